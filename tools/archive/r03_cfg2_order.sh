#!/bin/bash
# Config 2 inside the driver's command: timed before the headline config (the default) against after it (DSPFX_BENCH_EARLY= ,
# the order up to round 3), two runs each on one box.  The chip runs this light kernel 2-4 % slower for ~3 s after the large configs.
out=gpurun_out/r03_cfg2_order.txt
: > $out
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  order', d.get('timed_order'), ' headline %.4f ms  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))
for k,v in (d.get('other_configs') or {}).items():
    print('  %-9s step %8.2f us  kernel %8.2f us  frac %.3f  settle %s' % (k, v['ms_per_step']*1e3, v['roofline']['kernel_ms_avg']*1e3, v['roofline']['frac'], v.get('settle')))
"; }
for rep in 1 2; do
  echo "== default order (run $rep)" >> $out
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --paced-seconds 0 2>>$out.err | show >> $out
  echo "== config 2 after the headline config and config 3 (run $rep)" >> $out
  DSPFX_BENCH_EARLY= python bench.py --steps 20 --warmup 5 --no-cpu-baseline --paced-seconds 0 2>>$out.err | show >> $out
done
cat $out
