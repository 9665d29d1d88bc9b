#!/bin/bash
# Run on the GPU box (via gpurun): parity tests, variant sweep on the headline workload, rocprof kernel trace.
set -u
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /root/repo
python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
: > $OUT/sweep.jsonl
for v in "static=1,f=8,cpl=1" "static=1,f=8,cpl=2" "static=1,f=8,cpl=4" "static=1,f=16,cpl=1" "static=1,f=16,cpl=2" "static=1,f=4,cpl=4" "static=0,f=8" "static=0,f=4" "static=0,f=16"; do
  echo "== $v" | tee -a $OUT/sweep.log
  DSPFX_VARIANT="$v" python bench.py --steps 100 --warmup 200 --no-cpu-baseline 2>>$OUT/sweep.log | tee -a $OUT/sweep.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    r = d['roofline']; print(r['kernel'], 'ms/step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % r['kernel_ms_avg'], 'GB/s %.0f' % r['achieved'], 'frac %.3f' % r['frac'], 'samples/s %.3e' % d['value'])
"
done
# no-mix and no-link variants for reference
python bench.py --steps 100 --warmup 200 --no-cpu-baseline --no-mix 2>>$OUT/sweep.log | tee -a $OUT/sweep.jsonl | cut -c1-400
python bench.py --steps 100 --warmup 200 --no-cpu-baseline --link-flags 0 2>>$OUT/sweep.log | tee -a $OUT/sweep.jsonl | cut -c1-400
