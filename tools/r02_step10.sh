#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r02j; mkdir -p $OUT
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
  python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/drv_plain_$i.json 2>$OUT/drv_plain_$i.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o drv -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/drv_trace_bench.json 2>$OUT/drv_trace.err
for f in drv_plain_1 drv_plain_2 drv_trace_bench; do python3 - $OUT/$f.json <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
d=json.loads(lines[-1])
print(sys.argv[1].split('/')[-1], 'ms/step %.4f kern %.4f frac %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))
for k,o in (d.get('other_configs') or {}).items():
    print('  ',k, o.get('error') or 'ms/step %.4f kern %.4f frac %.3f %s'%(o['ms_per_step'],o['roofline']['kernel_ms_avg'],o['roofline']['frac'],o['plan'][0][:100]))
PY
done
