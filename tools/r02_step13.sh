#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r02m; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --config cfg4 --steps 50 --warmup 20"
for sg in 0 1 2 4 9 0; do
DSPFX_FIR_STAGGER=$sg $B > $OUT/cfg4_sg$sg.json 2>$OUT/cfg4_sg$sg.err
python3 - "$OUT/cfg4_sg$sg.json" <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
d=json.loads(lines[-1])
print(sys.argv[1].split('/')[-1], 'ms/step %.4f kern %.4f frac %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))
PY
done
