#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/r02l; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --config cfg4 --steps 50 --warmup 20"
$B > $OUT/cfg4_njt2.json 2>$OUT/cfg4_njt2.err
DSPFX_FIR_NJT=4 $B > $OUT/cfg4_njt4.json 2>$OUT/cfg4_njt4.err
$B --taps 1024 > $OUT/cfg4_t1024_njt2.json 2>/dev/null
$B --taps 256 > $OUT/cfg4_t256_njt2.json 2>/dev/null
for f in $OUT/cfg4_*.json; do python3 - "$f" <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
if not lines: print(sys.argv[1],'NO LINE'); sys.exit()
d=json.loads(lines[-1])
print(sys.argv[1].split('/')[-1], 'ms/step %.4f kern %.4f frac %.3f settle %s'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['config']['settle']['ms_per_step']))
PY
done
