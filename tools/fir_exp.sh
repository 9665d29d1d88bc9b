#!/bin/bash
# What the FIR sweep's matrix pipe delivers when one of its feeds is free (experiment builds, wrong results by design).
OUT=/root/repo/gpurun_out/firexp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --config cfg4 --steps 50 --warmup 20"
for v in "" firexp1 firexp2 firexp3 ""; do
  if [ -n "$v" ]; then export DSPFX_LIB=/root/repo/dsp-stuff_amd/csrc/libdspfx_$v.so; else unset DSPFX_LIB; fi
  $B > $OUT/cfg4_${v:-base}.json 2>$OUT/cfg4_${v:-base}.err
  python3 - "$OUT/cfg4_${v:-base}.json" <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
d=json.loads(lines[-1])
print(sys.argv[1].split('/')[-1], 'ms/step %.4f kern %.4f frac %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))
PY
done
