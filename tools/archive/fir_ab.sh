#!/bin/bash
# Same-box A/B of the FIR sweep: the shipped library against another build (DSPFX_LIB), config 4, alternating.
OUT=/root/repo/gpurun_out/firab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-others --config cfg4 --steps 50 --warmup 20"
for v in "" ${FIR_AB_LIBS:-rowmajor} "" ${FIR_AB_LIBS:-rowmajor}; do
  if [ -n "$v" ]; then export DSPFX_LIB=/root/repo/dsp-stuff_amd/csrc/libdspfx_$v.so; else unset DSPFX_LIB; fi
  $B > $OUT/cfg4_${v:-base}.json 2>$OUT/cfg4_${v:-base}.err
  python3 - "$OUT/cfg4_${v:-base}.json" <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
d=json.loads(lines[-1])
print(sys.argv[1].split('/')[-1], 'ms/step %.4f kern %.4f frac %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))
PY
done
