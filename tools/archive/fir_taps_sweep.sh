#!/bin/bash
# FIR kernel time against tap count, tiles per wave (DSPFX_FIR_NJT) and sweep form (DSPFX_FIR_SKEW), 262144 channels, B = 128.
OUT=/root/repo/gpurun_out/firsweep; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for T in ${FIR_SWEEP_TAPS:-64 256 384 512 1024 2048 4096}; do
  for njt in 2 4; do
    for skew in 1 0; do
      DSPFX_FIR_NJT=$njt DSPFX_FIR_SKEW=$skew python3 /root/repo/bench.py --no-cpu-baseline --no-others --config cfg4 --taps $T --steps 50 --warmup 40 2>/dev/null > $OUT/t${T}_n${njt}_s${skew}.json
      python3 - "$OUT/t${T}_n${njt}_s${skew}.json" $T $njt $skew <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')]
d=json.loads(lines[-1])
print('T %5s NJT %s skew %s   ms/step %.4f kern %.4f frac %.3f'%(sys.argv[2],sys.argv[3],sys.argv[4],d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))
PY
    done
  done
done
