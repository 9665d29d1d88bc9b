#!/bin/bash
# Round 6, VERDICT r05 #4: what could folding the fix-ups (a) and fusing the append (b) buy at config 4?  Timing-only bounds from an experiment
# build (make libdspfx_exp.so XFLAGS=-DDSPFX_FIR_EXP): DSPFX_FIR_EXP=1 skips the append launch (results wrong: an upper bound for a perfectly
# hidden append), =2 skips the two usually-empty fix-up launches, =3 both.
out=gpurun_out/r06_fir_fusion_bound.txt
: > $out
for e in 0 2 1 3 0; do
  DSPFX_LIB=$PWD/dsp-stuff_amd/csrc/libdspfx_exp.so DSPFX_FIR_EXP=$e python3 bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('DSPFX_FIR_EXP=$e  ms_per_step %.4f  kernel %s %.4f ms  frac %.4f  frac_by_step %.4f' % (d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], r['frac_by_step']))" >> $out
done
cat $out
