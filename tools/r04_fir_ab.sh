#!/bin/bash
# Config 4 (262144 channels x 4096 taps, B = 128) on the three steady-state sweeps, same box, alternating:
#   half  = fir_half_kernel (f16 hi + lo, 3 products / term, + the bf16 x 3 second pass over listed tiles: the default)
#   split = fir_split_kernel (bf16 x 3, 6 products / term: DSPFX_FIR_HALF=0)
#   f32   = fir_skew_kernel (v_mfma_f32_32x32x2_f32: DSPFX_FIR_SPLIT=0)
out=gpurun_out/r04_fir_ab.txt
: > $out
for i in 1 2; do
  for mode in half split f32; do
    echo "== $mode" >> $out
    env=""
    [ $mode = split ] && export DSPFX_FIR_HALF=0
    [ $mode = f32 ] && export DSPFX_FIR_SPLIT=0
    python bench.py --config cfg4 --steps 100 --warmup 20 --no-cpu-baseline --no-others --paced-seconds 0 2>>$out.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%.4f ms/step  kernel %s %.4f ms  bound %s frac %.3f  by step %.3f  algorithmic %.1f TFLOP/s  %s' % (d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['bound'], r['frac'], r['frac_by_step'], r.get('algorithmic_tflops', r['achieved']), {k: round(v, 3) for k, v in r.items() if k in ('frac_hbm', 'frac_f16_mfma')}))
" >> $out 2>&1
    unset DSPFX_FIR_HALF DSPFX_FIR_SPLIT
  done
done
cat $out
