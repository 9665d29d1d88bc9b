#!/bin/bash
# VERDICT r04 #3a, the cheap half: before building a packed {f16 hi, f16 lo} history ring that the append pass would write, how
# much can the config-4 sweep gain from not splitting (and peak-scanning) every sample 33 times?  libdspfx_exp.so built with
# -DDSPFX_HALF_NOSPLIT feeds the f32 bits to the matrix pipe unsplit (garbage results, the packed-ring sweep's instruction
# stream) -- an upper bound.  Alternating runs on one box; rocprofv3 kernel trace of each for the kernel's own time.
out=gpurun_out/r05_fir_nosplit_ab.txt
: > $out
run() {
  python bench.py --config cfg4 --steps 100 --warmup 20 --no-cpu-baseline --no-others --paced-seconds 0 2>>$out.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%.4f ms/step  kernel %s %.4f ms  frac_hbm %.3f  by step %.3f' % (d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r.get('frac_hbm', r['frac']), r['frac_by_step']))
" >> $out 2>&1
}
for i in 1 2 3; do
  echo "== default (split + peak scan in the sweep)" >> $out; run
  echo "== nosplit experiment build" >> $out; DSPFX_LIB=$PWD/dsp-stuff_amd/csrc/libdspfx_exp.so run
done
cat $out
