#!/bin/bash
# Config 4 on the two-part f16 sweep: round 4's form (the sweep splits the f32 history itself: DSPFX_FIR_PACKED=0) against the
# packed history ring written by the append pass (round 5), with 12 and 16 register slots in the sweep's chunk ring.
out=gpurun_out/r05_fir_packed_ab.txt
: > $out
run() {
  python bench.py --config cfg4 --steps 100 --warmup 20 --no-cpu-baseline --no-others --paced-seconds 0 2>>$out.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%.4f ms/step  kernel %s %.4f ms  frac_hbm %.3f  by step %.3f  step - kernel %.1f us' % (d['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r.get('frac_hbm', r['frac']), r['frac_by_step'], 1e3 * (d['ms_per_step'] - r['kernel_ms_avg'])))
" >> $out 2>&1
}
for i in 1 2 3; do
  echo "== round 4: split in the sweep (DSPFX_FIR_PACKED=0)" >> $out; DSPFX_FIR_PACKED=0 run
  echo "== packed history, 12 slots" >> $out; DSPFX_FIR_SLOTS=12 run
  echo "== packed history, 16 slots" >> $out; DSPFX_FIR_SLOTS=16 run
done
cat $out
