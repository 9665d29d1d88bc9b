#!/bin/bash
# Fresh processes on one box: how often does an install land in the slow placement mode before tuning, what does the adaptive
# tuning cost, where does it end.  ms_before = the chain kernel against the real buffers before dspfx_tune_placement (no bus),
# ms_after = the settled step after it (same-block bus included: +5 us).
out=gpurun_out/r03_placement_incidence.txt
: > $out
for i in $(seq 1 ${1:-14}); do
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-others --paced-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['config']['placement_tuning']; r=d['roofline']
print('run $i  untuned %.4f ms (%.3f of peak)  tuning %.2f s  %s  tuned+bus %.4f ms (%.3f)' % (t['ms_before'], 16.5*(1<<20)*128/(t['ms_before']*1e-3)/8e12, t['seconds'], [p.split(', ')[1] for p in d['config']['plan'] if 'ring' in p][0], r['kernel_ms_avg'], r['frac']))" >> $out
done
cat $out
