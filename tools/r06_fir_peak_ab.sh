#!/bin/bash
# A/B on ONE box: config 4 on the current library and on one with round 5's fir_kernels.hip (peak decision at the sweep's end)
for rep in 1 2 3 4 5 6; do
for lib in libdspfx.so libdspfx_oldfir.so libdspfx_prodec.so; do
  DSPFX_LIB=$PWD/dsp-stuff_amd/csrc/$lib python3 bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$lib  ms_per_step %.4f  kernel %s %.4f ms' % (d['ms_per_step'], r['kernel'], r['kernel_ms_avg']))"
done; done
