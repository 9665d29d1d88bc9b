#!/bin/bash
# config 4 with the history append fused into the sweep (default) and as a pass of its own (DSPFX_FIR_FUSE=0), f32 and split sweeps
out=gpurun_out/r03_fir_ab.txt
: > $out
for split in 0 1; do for fuse in 1 0; do
  echo "== DSPFX_FIR_SPLIT=$split DSPFX_FIR_FUSE=$fuse" >> $out
  DSPFX_FIR_SPLIT=$split DSPFX_FIR_FUSE=$fuse python bench.py --config cfg4 --steps 40 --warmup 10 --no-cpu-baseline --no-others --paced-seconds 0 2>>$out.err | python tools/show_bench.py >> $out 2>&1
done; done
cat $out
