#!/bin/bash
# In-launch bus tail: slices x batch variants (experiment builds: make -C dsp-stuff_amd/csrc libdspfx_tail_<slices>_<batch>.so)
out=gpurun_out/r03_tail_ab.txt
: > $out
for rep in 1 2; do
for v in "" _tail_32_32 _tail_128_32 _tail_64_16; do
  lib=$PWD/dsp-stuff_amd/csrc/libdspfx$v.so
  echo "== rep $rep  ${v:-default(64 slices, batch 32)}" >> $out
  DSPFX_LIB=$lib DSPFX_BENCH_MIX=inline python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-others --paced-seconds 0 2>/dev/null | python tools/show_bench.py >> $out 2>&1
done; done
cat $out
