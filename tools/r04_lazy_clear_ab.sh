#!/bin/bash
# What does the delay ring's lazy clear (SlotArgs::zero_rows: one scalar compare per chunk and delay node) cost the steady state?
# Same box, alternating: the shipped library against libdspfx_exp.so built with -DDSPFX_NO_LAZY_CLEAR (make libdspfx_exp.so XFLAGS=...).
out=gpurun_out/r04_lazy_clear_ab.txt
: > $out
for i in 1 2 3; do
  for lib in libdspfx.so libdspfx_exp.so; do
    echo "== $lib cfg5" >> $out
    DSPFX_LIB=$PWD/dsp-stuff_amd/csrc/$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-others --paced-seconds 0 2>>$out.err | python tools/show_bench.py >> $out 2>&1
    echo "== $lib cfg2" >> $out
    DSPFX_LIB=$PWD/dsp-stuff_amd/csrc/$lib python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline --no-others --paced-seconds 0 2>>$out.err | python tools/show_bench.py >> $out 2>&1
  done
done
cat $out
