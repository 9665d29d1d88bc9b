#!/bin/bash
# Same-block mix bus A/B on config 5's shard: in-kernel pipeline (bus two calls late) vs the bus of the same block inside the
# launch (mix_tail) vs the stand-alone reduction kernels behind the launch.  Each line: bench.py's own JSON.
out=gpurun_out/r03_bus_ab.txt
: > $out
for mode in pipe inline; do
  for tail in 1 0; do
    [ $mode = pipe ] && [ $tail = 0 ] && continue
    echo "== DSPFX_BENCH_MIX=$mode DSPFX_MIX_TAIL=$tail" >> $out
    DSPFX_BENCH_MIX=$mode DSPFX_MIX_TAIL=$tail python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-others 2>>$out.err | python tools/show_bench.py >> $out 2>&1
  done
done
cat $out
