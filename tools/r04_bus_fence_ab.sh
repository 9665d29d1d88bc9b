#!/bin/bash
# The in-launch bus' hand-over: default (write-through rows + s_waitcnt + relaxed tickets + sc1 reads) against the fence build
# (libdspfx_busfence.so: plain rows, agent-scope release / acquire).  Same box, alternating; config 5 (same-block bus) and
# config 2 with a bus (DSPFX_BENCH_BUS_ALL=1).  Bit-identity of the two is checked by tests/test_gpu_threads.py.
out=gpurun_out/r04_bus_fence_ab.txt
: > $out
for i in 1 2 3; do
  for lib in libdspfx.so libdspfx_busfence.so; do
    echo "== $lib cfg5 (bus of the same block)" >> $out
    DSPFX_LIB=$PWD/dsp-stuff_amd/csrc/$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-others --paced-seconds 0 2>>$out.err | python tools/show_bench.py >> $out 2>&1
    echo "== $lib cfg2 + bus" >> $out
    DSPFX_BENCH_BUS_ALL=1 DSPFX_LIB=$PWD/dsp-stuff_amd/csrc/$lib python bench.py --config cfg2 --steps 200 --warmup 20 --no-cpu-baseline --no-others --paced-seconds 0 2>>$out.err | python tools/show_bench.py >> $out 2>&1
  done
done
cat $out
