#!/bin/bash
# The driver's own command under rocprofv3 --kernel-trace --stats, twice un-profiled beside it; per-launch durations of every
# timed region (tools/trace_phases.py).  Outputs under gpurun_out/drivercmd/ (copy the summaries to profiles/).
OUT=/root/repo/gpurun_out/drivercmd; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_run1.json 2>$OUT/bench_run1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o drv -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2>$OUT/rocprof.err
python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_run2.json 2>$OUT/bench_run2.err
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 /root/repo/tools/trace_phases.py $T $OUT/bench_under_rocprof.json $OUT/timed_regions.json > $OUT/timed_regions.txt 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
tail -30 $OUT/timed_regions.txt
