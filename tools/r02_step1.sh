#!/bin/bash
# Round 2, step 1: the driver's own command under rocprofv3 + plain, and the idle-transient experiment.
set -u
OUT=/root/repo/gpurun_out/r02a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o drv -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/drv_trace_bench.json 2>$OUT/drv_trace.err
tail -c 600 $OUT/drv_trace_bench.json
for i in 1 2; do
  python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/drv_plain_$i.json 2>$OUT/drv_plain_$i.err
done
rocprofv3 --kernel-trace --output-format csv -d $OUT/idle -o idle -- python3 /root/repo/tools/idle_transient.py > $OUT/idle.json 2>$OUT/idle.err
find $OUT -name "*.csv" | head; ls -la $OUT
