#!/bin/bash
# Round 2, step 2: GPU suite, then the driver's command plain x2 and under rocprofv3.
set -u
OUT=/root/repo/gpurun_out/r02b; mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
  python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/drv_plain_$i.json 2>$OUT/drv_plain_$i.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o drv -- python3 /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/drv_trace_bench.json 2>$OUT/drv_trace.err
ls $OUT
