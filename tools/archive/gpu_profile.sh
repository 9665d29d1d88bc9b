#!/bin/bash
# rocprofv3 evidence for the headline bench command: kernel-trace stats + PMC HBM bytes (separate passes).
set -u
R=${1:-r01}
OUT=/root/repo/gpurun_out/prof_$R; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --steps 100 --warmup 200 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o cfg5 -- $B > $OUT/trace_bench.json 2>$OUT/trace.err
tail -1 $OUT/trace_bench.json | cut -c1-200
# PMC passes (counters only, with kernel-trace so dispatches are named)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o cfg5 -- $B > $OUT/pmc_${c}_bench.json 2>$OUT/pmc_$c.err
  # calibration: empty chain through the same kernel structure and access width = exactly 4 B read + 4 B written per sample
  DSPFX_VARIANT="static=1,f=8,cpl=2" rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_copy -o copy -- $B --config copy --link-flags 0 --no-mix > $OUT/pmc_${c}_copy_bench.json 2>$OUT/pmc_${c}_copy.err
done
find $OUT -name "*.csv" | head -30
