#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace --stats) of small-N configurations (cfg2 chain: gain>biquad>delay)
# for a few kernel variants: at these sizes the bench's wall-clock step is launch/host bound, the trace is not.
# usage: small_n_profile.sh "<channels list>" "<variant list>"
set -u
OUT=/root/repo/gpurun_out/small_n; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CH=${1:-"65536"}
VARS=${2:-"f=8,cpl=2 f=8,cpl=1 f=16,cpl=1 f=32,cpl=1"}
for n in $CH; do
for v in $VARS; do
  tag=n${n}_$(echo $v | tr -d '=,')
  DSPFX_VARIANT="$v" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o cfg2 -- python3 /root/repo/bench.py --config cfg2 --channels $n --steps 100 --warmup 200 --no-cpu-baseline --probe 1 > $OUT/$tag.json 2>$OUT/$tag.err
  echo "== N=$n $v: $(grep chain_kernel $OUT/$tag/cfg2_kernel_stats.csv | awk -F, '{print "avg_ns", $(NF-4), "min", $(NF-2), "max", $(NF-1)}')"
done
done
