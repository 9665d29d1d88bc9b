#!/bin/bash
# config 2 (65536 channels, 3-node chain, time-sliced kernel) with the three bus forms and without a bus
out=gpurun_out/r03_cfg2_ab.txt
: > $out
for mode in inline pipe nomix; do
  echo "== $mode" >> $out
  if [ $mode = nomix ]; then extra="--no-mix"; else extra=""; fi
  DSPFX_BENCH_BUS_ALL=1 DSPFX_BENCH_MIX=$mode python bench.py --config cfg2 --steps 200 --warmup 50 --no-cpu-baseline --no-others --paced-seconds 0 $extra 2>>$out.err | python tools/show_bench.py >> $out 2>&1
done
cat $out
