#!/bin/bash
# config 2 without a bus (as BASELINE words it): frame-major [B][N] against channel-tiled layouts, fresh process each
out=gpurun_out/r03_cfg2_layout_ab.txt
: > $out
for rep in 1 2 3; do
  for tile in ${TILES:-0 128 256 1024 4096 16384}; do
    echo "== tile $tile run $rep" >> $out
    python bench.py --config cfg2 --tile $tile --steps 200 --warmup 50 --no-cpu-baseline --no-others --paced-seconds 0 2>>$out.err | python tools/show_bench.py | cut -c1-90 >> $out 2>&1
  done
done
cat $out
