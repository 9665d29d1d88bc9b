#!/bin/bash
# headline config, same-block bus: frame-major [B][N] against channel-tiled layouts, fresh process each
out=gpurun_out/r03_cfg5_layout_ab.txt
: > $out
for rep in 1 2; do
  for tile in ${TILES:-0 128 256 512 1024 4096}; do
    echo "== tile $tile run $rep" >> $out
    python bench.py --config cfg5 --tile $tile --steps 200 --warmup 20 --no-cpu-baseline --no-others --paced-seconds 0 2>>$out.err | python tools/show_bench.py | cut -c1-150 >> $out 2>&1
  done
done
cat $out
