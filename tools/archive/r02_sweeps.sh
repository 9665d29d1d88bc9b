#!/bin/bash
OUT=/root/repo/gpurun_out/r02_sweeps; mkdir -p $OUT
cd /root/repo
timeout 900 python tools/chain_sweep.py ${SEED_CHAIN:-7000} 300 > $OUT/chain.log 2>&1; tail -2 $OUT/chain.log
timeout 900 python tools/graph_sweep.py ${SEED_GRAPH:-4000} 60 > $OUT/graph.log 2>&1; tail -3 $OUT/graph.log
timeout 1200 python tools/region_sweep.py ${SEED_REGION:-3000} 40 > $OUT/region.log 2>&1; tail -3 $OUT/region.log
