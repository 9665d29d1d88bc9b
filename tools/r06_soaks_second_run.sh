#!/bin/bash
out=gpurun_out/r06_soaks_second_run.txt
{
echo "second run, other seeds:"
echo "tools/r05_ring_soak.py 50000 1500:";  python3 tools/r05_ring_soak.py 50000 1500 | tail -1
echo "tools/r04_store_soak.py 51000 200:"; python3 tools/r04_store_soak.py 51000 200 | tail -1
echo "tools/r04_ragged_calls_soak.py 52000 200:"; python3 tools/r04_ragged_calls_soak.py 52000 200 | tail -1
echo "tools/chain_sweep.py 53000 250:"; python3 tools/chain_sweep.py 53000 250 | tail -1
echo "tools/graph_sweep.py 54000 80:"; python3 tools/graph_sweep.py 54000 80 | tail -2
echo "tools/soak.py 300:"; python3 tools/soak.py 300 | tail -1
} > $out 2>&1
grep -v amdgpu.ids $out
