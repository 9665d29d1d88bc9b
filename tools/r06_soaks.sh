#!/bin/bash
# Round 6: the long form of every soak on the final build (the time-boxed slices run in `pytest -m gpu`: tests/test_gpu_soak.py).
out=gpurun_out/r06_soaks.txt
{
echo "tools/r05_ring_soak.py 40000 600:";  python3 tools/r05_ring_soak.py 40000 600 | tail -1
echo "tools/r04_store_soak.py 41000 120:"; python3 tools/r04_store_soak.py 41000 120 | tail -1
echo "tools/r04_ragged_calls_soak.py 42000 100:"; python3 tools/r04_ragged_calls_soak.py 42000 100 | tail -1
echo "tools/r04_mailbox_soak.py 4 30000:"; python3 tools/r04_mailbox_soak.py 4 30000 | tail -5
echo "tools/r04_mailbox_soak.py 3 30000 (a world whose divisor's reciprocal is inexact: the checker now divides on the host):"; python3 tools/r04_mailbox_soak.py 3 30000 | tail -4
echo "tools/fir_soak.py 600:"; python3 tools/fir_soak.py 600 | tail -26
echo "tools/chain_sweep.py 43000 150:"; python3 tools/chain_sweep.py 43000 150 | tail -1
echo "tools/graph_sweep.py 44000 60:"; python3 tools/graph_sweep.py 44000 60 | tail -2
} > $out 2>&1
tail -50 $out
