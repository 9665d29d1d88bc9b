#!/bin/bash
# Round 6: (1) the driver's N = 1 command; (2) `python3 bench.py --gpus 2` with NO launcher in front of it on a one-GPU box
# (DSPFX_BENCH_SHARE_GPU=1: both ranks on GPU 0, torch's process group over gloo, the bus through the C ABI's mailbox communicator):
# bench.py starts its own ranks and reports both forms of the exchange (scaling_forms).  The numbers of (2) mean nothing as
# throughput (two processes time-slice one chip); the run proves launch, rendezvous, exchange and the line.
mkdir -p gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd.json 2> gpurun_out/r06_bench_driver_cmd.err
echo "rc=$?"; tail -c 3000 gpurun_out/r06_bench_driver_cmd.json
DSPFX_BENCH_SHARE_GPU=1 DSPFX_BENCH_COMM=abi timeout 900 python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r06_two_ranks_self_launched.json 2> gpurun_out/r06_two_ranks_self_launched.err
echo "rc=$?"; cat gpurun_out/r06_two_ranks_self_launched.json; grep -v "^W\|amdgpu.ids\|^$\|Gloo" gpurun_out/r06_two_ranks_self_launched.err | tail -12
