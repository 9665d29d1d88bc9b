#!/bin/bash
# The driver's N = 2 command on a one-GPU box: both ranks on GPU 0 (DSPFX_BENCH_SHARE_GPU=1: torch's process group over gloo),
# the mix bus through the C ABI's mailbox communicator (DSPFX_BENCH_COMM=abi) -- a real two-process exchange per block, which RCCL
# refuses on one device.  Checks the multi-rank path end to end on hardware: launch, rendezvous, per-rank engines, tuning,
# communicator validation, the same-block global bus, exchange latency, the paced leg, MAX over ranks, one JSON line.
out=gpurun_out/r04_two_ranks_one_gpu.txt
export DSPFX_BENCH_SHARE_GPU=1 DSPFX_BENCH_COMM=abi HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $out 2> $out.err
echo "rc=$?" >> $out
tail -c 4000 $out; echo; grep -v "^W\|amdgpu.ids\|^$" $out.err | tail -15
