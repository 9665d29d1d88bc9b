#!/bin/bash
# The driver's N = 2 command on a one-GPU box: both ranks on GPU 0, process group over gloo (DSPFX_BENCH_SHARE_GPU=1).  Checks that
# the multi-rank path runs end to end on hardware -- launch, rendezvous, per-rank engines, tuning, batched bus + collective, MAX over
# ranks, one JSON line -- not its speed.
out=gpurun_out/r03_two_ranks_one_gpu.txt
export DSPFX_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $out 2> $out.err
echo "rc=$?" >> $out
tail -c 3000 $out; echo; grep -v "^W\|amdgpu.ids\|^$" $out.err | tail -15
