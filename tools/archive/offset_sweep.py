#!/usr/bin/env python3
"""Sweep the output buffer's byte offset against a fixed engine (block-major ring, D multiple of 128,
so ring-vs-out alignment is the same on every step) and print kernel time per offset."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
os.environ["DSPFX_VARIANT"] = "static=1,f=8,cpl=2"
N, B, D = 1 << 20, 128, 4096
step = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
count = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
e = pkg.Engine(N, B, link_flags=3, tile_channels=256)
e.set_chain(chains.chain5(pkg, D))
e.profile_enable(64); e.profile_enable(0)
x = torch.empty(B * N, dtype=torch.float32, device=dev)
e.fill_noise(x, B, 0)
big = torch.empty(B * N + step * count // 4 + 1024, dtype=torch.float32, device=dev)
for _ in range(D // B + 2): e.process(x, out=big[:B * N], n_frames=B, stream=stream)
res = []
for k in range(count):
    off = k * step // 4
    y = big[off:off + B * N]
    for _ in range(2): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize()
    e.profile_enable(1)
    for _ in range(12): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(0)
    ms, n, _ = e.profile_read()
    res.append(ms / n)
print("step %d bytes" % step)
for i in range(0, count, 16):
    print("%8d KB: " % (i * step // 1024) + " ".join("%.3f" % v for v in res[i:i + 16]))
