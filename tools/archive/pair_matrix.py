#!/usr/bin/env python3
"""Kernel time for every (input buffer, output buffer) pair, for a pure copy chain and for the
delay-only chain (ring fixed): is the fast/slow mode a property of stream PAIRS?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
N, B, D = 1 << 20, 128, 4096
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
K = 5
xs = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(K)]
ys = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(K)]
def timeit(e, x, y, steps=12):
    for _ in range(2): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(1)
    for _ in range(steps): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(0)
    ms, n, _ = e.profile_read(); return ms / n
for name, chain, lf in (("copy (c2 static)", [], 0), ("delay-only (dyn)", [pkg.Reverb(delay_samples=D, decay=0.5)], 0)):
    os.environ["DSPFX_VARIANT"] = "static=1,f=8,cpl=2"
    e = pkg.Engine(N, B, link_flags=lf, tile_channels=256); e.set_chain(chain)
    e.profile_enable(64); e.profile_enable(0); e.fill_noise(xs[0], B, 0)
    for t in xs[1:]: t.copy_(xs[0])
    for _ in range(D // B + 2): e.process(xs[0], out=ys[0], n_frames=B, stream=stream)
    print("==", name, e.describe().strip().splitlines()[1])
    for i, x in enumerate(xs):
        print("  x%d: " % i + " ".join("%.4f" % timeit(e, x, y) for y in ys))
