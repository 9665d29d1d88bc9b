#!/usr/bin/env python3
"""Which allocation decides the fast/slow mode of the chain kernel?  (a) one engine, several output
buffers; (b) several engines, one output buffer; (c) several input buffers."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
os.environ["DSPFX_VARIANT"] = "static=1,f=8,cpl=2"
N, B, D = 1 << 20, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream

def mk_engine():
    e = pkg.Engine(N, B, link_flags=3, tile_channels=256)
    e.set_chain(chains.chain5(pkg, D))
    e.profile_enable(64); e.profile_enable(0)
    return e

def timeit(e, x, y, steps=30):
    for _ in range(3): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize()
    e.profile_enable(1)
    for _ in range(steps): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(0)
    ms, n, _ = e.profile_read()
    return ms / n

x = torch.empty(B * N, dtype=torch.float32, device=dev)
e0 = mk_engine(); e0.fill_noise(x, B, 0)
ys = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(6)]
for _ in range(D // B + 2): e0.process(x, out=ys[0], n_frames=B, stream=stream)
print("(a) one engine, 6 output buffers :", " ".join("%.4f" % timeit(e0, x, y) for y in ys))
xs = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(4)]
for t in xs: t.copy_(x)
print("(c) one engine, 4 input buffers  :", " ".join("%.4f" % timeit(e0, t, ys[0]) for t in xs))
engs = [mk_engine() for _ in range(5)]
for e in engs:
    for _ in range(D // B + 2): e.process(x, out=ys[0], n_frames=B, stream=stream)
print("(b) 5 engines, one in/out buffer :", " ".join("%.4f" % timeit(e, x, ys[0]) for e in engs))
print("(b') same engines, other out buf :", " ".join("%.4f" % timeit(e, x, ys[3]) for e in engs))
