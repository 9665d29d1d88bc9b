#!/usr/bin/env python3
"""(engine, output buffer) time matrix for the 5-node chain: static c2, static c1, interpreter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
N, B, D = 1 << 20, 128, 4096
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
x = torch.empty(B * N, dtype=torch.float32, device=dev)
ys = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(6)]
def timeit(e, x, y, steps=12):
    for _ in range(2): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(1)
    for _ in range(steps): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(0)
    ms, n, _ = e.profile_read(); return ms / n
first = True
for var in ("static=1,f=8,cpl=2", "static=1,f=8,cpl=1", "static=0,f=16"):
    os.environ["DSPFX_VARIANT"] = var
    print("==", var, "tile", tile)
    for k in range(3):
        e = pkg.Engine(N, B, link_flags=3, tile_channels=tile); e.set_chain(chains.chain5(pkg, D))
        e.profile_enable(64); e.profile_enable(0)
        if first: e.fill_noise(x, B, 0); first = False
        for _ in range(D // B + 2): e.process(x, out=ys[0], n_frames=B, stream=stream)
        print("  e%d: " % k + " ".join("%.4f" % timeit(e, x, y) for y in ys))
        e.close()
