#!/usr/bin/env python3
"""Distribution of kernel times over K fresh engines, ring row-skew off vs on (allocation luck is the
confounder, so look at the whole distribution)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
N, B, D, K = 1 << 20, 128, 4096, 8
os.environ["DSPFX_VARIANT"] = "static=1,f=8,cpl=2"
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
x = torch.empty(B * N, dtype=torch.float32, device=dev)
ys = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(3)]
def timeit(e, y, steps=32):
    for _ in range(2): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(1)
    for _ in range(steps): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(0)
    ms, n, _ = e.profile_read(); return ms / n
first = True
for rnd in range(2):
    for skew in ("0", "1"):
        os.environ["DSPFX_RING_ROWSKEW"] = skew
        engs = []
        for k in range(K):
            e = pkg.Engine(N, B, link_flags=3, tile_channels=256); e.set_chain(chains.chain5(pkg, D))
            e.profile_enable(64); e.profile_enable(0)
            if first: e.fill_noise(x, B, 0); first = False
            for _ in range(D // B + 2): e.process(x, out=ys[0], n_frames=B, stream=stream)
            engs.append(e)
        t = [[timeit(e, y) for e in engs] for y in ys]
        flat = sorted(v for row in t for v in row)
        print("rowskew=%s: " % skew + " | ".join(" ".join("%.3f" % v for v in row) for row in t), " -> slow(>0.40): %d/%d" % (sum(v > 0.40 for v in flat), len(flat)))
        for e in engs: e.close()
