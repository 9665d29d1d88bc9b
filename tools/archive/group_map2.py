#!/usr/bin/env python3
"""Per-ring-group S/F map of one engine for different (input, output) buffer combinations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
N, B, D = 1 << 20, 128, 24064
os.environ["DSPFX_RING_TUNE"] = "0"
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
xs = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(3)]
ys = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(3)]
G = D // 128
for name, var, mk in (("delay-only", "static=0,f=8", lambda: [pkg.Reverb(delay_samples=D, decay=0.5)]),
                      ("5-node", "static=1,f=8,cpl=2", lambda: chains.chain5(pkg, D))):
    os.environ["DSPFX_VARIANT"] = var
    e = pkg.Engine(N, B, link_flags=3, tile_channels=256); e.set_chain(mk())
    e.profile_enable(8); e.profile_enable(0); e.fill_noise(xs[0], B, 0)
    for t in xs[1:]: t.copy_(xs[0])
    maps = {}
    for (xi, yi) in ((0, 0), (1, 0), (2, 0), (0, 1), (0, 2), (0, 0)):
        ts = []
        for s in range(G):
            e.profile_enable(1); e.process(xs[xi], out=ys[yi], n_frames=B, stream=stream)
            torch.cuda.synchronize(); e.profile_enable(0)
            ms, n, _ = e.profile_read(); ts.append(ms)
        lo = sorted(ts)[G // 8]
        m = "".join("S" if t > 1.1 * lo else "F" for t in ts)
        maps[(xi, yi)] = m
        print("%-10s x%d y%d fast %.4f mean %.4f: %s" % (name, xi, yi, lo, sum(ts) / G, m))
    e.close()
