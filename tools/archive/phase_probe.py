#!/usr/bin/env python3
"""Does the ring phase (pos mod 128 relative to the block's frame index) decide the mode?  Per-step kernel
times for delay lengths that keep the phase at 0 (multiples of 128) or make it alternate (odd multiples of 64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
N, B = 1 << 20, 128
os.environ["DSPFX_VARIANT"] = "static=1,f=8,cpl=2"
os.environ["DSPFX_RING_ROWSKEW"] = "0"
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
x = torch.empty(B * N, dtype=torch.float32, device=dev)
y = torch.empty(B * N, dtype=torch.float32, device=dev)
first = True
for D in (4096, 4160, 4224, 4128, 24000):
    for k in range(2):
        e = pkg.Engine(N, B, link_flags=3, tile_channels=256); e.set_chain(chains.chain5(pkg, D))
        steps = 2 * (D // B + 1) + 4
        e.profile_enable(steps + 8); e.profile_enable(0)
        if first: e.fill_noise(x, B, 0); first = False
        for _ in range(4): e.process(x, out=y, n_frames=B, stream=stream)
        # per-step timing over two ring periods: one profile read per step (slow but exact)
        ts = []
        torch.cuda.synchronize()
        for s in range(steps):
            e.profile_enable(1)
            e.process(x, out=y, n_frames=B, stream=stream)
            torch.cuda.synchronize(); e.profile_enable(0)
            ms, n, _ = e.profile_read(); ts.append(ms)
        pos = [((4 + s) * B) % D % 128 for s in range(steps)]
        by = {}
        for p, t in zip(pos, ts): by.setdefault(p, []).append(t)
        print("D=%5d engine %d: " % (D, k) + "  ".join("phase %3d: n=%d mean %.4f min %.4f max %.4f" % (p, len(v), sum(v) / len(v), min(v), max(v)) for p, v in sorted(by.items())))
        e.close()
