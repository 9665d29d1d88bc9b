#!/usr/bin/env python3
"""Per-step kernel time over three ring periods at D=24000: F = fast (<0.39 ms), S = slow.  Are the slow
steps the same ring groups on every pass?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
N, B, D = 1 << 20, 128, 24064      # 188 groups exactly (multiple of 128) so step k maps to group k % 188
os.environ["DSPFX_VARIANT"] = "static=1,f=8,cpl=2"
os.environ["DSPFX_RING_ROWSKEW"] = "0"
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
x = torch.empty(B * N, dtype=torch.float32, device=dev)
ys = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(2)]
for k in range(2):
    e = pkg.Engine(N, B, link_flags=3, tile_channels=256); e.set_chain(chains.chain5(pkg, D))
    G = D // 128
    e.profile_enable(8); e.profile_enable(0)
    if k == 0: e.fill_noise(x, B, 0)
    rows = []
    for y in (ys[0], ys[0], ys[1]):
        ts = []
        for s in range(G):
            e.profile_enable(1); e.process(x, out=y, n_frames=B, stream=stream)
            torch.cuda.synchronize(); e.profile_enable(0)
            ms, n, _ = e.profile_read(); ts.append(ms)
        rows.append(ts)
    for i, ts in enumerate(rows):
        print("engine %d pass %d (out buffer %d): " % (k, i, 0 if i < 2 else 1) + "".join("S" if t > 0.39 else "F" for t in ts) + "  mean %.4f" % (sum(ts) / len(ts)))
    same = sum((a > 0.39) == (b > 0.39) for a, b in zip(rows[0], rows[1]))
    other = sum((a > 0.39) == (b > 0.39) for a, b in zip(rows[1], rows[2]))
    print("   same groups slow on pass 0 and 1: %d/%d;  pass 1 (buffer 0) vs pass 2 (buffer 1): %d/%d" % (same, G, other, G))
    e.close()
