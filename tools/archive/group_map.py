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
N, B, D = 1 << 20, 128, 24064      # 188 groups exactly
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
    if os.environ.get("DSPFX_DESCRIBE_GROUPS"):
        addrs = [int(l.split("@")[1], 16) for l in e.describe().splitlines() if l.strip().startswith("group")]
        print("   x=0x%x y0=0x%x y1=0x%x" % (x.data_ptr(), ys[0].data_ptr(), ys[1].data_ptr()))
        runs, prev = [], None
        for g, (a, t) in enumerate(zip(addrs, rows[0])):
            m = "S" if t > 0.39 else "F"
            if m != prev: runs.append([m, g, a, a]); prev = m
            runs[-1][3] = a
        for m, g, a0, a1 in runs: print("   %s from group %3d: VA 0x%x .. 0x%x  (GiB %.2f .. %.2f)" % (m, g, a0, a1, a0 / 2**30, a1 / 2**30))
    same = sum((a > 0.39) == (b > 0.39) for a, b in zip(rows[0], rows[1]))
    other = sum((a > 0.39) == (b > 0.39) for a, b in zip(rows[1], rows[2]))
    print("   same groups slow on pass 0 and 1: %d/%d;  pass 1 (buffer 0) vs pass 2 (buffer 1): %d/%d" % (same, G, other, G))
    e.close()
