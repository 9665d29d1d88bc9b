#!/usr/bin/env python3
"""A wider sweep than the test suite runs: random DAGs (tests/graphs.py random_dag) as one generated kernel
 (a) of the exact-arithmetic kinds against the oracle's node-by-node evaluation (ulp), 
 (b) of every fusable kind against the run-by-run evaluation on the GPU (bits),
 (c) graphs cut into a series of small kernels (segment_plan with max_nodes 2..6) against the one-kernel result (bits).
usage: graph_sweep.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
from __graft_entry__ import load_package
E = load_package()
from dsp_stuff_amd import graph as G
import graphs, graph_eval, oracle as O
from chains import ulp_diff
def run(s0=1000, cnt=100, budget_s=None):
    """see the module docstring; budget_s stops the sweep early (the test suite's time box); returns the counters."""
    t0 = time.time()
    worst, nonzero, bad, cuts, ran = 0, 0, [], 0, 0
    for seed in range(s0, s0 + cnt):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        ran += 1
        n = 4 + seed % 13                       # 4..16 nodes
        N, B, nf = 64, 128, 384
        tile = 64 if seed % 2 else 0
        text = graphs.random_dag(seed, n)
        x = O.noise(0x5EED1000 + seed, np.arange(N), np.arange(nf))
        ge = G.GraphEngine(text, N, B, tile_channels=tile, fused=True)
        got = np.empty_like(x)
        for f0 in range(0, nf, B):
            y = ge.process(torch.from_numpy(E.to_layout(x[f0:f0 + B], tile)).cuda(), B)
            torch.cuda.synchronize()
            got[f0:f0 + B] = E.from_layout(y.cpu().numpy(), B, N, tile)
        ref = graph_eval.run_graph(ge.g, x[:, :8])
        ge.close()
        if not np.isfinite(ref).all():
            continue
        d = int(ulp_diff(got[:, :8], ref).max())
        worst = max(worst, d)
        nonzero += int(np.abs(ref).max() > 0)
        if d > 1:
            bad.append(("oracle", seed, d))
        # (b) every kind, fused vs run by run
        text = graphs.random_dag(seed, n, libm=True)
        N2 = 1024
        a = G.GraphEngine(text, N2, B, fused=True)
        b = G.GraphEngine(text, N2, B, fused=False)
        xd = torch.empty(B * N2, dtype=torch.float32, device="cuda")
        for k in range(3):
            a.util.fill_noise(xd, B, k * B, 0x5EED2000 + seed)
            ya = a.process(xd, B).clone()
            yb = b.process(xd, B)
            torch.cuda.synchronize()
            if not torch.equal(ya.view(torch.int32), yb.view(torch.int32)):
                bad.append(("runs", seed, k))
                break
        a.close(); b.close()
        # (c) the series planner: cut as if a kernel held only m nodes, against the one-kernel result
        for text in (graphs.long_rig(seed, 5, dry_mix=bool(seed % 2)), graphs.random_dag(seed, 8)):
            for m in (2, 3, 4, 6):
                steps = G.segment_plan(G.Graph(text), m)
                if steps is None or len(steps) < 2:
                    continue
                cuts += 1
                a = G.GraphEngine(text, 256, B, max_nodes=m)
                b = G.GraphEngine(text, 256, B, fused=True)
                xs = torch.empty(B * 256, dtype=torch.float32, device="cuda")
                for k in range(3):
                    b.util.fill_noise(xs, B, k * B, 0x5EED3000 + seed)
                    ya = a.process(xs, B).clone()
                    yb = b.process(xs, B)
                    torch.cuda.synchronize()
                    if not torch.equal(ya.view(torch.int32), yb.view(torch.int32)):
                        bad.append(("segments", seed, m, k))
                        break
                a.close(); b.close()
    print("series plans checked: %d" % cuts)
    print("seeds %d..%d: worst ulp vs oracle %d (%d graphs with non-zero output), mismatches %s, %.0f s" % (s0, s0 + cnt - 1, worst, nonzero, bad, time.time() - t0))
    return dict(ran=ran, worst=worst, nonzero=nonzero, cuts=cuts, bad=bad, seconds=time.time() - t0)


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 100)
