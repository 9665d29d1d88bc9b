#!/usr/bin/env python3
"""A wider sweep than the test suite runs, for the region plan (graph.py region_plan + dspfx_process_io): random DAGs of
17..40 nodes (tests/graphs.py random_dag) and smaller ones cut as if a kernel held only 3..8 nodes,
 (a) exact-arithmetic kinds against the oracle's node-by-node evaluation (ulp, a few channels),
 (b) every fusable kind against the run-by-run evaluation on the GPU (bits).
usage: region_sweep.py [first_seed] [count]"""
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
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = time.time()
worst, bad, regions, kernels = 0, [], 0, 0
for seed in range(s0, s0 + cnt):
    n = 17 + seed % 24
    N, B, nf = 64, 128, 384
    text = graphs.random_dag(seed, n)
    x = O.noise(0x5EED4000 + seed, np.arange(N), np.arange(nf))
    ge = G.GraphEngine(text, N, B, regions=True)
    regions += 1
    kernels += sum(1 for st in ge.regions if st["kind"] == "region")
    got = np.empty_like(x)
    for f0 in range(0, nf, B):
        y = ge.process(torch.from_numpy(x[f0:f0 + B]).cuda(), B)
        torch.cuda.synchronize()
        got[f0:f0 + B] = y.cpu().numpy().reshape(B, N)
    ref = graph_eval.run_graph(ge.g, x[:, :6])
    ge.close()
    if np.isfinite(ref).all():
        d = int(ulp_diff(got[:, :6], ref).max())
        worst = max(worst, d)
        if d > 1:
            bad.append(("oracle", seed, d))
    # (b) every kind, regions (also cut small) vs run by run
    for m in (None, 3 + seed % 6):
        text = graphs.random_dag(seed, n if m is None else 6 + seed % 11, libm=True)
        a = G.GraphEngine(text, 256, B, regions=True, max_nodes=m)
        b = G.GraphEngine(text, 256, B, fused=False)
        xs = torch.empty(B * 256, dtype=torch.float32, device="cuda")
        for k in range(3):
            b.util.fill_noise(xs, B, k * B, 0x5EED5000 + seed)
            ya = a.process(xs, B).clone()
            yb = b.process(xs, B)
            torch.cuda.synchronize()
            if not torch.equal(ya.view(torch.int32), yb.view(torch.int32)):
                bad.append(("runs", seed, m, k))
                break
        a.close(); b.close()
print("region plans %d..%d: %d graphs of 17..40 nodes in %d region kernels; worst ulp vs oracle %d, mismatches %s, %.0f s" % (
    s0, s0 + cnt - 1, regions, kernels, worst, bad, time.time() - t0))
