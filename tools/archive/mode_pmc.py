#!/usr/bin/env python3
"""Workload for the fast/slow-mode PMC study: K engines x M launches of the 5-node chain (run under
rocprofv3 --pmc ...; tools/mode_pmc_report.py then splits dispatches into fast / slow by duration)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
os.environ["DSPFX_VARIANT"] = "static=1,f=8,cpl=2"
N, B, D, K, M = 1 << 20, 128, 4096, 6, 48
dev = torch.device("cuda", 0)
x = torch.empty(B * N, dtype=torch.float32, device=dev)
ys = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(2)]
engs = []
for k in range(K):
    e = pkg.Engine(N, B, link_flags=3, tile_channels=256); e.set_chain(chains.chain5(pkg, D))
    if k == 0: e.fill_noise(x, B, 0)
    engs.append(e)
for e in engs:
    for i in range(M):
        e.process(x, out=ys[i & 1], n_frames=B)
torch.cuda.synchronize()
