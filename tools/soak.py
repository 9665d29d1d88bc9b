#!/usr/bin/env python3
"""Soak run: many seeded random chains (all exact-arithmetic kinds + the libm kinds with their own bar) through the
default kernels and the forced two-channel interpreter, ragged sizes, both layouts, against the CPU oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from __graft_entry__ import load_package
import oracle as O
import test_gpu_parity as T
from chains import ulp_diff
dspfx = load_package()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
worst, cases = 0, 0
for seed in range(n_seeds):
    rng = np.random.default_rng(5000 + seed)
    os.environ.pop("DSPFX_VARIANT", None)
    if seed % 3 == 1:
        os.environ["DSPFX_VARIANT"] = "static=0,f=8,cpl=2"
    elif seed % 3 == 2:
        os.environ["DSPFX_VARIANT"] = "static=0,f=4,cpl=2"
    chain = [T._random_exact_node(dspfx, rng) for _ in range(int(rng.integers(1, 14)))]
    tile = int(rng.choice([0, 64, 256]))
    N = int(rng.choice([1, 3]) * tile) if tile else int(rng.choice([1, 2, 63, 100, 129, 130, 273, 418]))
    block = int(rng.choice([128, 256, 384, 768]))
    lf = int(rng.choice([0, 1, 3]))
    nf = 768
    x, side = T.noise_block(N, nf, seed=seed), T.noise_block(N, nf, seed=seed + 999)
    got = T.run_gpu(dspfx, torch, chain, x, link_flags=lf, block=block, side=side, tile=tile)
    ref = T.run_oracle(chain, x, lf, side)
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), ok), (seed, "finite pattern")
    d = int(ulp_diff(got[ok], ref[ok]).max()) if ok.any() else 0
    worst = max(worst, d); cases += 1
    assert d <= 1, (seed, d, [(n.kind, n.mode) for n in chain], N, block, tile, lf, os.environ.get("DSPFX_VARIANT"))
print("soak ok: %d random chains, worst %d ulp" % (cases, worst))
