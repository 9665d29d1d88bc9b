#!/usr/bin/env python3
"""A wider sweep than the test suite runs, for plain chains: seeded random chains (1..16 nodes of the exact-arithmetic kinds,
random parameters, ragged channel counts, both layouts, every link-flag setting, 128- or 256-frame blocks) through the
interpreter kernels and through the run-time specialised ones (DSPFX_JIT=1, incl. the one-kernel form of 9..16-node runs),
against the oracle: ulp and the sign of zeros.   usage: chain_sweep.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
from __graft_entry__ import load_package
E = load_package()
import test_gpu_parity as T
from chains import ulp_diff
def run(s0=5000, cnt=100, budget_s=None):
    """see the module docstring; budget_s stops the sweep early (the test suite's time box); returns the counters."""
    t0, worst, bad, ran = time.time(), 0, [], 0
    for seed in range(s0, s0 + cnt):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        rng = np.random.default_rng(seed)
        n_nodes = int(rng.integers(1, 17))
        chain = [T._random_exact_node(E, rng) for _ in range(n_nodes)]
        tile = int(rng.choice([0, 64]))
        N = int(rng.choice([64, 128, 320])) if tile else int(rng.choice([1, 63, 100, 129, 273, 192]))
        block = int(rng.choice([128, 256]))
        lf = int(rng.choice([0, 1, 3]))
        nf = 768
        x, side = T.noise_block(N, nf, seed=seed), T.noise_block(N, nf, seed=seed + 100000)
        ref = T.run_oracle(chain, x, lf, side)
        ok = np.isfinite(ref)
        for jit in ("0", "1"):
            os.environ["DSPFX_JIT"] = jit
            got = T.run_gpu(E, torch, chain, x, link_flags=lf, block=block, side=side, tile=tile)
            ran += 1
            if not np.array_equal(np.isfinite(got), ok):
                bad.append((seed, jit, "finite")); continue
            d = ulp_diff(got[ok], ref[ok])
            w = int(d.max()) if d.size else 0
            worst = max(worst, w)
            if w > 1 or not np.array_equal(np.signbit(got[ok]), np.signbit(ref[ok])):
                bad.append((seed, jit, w))
    print("chains %d..%d (%d runs: interpreter and run-time specialised): worst ulp vs oracle %d, failures %s, %.0f s" % (s0, s0 + cnt - 1, ran, worst, bad, time.time() - t0))
    return dict(ran=ran, worst=worst, bad=bad, seconds=time.time() - t0)


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 5000, int(sys.argv[2]) if len(sys.argv) > 2 else 100)
