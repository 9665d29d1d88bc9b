#!/usr/bin/env python3
"""Calls of RANDOM length (1..256 frames each, engine max_frames 256) on random chains of the exact-arithmetic kinds -- delays shorter
than a call included (the engine splits the call at the delay length) -- with slider stores between calls, under the product's
defaults; every sample against the oracle fed the same calls.   usage: r04_ragged_calls_soak.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
from __graft_entry__ import load_package
E = load_package()
import oracle as O
import test_gpu_parity as T
from chains import ulp_diff
def run(s0=50000, cnt=100, budget_s=None):
    """see the module docstring; budget_s stops the sweep early (the test suite's time box); returns the counters."""
    t0, worst, bad, ran, n_calls = time.time(), 0, [], 0, 0
    for seed in range(s0, s0 + cnt):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        rng = np.random.default_rng(seed)
        chain = []
        while len(chain) < int(rng.integers(1, 9)):
            n = T._random_exact_node(E, rng)
            if n.kind not in (E.ADD, E.MIX, E.ENVELOPE):
                chain.append(n)
        tile = int(rng.choice([0, 0, 64]))
        N = int(rng.choice([64, 192, 1024])) if tile else int(rng.choice([1, 63, 100, 273, 1000, 2085]))
        lf = int(rng.choice([0, 1, 3]))
        calls = [int(rng.integers(1, 257)) for _ in range(24)]
        total = sum(calls)
        x = T.noise_block(N, total, seed=seed)
        eng = E.Engine(N, 256, link_flags=lf, tile_channels=tile)
        eng.set_chain(chain)
        y = np.empty_like(x)
        nodes = [[O.node_from_desc(n.oracle_desc()) for n in chain] for _ in range(N)]
        ref = np.empty_like(x)
        f0 = 0
        for k, nf in enumerate(calls):
            if rng.random() < 0.25:
                i = int(rng.integers(0, len(chain)))
                kd = chain[i].kind
                st = {E.GAIN: (0, float(rng.uniform(0, 2))), E.BIQUAD: (int(rng.integers(3, 6)), float(rng.uniform(-1, 1))), E.LOW_PASS: (0, float(rng.uniform(0, 1))),
                      E.HIGH_PASS: (0, float(rng.uniform(0, 1))), E.REVERB: (0, float(rng.uniform(0, 0.9))), E.DISTORT: (0, float(rng.uniform(0.1, 6)))}.get(kd)
                if st:
                    eng.set_param(i, st[0], st[1])
                    for c in range(N):
                        nodes[c][i].set_param(st[0], st[1])
            if k == 12:
                eng.kernels_ready(60000)
            dx = torch.from_numpy(E.to_layout(x[f0:f0 + nf], tile)).cuda()
            dy = torch.empty_like(dx)
            eng.process(dx, out=dy, n_frames=nf)
            torch.cuda.synchronize()
            y[f0:f0 + nf] = E.from_layout(dy.cpu().numpy(), nf, N, tile)
            for c in range(N):
                ref[f0:f0 + nf, c] = O.chain_run(nodes[c], x[f0:f0 + nf, c], lf, block=128)
            f0 += nf
        eng.close()
        ran += 1
        n_calls += len(calls)
        ok = np.isfinite(ref)
        d = ulp_diff(y[ok], ref[ok]) if np.array_equal(np.isfinite(y), ok) else np.array([1 << 30])
        w = int(d.max()) if d.size else 0
        worst = max(worst, w)
        if w > 1 or not np.array_equal(np.signbit(y[ok]), np.signbit(ref[ok])):
            bad.append((seed, w, N, tile, lf, calls[:6], [n.kind for n in chain]))
        if (seed - s0) % 20 == 19:
            print("... %d runs, worst %d ulp, failures %s, %.0f s" % (seed - s0 + 1, worst, bad[:3], time.time() - t0), flush=True)
    print("seeds %d..%d: calls of 1..256 frames, worst ulp vs oracle %d, failures %s, %.0f s" % (s0, s0 + cnt - 1, worst, bad, time.time() - t0))
    return dict(ran=ran, calls=n_calls, worst=worst, bad=bad, seconds=time.time() - t0)


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 50000, int(sys.argv[2]) if len(sys.argv) > 2 else 100)
