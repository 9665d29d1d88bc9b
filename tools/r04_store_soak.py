#!/usr/bin/env python3
"""Slider stores in mid-stream on random chains, under the product's defaults: seeded random chains of the exact-arithmetic kinds
(1..10 nodes, delays included), ragged and whole channel counts from 1 to 4227, both layouts, every link-flag setting, blocks of 100,
128 or 256 frames; five to eight random stores per run (any node, any slider: biquad stores regenerate + reset, delay stores swap in
a zero ring, level / ratio stores just land) and a dspfx_reset; the engine starts on the interpreter and adopts the kernels the
background compiler makes for its shape wherever that happens (half of the runs wait for them in the middle, the others sleep at
random).  Every output sample against the oracle with the same stores at the same blocks: ulp and the sign of zeros.
usage: r04_store_soak.py [first_seed] [count]"""
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


def random_node(rng):
    while True:
        n = T._random_exact_node(E, rng)
        if n.kind not in (E.ADD, E.MIX):          # (side inputs have their own tests)
            return n


def random_store(rng, chain):
    """(node index, slider index, value) for a random slider of a random node"""
    for _ in range(50):
        i = int(rng.integers(0, len(chain)))
        k = chain[i].kind
        if k == E.GAIN:
            return i, 0, float(rng.uniform(0.0, 2.0))
        if k == E.BIQUAD:
            return i, int(rng.integers(3, 6)), float(rng.uniform(-1.0, 1.0))      # a feed-forward slider: the filter stays stable
        if k in (E.LOW_PASS, E.HIGH_PASS):
            return i, 0, float(rng.uniform(0.0, 1.0))
        if k == E.REVERB:
            return i, 0, float(rng.uniform(0.0, 0.9))                             # decay: a new zero ring all the same
        if k == E.DISTORT:
            return i, 0, float(rng.choice([rng.uniform(0.1, 6.0), 2.0, 6.0, 3.0]))  # incl. an even integer that is no power of two
        if k == E.ENVELOPE:
            return i, int(rng.integers(0, 2)), float(rng.choice([0.0, 5.0, 120.0]))
    return None


def run(s0=9000, cnt=60, budget_s=None):
    """see the module docstring; budget_s stops the sweep early (the test suite's time box); returns the counters."""
    t0, worst, bad, ran, adopted, stores = time.time(), 0, [], 0, 0, 0
    for seed in range(s0, s0 + cnt):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        rng = np.random.default_rng(seed)
        chain = [random_node(rng) for _ in range(int(rng.integers(1, 11)))]
        tile = int(rng.choice([0, 0, 64, 256]))
        N = int(rng.choice([256, 1024, 4096])) if tile else int(rng.choice([1, 63, 100, 273, 1000, 2085, 4227, 576]))
        block = int(rng.choice([128, 128, 256, 100]))
        lf = int(rng.choice([0, 1, 3]))
        nblocks = 20
        x = T.noise_block(N, block * nblocks, seed=seed)
        acts = {}
        for _ in range(int(rng.integers(5, 9))):
            st = random_store(rng, chain)
            if st:
                acts.setdefault(int(rng.integers(1, nblocks)), []).append(st)
        reset_at = int(rng.integers(1, nblocks)) if rng.random() < 0.3 else -1
        wait_at = nblocks // 2 if rng.random() < 0.5 else -1
        eng = E.Engine(N, block, link_flags=lf, tile_channels=tile)
        eng.set_chain(chain)
        y = np.empty_like(x)
        for k in range(nblocks):
            for (i, p, v) in acts.get(k, []):
                eng.set_param(i, p, v)
                stores += 1
            if k == reset_at:
                eng.reset()
            if k == wait_at:
                eng.kernels_ready(60000)
            elif wait_at < 0 and rng.random() < 0.3:
                time.sleep(float(rng.uniform(0.0, 0.15)))
            dx = torch.from_numpy(E.to_layout(x[k * block:(k + 1) * block], tile)).cuda()
            dy = torch.empty_like(dx)
            eng.process(dx, out=dy, n_frames=block)
            torch.cuda.synchronize()
            y[k * block:(k + 1) * block] = E.from_layout(dy.cpu().numpy(), block, N, tile)
        adopted += 1 if "jit_" in eng.describe() else 0
        eng.close()
        descs = [n.oracle_desc() for n in chain]
        ref = np.empty_like(x)
        for c in range(N):
            nodes = [O.node_from_desc(d) for d in descs]
            for k in range(nblocks):
                for (i, p, v) in acts.get(k, []):
                    nodes[i].set_param(p, v)
                if k == reset_at:
                    for nd in nodes:
                        nd.reset()
                ref[k * block:(k + 1) * block, c] = O.chain_run(nodes, x[k * block:(k + 1) * block, c], lf, block=min(block, 128))
        ran += 1
        ok = np.isfinite(ref)
        if not np.array_equal(np.isfinite(y), ok):
            bad.append((seed, "finite"))
            continue
        d = ulp_diff(y[ok], ref[ok])
        w = int(d.max()) if d.size else 0
        worst = max(worst, w)
        if w > 1 or not np.array_equal(np.signbit(y[ok]), np.signbit(ref[ok])):
            bad.append((seed, w, N, tile, block, lf, [n.kind for n in chain]))
        if (seed - s0) % 10 == 9:
            print("... %d runs, worst %d ulp, failures %s, %.0f s" % (ran, worst, bad, time.time() - t0), flush=True)
    print("seeds %d..%d: %d runs, %d stores, %d engines ended on run-time specialised kernels; worst ulp vs oracle %d, failures %s, %.0f s" % (
        s0, s0 + cnt - 1, ran, stores, adopted, worst, bad, time.time() - t0))
    return dict(ran=ran, stores=stores, adopted=adopted, worst=worst, bad=bad, seconds=time.time() - t0)


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 9000, int(sys.argv[2]) if len(sys.argv) > 2 else 60)
