#!/usr/bin/env python3
"""Randomised parity sweep of the FIR path against the oracle: tap counts, channel counts, block lengths, tap reloads in
mid-stream, both I/O layouts, every steady-state sweep (skewed, rectangular, split precision) and the exact kernel.
Integer data must come out bit for bit (every product and sum is exact in f32 and in bf16 parts); noise within the stated
relative RMS tolerance (1e-6) on the MFMA sweeps and bit for bit on the exact kernel.
usage: fir_sweep.py [n_cases] [seed0]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import oracle as O
from __graft_entry__ import load_package
fx = load_package()

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
TOL = 1e-6
bad = 0
worst = 0.0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    mode = ("skew", "rect", "split", "exact")[case % 4]
    for k in ("DSPFX_FIR_SKEW", "DSPFX_FIR_SPLIT", "DSPFX_FIR_KERNEL"):
        os.environ.pop(k, None)
    os.environ["DSPFX_FIR_KERNEL"] = "0" if mode == "exact" else "1"
    if mode == "rect":
        os.environ["DSPFX_FIR_SKEW"] = "0"
    os.environ["DSPFX_FIR_SPLIT"] = "1" if mode == "split" else "0"     # (the engine's default is the split sweep)
    tile = int(rng.choice([0, 0, 64, 256]))
    N = int(rng.integers(1, 6)) * tile if tile else int(rng.integers(1, 400))
    B = int(rng.choice([16, 48, 64, 100, 128, 128, 128, 256]))
    integers = bool(rng.integers(0, 2))
    fir_mode = int(rng.integers(0, 2))
    plan = [(int(rng.integers(16, 700)), int(rng.integers(2, 9)) * B) for _ in range(int(rng.integers(1, 4)))]
    mk = (lambda T: rng.integers(-4, 5, T).astype(np.float64)) if integers else (lambda T: rng.uniform(-1, 1, T) * np.exp(-np.arange(T) / 150.0))
    total = sum(n for _, n in plan)
    x = (rng.integers(-8, 9, (total, N)).astype(np.float32) if integers else rng.uniform(-1, 1, (total, N)).astype(np.float32))
    h0 = mk(plan[0][0])
    eng = fx.Engine(N, B, link_flags=0, tile_channels=tile)
    eng.set_chain([fx.Fir(h0, fir_mode)])
    nodes = [O.Node(O.FIR, mode=fir_mode, taps_reversed=h0[::-1]) for _ in range(N)]
    got, ref, f0 = [], [], 0
    for k, (T, n) in enumerate(plan):
        if k:
            h = mk(T)
            eng.set_taps(0, h, fir_mode)
            for nd in nodes:
                nd.set_taps(h[::-1])
        seg = x[f0:f0 + n]
        ys = np.empty_like(seg)
        for b0 in range(0, n, B):
            dx = torch.from_numpy(fx.to_layout(seg[b0:b0 + B], tile)).cuda()
            dy = torch.empty_like(dx)
            eng.process(dx, out=dy, n_frames=B)
            torch.cuda.synchronize()
            ys[b0:b0 + B] = fx.from_layout(dy.cpu().numpy(), B, N, tile)
        got.append(ys)
        ref.append(np.stack([np.concatenate([nodes[c].process(seg[i:i + 128, c]) for i in range(0, n, 128)]) for c in range(N)], axis=1))
        f0 += n
    got, ref = np.concatenate(got), np.concatenate(ref)
    if mode == "exact":
        ok = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        err = 0.0 if ok else float(np.abs(got - ref).max())
    elif integers:
        ok = np.array_equal(got, ref)
        err = 0.0 if ok else float(np.abs(got - ref).max())
    else:
        e = got.astype(np.float64) - ref
        err = float(np.sqrt(np.mean(e ** 2)) / max(np.sqrt(np.mean(ref.astype(np.float64) ** 2)), 1e-30))
        ok = err < TOL
    worst = max(worst, err)
    if not ok:
        bad += 1
        print("FAIL case", seed0 + case, mode, "N", N, "tile", tile, "B", B, "plan", plan, "integers", integers, "err", err)
    eng.close()
print("%d cases, %d failed, worst relative RMS (MFMA sweeps) %.3g" % (n_cases, bad, worst))
sys.exit(1 if bad else 0)
