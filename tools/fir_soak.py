#!/usr/bin/env python3
"""Long FIR runs against the oracle: hundreds of blocks, so that the history ring wraps many times and the absolute sample
clock grows large; every sweep kernel; integer data (bit for bit) and noise (relative RMS).  usage: fir_soak.py [blocks]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import oracle as O
from __graft_entry__ import load_package
fx = load_package()


def run(blocks=600, modes=("skew", "rect", "split", "half"), shapes=((333, 40, 128), (64, 70, 256), (1500, 33, 128))):
    """see the module docstring; returns [(T, N, B, data, mode, message)] of the failing legs and the number of legs run."""
    bad, legs = [], 0
    saved = {k: os.environ.get(k) for k in ("DSPFX_FIR_SKEW", "DSPFX_FIR_SPLIT", "DSPFX_FIR_HALF")}
    try:
        for T, N, B in shapes:
            rng = np.random.default_rng(T)
            for integers in (True, False):
                h = rng.integers(-4, 5, T).astype(np.float64) if integers else rng.uniform(-1, 1, T) * np.exp(-np.arange(T) / (T / 5.0))
                x = rng.integers(-8, 9, (blocks * B, N)).astype(np.float32) if integers else rng.uniform(-1, 1, (blocks * B, N)).astype(np.float32)
                ref = O.run_channels([fx.Fir(h).oracle_desc()], x, 0)
                for mode in modes:
                    for k in ("DSPFX_FIR_SKEW", "DSPFX_FIR_SPLIT", "DSPFX_FIR_HALF"):
                        os.environ.pop(k, None)
                    os.environ["DSPFX_FIR_SPLIT"] = "1" if mode in ("split", "half") else "0"     # (the engine's default is the two-part f16 sweep)
                    if mode == "split":
                        os.environ["DSPFX_FIR_HALF"] = "0"
                    if mode == "rect":
                        os.environ["DSPFX_FIR_SKEW"] = "0"
                    eng = fx.Engine(N, B, link_flags=0)
                    eng.set_chain([fx.Fir(h)])
                    dx = torch.from_numpy(x).cuda()
                    dy = torch.empty_like(dx)
                    for b in range(blocks):
                        eng.process(dx[b * B:(b + 1) * B], out=dy[b * B:(b + 1) * B], n_frames=B)
                    torch.cuda.synchronize()
                    y = dy.cpu().numpy()
                    if integers:
                        ok = np.array_equal(y, ref)
                        msg = "bit for bit" if ok else "MISMATCH max %.3g" % np.abs(y - ref).max()
                    else:
                        tail = slice(-min(64, blocks) * B, None)       # the last blocks: after every wrap of the ring
                        e = y[tail].astype(np.float64) - ref[tail]
                        err = np.sqrt(np.mean(e ** 2)) / np.sqrt(np.mean(ref[tail].astype(np.float64) ** 2))
                        ok = err < 1e-6
                        msg = "relative RMS over the last %d blocks %.3g" % (min(64, blocks), err)
                    legs += 1
                    if not ok:
                        bad.append((T, N, B, "integers" if integers else "noise", mode, msg))
                    print("T %4d N %3d B %3d %4d blocks %-8s %-5s %s" % (T, N, B, blocks, "integers" if integers else "noise", mode, msg))
                    eng.close()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    print("failures:", len(bad))
    return bad, legs


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 600)[0] else 0)
