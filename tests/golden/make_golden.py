#!/usr/bin/env python3
"""Generate the golden input/output vectors in tests/golden/*.npz.

The reference (Rust nightly) cannot run in this image and ships no fixtures, so these vectors
come from the C oracle (oracle/dspfx_oracle.c) and every case is cross-checked here against the
independent numpy-float32 model (oracle/numpy_model.py) before it is written: PARITY UNPINNED
upstream, pinned between two restatements.  Re-run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy_model as M   # noqa: E402
import oracle as O        # noqa: E402
from chains import rbj_highpass, rbj_lowpass, fir_taps   # noqa: E402

F = np.float32
SEED = 0x5EED0001


def noise(n, ch=0, n0=0):
    return O.noise(SEED, [ch], np.arange(n0, n0 + n))[:, 0]


def model_chain(descs, x, link_flags):
    """numpy model of a 1-channel chain, block-of-128 structure irrelevant for these nodes."""
    objs = []
    for d in descs:
        k, p = d["kind"], d.get("params") or []
        if k == O.BIQUAD:
            objs.append(M.Biquad(*p))
        elif k == O.LOW_PASS:
            objs.append(M.OnePole(p[0]))
        elif k == O.HIGH_PASS:
            objs.append(M.OnePole(p[0], high=True))
        elif k == O.REVERB:
            objs.append(M.Reverb(d["delay_len"], p[0]))
        elif k == O.FIR:
            objs.append(M.Fir(d["taps_reversed"], d.get("mode") == O.FIR_AVERAGE))
        else:
            objs.append(None)
    y = np.asarray(x, F)
    for i, (d, o) in enumerate(zip(descs, objs)):
        if (link_flags & 2) if i == 0 else (link_flags & 1):
            y = M.link_scale(y)
        k, p = d["kind"], d.get("params") or []
        if k == O.GAIN:
            y = M.gain(y, p[0])
        elif k == O.DISTORT:
            y = np.concatenate([M.distort(y[b:b + 128], p[0], d["mode"]) for b in range(0, len(y), 128)])
        elif k == O.OVERDRIVE:
            y = M.overdrive(y, *p[:3])
        elif k == O.CHEBYSHEV:
            y = M.chebyshev(y, *p[:2])
        else:
            y = o.run(y)
    return y


def case(name, descs, x, link_flags, exact=True, rtol=0.0):
    nodes = [O.node_from_desc(d) for d in descs]
    y = O.chain_run(nodes, x, link_flags)
    ym = model_chain(descs, x, link_flags)
    if exact:
        assert np.array_equal(y.view(np.uint32), ym.view(np.uint32)), f"{name}: oracle != numpy model"
    else:
        assert np.allclose(y, ym, rtol=rtol, atol=rtol), f"{name}: oracle !~ numpy model"
    meta = [{k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in d.items() if k != "taps_reversed"} for d in descs]
    taps = [np.asarray(d["taps_reversed"], np.float64) if d.get("taps_reversed") is not None else np.zeros(0) for d in descs]
    return name, dict(x=np.asarray(x, F), y=y, link_flags=np.int32(link_flags), meta=np.array(json.dumps(meta)),
                      **{f"taps{i}": t for i, t in enumerate(taps) if t.size})


def main():
    cases = []
    x = noise(1024)
    lp, hp = rbj_lowpass(1000.0, 0.7071), rbj_highpass(80.0, 0.7071)
    d3 = [dict(kind=O.GAIN, params=[0.8]), dict(kind=O.BIQUAD, params=lp), dict(kind=O.REVERB, params=[0.5], delay_len=256)]
    d5 = [dict(kind=O.BIQUAD, params=lp), dict(kind=O.DISTORT, params=[3.0], mode=O.SOFT_CLIP),
          dict(kind=O.REVERB, params=[0.5], delay_len=256), dict(kind=O.BIQUAD, params=hp), dict(kind=O.GAIN, params=[0.5])]
    for lf in (0, 1, 3):
        cases.append(case(f"chain3_link{lf}", d3, x, lf))       # KAT-10
        cases.append(case(f"chain5_link{lf}", d5, x, lf))
    xs = x[:256] * F(2.5)
    for mode in range(9):
        exact = mode not in (O.TANH, O.SIN, O.ATAN, O.FUZZ)
        cases.append(case(f"distort_mode{mode}", [dict(kind=O.DISTORT, params=[3.0], mode=mode)], xs, 0,
                          exact=exact, rtol=3e-6))
    cases.append(case("onepole", [dict(kind=O.LOW_PASS, params=[0.3]), dict(kind=O.HIGH_PASS, params=[0.9])], x[:512], 3))
    cases.append(case("overdrive", [dict(kind=O.OVERDRIVE, params=[5.0, 0.7, 0.9])], x[:256], 0, exact=False, rtol=3e-7))
    cases.append(case("chebyshev", [dict(kind=O.CHEBYSHEV, params=[4.0, 2.0])], x[:256], 0, exact=False, rtol=6e-7))
    h = np.array([1, -2, 3, 4, -1], np.float64)
    xi = np.round(x[:512] * 8).astype(F)
    cases.append(case("fir_int5", [dict(kind=O.FIR, params=[], mode=O.FIR_BALANCED, taps_reversed=h[::-1].copy())], xi, 0))
    cases.append(case("fir_200", [dict(kind=O.FIR, params=[], mode=O.FIR_AVERAGE, taps_reversed=fir_taps(200)[::-1].copy())],
                      x[:768], 3, exact=False, rtol=2e-6))
    for name, arrs in cases:
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"{name:18s} {os.path.getsize(path):6d} B")


if __name__ == "__main__":
    main()
