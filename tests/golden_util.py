"""Load the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_all():
    out = []
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        z = np.load(path)
        descs = json.loads(str(z["meta"]))
        for i, d in enumerate(descs):
            d["taps_reversed"] = z[f"taps{i}"] if f"taps{i}" in z.files else None
        out.append(dict(name=os.path.basename(path)[:-4], x=z["x"], y=z["y"], link_flags=int(z["link_flags"]), descs=descs))
    return out


# per-case bar in ulps (None = relative-to-peak bar for Fuzz / FIR); mirrors tests/test_gpu_parity.py
def bar_for(name):
    if name.startswith("distort_mode"):
        m = int(name[len("distort_mode"):])
        return {2: 2, 5: 1, 6: 1, 4: None}.get(m, 1)
    if name in ("overdrive", "chebyshev"):
        return 4
    if name.startswith("fir_"):
        return 0 if name == "fir_int5" else None
    return 1
