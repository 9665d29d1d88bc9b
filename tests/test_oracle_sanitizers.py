"""AddressSanitizer + UBSan pass over the CPU oracle (the checker must not have memory bugs of its
own).  GPU sanitizers are unavailable on the pool, so this is the CPU build only: the KAT-style
workload runs in a child interpreter with libasan preloaded."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import ctypes, sys
    sys.path.insert(0, %(oracle)r)
    import numpy as np
    import oracle as O
    O._lib = O._bind(ctypes.CDLL(%(lib)r))
    x = O.noise(1, [0], np.arange(1024))[:, 0]
    descs = [dict(kind=O.BIQUAD), dict(kind=O.DISTORT, params=[3.0], mode=O.FUZZ),
             dict(kind=O.REVERB, params=[0.5], delay_len=200),
             dict(kind=O.FIR, params=[], mode=0, taps_reversed=np.linspace(1, 0, 300)),
             dict(kind=O.MIX, params=[0.3]), dict(kind=O.OVERDRIVE, params=[5, .5, .5]),
             dict(kind=O.LOW_PASS, params=[0.2]), dict(kind=O.CHEBYSHEV, params=[2.0, 3.0])]
    nodes = [O.node_from_desc(d) for d in descs]
    y = O.chain_run(nodes, x, 3, side=x[::-1].copy(), ctl={(4, 0): x, (5, 1): x})
    for n in nodes:
        n.reset()
    y2 = O.chain_run(nodes, x[:100], 1, block=50)
    out, mix = O.run_noise_channels(descs[:4], 1, 0, 37, 0, 5, want_mix=True, n_threads=4)
    assert np.isfinite(y).all() and out.shape == (640, 37)
    print("sanitized run ok")
""")


def test_oracle_under_asan_ubsan():
    lib = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-200:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not found")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", SCRIPT % {"oracle": os.path.join(ROOT, "oracle"), "lib": lib}],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "sanitized run ok" in p.stdout
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
