import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fir_default: leave the FIR sweep at the engine's default (split precision) for this test")
    config.addinivalue_line("markers", "jit_async: leave the background specialisation of small engines on for this test")


@pytest.fixture(autouse=True)
def _f32_fir_sweep_unless_asked(monkeypatch, request):
    """The engine's default steady-state FIR sweep is the split-precision one (round 3).  The parity tests name the sweep they
    mean -- f32 (`kernel = 1 / rect`) or split (`kernel = split`, dspfx_set_fir_precision) -- so DSPFX_FIR_SPLIT=0 is the
    baseline of every test; tests marked `fir_default` see the default as a host would."""
    if "fir_default" not in request.keywords:
        monkeypatch.setenv("DSPFX_FIR_SPLIT", "0")
    # Small engines have their chain shape specialised in the background and switch kernels when it is ready (jit.hip): same
    # samples, but WHEN the switch happens is a matter of timing, and the tests compare engines block for block (the bus'
    # summation order follows the kernel).  Off by default here; tests marked `jit_async` exercise it.
    if "jit_async" not in request.keywords and os.environ.get("DSPFX_TEST_JIT_ASYNC") != "1":   # (=1: the whole suite with it on)
        monkeypatch.setenv("DSPFX_JIT_ASYNC", "0")


@pytest.fixture(scope="session")
def dspfx():
    """The product package (dsp-stuff_amd/), loaded through its in-tree HIP library."""
    from __graft_entry__ import load_package
    return load_package()
