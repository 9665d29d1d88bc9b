import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fir_f32: this test means the f32 FIR sweep (sets DSPFX_FIR_SPLIT=0; the product default is the split-precision sweep)")
    config.addinivalue_line("markers", "jit_frozen: this test needs the kernel it starts on to stay (sets DSPFX_JIT_ASYNC=0; by default small engines adopt kernels compiled in the background)")


@pytest.fixture(autouse=True)
def _product_defaults_unless_a_test_opts_out(monkeypatch, request):
    """The suite runs what ships: the split-precision FIR sweep and the background specialisation of small engines are ON,
    as a host would see them (round 4; round 3 forced both off for every test).  A test that is ABOUT the f32 sweep or
    about one particular kernel says so with a marker."""
    if "fir_f32" in request.keywords:
        monkeypatch.setenv("DSPFX_FIR_SPLIT", "0")
    if "jit_frozen" in request.keywords:
        monkeypatch.setenv("DSPFX_JIT_ASYNC", "0")


@pytest.fixture(scope="session", autouse=True)
def _package_is_importable():
    """`from dsp_stuff_amd import graph` in a test works whatever test a worker happens to run first (pytest-xdist deals them out
    in any order): the package directory `dsp-stuff_amd/` is registered under its importable name once per session."""
    from __graft_entry__ import load_package
    load_package()


@pytest.fixture(scope="session")
def dspfx():
    """The product package (dsp-stuff_amd/), loaded through its in-tree HIP library."""
    from __graft_entry__ import load_package
    return load_package()
