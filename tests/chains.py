"""Chains and parameters shared by the parity tests.  The workloads themselves live in the package
(dsp-stuff_amd/workloads.py: bench.py times them without importing from tests/); this module re-exports them and adds the
tests' own ulp distance."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from dsp_stuff_amd import workloads as _w  # noqa: E402

rbj_lowpass, rbj_highpass, chain3, chain5, fir_taps = _w.rbj_lowpass, _w.rbj_highpass, _w.chain3, _w.chain5, _w.fir_taps


def ulp_diff(a, b):
    """Distance in units of f32 ordering (sign-magnitude mapped to a line); NaN==NaN -> 0."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)
