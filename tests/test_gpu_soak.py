"""A time-boxed slice of every builder-run soak, under the driver's eyes.

The long runs (thousands of seeds, `profiles/r0N_soaks.txt`) come from the committed generators in tools/; this module imports
THE SAME generators and runs a fixed seed range of each, sized to 20 - 30 s, with the same comparison: every output sample against
the CPU oracle (tests' checker, oracle/), worst ulp == 0 and the sign of zeros for the exact-arithmetic kinds, bit for bit for
integer FIR data, the stated relative-RMS bar (1e-6) for FIR noise, closed-form sums for the mailbox exchange.  Each test prints
seeds x actions.  `budget_s` is a guard against a slow box (the sweep stops early, the test then insists on a minimum number of runs).
"""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _tool(name):
    for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__)), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    return importlib.import_module(name)


@pytest.fixture
def keep_env():
    """the sweeps select kernels through DSPFX_* variables: leave the process as it was found"""
    saved = {k: v for k, v in os.environ.items() if k.startswith("DSPFX_")}
    yield
    for k in [k for k in os.environ if k.startswith("DSPFX_")]:
        if k not in saved:
            del os.environ[k]
    os.environ.update(saved)


def test_ring_soak_slice(keep_env):
    """tools/r05_ring_soak.py: delay-ring length changes in mid-stream (seconds / decay stores, set_delay_len, reserve, trim, reset;
    half of the runs store from a second thread behind a held stream) on random chains -- 160 seeds."""
    r = _tool("r05_ring_soak").run(31000, 160, budget_s=45)
    print("ring soak: %d seeds x %d actions (%d blocks with stores from a second thread), worst %d ulp, %.0f s" % (
        r["ran"], r["actions"], r["threaded_blocks"], r["worst"], r["seconds"]))
    assert r["bad"] == [] and r["worst"] == 0 and r["ran"] >= 60


def test_store_soak_slice(keep_env):
    """tools/r04_store_soak.py: slider stores in mid-stream on random chains while the background compiler swaps kernels in -- 24 seeds."""
    r = _tool("r04_store_soak").run(32000, 24, budget_s=45)
    print("store soak: %d seeds x %d stores (%d engines ended on run-time kernels), worst %d ulp, %.0f s" % (
        r["ran"], r["stores"], r["adopted"], r["worst"], r["seconds"]))
    assert r["bad"] == [] and r["worst"] == 0 and r["ran"] >= 10


def test_ragged_calls_soak_slice(keep_env):
    """tools/r04_ragged_calls_soak.py: calls of 1..256 frames (delays shorter than a call included) with stores in between -- 40 seeds."""
    r = _tool("r04_ragged_calls_soak").run(33000, 40, budget_s=45)
    print("ragged calls: %d seeds x %d calls, worst %d ulp, %.0f s" % (r["ran"], r["calls"], r["worst"], r["seconds"]))
    assert r["bad"] == [] and r["worst"] == 0 and r["ran"] >= 15


def test_chain_sweep_slice(keep_env):
    """tools/chain_sweep.py: random chains of 1..16 nodes through the interpreter AND the run-time specialised kernels -- 24 seeds x 2."""
    r = _tool("chain_sweep").run(34000, 24, budget_s=45)
    print("chain sweep: %d runs (interpreter + specialised), worst %d ulp, %.0f s" % (r["ran"], r["worst"], r["seconds"]))
    assert r["bad"] == [] and r["worst"] == 0 and r["ran"] >= 20


def test_graph_sweep_slice(keep_env):
    """tools/graph_sweep.py: random DAGs as one generated kernel vs the oracle's node-by-node evaluation (ulp), vs the run-by-run
    evaluation (bits) and vs series of small kernels (bits) -- 8 seeds."""
    r = _tool("graph_sweep").run(35000, 8, budget_s=50)
    print("graph sweep: %d seeds, %d series plans, %d graphs with non-zero output, worst %d ulp, %.0f s" % (
        r["ran"], r["cuts"], r["nonzero"], r["worst"], r["seconds"]))
    assert r["bad"] == [] and r["worst"] == 0 and r["ran"] >= 3 and r["nonzero"] >= 1


def test_fir_soak_slice(keep_env):
    """tools/fir_soak.py: 160 blocks per leg (the history ring wraps 5 .. 320 times), every sweep kernel, three shapes; integers bit
    for bit, noise under the stated relative RMS of 1e-6."""
    bad, legs = _tool("fir_soak").run(160)
    print("fir soak: %d legs x 160 blocks, failures %s" % (legs, bad))
    assert bad == [] and legs == 24


def test_mailbox_soak_slice():
    """tools/r04_mailbox_soak.py: 3 processes on this GPU, 20 000 exchanges through dspfx_mix_allreduce's mailboxes with random
    sleeps and a 20 ms stall every 1000th exchange; every frame of every exchange against the closed-form sum."""
    res = _tool("r04_mailbox_soak").run(3, 20000, timeout=240)
    print("mailbox soak: %s" % res)
    assert len(res) == 3 and all(isinstance(r, dict) and r["bad"] == 0 and r["backend"] == "mailbox" for r in res), res
