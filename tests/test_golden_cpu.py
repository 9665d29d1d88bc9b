"""The CPU oracle must keep reproducing the committed golden vectors bit for bit (they were
generated from it and cross-checked against the numpy model: tests/golden/make_golden.py)."""
import numpy as np

import oracle as O
from golden_util import load_all


def test_golden_vectors_present():
    names = {g["name"] for g in load_all()}
    assert {"chain3_link3", "chain5_link3", "distort_mode1", "fir_int5", "onepole"} <= names
    assert len(names) >= 20


def test_oracle_reproduces_golden():
    for g in load_all():
        nodes = [O.node_from_desc(d) for d in g["descs"]]
        y = O.chain_run(nodes, g["x"], g["link_flags"])
        assert np.array_equal(y.view(np.uint32), g["y"].view(np.uint32)), g["name"]
