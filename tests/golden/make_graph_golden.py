#!/usr/bin/env python3
"""Golden vectors for whole saved graphs: tests/golden/graphs/<name>.npz = the DSPConfig document, a 3-channel noise input
and the output of the oracle's node-by-node evaluation (oracle/graph_eval.py, reference semantics of node.rs:267-352).
Made by the oracle (the reference itself cannot run here: parity unpinned upstream); they pin the oracle, the document
parser and the graph planners against drift, and the GPU tests compare the HIP path with them.
usage: python tests/golden/make_graph_golden.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
from __graft_entry__ import load_package
load_package()
from dsp_stuff_amd import graph as G
import graphs, graph_eval, oracle as O

CASES = {
    "diamond": graphs.diamond(), "lfo_tremolo": graphs.lfo_tremolo(), "fan_in_three": graphs.fan_in_three(),
    "routing_ba": graphs.routing("B", "A"), "cab_rig": graphs.cab_rig(), "cab_rig_wet_dry": graphs.cab_rig(dry=True),
    "long_rig_0": graphs.long_rig(0, 12), "long_rig_1_fir": graphs.long_rig(1, 12, fir_at=5),
    "long_rig_2_wet_dry": graphs.long_rig(2, 12, dry_mix=True), "long_rig_3_fir_wet_dry": graphs.long_rig(3, 12, fir_at=5, dry_mix=True),
}
for name, text in CASES.items():
    x = O.noise(0x5EED00AA, np.arange(3), np.arange(512))
    y = graph_eval.run_graph(G.Graph(text), x)
    assert np.isfinite(y).all() and np.abs(y).max() > 0, name
    np.savez_compressed(os.path.join(HERE, "graphs", name + ".npz"), doc=np.array(text), x=x, y=y)
    print(name, y.shape, float(np.abs(y).max()))
