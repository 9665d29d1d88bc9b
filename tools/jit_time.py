#!/usr/bin/env python3
"""How long dspfx_graph_set takes (hiprtc compile + module load) for a few graphs at 1 048 576 channels, and for a cached one."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
E = load_package()
from dsp_stuff_amd import graph as G
import graphs, torch
torch.zeros(1, device="cuda")
for name, text in [("diamond", graphs.diamond()), ("lfo_tremolo", graphs.lfo_tremolo()), ("random16", graphs.random_dag(116, 16, libm=True)), ("diamond again", graphs.diamond())]:
    plan = G.fused_plan(G.Graph(text))
    e = E.Engine(1 << 20, 128, tile_channels=256)
    t = time.time(); e.set_graph(*plan); dt = time.time() - t
    print("%s: set_graph %.2f s  %s" % (name, dt, [l for l in e.describe().splitlines() if l.startswith("stage")][0][:90]))
    e.close()
