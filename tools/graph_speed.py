#!/usr/bin/env python3
"""Step time of a whole saved graph (DAG) on GraphEngine: tests/graphs.py documents at 1 048 576 channels, as one
generated kernel (dspfx_graph_set) and evaluated run by run.   usage: graph_speed.py [name | random:<seed>:<nodes> | long:<seed> | cab_rig] [channels]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
pkg = load_package()
from dsp_stuff_amd import graph as G
import graphs
N, B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20, 128
name = sys.argv[1] if len(sys.argv) > 1 else "diamond"
text = graphs.cab_rig(4096) if name == "cab_rig" else graphs.long_rig(int(name.split(":")[1]), 12) if name.startswith("long:") else graphs.random_dag(*map(int, name.split(":")[1:]), libm=True) if name.startswith("random:") else getattr(graphs, name)()   # random:<seed>:<nodes>
x = torch.empty(B * N, dtype=torch.float32, device="cuda")
for fused in (None, False):
    ge = G.GraphEngine(text, N, B, tile_channels=256, fused=fused)
    ge.util.fill_noise(x, B, 0)
    ge.tune_placement(x, B)
    for _ in range(20): ge.process(x, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 100
    e0.record()
    for _ in range(steps): ge.process(x, B)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print("%s %s: %.4f ms/block, %.3e samples/s, %.0f GB/s of in+out" % (name, "run by run" if fused is False else ("one kernel " if ge.fused is not None else "segments   "), ms, N * B / ms * 1e3, 8 * N * B / ms / 1e6))
    print(ge.describe())
    ge.close()
