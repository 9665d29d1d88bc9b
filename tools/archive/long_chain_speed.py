#!/usr/bin/env python3
"""A 12-node chain at 1 048 576 channels through dspfx_chain_set and handed over as a graph (dspfx_graph_set).  Large
engines compile the run as one generated kernel either way; DSPFX_VARIANT=static=0 shows the chain engine's two-launch
form (8 + 4 nodes, interpreter), which is what profiles/r01_graph_one_kernel.txt compares against."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
E = load_package()
N, B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20, 128
chain = [E.Gain(0.9), E.BiQuad(1.0, -1.2, 0.5, 0.3, 0.2, 0.1), E.Distort(3.0, E.SOFT_CLIP), E.LowPass(0.3),
         E.Reverb(delay_samples=24000, decay=0.4), E.HighPass(0.2), E.Gain(1.1), E.Distort(2.0, E.HARD_CLIP),
         E.BiQuad(1.0, -0.5, 0.2, 0.4, 0.1, 0.0), E.Envelope(4.0, 100.0), E.LowPass(0.6), E.Gain(0.7)]
x = torch.empty(B * N, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for form in ("chain", "graph"):
    eng = E.Engine(N, B, link_flags=E.LINK_INTERNAL | E.LINK_INPUT if form == "chain" else 0, tile_channels=256)
    if form == "chain":
        eng.set_chain(chain)
    else:
        eng.set_graph(chain, [(E.GRAPH_INPUT, 0, E.PORT_MAIN)] + [(i, i + 1, E.PORT_MAIN) for i in range(len(chain))])
    eng.fill_noise(x, B, 0)
    eng.tune_placement(x, y, B)
    for _ in range(20): eng.process(x, out=y, n_frames=B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): eng.process(x, out=y, n_frames=B)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print("%s: %.4f ms/block, %.3e samples/s" % (form, ms, N * B / ms * 1e3))
    print("  " + " | ".join(l for l in eng.describe().splitlines() if l.startswith("stage")))
    eng.close()
