#!/usr/bin/env python3
"""Standard chain kernel, which (frames per chunk, channels per lane) at mid channel counts -- between the time-sliced kernels and
the large regime: every compiled variant forced in turn, placement tuned against the buffers in use, two fresh engines each;
us per 128-frame block (the engine's own choice first)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from chains import chain3, chain5
fx = load_package()
variants = ("", "ts=0,f=8,cpl=1", "ts=0,f=8,cpl=2", "ts=0,f=16,cpl=1", "ts=0,f=16,cpl=2", "ts=0,f=32,cpl=1", "ts=0,f=8,cpl=4")
for which, mk in (("chain5", lambda: chain5(fx, 24000)), ("chain3", lambda: chain3(fx, 24000))):
    for N in (114688, 131072, 163840, 196608, 262144):
        cells = []
        for var in variants:
            if var: os.environ["DSPFX_VARIANT"] = var
            else: os.environ.pop("DSPFX_VARIANT", None)
            ts_ = []
            name = "?"
            for rep in range(2):
                eng = fx.Engine(N, 128, link_flags=3, tile_channels=256)
                eng.set_chain(mk())
                s = torch.cuda.Stream()
                xs = [torch.empty(128 * N, device="cuda") for _ in range(2)]
                for k, x in enumerate(xs): eng.fill_noise(x, 128, k * 128, 1, s.cuda_stream)
                y = torch.empty_like(xs[0])
                eng.tune_placement(xs[0], y, 128)
                for k in range(300): eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for k in range(1000): eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
                e1.record(s); torch.cuda.synchronize()
                ts_.append(e0.elapsed_time(e1) * 1e3 / 1000)
                name = [l for l in eng.describe().splitlines() if l.startswith("stage")][-1].split("kernel ")[1].split(" ")[0]
                eng.close(); del eng, xs, y
            cells.append("%s %.1f/%.1f" % (name.split("_", 1)[1], ts_[0], ts_[1]))
        print("%s N %6d | %s" % (which, N, " | ".join(cells)), flush=True)
os.environ.pop("DSPFX_VARIANT", None)
