#!/usr/bin/env python3
"""Where the time-sliced kernel stops paying: channel counts around the engine's thresholds, both kernels forced in turn
(DSPFX_VARIANT ts=1 / ts=0), placement tuned against the buffers in use, three fresh engines each; us per 128-frame block."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from chains import chain3, chain5
fx = load_package()
for which, mk in (("chain3", lambda: chain3(fx, 24000)), ("chain5", lambda: chain5(fx, 24000))):
    for N in (57344, 65536, 73728, 81920, 90112, 98304, 114688, 131072):
        cols = []
        for ts in (1, 0):
            os.environ["DSPFX_VARIANT"] = "ts=%d" % ts
            ts_times = []
            for rep in range(3):
                eng = fx.Engine(N, 128, link_flags=3, tile_channels=256)
                eng.set_chain(mk())
                s = torch.cuda.Stream()
                xs = [torch.empty(128 * N, device="cuda") for _ in range(2)]
                for k, x in enumerate(xs): eng.fill_noise(x, 128, k * 128, 1, s.cuda_stream)
                y = torch.empty_like(xs[0])
                eng.tune_placement(xs[0], y, 128)
                for k in range(400): eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for k in range(1500): eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
                e1.record(s); torch.cuda.synchronize()
                ts_times.append(e0.elapsed_time(e1) * 1e3 / 1500)
                eng.close(); del eng, xs, y
            cols.append(ts_times)
        print("%s N %6d   time-sliced %s   standard %s" % (which, N, " ".join("%6.2f" % t for t in cols[0]), " ".join("%6.2f" % t for t in cols[1])), flush=True)
os.environ.pop("DSPFX_VARIANT", None)
