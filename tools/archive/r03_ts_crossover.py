#!/usr/bin/env python3
"""Time-sliced kernel against the standard one by channel count and layout (no bus, blocks of 128 frames back to back,
two rotating inputs): where the crossover lies in the frame-major layout and in the channel-tiled one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from chains import chain3, chain5
dspfx = load_package()
blocks = 2000
print("us per block (GPU events over %d blocks)            time-sliced   standard" % blocks)
for which, mk in (("chain3", lambda: chain3(dspfx, 24000)), ("chain5", lambda: chain5(dspfx, 24000))):
    for N in (32768, 49152, 65536, 81920, 98304, 131072, 196608):
        for tile in (0, 256):
            res = []
            for ts in (1, 0):
                os.environ["DSPFX_VARIANT"] = "ts=%d" % ts
                eng = dspfx.Engine(N, 128, link_flags=3, tile_channels=tile)
                eng.set_chain(mk())
                s = torch.cuda.Stream()
                xs = [torch.empty(128 * N, device="cuda") for _ in range(2)]
                for k, x in enumerate(xs):
                    eng.fill_noise(x, 128, k * 128, 1, s.cuda_stream)
                y = torch.empty(128 * N, device="cuda")
                for k in range(400):
                    eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for k in range(blocks):
                    eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
                e1.record(s)
                torch.cuda.synchronize()
                kern = [l for l in eng.describe().splitlines() if l.startswith("stage")][-1]
                res.append((e0.elapsed_time(e1) * 1e3 / blocks, "ts" if "time-sliced" in kern else "std"))
                del eng
            bps = (16.25 if which == "chain3" else 16.5) * N * 128
            print("%s N %6d %-11s   %7.2f (%s, %.3f)   %7.2f (%s, %.3f)" % (
                which, N, "frame-major" if tile == 0 else "tiled-256", res[0][0], res[0][1], bps / res[0][0] / 8e6,
                res[1][0], res[1][1], bps / res[1][0] / 8e6), flush=True)
os.environ.pop("DSPFX_VARIANT", None)
