#!/usr/bin/env python3
"""Block time at small channel counts (the reference's own use is a handful of channels): back-to-back blocks of 128 frames,
3-node and 5-node chain, with the same-block bus, per layout.  usage: r03_small_latency.py [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from chains import chain3, chain5
dspfx = load_package()
Ns = [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 4096, 4099, 16384, 65536]
blocks = 3000
for N in Ns:
    for tile in (0, 256):
        if tile and N % tile:
            continue
        for which, mk in (("chain3", lambda: chain3(dspfx, 24000)), ("chain5", lambda: chain5(dspfx, 24000))):
            for mix in (False, True):
                eng = dspfx.Engine(N, 128, link_flags=3, tile_channels=tile)
                eng.set_chain(mk())
                s = torch.cuda.Stream()
                x = torch.empty(128 * N, device="cuda"); eng.fill_noise(x, 128, 0, 1, s.cuda_stream)
                y = torch.empty(128 * N, device="cuda"); m = torch.zeros(128, device="cuda")
                for k in range(300):
                    eng.process(x, out=y, mix=m if mix else None, n_frames=128, stream=s.cuda_stream)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.time()
                e0.record(s)
                for k in range(blocks):
                    eng.process(x, out=y, mix=m if mix else None, n_frames=128, stream=s.cuda_stream)
                e1.record(s)
                t_host = time.time() - t0
                torch.cuda.synchronize()
                print("N %6d tile %3d %s bus %d : %7.2f us/block on the GPU, host submit %6.2f us/block   %s" % (
                    N, tile, which, mix, e0.elapsed_time(e1) * 1e3 / blocks, t_host * 1e6 / blocks, eng.describe().splitlines()[0][:100]), flush=True)
                del eng
