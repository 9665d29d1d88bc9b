#!/usr/bin/env python3
"""Long soak of the bus that is finished inside the chain launch (dspfx_process_bus / dspfx_process(mix)): tens of thousands
of blocks back to back at full size, with and without other kernels hammering HBM on a second stream, every block's bus
compared bit for bit with the stand-alone reduction kernels (DSPFX_MIX_TAIL=0) run on a second engine over the same input.
A stale read across the XCDs' L2s, a lost ticket or a counter that was not reset would show as a differing row.

usage: r03_tail_soak.py [blocks]          (writes nothing; prints one line per shape and `failures: n`)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from chains import chain3, chain5
dspfx = load_package()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
SEED = 0x5EED0001
failures = 0


def run(N, nf, make_chain, tile, tail, hammer, n_in=8):
    os.environ["DSPFX_MIX_TAIL"] = "1" if tail else "0"
    eng = dspfx.Engine(N, nf, link_flags=3, tile_channels=tile)
    eng.set_chain(make_chain())
    s = torch.cuda.Stream()
    xs = [torch.empty(nf * N, device="cuda") for _ in range(n_in)]
    for k, x in enumerate(xs):
        eng.fill_noise(x, nf, k * nf, SEED, s.cuda_stream)
    y = torch.empty(nf * N, device="cuda")
    bus = torch.zeros((blocks, nf), device="cuda")
    s2 = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device="cuda") if hammer else None
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(blocks):
        if hammer and k % 7 == 0:
            with torch.cuda.stream(s2):
                junk.mul_(1.0001)
        eng.process(xs[k % n_in], out=y, mix=bus[k], n_frames=nf, stream=s.cuda_stream)
    torch.cuda.synchronize()
    dt = time.time() - t0
    desc = eng.describe() if hasattr(eng, "describe") else ""
    del eng
    return bus, y.clone(), dt, desc


shapes = [
    ("config 5 shard", 1 << 20, 128, lambda: chain5(dspfx, 384), 256),
    ("config 5 shard, frame-major", 1 << 20, 128, lambda: chain5(dspfx, 384), 0),
    ("config 2 (time-sliced)", 1 << 16, 128, lambda: chain3(dspfx, 256), 256),
    ("ragged 1000003 ch", 1000003, 128, lambda: chain5(dspfx, 256), 0),
    ("B = 256", 1 << 19, 256, lambda: chain5(dspfx, 512), 256),
    ("few rows (4099 ch)", 4099, 128, lambda: chain3(dspfx, 128), 0),
]
for name, N, nf, mk, tile in shapes:
    ref, yref, _, _ = run(N, nf, mk, tile, tail=False, hammer=False)
    for hammer in (False, True):
        got, y, dt, desc = run(N, nf, mk, tile, tail=True, hammer=hammer)
        bad_rows = (got.view(torch.int32) != ref.view(torch.int32)).any(dim=1).nonzero().flatten()
        y_same = bool(torch.equal(y.view(torch.int32), yref.view(torch.int32)))
        ok = bad_rows.numel() == 0 and y_same
        failures += 0 if ok else 1
        print("%-28s %s  %6d blocks  %.3f ms/block  bus %s  last out %s" % (
            name, "with HBM traffic on a 2nd stream" if hammer else "alone                           ", blocks,
            dt * 1e3 / blocks, "identical" if bad_rows.numel() == 0 else "DIFFERS at blocks %s" % bad_rows[:8].tolist(),
            "identical" if y_same else "DIFFERS"), flush=True)
    del ref, got
    torch.cuda.empty_cache()
print("failures:", failures)
