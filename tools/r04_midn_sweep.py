#!/usr/bin/env python3
"""Chain kernels between the few-channel and the million-channel regime: us per 128-frame block and fraction of the
8 TB/s HBM peak at every multiple of 16384 channels, for the kernel the engine picks by default and for forced alternatives.
usage: r04_midn_sweep.py [chain3|chain5|both] [max channels]     (no bus, tiled-256 layout, D = 24000, 1500 blocks back to back)
       r04_midn_sweep.py table [chain3|chain5]                  the default kernel at every multiple of 16384 channels up to 1 M, after
                                                                dspfx_tune_placement (as bench.py does), by step"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from chains import chain3, chain5
dspfx = load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
nmax = int(sys.argv[2]) if len(sys.argv) > 2 and which != "table" else 262144
blocks = 1500
ARMS = [("default", None), ("ts c1", "ts=1,cpl=1"), ("ts c2", "ts=1,cpl=2"),
        ("std f32 c1", "ts=0,f=32,cpl=1"), ("std f16 c1", "ts=0,f=16,cpl=1"), ("std f8 c1", "ts=0,f=8,cpl=1"), ("std f8 c2", "ts=0,f=8,cpl=2")]
if os.environ.get("SWEEP_ARMS"):
    ARMS = [a for a in ARMS if a[0] in os.environ["SWEEP_ARMS"].split(";")]


def run(mk, N, var):
    if var is None:
        os.environ.pop("DSPFX_VARIANT", None)
    else:
        os.environ["DSPFX_VARIANT"] = var
    try:
        eng = dspfx.Engine(N, 128, link_flags=3, tile_channels=256)
        eng.set_chain(mk())
    except Exception as ex:
        return None, str(ex)[:40]
    s = torch.cuda.Stream()
    xs = [torch.empty(128 * N, device="cuda") for _ in range(2)]
    for k, x in enumerate(xs):
        eng.fill_noise(x, 128, k * 128, 1, s.cuda_stream)
    y = torch.empty(128 * N, device="cuda")
    for k in range(300):
        eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for k in range(blocks):
        eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    kern = [l for l in eng.describe().splitlines() if l.startswith("stage")][-1]
    name = kern.split("time-sliced ")[1].split(")")[0].split(";")[0] if "time-sliced" in kern else kern.split("fused kernel ")[1].split(" ")[0]
    us = e0.elapsed_time(e1) * 1e3 / blocks
    eng.close()
    return us, name


def table(cname):
    mkc, bps = (chain3, 16.25) if cname == "chain3" else (chain5, 16.5)
    print("== %s, default kernel, tuned placement: channels | us per block by step, launch gaps included (frac of 8 TB/s) | kernel" % cname, flush=True)
    for N in list(range(32768, 262144 + 1, 16384)) + list(range(393216, 1048576 + 1, 131072)):
        D = 24000 if N <= 262144 else 4800           # ring bytes moved per block do not depend on D; 1 M channels x 24000 rows would be 96 GB
        eng = dspfx.Engine(N, 128, link_flags=3, tile_channels=256)
        eng.set_chain(mkc(dspfx, D))
        eng.kernels_ready(120000)
        s = torch.cuda.Stream()
        xs = [torch.empty(128 * N, device="cuda") for _ in range(2)]
        for k, x in enumerate(xs):
            eng.fill_noise(x, 128, k * 128, 1, s.cuda_stream)
        y = torch.empty(128 * N, device="cuda")
        eng.tune_placement(xs[0], y, 128, stream=s.cuda_stream)
        nb = 1500 if N <= 262144 else 400
        for k in range(200):
            eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for k in range(nb):
            eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / nb
        floor = bps * N * 128 / 8e6
        kern = [l for l in eng.describe().splitlines() if l.startswith("stage")][-1]
        name = kern.split("time-sliced ")[1].split(")")[0].split(";")[0] if "time-sliced" in kern else kern.split("fused kernel ")[1].split(" ")[0]
        print("N %7d | %6.1f (%.3f) | %s" % (N, us, floor / us, name), flush=True)
        eng.close()
        del xs, y
        torch.cuda.empty_cache()


if which == "table":
    table(sys.argv[2] if len(sys.argv) > 2 else "chain3")
    sys.exit(0)
for cname, mk, bps, wcap in (("chain3", lambda: chain3(dspfx, 24000), 16.25, 4), ("chain5", lambda: chain5(dspfx, 24000), 16.5, 5)):
    if which not in ("both", cname):
        continue
    print("== %s: us per block incl. launch gaps (fraction of 8 TB/s) [kernel]" % cname, flush=True)
    for N in range(int(os.environ.get("SWEEP_NMIN", "32768")), nmax + 1, 16384):
        row = []
        for label, var in ARMS:
            v = var % wcap if var and "%d" in var else var
            us, name = run(mk, N, v)
            row.append("%s %s" % (label, "-" if us is None else "%.1f (%.3f) [%s]" % (us, bps * N * 128 / us / 8e6, name)))
        print("N %7d | " % N + " | ".join(row), flush=True)
os.environ.pop("DSPFX_VARIANT", None)
