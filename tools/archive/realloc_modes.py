#!/usr/bin/env python3
"""Is the speed mode a property of the process or of an engine's allocations?  One process, fixed sample buffers,
the chain is re-installed (state arrays, mix partials and the delay ring are re-allocated and re-tuned) several
times; each time a full ring revolution is timed.  Optionally (--fresh-io) the in/out buffers are re-allocated too."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--fresh-io", action="store_true")
ap.add_argument("--tune", action="store_true", help="dspfx_tune_placement with the real x / y after every install")
ap.add_argument("--new-engine", action="store_true", help="destroy and re-create the whole engine each round")
args = ap.parse_args()
pkg = load_package()
N, B, D = 1 << 20, 128, 24000
chain = chains.chain5(pkg, D)
eng = pkg.Engine(N, B, link_flags=3, tile_channels=256)
x = torch.empty(B * N, dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
eng.fill_noise(x, B, 0)
def measure():
    steps = 192
    for _ in range(200): eng.process(x, out=y, n_frames=B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): eng.process(x, out=y, n_frames=B)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
for r in range(args.rounds):
    if args.new_engine and r:
        eng.close(); eng = pkg.Engine(N, B, link_flags=3, tile_channels=256)
    eng.set_chain(chain)
    if args.fresh_io and r:
        x = torch.empty(B * N, dtype=torch.float32, device="cuda"); y = torch.empty_like(x); eng.fill_noise(x, B, 0)
    a = measure()
    if args.tune:
        eng.tune_placement(x, y, B)
    b = measure()
    ring = [l for l in eng.describe().splitlines() if "ring" in l]
    print("round %d: %.4f %.4f ms/step  %s" % (r, a, b, ring[-1][-45:] if ring else ""), flush=True)
