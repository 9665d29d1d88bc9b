"""While the background compiler works on one engine's shape, how long do ordinary calls of ANOTHER engine take on the main thread?"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from __graft_entry__ import load_package
E = load_package()
from chains import chain5
B = 128
busy = E.Engine(256, B, link_flags=3); busy.set_chain(chain5(E, 300))          # a compiled-in shape: no compiler involved
x = torch.zeros(B * 256, device="cuda"); y = torch.empty_like(x)
busy.process(x, out=y, n_frames=B); torch.cuda.synchronize()
def probe(label, seconds):
    t_end = time.time() + seconds
    worst = {"process+sync": 0.0, "engine create+chain_set(static)+close": 0.0, "hipMalloc 64MiB+free": 0.0}
    n = 0
    while time.time() < t_end:
        t = time.time(); busy.process(x, out=y, n_frames=B); torch.cuda.synchronize(); worst["process+sync"] = max(worst["process+sync"], time.time() - t)
        t = time.time(); e = E.Engine(128, B, link_flags=3); e.set_chain(chain5(E, 200)); e.close(); worst["engine create+chain_set(static)+close"] = max(worst["engine create+chain_set(static)+close"], time.time() - t)
        t = time.time(); z = torch.empty(64 << 20, dtype=torch.uint8, device="cuda"); del z; torch.cuda.empty_cache(); worst["hipMalloc 64MiB+free"] = max(worst["hipMalloc 64MiB+free"], time.time() - t)
        n += 1
    print(label, "(%d rounds): worst ms:" % n, {k: round(v * 1e3, 2) for k, v in worst.items()}, flush=True)
probe("idle compiler", 1.0)
engs = []
for k in range(4):                       # four new shapes: ~16 kernels for the background thread
    e = E.Engine(1000, B, link_flags=3)
    t = time.time(); e.set_chain([E.Gain(0.3 + 0.1 * k), E.LowPass(0.2), E.HighPass(0.1 * (k + 1)), E.Gain(1.1)] + [E.Gain(0.9)] * k); dt = time.time() - t
    print("chain_set of new shape %d: %.1f ms" % (k, dt * 1e3), flush=True)
    engs.append(e)
probe("compiler busy", 3.0)
for e in engs: e.kernels_ready(60000)
probe("compiler done", 1.0)
