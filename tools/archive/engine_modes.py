#!/usr/bin/env python3
"""Do engines (i.e. their ring/state allocations) fall into fast/slow classes?  K engines alive at once,
each timed with the same in/out buffers; chains: delay-only, biquad-only, 5-node."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
N, B, D = 1 << 20, 128, 4096
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = 6
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
x = torch.empty(B * N, dtype=torch.float32, device=dev)
y = torch.empty(B * N, dtype=torch.float32, device=dev)
def timeit(e, steps=12):
    for _ in range(2): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(1)
    for _ in range(steps): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(0)
    ms, n, _ = e.profile_read(); return ms / n
os.environ["DSPFX_VARIANT"] = "static=1,f=8,cpl=2"
for name, mk in (("delay-only", lambda: [pkg.Reverb(delay_samples=D, decay=0.5)]),
                 ("2 biquads", lambda: [pkg.BiQuad(), pkg.BiQuad(1, -0.5, 0.1, 0.3, 0.2, 0.1)]),
                 ("5-node", lambda: chains.chain5(pkg, D))):
    engs = []
    for k in range(K):
        e = pkg.Engine(N, B, link_flags=3, tile_channels=tile); e.set_chain(mk())
        e.profile_enable(64); e.profile_enable(0)
        if not engs: e.fill_noise(x, B, 0)
        for _ in range(D // B + 2): e.process(x, out=y, n_frames=B, stream=stream)
        engs.append(e)
    print("%-10s tile %3d: " % (name, tile) + " ".join("%.4f" % timeit(e) for e in engs), "| again:", " ".join("%.4f" % timeit(e) for e in engs[:3]))
    for e in engs: e.close()
