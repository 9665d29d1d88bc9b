#!/usr/bin/env python3
"""Same engines, same buffers: kernel time with the XCD-contiguous block mapping off / on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains
pkg = load_package()
N, B = 1 << 20, 128
D = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 256
var = sys.argv[3] if len(sys.argv) > 3 else "static=1,f=8,cpl=2"
K = 6 if D <= 4096 else 2
os.environ["DSPFX_VARIANT"] = var
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
x = torch.empty(B * N, dtype=torch.float32, device=dev)
y = torch.empty(B * N, dtype=torch.float32, device=dev)
def timeit(e, steps=32):
    for _ in range(2): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(1)
    for _ in range(steps): e.process(x, out=y, n_frames=B, stream=stream)
    torch.cuda.synchronize(); e.profile_enable(0)
    ms, n, _ = e.profile_read(); return ms / n
engs = []
for k in range(K):
    e = pkg.Engine(N, B, link_flags=3, tile_channels=tile); e.set_chain(chains.chain5(pkg, D))
    e.profile_enable(64); e.profile_enable(0)
    if not engs: e.fill_noise(x, B, 0)
    for _ in range(min(D // B + 2, 200)): e.process(x, out=y, n_frames=B, stream=stream)
    engs.append(e)
for rnd in range(2):
    for mode in ("0", "1"):     # identity / XCD-contiguous (the hashed and bit-reversed mappings were measured equal and removed)
        os.environ["DSPFX_XCD_REMAP"] = mode
        print("D=%d tile=%d %s remap=%s: " % (D, tile, var, mode) + " ".join("%.4f" % timeit(e) for e in engs))
