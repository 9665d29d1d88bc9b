import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import torch, oracle as O
from __graft_entry__ import load_package
from chains import fir_taps
pkg = load_package()
T, N, blocks = 4096, 32, 48
x = O.noise(0x5EED0001, np.arange(N), np.arange(128 * blocks))
ch = [pkg.Fir(fir_taps(T))]
ref = O.run_channels([n.oracle_desc() for n in ch], x, 0)
import os
for k, split in (("1", "1"), ("1", "0"), ("0", "0")):
    os.environ["DSPFX_FIR_KERNEL"] = k
    os.environ["DSPFX_FIR_SPLIT"] = split
    eng = pkg.Engine(N, 128, link_flags=0); eng.set_chain(ch)
    dx = torch.from_numpy(x).cuda(); dy = torch.empty_like(dx)
    for b in range(blocks): eng.process(dx[b*128:(b+1)*128], out=dy[b*128:(b+1)*128])
    torch.cuda.synchronize(); y = dy.cpu().numpy()
    e = y.astype(np.float64) - ref
    for name, sl in (("warm-up", slice(0, 4096)), ("steady", slice(4096, None))):
        print("kernel", k, "split" if split == "1" else "     ", name, "rel RMS %.3g" % (np.sqrt(np.mean(e[sl]**2)) / np.sqrt(np.mean(ref[sl].astype(np.float64)**2))), "max abs %.3g" % np.abs(e[sl]).max(), "peak %.3g" % np.abs(ref[sl]).max())
