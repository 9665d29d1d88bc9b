"""Per-block time of a FIR node at 262144 channels for several filter lengths and every sweep (half = the default)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from __graft_entry__ import load_package
fx = load_package()
N, B = 262144, 128
x = torch.empty(B * N, device="cuda"); y = torch.empty_like(x)
for T in (64, 256, 1024, 2048, 4096):
    taps = (np.random.default_rng(T).standard_normal(T) / np.sqrt(T)).astype(np.float64)
    row = []
    for name, prec in (("half", fx.FIR_PRECISION_HALF), ("split", fx.FIR_PRECISION_SPLIT), ("f32", fx.FIR_PRECISION_F32)):
        eng = fx.Engine(N, B, link_flags=3)
        eng.set_chain([fx.Fir(taps)])
        eng.set_fir_precision(0, prec)
        eng.fill_noise(x, B, 0)
        for k in range(T // B + 40):
            eng.process(x, out=y, n_frames=B)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(100):
            eng.process(x, out=y, n_frames=B)
        e1.record(); torch.cuda.synchronize()
        row.append("%s %.3f ms" % (name, e0.elapsed_time(e1) / 100))
        eng.close()
    print("T %5d: %s" % (T, "  ".join(row)), flush=True)
