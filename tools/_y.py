import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from chains import chain3, chain5
dspfx = load_package()
blocks = 2000
def mk_generic():
    return [dspfx.Gain(0.8), dspfx.LowPass(0.3), dspfx.Distort(2.0, dspfx.TANH), dspfx.Reverb(delay_samples=24000, decay=0.5), dspfx.HighPass(0.6)]
for which, mk in (("chain3 (compiled shape)", lambda: chain3(dspfx, 24000)), ("generic 5 nodes (interpreter)", mk_generic)):
    for N in (64, 1024, 4096, 16384, 32768):
        eng = dspfx.Engine(N, 128, link_flags=3, tile_channels=0)
        eng.set_chain(mk())
        s = torch.cuda.Stream()
        x = torch.empty(128 * N, device="cuda"); eng.fill_noise(x, 128, 0, 1, s.cuda_stream)
        y = torch.empty(128 * N, device="cuda")
        eng.profile_enable(blocks + 400) if False else None
        for k in range(300):
            eng.process(x, out=y, n_frames=128, stream=s.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for k in range(blocks):
            eng.process(x, out=y, n_frames=128, stream=s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        kern = [l for l in eng.describe().splitlines() if l.startswith("stage")][-1]
        print("%-30s N %6d  %7.2f us/block   %s" % (which, N, e0.elapsed_time(e1) * 1e3 / blocks, kern[:100]), flush=True)
        del eng
