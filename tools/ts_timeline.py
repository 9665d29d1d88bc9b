#!/usr/bin/env python3
"""Phase timeline of the time-sliced chain kernel (config 2), from a debug build with -DDSPFX_TS_TRACE
(dsp-stuff_amd/csrc/libdspfx_tstrace.so; see chain_kernels.hip.h).  Per wave: 100 MHz wall-clock stamps at entry, after the
loads are issued, before each node, before / after the output stores, at exit.  Prints percentiles over all waves, relative to
the earliest entry of the launch, per time slice q."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DSPFX_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsp-stuff_amd", "csrc", "libdspfx_tstrace.so"))
import torch
from __graft_entry__ import load_package
fx = load_package()
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
which = sys.argv[2] if len(sys.argv) > 2 else "chain3"        # chain3 (config 2's) | chain5 (the headline chain)
B = 128
lib = ctypes.CDLL(os.environ["DSPFX_LIB"])
trace_fn = lib.dspfx_debug_ts_trace if which == "chain3" else lib.dspfx_debug_ts_trace5
trace_fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
cfg = dict(bench.CONFIGS["cfg2" if which == "chain3" else "cfg5"]); cfg["channels"] = N
chain = bench.build_chain(fx, cfg)
eng = fx.Engine(N, B, link_flags=3, tile_channels=256)
eng.set_chain(chain)
x = torch.empty(B * N, device="cuda"); y = torch.empty_like(x)
eng.fill_noise(x, B, 0)
for _ in range(300):
    eng.process(x, out=y, n_frames=B)
torch.cuda.synchronize()
trace_fn(None, 0, 1)
eng.process(x, out=y, n_frames=B)
torch.cuda.synchronize()
G = min(4096, N // 64)
buf = np.zeros(G * 4 * 16, np.uint64)
assert trace_fn(buf.ctypes.data, buf.size, 0) == 0
t = buf.reshape(G, 4, 16).astype(np.float64)
valid = t[:, :, 0] > 0
t0 = t[:, :, 0][valid].min()
us = (t - t0) / 100.0
if which == "chain3":
    names = {0: "entry", 1: "loads issued", 2: "node0 (gain) begins", 3: "node1 (biquad) begins", 4: "node2 (delay) begins", 12: "nodes done", 13: "out stores issued", 14: "exit"}
    spans = ((0, 1, "issue loads"), (1, 2, "(address math)"), (2, 3, "gain incl. wait for the block's samples"), (3, 4, "biquad incl. turns"), (4, 12, "delay incl. wait for taps + ring stores"), (12, 13, "issue out stores"), (13, 14, "exit"))
else:
    names = {0: "entry", 1: "loads issued", 2: "node0 (biquad) begins", 3: "node1 (softclip) begins", 4: "node2 (delay) begins", 5: "node3 (biquad) begins", 6: "node4 (gain) begins", 12: "nodes done", 13: "out stores issued", 14: "exit"}
    spans = ((0, 1, "issue loads"), (1, 2, "(address math)"), (2, 3, "biquad incl. wait for the samples + turns"), (3, 4, "softclip"), (4, 5, "delay incl. wait for taps + ring stores"), (5, 6, "second biquad incl. turns"), (6, 12, "gain"), (12, 13, "issue out stores"), (13, 14, "exit"))
print("N = %d, %d workgroups of 4 waves; microseconds after the first wave's entry: min / p10 / median / p90 / max" % (N, G))
for q in range(4):
    print(" slice q = %d" % q)
    for k, nm in names.items():
        v = us[:, q, k][valid[:, q]]
        print("   %-24s %6.2f %6.2f %6.2f %6.2f %6.2f" % (nm, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
d = us[:, :, 14] - us[:, :, 0]
print(" wave lifetime: median %.2f  p90 %.2f  max %.2f" % (np.median(d[valid]), np.percentile(d[valid], 90), d[valid].max()))
for a, b, nm in spans:
    dd = (us[:, :, b] - us[:, :, a])[valid]
    print("   %-44s median %6.2f  p90 %6.2f" % (nm, np.median(dd), np.percentile(dd, 90)))
