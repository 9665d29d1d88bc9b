#!/usr/bin/env python3
"""Where and when the waves of ONE launch of the standard / software-pipelined chain kernel ran (debug build libdspfx_tstrace.so,
-DDSPFX_TS_TRACE): per wave entry and exit stamps and HW_ID / XCC_ID.  Prints, for the last of 300 back-to-back launches: how many
waves each CU got, how a wave's lifetime depends on how many shared its CU, and when the CUs went idle.
usage: wg_timeline.py N [DSPFX_VARIANT string]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("DSPFX_LIB", os.path.join(ROOT, "dsp-stuff_amd", "csrc", "libdspfx_tstrace.so"))
N = int(sys.argv[1])
if len(sys.argv) > 2:
    os.environ["DSPFX_VARIANT"] = sys.argv[2]
import torch
from __graft_entry__ import load_package
from chains import chain3
fx = load_package()
lib = ctypes.CDLL(os.environ["DSPFX_LIB"])
fn = lib.dspfx_debug_wg_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
eng = fx.Engine(N, 128, link_flags=3, tile_channels=256)
eng.set_chain(chain3(fx, 24000))
eng.kernels_ready(60000)
kern = [l for l in eng.describe().splitlines() if l.startswith("stage")][-1]
xs = [torch.empty(128 * N, device="cuda") for _ in range(2)]
for k, x in enumerate(xs):
    eng.fill_noise(x, 128, k * 128)
y = torch.empty(128 * N, device="cuda")
if os.environ.get("WG_TUNE"):
    eng.tune_placement(xs[0], y, 128)
for k in range(300):
    eng.process(xs[k & 1], out=y, n_frames=128)
torch.cuda.synchronize()
fn(None, 0, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for k in range(20):
    if k == 10:
        e0.record()
    eng.process(xs[k & 1], out=y, n_frames=128)
e1.record()
torch.cuda.synchronize()
W = 32768
buf = np.zeros(W * 3, np.uint64)
assert fn(buf.ctypes.data, buf.size, 0) == 0
t = buf.reshape(W, 3)
t = t[t[:, 0] > 0]
t_in = (t[:, 0] - t[:, 0].min()).astype(np.float64) / 100.0
t_out = (t[:, 1] - t[:, 0].min()).astype(np.float64) / 100.0
hw = (t[:, 2] & 0xffffffff).astype(np.int64)
xcc = (t[:, 2] >> 32).astype(np.int64) & 0xf
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(kern)
print("N = %d: %d waves traced; launch (last of 10 back to back) %.1f us per block by events; first entry -> last exit %.1f us" %
      (N, len(t), e0.elapsed_time(e1) * 100, t_out.max()))
ids, counts = np.unique(cuid, return_counts=True)
print("CUs that ran waves: %d; waves per CU: " % len(ids) + ", ".join("%d CUs x %d" % ((counts == c).sum(), c) for c in sorted(set(counts))))
sid, scounts = np.unique(cuid * 4 + simd, return_counts=True)
print("SIMDs that ran waves: %d; waves per SIMD: " % len(sid) + ", ".join("%d x %d" % ((scounts == c).sum(), c) for c in sorted(set(scounts))))
per_cu = dict(zip(ids, counts))
share = np.array([per_cu[c] for c in cuid])
life = t_out - t_in
for c in sorted(set(counts)):
    m = share == c
    print("  waves on CUs with %2d waves: entry median %5.1f (max %5.1f), lifetime median %5.1f p90 %5.1f, exit median %5.1f max %5.1f us" %
          (c, np.median(t_in[m]), t_in[m].max(), np.median(life[m]), np.percentile(life[m], 90), np.median(t_out[m]), t_out[m].max()))
per_s = dict(zip(sid, scounts))
sshare = np.array([per_s[c] for c in cuid * 4 + simd])
for c in sorted(set(scounts)):
    m = sshare == c
    print("  waves on SIMDs with %2d waves: lifetime median %5.1f p90 %5.1f, exit max %5.1f us" % (c, np.median(life[m]), np.percentile(life[m], 90), t_out[m].max()))
edges = np.arange(0, t_out.max() + 5, 5.0)
print("  waves alive at t (us): " + " ".join("%d:%d" % (e, ((t_in <= e) & (t_out > e)).sum()) for e in edges))
print("  per XCC waves: " + " ".join("%d:%d" % (x, (xcc == x).sum()) for x in sorted(set(xcc))))
print("  lifetime by XCC: " + " ".join("%d:%.1f" % (x, np.median(life[xcc == x])) for x in sorted(set(xcc))))
print("  lifetime by SE within XCC: " + " ".join("%d:%.1f" % (x, np.median(life[se == x])) for x in sorted(set(se))))
print("  lifetime by CU index: " + " ".join("%d:%.1f" % (x, np.median(life[cu == x])) for x in sorted(set(cu))))
idx = np.nonzero(buf.reshape(W, 3)[:, 0] > 0)[0]
bins = np.array_split(np.arange(len(idx)), 16)
print("  lifetime by wave index (16 ranges): " + " ".join("%.1f" % np.median(life[b]) for b in bins))
slow = life > np.percentile(life, 90)
print("  slowest 10%%: XCC histogram %s, wave-index histogram (16 ranges) %s" % (np.bincount(xcc[slow], minlength=8).tolist(), [int(slow[b].sum()) for b in bins]))
