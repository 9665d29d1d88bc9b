#!/usr/bin/env python3
"""Which buffers decide the two speeds seen above 131072 channels: the engine's rings or the caller's sample blocks?
usage: r04_midn_noise2.py N"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from chains import chain3
dspfx = load_package()
hip = C.CDLL("libamdhip64.so")
N = int(sys.argv[1])
s = torch.cuda.Stream()


def engine():
    eng = dspfx.Engine(N, 128, link_flags=3, tile_channels=256)
    eng.set_chain(chain3(dspfx, 24000))
    eng.kernels_ready(60000)
    return eng


def timeit(eng, xs, y, blocks=1500):
    for k in range(200):
        eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for k in range(blocks):
        eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
    e1.record(s)
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / blocks, 1)


def bufs(eng):
    xs = [torch.empty(128 * N, device="cuda") for _ in range(2)]
    for k, x in enumerate(xs):
        eng.fill_noise(x, 128, k * 128, 1, s.cuda_stream)
    return xs, torch.empty(128 * N, device="cuda")


def raw(nbytes, contig):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(0x4 if contig else 0)) if contig else hip.hipMalloc(C.byref(p), C.c_size_t(nbytes))
    assert rc == 0, rc
    return p.value


E = engine()
xs, y = bufs(E)
print("N", N, "floor(1.0 of 8 TB/s) %.1f us" % (16.25 * 128 * N / 8e6))
print("same engine, same blocks, 3x     :", [timeit(E, xs, y) for _ in range(3)], flush=True)
r, keep = [], []
for k in range(6):
    keep.append(torch.empty(int(29e6) * (k + 1), device="cuda"))
    xs2, y2 = bufs(E)
    r.append(timeit(E, xs2, y2))
    keep += xs2 + [y2]
print("same engine, new torch blocks    :", r, flush=True)
r = []
for k in range(6):
    ps = [raw(512 * N, False) for _ in range(3)]
    for j in range(2):
        E.fill_noise(ps[j], 128, j * 128, 1, s.cuda_stream)
    r.append((timeit(E, ps[:2], ps[2]), [hex(p & 0xffffff) for p in ps]))
print("same engine, new hipMalloc blocks:", r, flush=True)
r = []
for k in range(6):
    ps = [raw(512 * N, True) for _ in range(3)]
    for j in range(2):
        E.fill_noise(ps[j], 128, j * 128, 1, s.cuda_stream)
    r.append(timeit(E, ps[:2], ps[2]))
print("same engine, contiguous blocks   :", r, flush=True)
r = []
for k in range(6):
    E2 = engine()
    r.append(timeit(E2, xs, y))
    keep.append(torch.empty(int(31e6) * (k + 1), device="cuda"))
    E2.close()
print("new engine, same blocks          :", r, flush=True)
os.environ["DSPFX_CONTIG"] = "1"
r = []
for k in range(6):
    E2 = engine()
    r.append(timeit(E2, xs, y))
    E2.close()
print("new engine (contig rings), same  :", r, flush=True)
r = []
ps = [raw(512 * N, True) for _ in range(3)]
for j in range(2):
    E.fill_noise(ps[j], 128, j * 128, 1, s.cuda_stream)
for k in range(6):
    E2 = engine()
    r.append(timeit(E2, ps[:2], ps[2]))
    E2.close()
print("contig rings + contig blocks     :", r, flush=True)
del os.environ["DSPFX_CONTIG"]
t0 = timeit(E, xs, y)
E.tune_placement(xs[0], y, 128, stream=s.cuda_stream)
print("tune_placement: before", t0, "after", timeit(E, xs, y), flush=True)
