#!/usr/bin/env python3
"""How much of the mid-N spread is WHERE the rings landed: the same kernel, a fresh engine each time (new ring allocations),
with and without the setup-time placement tuning (DSPFX_RING_TUNE=1) and with physically contiguous rings (DSPFX_CONTIG=1).
usage: r04_midn_noise.py [reps]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from __graft_entry__ import load_package
    from chains import chain3, chain5
    dspfx = load_package()
    N, reps = int(sys.argv[2]), int(sys.argv[3])
    out = []
    keep = []
    for r in range(reps):
        eng = dspfx.Engine(N, 128, link_flags=3, tile_channels=256)
        eng.set_chain(chain3(dspfx, 24000))
        eng.kernels_ready(60000)
        s = torch.cuda.Stream()
        xs = [torch.empty(128 * N, device="cuda") for _ in range(2)]
        for k, x in enumerate(xs):
            eng.fill_noise(x, 128, k * 128, 1, s.cuda_stream)
        y = torch.empty(128 * N, device="cuda")
        for k in range(300):
            eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for k in range(1500):
            eng.process(xs[k & 1], out=y, n_frames=128, stream=s.cuda_stream)
        e1.record(s)
        torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) * 1e3 / 1500, 1))
        keep.append(torch.empty(int(37e6) * (r + 1), device="cuda"))     # shift where the next engine's rings land
        eng.close()
    print(json.dumps(out))
    sys.exit(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for N in (65536, 98304, 131072, 163840, 196608, 262144):
    for label, env in (("plain", {}), ("tuned", {"DSPFX_RING_TUNE": "1"}), ("contig", {"DSPFX_CONTIG": "1"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child", str(N), str(reps)], env=e, capture_output=True, text=True, timeout=600)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]
        print("N %7d %-6s us/block %s   (1.0 of 8 TB/s = %.1f us)" % (N, label, line, 16.25 * 128 * N / 8e6), flush=True)
