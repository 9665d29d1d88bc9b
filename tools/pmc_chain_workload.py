#!/usr/bin/env python3
"""Workload of the counter passes (tools/pmc_chain.sh): for one BASELINE chain config, launches of a kernel whose
traffic is KNOWN -- the empty chain with the same access width (copy_f8_c<cpl>: reads N*B*4 bytes, writes N*B*4 bytes) --
followed by launches of the config's own kernel exactly as bench.py launches it (same engine settings, same bus form).
Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
import bench  # noqa: E402

cfg_name = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = bench.CONFIGS[cfg_name]
pkg = load_package()
N, B = cfg["channels"], cfg["frames"]
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev, priority=-1)
torch.cuda.set_stream(stream)
tile = cfg.get("tile", 256)
eng = pkg.Engine(N, B, link_flags=3, tile_channels=tile)
eng.set_chain(bench.build_chain(pkg, cfg))
desc = eng.describe()
kern = [l for l in desc.splitlines() if l.startswith("stage")][-1]
ts = "time-sliced" in kern and B == 128
name = kern.split("time-sliced ")[1].split(")")[0] if ts else kern.split("kernel ")[1].split(" ")[0]
cpl = int(name.rsplit("_c", 1)[1])
# the calibration kernel: the empty chain at the same channels per lane (same load / store widths)
os.environ["DSPFX_VARIANT"] = "cpl=%d" % cpl
cal = pkg.Engine(N, B, link_flags=3, tile_channels=tile)
cal.set_chain([])
del os.environ["DSPFX_VARIANT"]
xs = [torch.empty(B * N, dtype=torch.float32, device=dev) for _ in range(2)]
y = torch.empty(B * N, dtype=torch.float32, device=dev)
m = torch.zeros(B, dtype=torch.float32, device=dev)
for j, x in enumerate(xs):
    eng.fill_noise(x, B, j * B, stream=stream.cuda_stream)
for k in range(launches):
    cal.process(xs[k & 1], out=y, n_frames=B, stream=stream.cuda_stream)
for k in range(launches):
    if cfg.get("mix", True):
        eng.process_bus(xs[k & 1], y, m, B, n_connected=N, stream=stream.cuda_stream)
    else:
        eng.process(xs[k & 1], out=y, n_frames=B, stream=stream.cuda_stream)
torch.cuda.synchronize()
print("PMCINFO", cfg_name, N, B, name, "copy_f8_c%d" % cpl, eng.algorithmic_bytes_per_sample(B), flush=True)
print(desc, flush=True)
print(cal.describe(), flush=True)
