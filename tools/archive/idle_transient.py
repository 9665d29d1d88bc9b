#!/usr/bin/env python3
"""What an idle gap does to the chain kernel's launch durations (run under rocprofv3 --kernel-trace).

One cfg5 engine (1 048 576 channels, 5-node chain, tuned placement); bursts of 160 back-to-back launches, each burst
preceded by a host-side idle of a different length.  A cos_ marker kernel follows every burst; tools/idle_report.py
reads the trace and prints the per-launch durations of every burst."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
import chains

pkg = load_package()
dev = torch.device("cuda", 0)
N, B = 1 << 20, 128
eng = pkg.Engine(N, B, link_flags=3, device=0, tile_channels=256)
eng.set_chain(chains.chain5(pkg, 24000))
cs = torch.cuda.Stream(device=dev, priority=-1)
torch.cuda.set_stream(cs)
s = cs.cuda_stream
x = torch.empty(B * N, dtype=torch.float32, device=dev)
y = torch.empty_like(x)
mark = torch.zeros(64, device=dev)
eng.fill_noise(x, B, 0, 0x5EED0001, s)
eng.tune_placement(x, y, 128, stream=s)
eng.fill_noise(x, B, 0, 0x5EED0001, s)
for k in range(1200):       # settle
    eng.process(x, out=y, n_frames=B, stream=s)
torch.cuda.synchronize()
mark.cos_()
idles = [0.0, 0.001, 0.003, 0.01, 0.03, 0.1, 0.3, 1.0, 3.0, 0.0]
log = []
for idle in idles:
    torch.cuda.synchronize()
    if idle:
        time.sleep(idle)
    for k in range(160):
        eng.process(x, out=y, n_frames=B, stream=s)
    torch.cuda.synchronize()
    mark.cos_()
    log.append(idle)
torch.cuda.synchronize()
print(json.dumps({"idles_s": log, "burst": 160}))
