#!/usr/bin/env python3
"""Paced real-time run of config 5's shard (one 128-frame block every 2.667 ms, bus of the same block) while a second thread
moves the Reverb node's `seconds` slider -- Reverb::refresh_seconds, reverb.rs:55-71, which the generated render() calls on every
frame a drag changes the value (dsp-stuff-derive/src/lib.rs:560-568).  VERDICT r04 #1: deadlines across a shrink, a growth
within the ring's capacity, a growth beyond it (the storing thread allocates), and a 60 Hz drag.

    python tools/r05_seconds_drag_paced.py [--channels 1048576] [--out gpurun_out/r05_seconds_drag_paced.json]
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=1 << 20)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_seconds_drag_paced.json"))
    args = ap.parse_args()
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    import chains
    pkg = load_package()
    N, B, tile = args.channels, 128, 256
    chain = chains.chain5(pkg)
    chain[2] = pkg.Reverb(seconds=0.5, decay=0.5)
    eng = pkg.Engine(N, B, link_flags=3, tile_channels=tile)
    eng.set_chain(chain)
    eng.kernels_ready(120000)
    s = torch.cuda.Stream(priority=-1)
    torch.cuda.set_stream(s)
    stream = s.cuda_stream
    xs = [torch.empty(B * N, dtype=torch.float32, device="cuda") for _ in range(2)]
    y = torch.empty(B * N, dtype=torch.float32, device="cuda")
    mixes = [torch.zeros(B, dtype=torch.float32, device="cuda") for _ in range(4)]
    for j, x in enumerate(xs):
        eng.fill_noise(x, B, j * B, 0x5EED0001, stream)
    eng.tune_placement(xs[0], y, B, stream=stream)
    for k in range(400):                                   # settle: more than one revolution of the ring, back to back
        eng.process_bus(xs[k & 1], y, mixes[k & 3], B, n_connected=N, stream=stream)
    torch.cuda.synchronize()

    period = B / 48000.0
    # the script of the other thread: (seconds after the start, what, slider values)
    plan = [
        (1.0, "shrink 0.5 -> 0.25 s (24000 -> 12000 samples)", [0.25]),
        (2.0, "growth within capacity 0.25 -> 0.5 s", [0.5]),
        (3.0, "growth BEYOND capacity 0.5 -> 0.75 s (36000 samples: %d more groups of %d MiB, allocated by the storing thread)"
         % (282 - 188, (128 * N * 4) >> 20), [0.75]),
        (5.0, "shrink 0.75 -> 0.5 s", [0.5]),
        (5.5, "growth within capacity 0.5 -> 0.75 s", [0.75]),
        (6.5, "60 Hz drag 0.75 -> 0.30 -> 0.75 s, 120 stores", [0.75 - 0.45 * (1 - abs(1 - i / 60.0)) for i in range(1, 121)]),
    ]
    total_s = 9.5
    n = int(total_s / period)
    events = []
    t_start = [0.0]
    started = threading.Event()

    def gui():
        started.wait()
        for at, what, values in plan:
            calls = []
            for i, v in enumerate(values):
                due = t_start[0] + at + i / 60.0
                while time.perf_counter() < due:
                    time.sleep(2e-4)
                t0 = time.perf_counter()
                seq = eng.set_param_seq(2, 1, float(np.float32(v)))
                calls.append(((time.perf_counter() - t0) * 1e3, seq, t0 - t_start[0]))
            events.append({"at_s": at, "what": what, "stores": len(values), "store_call_ms_max": max(c[0] for c in calls),
                           "store_call_ms_median": float(np.median([c[0] for c in calls])), "first_seq": calls[0][1], "last_seq": calls[-1][1]})

    th = threading.Thread(target=gui)
    th.start()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    lat, lag, sub = np.empty(n), np.empty(n), np.empty(n)
    import gc
    gc.disable()
    t_next = time.perf_counter() + 0.01
    t_start[0] = t_next
    started.set()
    for k in range(n):
        while True:
            now = time.perf_counter()
            if now >= t_next:
                break
            if t_next - now > 3e-4:
                time.sleep(1e-4)
        t_sub = time.perf_counter()
        ev[k][0].record()
        eng.process_bus(xs[k & 1], y, mixes[k & 3], B, n_connected=N, stream=stream)
        sub[k] = time.perf_counter() - t_sub
        ev[k][1].record()
        while not ev[k][1].query():
            pass
        lat[k] = time.perf_counter() - t_sub
        lag[k] = t_sub - t_next
        t_next += period
    th.join()
    gc.enable()
    gpu = np.array([a.elapsed_time(b) for a, b in ev])
    log = {seq: frame for seq, frame, node, param, value in eng.param_log()}
    blk0 = eng.frames_submitted() // B - n                 # block index of the paced run's first block
    q = lambda v, p: float(np.percentile(v, p))
    for e in events:
        k0 = log.get(e["first_seq"], 0) // B - blk0
        k1 = log.get(e["last_seq"], 0) // B - blk0
        lo, hi = max(0, k0 - 2), min(n, k1 + 40)
        e.update({"landed_at_blocks": [int(k0), int(k1)],
                  "latency_ms_max_around": float(lat[lo:hi].max()) * 1e3, "gpu_ms_max_around": float(gpu[lo:hi].max()),
                  "submit_call_ms_max_around": float(sub[lo:hi].max()) * 1e3,
                  "deadline_misses_around": int((lat[lo:hi] > period).sum())})
    res = {"what": "paced run of config 5's shard with the Reverb seconds slider moved from a second thread (tools/r05_seconds_drag_paced.py)",
           "channels": N, "period_ms": period * 1e3, "blocks": n,
           "latency_ms": {"p50": q(lat, 50) * 1e3, "p99": q(lat, 99) * 1e3, "max": float(lat.max()) * 1e3},
           "gpu_ms": {"p50": q(gpu, 50), "p99": q(gpu, 99), "max": float(gpu.max())},
           "submit_call_ms": {"p50": q(sub, 50) * 1e3, "p99": q(sub, 99) * 1e3, "max": float(sub.max()) * 1e3},
           "timer_lag_ms": {"p50": q(lag, 50) * 1e3, "p99": q(lag, 99) * 1e3, "max": float(lag.max()) * 1e3},
           "deadline_misses": int((lat > period).sum()),
           "stores": events, "plan": eng.describe().strip().split("\n")[1:]}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))
    eng.close()


if __name__ == "__main__":
    main()
