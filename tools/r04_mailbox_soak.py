#!/usr/bin/env python3
"""The mailbox all-reduce under drift: `world` processes on the box's one GPU, `count` exchanges back to back through
dspfx_mix_allreduce, each rank contributing small integers that depend on (rank, exchange, frame) so that the expected sum is known
in closed form and exact in f32; ranks sleep at random (host side) and every 1000th exchange one rank stalls 20 ms, so the others
spin in the kernel.  Every exchange is checked on the device (mismatching frames are counted); block lengths alternate between 128
and 256 frames.   usage: r04_mailbox_soak.py [world] [count]"""
import os, sys, subprocess, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "rank":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from __graft_entry__ import load_package
    fx = load_package()
    rank, world, uid, count = int(sys.argv[2]), int(sys.argv[3]), bytes.fromhex(sys.argv[4]), int(sys.argv[5])
    eng = fx.Engine(64, 256, link_flags=0)
    eng.set_chain([fx.Gain(1.0)])
    comm = fx.Comm(0, world, rank, uid)
    s = torch.cuda.Stream()
    N_total = 64 * world
    div = fx.link_divisor(N_total)
    rng = np.random.default_rng(rank)
    f = torch.arange(256, device="cuda", dtype=torch.float32)
    bad = torch.zeros((), device="cuda", dtype=torch.int64)
    wsum = float(sum(r + 1 for r in range(world)))
    bufs = [torch.empty(256, device="cuda") for _ in range(8)]
    # expected bus of every exchange, divided on the HOST (numpy: IEEE f32 division, what node.rs:189-191 does).  torch's own
    # `tensor / python_float` multiplies by the reciprocal on the device, which differs from a true division for most divisors
    # (world = 3: f32(192.0001) -- 50 of the 97 integers; worlds 2 and 4 happen to agree), so it cannot be the checker.
    ks = np.arange(count, dtype=np.int64)
    bases = np.mod(np.arange(256, dtype=np.float32)[None, :] + ((ks * 131) % 9973).astype(np.float32)[:, None], np.float32(97.0)).astype(np.float32)
    wants = torch.from_numpy((bases * np.float32(wsum)) / np.float32(div)).cuda()
    t0 = time.time()
    with torch.cuda.stream(s):
        for k in range(count):
            B = 128 if k % 3 else 256
            base = torch.remainder(f[:B] + float((k * 131) % 9973), 97.0)            # integers 0..96, the same on every rank
            buf = bufs[k % 8]
            buf[:B] = base * float(rank + 1)
            eng.mix_allreduce(comm, buf, B, N_total, s.cuda_stream)
            bad += (buf[:B] != wants[k, :B]).sum()
            if k % 1000 == 999 and (k // 1000) % world == rank:
                s.synchronize(); time.sleep(0.02)
            elif rng.random() < 0.002:
                time.sleep(float(rng.uniform(0, 0.002)))
    s.synchronize()
    print(json.dumps(dict(rank=rank, bad=int(bad.item()), seconds=round(time.time() - t0, 1), backend=comm.backend)))
    comm.close()
    sys.exit(0)


def run(world=4, count=100000, timeout=3000):
    """Start `world` rank processes (this file with the `rank` argument) on GPU 0; returns one dict per rank (or a string with the
    failing rank's exit status and stderr tail)."""
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    fx = load_package()
    uid = fx.comm_unique_id("mailbox")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "rank", str(r), str(world), uid.hex(), str(count)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    res = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, e = p.communicate()
            res.append("timeout %s" % e[-500:])
            continue
        res.append(json.loads(o.strip().splitlines()[-1]) if p.returncode == 0 and o.strip() else "rc=%d %s" % (p.returncode, e[-500:]))
    print("%d ranks x %d exchanges through the mailboxes:" % (world, count))
    for r in res:
        print("  ", json.dumps(r) if isinstance(r, dict) else r)
    return res


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 100000)
