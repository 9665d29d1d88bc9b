#!/usr/bin/env python3
"""bench.py -- headline benchmark of the effect-chain hot path on MI355X.

Metric (BASELINE.json): mono-channel-samples/sec through the 5-node chain at
128-frame blocks on 1/2/4/8 GPUs.  One "step" = one 128-frame block through the
whole chain for every channel this rank owns (inputs resident in HBM), including
the mix-bus reduction (and, for N>1, its RCCL all-reduce over xGMI).

Default workload (per GPU; weak scaling): BASELINE config 5's shard --
1 048 576 channels, biquad(LP 1 kHz) -> distort(SoftClip, 3.0) -> delay(D=24000,
decay .5) -> biquad(HP 80 Hz) -> gain(.5), B=128, link scaling on every hop,
mix bus.  Other BASELINE configs are parity-test cases; `--config` can time them
for DESIGN.md but the driver's line is the default.

Output: ONE JSON line on stdout -- compact_line() of the full record: every contract field as measured, every config's numbers,
under 7 KB so that a driver that keeps the last 8 KB of stdout keeps all of it; the full record (plans, notes, the CPU baseline's
thread legs) goes to stderr ("bench.py detail: ...") and to gpurun_out/bench_detail_<N>gpu.json.

    python bench.py --gpus 1 --steps 200 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 200
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA dense peak
SEED = 0x5EED0001

CONFIGS = {
    # name: (channels per GPU, frames per block, chain, delay, description)
    "cfg5": dict(channels=1 << 20, frames=128, chain="chain5", delay=24000,
                 desc="BASELINE config 5 shard: 1048576 ch/GPU, biquad>softclip>delay(24000)>biquad>gain, B=128, mix bus"),
    # the mix bus is part of config 5 only (BASELINE.json: "... with RCCL xGMI mix-bus all-reduce"); configs 2 - 4 are timed as
    # BASELINE words them, without one (--bus-everywhere / DSPFX_BENCH_BUS_ALL=1 times them with the same-block bus too)
    "cfg3": dict(channels=1 << 20, frames=256, chain="chain5", delay=24000, mix=False,
                 desc="BASELINE config 3: 1048576 ch, 5-node chain, B=256"),
    "cfg2": dict(channels=1 << 16, frames=128, chain="chain3", delay=24000, mix=False,
                 desc="BASELINE config 2: 65536 ch, gain>biquad>delay(24000), B=128"),
    # diagnostics (memory-pattern ceilings of the chain kernel), not BASELINE configs
    "copy": dict(channels=1 << 20, frames=128, chain="copy", delay=0, desc="diagnostic: empty chain (8 B/sample)"),
    "delay": dict(channels=1 << 20, frames=128, chain="delay", delay=24000,
                  desc="diagnostic: delay line only (16 B/sample)"),
    "cfg4": dict(channels=1 << 18, frames=128, chain="fir", delay=0, taps=4096, mix=False,
                 desc="BASELINE config 4: 262144 ch, 4096-tap FIR, B=128"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=200)   # >= 188 blocks fills the 24000-sample rings with real data
    ap.add_argument("--config", default="cfg5", choices=sorted(CONFIGS))
    ap.add_argument("--channels", type=int, default=None, help="override channels per GPU")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--delay", type=int, default=None)
    ap.add_argument("--taps", type=int, default=None)
    ap.add_argument("--no-mix", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="do not time cfg3 / cfg2 / cfg4 after the headline config")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU-baseline duration")
    ap.add_argument("--link-flags", type=int, default=3)
    ap.add_argument("--probe", type=int, default=1,
                    help="also rate 3 x K candidate sample buffers before the timed region (default 1 = off)")
    ap.add_argument("--paced-seconds", type=float, default=5.0,
                    help="length of the paced real-time run of the headline config (0 = skip; --paced runs nothing else)")
    ap.add_argument("--paced", action="store_true", help="only the paced real-time run (plus the short settle before it)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: run the launch / rendezvous / communicator / bus-batching control flow on the host (gloo) and print "
                         "a line marked dry_run -- what tests/test_bench_dryrun_cpu.py executes under torch.distributed.run")
    ap.add_argument("--tile", type=int, default=None,
                    help="channel-tiled HBM layout [N/W][B][W]; 0 = frame-major [B][N].  Default: the config's own `tile` entry, else 256 (the "
                         "engine-native tiling; tools/archive/r03_cfg2_layout_ab.sh, r03_cfg5_layout_ab.sh sweep it)")
    return ap.parse_args()


def build_chain(pkg, cfg):
    from dsp_stuff_amd import workloads as chains      # BASELINE's chains live in the package, not under tests/
    if cfg["chain"] == "chain5":
        return chains.chain5(pkg, cfg["delay"])
    if cfg["chain"] == "chain3":
        return chains.chain3(pkg, cfg["delay"])
    if cfg["chain"] == "copy":
        return []
    if cfg["chain"] == "delay":
        return [pkg.Reverb(delay_samples=cfg["delay"], decay=0.5)]
    if cfg["chain"] == "fir":
        return [pkg.Fir(chains.fir_taps(cfg["taps"]))]
    raise ValueError(cfg["chain"])


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_topology():
    """(physical cores, logical CPUs, sockets) of this host from /proc/cpuinfo."""
    cores, sockets = set(), set()
    phys = core = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                    sockets.add(phys)
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return (len(cores) or logical), logical, (len(sockets) or 1)


def cpu_quota():
    """CPUs this process may use: (affinity mask size, cgroup CPU quota in cores or None)."""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return aff, quota


def cpu_baseline(chain, cfg, link_flags, target_s, with_single=True):
    """The reference CPU path stand-in: the C restatement (oracle/), built -O3
    -march=native -ffp-contract=off ON THIS BOX, channel-at-a-time / node-at-a-time /
    128-frame blocks like node.rs:267-352 (gather, /1.0001, process, scatter per node), threaded
    over channels.  Timed inside the C function (orc_bench_chain) between a start barrier
    and the last thread's finish; node state is cloned once per thread and reset per channel, the input
    is a per-thread table of noise blocks made before the barrier: nothing but the chain is timed.
    Legs: 1, 16, 64, ... threads up to every logical CPU, each sized from a probe for an equal share of
    target_s; `value` is the best leg (all the box can do), `scaling` lists them all."""
    from __graft_entry__ import load_oracle
    import ctypes as C
    O = load_oracle()
    physical, logical, sockets = cpu_topology()
    native = False
    try:   # rebuild for this host's ISA (the in-tree native .so was built elsewhere)
        tmp = tempfile.mkdtemp(prefix="dspfx_cpu_")
        so = os.path.join(tmp, "liboracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
                               "-std=c11", "-shared", "-o", so, os.path.join(ROOT, "oracle", "dspfx_oracle.c"),
                               "-lm", "-lpthread"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = O._bind(C.CDLL(so))
        native = True
    except Exception:
        L = O.lib()
    descs = [n.oracle_desc() for n in chain]
    block = 128
    is_fir = cfg["chain"] == "fir"
    # blocks per channel: more than one revolution of the delay ring (188 blocks at D = 24000), so the ring is read
    # back and not only written; FIR: 48 blocks = 6144 samples, most of them past the 4096-sample fill phase
    nb = 48 if is_fir else 256

    def run(n_channels, threads):
        protos = [O.node_from_desc(d, _lib=L) for d in descs]
        hs = (C.c_void_p * len(protos))(*[p.h for p in protos])
        w = C.c_double()
        L.orc_bench_chain(hs, len(protos), link_flags, SEED, n_channels, nb, block, threads, C.byref(w))
        return max(w.value, 1e-9)

    legs = sorted({t for t in (1, 16, 64, physical, logical) if 1 <= t <= logical})
    share = target_s / (len(legs) + 0.5)
    scaling = []
    for t in legs:
        probe_ch = t * (1 if is_fir else 2)
        tp = run(probe_ch, t)                                   # >= one ring period per channel: the rate transfers
        ch = int(max(t, min(1 << 20, probe_ch * share / tp)))
        ch -= ch % t
        ch = max(ch, t)
        tw = run(ch, t)
        scaling.append({"threads": t, "value": ch * nb * block / tw, "channels": ch, "seconds": round(tw, 2)})
    one = scaling[0]["value"]
    # the same thread counts on a register-only loop: what parallelism does this process really get (CPU quota, SMT)?
    spin = {}
    try:
        w = C.c_double()
        L.orc_bench_spin(1, 20_000_000, C.byref(w))
        iters = int(max(1e6, 0.25 * 20_000_000 / max(w.value, 1e-6)))
        L.orc_bench_spin(1, iters, C.byref(w))
        t1 = w.value
        for t in legs:
            L.orc_bench_spin(t, iters, C.byref(w))
            spin[t] = round(t * t1 / max(w.value, 1e-9), 2)
    except Exception:
        spin = {}
    for e in scaling:
        e["speedup"] = round(e["value"] / one, 2)
        if e["threads"] in spin:
            e["register_loop_speedup"] = spin[e["threads"]]
    affinity, quota = cpu_quota()
    best = max(scaling, key=lambda e: e["value"])
    res = {"value": best["value"], "unit": "samples/s", "cores": best["threads"], "physical_cores": physical,
           "threads": logical, "sockets": sockets, "affinity_cpus": affinity, "cgroup_cpu_quota": quota,
           "cpu_model": cpu_model(), "kind": "port",
           # which object ran: the oracle rebuilt for THIS host's ISA, or -- on a box without gcc -- the prebuilt in-tree one
           "build": ("gcc -O3 -march=native -ffp-contract=off, built on this box" if native else "prebuilt oracle/liboracle.so (-O2 -ffp-contract=off): no compiler on this box"),
           "sample": f"{best['channels']} channels x {nb} blocks of 128 frames on {best['threads']} threads, same chain/params, "
                     f"hashed-noise input table made before the timed region, {best['seconds']} s; "
                     f"oracle/dspfx_oracle.c {'-O3 -march=native' if native else '-O2'} -ffp-contract=off, "
                     f"pthreads over channels (orc_bench_chain); excludes the reference's tokio/ring/pool overhead",
           "scaling": scaling,
           "scaling_note": "speedup = chain samples/s over the one-thread leg; register_loop_speedup = the same thread count on a loop "
                           "that touches no memory: where the two agree the chain scales as far as this process is given CPUs "
                           "(affinity mask / cgroup quota above; the one-thread leg also runs at the single-core boost clock), where "
                           "the chain falls behind the register loop it is bound by the memory system"}
    if with_single:
        res["single_thread"] = {"value": one, "unit": "samples/s", "cores": 1,
                                "sample": f"{scaling[0]['channels']} channels x {nb} blocks, one thread, {scaling[0]['seconds']} s"}
    return res


def make_comm(ctx, eng=None):
    """The mix bus' own communicator behind the C ABI (rank 0 makes the id, torch.distributed only carries the bytes).
    Backends are tried in order -- the one-shot mailbox all-reduce (peer writes over xGMI, rank-ordered sum), then RCCL behind
    the same ABI -- and each is VALIDATED before it is trusted: 64 exchanges of known values, checked on every rank.  Any
    failure on any rank sends ALL ranks on to the next backend, and in the end to torch.distributed's all_reduce; the line
    says which one ran (config.collective): a run on N GPUs must not die on the plumbing of an alternative call path."""
    import torch
    import torch.distributed as dist
    if ctx.comm is not None or getattr(ctx, "comm_fallback", None) or not ctx.use_dist or os.environ.get("DSPFX_BENCH_COMM", "abi") != "abi":
        return
    notes = []
    for backend in [b for b in os.environ.get("DSPFX_BENCH_COMM_BACKENDS", "mailbox,rccl").split(",") if b]:
        err = ""
        idt = torch.zeros(ctx.pkg.COMM_ID_BYTES, dtype=torch.uint8, device=ctx.dev)
        try:
            if ctx.rank == 0:
                idt.copy_(torch.tensor(list(ctx.pkg.comm_unique_id(backend)), dtype=torch.uint8))
        except Exception as ex:
            err = "unique id: %s" % ex
        dist.broadcast(idt, 0)
        comm = None
        if os.environ.get("DSPFX_BENCH_COMM_FAIL") in ("1", backend):       # exercise the fallbacks below
            err = "forced by DSPFX_BENCH_COMM_FAIL"
        elif not bool(idt.any().item()):
            err = err or "rank 0 could not make a unique id"
        else:
            try:
                comm = ctx.pkg.Comm(ctx.local_rank, ctx.world, ctx.rank, bytes(idt.cpu().tolist()))
                if eng is not None and not getattr(ctx, "dry", False):
                    t = torch.empty(128, dtype=torch.float32, device=ctx.dev)
                    for r in range(64):
                        t.fill_(float((ctx.rank + 1) * (r + 1)))
                        eng.mix_allreduce(comm, t, 128, 0, torch.cuda.current_stream().cuda_stream)
                        want = float((r + 1) * ctx.world * (ctx.world + 1) // 2)
                        if not bool((t == want).all().item()):
                            raise RuntimeError("validation exchange %d gave %r, expected %r" % (r, t[:4].tolist(), want))
            except Exception as ex:
                err = "%s: %s" % (backend, ex)
        okt = torch.tensor([0.0 if err else 1.0], device=ctx.dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if okt.item() >= 1.0:
            ctx.comm = comm
            ctx.comm_notes = notes
            break
        if comm is not None:
            try:
                comm.close()
            except Exception:
                pass
        notes.append(err or "%s: another rank failed" % backend)
        print("bench.py rank %d: communicator backend %s unavailable (%s)" % (ctx.rank, backend, notes[-1]), file=sys.stderr)
    if ctx.comm is None:
        ctx.comm_fallback = "; ".join(notes) or "no backend"
        print("bench.py rank %d: C-ABI communicator unavailable (%s); using torch.distributed all_reduce" % (ctx.rank, ctx.comm_fallback),
              file=sys.stderr)
    dist.barrier()
    if not getattr(ctx, "dry", False):
        torch.cuda.synchronize()


class Ctx:
    """What one process shares between the configs it measures."""
    pass


def paced_run(torch, eng, xs, y, mixes, B, total_channels, stream, seconds, pbus=None, start_at=None):
    """Real-time operation: ONE block every block period (B / 48 kHz = 2.667 ms at B = 128) from a host timer, each block
    with the Output node's bus of the SAME block (dspfx_process_bus).  Per block: host time from the submit call to the
    moment the host sees the block's `out` AND `mix` complete (an event polled right behind the launch), and the GPU time
    between two events around the launch.  The chip idles 2.3 ms between blocks: this is the pattern north_star's
    '<128-sample block latency' is about, not the back-to-back throughput of the timed region."""
    import numpy as np
    period = B / 48000.0
    n = max(8, int(seconds / period))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    lat = np.empty(n)
    lag = np.empty(n)
    n_in = len(xs)
    torch.cuda.synchronize()
    import gc
    gc_was = gc.isenabled()
    gc.disable()                                      # a collection in the middle of a block is the host's jitter, not the engine's
    t_next = time.perf_counter() + (0.005 if start_at is None else max(0.005, start_at - time.time()))
    t_begin = t_next
    for k in range(n):
        while True:                                   # sleep most of the gap, spin the last 300 us
            now = time.perf_counter()
            if now >= t_next:
                break
            if t_next - now > 3e-4:
                time.sleep(1e-4)
        t_sub = time.perf_counter()
        ev[k][0].record()
        if pbus is not None:                          # several ranks: the block, then the exchange of its bus, on the compute stream
            pbus.step(xs[k % n_in], y)
        else:
            eng.process_bus(xs[k % n_in], y, mixes[k & 3], B, n_connected=total_channels, stream=stream)
        ev[k][1].record()
        while not ev[k][1].query():
            pass
        lat[k] = time.perf_counter() - t_sub
        lag[k] = t_sub - t_next
        t_next += period
    wall = time.perf_counter() - t_begin
    if gc_was:
        gc.enable()
    gpu = np.array([a.elapsed_time(b) for a, b in ev])
    q = lambda v, p: float(np.percentile(v, p))
    return {"what": "one block per block period from a host timer; bus of the same block (dspfx_process_bus); latency = submit call -> host sees out and mix complete",
            "period_ms": period * 1e3, "blocks": n, "seconds": wall,
            "latency_ms": {"p50": q(lat, 50) * 1e3, "p99": q(lat, 99) * 1e3, "max": float(lat.max()) * 1e3},
            "gpu_ms": {"p50": q(gpu, 50), "p99": q(gpu, 99), "max": float(gpu.max())},
            "timer_lag_ms": {"p50": q(lag, 50) * 1e3, "p99": q(lag, 99) * 1e3, "max": float(lag.max()) * 1e3},
            "deadline_misses": int((lat > period).sum())}


def measure(ctx, args, cfg_name, steps, warmup, overrides=None, extras=False):
    """One configuration: build the engine, settle, W warm-up steps, K timed steps.  Returns a dict.
    extras: also the cold figure (first launches after an idle gap) and the paced real-time run (one block per block period)."""
    import torch
    import torch.distributed as dist
    pkg, P, dev, world = ctx.pkg, ctx.P, ctx.dev, ctx.world
    cfg = dict(CONFIGS[cfg_name])
    cfg.update(overrides or {})
    N, B = cfg["channels"], cfg["frames"]
    use_mix = not args.no_mix and (cfg.get("mix", True) or os.environ.get("DSPFX_BENCH_BUS_ALL") == "1")
    chain = build_chain(pkg, cfg)
    is_fir = cfg["chain"] == "fir"

    shard = P.weak_shard(N, world, ctx.rank)      # weak scaling: N channels on every rank
    tile = args.tile if args.tile is not None else cfg.get("tile", 256)
    eng = pkg.Engine(shard.channels, B, link_flags=args.link_flags, device=ctx.local_rank,
                     channel_offset=shard.offset, tile_channels=tile)
    eng.set_chain(chain)
    eng.kernels_ready(120000)     # setup time: a shape without a compiled-in kernel gets its run-time kernel BEFORE anything is timed
    stream = ctx.compute_stream.cuda_stream

    # ---- every host-side allocation happens HERE, before any settling: the chip's power management reacts to
    # idle gaps of a few tens of ms with a transient of ~50 launches (fast, then 5-15 % slow, then steady:
    # profiles/r02_idle_transient.txt), so nothing may pause the queue between settling and the timed region
    n_in = int(os.environ.get("DSPFX_BENCH_NIN", "2"))

    def alloc_buf(block=None):
        t = torch.empty(B * N, dtype=torch.float32, device=dev)
        if block is not None:
            eng.fill_noise(t, B, block * B, SEED, stream)
        return t

    mixes = [torch.zeros(B, dtype=torch.float32, device=dev) for _ in range(4)] if use_mix else [None] * 4
    total_channels = shard.total_channels
    # Mix bus.  Default ("inline"): the Output node's bus of the SAME block, finished inside the chain launch
    # (dspfx_process_bus: the launch's last workgroups complete the sum and apply the Output hop) -- one launch per block,
    # nothing else on the compute stream, and the bus is ready when the block's samples are (round 3; round 2 reported the
    # "pipe" form, dspfx_process_mixpipe, which delivers the bus two calls late and is 1.6 % faster: timed after the
    # headline region as `bus_two_calls_late`).  Several GPUs: the per-rank sums of BATCH blocks all-reduced in one RCCL
    # call on the second stream (parallel.PipelinedMixBus over either form).
    # DSPFX_BENCH_MIX = inline | pipe | deferred selects the path for A/B runs.
    dist_run = ctx.use_dist or world > 1
    mix_mode = os.environ.get("DSPFX_BENCH_MIX", "inline")
    mix_stream = ctx.mix_stream
    ms = mix_stream.cuda_stream
    # Blocks per exchange.  1 (default since round 4): the GLOBAL bus of every block, exchanged right behind the block's own
    # launch on the compute stream -- the Output node as the reference has it, a few microseconds after the samples.  8 was
    # rounds 1-3's default: one collective per 8 blocks on the second stream, the bus up to 21 ms late (a throughput option).
    BATCH = int(os.environ.get("DSPFX_BENCH_MIX_BATCH", "1"))
    bus_world = 2 if (ctx.use_dist and world == 1) else world          # forced-dist: take the collective path
    bus = P.MixBus(total_channels, B, lambda m, nf, n: eng.mix_finish(m, nf, n, ms), world=bus_world)
    pipe_fill = [0]

    # one full revolution of the delay ring per measurement: ring groups differ in placement quality too, and a
    # probe that only walks a few of them mispredicts the timed region (seen: 0.355 probed, 0.406 timed)
    probe_steps = max(24, -(-int(cfg.get("delay") or 0) // B))

    def probe(xs_, y_, steps_=probe_steps):
        for k in range(4):
            eng.process(xs_[k % len(xs_)], out=y_, n_frames=B, stream=stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(steps_):
            eng.process(xs_[k % len(xs_)], out=y_, n_frames=B, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps_

    probe_log = None
    if args.probe > 1 and not is_fir:
        # Optional rating of candidate sample buffers (--probe K): each is rated on its own, first as the input
        # (fixed output buffer), then the rest as the output (best input).
        K = 3 * args.probe
        bufs = [alloc_buf(i % n_in) for i in range(K)]
        ref_out = K - 1
        t_in = [probe([bufs[i]], bufs[ref_out]) for i in range(K - 1)]
        order = sorted(range(K - 1), key=lambda i: t_in[i])
        ins = order[:n_in]
        outs = [i for i in range(K) if i not in ins]
        t_out = {i: probe([bufs[ins[0]]], bufs[i]) for i in outs}
        out_i = min(outs, key=lambda i: t_out[i])
        xs, y = [bufs[i] for i in ins], bufs[out_i]
        probe_log = {"buffers": K, "as_input_ms": [round(t, 4) for t in t_in], "inputs": ins,
                     "as_output_ms": {str(i): round(t, 4) for i, t in t_out.items()}, "output": out_i}
        del bufs
        torch.cuda.empty_cache()
    else:
        xs, y = [alloc_buf(i) for i in range(n_in)], alloc_buf()
    # Placement tuning against the buffers the timed region will use (dspfx_tune_placement): how fast a ring group
    # streams depends on where it sits relative to the OUTPUT buffer it is streamed with, so the engine re-times
    # its candidate groups with the real chain kernel on these buffers and keeps the fastest.  This is what removes
    # the slow mode (profiles/r01_placement.txt: 0.39-0.40 -> 0.355 ms in every one of 18 installs).  Setup only.
    tune_log = None
    if os.environ.get("DSPFX_BENCH_TUNE", "1") == "1" and not is_fir and (cfg.get("delay") or 0) * N * 4 >= (1 << 30):
        before = probe(xs, y)
        t_tune = time.perf_counter()
        eng.tune_placement(xs[0], y, min(B, 128), stream=stream)
        torch.cuda.synchronize()
        tune_s = time.perf_counter() - t_tune
        tune_log = {"ms_before": round(before, 4), "seconds": round(tune_s, 2)}
    for j, x_ in enumerate(xs):     # block j of the noise stream in input j
        eng.fill_noise(x_, B, j * B, SEED, stream)
    if os.environ.get("DSPFX_BENCH_ZERO_INPUT") == "1":     # diagnosis only (is a kernel power-limited? zeros toggle nothing); reported in config
        with torch.cuda.stream(ctx.compute_stream):
            for x_ in xs:
                x_.zero_()
    if os.environ.get("DSPFX_BENCH_COMM_EARLY", "0") != "1":
        make_comm(ctx, eng)         # after the engine's large allocations and the tuning (A/B: DSPFX_BENCH_COMM_EARLY=1)
    pbus = (P.PipelinedMixBus(eng, total_channels, B, ctx.compute_stream, mix_stream, bus_world, batch=BATCH, device=dev,
                              comm=ctx.comm, same_block=(mix_mode == "inline"))
            if (use_mix and dist_run and mix_mode in ("pipe", "inline")) else None)

    def step(k):
        if not use_mix:
            eng.process(xs[k % n_in], out=y, n_frames=B, stream=stream)
            return
        if pbus is not None:
            pbus.step(xs[k % n_in], y)
            return
        m = mixes[k & 3]
        if mix_mode == "pipe":
            eng.process_mixpipe(xs[k % n_in], y, m, B, n_connected=total_channels, stream=stream)
            pipe_fill[0] += 1
            return
        if mix_mode == "inline" and not dist_run:        # the bus of THIS block, finished inside its own launch
            eng.process_bus(xs[k % n_in], y, m, B, n_connected=total_channels, stream=stream)
            return
        eng.process_partials(xs[k % n_in], out=y, n_frames=B, stream=stream)
        eng.mix_collect(m, B, stream=ms)
        with torch.cuda.stream(mix_stream):
            bus.submit(m)

    def drain():
        if pbus is not None:
            pbus.drain()
            return
        if not use_mix or (mix_mode == "inline" and not dist_run):
            return                                  # nothing is pending: the bus (if any) was finished inside each block's own launch
        if use_mix and mix_mode == "pipe":
            if pipe_fill[0]:
                eng.mixpipe_flush(mixes[2] if pipe_fill[0] >= 2 else None, mixes[3], n_connected=total_channels, stream=stream)
                pipe_fill[0] = 0
            return
        with torch.cuda.stream(mix_stream):
            bus.drain()

    def fence():
        torch.cuda.synchronize()
        if ctx.use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Kernel timing.  Per-launch HIP events drain the queue between kernels (measured: +5..20 us per
    # step), so when the compute stream holds nothing but the dominant kernel (single fused stage, mix
    # bus on its own stream) its average launch duration is taken from ONE event pair around the timed
    # region on that stream (gaps included: a slight under-estimate of the kernel's rate).  Otherwise
    # (several kernels per step) every launch of the dominant stage is bracketed by its own events.
    stage_lines = [l for l in eng.describe().splitlines() if l.startswith("stage")]
    n_stages = len(stage_lines)
    region_timing = n_stages == 1
    if not region_timing:
        eng.profile_enable(steps + 8)
        eng.profile_enable(0)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    es0, es1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    # ---- settle: the step of the timed region, back to back, for >= one revolution of the delay rings (they then
    # hold real data) and >= SETTLE seconds of GPU time, so that clocks / power are in their steady state.  The W
    # warm-up steps follow without any host-side pause, then drain + fence (microseconds), then the timed region.
    settle_s = float(os.environ.get("DSPFX_BENCH_SETTLE", "0.3"))
    for k in range(8):
        step(k)
    es0.record()
    for k in range(16):
        step(k)
    es1.record()
    es1.synchronize()
    est_ms = max(es0.elapsed_time(es1) / 16, 1e-3)
    if ctx.use_dist:     # every rank must run the SAME number of steps: the batched bus collectives are counted in steps
        t_est = torch.tensor([est_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t_est, op=dist.ReduceOp.MAX)
        est_ms = float(t_est.item())
    settle_steps = int(max(probe_steps + 8, min(20000, settle_s * 1e3 / est_ms)))
    es0.record()
    for k in range(settle_steps):
        step(k)
    es1.record()
    for k in range(warmup):
        step(k)
    drain()
    fence()
    settle_ms = es0.elapsed_time(es1) / settle_steps
    if tune_log is not None:
        tune_log["ms_after"] = round(settle_ms, 4)
    if not region_timing:
        eng.profile_enable(1)
    # one-workgroup marker kernels (sin_ before, cos_ after) delimit the timed region in a rocprofv3 kernel trace
    # (tools/trace_phases.py).  Both are outside the host-clock window AND off the GPU's timeline of the region: the first has
    # finished before the clock starts (it used to be queued right in front of the first step, whose launch then waited for it),
    # the second is queued after the clock has stopped (round 5: VERDICT r04 #5, the fixed cost of a 20-launch region).
    def timed_region(step_fn, drain_fn, e0, e1):
        """The contract's timed region: barrier + synchronize, K steps, barrier + synchronize; returns the host-clock seconds (this rank),
        the submit time and the moment the region's closing event was seen.  What is still pending behind the K-th step (the pipelined
        forms' flush and last exchange) is SUBMITTED before the host waits for anything (round 6, ADVICE r05: the wait used to come first,
        which serialised the flush's launch latency behind an idle device)."""
        ctx.mark.sin_()
        fence()
        t0_ = time.perf_counter()
        e0.record()
        for k in range(steps):
            step_fn(k)
        e1.record()
        t_sub_ = time.perf_counter() - t0_     # host-side submission time of the K steps
        drain_fn()
        e1.synchronize()                       # a spinning wait on the region's own event: the synchronize below finds an idle device
        t_evt_ = time.perf_counter() - t0_
        fence()
        dt_ = time.perf_counter() - t0_
        ctx.mark.cos_()
        torch.cuda.synchronize()
        return dt_, t_sub_, t_evt_

    dt, t_submitted, t_event = timed_region(step, drain, ev0, ev1)
    region_ms = ev0.elapsed_time(ev1)
    if region_timing:
        kern_ms_total, kern_launches = region_ms, steps
        kern_name = stage_lines[-1].split("kernel ")[1].split(" ")[0]
        if "time-sliced " in stage_lines[-1] and B == 128:       # whole 128-frame blocks of a few-channel engine run the time-sliced kernel
            kern_name = stage_lines[-1].split("time-sliced ")[1].split(")")[0]
        kern_method = "one HIP-event pair around the timed region on the compute stream / launches"
    else:
        eng.profile_enable(0)
        kern_ms_total, kern_launches, kern_name = eng.profile_read()
        kern_method = "HIP events around every launch of the dominant stage"

    if ctx.use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    samples = float(total_channels) * B * steps
    value = samples / dt
    bps = eng.algorithmic_bytes_per_sample(B)
    kern_ms = kern_ms_total / max(kern_launches, 1)
    if is_fir:
        flops = 2.0 * cfg["taps"] * N * B
        achieved = flops / (kern_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F32_PEAK_TFLOPS}
        if kern_name == "fir_split_kernel":
            # split precision: every f32 operand as three bf16, six bf16 MFMAs (32x32x16) per 16 taps and 32x32 tile.  The
            # pipe it runs on is the bf16 one, so the fraction is EXECUTED bf16 flop/s over the dense bf16 peak; the
            # algorithmic (2 T flop per sample) rate and its ratio to the f32 MFMA peak are given beside it.
            T = cfg["taps"]
            koff = (1 - T) % 16                                  # whole 128-frame blocks: n0 is a multiple of 16
            n_iter = (koff + T + 30) // 16 + 1
            executed = ((N + 31) // 32) * ((B + 127) // 128) * n_iter * 24 * 32768.0
            ex_rate = executed / (kern_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "dtype": "bf16x3 (f32 operands split exactly into three bf16; f32 accumulation)",
                    "achieved": ex_rate, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ex_rate / MFMA_BF16_PEAK_TFLOPS,
                    "algorithmic_tflops": achieved, "algorithmic_over_f32_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS,
                    "note": "power-limited: shader clock 1.79-1.82 GHz under this kernel, matrix pipe 70-71 % busy (profiles/r03_fir_split_pmc.json)"}
        if kern_name in ("fir_half_kernel", "fir_halfp_kernel"):
            # two-part f16: every f32 operand as f16 hi + f16 lo, THREE f16 MFMAs (32x32x16) per 16 taps and 32x32 tile.  Half the
            # matrix-pipe time of the bf16 x 3 sweep: the sweep now sits between its two roofs, so both fractions are given and
            # `bound` names the nearer one -- HBM: the sweep's own algorithmic bytes (the window's (T - 1 + B) rows re-read + the
            # block written, per channel and block; the append is another kernel) over the 8 TB/s peak; matrix pipe: EXECUTED
            # f16 flop/s over the dense f16 peak.  The algorithmic 2 T flop per sample rate of SURVEY 8d is beside them.
            T = cfg["taps"]
            koff = (1 - T) % 16
            n_iter = (koff + T + 30) // 16 + 1
            executed = ((N + 31) // 32) * ((B + 127) // 128) * n_iter * 12 * 32768.0
            ex_rate = executed / (kern_ms * 1e-3) / 1e12
            sweep_bytes = ((T - 1 + B) * 4.0 / B + 4.0) * N * B
            hbm_rate = sweep_bytes / (kern_ms * 1e-3) / 1e9
            f_hbm, f_mfma = hbm_rate / HBM_PEAK_GBPS, ex_rate / MFMA_BF16_PEAK_TFLOPS
            roof = ({"bound": "hbm", "achieved": hbm_rate, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": f_hbm} if f_hbm >= f_mfma else
                    {"bound": "mfma", "achieved": ex_rate, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": f_mfma})
            roof.update({"dtype": "f16x2 (f32 operands as f16 hi + f16 lo, three products per term; f32 accumulation; bf16 x 3 second pass "
                                  "over the tiles outside f16's range)",
                         "frac_hbm": f_hbm, "hbm_gbps": hbm_rate, "sweep_bytes_per_launch": sweep_bytes,
                         "frac_f16_mfma": f_mfma, "executed_tflops": ex_rate,
                         "algorithmic_tflops": achieved, "algorithmic_over_f32_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS,
                         "note": "fir_halfp_kernel = the sweep over the packed {f16 hi, f16 lo} history the append pass writes (round 5); "
                                 "counters: profiles/r05_fir_halfp_pmc.json"})
    else:
        achieved = bps * N * B / (kern_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS}
    roof["frac_by_step"] = roof["frac"] * kern_ms / (dt * 1e3 / steps) if dt > 0 else None     # the same work over the whole step (launch gaps and the step's other kernels included)
    roof.update({"kernel": kern_name, "kernel_ms_avg": kern_ms, "launches": kern_launches, "timing": kern_method,
                 "algorithmic_bytes_per_sample": bps, "traffic": None})
    # HBM bytes per launch from the committed PMC pass of this same command, if one exists (not measured in this run)
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr):
        try:
            t = json.load(open(tr))
            ent = t.get(f"{cfg_name}{'split' if kern_name == 'fir_split_kernel' else 'half' if kern_name in ('fir_half_kernel', 'fir_halfp_kernel') else ''}:{N}:{B}")
            if ent:
                roof["traffic"] = ent["hbm_bytes_per_launch"]
                roof["traffic_source"] = ent.get("source")
        except Exception:
            pass
    # ---- cold: the first launches after an idle gap, which the settle phase above deliberately keeps out of the timed region
    # (the socket's power management answers an idle of >= 10 ms with ~50 slower launches: profiles/r02_idle_transient.txt)
    cold = None
    paced = None
    alt_bus = None
    if extras and not dist_run:
        idle_s = float(os.environ.get("DSPFX_BENCH_COLD_IDLE", "0.15"))
        ec0, ec1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        time.sleep(idle_s)
        ec0.record()
        for k in range(20):
            step(k)
        ec1.record()
        drain()
        fence()
        cold = {"idle_ms": idle_s * 1e3, "steps": 20, "ms_per_step": ec0.elapsed_time(ec1) / 20,
                "note": "first 20 launches after the idle gap, one event pair; the settled figure is ms_per_step"}
        if use_mix and mix_mode == "inline" and not is_fir:
            # the round-2 form on the same engine and buffers: in-kernel pipeline, bus two calls late
            ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for k in range(60):
                eng.process_mixpipe(xs[k % n_in], y, mixes[k & 3], B, n_connected=total_channels, stream=stream)
            ea0.record()
            for k in range(max(steps, 20)):
                eng.process_mixpipe(xs[k % n_in], y, mixes[k & 3], B, n_connected=total_channels, stream=stream)
            ea1.record()
            eng.mixpipe_flush(mixes[2], mixes[3], n_connected=total_channels, stream=stream)
            fence()
            alt_ms = ea0.elapsed_time(ea1) / max(steps, 20)
            alt_bus = {"what": "dspfx_process_mixpipe: the bus of block k-2 rides in block k's launch", "ms_per_step": alt_ms,
                       "frac": bps * N * B / (alt_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        if use_mix and args.paced_seconds > 0:
            paced = paced_run(torch, eng, xs, y, mixes, B, total_channels, stream, args.paced_seconds)
    # ---- several ranks: what one exchange of the bus costs (the kernel of dspfx_mix_allreduce alone, ranks in step), and the
    # paced leg with the GLOBAL bus of every block
    exchange = None
    if dist_run and use_mix and ctx.comm is not None:
        t = torch.zeros(B, dtype=torch.float32, device=dev)
        fence()
        xe = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for a_, b_ in xe:
            a_.record()
            eng.mix_allreduce(ctx.comm, t, B, total_channels, stream)
            b_.record()
        fence()
        us = sorted(1e3 * a_.elapsed_time(b_) for a_, b_ in xe[20:])
        exchange = {"what": "one dspfx_mix_allreduce of %d floats + Output hop, events around the call, ranks in step" % B, "backend": ctx.comm.backend,
                    "us_p50": us[len(us) // 2], "us_p99": us[int(len(us) * 0.99)], "us_max": us[-1]}
        if extras and args.paced_seconds > 0 and pbus is not None and pbus.inline:
            t0 = torch.tensor([time.time() + 0.2], dtype=torch.float64, device=dev)
            dist.broadcast(t0, 0)                     # all ranks start their block clock at the same wall-clock instant
            paced = paced_run(torch, eng, xs, y, mixes, B, total_channels, stream, args.paced_seconds, pbus=pbus,
                              start_at=float(t0.item()))
            pbus.drain()
            fence()
    # ---- several ranks: BOTH forms of the bus exchange in the same run.  `value` above is the same-block form -- the
    # Output node's semantics (nodes/output.rs:215-249 over node.rs:162-194: the GLOBAL bus of block k with block k's samples), its
    # exchange queued on the compute stream behind every chain kernel.  The overlapped form lets the bus ride two calls late
    # (dspfx_process_mixpipe) and exchanges it on the second stream, under the next blocks' chain kernels: the weak-scaling ceiling
    # without the exchange on the critical path.  Same engine, same buffers, same K, same barrier + MAX-over-ranks clock.
    forms = None
    if dist_run and use_mix and pbus is not None and pbus.inline and os.environ.get("DSPFX_BENCH_FORMS", "1") == "1":
        forms = {"inline": {"what": "same-block bus, exchange on the compute stream behind each chain kernel (= value)",
                            "value": value, "ms_per_step": dt * 1e3 / steps, "bus_delay_blocks": 0}}
        try:
            ovl_batch = int(os.environ.get("DSPFX_BENCH_OVERLAP_BATCH", "1"))
            pb2 = P.PipelinedMixBus(eng, total_channels, B, ctx.compute_stream, mix_stream, bus_world, batch=ovl_batch, device=dev,
                                    comm=ctx.comm, same_block=False)
            for k in range(max(warmup, 3 * ovl_batch + 8)):
                pb2.step(xs[k % n_in], y)
            pb2.drain()
            fence()
            eo0, eo1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dt2, _, _ = timed_region(lambda k: pb2.step(xs[k % n_in], y), pb2.drain, eo0, eo1)
            if ctx.use_dist:
                t = torch.tensor([dt2], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt2 = float(t.item())
            forms["overlapped"] = {"what": "bus two calls late (dspfx_process_mixpipe), exchanged on the second stream, %d block(s) per exchange" % ovl_batch,
                                   "value": samples / dt2, "ms_per_step": dt2 * 1e3 / steps, "gpu_event_ms_per_step": eo0.elapsed_time(eo1) / steps,
                                   "bus_delay_blocks": 2 + ovl_batch - 1}
            del pb2
        except Exception as ex:          # reporting only: never lose the headline line
            forms["overlapped"] = {"error": str(ex)[:300]}
            fence()
        # the third form: the SAME bus as `inline` (block k's, one exchange per block), but the exchange queued on the second
        # stream behind one event -- block k + 1's chain kernel does not wait for it, the bus completes a few microseconds after
        # its block's samples.  Same semantics as `value`, the exchange off the critical path.
        try:
            pb3 = P.PipelinedMixBus(eng, total_channels, B, ctx.compute_stream, mix_stream, bus_world, batch=1, device=dev,
                                    comm=ctx.comm, same_block=True, exchange_on_compute=False)
            for k in range(max(warmup, 8)):
                pb3.step(xs[k % n_in], y)
            pb3.drain()
            fence()
            ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dt3, _, _ = timed_region(lambda k: pb3.step(xs[k % n_in], y), pb3.drain, ea0, ea1)
            if ctx.use_dist:
                t = torch.tensor([dt3], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt3 = float(t.item())
            forms["same_block_second_stream"] = {"what": "same-block bus (as inline), exchange on the second stream behind one event per block",
                                                 "value": samples / dt3, "ms_per_step": dt3 * 1e3 / steps,
                                                 "gpu_event_ms_per_step": ea0.elapsed_time(ea1) / steps, "bus_delay_blocks": 0}
            del pb3
        except Exception as ex:
            forms["same_block_second_stream"] = {"error": str(ex)[:300]}
            fence()
    res = {
        "value": value, "ms_per_step": dt * 1e3 / steps, "roofline": roof, "chain": chain, "cfg": cfg, "cold": cold, "paced": paced, "alt_bus": alt_bus,
        "config": {"workload": cfg["desc"], "channels_per_gpu": N, "frames_per_block": B,
                   "delay_len": cfg.get("delay"), "taps": cfg.get("taps"), "link_flags": args.link_flags, **({"zero_input": True} if os.environ.get("DSPFX_BENCH_ZERO_INPUT") == "1" else {}),
                   "mix_bus": (mix_mode if use_mix else False), "parallelism": f"channel-shard x{world}",
                   "collective": (None if not dist_run else ("dspfx_mix_allreduce, backend %s, %d block(s) per exchange%s" % (
                                      ctx.comm.backend, BATCH, "".join(" [not %s]" % n for n in getattr(ctx, "comm_notes", []))))
                                  if ctx.comm is not None
                                  else "torch.distributed all_reduce" + (" (fallback: %s)" % ctx.comm_fallback if getattr(ctx, "comm_fallback", None) else "")),
                   "bus_exchange": exchange, "scaling_forms": forms,
                   "placement_probe": probe_log, "placement_tuning": tune_log,
                   "settle": {"steps": settle_steps, "ms_per_step": round(settle_ms, 4)},
                   "layout": f"channel-tiled [N/{tile}][B][{tile}]" if tile else "frame-major [B][N]",
                   "plan": eng.describe().strip().split("\n")[1:]},
        "gpu_event_ms_per_step": region_ms / steps, "host_submit_ms_per_step": t_submitted * 1e3 / steps,
        "block_budget_ms": B / 48.0,
        # what the host-clock window holds beside the K launches: the first launch's way to the GPU, the event records and the wait for
        # the last one -- a fixed cost per REGION (value = samples / dt pays it; at K = 20 it is most of frac - frac_by_step for config 2)
        "region_fixed_cost_us": (dt * 1e3 - kern_ms * steps) * 1e3 if region_timing else (dt * 1e3 - region_ms) * 1e3,
        "region_host_us": {"submit": t_submitted * 1e6, "event_seen": t_event * 1e6, "synchronized": dt * 1e6, "gpu_events": region_ms * 1e3},
    }
    del pbus, bus, xs, y, mixes
    eng.close()
    torch.cuda.empty_cache()
    return res


def host_path(ctx, args, blocks=20):
    """The boundary a Rust host actually crosses: dspfx_process_host from page-locked host buffers (dspfx_host_alloc), H2D + chain + D2H
    per block -- what host/rust/src/gpu_bank.rs's GpuBank::process costs per 128-frame block (node.rs:135-146, 271-288).  PCIe-bound and
    NEVER `value`: (a) config 5's shard, (b) the largest power-of-two channel count whose p99 block time stays inside the 2.667 ms block
    period (= channels one GPU serves in real time from host memory).  `blocks` blocks each after 4 untimed ones; every block timed on
    its own with the host clock (the call returns when `out` is complete in host memory)."""
    import numpy as np
    pkg = ctx.pkg
    B = 128
    period = B / 48.0

    def leg(n_channels):
        cfg = dict(CONFIGS["cfg5"], channels=n_channels)
        eng = pkg.Engine(n_channels, B, link_flags=args.link_flags, device=ctx.local_rank)
        eng.set_chain(build_chain(pkg, cfg))
        eng.kernels_ready(120000)
        px, py = pkg.PinnedArray((B, n_channels)), pkg.PinnedArray((B, n_channels))
        try:
            rng = np.random.default_rng(SEED)
            px.array[:] = rng.uniform(-1.0, 1.0, (B, n_channels)).astype(np.float32)
            for _ in range(4):
                eng.process_host(px.array, out=py.array)
            ms = np.empty(blocks)
            for k in range(blocks):
                t0 = time.perf_counter()
                eng.process_host(px.array, out=py.array)
                ms[k] = (time.perf_counter() - t0) * 1e3
            nbytes = float(px.array.nbytes)
            p50 = float(np.percentile(ms, 50))
            return {"channels": n_channels, "blocks": blocks, "ms_per_block_p50": p50, "ms_per_block_p99": float(np.percentile(ms, 99)),
                    "ms_per_block_max": float(ms.max()), "gbps_each_direction": nbytes / (p50 * 1e-3) / 1e9,
                    "samples_per_s": n_channels * B / (p50 * 1e-3), "realtime_channels": n_channels * B / (p50 * 1e-3) / 48000.0,
                    "block_budget_ms": period, "inside_budget_p99": bool(np.percentile(ms, 99) < period)}
        finally:
            px.close()
            py.close()
            eng.close()

    out = {"what": "dspfx_process_host from dspfx_host_alloc buffers: H2D + 5-node chain + D2H per 128-frame block (GpuBank::process); PCIe-bound, never `value`",
           "cfg5_shard": leg(CONFIGS["cfg5"]["channels"]), "largest_realtime_pow2": None, "tried": []}
    n = 1 << 18
    while n >= 1 << 12:
        r = leg(n)
        out["tried"].append({"channels": n, "ms_per_block_p99": r["ms_per_block_p99"]})
        if r["inside_budget_p99"]:
            out["largest_realtime_pow2"] = r
            break
        n >>= 1
    return out


def launch_env(args, ctx):
    """One process per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment torch.distributed.run (the driver's
    launcher, or self_launch below) sets."""
    world = ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = ctx.rank = int(os.environ.get("RANK", "0"))
    ctx.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1: launch with torch.distributed.run, or with no launcher environment at all "
                         "(bench.py then starts its own ranks)")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    return world, rank


def visible_gpus():
    """GPUs this process could use, WITHOUT initialising one (torch.cuda.device_count() only counts on this image)."""
    import torch
    return int(torch.cuda.device_count())


def self_launch(args, argv):
    """`python3 bench.py --gpus N` (N > 1) with no launcher environment: start the N ranks as FRESH child processes -- the
    driver's own launcher command, `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <same arguments>` -- BEFORE this process makes any GPU call, let rank 0's one JSON line through
    on stdout, and exit with the launcher's status.  Fewer than N visible GPUs is an error with a one-line reason, never a
    silent N = 1 (DSPFX_BENCH_SHARE_GPU=1, the one-GPU test rig, needs one; --dry-run needs none)."""
    import socket
    n = args.gpus
    # under a profiler whose preloaded library has already initialised the GPU in THIS process, starting other programs from it is
    # what the GPU pool forbids (an exec from a GPU-initialised process): say so instead of trying
    if any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY", "ROCPROF_ATT_LIBRARY_PATH")) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        print("bench.py: --gpus %d under a profiler: this process may already hold the GPU and must not start the ranks itself; "
              "profile with the launcher in front (rocprofv3 ... -- python3 -m torch.distributed.run ... bench.py --gpus %d)" % (n, n), file=sys.stderr)
        raise SystemExit(3)
    if not args.dry_run:
        need = 1 if os.environ.get("DSPFX_BENCH_SHARE_GPU") == "1" else n
        have = visible_gpus()
        if have < need:
            print(f"bench.py: --gpus {n} needs {need} visible GPU(s), this host shows {have}: not running", file=sys.stderr)
            raise SystemExit(3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: the mailbox communicator's peer mappings and RCCL need it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print("bench.py: no launcher environment; starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr)
    sys.stderr.flush()
    rc = subprocess.call(cmd, env=env)
    raise SystemExit(rc if rc >= 0 else 128 - rc)


class _DryEngine:
    """--dry-run: stands in for the engine on a host without a GPU.  The 'bus' of block k on rank r is a known vector, delivered
    like the engine delivers it (process_bus: same call; process_mixpipe: two calls late, un-normalised)."""

    def __init__(self, torch, rank, B, divisor):
        self.torch, self.rank, self.B, self.div, self.k = torch, rank, B, divisor, 0

    def bus(self, k, rank=None):
        r = self.rank if rank is None else rank
        return self.torch.arange(self.B, dtype=self.torch.float32) * 0.5 + (k * 7 + r * 1000)

    def process_bus(self, x, out, mix, n_frames, n_connected=0, side=None, stream=0):
        mix.copy_(self.bus(self.k))
        self.k += 1

    def process_mixpipe(self, x, out, mix, n_frames, n_connected=0, side=None, stream=0):
        if self.k >= 2:
            mix.copy_(self.bus(self.k - 2))
        self.k += 1

    def mixpipe_flush(self, mix_older, mix_newer, n_connected=0, stream=0):
        if self.k >= 2:
            mix_older.copy_(self.bus(self.k - 2))
        mix_newer.copy_(self.bus(self.k - 1))

    def mix_finish(self, mix, n_frames, n_connected, stream=0):
        mix /= self.div


def dry_run(args):
    """The host-side control flow of a --gpus N run without touching a GPU: environment, rendezvous (gloo), the
    communicator's id broadcast and its all-ranks fallback, channel sharding, the batched bus of parallel.PipelinedMixBus over
    a real collective, the MAX-over-ranks timing and rank 0's line."""
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    ctx = Ctx()
    ctx.dry = True
    world, rank = launch_env(args, ctx)
    if os.environ.get("DSPFX_BENCH_DRY_FAIL_RANK") == str(rank):        # tests: a rank that dies must fail the whole launch
        raise SystemExit("dry run: rank %d told to fail" % rank)
    ctx.use_dist = world > 1
    ctx.dev = torch.device("cpu")
    if ctx.use_dist:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx.pkg = load_package()
    from dsp_stuff_amd import parallel as P
    ctx.comm = None
    make_comm(ctx)                        # no device: every rank must land in the torch.distributed fallback together
    cfg = CONFIGS[args.config]
    N, B = args.channels or cfg["channels"], args.frames or cfg["frames"]
    shard = P.weak_shard(N, world, rank)
    mix_mode = os.environ.get("DSPFX_BENCH_MIX", "inline")
    batch = int(os.environ.get("DSPFX_BENCH_MIX_BATCH", "1"))
    div = float(ctx.pkg.link_divisor(shard.total_channels))
    eng = _DryEngine(torch, rank, B, div)
    pbus = P.PipelinedMixBus(eng, shard.total_channels, B, None, None, world, batch=batch, device="cpu",
                             same_block=(mix_mode == "inline"), order=P.HostOrder())
    if ctx.use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.warmup + args.steps):
        pbus.step(None, None)
    pbus.drain()
    if ctx.use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    n = args.warmup + args.steps
    ok = True
    for j, row in pbus.results().items():
        want = sum(eng.bus(j, r) for r in range(world)) / div
        ok = ok and bool(torch.allclose(row, want, rtol=1e-6))
    # the overlapped form of measure()'s scaling_forms: bus two calls late, exchanged off the compute stream's order
    forms = None
    if ctx.use_dist and mix_mode == "inline":
        eng2 = _DryEngine(torch, rank, B, div)
        pb2 = P.PipelinedMixBus(eng2, shard.total_channels, B, None, None, world, batch=int(os.environ.get("DSPFX_BENCH_OVERLAP_BATCH", "1")),
                                device="cpu", same_block=False, order=P.HostOrder())
        t1 = time.perf_counter()
        for k in range(args.steps):
            pb2.step(None, None)
        pb2.drain()
        dist.barrier()
        dt2 = time.perf_counter() - t1
        for j, row in pb2.results().items():
            want = sum(eng2.bus(j, r) for r in range(world)) / div
            ok = ok and bool(torch.allclose(row, want, rtol=1e-6))
        eng3 = _DryEngine(torch, rank, B, div)
        pb3 = P.PipelinedMixBus(eng3, shard.total_channels, B, None, None, world, batch=1, device="cpu", same_block=True, order=P.HostOrder(),
                                exchange_on_compute=False)
        t1 = time.perf_counter()
        for k in range(args.steps):
            pb3.step(None, None)
        pb3.drain()
        dist.barrier()
        dt3 = time.perf_counter() - t1
        for j, row in pb3.results().items():
            want = sum(eng3.bus(j, r) for r in range(world)) / div
            ok = ok and bool(torch.allclose(row, want, rtol=1e-6))
        forms = {"inline": {"ms_per_step": dt * 1e3 / (args.warmup + args.steps), "bus_delay_blocks": 0},
                 "overlapped": {"ms_per_step": dt2 * 1e3 / args.steps, "bus_delay_blocks": 2 + pb2.batch - 1, "bus_checked_blocks": len(pb2.results())},
                 "same_block_second_stream": {"ms_per_step": dt3 * 1e3 / args.steps, "bus_delay_blocks": 0, "bus_checked_blocks": len(pb3.results())}}
    t = torch.tensor([dt, 0.0 if ok else 1.0], dtype=torch.float64)
    if ctx.use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({
            "dry_run": True, "metric": "mono-channel-samples/sec through 5-node chain @128-frame blocks", "value": None,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t[0]) * 1e3 / n,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["desc"], "channels_per_gpu": N, "total_channels": shard.total_channels, "channel_offset_of_last_rank": P.weak_shard(N, world, world - 1).offset,
                       "frames_per_block": B, "mix_bus": mix_mode, "parallelism": f"channel-shard x{world}",
                       "collective": ("torch.distributed all_reduce (fallback: %s)" % ctx.comm_fallback) if getattr(ctx, "comm_fallback", None) else
                                     ("dspfx_mix_allreduce" if ctx.comm is not None else None)},
            "scaling_forms": forms, "bus_checked_blocks": len(pbus.results()), "bus_ok": bool(t[1] == 0.0)}))
    if ctx.use_dist:
        dist.destroy_process_group()
    if t[1] != 0.0:
        raise SystemExit("dry run: a bus row differs from the sum over ranks")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args, sys.argv[1:])
    if args.dry_run:
        return dry_run(args)
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    ctx = Ctx()
    world, rank = launch_env(args, ctx)
    # DSPFX_BENCH_SHARE_GPU=1 (a test rig for one-GPU boxes, tools/r04_two_ranks_one_gpu.sh): every rank uses GPU 0 and the
    # process group runs over gloo -- RCCL refuses two ranks on one device.  Everything else is the N > 1 path as the driver
    # launches it: sharding, per-rank engines and tuning, the batched bus with its collective on the second stream, barriers,
    # MAX over ranks, one line from rank 0.  The numbers of such a run mean nothing (the ranks time-slice one chip).
    share_gpu = os.environ.get("DSPFX_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        ctx.local_rank = 0
        os.environ.setdefault("DSPFX_BENCH_COMM", "torch")
    if visible_gpus() <= ctx.local_rank:
        raise SystemExit(f"bench.py rank {rank}: local rank {ctx.local_rank} has no GPU (this host shows {visible_gpus()}): not running")
    torch.cuda.set_device(ctx.local_rank)
    dev = ctx.dev = torch.device("cuda", ctx.local_rank)
    # DSPFX_BENCH_FORCE_DIST=1 initialises RCCL even for one rank so the collective code path can be
    # exercised on a 1-GPU box (the driver launches the real N>1 runs with torch.distributed.run)
    ctx.use_dist = world > 1 or os.environ.get("DSPFX_BENCH_FORCE_DIST") == "1"
    if ctx.use_dist:
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ctx.pkg = load_package()
    from dsp_stuff_amd import parallel as P
    ctx.P = P
    # the chain kernels get their own high-priority stream: the mix bus' small kernels and the RCCL
    # all-reduce on the second stream must not delay the next block's launch
    if os.environ.get("DSPFX_BENCH_HIPRIO", "1") == "1":
        ctx.compute_stream = torch.cuda.Stream(device=dev, priority=-1)
        torch.cuda.set_stream(ctx.compute_stream)
    else:
        ctx.compute_stream = torch.cuda.current_stream()
    ctx.mix_stream = torch.cuda.Stream(device=dev)
    ctx.mark = torch.zeros(64, dtype=torch.float32, device=dev)
    # everything that is lazy on first use (torch code objects of the marker kernels, RCCL's communicator) is used
    # once NOW: a first call between warm-up and the timed region stalls the queue for ~20 ms, and an idle gap of
    # >= 10 ms is followed by ~40 slow launches (profiles/r02_idle_transient.txt)
    ctx.mark.sin_()
    ctx.mark.cos_()
    ctx.comm = None
    if ctx.use_dist:
        dist.barrier()
        dist.all_reduce(ctx.mark)
        # The mix bus' own communicator lives behind the C ABI (dspfx_comm_create / dspfx_mix_allreduce: what a Rust or
        # C++ host calls); it is made in measure().  DSPFX_BENCH_COMM=torch keeps torch.distributed's all_reduce (A/B).
        if os.environ.get("DSPFX_BENCH_COMM_EARLY", "0") == "1":
            make_comm(ctx)
    torch.cuda.synchronize()

    over = {k: getattr(args, k) for k in ("channels", "frames", "delay", "taps") if getattr(args, k) is not None}
    if args.paced:
        args.no_others = True
        args.no_cpu_baseline = True
    # Config 2 is timed BEFORE the headline config.  After seconds of the large configs the chip runs this light, latency-
    # sensitive kernel 2-4 % slower for about three seconds (23.4-24.0 us against the 23.1 us it holds for as long as it runs
    # alone: profiles/r03_small_n.txt) -- the previous workload's power state, not config 2's.  DSPFX_BENCH_EARLY= (empty)
    # restores the old order (tools/archive/r03_cfg2_order.sh).
    want_others = args.config == "cfg5" and world == 1 and not over and not args.no_others
    other_names = [n for n in os.environ.get("DSPFX_BENCH_OTHERS", "cfg3,cfg2,cfg4").split(",") if n]
    early_names = [n for n in os.environ.get("DSPFX_BENCH_EARLY", "cfg2").split(",") if n and n in other_names] if want_others else []
    timed_order = []

    def other_entry(name):
        o = measure(ctx, args, name, args.steps, args.warmup)
        timed_order.append(name)
        return {"workload": o["config"]["workload"], "value": o["value"], "unit": "samples/s",
                "ms_per_step": o["ms_per_step"], "block_budget_ms": o["block_budget_ms"],
                "roofline": o["roofline"], "plan": o["config"]["plan"], "region_fixed_cost_us": o.get("region_fixed_cost_us"),
                "settle": o["config"]["settle"], "placement_tuning": o["config"]["placement_tuning"]}

    early = {}
    for name in early_names:
        try:
            early[name] = other_entry(name)
        except Exception as ex:
            early[name] = {"error": str(ex)[:300]}
    r = measure(ctx, args, args.config, args.steps, args.warmup, over, extras=(world == 1))
    timed_order.append(args.config)

    # The other single-GPU BASELINE configs, timed in the same run with the same K / W (extra key; the line's
    # `value` stays the headline config's).  Default invocation on one GPU only.
    others = None
    if want_others:
        others = {}
        for name in other_names:
            if name in early:
                others[name] = early[name]
                continue
            try:
                others[name] = other_entry(name)
            except Exception as ex:   # never lose the headline line
                others[name] = {"error": str(ex)[:300]}
        # config 4 once more through the f32 sweep (DSPFX_FIR_SPLIT=0: v_mfma_f32_32x32x2_f32, the reference's own data type on the
        # matrix pipe; the engine's default since round 3 is the split-precision sweep on the bf16 pipe, same stated tolerance)
        try:
            os.environ["DSPFX_FIR_SPLIT"] = "0"
            o = measure(ctx, args, "cfg4", args.steps, args.warmup)
            timed_order.append("cfg4_f32")
            others["cfg4_f32"] = {"workload": o["config"]["workload"] + " [DSPFX_FIR_SPLIT=0: f32 sweep]", "value": o["value"], "unit": "samples/s",
                                  "ms_per_step": o["ms_per_step"], "block_budget_ms": o["block_budget_ms"], "region_fixed_cost_us": o.get("region_fixed_cost_us"),
                                  "roofline": o["roofline"], "plan": o["config"]["plan"]}
        except Exception as ex:
            others["cfg4_f32"] = {"error": str(ex)[:300]}
        finally:
            os.environ.pop("DSPFX_FIR_SPLIT", None)
        # ... and through the bf16 x 3 sweep (round 3's default; now the second pass of the two-part f16 sweep)
        try:
            os.environ["DSPFX_FIR_HALF"] = "0"
            o = measure(ctx, args, "cfg4", args.steps, args.warmup)
            timed_order.append("cfg4_split")
            others["cfg4_split"] = {"workload": o["config"]["workload"] + " [DSPFX_FIR_HALF=0: bf16 x 3 sweep]", "value": o["value"], "unit": "samples/s",
                                    "ms_per_step": o["ms_per_step"], "block_budget_ms": o["block_budget_ms"], "region_fixed_cost_us": o.get("region_fixed_cost_us"),
                                    "roofline": o["roofline"], "plan": o["config"]["plan"]}
        except Exception as ex:
            others["cfg4_split"] = {"error": str(ex)[:300]}
        finally:
            os.environ.pop("DSPFX_FIR_HALF", None)

    if rank != 0:
        if ctx.use_dist:
            dist.destroy_process_group()
        return

    value = r["value"]
    line = {
        "metric": "mono-channel-samples/sec through 5-node chain @128-frame blocks" if args.config == "cfg5"
                  else f"mono-channel-samples/sec ({args.config})",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": r["config"],
        "roofline": r["roofline"],
        "realtime_channels": value / 48000.0,
        "block_latency_ms": r["ms_per_step"], "block_budget_ms": r["block_budget_ms"],
        "gpu_event_ms_per_step": r["gpu_event_ms_per_step"], "host_submit_ms_per_step": r["host_submit_ms_per_step"],
    }
    # several ranks: what exchanged the bus, at the top level of the line -- the backend that actually RAN,
    # what one exchange cost, and any fallback that was taken on the way there
    if ctx.use_dist:
        line["bus_exchange"] = r["config"].get("bus_exchange")
        line["scaling_forms"] = r["config"].get("scaling_forms")
        line["collective_backend"] = (ctx.comm.backend if ctx.comm is not None else "torch.distributed all_reduce")
        line["collective_fallback"] = (getattr(ctx, "comm_fallback", None) or "; ".join(getattr(ctx, "comm_notes", []) or []) or None)
    line["region_fixed_cost_us"] = r.get("region_fixed_cost_us")
    line["region_host_us"] = r.get("region_host_us")
    if r.get("cold") is not None:
        line["cold"] = r["cold"]
    if others is not None:
        line["timed_order"] = timed_order          # the order of the timed regions in this run (tools/trace_phases.py)
    if r.get("paced") is not None:
        line["paced"] = r["paced"]
    if r.get("alt_bus") is not None:
        line["bus_two_calls_late"] = r["alt_bus"]
    if others is not None:
        line["other_configs"] = others
    if world == 1 and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(r["chain"], r["cfg"], args.link_flags, args.cpu_seconds)
            line["cpu_baseline"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        except Exception as ex:   # the baseline is reporting only; never lose the GPU line
            line["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
                                    "sample": f"failed: {ex}"}
    else:
        line["cpu_baseline"] = None      # N > 1, or switched off with --no-cpu-baseline
    if world == 1 and want_others and os.environ.get("DSPFX_BENCH_HOST_PATH", "1") == "1":
        try:
            line["host_path"] = host_path(ctx, args)
        except Exception as ex:          # reporting only
            line["host_path"] = {"error": str(ex)[:300]}
    # The driver keeps the last 8 KB of stdout: the line that goes there is the COMPACT one (every contract field, every
    # config's numbers); the full record -- plans, notes, the CPU baseline's thread legs -- goes to stderr and to
    # gpurun_out/bench_detail_<N>gpu.json.
    detail = json.dumps(line)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_detail_%dgpu.json" % world), "w") as f:
            f.write(detail + "\n")
    except OSError:
        pass
    print("bench.py detail: " + detail, file=sys.stderr)
    print(json.dumps(compact_line(line)))
    if ctx.use_dist:
        dist.destroy_process_group()


def _round(v, digits=6):
    """Floats to `digits` significant digits (the line is read by people and by an 8 KB tail)."""
    if isinstance(v, float):
        return float(("%." + str(digits) + "g") % v) if v == v and abs(v) != float("inf") else v
    if isinstance(v, dict):
        return {k: _round(x, digits) for k, x in v.items()}
    if isinstance(v, list):
        return [_round(x, digits) for x in v]
    return v


def compact_line(line):
    """The one JSON line of the contract, small enough (< 7 KB) for the driver's 8 KB stdout tail to hold ALL of it: every contract
    field as it is; the long texts (plans, notes, provenance sentences, the CPU baseline's thread legs) stay in the detail record."""
    def roof(r):
        if not r:
            return r
        keep = ("bound", "achieved", "peak", "unit", "frac", "frac_by_step", "traffic", "kernel", "kernel_ms_avg", "launches",
                "algorithmic_bytes_per_sample", "frac_hbm", "frac_f16_mfma", "algorithmic_tflops")
        o = {k: r[k] for k in keep if k in r}
        if r.get("traffic_source"):
            o["traffic_source"] = r["traffic_source"].split(" ")[0]          # the file under profiles/
        return o

    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data") if k in line}
    c = dict(line.get("config") or {})
    plan = [l for l in (c.pop("plan", None) or []) if l.startswith(("stage", "node"))]
    c.pop("placement_probe", None)
    if "scaling_forms" in line:                      # several ranks: both are top-level keys of the line
        c.pop("scaling_forms", None)
        c.pop("bus_exchange", None)
    c["plan"] = [l[:160] for l in plan]
    out["config"] = c
    out["roofline"] = roof(line.get("roofline"))
    cb = line.get("cpu_baseline")
    if cb:
        cb = {k: cb[k] for k in ("value", "unit", "cores", "kind", "build", "cpu_model", "cgroup_cpu_quota", "gpu_over_cpu") if k in cb}
        full = line["cpu_baseline"]
        cb["sample"] = (full.get("sample") or "")[:150]
        if full.get("single_thread"):
            cb["single_thread_value"] = full["single_thread"]["value"]
    out["cpu_baseline"] = cb
    for k in ("bus_exchange", "scaling_forms", "collective_backend", "collective_fallback", "region_fixed_cost_us", "realtime_channels", "block_latency_ms",
              "block_budget_ms", "gpu_event_ms_per_step", "host_submit_ms_per_step", "timed_order"):
        if k in line:
            out[k] = line[k]
    if line.get("cold"):
        out["cold"] = {k: line["cold"][k] for k in ("idle_ms", "steps", "ms_per_step")}
    if line.get("paced"):
        out["paced"] = {k: v for k, v in line["paced"].items() if k != "what"}
    if line.get("host_path"):
        h = line["host_path"]
        out["host_path"] = h if "error" in h else {
            k: ({kk: v[kk] for kk in ("channels", "ms_per_block_p50", "ms_per_block_p99", "gbps_each_direction", "realtime_channels", "inside_budget_p99")}
                if isinstance(v, dict) else v) for k, v in h.items() if k in ("cfg5_shard", "largest_realtime_pow2")}
    if line.get("bus_two_calls_late"):
        out["bus_two_calls_late"] = {k: v for k, v in line["bus_two_calls_late"].items() if k != "what"}
    if line.get("other_configs"):
        oc = {}
        for name, o in line["other_configs"].items():
            if "error" in o:
                oc[name] = o
                continue
            oc[name] = {"value": o["value"], "ms_per_step": o["ms_per_step"], "roofline": roof(o.get("roofline")),
                        "region_fixed_cost_us": o.get("region_fixed_cost_us")}
        out["other_configs"] = oc
    out["detail"] = "stderr ('bench.py detail: ...') and gpurun_out/bench_detail_%dgpu.json" % line.get("n_gpus", 1)
    return {k: (v if isinstance(v, (int, float)) or k == "roofline" else _round(v, 7)) for k, v in out.items()}      # the contract's own numbers stay as measured


if __name__ == "__main__":
    main()
