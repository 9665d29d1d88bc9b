"""The mix bus ACROSS DEVICES (north_star: "RCCL over xGMI only for the final mix-bus all-reduce"; nodes/output.rs:215-249 feeding
node.rs:162-194 as the global bus).  Every test here needs at least two GPUs and skips itself on a one-GPU box -- the boxes this
repository is built on have one, so these are the tests that run THEMSELVES the first time the suite meets a multi-GPU node
(VERDICT r04 #2).  The same rank script runs in tests/test_gpu_threads.py with every rank on device 0, so it cannot rot.

What they pin down, per backend (mailbox: hipIpc peer writes; rccl: ncclAllReduce behind the same ABI):
  * one fresh process per rank, one device per rank, every block's bus through dspfx_process_bus -> dspfx_mix_allreduce;
  * every rank ends with the same bits; for the mailbox they are the rank-ordered f32 sum ((0 + b0) + b1) + ... of the
    rank-local buses, bit for bit; for RCCL the sum in whatever order its topology search picked (exact on integer data);
  * the exchange's latency over xGMI is printed (events around the call, ranks drifting).
And two risks nobody has seen run: threads of ONE process on different devices (peer access instead of
hipIpc), and a mailbox in ordinary coarse-grained memory (DSPFX_COMM_COARSE=1), whose polling loads the owner's L2 may serve
stale -- the reason the library refuses to fall back to it on its own."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32
B = 128


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


NDEV = _device_count()
# DSPFX_TEST_MULTIDEVICE_ON_ONE=1: run the test BODIES on a one-GPU box, every rank on device 0 (mailbox only: RCCL refuses two
# ranks on one device) -- how these functions were exercised where they were written; proves nothing about xGMI.
FORCED = os.environ.get("DSPFX_TEST_MULTIDEVICE_ON_ONE") == "1" and NDEV == 1
needs_two = pytest.mark.skipif(NDEV < 2 and not FORCED, reason="needs at least two GPUs (this box has %d)" % NDEV)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def _run_ranks(dspfx, tmp_path, world, backend, exact, N, blocks, extra_env=None, timeout=600):
    from test_gpu_threads import _RANK_SCRIPT
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    uid = dspfx.comm_unique_id(backend)
    env = dict(os.environ, **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, "-c", _RANK_SCRIPT.replace("ROOT", repr(root)), str(r), str(world), uid.hex(), str(tmp_path), str(N),
                               str(blocks), "1" if exact else "0", str(r % NDEV), backend], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    outs, errs = [], []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        errs.append((p.returncode, e[-3000:]))
        outs.append(o)
    return outs, errs


def _local_buses(dspfx, torch, world, N, blocks, exact):
    """The rank-local, un-normalised buses of the same shards, one engine after the other on device 0."""
    from chains import chain5
    n_loc = N // world
    local = []
    for r in range(world):
        eng = dspfx.Engine(n_loc, B, link_flags=0 if exact else 3, channel_offset=r * n_loc, tile_channels=256)
        eng.set_chain([dspfx.Gain(1.0)] if exact else chain5(dspfx, 256))
        x = torch.empty(B * n_loc, device="cuda")
        y = torch.empty_like(x)
        b = torch.zeros((blocks, B), device="cuda")
        for k in range(blocks):
            eng.fill_noise(x, B, k * B, 0x5EED0001)
            if exact:
                x.mul_(8.0).round_()
            eng.process_bus(x, y, b[k], B, n_connected=0)
        torch.cuda.synchronize()
        local.append(b.cpu().numpy())
        eng.close()
    return local


@needs_two
@pytest.mark.parametrize("backend", ["mailbox", "rccl"])
@pytest.mark.parametrize("world", sorted({2, min(8, max(2, NDEV))}))
@pytest.mark.parametrize("exact", [False, True])
def test_every_blocks_bus_across_devices(dspfx, torch_cuda, tmp_path, backend, world, exact):
    torch = torch_cuda
    if FORCED and backend == "rccl":
        pytest.skip("RCCL refuses two ranks on one device")
    N, blocks = world * 256 * 12, 48
    outs, errs = _run_ranks(dspfx, tmp_path, world, backend, exact, N, blocks)
    for rc, e in errs:
        assert rc == 0, e
    info = [json.loads(o.strip().splitlines()[-1]) for o in outs]
    assert sorted(i["device"] for i in info) == sorted(r % NDEV for r in range(world))
    buses = [np.load(tmp_path / ("bus%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(buses[0].view(np.uint32), buses[r].view(np.uint32)), (backend, r)     # every rank: the same bits
    local = _local_buses(dspfx, torch, world, N, blocks, exact)
    acc = np.zeros_like(local[0])
    for b in local:
        acc = (acc + b).astype(F)
    want = (acc / dspfx.link_divisor(N)).astype(F)
    if backend == "mailbox" or exact:
        assert np.array_equal(buses[0].view(np.uint32), want.view(np.uint32)), backend           # rank order, bit for bit
    else:
        assert np.allclose(buses[0], want, rtol=1e-5, atol=1e-6)
    assert np.abs(want).max() > 0
    print("\n%s exchange over %d devices (%d ranks): p50 %s us, fastest %s us, slowest %s us" % (
        backend, min(world, NDEV), world, [round(i["us_p50"], 1) for i in info], [round(i["us_min"], 1) for i in info], [round(i["us_max"], 1) for i in info]))


@needs_two
def test_coarse_grained_mailbox_across_devices_is_why_there_is_no_fallback(dspfx, torch_cuda, tmp_path):
    """DSPFX_COMM_COARSE=1 puts the mailbox into ordinary device memory.  Peer writes land in HBM, but the owner's polling
    loads may be served from its L2 for ever: the exchange then times out into NaNs.  Either outcome is recorded -- correct
    buses disprove the risk on this machine, stale ones are the reason dspfx_comm_create fails instead of falling back."""
    torch = torch_cuda
    world, N, blocks = 2, 2 * 256 * 12, 24
    outs, errs = _run_ranks(dspfx, tmp_path, world, "mailbox", True, N, blocks, extra_env={"DSPFX_COMM_COARSE": "1", "DSPFX_COMM_SPIN": str(1 << 18)}, timeout=300)
    ok = all(rc == 0 for rc, _ in errs)
    if ok:
        buses = [np.load(tmp_path / ("bus%d.npy" % r)) for r in range(world)]
        local = _local_buses(dspfx, torch, world, N, blocks, True)
        want = ((local[0] + local[1]).astype(F) / dspfx.link_divisor(N)).astype(F)
        ok = all(np.array_equal(b.view(np.uint32), want.view(np.uint32)) for b in buses)
    print("\ncoarse-grained mailbox across two devices: %s" % ("correct on this machine" if ok else "STALE / timed out (as feared)"))
    if not ok:
        pytest.xfail("a coarse-grained mailbox is served stale across devices: no silent fallback (comm.hip, mailbox_join)")


@needs_two
def test_thread_ranks_of_one_process_on_two_devices(dspfx, torch_cuda):
    """Ranks that are THREADS of one process share an address space: a peer's mailbox is its raw pointer, which another
    device may only write after hipDeviceEnablePeerAccess (mailbox_join enables it, or refuses).  Two threads, two devices,
    exact integer buses against the closed form."""
    torch = torch_cuda
    world, blocks = 2, 32
    uid = dspfx.comm_unique_id("mailbox")
    res, errs = {}, []

    def rank_thread(r):
        try:
            d = r % NDEV
            torch.cuda.set_device(d)
            eng = dspfx.Engine(256, B, link_flags=0, device=d)
            eng.set_chain([])
            comm = dspfx.Comm(d, world, r, uid)
            s = torch.cuda.Stream(device=d)
            out = torch.zeros((blocks, B), device="cuda:%d" % d)
            with torch.cuda.stream(s):
                for k in range(blocks):
                    out[k] = torch.arange(B, device="cuda:%d" % d, dtype=torch.float32) * (r + 1) + k
                    eng.mix_allreduce(comm, out[k], B, 0, s.cuda_stream)
            s.synchronize()
            res[r] = out.cpu().numpy()
            comm.close()
            eng.close()
        except Exception as ex:      # noqa: BLE001
            errs.append((r, repr(ex)))

    ts = [threading.Thread(target=rank_thread, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errs, errs
    want = np.stack([np.arange(B, dtype=F) * 3 + 2 * k for k in range(blocks)])
    for r in range(world):
        assert np.array_equal(res[r], want), r


@pytest.mark.skipif(NDEV < 2, reason="needs at least two GPUs (this box has %d)" % NDEV)
@pytest.mark.parametrize("world", sorted({2, min(8, max(NDEV, 2))}))
def test_the_bare_bench_command_scales_itself_across_devices(world):
    """`python3 bench.py --gpus N` as the driver may issue it -- no launcher, one rank per DEVICE, RCCL for torch's group, the C ABI's
    communicator for the bus: one line, the whole-job value = N x the shard, the backend that ran named at the top level, all three
    forms of the exchange timed.  (A reduced shard: this test is about the launch and the exchange, the curve is the driver's.)  The
    one-GPU twin of this test runs every round with both ranks on device 0 (tests/test_gpu_fullsize.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID", "DSPFX_BENCH_SHARE_GPU")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "5", "--channels", "262144",
                        "--delay", "1024"], capture_output=True, text=True, timeout=1800, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["parallelism"] == "channel-shard x%d" % world
    assert abs(d["value"] - world * 262144 * 128 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["collective_backend"] in ("mailbox", "rccl", "torch.distributed all_reduce")
    f = d["scaling_forms"]
    assert all(f[k].get("value", 0) > 0 for k in ("inline", "overlapped", "same_block_second_stream")), f
    print("bare --gpus %d: backend %s (fallback: %s), exchange p50 %s us; samples/s inline %.4g, second stream %.4g, two calls late %.4g" % (
        world, d["collective_backend"], d["collective_fallback"], (d.get("bus_exchange") or {}).get("us_p50"),
        f["inline"]["value"], f["same_block_second_stream"]["value"], f["overlapped"]["value"]))
