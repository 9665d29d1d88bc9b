"""The N>1 path on CPU: world-size-2 gloo runs of the host logic in dsp-stuff_amd/parallel.py
(channel sharding + mix-bus all-reduce + Output-hop divisor), checked against the single-process
oracle.  The per-rank partial sums come from the oracle here (there is no GPU); on the GPU box
the same MixBus object is fed by the engine's fused epilogue (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, blocks, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from __graft_entry__ import load_package
    import chains
    pkg = load_package()
    from dsp_stuff_amd import parallel as P   # noqa: E402  (submodule of the loaded package)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sh = P.shard_channels(total, world, rank)
    chain = chains.chain5(pkg, delay=128)
    descs = [n.oracle_desc() for n in chain]
    # per-rank partial mix (what the fused epilogue + mix_reduce produce on the GPU)
    _, part = O.run_noise_channels(descs, 0x5EED0001, sh.offset, sh.channels, 0, blocks, link_flags=3,
                                   want_out=False, want_mix=True)
    B = 128
    div = O.link_divisor(total)
    finished = []

    def finish(mix, n_frames, n_total):     # Output hop (node.rs:189-191) on a CPU tensor
        assert n_frames == B and n_total == total
        mix /= float(div)
        finished.append(mix.clone())

    bus = P.MixBus(total, B, finish)
    for b in range(blocks):
        m = torch.from_numpy(part[b * B:(b + 1) * B].astype(np.float32))
        bus.submit(m)
        assert len(finished) == b            # pipelined: block b completes on the next submit
    bus.drain()
    assert len(finished) == blocks
    if rank == 0:
        out_q.put(np.concatenate([f.numpy() for f in finished]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [256, 1001])
def test_mix_bus_world2_matches_single_process_oracle(total):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from __graft_entry__ import load_package
    import chains
    pkg = load_package()
    blocks, world = 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, blocks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    chain = chains.chain5(pkg, delay=128)
    _, full = O.run_noise_channels([n.oracle_desc() for n in chain], 0x5EED0001, 0, total, 0, blocks,
                                   link_flags=3, want_out=False, want_mix=True)
    expect = (full.astype(np.float32) / O.link_divisor(total)).astype(np.float32)
    assert np.allclose(got, expect, rtol=1e-5, atol=1e-7)


def test_shard_channels_properties():
    from __graft_entry__ import load_package
    load_package()
    from dsp_stuff_amd import parallel as P
    for total in (0, 1, 7, 8, 1000, 8388608):
        for world in (1, 2, 3, 8):
            shards = [P.shard_channels(total, world, r) for r in range(world)]
            assert sum(s.channels for s in shards) == total
            assert shards[0].offset == 0
            for a, b in zip(shards, shards[1:]):
                assert b.offset == a.offset + a.channels          # contiguous, ordered, no overlap
            assert max(s.channels for s in shards) - min(s.channels for s in shards) <= 1
    s = P.weak_shard(1 << 20, 8, 3)
    assert (s.offset, s.channels, s.total_channels) == (3 << 20, 1 << 20, 8 << 20)
    with pytest.raises(ValueError):
        P.shard_channels(10, 2, 2)


def test_mix_bus_world1_finishes_immediately():
    from __graft_entry__ import load_package
    load_package()
    from dsp_stuff_amd import parallel as P
    calls = []
    bus = P.MixBus(64, 128, lambda m, nf, n: calls.append((nf, n)), world=1)
    bus.submit(torch.zeros(128))
    assert calls == [(128, 64)]
    bus.drain()
    assert calls == [(128, 64)]


# ---- PipelinedMixBus (what bench.py --gpus N runs) with two ranks holding DIFFERENT data --------------------------------

class _FakeEngine:
    """Stands in for dsp_stuff_amd.Engine in a host-only run: the 'chain' is the oracle's partial bus of this rank's
    channel shard (precomputed per block), delivered the way the engine delivers it -- in the same call (process_bus)
    or two calls late (process_mixpipe + mixpipe_flush), un-normalised; mix_finish is the Output hop."""

    def __init__(self, partial, B, divisor):
        self.partial, self.B, self.div = partial, B, divisor
        self.k = 0
        self.calls = []

    def _bus(self, k):
        return torch.from_numpy(self.partial[k * self.B:(k + 1) * self.B].astype(np.float32))

    def process_bus(self, x, out, mix, n_frames, n_connected=0, side=None, stream=0):
        assert n_frames == self.B and n_connected == 0 and mix is not None
        mix.copy_(self._bus(self.k))
        self.k += 1

    def process_mixpipe(self, x, out, mix, n_frames, n_connected=0, side=None, stream=0):
        assert n_frames == self.B and n_connected == 0
        if self.k >= 2:
            assert mix is not None
            mix.copy_(self._bus(self.k - 2))
        else:
            assert mix is None
        self.k += 1

    def mixpipe_flush(self, mix_older, mix_newer, n_connected=0, stream=0):
        if self.k >= 2:
            mix_older.copy_(self._bus(self.k - 2))
        mix_newer.copy_(self._bus(self.k - 1))

    def mix_finish(self, mix, n_frames, n_connected, stream=0):
        self.calls.append(n_frames)
        mix /= float(self.div)


def _pipelined_worker(rank, world, port, total, blocks, batch, same_block, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from __graft_entry__ import load_package
    import chains
    pkg = load_package()
    from dsp_stuff_amd import parallel as P   # noqa: E402
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sh = P.shard_channels(total, world, rank)
    descs = [n.oracle_desc() for n in chains.chain5(pkg, delay=128)]
    B = 128
    _, part = O.run_noise_channels(descs, 0x5EED0001, sh.offset, sh.channels, 0, blocks, link_flags=3,
                                   want_out=False, want_mix=True)
    eng = _FakeEngine(part, B, O.link_divisor(total))
    second = same_block == "second_stream"         # the same-block bus, exchanged off the compute stream's order (round 6)
    same_block = bool(same_block)
    pb = P.PipelinedMixBus(eng, total, B, None, None, world, batch=batch, device="cpu", same_block=same_block,
                           order=P.HostOrder(), exchange_on_compute=(False if second else None))
    assert pb.inline == (same_block and batch == 1 and not second)
    lag = 0 if same_block else 2
    got = {}
    for k in range(blocks):
        pb.step(None, None)
        # a finished batch is complete one submit later (MixBus pipelines the collective); read it before its ring is reused
        j_done = k - lag + 1 - batch            # blocks below this index belong to batches whose successor was submitted
        if j_done > 0 and j_done % batch == 0:
            for j in range(j_done - batch, j_done):
                got[j] = pb._row(j).clone().numpy()
    pb.drain()
    for j, r in pb.results().items():
        if j not in got:
            got[j] = r.clone().numpy()
    assert sorted(got) == list(range(blocks)), sorted(got)
    assert all(nf == batch * B for nf in eng.calls)           # one Output hop per batch, over the whole ring
    if rank == 0:
        out_q.put(np.concatenate([got[j] for j in range(blocks)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("same_block", [False, True, "second_stream"])
@pytest.mark.parametrize("blocks,batch", [(1, 4), (2, 4), (7, 4), (8, 4), (13, 4), (26, 8), (33, 2), (1, 1), (9, 1)])
def test_pipelined_mix_bus_world2_every_block_matches_the_oracle(blocks, batch, same_block):
    """Two ranks, different channel shards, the batched collective path of bench.py --gpus N: rings, batches, the
    partly filled last batch and the drain, with the engine's delivery (same block / two calls late) faked on the host.
    Every block's bus equals the single-process oracle's sum over ALL channels divided by f32(0.0001 + N)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from __graft_entry__ import load_package
    import chains
    pkg = load_package()
    total, world = 301, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, total, blocks, batch, same_block, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, full = O.run_noise_channels([n.oracle_desc() for n in chains.chain5(pkg, delay=128)], 0x5EED0001, 0, total, 0, blocks,
                                   link_flags=3, want_out=False, want_mix=True)
    expect = (full.astype(np.float32) / O.link_divisor(total)).astype(np.float32)
    assert np.allclose(got, expect, rtol=1e-5, atol=1e-7)
