"""Host-side logic of the multi-GPU path: one process per GPU, channels sharded by contiguous
ranges (no data-path collective), one exchange step for the mix bus.

The reference has no distributed anything; its "mix bus" is an
Output node averaging all connected pipes: sum over pipes, then divide by f32(0.0001 + n)
(dsp-stuff/src/node.rs:162-194, nodes/output.rs:215-249).  Sharded, that becomes
    per-rank partial sum [B]  ->  all-reduce(sum) over RCCL/xGMI  ->  / link_divisor(N_total).
The functions are backend-agnostic (`nccl` == RCCL on ROCm for the GPUs, `gloo` in the CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional


@dataclass(frozen=True)
class Shard:
    """Contiguous channel range owned by one rank."""
    rank: int
    world: int
    total_channels: int
    offset: int      # global index of local channel 0 == dspfx_engine_desc.channel_offset
    channels: int    # local N


def shard_channels(total_channels: int, world: int, rank: int) -> Shard:
    """Even contiguous split; the first `total % world` ranks own one extra channel."""
    if world < 1 or not (0 <= rank < world) or total_channels < 0:
        raise ValueError(f"bad shard request total={total_channels} world={world} rank={rank}")
    base, rem = divmod(total_channels, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return Shard(rank, world, total_channels, offset, count)


def weak_shard(channels_per_rank: int, world: int, rank: int) -> Shard:
    """Weak scaling: every rank owns `channels_per_rank` channels (bench.py)."""
    return Shard(rank, world, channels_per_rank * world, rank * channels_per_rank, channels_per_rank)


class MixBus:
    """Pipelined cross-rank mix bus.

    submit(mix) starts the all-reduce of one block's per-rank partial sums asynchronously and
    completes the previous block (wait + `finish(mix_prev, n_frames, total_channels)`), so the
    collective of block k overlaps the chain kernel of block k+1.  `finish` is the Output-node
    hop (node.rs:189-191): on the GPU it is `Engine.mix_finish`; it is a callback so this module
    holds no arithmetic of its own.  With world == 1 there is no collective and finish runs at once.
    """

    def __init__(self, total_channels: int, n_frames: int,
                 finish: Callable[[object, int, int], None], group=None, world: Optional[int] = None,
                 allreduce: Optional[Callable[[object, int, int], None]] = None):
        self.total_channels = int(total_channels)
        self.n_frames = int(n_frames)
        self.finish = finish
        self.group = group
        # allreduce(mix, n_frames, total_channels): the C-ABI collective (Engine.mix_allreduce over a dsp_stuff_amd.Comm:
        # RCCL all-reduce + Output hop, stream-ordered on the caller's current stream).  When given it replaces the
        # torch.distributed call below -- the path a Rust / C++ host takes; torch.distributed stays for the CPU (gloo) tests.
        self.allreduce = allreduce
        if world is None:
            import torch.distributed as dist
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.world = world
        self._pending = None

    def submit(self, mix):
        if self.allreduce is not None:
            self.allreduce(mix, self.n_frames, self.total_channels)
            return
        if self.world == 1:
            self.finish(mix, self.n_frames, self.total_channels)
            return
        import torch.distributed as dist
        work = dist.all_reduce(mix, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._complete_pending()
        self._pending = (work, mix)

    def _complete_pending(self):
        if self._pending is not None:
            work, mix = self._pending
            work.wait()
            self.finish(mix, self.n_frames, self.total_channels)
            self._pending = None

    def drain(self):
        """Complete the last submitted block (call before reading its mix buffer)."""
        self._complete_pending()


class StreamOrder:
    """The ordering PipelinedMixBus needs between its two streams, on real HIP streams (torch wraps them).  Split out so
    that the ring / batch / drain logic can be driven without a GPU (HostOrder) in the world-size-2 gloo tests."""

    def __init__(self, torch, compute_stream, mix_stream):
        self.torch, self.cs, self.ms = torch, compute_stream, mix_stream
        self.compute_handle, self.mix_handle = compute_stream.cuda_stream, mix_stream.cuda_stream

    def mix_after_compute(self):
        """Everything queued on the compute stream so far happens before what the mix stream is given next (one event)."""
        ev = self.torch.cuda.Event()
        ev.record(self.cs)
        self.ms.wait_event(ev)

    def on_mix(self):
        return self.torch.cuda.stream(self.ms)

    def mark_mix(self):
        ev = self.torch.cuda.Event()
        ev.record(self.ms)
        return ev

    def compute_after(self, ev):
        """The compute stream waits for a mark of the mix stream (no packet when it has long passed)."""
        if not ev.query():
            self.cs.wait_event(ev)


class HostOrder:
    """StreamOrder for a host-only run: every call completes before it returns, so there is nothing to order."""
    compute_handle = mix_handle = 0

    def mix_after_compute(self):
        pass

    def on_mix(self):
        import contextlib
        return contextlib.nullcontext()

    def mark_mix(self):
        return None

    def compute_after(self, ev):
        pass


class PipelinedMixBus:
    """Cross-rank mix bus behind the chain kernels of a rank, which run back to back on the compute stream.

    Each launch leaves a rank-local, un-normalised bus in a row of a ring buffer: the bus of ITS OWN block when
    `same_block` (Engine.process_bus: the launch's last workgroups finish the sum), else of the block submitted two calls
    earlier (Engine.process_mixpipe, the in-kernel pipeline).  Every `batch` blocks ONE event marker is put on the compute
    stream, and the second stream all-reduces the `batch` rows in one RCCL call and applies the Output-node hop with the
    global channel count (`MixBus` over a [batch * n_frames] buffer).  Three rings decouple the streams: a ring is reused
    two submits after its own, when its collective and division have long finished (checked with an event query, no stall
    in steady state).  `results()` after `drain()` returns {block index: tensor view} for the blocks still in the rings.
    `order` = StreamOrder (GPU) or HostOrder (tests); `engine` needs process_bus / process_mixpipe / mixpipe_flush /
    mix_finish (/ mix_allreduce with a `comm`).

    batch = 1 with same_block (the default of bench.py --gpus N since round 4) is the Output node as the reference has it: the
    GLOBAL bus of every block, ready a few microseconds after the block's own samples -- the exchange (one kernel of one
    workgroup with the mailbox backend of dspfx_mix_allreduce) is queued on the COMPUTE stream right behind the chain kernel:
    no event, no second stream, no ring hand-over; north_star's "< 128-sample block latency" holds for the bus too.  Larger
    batches trade that for throughput: the bus arrives up to `batch` blocks late (batch = 8: 21 ms at 48 kHz).
    """

    def __init__(self, engine, total_channels: int, n_frames: int, compute_stream, mix_stream, world: int,
                 batch: int = 8, device=None, comm=None, same_block: bool = False, order=None, group=None,
                 exchange_on_compute: Optional[bool] = None):
        import torch
        self.torch, self.eng = torch, engine
        self.nf, self.batch = int(n_frames), int(batch)
        self.order = order if order is not None else StreamOrder(torch, compute_stream, mix_stream)
        self.same_block = bool(same_block)
        # one exchange per block on the compute stream itself (the default for same_block, batch = 1).  exchange_on_compute=False
        # (round 6) keeps the SAME bus -- block k's, exchanged once per block -- but queues the exchange on the second stream behind
        # one event, so block k + 1's chain kernel does not wait for it: the bus completes a few microseconds after its block's
        # samples, off the critical path (bench.py: scaling_forms.same_block_second_stream).
        self.inline = (self.same_block and self.batch == 1) if exchange_on_compute is None else (bool(exchange_on_compute) and self.same_block and self.batch == 1)
        self.lag = 0 if self.same_block else 2       # calls between a block's submission and its bus
        self.rings = [torch.zeros(self.batch * self.nf, dtype=torch.float32, device=device) for _ in range(3)]
        mh = self.order.compute_handle if self.inline else self.order.mix_handle
        # comm (dsp_stuff_amd.Comm): the collective goes through the C ABI (dspfx_mix_allreduce: RCCL + Output hop on the
        # second stream); without one, torch.distributed's all_reduce on that stream
        allreduce = (lambda m, nf, n: engine.mix_allreduce(comm, m, nf, n, mh)) if comm is not None else None
        self.bus = MixBus(total_channels, self.batch * self.nf,
                          lambda m, nf, n: engine.mix_finish(m, nf, n, mh), world=world,
                          allreduce=allreduce, group=group)
        self.count = 0            # blocks submitted since the last drain
        self.submitted = 0        # batches handed to the second stream
        self.events = {}          # batch index -> mark of the second stream after its submit
        self.held = 0             # blocks of the last drain that results() can still serve

    def _row(self, j):
        q, r = divmod(j, self.batch)
        return self.rings[q % 3][r * self.nf:(r + 1) * self.nf]

    def _submit(self, q):
        if self.inline:                              # stream order does it all: chain kernel, exchange + Output hop, next chain kernel
            self.bus.submit(self.rings[q % 3])
            self.bus.drain()
            self.submitted = q + 1
            return
        self.order.mix_after_compute()
        with self.order.on_mix():
            self.bus.submit(self.rings[q % 3])       # also completes batch q-1 (wait + Output hop)
            done = self.order.mark_mix()
        self.events[q] = done
        self.events.pop(q - 3, None)
        self.submitted = q + 1

    def step(self, x, out, side=None):
        j = self.count - self.lag                    # this call delivers the bus of block j
        row = None
        if j >= 0:
            q, r = divmod(j, self.batch)
            if not self.inline and r == 0 and (q - 2) in self.events:    # ring q % 3 last held batch q-3, finished by submit q-2
                self.order.compute_after(self.events[q - 2])
            row = self._row(j)
        if self.same_block:
            self.eng.process_bus(x, out, row, self.nf, n_connected=0, side=side, stream=self.order.compute_handle)
        else:
            self.eng.process_mixpipe(x, out, row, self.nf, n_connected=0, side=side, stream=self.order.compute_handle)
        self.count += 1
        if j >= 0 and (j + 1) % self.batch == 0:
            self._submit(j // self.batch)

    def drain(self):
        n = self.count
        if n:
            if not self.same_block:
                self.eng.mixpipe_flush(self._row(n - 2) if n >= 2 else None, self._row(n - 1), n_connected=0,
                                       stream=self.order.compute_handle)
            for q in range(self.submitted, (n - 1) // self.batch + 1):    # partly filled rings: unused rows ride along
                self._submit(q)
        with self.order.on_mix():
            self.bus.drain()
        self.held = n
        self.count, self.submitted = 0, 0
        self.events.clear()

    def results(self):
        """After drain(): {block index: view of its finished bus} for the blocks the three rings still hold (the last
        2 * batch + the partly filled batch at least)."""
        n = self.held
        first_batch = max(0, (n - 1) // self.batch - 2) if n else 0
        return {j: self._row(j) for j in range(first_batch * self.batch, n)}
