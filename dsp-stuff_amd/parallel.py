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
                 finish: Callable[[object, int, int], None], group=None, world: Optional[int] = None):
        self.total_channels = int(total_channels)
        self.n_frames = int(n_frames)
        self.finish = finish
        self.group = group
        if world is None:
            import torch.distributed as dist
            world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.world = world
        self._pending = None

    def submit(self, mix):
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
