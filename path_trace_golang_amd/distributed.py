"""One-process-per-GPU frame assembly over torch.distributed.

The render path shards by tile with no data-path collective; the only exchange step is
collecting the per-tile framebuffers on rank 0: one gather of equal-sized uint8 buffers
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests), then an
untile.  Many-to-one over seven disjoint xGMI links, ~1 MB per peer at 1080p.
"""
from __future__ import annotations

from typing import Callable, List, Optional

from . import tiling


def gather_tiles(tiles, rank: int, world: int, scratch: Optional[List] = None):
    """Gathers every rank's tile buffer (same shape on all ranks) on rank 0.

    Returns the list of buffers in shard order on rank 0 (just [tiles] when world == 1) and
    None elsewhere.  `scratch` may hold preallocated receive buffers to keep the timed loop
    allocation-free.
    """
    if world == 1:
        return [tiles]
    import torch
    import torch.distributed as dist

    if rank == 0:
        bufs = scratch if scratch is not None else [torch.empty_like(tiles) for _ in range(world)]
        dist.gather(tiles, bufs, dst=0)
        return bufs
    dist.gather(tiles, None, dst=0)
    return None


def assemble_frame(tiles, width: int, height: int, rank: int, world: int,
                   untile: Callable, scratch: Optional[List] = None):
    """gather + untile; `untile(list_of_shard_buffers, stride_tiles)` runs on rank 0 only."""
    bufs = gather_tiles(tiles, rank, world, scratch)
    if rank != 0:
        return None
    return untile(bufs, tiling.max_shard_tiles(width, height, world))
