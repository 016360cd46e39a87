"""32x32 tile sharding of a frame over ranks/devices (host-side logic, pure Python/numpy).

The tile is the reference's own work granule (renderIntoCPU's tile queue,
/root/reference/internal/engine/renderer.go:132-157).  Tiles are numbered row-major,
t = ty*ntx + tx, and dealt round-robin: shard k of n owns t = k, k+n, k+2n, ... so
expensive image regions spread over all GPUs.  A shard's tile buffer holds its tiles in
that order, each tile as [32][32][C] row-major; pixels of edge tiles that fall outside the
frame are zero.  libptcore uses the same convention (pt_shard_tiles, pt_untile_device);
the numpy versions here exist so the multi-rank host path can be tested without a GPU.
"""
from __future__ import annotations

import numpy as np

TILE = 32


def tile_grid(width: int, height: int):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def shard_tiles(width: int, height: int, index: int, count: int):
    """Global tile ids owned by shard `index` of `count`, in buffer order."""
    ntx, nty = tile_grid(width, height)
    return list(range(index, ntx * nty, count))


def max_shard_tiles(width: int, height: int, count: int) -> int:
    ntx, nty = tile_grid(width, height)
    return (ntx * nty + count - 1) // count


def tile_rect(width: int, height: int, t: int):
    """Pixel rectangle (x0, y0, x1, y1) of tile t clipped to the frame."""
    ntx, _ = tile_grid(width, height)
    tx, ty = t % ntx, t // ntx
    return tx * TILE, ty * TILE, min((tx + 1) * TILE, width), min((ty + 1) * TILE, height)


def tile_from_frame(frame: np.ndarray, index: int, count: int, stride_tiles: int = 0) -> np.ndarray:
    """Packs a full frame [H, W, C] into shard `index`'s tile buffer [ntiles, 32, 32, C]."""
    h, w, c = frame.shape
    ids = shard_tiles(w, h, index, count)
    n = max(stride_tiles, len(ids))
    out = np.zeros((n, TILE, TILE, c), frame.dtype)
    for i, t in enumerate(ids):
        x0, y0, x1, y1 = tile_rect(w, h, t)
        out[i, : y1 - y0, : x1 - x0] = frame[y0:y1, x0:x1]
    return out


def untile(gathered, width: int, height: int, count: int, channels: int, dtype, stride_tiles: int = 0) -> np.ndarray:
    """Inverse of the gather: `gathered` is a list (or concatenation) of shard buffers in shard
    order; shard k starts at tile k*stride_tiles, or right after shard k-1 when stride_tiles == 0."""
    if isinstance(gathered, (list, tuple)):
        gathered = np.concatenate([np.asarray(g).reshape(-1, TILE, TILE, channels) for g in gathered])
    buf = np.asarray(gathered).reshape(-1, TILE, TILE, channels)
    frame = np.zeros((height, width, channels), dtype)
    before = 0
    for k in range(count):
        ids = shard_tiles(width, height, k, count)
        base = k * stride_tiles if stride_tiles else before
        for i, t in enumerate(ids):
            x0, y0, x1, y1 = tile_rect(width, height, t)
            frame[y0:y1, x0:x1] = buf[base + i, : y1 - y0, : x1 - x0]
        before += len(ids)
    return frame
