"""The N > 1 host path on CPU: two gloo ranks shard a frame by interleaved 32x32 tiles,
gather the per-tile framebuffers on rank 0 and untile.  Pixels come from the CPU oracle
(window renders of each owned tile), so the assembled frame must be byte-identical to a
single-rank render: each pixel is computed wholly on one rank."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, scene_path

W, H, SPP, DEPTH, SEED = 80, 70, 2, 4, 5   # 3 x 3 tiles, ragged right and bottom edges


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, into_views=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import ora
    from path_trace_golang_amd import distributed, tiling

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = ora.Scene.load(scene_path("example_simple"))
    stride = tiling.max_shard_tiles(W, H, world)
    frame = np.zeros((H, W, 4), np.uint8)
    for t in tiling.shard_tiles(W, H, rank, world):
        x0, y0, x1, y1 = tiling.tile_rect(W, H, t)
        r = ora.render(sc, W, H, SPP, DEPTH, seed=SEED, workers=1, window=(x0, y0, x1, y1), want=("rgba",))
        frame[y0:y1, x0:x1] = r["rgba"][y0:y1, x0:x1]
    tiles = torch.from_numpy(tiling.tile_from_frame(frame, rank, world, stride).reshape(-1))

    def untile(bufs, stride_tiles):
        return tiling.untile([b.numpy() for b in bufs], W, H, world, 4, np.uint8, stride_tiles)

    scratch = None
    packed = None
    if into_views and rank == 0:
        # what bench.py does: the receive list is views of one contiguous buffer, so the gather needs no copy before the untile
        packed = torch.empty(world * tiles.numel(), dtype=torch.uint8)
        scratch = list(packed.split(tiles.numel()))
    out = distributed.assemble_frame(tiles, W, H, rank, world, untile, scratch)
    if rank == 0:
        if into_views:  # the shards really landed back to back in `packed`
            flat = np.concatenate([b.numpy() for b in scratch])
            assert np.array_equal(flat, packed.numpy())
            assert scratch[1].data_ptr() == packed.data_ptr() + tiles.numel()
        np.save(out_path, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_matches_single_rank(tmp_path, oracle):
    import torch.multiprocessing as mp

    out = str(tmp_path / "frame.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    ref = oracle.render(oracle.Scene.load(scene_path("example_simple")), W, H, SPP, DEPTH, seed=SEED, want=("rgba",))
    assert np.array_equal(got, ref["rgba"])


@pytest.mark.timeout(300)
def test_two_rank_gather_into_views_of_one_buffer(tmp_path, oracle):
    import torch.multiprocessing as mp

    out = str(tmp_path / "frame.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out, True), nprocs=2, join=True)
    got = np.load(out)
    ref = oracle.render(oracle.Scene.load(scene_path("example_simple")), W, H, SPP, DEPTH, seed=SEED, want=("rgba",))
    assert np.array_equal(got, ref["rgba"])


def test_tiling_roundtrip_and_layout():
    from path_trace_golang_amd import tiling

    rng = np.random.default_rng(0)
    for (w, h, n) in [(80, 70, 2), (1920, 1080, 8), (33, 31, 3), (64, 64, 5), (400, 225, 4)]:
        frame = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
        ntx, nty = tiling.tile_grid(w, h)
        owned = [tiling.shard_tiles(w, h, k, n) for k in range(n)]
        assert sorted(t for o in owned for t in o) == list(range(ntx * nty))     # a partition
        assert max(len(o) for o in owned) == tiling.max_shard_tiles(w, h, n)
        assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1      # balanced
        for stride in (0, tiling.max_shard_tiles(w, h, n)):
            bufs = [tiling.tile_from_frame(frame, k, n, stride) for k in range(n)]
            assert np.array_equal(tiling.untile(bufs, w, h, n, 3, np.uint8, stride), frame)
    # reference tile grid sizes: 1080p -> 60 x 34 tiles
    assert tiling.tile_grid(1920, 1080) == (60, 34) and tiling.tile_grid(800, 600) == (25, 19)
