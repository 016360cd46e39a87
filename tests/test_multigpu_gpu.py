"""Multi-GPU paths on two or more PHYSICAL devices (skipped on a one-GPU box, so it runs the first time a multi-GPU
lease exists): the frame must be byte-identical for any device count, because every pixel's whole sample loop stays
on one device (32x32 tiles, the reference's own granule, renderer.go:132-157, owned by tile index mod N).

  * in process: pt_create with ordinals [0,1] and [0..n-1] -- per-device streams, hipMemcpyPeerAsync of the tile
    buffers to devices[0] over xGMI, untile there;
  * per device: pt_render_tiles_device with shard k on device k (what one bench.py rank does), the shard buffers
    brought to device 0 by torch (peer copy) and untiled with pt_untile_device.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _ndev():
    import torch

    return torch.cuda.device_count()  # counting devices does not initialise the GPU on this image


needs_two = pytest.mark.skipif(_ndev() < 2, reason="needs at least two physical GPUs")

W, H, SPP, DEPTH, SEED = 1920, 1080, 4, 8, 1  # BASELINE config 4 at reduced spp


def _one_device_frame(gpu_ctx):
    from path_trace_golang_amd import capi, hip, scene

    sc = scene.load(scene_path("gpu_showcase"))
    img = np.zeros((H, W, 4), np.uint8)
    acc = np.zeros((H, W, 3))
    st = hip.render(sc, hip.RenderConfig(W, H, SPP, DEPTH, SEED), img, None, acc, ctx=gpu_ctx)
    return sc, img, acc, st


@needs_two
def test_in_process_devices_with_the_rccl_gather(gpu_ctx, monkeypatch):
    """The same frame with the tiles collected through RCCL (PTCORE_GATHER=rccl: grouped ncclSend / ncclRecv over xGMI)."""
    from path_trace_golang_amd import capi, hip

    sc, img1, acc1, st1 = _one_device_frame(gpu_ctx)
    monkeypatch.setenv("PTCORE_GATHER", "rccl")
    n = _ndev()
    for devs in ([0, 1], list(range(n)), [1, 0]):
        with capi.Context(devices=devs) as ctx:
            assert capi.load().pt_debug_gather_mode(ctx.handle) == 1
            for _ in range(2):  # a communicator serves frame after frame
                img = np.zeros((H, W, 4), np.uint8)
                acc = np.zeros((H, W, 3))
                st = hip.render(sc, hip.RenderConfig(W, H, SPP, DEPTH, SEED), img, None, acc, ctx=ctx)
                assert st["num_devices"] == len(devs) and st["segments"] == st1["segments"]
                assert np.array_equal(img, img1), "frame differs on devices %s" % devs
                assert np.array_equal(acc, acc1)


@needs_two
def test_in_process_devices_give_the_one_device_frame(gpu_ctx):
    from path_trace_golang_amd import capi, hip

    sc, img1, acc1, st1 = _one_device_frame(gpu_ctx)
    n = _ndev()
    for devs in ([0, 1], list(range(n)), [1, 0]):
        with capi.Context(devices=devs) as ctx:
            img = np.zeros((H, W, 4), np.uint8)
            acc = np.zeros((H, W, 3))
            st = hip.render(sc, hip.RenderConfig(W, H, SPP, DEPTH, SEED), img, None, acc, ctx=ctx)
            assert st["num_devices"] == len(devs)
            assert st["segments"] == st1["segments"] and st["draws"] == st1["draws"] and st["samples"] == st1["samples"]
            assert np.array_equal(img, img1), "frame differs on devices %s" % devs
            assert np.array_equal(acc, acc1)
            # a second frame on the same context (buffers and peer mappings reused)
            img[:] = 0
            hip.render(sc, hip.RenderConfig(W, H, SPP, DEPTH, SEED), img, None, None, ctx=ctx)
            assert np.array_equal(img, img1)


@needs_two
def test_one_shard_per_device_then_untile_on_device_zero(gpu_ctx):
    import torch

    from path_trace_golang_amd import capi, hip, tiling

    L = capi.load()
    sc, img1, acc1, st1 = _one_device_frame(gpu_ctx)
    flat = hip.FlatScene(sc)
    cfg = hip.pt_config(hip.RenderConfig(W, H, SPP, DEPTH, SEED))
    n = _ndev()
    for world in sorted({2, n}):
        stride = tiling.max_shard_tiles(W, H, world)
        dev0 = torch.device("cuda", 0)
        packed = torch.zeros(world * stride * 4096, dtype=torch.uint8, device=dev0)
        packed_acc = torch.zeros(world * stride * 3072, dtype=torch.float64, device=dev0)
        segs = 0
        for k in range(world):
            dev = torch.device("cuda", k)
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev)
                t = torch.zeros(stride * 4096, dtype=torch.uint8, device=dev)
                a = torch.zeros(stride * 3072, dtype=torch.float64, device=dev)
                st = capi.PtStats()
                sh = capi.PtShard(k, world)
                with capi.Context(devices=[k]) as ctx:
                    capi.check(L.pt_render_tiles_device(ctx.handle, C.byref(flat.c), C.byref(cfg), C.byref(sh),
                                                        C.c_void_p(t.data_ptr()), C.c_void_p(a.data_ptr()),
                                                        C.c_void_p(stream.cuda_stream), C.byref(st)))
                    torch.cuda.synchronize(dev)
                segs += st.segments
                # the exchange step: shard k's tiles land at k * stride tiles of the buffer on device 0 (peer copy)
                packed[k * stride * 4096:(k + 1) * stride * 4096].copy_(t)
                packed_acc[k * stride * 3072:(k + 1) * stride * 3072].copy_(a)
        torch.cuda.synchronize(dev0)
        assert segs == st1["segments"]
        frame = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev0)
        facc = torch.zeros((H, W, 3), dtype=torch.float64, device=dev0)
        with torch.cuda.device(dev0):
            s0 = torch.cuda.current_stream(dev0)
            capi.check(L.pt_untile_device(gpu_ctx.handle, W, H, world, stride, C.c_void_p(packed.data_ptr()),
                                          C.c_void_p(packed_acc.data_ptr()), C.c_void_p(frame.data_ptr()), W * 4,
                                          C.c_void_p(facc.data_ptr()), C.c_void_p(s0.cuda_stream)))
            torch.cuda.synchronize(dev0)
        assert np.array_equal(frame.cpu().numpy(), img1), "world %d" % world
        assert np.array_equal(facc.cpu().numpy(), acc1)


def test_listing_one_device_twice_still_matches(gpu_ctx):
    """Runs on any box: the same frame through the in-process multi-device path with ordinal 0 listed twice."""
    from path_trace_golang_amd import capi, hip

    sc, img1, acc1, st1 = _one_device_frame(gpu_ctx)
    with capi.Context(devices=[0, 0]) as ctx:
        img = np.zeros((H, W, 4), np.uint8)
        st = hip.render(sc, hip.RenderConfig(W, H, SPP, DEPTH, SEED), img, None, None, ctx=ctx)
        assert st["segments"] == st1["segments"] and np.array_equal(img, img1)
