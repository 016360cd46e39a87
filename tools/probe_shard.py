#!/usr/bin/env python3
"""Time of ONE rank's share of the bench frame for world sizes 1, 2, 4, 8 on one GPU (what strong scaling can reach):
   python tools/probe_shard.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from path_trace_golang_amd import capi, hip, scene
L = capi.load()
ctx = capi.Context(devices=[0])
dev = torch.device("cuda", 0)
sc = scene.load("scenes/gpu_showcase.json")
flat = hip.FlatScene(sc)
W, H = 1920, 1080
cfg = hip.pt_config(hip.RenderConfig(W, H, 1024, 8, 1))
stream = torch.cuda.current_stream(dev)
base = None
for world in (1, 2, 4, 8):
    for rank in sorted({0, world - 1}):
        shard = capi.PtShard(rank, world)
        ntl = C.c_int32()
        capi.check(L.pt_shard_tiles(W, H, C.byref(shard), C.byref(ntl), None, None))
        tiles = torch.zeros(ntl.value * 4096, dtype=torch.uint8, device=dev)
        def step():
            st = capi.PtStats()
            capi.check(L.pt_render_tiles_device(ctx.handle, C.byref(flat.c), C.byref(cfg), C.byref(shard),
                                                C.c_void_p(tiles.data_ptr()), None, C.c_void_p(stream.cuda_stream), C.byref(st)))
            return st
        step(); torch.cuda.synchronize()
        t = time.perf_counter(); sts = [step() for _ in range(3)]; torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        if base is None: base = dt
        print("world %d rank %d: %d tiles, %.1f ms per frame share (ideal %.1f), %d launches, chunk %d spp -> efficiency %.3f"
              % (world, rank, ntl.value, dt * 1e3, base * 1e3 / world, sts[0].trace_launches, sts[0].spp_chunk, base / world / dt), flush=True)
