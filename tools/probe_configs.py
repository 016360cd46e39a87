#!/usr/bin/env python3
"""BASELINE configs C1-C5 on one GPU (C5 also at a reduced spp): segments/s, samples/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_trace_golang_amd import capi, hip, scene
ctx = capi.Context(ndev=1)
for tag, name, w, h, spp, d in [("C1", "example_simple", 256, 256, 16, 4), ("C2", "test_scene", 800, 600, 256, 8),
                                ("C3", "metal_glass_room", 1920, 1080, 1024, 12), ("C4", "gpu_showcase", 1920, 1080, 1024, 8),
                                ("C5/16 (256 of 4096 spp)", "test_comprehensive", 3840, 2160, 256, 16),
                                ("C5 (full, one GPU)", "test_comprehensive", 3840, 2160, 4096, 16)]:
    sc = scene.load("scenes/%s.json" % name)
    img = np.zeros((h, w, 4), np.uint8)
    cfg = hip.RenderConfig(w, h, spp, d, 1)
    hip.render(sc, hip.RenderConfig(w, h, min(spp, 256), d, 1), img, ctx=ctx)  # warm: allocates the frame's (up to 16 GiB of) job buffers
    t = time.time(); st = hip.render(sc, cfg, img, ctx=ctx); dt = time.time() - t
    print("%s %s %dx%d spp %d depth %d: %.3f s  %.0f M segments/s  %.0f M samples/s  (%.2f segments/sample, chunk %d, %d launches)"
          % (tag, name, w, h, spp, d, dt, st["segments"] / dt / 1e6, st["samples"] / dt / 1e6, st["segments"] / st["samples"],
             st["spp_chunk"], st["trace_launches"]), flush=True)
