#!/usr/bin/env python3
"""PTCORE_SCAN=verify on BASELINE configs C2-C5 at their FULL sizes: both closest-hit strategies on every segment.
   python tools/verify_full_configs.py"""
import os, sys, time
os.environ.setdefault("PTCORE_SCAN", "verify")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_trace_golang_amd import capi, hip, scene
ctx = capi.Context(ndev=1)
L = capi.load()
total = 0
for tag, name, w, h, spp, d in [("C2", "test_scene", 800, 600, 256, 8), ("C3", "metal_glass_room", 1920, 1080, 1024, 12),
                                ("C4", "gpu_showcase", 1920, 1080, 1024, 8), ("C5", "test_comprehensive", 3840, 2160, 4096, 16)]:
    sc = scene.load("scenes/%s.json" % name)
    img = np.zeros((h, w, 4), np.uint8)
    t = time.time()
    st = hip.render(sc, hip.RenderConfig(w, h, spp, d, 1), img, ctx=ctx)
    total += st["segments"] + st["exit_scans"]
    print("%s %s %dx%d spp %d depth %d: %d scans so far, %d mismatches (cumulative), %.1f s"
          % (tag, name, w, h, spp, d, total, L.pt_debug_scan_mismatches(ctx.handle), time.time() - t), flush=True)
sys.exit(0 if L.pt_debug_scan_mismatches(ctx.handle) == 0 else 1)
