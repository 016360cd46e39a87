#!/usr/bin/env python3
"""Runs the PTCORE_SCAN=verify build on big workloads: both closest-hit strategies on every segment,
prints how many segments disagreed (must be 0)."""
import os, sys
os.environ.setdefault("PTCORE_SCAN", "verify")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_trace_golang_amd import capi, hip, scene
ctx = capi.Context(ndev=1)
L = capi.load()
total = 0
scale = int(os.environ.get("VERIFY_SCALE", "1"))
for name, w, h, spp, d in [("gpu_showcase", 1920, 1080, 32 * scale, 8), ("metal_glass_room", 1920, 1080, 32 * scale, 12),
                           ("test_comprehensive", 1920, 1080, 16 * scale, 16), ("test_scene", 800, 600, 64 * scale, 8),
                           ("example_simple", 400, 225, 64 * scale, 20)]:
    sc = scene.load("scenes/%s.json" % name)
    img = np.zeros((h, w, 4), np.uint8)
    st = hip.render(sc, hip.RenderConfig(w, h, spp, d, 7), img, ctx=ctx)
    mm = L.pt_debug_scan_mismatches(ctx.handle)
    total += st["segments"] + st["exit_scans"]
    print("%-20s %dx%d spp %d depth %d: %d scans so far, %d mismatches (cumulative)" % (name, w, h, spp, d, total, mm), flush=True)
sys.exit(0 if L.pt_debug_scan_mismatches(ctx.handle) == 0 else 1)
