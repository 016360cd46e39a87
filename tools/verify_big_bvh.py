#!/usr/bin/env python3
"""PTCORE_SCAN=verify_bvh on very large synthetic scenes at a small frame (the plain scan visits every object):
   python tools/verify_big_bvh.py [n ...]"""
import os, sys
os.environ["PTCORE_SCAN"] = "verify_bvh"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_trace_golang_amd import capi, hip, synth
ctx = capi.Context(ndev=1)
L = capi.load()
for n in [int(a) for a in sys.argv[1:]] or [100000, 1000000]:
    sc = hip.FlatScene(synth.make_scene(n, 2))
    w, h, spp, d = (256, 144, 4, 8) if n <= 100000 else (128, 72, 2, 8)
    img = np.zeros((h, w, 4), np.uint8)
    st = hip.render(sc, hip.RenderConfig(w, h, spp, d, 11), img, ctx=ctx)
    print("n=%d %dx%d spp %d: %d scans, %d mismatches (cumulative), trace %.0f ms"
          % (n, w, h, spp, st["segments"] + st["exit_scans"], L.pt_debug_scan_mismatches(ctx.handle), st["trace_ms"]), flush=True)
sys.exit(0 if L.pt_debug_scan_mismatches(ctx.handle) == 0 else 1)
