#!/usr/bin/env python3
"""Throughput of the BVH path on synthetic scenes: python tools/probe_synth.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_trace_golang_amd import capi, hip, synth
ctx = capi.Context(ndev=1)
L = capi.load()
ns = [int(a) for a in sys.argv[1:]] or [1000, 10000, 100000, 1000000]
w, h, spp, d = 1920, 1080, 16, 8
for n in ns:
    t = time.time(); sc = hip.FlatScene(synth.make_scene(n, 1)); tg = time.time() - t
    img = np.zeros((h, w, 4), np.uint8)
    cfg = hip.RenderConfig(w, h, spp, d, int(os.environ.get("PROBE_SEED", "1")))
    t = time.time(); st = hip.render(sc, cfg, img, ctx=ctx); t1 = time.time() - t
    t = time.time(); st = hip.render(sc, cfg, img, ctx=ctx); dt = time.time() - t
    mm = L.pt_debug_scan_mismatches(ctx.handle) if os.environ.get("PTCORE_SCAN", "").startswith("verify") else -1
    kms = st["trace_ms"] + st["glass_ms"]  # scan / traversal passes + shading passes (glass_kernel or the wavefront form's)
    print("n=%d pipeline=%s gen %.1fs first %.2fs | %dx%d spp %d: wall %.3fs scan %.1fms shade %.1fms raygen %.1fms  kernel rate %.1f Mseg/s  wall rate %.1f Mseg/s  seg/samp %.2f  mismatches %d"
          % (n, os.environ.get("PTCORE_PIPELINE", "default"), tg, t1, w, h, spp, dt, st["trace_ms"], st["glass_ms"], st["raygen_ms"],
             st["segments"] / kms / 1e3, st["segments"] / dt / 1e6, st["segments"] / st["samples"], mm), flush=True)
