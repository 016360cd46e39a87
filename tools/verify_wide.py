#!/usr/bin/env python3
"""PTCORE_SCAN=verify_wide on synthetic 40 / 64 / 100 / 128-object scenes at 1920x1080: the grouped bitmask scan and
the plain object-by-object loop on every segment, disagreements counted (must be 0).  VERIFY_SPP scales the run."""
import os, sys
os.environ["PTCORE_SCAN"] = "verify_wide"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_trace_golang_amd import capi, hip, synth
ctx = capi.Context(ndev=1)
L = capi.load()
total = 0
spp = int(os.environ.get("VERIFY_SPP", "24"))
for n in (40, 64, 100, 128):
    sc = synth.make_scene(n, seed=n)
    img = np.zeros((1080, 1920, 4), np.uint8)
    st = hip.render(sc, hip.RenderConfig(1920, 1080, spp, 8, 7), img, ctx=ctx)
    total += st["segments"] + st["exit_scans"]
    print("synthetic %3d objects 1920x1080 spp %d depth 8 (verify_wide): %d scans so far, %d mismatches (cumulative)"
          % (n, spp, total, L.pt_debug_scan_mismatches(ctx.handle)), flush=True)
sys.exit(0 if L.pt_debug_scan_mismatches(ctx.handle) == 0 else 1)
