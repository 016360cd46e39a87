#!/usr/bin/env python3
"""Runs the diagnostic (PTCORE_PROFILE=1) build on a workload and prints per-section shares.
   python tools/section_profile.py [scene] [w] [h] [spp] [depth]"""
import ctypes as C, os, sys
os.environ["PTCORE_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_trace_golang_amd import capi, hip, scene
name = sys.argv[1] if len(sys.argv) > 1 else "gpu_showcase"
w, h, spp, d = (int(x) for x in (sys.argv[2:6] if len(sys.argv) >= 6 else (1920, 1080, 42, 8)))
# "shade": hit record + emitted + scatter of every material kind in one section since round 2 (shade_hit); cosine, diel and
# unitdir are no longer timed apart; diel / exitpost only run in the all-in-one form (PTCORE_SPLIT_ROUNDS=0)
SEC = ["iter", "raygen", "hist0", "scan", "hist1", "hist2", "shade", "cos|node", "(diel)", "exitpost", "rr",
       "finish", "sky", "udir|test", "broad", "nar_sph", "nar_box", "plane"]
ctx = capi.Context(ndev=1)
if name.startswith("synth:"):
    from path_trace_golang_amd import synth
    sc = synth.make_scene(int(name.split(":")[1]), 1)
else:
    sc = scene.load("scenes/%s.json" % name)
img = np.zeros((h, w, 4), np.uint8)
st = hip.render(sc, hip.RenderConfig(w, h, spp, d, 1), img, ctx=ctx)
buf = (C.c_uint64 * (3 * len(SEC)))()
n = capi.load().pt_debug_profile(ctx.handle, buf, len(buf))
it_exec, it_lanes, it_cyc = buf[0], buf[1], buf[2]
print("%s %dx%d spp %d depth %d: trace %.1f ms, %d segs, %d exits" % (name, w, h, spp, d, st["trace_ms"], st["segments"], st["exit_scans"]))
print("%-10s %14s %10s %8s %12s %10s" % ("section", "wave-execs", "exec/iter", "lanes%", "cycles-share", "cyc/exec"))
for i, s in enumerate(SEC):
    e, l, c = buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]
    print("%-10s %14d %10.3f %8.1f %12.3f %10.0f" % (s, e, e / max(it_exec, 1), 100.0 * l / max(64 * e, 1), c / max(it_cyc, 1), c / max(e, 1)))

if name.startswith("synth:"):
    g = lambda i, k: buf[3 * SEC.index(i) + k]
    print("object batches per closest-hit scan: <4: %d  <16: %d  <64: %d  <256: %d  <1024: %d  >=1024: %d" % (
        g("hist0", 0), g("hist0", 2), g("hist1", 0), g("hist1", 2), g("hist2", 0), g("hist2", 2)))
    print("node visits (lanes): %d, of them in the LDS copy of the top of the tree: %d, in a copy four times as large: %d" % (
        g("nar_box", 1), g("(diel)", 0), g("(diel)", 1)))
    print("exit searches (even bins only): <4: %d  [16,64): %d  [256,1024): %d" % (g("hist0", 1), g("hist1", 1), g("hist2", 1)))

os.environ["PTCORE_VERBOSE"] = "1"
capi.load().pt_debug_scan_mismatches(ctx.handle)
