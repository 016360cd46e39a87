#!/usr/bin/env python3
"""Launch-parameter sweep on C4 at 256 spp (each setting in its own process: the knobs are read at pt_create)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from path_trace_golang_amd import capi, hip, scene
ctx = capi.Context(ndev=1)
sc = scene.load(%r + "/scenes/gpu_showcase.json")
img = np.zeros((1080, 1920, 4), np.uint8)
cfg = hip.RenderConfig(1920, 1080, 256, 8, 1)
hip.render(sc, hip.RenderConfig(1920, 1080, 32, 8, 1), img, ctx=ctx)
best = 0
for _ in range(3):
    t = time.time(); st = hip.render(sc, cfg, img, ctx=ctx); dt = time.time() - t
    best = max(best, st["segments"] / dt / 1e6)
print("%%.0f" %% best)
''' % (ROOT, ROOT)
def run(env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e)
    return r.stdout.strip() or r.stderr.strip()[-200:]
print("default", run({}), flush=True)
for claim in (64, 128, 512, 1024, 4096):
    print("claim", claim, run({"PTCORE_CLAIM": str(claim)}), flush=True)
for b in (1, 2, 3, 4, 5, 6):
    print("blocks_per_cu", b, run({"PTCORE_BLOCKS_PER_CU": str(b)}), flush=True)
for mb in (1024, 3072, 12288, 24576):
    print("budget_mb", mb, run({"PTCORE_L_BUDGET_MB": str(mb)}), flush=True)
