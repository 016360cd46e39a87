#!/usr/bin/env python3
"""What a Go host gets: pt_render through the C ABI exactly as INTEGRATION.md binds it -- library defaults, no
environment knobs, host image (the 8.3 MB RGBA8 frame is copied back inside the call) -- on BASELINE config 4 at full size
(gpu.go:2534-2546: one call per frame):
  * a fresh context: pt_create + the first frame (what the one-shot CLI pays: job buffers are allocated, the scene is uploaded);
  * five more frames on the same context (what ui/app.go:190 or any host that renders frame after frame gets).
Writes one JSON object (profiles/r04_pt_render_default.json when run by tools/r04_default_path.sh)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_trace_golang_amd import capi, hip, scene

for k in ("PTCORE_L_BUDGET_MB", "PTCORE_CLAIM", "PTCORE_SPLIT_ROUNDS"):
    if os.environ.get(k) and "--keep-env" not in sys.argv:
        raise SystemExit("unset %s: this measures the library's defaults" % k)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = hip.FlatScene(scene.load(os.path.join(root, "scenes", "gpu_showcase.json")))
w, h, spp, depth = 1920, 1080, 1024, 8
cfg = hip.RenderConfig(w, h, spp, depth, 1)
img = np.zeros((h, w, 4), np.uint8)
capi.load()
t0 = time.perf_counter()
ctx = capi.Context(ndev=1)
t1 = time.perf_counter()
st = hip.render(sc, cfg, img, ctx=ctx)
t2 = time.perf_counter()
frames = []
for _ in range(int(os.environ.get("FRAMES", "5"))):
    t = time.perf_counter()
    st = hip.render(sc, cfg, img, ctx=ctx)
    frames.append((time.perf_counter() - t) * 1e3)
steady = sorted(frames)[len(frames) // 2]
out = {"workload": "scenes/gpu_showcase.json %dx%d, %d spp, max depth %d, seed 1 (BASELINE config 4) through pt_render, host image, library defaults" % (w, h, spp, depth),
       "job_buffer_budget_mib": os.environ.get("PTCORE_L_BUDGET_MB", "library default"),
       "pt_create_ms": (t1 - t0) * 1e3, "first_frame_ms": (t2 - t1) * 1e3, "fresh_context_one_frame_ms": (t2 - t0) * 1e3,
       "later_frames_ms": frames, "steady_frame_ms_median": steady, "spp_chunk": st["spp_chunk"], "passes_per_frame": -(-spp // st["spp_chunk"]),
       "segments_per_frame": st["segments"], "msegments_per_s_steady": st["segments"] / steady / 1e3,
       "msegments_per_s_fresh_context": st["segments"] / ((t2 - t0) * 1e3) / 1e3,
       "device_ms_last_frame": st["device_ms"], "checksum_rgba": int(img.astype(np.uint64).sum())}
print(json.dumps(out))
