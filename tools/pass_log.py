#!/usr/bin/env python3
"""What every trace pass of a frame does (segments, parked paths, time): PTCORE_DEBUG_PASS_LOG=1 python tools/pass_log.py [scene w h spp depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PTCORE_DEBUG_PASS_LOG", "1")
import numpy as np
from path_trace_golang_amd import capi, hip, scene
a = sys.argv[1:]
name = a[0] if a else "gpu_showcase"
w, h, spp, d = (int(x) for x in a[1:5]) if len(a) >= 5 else (1920, 1080, 256, 8)
sc = scene.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", name + ".json"))
with capi.Context(ndev=1) as ctx:
    img = np.zeros((h, w, 4), np.uint8)
    for rep in range(2):
        print("--- frame %d" % rep, file=sys.stderr, flush=True)
        st = hip.render(sc, hip.RenderConfig(w, h, spp, d, 1), img, ctx=ctx)
    print({k: st[k] for k in ("segments", "samples", "trace_ms", "glass_ms", "raygen_ms", "spp_chunk")})
