import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ora
from path_trace_golang_amd import capi, hip, scene
sys.path.insert(0, "tests")
import test_edge_scenes_gpu as T
objs = [{"type": "plane", "position": T.V(0, 0, 0), "material_id": "d"},
        {"type": "sphere", "position": T.V(0, 1, 6), "size": T.V(2, 0, 0), "material_id": "g"},
        {"type": "box", "position": T.V(0, 1, 0), "size": T.V(1, 2, 1), "material_id": "m"}]
huge = dict(T.CAM, position=T.V(0, 1e12, 3e12), focus_dist=0, fov=1e-9)
doc = {"camera": huge, "sky": T.SKY, "objects": objs, "materials": T.MATS}
w, h, spp, depth, seed = 48, 32, 3, 6, 3
o = ora.render(ora.Scene(doc), w, h, spp, depth, seed=seed)
ctx = capi.Context(ndev=1)
img = np.zeros((h, w, 4), np.uint8); acc = np.zeros((h, w, 3)); nseg = np.zeros((h, w), np.uint32); ndraw = np.zeros((h, w), np.uint32)
st = hip.render(scene.Scene.decode(doc), hip.RenderConfig(w, h, spp, depth, seed, 0, capi.PT_FLAG_PIXEL_STATS), img, None, acc, nseg, ndraw, ctx=ctx)
print(os.environ.get("PTCORE_SCAN"), "gpu segs", st["segments"], "oracle", o["stats"]["segments"], "mismatch px", int((nseg != o["nseg"]).sum()),
      "scan mismatches", capi.load().pt_debug_scan_mismatches(ctx.handle))
ys, xs = np.nonzero(nseg != o["nseg"])
for y, x in list(zip(ys, xs))[:5]:
    print((x, y), nseg[y, x], o["nseg"][y, x], ndraw[y, x], o["ndraw"][y, x], acc[y, x], o["accum"][y, x])
