#!/usr/bin/env python3
"""Soak: many frames of alternating configurations on one context; every repeat of a configuration must give
the same bytes and the device memory in use must stay flat.   python tools/soak.py [rounds]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from path_trace_golang_amd import capi, hip, scene, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ctx = capi.Context(ndev=1)
cases = [("gpu_showcase", 1920, 1080, 128, 8), ("test_scene", 800, 600, 64, 8), ("example_simple", 33, 17, 700, 5),
         ("metal_glass_room", 1280, 720, 64, 12), ("synth:5000", 640, 360, 16, 6), ("test_comprehensive", 257, 129, 33, 16)]
scenes = {}
for name, *_ in cases:
    scenes[name] = hip.FlatScene(synth.make_scene(int(name.split(":")[1]), 4) if name.startswith("synth:") else scene.load("scenes/%s.json" % name))
seen, mem = {}, []
t0 = time.time()
for r in range(rounds):
    for name, w, h, spp, d in cases:
        img = np.zeros((h, w, 4), np.uint8)
        hip.render(scenes[name], hip.RenderConfig(w, h, spp, d, 9), img, ctx=ctx)
        dig = hashlib.sha256(img.tobytes()).hexdigest()
        assert seen.setdefault(name, dig) == dig, "frame of %s changed in round %d" % (name, r)
    free, total = torch.cuda.mem_get_info(0)
    mem.append(total - free)
print("%d frames in %.1f s; device memory in use after each round (GiB): %s"
      % (rounds * len(cases), time.time() - t0, " ".join("%.2f" % (m / 2 ** 30) for m in mem)))
assert max(mem[2:]) - min(mem[2:]) < 64 << 20, "device memory is not flat"
print("soak ok")
