#!/usr/bin/env python3
"""Quick throughput probe of the scene files at reduced sample counts (one GPU): python tools/probe.py"""
import sys, time, numpy as np
sys.path.insert(0,'.')
from path_trace_golang_amd import capi, hip, scene
ctx = capi.Context(ndev=1)
for name,w,h,spp,d in [("gpu_showcase",1920,1080,64,8),("gpu_showcase",1920,1080,256,8),("metal_glass_room",1920,1080,64,12),("test_scene",800,600,256,8),("test_comprehensive",1920,1080,64,16)]:
    sc = scene.load(f"scenes/{name}.json")
    img = np.zeros((h,w,4),np.uint8)
    cfg = hip.RenderConfig(w,h,spp,d,1)
    st = hip.render(sc,cfg,img,ctx=ctx)
    t=time.time(); st = hip.render(sc,cfg,img,ctx=ctx); dt=time.time()-t
    print(name,w,h,spp,d, "wall %.3fs trace %.1fms resolve %.1fms chunk %d launches %d"%(dt,st['trace_ms'],st['resolve_ms'],st['spp_chunk'],st['trace_launches']),
          "Mseg/s %.1f Msamp/s %.1f seg/samp %.2f exit/seg %.3f draws/seg %.2f"%(st['segments']/dt/1e6, st['samples']/dt/1e6, st['segments']/st['samples'], st['exit_scans']/st['segments'], st['draws']/st['segments']), flush=True)
