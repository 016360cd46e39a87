#!/usr/bin/env python3
"""Regenerates the golden fixtures from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py

For each of the five scene files: 64x36, 16 spp, depth 8, seed 1 -> <scene>.npz holding
rgba uint8 [36,64,4], accum float64 [36,64,3] (raw radiance sums), nseg / ndraw uint32 [36,64]
and the totals.  The reference has no fixtures of its own and its RNG is time-seeded, so these
pin the ORACLE (and through it the HIP path) against regressions; they are not reference output.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ora  # noqa: E402

CFG = dict(width=64, height=36, spp=16, depth=8, seed=1)
SCENES = ["example_simple", "test_scene", "metal_glass_room", "gpu_showcase", "test_comprehensive"]

if __name__ == "__main__":
    for name in SCENES:
        sc = ora.Scene.load(os.path.join(ROOT, "scenes", name + ".json"))
        r = ora.render(sc, CFG["width"], CFG["height"], CFG["spp"], CFG["depth"], seed=CFG["seed"])
        st = r["stats"]
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgba=r["rgba"], accum=r["accum"], nseg=r["nseg"],
                            ndraw=r["ndraw"], totals=np.array([st["samples"], st["segments"], st["exit_scans"],
                                                               st["draws"]], np.uint64),
                            cfg=np.array([CFG["width"], CFG["height"], CFG["spp"], CFG["depth"], CFG["seed"]], np.int64))
        print(name, st["segments"], st["draws"])
