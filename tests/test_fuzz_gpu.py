"""Randomised scenes through every closest-hit strategy (candidate bitmasks, BVH, plain scan) and every form of the loop vs the oracle.
Objects overlap, touch, nest, share centres and sizes on a coarse grid on purpose: exact ties and rays that
start inside several objects are the cases where a culled strategy could differ from the sequential loop."""
import os

import numpy as np
import pytest

from conftest import render_vs_oracle

pytestmark = pytest.mark.gpu

MAT_KINDS = ["lambert", "metal", "dielectric", "emissive", "mirror"]


def _random_doc(rng, nobj):
    mats = []
    for i in range(rng.integers(1, 8)):
        k = MAT_KINDS[int(rng.integers(len(MAT_KINDS)))]
        m = {"id": "m%d" % i, "type": k, "albedo": dict(zip("rgb", rng.uniform(0.1, 1.0, 3).round(3).tolist())),
             "rough": float(rng.choice([0.0, 0.0, 0.05, 0.5, 1.0])), "ior": float(rng.choice([0.0, 1.1, 1.5, 2.4])),
             "emit": dict(zip("rgb", rng.uniform(0.2, 1.0, 3).round(3).tolist())), "power": float(rng.uniform(1, 8)),
             "absorption": dict(zip("rgb", rng.choice([0.0, 0.0, 0.2, 1.0], 3).tolist())),
             "smoothness": float(rng.choice([0.0, 0.0, 0.7, 1.0]))}
        mats.append(m)
    objs = []
    grid = lambda lo, hi: float(rng.integers(lo * 2, hi * 2 + 1)) / 2.0  # half-unit grid: coincident faces are common
    for i in range(nobj):
        kind = rng.choice(["sphere", "box", "box", "sphere", "sphere_light", "plane"], p=[0.3, 0.25, 0.15, 0.15, 0.1, 0.05])
        pos = {"x": grid(-3, 3), "y": grid(0, 4), "z": grid(-3, 3)}
        if kind == "box":
            size = {"x": grid(0, 3), "y": grid(0, 3), "z": grid(0, 3)}
        else:
            size = {"x": float(rng.choice([0.25, 0.5, 1.0, 1.5])), "y": 0, "z": 0}
        objs.append({"id": "o%d" % i, "type": str(kind), "position": pos, "size": size,
                     "material_id": "m%d" % int(rng.integers(len(mats) + 1))})  # sometimes a missing id
    cam = {"position": {"x": grid(-2, 2), "y": grid(1, 3), "z": 7.0}, "target": {"x": 0, "y": 1.5, "z": 0},
           "up": {"x": 0, "y": 1, "z": 0}, "fov": float(rng.choice([35, 60, 90])), "aperture": float(rng.choice([0, 0, 0.2])),
           "focus_dist": float(rng.choice([0, 7])), "aspect_ratio": float(rng.choice([0, 1.7777778]))}
    sky = [None, {"type": "gradient", "horizon": {"r": 1, "g": 1, "b": 1}, "zenith": {"r": 0.3, "g": 0.5, "b": 1}},
           {"type": "solid", "color": {"r": 0.7, "g": 0.8, "b": 0.9}}][int(rng.integers(3))]
    return {"camera": cam, "objects": objs, "materials": mats, "sky": sky, "background": {"r": 0.1, "g": 0.1, "b": 0.15}}


@pytest.fixture(scope="module")
def contexts():
    from path_trace_golang_amd import capi

    out = {}
    old = os.environ.get("PTCORE_SCAN")
    oldp = os.environ.get("PTCORE_PIPELINE")
    try:
        for mode in ("broad", "wide", "bvh", "uniform"):
            os.environ["PTCORE_SCAN"] = mode
            out[mode] = capi.Context(ndev=1)
        # the two pass forms of the loop over the hierarchy: the FP64 traversal pass, and the FP32 walk + exact pass (ties,
        # nested and coincident objects are where a bound that is not certain, or a core that is not inside, would show)
        os.environ["PTCORE_SCAN"] = "bvh"
        for pipe in ("wavefront", "walk32"):
            os.environ["PTCORE_PIPELINE"] = pipe
            out["bvh/" + pipe] = capi.Context(ndev=1)
    finally:
        for k, v in (("PTCORE_SCAN", old), ("PTCORE_PIPELINE", oldp)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    yield out
    for c in out.values():
        c.close()


def _one_random_scene(contexts, oracle, rng, nobj, w, h, spp, depth, seed):
    from path_trace_golang_amd import capi, hip, scene

    doc = _random_doc(rng, nobj)
    o = oracle.render(oracle.Scene(doc), w, h, spp, depth, seed=seed)
    sc = scene.Scene.decode(doc)
    for mode, ctx in contexts.items():
        # the counting build and the one that ships, of every strategy
        render_vs_oracle(ctx, sc, o, w, h, spp, depth, seed, tag=(mode, seed))
    return o["stats"]["segments"]


@pytest.mark.parametrize("seed", range(40))
def test_random_scenes_all_strategies(contexts, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(4):
        _one_random_scene(contexts, oracle, rng, int(rng.integers(1, 45)) if _ < 3 else int(rng.integers(45, 150)), 32, 20, 3, 7, seed + 1)


@pytest.mark.skipif(not os.environ.get("PT_SOAK_SECONDS"), reason="long run: PT_SOAK_SECONDS=<seconds> [PT_SOAK_SEED=<n>]")
def test_random_scenes_soak(contexts, oracle):
    # the fuzz above with fresh seeds for as long as asked (evidence runs: profiles/r02_fuzz_soak.txt)
    import time

    t0, budget = time.time(), float(os.environ["PT_SOAK_SECONDS"])
    rng = np.random.default_rng(int(os.environ.get("PT_SOAK_SEED", "90001")))
    scenes = segs = 0
    while time.time() - t0 < budget:
        nobj = int(rng.integers(1, 45)) if rng.random() < 0.7 else int(rng.integers(45, 300))
        w, h = int(rng.choice([8, 32, 33, 50])), int(rng.choice([8, 20, 33]))
        segs += _one_random_scene(contexts, oracle, rng, nobj, w, h, int(rng.choice([1, 3, 6])), int(rng.choice([2, 7, 12])),
                                  int(rng.integers(1, 1 << 40)))
        scenes += 1
        if scenes % 50 == 0:
            print("soak: %d scenes, %.0f s" % (scenes, time.time() - t0), flush=True)  # a silent GPU job is taken to be hung
    print("soak: %d random scenes x %d strategies, %d oracle segments each way, %.0f s, all equal" % (scenes, len(contexts), segs, time.time() - t0),
          flush=True)


def test_random_render_configurations_match_oracle(gpu_ctx, oracle):
    # 40 random (scene file, frame size, spp, depth, chunk, seed) combinations, including 0 spp, depth 0,
    # 1-pixel frames and chunk sizes that do not divide the sample count
    from conftest import SCENE_NAMES, scene_path
    from path_trace_golang_amd import capi, hip, scene

    rng = np.random.default_rng(20241)
    for case in range(40):
        name = SCENE_NAMES[int(rng.integers(len(SCENE_NAMES)))]
        w = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 97, 130]))
        h = int(rng.choice([1, 3, 8, 24, 32, 40, 65, 90]))
        spp = int(rng.choice([0, 1, 2, 5, 17, 64]))
        depth = int(rng.choice([0, 1, 2, 3, 4, 8, 12, 30]))
        chunk = int(rng.choice([0, 1, 3, 7, 100]))
        seed = int(rng.integers(1, 1 << 40))
        sc = scene.load(scene_path(name))
        o = oracle.render(oracle.Scene.load(scene_path(name)), w, h, spp, depth, seed=seed)
        tag = "case %d: %s %dx%d spp %d depth %d chunk %d seed %d" % (case, name, w, h, spp, depth, chunk, seed)
        render_vs_oracle(gpu_ctx, sc, o, w, h, spp, depth, seed, chunk=chunk, tag=tag)


def test_random_large_scenes_match_oracle(gpu_ctx, oracle):
    # 16 random synthetic scenes beyond the candidate bitmasks (BVH path) at random frame sizes, chunks and depths
    from path_trace_golang_amd import capi, hip, synth

    rng = np.random.default_rng(77031)
    for case in range(16):
        n = int(rng.choice([33, 40, 65, 100, 257, 600, 1500]))
        w = int(rng.choice([1, 17, 33, 64, 90]))
        h = int(rng.choice([2, 24, 32, 50]))
        spp = int(rng.choice([1, 3, 8]))
        depth = int(rng.choice([1, 3, 6, 10]))
        chunk = int(rng.choice([0, 1, 2, 5]))
        seed = int(rng.integers(1, 1 << 40))
        sc = synth.make_scene(n, int(rng.integers(1, 1000)))
        o = oracle.render(oracle.Scene(sc.encode()), w, h, spp, depth, seed=seed)
        tag = "case %d: %d objects %dx%d spp %d depth %d chunk %d seed %d" % (case, n, w, h, spp, depth, chunk, seed)
        render_vs_oracle(gpu_ctx, sc, o, w, h, spp, depth, seed, chunk=chunk, tag=tag)
