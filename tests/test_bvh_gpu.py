"""Scenes with more than 128 spheres or 128 boxes take the BVH path, 33-128 the grouped candidate masks.  The hierarchy only decides
which objects get the exact FP64 test, so the result must equal the reference's linear scan (the oracle)
exactly: same per-pixel segment/draw counts, same 8-bit image."""
import numpy as np
import pytest

from conftest import render_vs_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,w,h,spp,depth", [(40, 64, 48, 4, 6), (65, 64, 48, 4, 6), (100, 64, 48, 4, 6), (300, 64, 48, 4, 6),
                                             (3000, 48, 32, 2, 5)])
def test_synthetic_scene_matches_oracle(gpu_ctx, oracle, n, w, h, spp, depth):
    # 40 / 65 / 100 objects: the grouped candidate masks (trace_kernel<*,*,5,*>); 300 / 3000: the hierarchy (<*,*,3,false>);
    # each in the counting build and in the one that ships
    from path_trace_golang_amd import synth

    sc = synth.make_scene(n, 4)
    o = oracle.render(oracle.Scene(sc.encode()), w, h, spp, depth, seed=4)
    render_vs_oracle(gpu_ctx, sc, o, w, h, spp, depth, 4, tag="%d objects" % n)


def test_large_scene_renders_and_is_device_count_invariant():
    # 20,000 objects: nodes and objects live in HBM; two virtual devices must give the same pixels as one
    from path_trace_golang_amd import capi, hip, synth

    sc = synth.make_scene(20000, 9)
    w, h, spp, depth = 160, 96, 2, 6
    outs = []
    for devs in ([0], [0, 0]):
        with capi.Context(devices=devs) as ctx:
            img = np.zeros((h, w, 4), np.uint8)
            acc = np.zeros((h, w, 3))
            st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 2), img, None, acc, ctx=ctx)
            assert st["samples"] == w * h * spp and np.all(np.isfinite(acc)) and np.all(img[..., 3] == 255)
            outs.append((img, acc, st["segments"]))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]


@pytest.mark.parametrize("cam_scale", [3.0e3, 3.0e5, 1.0e8, 1.0e11])
def test_far_cameras_through_the_bvh(gpu_ctx, oracle, cam_scale):
    # Seen from thousands of scene sizes away the reference's FP64 sphere discriminant cancels and reports
    # hits the geometry does not have; the BVH widens its bounds per ray so that those survive (clip_ray).
    from path_trace_golang_amd import capi, hip, scene, synth

    sc = synth.make_scene(400, 11)  # more than 128 of a kind: the hierarchy, not the grouped masks
    doc = sc.encode()
    cam = doc["camera"]
    t = cam["target"]
    cam["position"] = {"x": t["x"] + 0.3 * cam_scale, "y": t["y"] + 0.5 * cam_scale, "z": t["z"] + cam_scale}
    cam["fov"] = 1500.0 / cam_scale  # the scene (about 30 units across) fills the frame
    cam["aperture"] = 0
    cam["focus_dist"] = 0
    sc = scene.Scene.decode(doc)
    w, h, spp, depth, seed = 64, 48, 3, 6, 5
    o = oracle.render(oracle.Scene(doc), w, h, spp, depth, seed=seed)
    assert o["stats"]["segments"] > w * h * spp  # the camera does see the scene
    render_vs_oracle(gpu_ctx, sc, o, w, h, spp, depth, seed, tag="camera %g away" % cam_scale)


def test_geometrically_spaced_objects(gpu_ctx, oracle):
    # 200 spheres whose positions and radii grow by 1.2x each (twelve decades): the deepest tree the builder
    # makes, a scene bound that dwarfs the objects near the camera, and every ray "far" for most of them
    from path_trace_golang_amd import capi, hip, scene

    objs = [{"type": "plane", "position": {"x": 0, "y": -1, "z": 0}, "material_id": "d"}]
    for k in range(200):
        s = 1.2 ** k
        objs.append({"type": "sphere" if k % 3 else "box", "position": {"x": 0.02 * s, "y": 0.01 * s, "z": -0.05 * s},
                     "size": {"x": 0.004 * s, "y": 0.004 * s, "z": 0.004 * s}, "material_id": "g" if k % 5 == 0 else "d"})
    doc = {"camera": {"position": {"x": 0, "y": 0.2, "z": 1.5}, "target": {"x": 0, "y": 0, "z": -1}, "up": {"x": 0, "y": 1, "z": 0},
                      "fov": 60},
           "sky": {"type": "gradient", "horizon": {"r": 1, "g": 1, "b": 1}, "zenith": {"r": 0.4, "g": 0.6, "b": 1.0}},
           "materials": [{"id": "d", "type": "lambert", "albedo": {"r": 0.7, "g": 0.6, "b": 0.5}},
                         {"id": "g", "type": "dielectric", "ior": 1.5, "albedo": {"r": 1, "g": 1, "b": 1}}],
           "objects": objs}
    sc = scene.Scene.decode(doc)
    w, h, spp, depth, seed = 64, 48, 4, 8, 3
    o = oracle.render(oracle.Scene(doc), w, h, spp, depth, seed=seed)
    render_vs_oracle(gpu_ctx, sc, o, w, h, spp, depth, seed)


def test_straggler_threshold_does_not_change_pixels(oracle):
    # PTCORE_BVH_MIN_LANES decides when unfinished walks are put off to the wave's next trip (0: never,
    # 64: as soon as any lane is done); it is a scheduling knob only
    import os

    from path_trace_golang_amd import capi, hip, synth

    sc = synth.make_scene(700, 21)
    w, h, spp, depth, seed = 96, 64, 4, 8, 2
    o = oracle.render(oracle.Scene(sc.encode()), w, h, spp, depth, seed=seed)
    old = os.environ.get("PTCORE_BVH_MIN_LANES")
    try:
        for t in ("0", "1", "8", "40", "64"):
            os.environ["PTCORE_BVH_MIN_LANES"] = t
            with capi.Context(ndev=1) as ctx:
                img = np.zeros((h, w, 4), np.uint8)
                nseg = np.zeros((h, w), np.uint32)
                ndraw = np.zeros((h, w), np.uint32)
                st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed, 0, capi.PT_FLAG_PIXEL_STATS), img, None, None, nseg,
                                ndraw, ctx=ctx)
                assert st["segments"] == o["stats"]["segments"] and st["draws"] == o["stats"]["draws"], t
                assert np.array_equal(nseg, o["nseg"]) and np.array_equal(ndraw, o["ndraw"]), t
                assert np.array_equal(img, o["rgba"]), t
    finally:
        os.environ.pop("PTCORE_BVH_MIN_LANES", None)
        if old is not None:
            os.environ["PTCORE_BVH_MIN_LANES"] = old
