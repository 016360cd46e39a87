"""PTCORE_PIPELINE=walk32 (csrc/pt_walk32.h): the BVH walk in FP32 only -- it lists candidate objects and shrinks its bound
only by certain hits (cores) -- followed by the exact FP64 pass on the candidates.  The hierarchy, the FP32 arithmetic and
the cores only decide WHICH objects get the reference's test (objects.go:37-61, :141-179): the result must be the
oracle's, exactly, like every other form of the loop: per-pixel segment and draw counts, the 8-bit image, FP64 sums."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _render(monkeypatch, sc, w, h, spp, depth, seed, scan="bvh"):
    from path_trace_golang_amd import capi, hip

    monkeypatch.setenv("PTCORE_PIPELINE", "walk32")
    monkeypatch.setenv("PTCORE_SCAN", scan)
    with capi.Context(ndev=1) as ctx:  # pipeline and scan are chosen by pt_create
        img = np.zeros((h, w, 4), np.uint8)
        acc = np.zeros((h, w, 3))
        nseg = np.zeros((h, w), np.uint32)
        ndraw = np.zeros((h, w), np.uint32)
        st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed, 0, capi.PT_FLAG_PIXEL_STATS), img, None, acc, nseg, ndraw, ctx=ctx)
        mism = capi.load().pt_debug_scan_mismatches(ctx.handle)
    return st, img, acc, nseg, ndraw, mism


def _same_as_oracle(o, st, img, acc, nseg, ndraw, depth):
    assert st["segments"] == o["stats"]["segments"] and st["draws"] == o["stats"]["draws"] and st["exit_scans"] == o["stats"]["exit_scans"]
    assert np.array_equal(nseg, o["nseg"]) and np.array_equal(ndraw, o["ndraw"])
    assert np.array_equal(img, o["rgba"])
    ref = o["accum"]
    assert np.all(np.abs(acc - ref) <= 4 * depth * 2.0 ** -52 * np.maximum(np.abs(ref), 1e-300))


@pytest.mark.parametrize("name", ["gpu_showcase", "metal_glass_room", "test_comprehensive", "test_scene"])
def test_reference_scenes_through_the_fp32_walk(monkeypatch, oracle, gpu_ctx, name):
    """The reference's own scenes, forced onto the hierarchy: glass spheres and boxes, lights, lens."""
    from path_trace_golang_amd import scene

    w, h, spp, depth = 96, 54, 5, 9
    o = oracle.render(oracle.Scene.load(scene_path(name)), w, h, spp, depth, seed=4)
    st, img, acc, nseg, ndraw, _ = _render(monkeypatch, scene.load(scene_path(name)), w, h, spp, depth, 4)
    _same_as_oracle(o, st, img, acc, nseg, ndraw, depth)


@pytest.mark.parametrize("n,w,h,spp,depth", [(300, 64, 48, 4, 6), (3000, 64, 40, 3, 7)])
def test_synthetic_scenes_through_the_fp32_walk(monkeypatch, oracle, gpu_ctx, n, w, h, spp, depth):
    from path_trace_golang_amd import synth

    sc = synth.make_scene(n, seed=5)
    o = oracle.render(oracle.Scene(sc.encode()), w, h, spp, depth, seed=6)
    st, img, acc, nseg, ndraw, _ = _render(monkeypatch, sc, w, h, spp, depth, 6)
    _same_as_oracle(o, st, img, acc, nseg, ndraw, depth)


def test_verify_mode_counts_no_disagreement(monkeypatch, gpu_ctx):
    """PTCORE_SCAN=verify_bvh: the exact pass also runs the reference's loop over every object and counts the segments on
    which the two answers differ."""
    from path_trace_golang_amd import synth

    sc = synth.make_scene(20000, seed=3)
    before = _render(monkeypatch, sc, 16, 8, 1, 2, 1, scan="verify_bvh")[5]  # the counter is cumulative per process
    st, _, _, _, _, after = _render(monkeypatch, sc, 96, 64, 2, 6, 7, scan="verify_bvh")
    assert st["segments"] > 30000
    assert after - before == 0


@pytest.mark.parametrize("cam_scale", [3.0e3, 1.0e8])
def test_far_cameras_take_the_slow_list(monkeypatch, oracle, gpu_ctx, cam_scale):
    """Rays from thousands of scene sizes away are not walked in FP32: the FP64 traversal answers them (widened bounds)."""
    from path_trace_golang_amd import scene, synth

    sc = synth.make_scene(400, 11)
    doc = sc.encode()
    cam = doc["camera"]
    t = cam["target"]
    cam["position"] = {"x": t["x"] + 0.3 * cam_scale, "y": t["y"] + 0.5 * cam_scale, "z": t["z"] + cam_scale}
    cam["fov"] = 1500.0 / cam_scale  # the scene (about 30 units across) fills the frame
    cam["aperture"] = 0
    cam["focus_dist"] = 0
    sc2 = scene.Scene.decode(doc)
    w, h, spp, depth = 48, 32, 2, 5
    o = oracle.render(oracle.Scene(doc), w, h, spp, depth, seed=2)
    assert o["stats"]["segments"] > w * h * spp  # the camera does see the scene
    st, img, acc, nseg, ndraw, _ = _render(monkeypatch, sc2, w, h, spp, depth, 2)
    _same_as_oracle(o, st, img, acc, nseg, ndraw, depth)


def test_full_size_frame_equals_the_all_in_one_loop(monkeypatch, gpu_ctx):
    """10 000 objects at 1920x1080: millions of paths through the candidate lists, byte-equal to the all-in-one BVH loop."""
    from path_trace_golang_amd import hip, synth

    sc = synth.make_scene(10000, 1)
    w, h, spp, depth = 1920, 1080, 2, 8
    ref = np.zeros((h, w, 4), np.uint8)
    acc_ref = np.zeros((h, w, 3))
    st_ref = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 2), ref, None, acc_ref, ctx=gpu_ctx)
    st, img, acc, _, _, _ = _render(monkeypatch, sc, w, h, spp, depth, 2, scan="")
    assert st["segments"] == st_ref["segments"] and st["draws"] == st_ref["draws"]
    assert np.array_equal(img, ref) and np.array_equal(acc, acc_ref)
