"""The wavefront form of the render loop (csrc/pt_wavefront.h, PTCORE_PIPELINE=wavefront): traversal, shading and exit
passes over path-state queues in HBM, one level of rayColorOpt's recursion (renderer.go:286-404) at a time, optionally
with the paths of a level reordered by direction octant and origin cell.  It is the A/B of the all-in-one loop and must
give the oracle's answer exactly, like it: per-pixel segment and draw counts, the 8-bit image, FP64 sums."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _render(monkeypatch, sc, w, h, spp, depth, seed, sort):
    from path_trace_golang_amd import capi, hip

    monkeypatch.setenv("PTCORE_PIPELINE", "wavefront")
    monkeypatch.setenv("PTCORE_WF_SORT", "1" if sort else "0")
    with capi.Context(ndev=1) as ctx:  # the pipeline is chosen by pt_create
        img = np.zeros((h, w, 4), np.uint8)
        acc = np.zeros((h, w, 3))
        nseg = np.zeros((h, w), np.uint32)
        ndraw = np.zeros((h, w), np.uint32)
        st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed, 0, capi.PT_FLAG_PIXEL_STATS), img, None, acc, nseg, ndraw, ctx=ctx)
    return st, img, acc, nseg, ndraw


def _same_as_oracle(o, st, img, acc, nseg, ndraw, depth):
    assert st["segments"] == o["stats"]["segments"] and st["draws"] == o["stats"]["draws"] and st["exit_scans"] == o["stats"]["exit_scans"]
    assert np.array_equal(nseg, o["nseg"]) and np.array_equal(ndraw, o["ndraw"])
    assert np.array_equal(img, o["rgba"])
    ref = o["accum"]
    assert np.all(np.abs(acc - ref) <= 4 * depth * 2.0 ** -52 * np.maximum(np.abs(ref), 1e-300))


@pytest.mark.parametrize("name", ["gpu_showcase", "metal_glass_room", "test_comprehensive"])
def test_reference_scenes_in_the_wavefront_form(monkeypatch, oracle, gpu_ctx, name):
    from path_trace_golang_amd import scene

    w, h, spp, depth = 96, 54, 5, 9
    o = oracle.render(oracle.Scene.load(scene_path(name)), w, h, spp, depth, seed=4)
    _same_as_oracle(o, *_render(monkeypatch, scene.load(scene_path(name)), w, h, spp, depth, 4, False), depth)


@pytest.mark.parametrize("name,depth", [("metal_glass_room", 14), ("gpu_showcase", 21)])
def test_deep_paths_take_the_early_out_of_the_level_loop(monkeypatch, oracle, gpu_ctx, name, depth):
    """From level 8 on the host reads the number of paths left every fourth level and stops when there are none
    (csrc/ptcore.hip, dev_step_wavefront): only reachable with max_depth >= 10 -- the "final" preset asks for 80."""
    from path_trace_golang_amd import scene

    w, h, spp = 64, 36, 3
    o = oracle.render(oracle.Scene.load(scene_path(name)), w, h, spp, depth, seed=8)
    _same_as_oracle(o, *_render(monkeypatch, scene.load(scene_path(name)), w, h, spp, depth, 8, False), depth)


@pytest.mark.parametrize("sort", [False, True])
def test_bvh_scene_in_the_wavefront_form(monkeypatch, oracle, gpu_ctx, sort):
    from path_trace_golang_amd import synth

    sc = synth.make_scene(700, seed=9)
    doc = sc.encode()
    w, h, spp, depth = 80, 45, 3, 7
    o = oracle.render(oracle.Scene(doc), w, h, spp, depth, seed=6)
    st, img, acc, nseg, ndraw = _render(monkeypatch, sc, w, h, spp, depth, 6, sort)
    _same_as_oracle(o, st, img, acc, nseg, ndraw, depth)


def test_wavefront_full_size_frame_equals_the_all_in_one_loop(monkeypatch, gpu_ctx):
    """C4 at reduced spp: millions of paths through the queues (windows, holes, several levels), byte-equal frames."""
    from path_trace_golang_amd import hip, scene

    sc = scene.load(scene_path("gpu_showcase"))
    w, h, spp, depth = 1920, 1080, 3, 8
    ref = np.zeros((h, w, 4), np.uint8)
    acc_ref = np.zeros((h, w, 3))
    st_ref = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 2), ref, None, acc_ref, ctx=gpu_ctx)
    st, img, acc, _, _ = _render(monkeypatch, sc, w, h, spp, depth, 2, False)
    assert st["segments"] == st_ref["segments"] and st["draws"] == st_ref["draws"]
    assert np.array_equal(img, ref) and np.array_equal(acc, acc_ref)
