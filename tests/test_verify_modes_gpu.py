"""The self-checking instantiations of the trace kernel (PTCORE_SCAN=verify / verify_wide / verify_bvh): every scan is
done twice, by the culled strategy and by the reference's plain object-by-object loop (renderer.go:297-302), and
disagreements are counted (pt_debug_scan_mismatches).  Checks that (a) they count 0 on scenes of each size class and
(b) the counter can move: with candidate bits dropped on purpose (PTCORE_DEBUG_DROP) it must be > 0."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(monkeypatch, mode, sc, w, h, spp, depth, drop=None):
    from path_trace_golang_amd import capi, hip

    monkeypatch.setenv("PTCORE_SCAN", mode)
    if drop is None:
        monkeypatch.delenv("PTCORE_DEBUG_DROP", raising=False)
    else:
        monkeypatch.setenv("PTCORE_DEBUG_DROP", drop)
    L = capi.load()
    with capi.Context(ndev=1) as ctx:  # PTCORE_SCAN is read by pt_create
        before = L.pt_debug_scan_mismatches(ctx.handle)
        img = np.zeros((h, w, 4), np.uint8)
        st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 11), img, ctx=ctx)
        return L.pt_debug_scan_mismatches(ctx.handle) - before, st, img


@pytest.mark.parametrize("nobj", [40, 64, 100, 128])
def test_verify_wide_counts_nothing_on_grouped_scenes(monkeypatch, gpu_ctx, nobj):
    from path_trace_golang_amd import synth

    sc = synth.make_scene(nobj, seed=nobj)
    mm, st, img = _run(monkeypatch, "verify_wide", sc, 320, 180, 8, 8)
    assert st["segments"] > 320 * 180 * 8 and mm == 0
    # and the frame is the shipping strategy's frame
    monkeypatch.delenv("PTCORE_SCAN")
    from path_trace_golang_amd import hip
    ref = np.zeros_like(img)
    hip.render(sc, hip.RenderConfig(320, 180, 8, 8, 11), ref, ctx=gpu_ctx)
    assert np.array_equal(img, ref)


def test_verify_counter_moves_when_candidates_are_dropped(monkeypatch, gpu_ctx):
    """Negative control: with the first eight candidate bits of every group cleared, the culled scan loses real hits and
    the checker must say so -- for the single-group scan (reference scenes) and for the grouped one."""
    from conftest import scene_path
    from path_trace_golang_amd import scene, synth

    mm, _, _ = _run(monkeypatch, "verify", scene.load(scene_path("gpu_showcase")), 160, 90, 4, 8, drop="0xff")
    assert mm > 0
    mm, _, _ = _run(monkeypatch, "verify_wide", synth.make_scene(100, seed=3), 160, 90, 4, 8, drop="0xff")
    assert mm > 0
    # the same runs without the knob are clean
    mm, _, _ = _run(monkeypatch, "verify", scene.load(scene_path("gpu_showcase")), 160, 90, 4, 8)
    assert mm == 0
    mm, _, _ = _run(monkeypatch, "verify_wide", synth.make_scene(100, seed=3), 160, 90, 4, 8)
    assert mm == 0


def test_verify_bvh_counts_nothing(monkeypatch, gpu_ctx):
    from path_trace_golang_amd import synth

    mm, st, _ = _run(monkeypatch, "verify_bvh", synth.make_scene(600, seed=2), 160, 90, 4, 8)
    assert st["segments"] > 0 and mm == 0
