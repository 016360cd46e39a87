"""The failure path of the path-state queues: a queue that cannot hold what a pass appends must fail the frame
with PT_ERR_STATE -- never write outside its buffers, never return a wrong image -- and leave the context usable.

PTCORE_DEBUG_QUEUE_CAP=<entries> (read when a frame opens, honoured only when set) allocates the queues with that many
entries instead of one per job plus the window slack, which is the only way to get an overflow: the host sizes the queues
from the very grids it launches (queue_slack in csrc/ptcore.hip).  Round 2 found out the hard way what an undersized queue
does without the bound checks (a GPU memory fault); this test is the proof that they hold."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _frame(ctx, sc, w, h, spp, depth, seed):
    from path_trace_golang_amd import capi, hip

    img = np.zeros((h, w, 4), np.uint8)
    acc = np.zeros((h, w, 3))
    nseg = np.zeros((h, w), np.uint32)
    ndraw = np.zeros((h, w), np.uint32)
    st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed, 0, capi.PT_FLAG_PIXEL_STATS), img, None, acc, nseg, ndraw, ctx=ctx)
    return st, img, acc, nseg, ndraw


@pytest.mark.parametrize("pipeline", ["mega", "wavefront"])
def test_queue_overflow_fails_the_frame_and_nothing_else(monkeypatch, oracle, gpu_ctx, pipeline):
    from path_trace_golang_amd import capi, scene

    sc = scene.load(scene_path("gpu_showcase"))  # BASELINE config 4's scene: glass, so both queues are in use
    w, h, spp, depth, seed = 320, 180, 6, 8, 5
    monkeypatch.setenv("PTCORE_PIPELINE", pipeline)
    with capi.Context(ndev=1) as ctx:  # the form is chosen by pt_create: split passes (default) or the wavefront form
        monkeypatch.setenv("PTCORE_DEBUG_QUEUE_CAP", "3000")  # the frame parks ~50 000 paths per pass
        with pytest.raises(capi.PtError) as ei:
            _frame(ctx, sc, w, h, spp, depth, seed)
        assert ei.value.code == capi.PT_ERR_STATE and "queue overflowed" in str(ei.value)
        # a second frame with the same undersized queues fails the same way (no state left behind by the first)
        with pytest.raises(capi.PtError):
            _frame(ctx, sc, w, h, spp, depth, seed)
        # and with the knob gone the same context renders the oracle's frame exactly
        monkeypatch.delenv("PTCORE_DEBUG_QUEUE_CAP")
        st, img, acc, nseg, ndraw = _frame(ctx, sc, w, h, spp, depth, seed)
    o = oracle.render(oracle.Scene.load(scene_path("gpu_showcase")), w, h, spp, depth, seed=seed)
    assert st["segments"] == o["stats"]["segments"] and st["draws"] == o["stats"]["draws"]
    assert np.array_equal(nseg, o["nseg"]) and np.array_equal(ndraw, o["ndraw"])
    assert np.array_equal(img, o["rgba"])
    assert np.all(np.abs(acc - o["accum"]) <= 4 * depth * 2.0 ** -52 * np.maximum(np.abs(o["accum"]), 1e-300))


def test_queue_overflow_in_the_bvh_primary_pass(monkeypatch, oracle):
    """BVH scenes: primary_bvh_kernel hands every path that goes on to the per-lane loop through the continuation queue;
    an undersized queue must fail the frame the same way."""
    from path_trace_golang_amd import capi, synth

    sc = synth.make_scene(300, 6)
    w, h, spp, depth, seed = 160, 90, 4, 6, 3
    monkeypatch.delenv("PTCORE_PIPELINE", raising=False)
    with capi.Context(ndev=1) as ctx:
        monkeypatch.setenv("PTCORE_DEBUG_QUEUE_CAP", "2000")  # the primary pass hands on ~50 000 paths
        with pytest.raises(capi.PtError) as ei:
            _frame(ctx, sc, w, h, spp, depth, seed)
        assert ei.value.code == capi.PT_ERR_STATE and "queue overflowed" in str(ei.value)
        monkeypatch.delenv("PTCORE_DEBUG_QUEUE_CAP")
        st, img, acc, nseg, ndraw = _frame(ctx, sc, w, h, spp, depth, seed)
    o = oracle.render(oracle.Scene(sc.encode()), w, h, spp, depth, seed=seed)
    assert st["segments"] == o["stats"]["segments"] and st["draws"] == o["stats"]["draws"]
    assert np.array_equal(nseg, o["nseg"]) and np.array_equal(ndraw, o["ndraw"]) and np.array_equal(img, o["rgba"])
