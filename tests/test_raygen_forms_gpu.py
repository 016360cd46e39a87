"""Ray generation for a thin-lens camera (camera.go:60-74, math.go:74-84) exists in three forms: one job per lane
(`PTCORE_RAYGEN=simple`), a lane walking down its column of jobs (`column`, round 2) and the wave's jobs as one pool for the
rejection walk (the default since round 4).  The draws of a job and their order are the reference's in all of them, so pixels,
per-pixel draw counts and sums must be the oracle's -- on ragged frames too, where a wave's pool holds jobs that lie outside the
frame, with more samples per pixel than one pass holds, and on a frame cut into several passes."""
import numpy as np
import pytest

from conftest import render_vs_oracle, scene_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("form", ["pool", "column", "simple"])
@pytest.mark.parametrize("w,h,spp,chunk", [(96, 54, 5, 0), (33, 17, 9, 0), (70, 45, 11, 4), (1, 1, 130, 0)])
def test_every_form_of_the_lens_walk_gives_the_oracle_frame(monkeypatch, oracle, form, w, h, spp, chunk):
    from path_trace_golang_amd import capi, scene

    name, depth, seed = "gpu_showcase", 6, 21  # aperture 0.1
    monkeypatch.setenv("PTCORE_RAYGEN", form)
    o = oracle.render(oracle.Scene.load(scene_path(name)), w, h, spp, depth, seed=seed)
    assert int(o["ndraw"].min()) >= 5 * spp  # u, v and at least one lens attempt per sample
    with capi.Context(ndev=1) as ctx:
        render_vs_oracle(ctx, scene.load(scene_path(name)), o, w, h, spp, depth, seed, chunk=chunk, tag=form)


def test_a_wide_lens_and_a_closed_one(monkeypatch, oracle):
    """Aperture far larger than the scene (every ray leaves from somewhere else) and aperture 0 (no lens draws at all: the
    pinhole kernel) through the same code path of the host."""
    from path_trace_golang_amd import capi, scene

    w, h, spp, depth, seed = 64, 40, 7, 5, 3
    monkeypatch.delenv("PTCORE_RAYGEN", raising=False)
    for aperture in (25.0, 0.0):
        sc = scene.load(scene_path("metal_glass_room"))
        sc.camera.aperture = aperture
        o = oracle.render(oracle.Scene(sc.encode()), w, h, spp, depth, seed=seed)
        if aperture == 0.0:
            assert np.all(o["ndraw"] >= 2 * spp)
        with capi.Context(ndev=1) as ctx:
            render_vs_oracle(ctx, sc, o, w, h, spp, depth, seed, tag="aperture %g" % aperture)
