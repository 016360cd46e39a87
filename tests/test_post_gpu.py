"""HIP post-process passes vs the oracle restatement: byte-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h", [(64, 40), (3, 3), (2, 5), (257, 129)])
def test_post_passes_match_oracle(gpu_ctx, oracle, w, h):
    from path_trace_golang_amd import hip

    rng = np.random.default_rng(w * 1000 + h)
    acc = rng.gamma(0.7, 3.0, (h, w, 3)) * 7
    acc[rng.random((h, w)) < 0.02] = 0.0
    spp = 7
    combos = [dict(tonemap=True), dict(tonemap=True, denoise=True), dict(tonemap=True, denoise=True, smooth=True),
              dict(tonemap=True, denoise=True, sigma_s=0.6, sigma_r=0.4, smooth=True, smooth_radius=5, smooth_strength=0.8),
              dict(denoise=True), dict(smooth=True, smooth_radius=1, smooth_strength=1.0)]
    for kw in combos:
        start = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        start[..., 3] = 255
        a, b = start.copy(), start.copy()
        oracle.post_process(a, accum=acc if kw.get("tonemap") else None, spp=spp, **kw)
        hip.post_process(b, hip.PostConfig(**kw), acc if kw.get("tonemap") else None, spp, ctx=gpu_ctx)
        assert np.array_equal(a, b), kw


def test_post_on_a_render_and_strided_rows(gpu_ctx, oracle):
    from conftest import scene_path
    from path_trace_golang_amd import hip, scene

    w, h, spp = 96, 54, 8
    sc = scene.load(scene_path("gpu_showcase"))
    big = np.zeros((h, w + 5, 4), np.uint8)
    img = big[:, :w]
    acc = np.zeros((h, w, 3))
    hip.render(sc, hip.RenderConfig(w, h, spp, 6, 1), img, None, acc, ctx=gpu_ctx)
    ref = np.ascontiguousarray(img).copy()
    post = hip.PostConfig.from_env({})  # the reference's defaults: tone map + denoise, no smoothing
    assert post.tonemap and post.denoise and not post.smooth and (post.sigma_s, post.sigma_r) == (1.0, 0.15)
    hip.post_process(img, post, acc, spp, ctx=gpu_ctx)
    oracle.post_process(ref, tonemap=True, denoise=True, accum=acc, spp=spp)
    assert np.array_equal(np.ascontiguousarray(img), ref) and not big[:, w:].any()
    env = {"PATHTRACER_GPU_DENOISE": "off", "PATHTRACER_GPU_SMOOTH": "on", "PATHTRACER_GPU_SMOOTH_RADIUS": "9",
           "PATHTRACER_GPU_SMOOTH_STRENGTH": "0.25", "PATHTRACER_GPU_DENOISE_SIGMA_R": "x"}
    p2 = hip.PostConfig.from_env(env)
    assert (p2.denoise, p2.smooth, p2.smooth_radius, p2.smooth_strength, p2.sigma_r) == (False, True, 5, 0.25, 0.15)
