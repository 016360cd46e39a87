import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENE_NAMES = ["example_simple", "test_scene", "metal_glass_room", "gpu_showcase", "test_comprehensive"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name: str) -> str:
    return os.path.join(SCENES, name + ".json")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ora

    ora.lib()
    return ora


@pytest.fixture(scope="session")
def gpu_ctx():
    """One pt_ctx for the whole GPU session; fails loudly if the HIP library is missing."""
    # some gpu tests hand torch device memory to the C ABI: torch must load ITS libamdhip64.so.7 before
    # libptcore.so is loaded, so both share one HIP runtime (the loader matches the SONAME)
    import torch  # noqa: F401

    from path_trace_golang_amd import build, capi

    build.build_all()  # no-op when the in-tree artefacts are current (hipcc exists on the GPU box too)
    capi.load()
    if capi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the GPU box")
    ctx = capi.Context(ndev=1)
    yield ctx
    ctx.close()


# ---------------------------------------------------------------- HIP path vs oracle, both builds of every kernel
#
# PT_FLAG_PIXEL_STATS selects the STATS=true instantiations of the kernels (per-pixel segment / draw counters); what a
# host gets by default -- and what bench.py times -- are the STATS=false ones, which are different binaries (registers,
# occupancy, spills).  `render_vs_oracle` therefore renders BOTH and holds each to the parity bar of DESIGN.md section 4:
# totals equal, 8-bit image equal, FP64 sums within 4*depth*2^-52 relative (NaN where the oracle has NaN); with the flag
# also the per-pixel counters; and the two GPU frames bit-equal to each other.

def render_vs_oracle(ctx, sc, o, w, h, spp, depth, seed, chunk=0, window=None, tag="", forms=("stats", "shipping")):
    """Renders `sc` on the GPU in the given forms and compares each with the oracle output `o`
    (whole frame, or the oracle's `window` = (x0, y0, x1, y1) of it).  Returns {form: (img, acc, nseg, ndraw, st)}."""
    import numpy as np

    from path_trace_golang_amd import capi, hip

    sl = (slice(None), slice(None)) if window is None else (slice(window[1], window[3]), slice(window[0], window[2]))
    out = {}
    for form in forms:
        stats = form == "stats"
        img = np.zeros((h, w, 4), np.uint8)
        acc = np.zeros((h, w, 3))
        nseg = np.zeros((h, w), np.uint32) if stats else None
        ndraw = np.zeros((h, w), np.uint32) if stats else None
        st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed, chunk, capi.PT_FLAG_PIXEL_STATS if stats else 0), img, None,
                        acc, nseg, ndraw, ctx=ctx)
        t = (tag, form)
        if window is None:
            for k in ("samples", "segments", "exit_scans", "draws"):
                assert st[k] == o["stats"][k], (t, k, st[k], o["stats"][k])
        if stats:
            assert np.array_equal(nseg[sl], o["nseg"][sl]) and np.array_equal(ndraw[sl], o["ndraw"][sl]), t
        assert np.array_equal(img[sl], o["rgba"][sl]), (t, int(np.count_nonzero(img[sl] != o["rgba"][sl])))
        ref, got = o["accum"][sl], acc[sl]
        ok = (np.isnan(got) & np.isnan(ref)) | (np.abs(got - ref) <= 4 * max(depth, 1) * 2.0 ** -52 * np.maximum(np.abs(ref), 1e-300))
        assert np.all(ok), t
        out[form] = (img, acc, nseg, ndraw, st)
    if len(out) == 2:
        a, b = out["stats"], out["shipping"]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True), (tag, "the two builds differ")
        for k in ("samples", "segments", "exit_scans", "draws"):
            assert a[4][k] == b[4][k], (tag, k)
    return out
