"""Job-buffer budget (PTCORE_L_BUDGET_MB, DESIGN 8 "Passes per frame"): the samples per pass follow from it, the frame does not
(ordered accumulation); a device that cannot give the budget halves the pass until the buffers fit and keeps that size for the
frames that follow instead of trying the full budget again on every frame."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def _frames(monkeypatch, budget_mb, name, w, h, spp, depth, seed, n=1):
    from path_trace_golang_amd import capi, hip, scene

    if budget_mb is None:
        monkeypatch.delenv("PTCORE_L_BUDGET_MB", raising=False)
    else:
        monkeypatch.setenv("PTCORE_L_BUDGET_MB", str(budget_mb))
    sc = hip.FlatScene(scene.load(scene_path(name)))
    out = []
    with capi.Context(ndev=1) as ctx:  # the budget is read by pt_create
        for _ in range(n):
            img = np.zeros((h, w, 4), np.uint8)
            acc = np.zeros((h, w, 3))
            st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed), img, None, acc, ctx=ctx)
            out.append((st, img, acc))
    return out


def test_the_frame_does_not_depend_on_the_budget(monkeypatch):
    name, w, h, spp, depth = "test_scene", 320, 180, 48, 8
    (st_a, img_a, acc_a), = _frames(monkeypatch, None, name, w, h, spp, depth, 5)
    (st_b, img_b, acc_b), = _frames(monkeypatch, 64, name, w, h, spp, depth, 5)
    assert st_a["spp_chunk"] == spp and 1 <= st_b["spp_chunk"] <= 4  # 64 MiB hold about three samples of every pixel slot
    assert st_b["trace_launches"] > st_a["trace_launches"]
    assert st_a["segments"] == st_b["segments"] and st_a["draws"] == st_b["draws"]
    assert np.array_equal(img_a, img_b)
    assert np.array_equal(acc_a, acc_b)  # same additions in the same order, whatever the pass size


def test_a_budget_beyond_the_device_halves_the_pass_once(monkeypatch, capfd):
    # 3840x2160: 8.36 M pixel slots x 128 spp x 310 B of job buffers (rays, radiance, two path-state queues) = 331 GB, more than
    # the 288 GB of an MI355X; half of it fits (a quarter, an eighth ... on a device that someone else is using too)
    from path_trace_golang_amd import capi, hip, scene

    name, w, h, spp, depth = "gpu_showcase", 3840, 2160, 128, 8
    monkeypatch.setenv("PTCORE_VERBOSE", "1")
    monkeypatch.setenv("PTCORE_L_BUDGET_MB", "400000")
    sc = hip.FlatScene(scene.load(scene_path(name)))
    big = []
    with capi.Context(ndev=1) as ctx:
        for frame in range(2):
            capfd.readouterr()
            img = np.zeros((h, w, 4), np.uint8)
            acc = np.zeros((h, w, 3))
            st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, 3), img, None, acc, ctx=ctx)
            halvings = capfd.readouterr().err.count("short of memory")
            if frame == 0:
                assert halvings >= 1 and st["spp_chunk"] == spp >> halvings
            else:
                assert halvings == 0 and st["spp_chunk"] == big[0][0]["spp_chunk"]  # starts from the size that fitted
            big.append((st, img, acc))
    monkeypatch.delenv("PTCORE_VERBOSE")
    (st_ref, img_ref, acc_ref), = _frames(monkeypatch, None, name, w, h, spp, depth, 3)
    for st, img, acc in big:
        assert st["segments"] == st_ref["segments"]
        assert np.array_equal(img, img_ref)
        assert np.array_equal(acc, acc_ref)
