"""The split form of the render loop: dielectric hits leave the trace kernel through a path-state queue, glass_kernel
scatters them, searches the exit (renderer.go:316-371) and hands survivors back through the continuation queue.  Whatever the
number of split rounds (PTCORE_SPLIT_ROUNDS; 0 = the all-in-one loop of round 1), pixels, per-pixel counters and sums must be
the oracle's."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu

V = lambda x, y, z: {"x": x, "y": y, "z": z}  # noqa: E731
CAM = {"position": V(0, 1.2, 5), "target": V(0, 1, 0), "up": V(0, 1, 0), "fov": 45, "aperture": 0.05, "focus_dist": 5, "aspect_ratio": 0}
SKY = {"type": "gradient", "horizon": {"r": 1, "g": 1, "b": 1}, "zenith": {"r": 0.4, "g": 0.6, "b": 1.0}}
MATS = [{"id": "d", "type": "lambert", "albedo": {"r": 0.7, "g": 0.6, "b": 0.5}},
        {"id": "g", "type": "dielectric", "ior": 1.5, "albedo": {"r": 1, "g": 1, "b": 1}},
        {"id": "t", "type": "dielectric", "ior": 1.33, "albedo": {"r": 1, "g": 1, "b": 1}, "absorption": {"r": 0.3, "g": 0.05, "b": 0.0}},
        {"id": "m", "type": "metal", "albedo": {"r": 0.9, "g": 0.9, "b": 0.9}, "rough": 0.2},
        {"id": "e", "type": "emissive", "emit": {"r": 1, "g": 0.9, "b": 0.8}, "power": 6}]


def _render(monkeypatch, rounds, sc, w, h, spp, depth, seed, stats=True):
    from path_trace_golang_amd import capi, hip

    if rounds is None:
        monkeypatch.delenv("PTCORE_SPLIT_ROUNDS", raising=False)
    else:
        monkeypatch.setenv("PTCORE_SPLIT_ROUNDS", str(rounds))
    with capi.Context(ndev=1) as ctx:  # read by pt_create
        img = np.zeros((h, w, 4), np.uint8)
        acc = np.zeros((h, w, 3))
        nseg = np.zeros((h, w), np.uint32) if stats else None
        ndraw = np.zeros((h, w), np.uint32) if stats else None
        st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed, 0, capi.PT_FLAG_PIXEL_STATS if stats else 0), img, None, acc, nseg,
                        ndraw, ctx=ctx)
    return st, img, acc, nseg, ndraw


def _same_as_oracle(o, st, img, acc, nseg, ndraw, depth):
    assert st["segments"] == o["stats"]["segments"] and st["draws"] == o["stats"]["draws"] and st["exit_scans"] == o["stats"]["exit_scans"]
    if nseg is not None:
        assert np.array_equal(nseg, o["nseg"]) and np.array_equal(ndraw, o["ndraw"])
    assert np.array_equal(img, o["rgba"])
    ref = o["accum"]
    assert np.all(np.abs(acc - ref) <= 4 * depth * 2.0 ** -52 * np.maximum(np.abs(ref), 1e-300))


@pytest.mark.parametrize("rounds", [0, 1, 2, 3, 12])
def test_any_number_of_split_rounds_gives_the_oracle_frame(monkeypatch, oracle, gpu_ctx, rounds):
    from path_trace_golang_amd import scene

    name, w, h, spp, depth = "metal_glass_room", 96, 54, 6, 12
    o = oracle.render(oracle.Scene.load(scene_path(name)), w, h, spp, depth, seed=8)
    st, img, acc, nseg, ndraw = _render(monkeypatch, rounds, scene.load(scene_path(name)), w, h, spp, depth, 8)
    _same_as_oracle(o, st, img, acc, nseg, ndraw, depth)
    st2, img2, acc2, _, _ = _render(monkeypatch, rounds, scene.load(scene_path(name)), w, h, spp, depth, 8, stats=False)
    _same_as_oracle(o, st2, img2, acc2, None, None, depth)  # the build that ships
    assert np.array_equal(acc, acc2)
    if rounds == 0:
        assert st["glass_events"] == 0 and st["glass_launches"] == 0
    else:
        assert st["glass_events"] > 0 and st["continuations"] > 0 and st["glass_launches"] >= 1


def test_paths_that_creep_through_a_glass_box_use_every_round(monkeypatch, oracle, gpu_ctx):
    """A glass box fills the view: a ray inside a box hits it again at t = tMin on every level (objects.go:141-222 returns the
    entry point only), so a path parks in the glass queue once per level until the depth runs out -- queues, windows and
    the all-in-one pass behind the split rounds all see work; a tinted glass sphere inside adds Beer-Lambert epilogues."""
    from path_trace_golang_amd import scene

    objs = [{"type": "plane", "position": V(0, 0, 0), "material_id": "d"},
            {"type": "box", "position": V(0, 1.2, 1.5), "size": V(6, 3, 1.5), "material_id": "g"},
            {"type": "sphere", "position": V(-0.8, 1, -1), "size": V(0.8, 0, 0), "material_id": "t"},
            {"type": "sphere", "position": V(1, 0.7, -0.5), "size": V(0.7, 0, 0), "material_id": "m"},
            {"type": "sphere_light", "position": V(0, 4, 0), "size": V(0.6, 0, 0), "material_id": "e"}]
    doc = {"camera": CAM, "sky": SKY, "objects": objs, "materials": MATS}
    w, h, spp, depth = 80, 48, 5, 10
    o = oracle.render(oracle.Scene(doc), w, h, spp, depth, seed=3)
    assert o["stats"]["exit_scans"] > 2 * w * h * spp  # glass nearly everywhere, several bounces deep
    for rounds in (None, 4):
        for stats in (True, False):
            st, img, acc, nseg, ndraw = _render(monkeypatch, rounds, scene.Scene.decode(doc), w, h, spp, depth, 3, stats=stats)
            _same_as_oracle(o, st, img, acc, nseg, ndraw, depth)
            assert st["glass_events"] > w * h * spp


def test_full_size_frame_is_the_same_with_and_without_split_rounds(monkeypatch, gpu_ctx):
    """C3 at reduced spp: tens of millions of paths through the queues (many windows per wave, holes, the continuation intake
    of the next pass), byte-equal to the all-in-one loop."""
    from path_trace_golang_amd import scene

    sc = scene.load(scene_path("metal_glass_room"))
    w, h, spp, depth = 1920, 1080, 6, 12
    frames = {}
    for rounds in (0, 2):
        st, img, acc, _, _ = _render(monkeypatch, rounds, sc, w, h, spp, depth, 1, stats=False)  # the builds that ship
        frames[rounds] = (st, img, acc)
    a, b = frames[0], frames[2]
    assert a[0]["segments"] == b[0]["segments"] and a[0]["draws"] == b[0]["draws"] and a[0]["exit_scans"] == b[0]["exit_scans"]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
