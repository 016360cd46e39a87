"""Edge-case scenes through the C ABI vs the oracle: empty worlds, odd geometry, rays with non-finite
components (which must follow the reference's sequential NaN behaviour through the plain-scan fallback)."""
import numpy as np
import pytest

from conftest import render_vs_oracle

pytestmark = pytest.mark.gpu

CAM = {"position": {"x": 0, "y": 1, "z": 6}, "target": {"x": 0, "y": 1, "z": 0}, "up": {"x": 0, "y": 1, "z": 0}, "fov": 50,
       "aperture": 0, "focus_dist": 6, "aspect_ratio": 0}
SKY = {"type": "gradient", "horizon": {"r": 1, "g": 1, "b": 1}, "zenith": {"r": 0.4, "g": 0.6, "b": 1.0}}
MATS = [{"id": "d", "type": "lambert", "albedo": {"r": 0.7, "g": 0.6, "b": 0.5}},
        {"id": "g", "type": "dielectric", "ior": 1.5, "albedo": {"r": 1, "g": 1, "b": 1}, "absorption": {"r": 0.2, "g": 0.1, "b": 0}},
        {"id": "m", "type": "metal", "albedo": {"r": 0.9, "g": 0.9, "b": 0.9}, "rough": 0.3},
        {"id": "r", "type": "lambert", "albedo": {"r": 0.6, "g": 0.2, "b": 0.2}, "rough": 0.5},
        {"id": "e", "type": "emissive", "emit": {"r": 1, "g": 1, "b": 1}, "power": 5}]


def V(x, y, z):
    return {"x": x, "y": y, "z": z}


def _check(gpu_ctx, oracle, doc, w=48, h=32, spp=3, depth=6, seed=3):
    from path_trace_golang_amd import scene

    sc = scene.Scene.decode(doc)
    o = oracle.render(oracle.Scene(doc), w, h, spp, depth, seed=seed)
    render_vs_oracle(gpu_ctx, sc, o, w, h, spp, depth, seed)  # the counting build and the one that ships
    return o


def test_empty_world_and_planes_only(gpu_ctx, oracle):
    _check(gpu_ctx, oracle, {"camera": CAM, "sky": SKY, "objects": [], "materials": []})
    _check(gpu_ctx, oracle, {"camera": CAM, "background": {"r": 0.2, "g": 0.3, "b": 0.4}, "objects": [], "materials": []})
    planes = [{"type": "plane", "position": V(0, 0, 0), "material_id": "d"}, {"type": "plane", "position": V(0, 3, 0), "material_id": "g"},
              {"type": "plane", "position": V(0, 5, 0), "material_id": "e"}]
    _check(gpu_ctx, oracle, {"camera": CAM, "sky": SKY, "objects": planes, "materials": MATS})


def test_odd_geometry(gpu_ctx, oracle):
    objs = [{"type": "plane", "position": V(0, 0, 0), "material_id": "d"},
            {"type": "sphere", "position": V(-1.5, 1, 0), "size": V(-0.8, 0, 0), "material_id": "m"},    # negative radius
            {"type": "sphere", "position": V(1.5, 1, 0), "size": V(0, 0, 0), "material_id": "d"},        # zero radius: 1/r = inf
            {"type": "box", "position": V(0, 0.5, 1), "size": V(-1, 1, 1), "material_id": "d"},          # min > max on x: never hit
            {"type": "box", "position": V(0, 2.5, 0), "size": V(1, 0, 1), "material_id": "m"},           # zero thickness: t1 <= t0
            {"type": "box", "position": V(0, 1, -2), "size": V(6, 2, 0.5), "material_id": "r"},          # rough lambert (unit-sphere draws)
            {"type": "sphere", "position": V(0, 1, 0), "size": V(0.7, 0, 0), "material_id": "g"},
            {"type": "sphere", "position": V(0, 1, 0), "size": V(0.7, 0, 0), "material_id": "g"},        # exact duplicate: every hit is a tie
            {"type": "sphere_light", "position": V(0, 4, 1), "size": V(0.5, 0, 0), "material_id": "e"}]
    _check(gpu_ctx, oracle, {"camera": CAM, "sky": SKY, "objects": objs, "materials": MATS}, spp=4, depth=8)


def test_coincident_boxes_and_touching_faces(gpu_ctx, oracle):
    # exact ties: two boxes sharing a face plane and overlapping in a corner, a sphere tangent to the floor
    objs = [{"type": "plane", "position": V(0, 0, 0), "material_id": "d"},
            {"type": "box", "position": V(-1, 1, 0), "size": V(2, 2, 2), "material_id": "d"},
            {"type": "box", "position": V(1, 1, 0), "size": V(2, 2, 2), "material_id": "m"},
            {"type": "box", "position": V(0, 1, 0), "size": V(2, 2, 2), "material_id": "r"},
            {"type": "sphere", "position": V(0, 3, 0), "size": V(1, 0, 0), "material_id": "g"},
            {"type": "box", "position": V(0, 2.5, 0), "size": V(1, 1, 1), "material_id": "g"}]
    _check(gpu_ctx, oracle, {"camera": CAM, "sky": SKY, "objects": objs, "materials": MATS}, spp=4, depth=10)


def test_camera_inside_glass_and_far_away(gpu_ctx, oracle):
    objs = [{"type": "plane", "position": V(0, 0, 0), "material_id": "d"},
            {"type": "sphere", "position": V(0, 1, 6), "size": V(2, 0, 0), "material_id": "g"},   # the camera sits inside this one
            {"type": "box", "position": V(0, 1, 0), "size": V(1, 2, 1), "material_id": "m"}]
    _check(gpu_ctx, oracle, {"camera": CAM, "sky": SKY, "objects": objs, "materials": MATS})
    far = dict(CAM, position=V(0, 300, 5000), focus_dist=0)  # far outside the scene cube: FP64 clip + re-based FP32 tests
    _check(gpu_ctx, oracle, {"camera": far, "sky": SKY, "objects": objs, "materials": MATS})
    huge = dict(CAM, position=V(0, 1e12, 3e12), focus_dist=0, fov=1e-9)
    _check(gpu_ctx, oracle, {"camera": huge, "sky": SKY, "objects": objs, "materials": MATS})


def test_degenerate_camera_gives_nan_rays_like_the_reference(gpu_ctx, oracle):
    # position == target: w = unit(0) = 0, u = unit(up x 0) = 0 -> every ray direction is the zero vector, a = 0 and
    # the sphere root is 0/0 = NaN, which the sequential loop "accepts" (objects.go:56-60 compares false).  The HIP
    # path must reproduce that through its plain-scan fallback.
    cam = dict(CAM, target=V(0, 1, 6), focus_dist=0)
    objs = [{"type": "plane", "position": V(0, 0, 0), "material_id": "d"},
            {"type": "sphere", "position": V(0, 1, 0), "size": V(1, 0, 0), "material_id": "d"},
            {"type": "box", "position": V(2, 1, 0), "size": V(1, 2, 1), "material_id": "m"},
            {"type": "sphere", "position": V(-2, 1, 0), "size": V(1, 0, 0), "material_id": "g"}]
    o = _check(gpu_ctx, oracle, {"camera": cam, "sky": SKY, "objects": objs, "materials": MATS}, w=16, h=12, spp=2, depth=5)
    assert o["stats"]["segments"] >= 16 * 12 * 2


def test_many_planes_and_more_than_32_of_a_kind(gpu_ctx, oracle):
    # 40 spheres exceed the 32-bit candidate mask of the flat scan: the scene takes the BVH path
    rng = np.random.default_rng(5)
    objs = [{"type": "plane", "position": V(0, 0, 0), "material_id": "d"}, {"type": "plane", "position": V(0, 9, 0), "material_id": "e"}]
    for i in range(40):
        p = rng.uniform([-3, 0.3, -3], [3, 3, 3])
        objs.append({"type": "sphere", "position": V(*p), "size": V(0.3, 0, 0), "material_id": "dgmr"[i % 4]})
    _check(gpu_ctx, oracle, {"camera": CAM, "sky": SKY, "objects": objs, "materials": MATS}, spp=3, depth=6)


def test_unknown_object_material_and_sky_types_follow_the_reference(gpu_ctx, oracle):
    """The reference's default branches through the HIP path: an unknown object type is skipped
    (objects.go:237-266), an unknown material type is lambert (materials.go:51-53), a sky block with an
    unknown type falls back to `background` (renderer.go:84-88)."""
    mats = MATS + [{"id": "v", "type": "velvet", "albedo": {"r": 0.3, "g": 0.8, "b": 0.4}, "rough": 0.2}]
    objs = [{"type": "plane", "position": V(0, 0, 0), "material_id": "d"},
            {"type": "torus", "position": V(0, 1, 2), "size": V(1, 1, 1), "material_id": "m"},          # dropped from the world
            {"type": "sphere", "position": V(-1.2, 1, 0), "size": V(0.9, 0, 0), "material_id": "v"},     # lambert, rough 0.2
            {"type": "box", "position": V(1.2, 0.75, 0), "size": V(1.5, 1.5, 1.5), "material_id": "v"},
            {"type": "sphere", "position": V(0, 2.2, -1), "size": V(0.6, 0, 0), "material_id": "nope"}]  # missing id: zero material
    weird_sky = {"type": "weird", "color": {"r": 9, "g": 9, "b": 9}, "horizon": {"r": 1, "g": 0, "b": 0}, "zenith": {"r": 0, "g": 0, "b": 1}}
    bg = {"r": 0.25, "g": 0.5, "b": 0.75}
    o = _check(gpu_ctx, oracle, {"camera": CAM, "sky": weird_sky, "background": bg, "objects": objs, "materials": mats}, spp=4, depth=6)
    # the torus really is gone: the same scene without it gives the same picture
    o2 = _check(gpu_ctx, oracle, {"camera": CAM, "sky": weird_sky, "background": bg, "objects": objs[:1] + objs[2:], "materials": mats},
                spp=4, depth=6)
    assert np.array_equal(o["rgba"], o2["rgba"])
    # a primary ray that leaves the scene returns `background`, not any of the sky block's colours
    top = o["accum"][0, 0] / 4.0
    assert np.allclose(top, [0.25, 0.5, 0.75], rtol=0, atol=1e-15)
    # each on its own as well
    _check(gpu_ctx, oracle, {"camera": CAM, "sky": SKY, "objects": objs[:2], "materials": mats})
    _check(gpu_ctx, oracle, {"camera": CAM, "sky": SKY, "objects": [objs[0], objs[2]], "materials": mats})
    _check(gpu_ctx, oracle, {"camera": CAM, "sky": weird_sky, "background": bg, "objects": [], "materials": []})


def test_non_finite_geometry_follows_the_reference_loop(gpu_ctx, oracle):
    # a plane whose point has an infinite x: (px - ox) * 0 = NaN, t = NaN, and `t < tMin || t > tMax` lets it through
    # (objects.go:107-112); the NaN then sits in `closest` for the objects after it.  Same for a sphere with a NaN centre.
    # The JSON loader refuses such numbers like encoding/json does, so they can only arrive through the C ABI: the test
    # patches the decoded scene.
    import copy

    from path_trace_golang_amd import capi, hip, scene

    inf, nan = float("inf"), float("nan")
    base = [{"type": "sphere", "position": V(-1.2, 1, 0), "size": V(0.8, 0, 0), "material_id": "m"},
            {"type": "box", "position": V(1.2, 0.6, 0), "size": V(1, 1.2, 1), "material_id": "d"}]
    w, h, spp, depth, seed = 32, 20, 2, 5, 3
    for typ, mat, axis, val in (("plane", "d", "x", inf), ("plane", "d", "z", -inf), ("sphere", "g", "x", nan), ("box", "d", "y", inf)):
        odd = {"type": typ, "position": V(0, 1, 0), "size": V(0.5, 1, 1), "material_id": mat}
        for objs, k in ((base + [odd], 2), ([odd] + base, 0)):
            doc = {"camera": CAM, "sky": SKY, "objects": copy.deepcopy(objs), "materials": MATS}
            sc = scene.Scene.decode(doc)
            setattr(sc.objects[k].position, axis, val)
            doc["objects"][k]["position"][axis] = val
            o = oracle.render(oracle.Scene(doc), w, h, spp, depth, seed=seed)
            render_vs_oracle(gpu_ctx, sc, o, w, h, spp, depth, seed, tag=(typ, axis, k))  # <*,*,SCAN_UNIFORM,false>, both builds
