"""Conservativeness of the FP32 culls under geometry built to graze (every FP32 bound is inflated by the margin
m = B / 4096 of the scene bound B and rounded outward; the claim is that no object the reference's FP64 tests accept is ever
culled).  Rendered frames and random scenes exercise that claim by accident; these cameras do it on purpose:

  * tiny fields of view aimed TANGENTIALLY at spheres, the frame a band of +-4 m around the limb (the line of sight passes
    within k*m of the radius for every k in [-4, 4], on both sides of the discriminant's zero);
  * along box faces (the slab parameter of the face's axis is 0 * inf or a huge number), through edges and corners (t0 == t1);
  * origins at 3.4 ... 4.1 scene sizes from the centre: the switch between "scan from the origin" and "clip against the scene
    cube, scan from the entry point" (clip_bound = 3.5 B) and the bound the FP32 error analysis holds for (4 B);
  * origins around the `far` threshold (reach * 3e-8 = m / 4, about 2000 scene sizes): below it the culls run, above it the
    bitmask scans take the reference's loop and the hierarchy widens every bound by the lane's own slack.

For every camera and every size class of scene (<= 32 of a kind: bitmask scan; <= 128: grouped masks; more: hierarchy):
the self-checking instantiation (PTCORE_SCAN=verify / verify_wide / verify_bvh: every scan ALSO by the reference's plain loop,
renderer.go:297-302, objects.go:37-61, :141-179) must count 0 disagreements, and the frame of the default strategy must be the
oracle's frame in the counting build and in the build that ships."""
import math
import os

import numpy as np
import pytest

from conftest import render_vs_oracle

pytestmark = pytest.mark.gpu

V = lambda x, y, z: {"x": float(x), "y": float(y), "z": float(z)}  # noqa: E731
MATS = [{"id": "d", "type": "lambert", "albedo": {"r": 0.7, "g": 0.6, "b": 0.5}},
        {"id": "g", "type": "dielectric", "ior": 1.5, "albedo": {"r": 1, "g": 1, "b": 1}, "absorption": {"r": 0.2, "g": 0.1, "b": 0}},
        {"id": "m", "type": "metal", "albedo": {"r": 0.9, "g": 0.9, "b": 0.9}, "rough": 0.0},
        {"id": "r", "type": "metal", "albedo": {"r": 0.8, "g": 0.7, "b": 0.6}, "rough": 0.4},
        {"id": "e", "type": "emissive", "emit": {"r": 1, "g": 1, "b": 1}, "power": 4}]
SKY = {"type": "gradient", "horizon": {"r": 1, "g": 1, "b": 1}, "zenith": {"r": 0.4, "g": 0.6, "b": 1.0}}
BOUND = 8.0           # every object lies inside [-8, 8]^3 (the far wall reaches it exactly): the library's B
M = BOUND / 4096.0    # its margin


def _objects(n_s, n_b, seed):
    """The probes (objects 1-6) first, then filler on a half-unit grid (coincident faces and centres are common)."""
    rng = np.random.default_rng(seed)
    objs = [{"type": "plane", "position": V(0, 0, 0), "material_id": "d"},
            {"type": "sphere", "position": V(0, 2, 0), "size": V(1.5, 0, 0), "material_id": "d"},        # 1: limb probe
            {"type": "sphere", "position": V(-4, 1, 2), "size": V(1, 0, 0), "material_id": "g"},         # 2: glass, tangent to the floor
            {"type": "sphere", "position": V(4.5, 3, -3), "size": V(0.25, 0, 0), "material_id": "m"},    # 3: small, far from the centre
            {"type": "box", "position": V(3, 1, 2), "size": V(2, 2, 2), "material_id": "d"},             # 4: faces at x = 2, 4; y = 0, 2
            {"type": "box", "position": V(-2, 2.5, -4), "size": V(3, 1, 1), "material_id": "g"},         # 5: glass slab
            {"type": "box", "position": V(0, 4, -7.5), "size": V(16, 8, 1), "material_id": "r"}]         # 6: wall out to the scene bound
    while sum(o["type"] != "box" for o in objs) - 1 < n_s:
        p = [float(rng.integers(-12, 13)) / 2, float(rng.integers(1, 12)) / 2, float(rng.integers(-12, 13)) / 2]
        objs.append({"type": "sphere_light" if rng.random() < 0.05 else "sphere", "position": V(*p),
                     "size": V(float(rng.choice([0.25, 0.5, 0.75])), 0, 0), "material_id": str(rng.choice(list("ddgmre")))})
    while sum(o["type"] == "box" for o in objs) < n_b:
        p = [float(rng.integers(-12, 13)) / 2, float(rng.integers(1, 12)) / 2, float(rng.integers(-12, 13)) / 2]
        s = [float(rng.integers(1, 4)) / 2 for _ in range(3)]
        objs.append({"type": "box", "position": V(*p), "size": V(*s), "material_id": str(rng.choice(list("ddgmr")))})
    return objs


def _cam(pos, target, half_height, aperture=0.0):
    d = math.dist(pos, target)
    fov = 2.0 * math.degrees(math.atan(half_height / d))
    return {"position": V(*pos), "target": V(*target), "up": V(0, 1, 0), "fov": fov, "aperture": aperture, "focus_dist": 0,
            "aspect_ratio": 0}


def _cameras():
    cams = {}
    # --- sphere limbs: the target is a point ON the sphere's silhouette as seen from the camera, the frame +-4 m around it
    for name, c, r, pos in (("limb, axis-aligned view", (0, 2, 0), 1.5, (0, 2, 6)),          # direction (0, 0, -1): two zero components
                            ("limb, oblique view", (0, 2, 0), 1.5, (5, 4.5, 6.5)),
                            ("limb of the glass sphere", (-4, 1, 2), 1.0, (1, 1.5, 7.5)),
                            ("limb of the small far sphere", (4.5, 3, -3), 0.25, (-6, 5, 7))):
        c, pos = np.array(c, float), np.array(pos, float)
        to_c = c - pos
        dist = np.linalg.norm(to_c)
        side = np.cross(to_c, [0.0, 1.0, 0.0])
        side /= np.linalg.norm(side)
        # tangent point: at angle asin(r / dist) off the axis; the tangent line touches the sphere at distance sqrt(dist^2 - r^2)
        tl = math.sqrt(dist * dist - r * r)
        ax = to_c / dist
        tangent_dir = ax * (tl / dist) + side * (r / dist)
        cams[name] = _cam(tuple(pos), tuple(pos + tangent_dir * tl), 4 * M)
    # --- box faces: looking exactly along the face x = 4 of box 4 (the ray's x never changes: dx = 0 on the centre column),
    #     along its top face, and along the floor-touching bottom face (y = 0: coincident with the plane)
    cams["along a box face (x = 4)"] = _cam((4, 1, 9), (4, 1, 2), 4 * M)
    cams["along a box top (y = 2)"] = _cam((-7, 2, 2.5), (3, 2, 2.5), 4 * M)
    cams["along the floor under a box"] = _cam((3, 0.0, 9), (3, 0.0, 2), 4 * M)
    # --- edges and corners: lines that touch the box in ONE point (outside it on either side of the contact: t0 == t1 there,
    #     a short chord on one side of the frame, a miss on the other)
    cams["tangent to a box edge"] = _cam((1, 5, 2.3), (4, 2, 2.3), 4 * M)           # edge x = 4, y = 2 of box 4; direction (1, -1, 0)
    cams["tangent to a box corner"] = _cam((0, 6, 1), (4, 2, 3), 4 * M)             # corner (4, 2, 3); direction (1, -1, 0.5)
    cams["tangent to the slab's corner"] = _cam((-4.5, 7, -4.7), (-0.5, 3, -3.5), 4 * M)  # glass slab corner; direction (1, -1, 0.3)
    cams["into a box corner"] = _cam((8, 6, 7), (4, 2, 3), 4 * M)                   # the three faces meet mid-frame; direction (-1, -1, -1)
    # --- origins around the clip switch (3.5 B) and the analysed range (4 B), looking back at the scene
    for k in (3.4, 3.5, 3.6, 4.0, 4.1):
        cams["origin at %.1f scene sizes" % k] = _cam((0.3 * BOUND, 0.2 * BOUND * k, BOUND * k), (0, 2, 0), 5.0)
    cams["origin at 3.5 scene sizes on the axis"] = _cam((0, 2, 3.5 * BOUND), (0, 2, 0), 5.0)
    # --- around the `far` threshold: reach * 3e-8 = m / 4  <=>  reach = B / 4096 / 1.2e-7 = 2034.5 B
    for k in (0.9, 0.99, 1.01, 1.1):
        z = 2034.5 * BOUND * k
        cams["origin at %.2f of the far threshold" % k] = _cam((0.01 * z, 0.02 * z, z), (0, 2, 0), 6.0)
    return cams


CAMERAS = _cameras()
CLASSES = {"verify": (10, 10), "verify_wide": (44, 40), "verify_bvh": (180, 160)}


@pytest.fixture(scope="module")
def verify_contexts():
    from path_trace_golang_amd import capi

    out = {}
    old = os.environ.get("PTCORE_SCAN")
    try:
        for mode in CLASSES:
            os.environ["PTCORE_SCAN"] = mode  # read by pt_create
            out[mode] = capi.Context(ndev=1)
    finally:
        if old is None:
            os.environ.pop("PTCORE_SCAN", None)
        else:
            os.environ["PTCORE_SCAN"] = old
    yield out
    for c in out.values():
        c.close()


@pytest.mark.parametrize("mode", list(CLASSES))
def test_grazing_cameras(gpu_ctx, oracle, verify_contexts, mode):
    from path_trace_golang_amd import capi, hip, scene

    L = capi.load()
    n_s, n_b = CLASSES[mode]
    objs = _objects(n_s, n_b, seed=7)
    w, h, spp, depth, seed = 64, 36, 3, 5, 9
    vctx = verify_contexts[mode]
    for name, cam in CAMERAS.items():
        doc = {"camera": cam, "sky": SKY, "objects": objs, "materials": MATS}
        sc = scene.Scene.decode(doc)
        o = oracle.render(oracle.Scene(doc), w, h, spp, depth, seed=seed)
        assert o["stats"]["segments"] >= w * h * spp, name
        # (1) the culled strategy against the reference's loop on every scan of the frame
        before = L.pt_debug_scan_mismatches(vctx.handle)
        img = np.zeros((h, w, 4), np.uint8)
        st = hip.render(sc, hip.RenderConfig(w, h, spp, depth, seed), img, ctx=vctx)
        assert L.pt_debug_scan_mismatches(vctx.handle) - before == 0, (mode, name)
        assert st["segments"] == o["stats"]["segments"] and np.array_equal(img, o["rgba"]), (mode, name)
        # (2) the default strategy for this scene size, counting build and shipping build, against the oracle
        render_vs_oracle(gpu_ctx, sc, o, w, h, spp, depth, seed, tag=(mode, name))


def test_the_probes_do_graze(oracle):
    """The cameras above are only worth something if their frames really straddle the surfaces: with the probes alone (no floor,
    no wall behind them) some primary rays of every tangential frame must hit and some must reach the sky."""
    objs = _objects(10, 10, seed=7)[1:6]
    for name, cam in CAMERAS.items():
        if not (name.startswith("limb") or name.startswith("along") or name.startswith("tangent")):
            continue
        doc = {"camera": cam, "sky": SKY, "objects": objs, "materials": MATS}
        o = oracle.render(oracle.Scene(doc), 64, 36, 1, 1, seed=9)
        hit = float((o["accum"].sum(axis=2) == 0).mean())  # depth 1: a hit contributes nothing, a miss the sky
        assert 0.02 < hit < 0.98, (name, hit)
