"""Synthetic large scenes in the reference's JSON scene schema.

`make_scene(n, seed)` returns a scene.Scene with n finite objects (spheres, sphere lights and
axis-aligned boxes of the five material kinds) scattered inside a closed room over a ground plane,
reproducibly from `seed`.  It is the only regime where the BVH path, HBM-resident nodes and
incoherent node fetches are exercised: the five scene files of the reference hold 11-44 objects.
"""
from __future__ import annotations

import numpy as np

from . import scene as scn


def make_scene(n: int, seed: int = 1, room: float = 40.0, light_every: int = 23) -> scn.Scene:
    rng = np.random.default_rng(seed)
    sc = scn.Scene(name="synthetic-%d-seed-%d" % (n, seed))
    sc.camera = scn.Camera(position=scn.Vec3(0.0, room * 0.3, room * 0.48), target=scn.Vec3(0.0, room * 0.2, 0.0),
                           up=scn.Vec3(0, 1, 0), fov=55.0, aperture=0.05, focus_dist=room * 0.5, aspect_ratio=0.0)
    sc.sky = scn.Sky(type="gradient", horizon=scn.Color(0.9, 0.9, 1.0), zenith=scn.Color(0.3, 0.5, 1.0))
    sc.background = scn.Color(0.0, 0.0, 0.0)
    mats = [
        scn.Material(id="floor", type="lambert", albedo=scn.Color(0.6, 0.6, 0.6)),
        scn.Material(id="wall", type="lambert", albedo=scn.Color(0.75, 0.7, 0.65)),
        scn.Material(id="red", type="lambert", albedo=scn.Color(0.8, 0.2, 0.2)),
        scn.Material(id="green", type="lambert", albedo=scn.Color(0.2, 0.8, 0.3)),
        scn.Material(id="blue", type="lambert", albedo=scn.Color(0.2, 0.3, 0.9)),
        scn.Material(id="gold", type="metal", albedo=scn.Color(1.0, 0.8, 0.3), rough=0.2),
        scn.Material(id="chrome", type="metal", albedo=scn.Color(0.9, 0.9, 0.9), rough=0.0),
        scn.Material(id="glass", type="dielectric", albedo=scn.Color(1, 1, 1), ior=1.5),
        scn.Material(id="tinted", type="dielectric", albedo=scn.Color(1, 1, 1), ior=1.5, absorption=scn.Color(0.3, 0.05, 0.0)),
        scn.Material(id="mirror", type="mirror", albedo=scn.Color(0.95, 0.95, 0.95)),
        scn.Material(id="lamp", type="emissive", emit=scn.Color(1.0, 0.9, 0.7), power=12.0),
    ]
    sc.materials = mats
    # one object in ten is glass (the reference scenes hold 1-5 glass objects among 11-44)
    surf = ["red", "green", "blue", "gold", "chrome", "mirror", "red", "green", "blue", "gold", "red", "green", "blue",
            "chrome", "mirror", "red", "green", "blue", "glass", "tinted"]
    h = room * 0.5
    objs = [scn.Object(id="ground", type="plane", position=scn.Vec3(0, 0, 0), size=scn.Vec3(0, 0, 0), material_id="floor")]
    wall_t = 0.5
    for name, pos, size in [("back", (0, h, -h), (room, room, wall_t)), ("left", (-h, h, 0), (wall_t, room, room)),
                            ("right", (h, h, 0), (wall_t, room, room)), ("top", (0, room, 0), (room, wall_t, room))]:
        objs.append(scn.Object(id="wall-" + name, type="box", position=scn.Vec3(*pos), size=scn.Vec3(*size), material_id="wall"))
    n_free = max(0, n - len(objs) + 1)  # the plane is not a finite object
    # object size shrinks with the count so that the room stays sparsely filled
    base = max(0.05, min(1.5, 0.22 * room / max(1.0, n_free) ** (1.0 / 3.0)))
    for i in range(n_free):
        p = rng.uniform([-h * 0.9, base, -h * 0.9], [h * 0.9, room * 0.8, h * 0.9])
        s = base * rng.uniform(0.5, 1.5)
        if light_every and i % light_every == 0:
            objs.append(scn.Object(id="l%d" % i, type="sphere_light", position=scn.Vec3(*p), size=scn.Vec3(s * 0.6, 0, 0),
                                   material_id="lamp"))
        elif rng.random() < 0.6:
            objs.append(scn.Object(id="s%d" % i, type="sphere", position=scn.Vec3(*p), size=scn.Vec3(s, 0, 0),
                                   material_id=surf[int(rng.integers(len(surf)))]))
        else:
            e = s * rng.uniform(0.6, 1.8, 3)
            objs.append(scn.Object(id="b%d" % i, type="box", position=scn.Vec3(*p), size=scn.Vec3(*e),
                                   material_id=surf[int(rng.integers(len(surf)))]))
    sc.objects = objs
    return sc
