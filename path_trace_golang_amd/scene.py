"""Scene model and JSON I/O: Python mirror of the reference's internal/scene package.

Types and field names follow /root/reference/internal/scene/scene.go:9-158; `load`
and `save` follow io.go:10-38.  Decoding behaves like Go's encoding/json into those
structs: absent keys leave zero values, keys match case-insensitively, `"sky": null`
(or no "sky") leaves Scene.sky = None, unknown keys are ignored.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import List, Optional

MATERIAL_LAMBERT = "lambert"
MATERIAL_METAL = "metal"
MATERIAL_DIELECTRIC = "dielectric"
MATERIAL_EMISSIVE = "emissive"
MATERIAL_MIRROR = "mirror"

OBJECT_SPHERE = "sphere"
OBJECT_PLANE = "plane"
OBJECT_BOX = "box"
OBJECT_SPHERE_LIGHT = "sphere_light"


def _get(d, key, default=None):
    if not isinstance(d, dict):
        return default
    if key in d:
        v = d[key]
    else:
        v = default
        lk = key.lower()
        for k, vv in d.items():
            if isinstance(k, str) and k.lower() == lk:
                v = vv
                break
    return default if v is None else v


def _f(d, key) -> float:
    return float(_get(d, key, 0.0))


def _i(d, key) -> int:
    return int(_get(d, key, 0))


@dataclass
class Vec3:  # scene.go:9-13
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0

    @classmethod
    def decode(cls, d) -> "Vec3":
        return cls(_f(d, "x"), _f(d, "y"), _f(d, "z"))

    def encode(self) -> dict:
        return {"x": self.x, "y": self.y, "z": self.z}

    def as_list(self):
        return [self.x, self.y, self.z]


@dataclass
class Color:  # scene.go:16-20
    r: float = 0.0
    g: float = 0.0
    b: float = 0.0

    @classmethod
    def decode(cls, d) -> "Color":
        return cls(_f(d, "r"), _f(d, "g"), _f(d, "b"))

    def encode(self) -> dict:
        return {"r": self.r, "g": self.g, "b": self.b}

    def as_list(self):
        return [self.r, self.g, self.b]


@dataclass
class Camera:  # scene.go:24-32
    position: Vec3 = field(default_factory=Vec3)
    target: Vec3 = field(default_factory=Vec3)
    up: Vec3 = field(default_factory=Vec3)
    fov: float = 0.0
    aperture: float = 0.0
    focus_dist: float = 0.0
    aspect_ratio: float = 0.0

    @classmethod
    def decode(cls, d) -> "Camera":
        return cls(Vec3.decode(_get(d, "position", {})), Vec3.decode(_get(d, "target", {})),
                   Vec3.decode(_get(d, "up", {})), _f(d, "fov"), _f(d, "aperture"), _f(d, "focus_dist"),
                   _f(d, "aspect_ratio"))

    def encode(self) -> dict:
        return {"position": self.position.encode(), "target": self.target.encode(), "up": self.up.encode(),
                "fov": self.fov, "aperture": self.aperture, "focus_dist": self.focus_dist,
                "aspect_ratio": self.aspect_ratio}


@dataclass
class Material:  # scene.go:41-63
    id: str = ""
    type: str = ""
    albedo: Color = field(default_factory=Color)
    rough: float = 0.0
    ior: float = 0.0
    emit: Color = field(default_factory=Color)
    power: float = 0.0
    absorption: Color = field(default_factory=Color)
    smoothness: float = 0.0
    reflectivity: float = 0.0
    tint: Color = field(default_factory=Color)
    absorption_scale: float = 0.0

    @classmethod
    def decode(cls, d) -> "Material":
        return cls(str(_get(d, "id", "")), str(_get(d, "type", "")), Color.decode(_get(d, "albedo", {})),
                   _f(d, "rough"), _f(d, "ior"), Color.decode(_get(d, "emit", {})), _f(d, "power"),
                   Color.decode(_get(d, "absorption", {})), _f(d, "smoothness"), _f(d, "reflectivity"),
                   Color.decode(_get(d, "tint", {})), _f(d, "absorption_scale"))

    def encode(self) -> dict:
        return {"id": self.id, "type": self.type, "albedo": self.albedo.encode(), "rough": self.rough,
                "ior": self.ior, "emit": self.emit.encode(), "power": self.power,
                "absorption": self.absorption.encode(), "smoothness": self.smoothness,
                "reflectivity": self.reflectivity, "tint": self.tint.encode(),
                "absorption_scale": self.absorption_scale}


@dataclass
class Object:  # scene.go:76-84
    id: str = ""
    type: str = ""
    position: Vec3 = field(default_factory=Vec3)
    size: Vec3 = field(default_factory=Vec3)
    material_id: str = ""

    @classmethod
    def decode(cls, d) -> "Object":
        return cls(str(_get(d, "id", "")), str(_get(d, "type", "")), Vec3.decode(_get(d, "position", {})),
                   Vec3.decode(_get(d, "size", {})), str(_get(d, "material_id", "")))

    def encode(self) -> dict:
        return {"id": self.id, "type": self.type, "position": self.position.encode(), "size": self.size.encode(),
                "material_id": self.material_id}


@dataclass
class RenderSettings:  # scene.go:87-92
    width: int = 0
    height: int = 0
    samples_per_px: int = 0
    max_depth: int = 0

    @classmethod
    def decode(cls, d) -> "RenderSettings":
        return cls(_i(d, "width"), _i(d, "height"), _i(d, "samples_per_px"), _i(d, "max_depth"))

    def encode(self) -> dict:
        return {"width": self.width, "height": self.height, "samples_per_px": self.samples_per_px,
                "max_depth": self.max_depth}


@dataclass
class Fog:  # scene.go:96-131 -- carried for round-tripping; the CPU engine ignores it
    density: float = 0.0
    color: Color = field(default_factory=Color)
    scatter: float = 0.0
    sigma_s: float = 0.0
    sigma_a: float = 0.0
    g: float = 0.0
    hetero_strength: float = 0.0
    noise_scale: float = 0.0
    noise_octaves: int = 0
    affect_sky: bool = False
    gpu_volumetric: bool = False

    @classmethod
    def decode(cls, d) -> "Fog":
        return cls(_f(d, "density"), Color.decode(_get(d, "color", {})), _f(d, "scatter"), _f(d, "sigma_s"),
                   _f(d, "sigma_a"), _f(d, "g"), _f(d, "hetero_strength"), _f(d, "noise_scale"),
                   _i(d, "noise_octaves"), bool(_get(d, "affect_sky", False)), bool(_get(d, "gpu_volumetric", False)))

    def encode(self) -> dict:
        return {"density": self.density, "color": self.color.encode(), "scatter": self.scatter,
                "sigma_s": self.sigma_s, "sigma_a": self.sigma_a, "g": self.g,
                "hetero_strength": self.hetero_strength, "noise_scale": self.noise_scale,
                "noise_octaves": self.noise_octaves, "affect_sky": self.affect_sky,
                "gpu_volumetric": self.gpu_volumetric}


@dataclass
class Sky:  # scene.go:135-140
    type: str = ""
    color: Color = field(default_factory=Color)
    horizon: Color = field(default_factory=Color)
    zenith: Color = field(default_factory=Color)

    @classmethod
    def decode(cls, d) -> "Sky":
        return cls(str(_get(d, "type", "")), Color.decode(_get(d, "color", {})),
                   Color.decode(_get(d, "horizon", {})), Color.decode(_get(d, "zenith", {})))

    def encode(self) -> dict:
        return {"type": self.type, "color": self.color.encode(), "horizon": self.horizon.encode(),
                "zenith": self.zenith.encode()}


@dataclass
class Scene:  # scene.go:143-158
    name: str = ""
    camera: Camera = field(default_factory=Camera)
    objects: List[Object] = field(default_factory=list)
    materials: List[Material] = field(default_factory=list)
    settings: RenderSettings = field(default_factory=RenderSettings)
    background: Color = field(default_factory=Color)
    sky: Optional[Sky] = None
    fog: Optional[Fog] = None

    @classmethod
    def decode(cls, doc: dict) -> "Scene":
        if not isinstance(doc, dict):
            raise ValueError("decode scene: top-level JSON value is not an object")
        sky = _get(doc, "sky", None)
        fog = _get(doc, "fog", None)
        return cls(str(_get(doc, "name", "")), Camera.decode(_get(doc, "camera", {})),
                   [Object.decode(o) for o in (_get(doc, "objects", []) or [])],
                   [Material.decode(m) for m in (_get(doc, "materials", []) or [])],
                   RenderSettings.decode(_get(doc, "settings", {})), Color.decode(_get(doc, "background", {})),
                   Sky.decode(sky) if isinstance(sky, dict) else None,
                   Fog.decode(fog) if isinstance(fog, dict) else None)

    def encode(self) -> dict:
        d = {"name": self.name, "camera": self.camera.encode(), "objects": [o.encode() for o in self.objects],
             "materials": [m.encode() for m in self.materials], "settings": self.settings.encode(),
             "background": self.background.encode(), "sky": self.sky.encode() if self.sky is not None else None}
        if self.fog is not None:  # `json:"fog,omitempty"`
            d["fog"] = self.fog.encode()
        return d


def load(path: str) -> Scene:
    """scene.Load, io.go:10-22."""
    try:
        f = open(path, "r", encoding="utf-8")
    except OSError as e:
        raise OSError("open scene: %s" % e) from e
    with f:
        try:
            doc = json.load(f)
        except json.JSONDecodeError as e:
            raise ValueError("decode scene: %s" % e) from e
    return Scene.decode(doc)


def save(path: str, sc: Scene) -> None:
    """scene.Save, io.go:25-38 (two-space indent, trailing newline)."""
    try:
        f = open(path, "w", encoding="utf-8")
    except OSError as e:
        raise OSError("create scene: %s" % e) from e
    with f:
        json.dump(sc.encode(), f, indent=2, ensure_ascii=False)
        f.write("\n")
