"""Scene model and JSON I/O: Python mirror of the reference's internal/scene package.

Types and field names follow /root/reference/internal/scene/scene.go:9-158; `load`
and `save` follow io.go:10-38.  Decoding behaves like Go's encoding/json into those
structs: absent keys and nulls leave zero values, keys match case-insensitively, `"sky": null`
(or no "sky") leaves Scene.sky = None, unknown keys are ignored, and a value of the wrong JSON
type for its field (or a number that overflows float64) fails the load like json.Unmarshal's
UnmarshalTypeError does.
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


def _kind(v) -> str:
    return ("bool" if isinstance(v, bool) else "number" if isinstance(v, (int, float)) else "string" if isinstance(v, str)
            else "array" if isinstance(v, list) else "object" if isinstance(v, dict) else type(v).__name__)


def _mismatch(v, key: str, want: str):
    # encoding/json reports the first type mismatch as an UnmarshalTypeError and scene.Load fails with it (io.go:17-19)
    return ValueError("decode scene: json: cannot unmarshal %s into field %s of type %s" % (_kind(v), key, want))


def _get(d, key, default=None):
    """Value of `key` (exact match first, then case-insensitively, like encoding/json); null counts as absent."""
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
    v = _get(d, key, 0.0)
    if isinstance(v, bool) or not isinstance(v, (int, float)):
        raise _mismatch(v, key, "float64")
    v = float(v)
    if v != v or v in (float("inf"), float("-inf")):  # 1e999, or the NaN / Infinity literals Python's parser lets through
        raise ValueError("decode scene: json: number out of range for field %s of type float64" % key)
    return v


def _i(d, key) -> int:
    v = _get(d, key, 0)
    if isinstance(v, bool) or not isinstance(v, int):  # 1.0 and 1e3 are not integer literals for encoding/json either
        raise _mismatch(v, key, "int")
    if not -(1 << 63) <= v < (1 << 63):
        raise ValueError("decode scene: json: number out of range for field %s of type int" % key)
    return v


def _s(d, key) -> str:
    v = _get(d, key, "")
    if not isinstance(v, str):
        raise _mismatch(v, key, "string")
    return v


def _b(d, key) -> bool:
    v = _get(d, key, False)
    if not isinstance(v, bool):
        raise _mismatch(v, key, "bool")
    return v


def _o(d, key) -> dict:
    """Nested struct: absent or null gives the zero value (an empty object here)."""
    v = _get(d, key, None)
    if v is None:
        return {}
    if not isinstance(v, dict):
        raise _mismatch(v, key, "struct")
    return v


def _a(d, key) -> list:
    v = _get(d, key, None)
    if v is None:
        return []
    if not isinstance(v, list):
        raise _mismatch(v, key, "slice")
    return v


def _elem(v, key: str) -> dict:
    if v is None:  # a null element leaves the zero struct
        return {}
    if not isinstance(v, dict):
        raise _mismatch(v, key, "struct")
    return v


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
        return cls(Vec3.decode(_o(d, "position")), Vec3.decode(_o(d, "target")), Vec3.decode(_o(d, "up")), _f(d, "fov"),
                   _f(d, "aperture"), _f(d, "focus_dist"), _f(d, "aspect_ratio"))

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
        return cls(_s(d, "id"), _s(d, "type"), Color.decode(_o(d, "albedo")), _f(d, "rough"), _f(d, "ior"),
                   Color.decode(_o(d, "emit")), _f(d, "power"), Color.decode(_o(d, "absorption")), _f(d, "smoothness"),
                   _f(d, "reflectivity"), Color.decode(_o(d, "tint")), _f(d, "absorption_scale"))

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
        return cls(_s(d, "id"), _s(d, "type"), Vec3.decode(_o(d, "position")), Vec3.decode(_o(d, "size")),
                   _s(d, "material_id"))

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
        return cls(_f(d, "density"), Color.decode(_o(d, "color")), _f(d, "scatter"), _f(d, "sigma_s"), _f(d, "sigma_a"),
                   _f(d, "g"), _f(d, "hetero_strength"), _f(d, "noise_scale"), _i(d, "noise_octaves"), _b(d, "affect_sky"),
                   _b(d, "gpu_volumetric"))

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
        return cls(_s(d, "type"), Color.decode(_o(d, "color")), Color.decode(_o(d, "horizon")), Color.decode(_o(d, "zenith")))

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
    # Go distinguishes a nil slice (no key, or null: Save writes null) from an empty one (Save writes [])
    objects_nil: bool = field(default=True, compare=False)
    materials_nil: bool = field(default=True, compare=False)

    @classmethod
    def decode(cls, doc: dict) -> "Scene":
        if not isinstance(doc, dict):
            raise ValueError("decode scene: top-level JSON value is not an object")
        sky = _get(doc, "sky", None)  # *Sky / *Fog: null or absent leaves the nil pointer
        fog = _get(doc, "fog", None)
        return cls(_s(doc, "name"), Camera.decode(_o(doc, "camera")),
                   [Object.decode(_elem(o, "objects")) for o in _a(doc, "objects")],
                   [Material.decode(_elem(m, "materials")) for m in _a(doc, "materials")],
                   RenderSettings.decode(_o(doc, "settings")), Color.decode(_o(doc, "background")),
                   Sky.decode(_o(doc, "sky")) if sky is not None else None,
                   Fog.decode(_o(doc, "fog")) if fog is not None else None,
                   _get(doc, "objects", None) is None, _get(doc, "materials", None) is None)

    def encode(self) -> dict:
        d = {"name": self.name, "camera": self.camera.encode(),
             "objects": None if (self.objects_nil and not self.objects) else [o.encode() for o in self.objects],
             "materials": None if (self.materials_nil and not self.materials) else [m.encode() for m in self.materials],
             "settings": self.settings.encode(),
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
        text = f.read()
    try:
        # json.Decoder.Decode reads ONE value and leaves whatever follows it unread (io.go:17)
        doc, _ = json.JSONDecoder().raw_decode(text.lstrip(" \t\r\n"))
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
        f.write(dumps(sc))


def _go_float(f: float) -> str:
    """encoding/json's float64 formatting (encode.go floatEncoder): shortest digits that round-trip, 'f' layout
    unless |f| < 1e-6 or |f| >= 1e21, then 'e' layout with a one-digit negative exponent cleaned up (e-07 -> e-7)."""
    import math
    from decimal import Decimal

    if math.isnan(f) or math.isinf(f):
        raise ValueError("encode scene: json: unsupported value: %r" % f)
    if f == 0:
        return "-0" if math.copysign(1.0, f) < 0 else "0"
    a = abs(f)
    d = Decimal(repr(float(f)))  # repr gives the shortest round-trip digits, like strconv's -1 precision
    if a < 1e-6 or a >= 1e21:
        sign, digits, exp = d.as_tuple()
        digs = "".join(map(str, digits)).rstrip("0") or "0"
        e10 = exp + len(digits) - 1
        mant = digs[0] + ("." + digs[1:] if len(digs) > 1 else "")
        es = "%s%02d" % ("-" if e10 < 0 else "+", abs(e10))
        if e10 < 0 and abs(e10) < 10:
            es = "-%d" % abs(e10)
        return ("-" if sign else "") + mant + "e" + es
    s = format(d, "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    return s


def _go_string(v: str) -> str:
    text = json.dumps(v, ensure_ascii=False)  # ", \\, \b, \f, \n, \r, \t and \u00xx like encoding/json
    for ch, esc in (("<", "\\u003c"), (">", "\\u003e"), ("&", "\\u0026"), ("\u2028", "\\u2028"), ("\u2029", "\\u2029"),
                    ("\x7f", "\x7f")):
        text = text.replace(ch, esc)
    return text


def _go_value(v, level: int) -> str:
    pad, pad_in = "  " * level, "  " * (level + 1)
    if v is None:
        return "null"
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, int):
        return str(v)
    if isinstance(v, float):
        return _go_float(v)
    if isinstance(v, str):
        return _go_string(v)
    if isinstance(v, list):
        if not v:
            return "[]"
        return "[\n" + ",\n".join(pad_in + _go_value(e, level + 1) for e in v) + "\n" + pad + "]"
    if not v:
        return "{}"
    return "{\n" + ",\n".join(pad_in + _go_string(k) + ": " + _go_value(e, level + 1) for k, e in v.items()) + "\n" + pad + "}"


def dumps(sc: Scene) -> str:
    """The text scene.Save writes (io.go:25-38): json.Encoder with SetIndent("", "  "), its float and string
    formatting (HTML escaping on), a trailing newline."""
    return _go_value(sc.encode(), 0) + "\n"
