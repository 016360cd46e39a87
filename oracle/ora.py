"""ctypes harness for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Loads oracle/libptoracle.so (built by oracle/Makefile) and feeds it scene JSON
decoded the way Go's encoding/json fills scene.Scene (internal/scene/scene.go,
internal/scene/io.go:10-22): missing keys are zero values, keys match
case-insensitively, `"sky": null` leaves Sky nil.  Nothing under
path_trace_golang_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("PTORACLE_LIB") or os.path.join(_HERE, "libptoracle.so")  # PTORACLE_LIB: a sanitizer build (oracle/Makefile)


class OraMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("_pad", C.c_int32), ("albedo", C.c_double * 3), ("rough", C.c_double),
                ("ior", C.c_double), ("emit", C.c_double * 3), ("power", C.c_double),
                ("absorption", C.c_double * 3), ("smoothness", C.c_double)]


class OraObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("position", C.c_double * 3),
                ("size", C.c_double * 3)]


class OraCamera(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("target", C.c_double * 3), ("up", C.c_double * 3),
                ("fov", C.c_double), ("aperture", C.c_double), ("focus_dist", C.c_double),
                ("aspect_ratio", C.c_double)]


class OraSky(C.Structure):
    _fields_ = [("sky_type", C.c_int32), ("_pad", C.c_int32), ("background", C.c_double * 3),
                ("color", C.c_double * 3), ("horizon", C.c_double * 3), ("zenith", C.c_double * 3)]


class OraScene(C.Structure):
    _fields_ = [("camera", OraCamera), ("sky", OraSky), ("nmaterials", C.c_int32), ("nobjects", C.c_int32),
                ("materials", C.POINTER(OraMaterial)), ("objects", C.POINTER(OraObject))]


class OraConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("seed", C.c_uint64), ("workers", C.c_int32), ("_pad", C.c_int32)]


class OraPostConfig(C.Structure):
    _fields_ = [("tonemap", C.c_int32), ("denoise", C.c_int32), ("sigma_s", C.c_double), ("sigma_r", C.c_double),
                ("smooth", C.c_int32), ("smooth_radius", C.c_int32), ("smooth_strength", C.c_double)]


class OraStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("exit_scans", C.c_uint64),
                ("draws", C.c_uint64), ("seconds", C.c_double), ("workers", C.c_int32), ("_pad", C.c_int32)]


def build(force: bool = False) -> str:
    """Compile the oracle if the .so is missing or older than its sources."""
    srcs = [os.path.join(_HERE, f) for f in ("pt_oracle.c", "pt_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "libptoracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.ora_render.restype = C.c_int
        L.ora_render.argtypes = [C.POINTER(OraScene), C.POINTER(OraConfig), C.c_void_p, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.POINTER(OraStats)]
        L.ora_render_window.restype = C.c_int
        L.ora_render_window.argtypes = [C.POINTER(OraScene), C.POINTER(OraConfig), C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(OraStats)]
        L.ora_sample.restype = None
        L.ora_sample.argtypes = [C.POINTER(OraScene), C.POINTER(OraConfig), C.c_int32, C.c_int32, C.c_int32, dp,
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        for name in ("ora_sin", "ora_cos", "ora_tan", "ora_exp"):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [C.c_double]
        for name in ("ora_pow", "ora_min", "ora_max"):
            f = getattr(L, name)
            f.restype = C.c_double
            f.argtypes = [C.c_double, C.c_double]
        L.ora_stream_init.restype = C.c_uint64
        L.ora_stream_init.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.ora_stream_next.restype = C.c_double
        L.ora_stream_next.argtypes = [C.POINTER(C.c_uint64)]
        L.ora_hit.restype = C.c_int
        L.ora_hit.argtypes = [C.c_int32, dp, dp, C.c_double, dp, dp, C.c_double, C.c_double, dp]
        L.ora_convert_material.restype = None
        L.ora_convert_material.argtypes = [C.POINTER(OraMaterial), dp]
        L.ora_camera_setup.restype = None
        L.ora_camera_setup.argtypes = [C.POINTER(OraCamera), C.c_int32, C.c_int32, dp]
        L.ora_post_process.restype = None
        L.ora_post_process.argtypes = [C.POINTER(OraPostConfig), C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_int32]
        L.ora_aces_tonemap.restype = C.c_float
        L.ora_aces_tonemap.argtypes = [C.c_float]
        L.ora_finish_pixel.restype = None
        L.ora_finish_pixel.argtypes = [dp, C.c_int32, C.POINTER(C.c_uint8)]
        _lib = L
    return _lib


# ---------------------------------------------------------------- scene JSON

_MAT_TYPES = {"lambert": 0, "metal": 1, "dielectric": 2, "emissive": 3, "mirror": 4}
_OBJ_TYPES = {"sphere": 0, "plane": 1, "box": 2, "sphere_light": 3}


def _get(d, key, default=None):
    """encoding/json matches keys case-insensitively (exact match preferred)."""
    if not isinstance(d, dict):
        return default
    if key in d:
        v = d[key]
    else:
        v = default
        for k, vv in d.items():
            if isinstance(k, str) and k.lower() == key.lower():
                v = vv
                break
    return default if v is None else v


def _num(d, key):
    return float(_get(d, key, 0.0))


def _vec(d, key, names):
    sub = _get(d, key, {})
    return [_num(sub, n) for n in names]


class Scene:
    """Owns the ctypes arrays behind an OraScene."""

    def __init__(self, doc: dict):
        self.doc = doc
        mats = _get(doc, "materials", []) or []
        objs = _get(doc, "objects", []) or []
        self._mats = (OraMaterial * max(1, len(mats)))()
        ids = {}
        for i, m in enumerate(mats):
            om = self._mats[i]
            # unknown type strings fall to convertMaterial's default branch (lambert)
            om.type = _MAT_TYPES.get(_get(m, "type", ""), 0)
            om.albedo[:] = _vec(m, "albedo", "rgb")
            om.rough = _num(m, "rough")
            om.ior = _num(m, "ior")
            om.emit[:] = _vec(m, "emit", "rgb")
            om.power = _num(m, "power")
            om.absorption[:] = _vec(m, "absorption", "rgb")
            om.smoothness = _num(m, "smoothness")
            ids[_get(m, "id", "")] = i  # map assignment: the last duplicate id wins (objects.go:227-229)
        self._objs = (OraObject * max(1, len(objs)))()
        for i, o in enumerate(objs):
            oo = self._objs[i]
            oo.type = _OBJ_TYPES.get(_get(o, "type", ""), -1)
            oo.material = ids.get(_get(o, "material_id", ""), -1)
            oo.position[:] = _vec(o, "position", "xyz")
            oo.size[:] = _vec(o, "size", "xyz")
        sc = OraScene()
        cam = _get(doc, "camera", {})
        sc.camera.position[:] = _vec(cam, "position", "xyz")
        sc.camera.target[:] = _vec(cam, "target", "xyz")
        sc.camera.up[:] = _vec(cam, "up", "xyz")
        sc.camera.fov = _num(cam, "fov")
        sc.camera.aperture = _num(cam, "aperture")
        sc.camera.focus_dist = _num(cam, "focus_dist")
        sc.camera.aspect_ratio = _num(cam, "aspect_ratio")
        sky = _get(doc, "sky", None)
        sc.sky.background[:] = _vec(doc, "background", "rgb")
        if isinstance(sky, dict):
            t = _get(sky, "type", "")
            sc.sky.sky_type = 1 if t == "gradient" else 2 if t == "solid" else 0
            sc.sky.color[:] = _vec(sky, "color", "rgb")
            sc.sky.horizon[:] = _vec(sky, "horizon", "rgb")
            sc.sky.zenith[:] = _vec(sky, "zenith", "rgb")
        else:
            sc.sky.sky_type = 0
        sc.nmaterials = len(mats)
        sc.nobjects = len(objs)
        sc.materials = C.cast(self._mats, C.POINTER(OraMaterial))
        sc.objects = C.cast(self._objs, C.POINTER(OraObject))
        self.c = sc

    @classmethod
    def load(cls, path: str) -> "Scene":
        with open(path, "r", encoding="utf-8") as f:
            return cls(json.load(f))


def render(scene: Scene, width: int, height: int, spp: int, depth: int, seed: int = 1, workers: int = 0,
           window=None, want=("rgba", "accum", "nseg", "ndraw")):
    """Returns dict(rgba uint8[H,W,4], accum f64[H,W,3], nseg u32[H,W], ndraw u32[H,W], stats)."""
    L = lib()
    cfg = OraConfig(width, height, spp, depth, seed, workers, 0)
    rgba = np.zeros((height, width, 4), np.uint8) if "rgba" in want else None
    accum = np.zeros((height, width, 3), np.float64) if "accum" in want else None
    nseg = np.zeros((height, width), np.uint32) if "nseg" in want else None
    ndraw = np.zeros((height, width), np.uint32) if "ndraw" in want else None
    st = OraStats()

    def p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    if window is None:
        L.ora_render(C.byref(scene.c), C.byref(cfg), p(rgba), width * 4, p(accum), p(nseg), p(ndraw), C.byref(st))
    else:
        x0, y0, x1, y1 = window
        L.ora_render_window(C.byref(scene.c), C.byref(cfg), x0, y0, x1, y1, p(rgba), width * 4, p(accum), p(nseg),
                            p(ndraw), C.byref(st))
    return {"rgba": rgba, "accum": accum, "nseg": nseg, "ndraw": ndraw,
            "stats": {"samples": st.samples, "segments": st.segments, "exit_scans": st.exit_scans,
                      "draws": st.draws, "seconds": st.seconds, "workers": st.workers}}


def sample(scene: Scene, width: int, height: int, spp: int, depth: int, seed: int, x: int, y: int, s: int):
    L = lib()
    cfg = OraConfig(width, height, spp, depth, seed, 1, 0)
    out = (C.c_double * 3)()
    a, b = C.c_uint32(), C.c_uint32()
    L.ora_sample(C.byref(scene.c), C.byref(cfg), x, y, s, out, C.byref(a), C.byref(b))
    return list(out), a.value, b.value


def hit(kind: int, a, b, radius, orig, dir_, tmin, tmax):
    L = lib()
    d3 = C.c_double * 3
    out = (C.c_double * 8)()
    ok = L.ora_hit(kind, d3(*a), d3(*b), radius, d3(*orig), d3(*dir_), tmin, tmax, out)
    return bool(ok), {"t": out[0], "p": [out[1], out[2], out[3]], "n": [out[4], out[5], out[6]],
                      "front": bool(out[7])}


def post_process(img, tonemap=False, denoise=False, sigma_s=1.0, sigma_r=0.15, smooth=False, smooth_radius=2,
                 smooth_strength=0.5, accum=None, spp=1):
    """In-place CPU restatement of the reference GPU backend's post passes on img (uint8 [H, W, 4])."""
    L = lib()
    cfg = OraPostConfig(int(tonemap), int(denoise), sigma_s, sigma_r, int(smooth), smooth_radius, smooth_strength)
    h, w = img.shape[0], img.shape[1]
    L.ora_post_process(C.byref(cfg), accum.ctypes.data_as(C.c_void_p) if accum is not None else None, spp,
                       img.ctypes.data_as(C.c_void_p), int(img.strides[0]), w, h)
