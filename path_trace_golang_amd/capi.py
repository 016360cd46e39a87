"""ctypes binding of include/ptcore.h (libptcore.so).

This is the only way Python reaches the renderer: there is no Python or CPU
rendering path in this package.  If the library has not been built, or no HIP
device is present, the calls raise -- they never fall back.

A process that also uses torch must `import torch` BEFORE the first capi.load(): torch bundles
its own libamdhip64.so.7, and loading it first lets libptcore.so bind to that same copy (one HIP
runtime, so torch streams and device pointers are valid inside libptcore).  The other order
loads two HIP runtimes and torch then reports no GPUs.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTCORE_LIB") or os.path.join(PKG, "libptcore.so")  # PTCORE_LIB: another build of the same ABI (A/B runs)

PT_ABI_VERSION = 3
PT_OK, PT_ERR_INVALID, PT_ERR_NO_DEVICE, PT_ERR_HIP, PT_ERR_NOMEM, PT_ERR_STATE = range(6)
PT_MAT_LAMBERT, PT_MAT_METAL, PT_MAT_DIELECTRIC, PT_MAT_EMISSIVE, PT_MAT_MIRROR = range(5)
PT_OBJ_UNKNOWN, PT_OBJ_SPHERE, PT_OBJ_PLANE, PT_OBJ_BOX, PT_OBJ_SPHERE_LIGHT = -1, 0, 1, 2, 3
PT_SKY_BACKGROUND, PT_SKY_GRADIENT, PT_SKY_SOLID = range(3)
PT_FLAG_NONE, PT_FLAG_PIXEL_STATS = 0, 1

_d3 = C.c_double * 3


class PtMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("reserved", C.c_int32), ("albedo", _d3), ("rough", C.c_double),
                ("ior", C.c_double), ("emit", _d3), ("power", C.c_double), ("absorption", _d3),
                ("smoothness", C.c_double)]


class PtObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("position", _d3), ("size", _d3)]


class PtCamera(C.Structure):
    _fields_ = [("position", _d3), ("target", _d3), ("up", _d3), ("fov", C.c_double), ("aperture", C.c_double),
                ("focus_dist", C.c_double), ("aspect_ratio", C.c_double)]


class PtSky(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("background", _d3), ("color", _d3),
                ("horizon", _d3), ("zenith", _d3)]


class PtScene(C.Structure):
    _fields_ = [("camera", PtCamera), ("sky", PtSky), ("num_materials", C.c_int32), ("num_objects", C.c_int32),
                ("materials", C.POINTER(PtMaterial)), ("objects", C.POINTER(PtObject))]


class PtConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples_per_px", C.c_int32),
                ("max_depth", C.c_int32), ("seed", C.c_uint64), ("spp_chunk", C.c_int32), ("flags", C.c_int32)]


class PtPostConfig(C.Structure):
    _fields_ = [("tonemap", C.c_int32), ("denoise", C.c_int32), ("sigma_s", C.c_double), ("sigma_r", C.c_double),
                ("smooth", C.c_int32), ("smooth_radius", C.c_int32), ("smooth_strength", C.c_double)]


class PtShard(C.Structure):
    _fields_ = [("index", C.c_int32), ("count", C.c_int32)]


class PtStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("exit_scans", C.c_uint64),
                ("draws", C.c_uint64), ("seconds", C.c_double), ("trace_ms", C.c_double),
                ("resolve_ms", C.c_double), ("device_ms", C.c_double), ("trace_launches", C.c_int32),
                ("resolve_launches", C.c_int32), ("spp_chunk", C.c_int32), ("num_devices", C.c_int32),
                ("per_device_ms", C.c_double * 8), ("raygen_ms", C.c_double), ("glass_ms", C.c_double),
                ("trace_split_ms", C.c_double), ("glass_launches", C.c_int32), ("trace_split_launches", C.c_int32),
                ("glass_events", C.c_uint64), ("continuations", C.c_uint64), ("split_cont_in", C.c_uint64),
                ("split_finished", C.c_uint64), ("shader_clock_mhz", C.c_double)]

    def as_dict(self) -> dict:
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "per_device_ms"}
        d["per_device_ms"] = list(self.per_device_ms)[: max(1, self.num_devices)]
        return d


# every symbol include/ptcore.h declares: (name, restype, argtypes)
_vp = C.c_void_p
SYMBOLS = [
    ("pt_abi_version", C.c_int32, []),
    ("pt_last_error", C.c_char_p, []),
    ("pt_device_count", C.c_int32, [C.POINTER(C.c_int32)]),
    ("pt_create", C.c_int32, [C.POINTER(C.c_int32), C.c_int32, C.POINTER(_vp)]),
    ("pt_destroy", None, [_vp]),
    ("pt_render", C.c_int32, [_vp, C.POINTER(PtScene), C.POINTER(PtConfig), _vp, C.c_int32, _vp, _vp, _vp,
                               C.POINTER(PtStats)]),
    ("pt_begin", C.c_int32, [_vp, C.POINTER(PtScene), C.POINTER(PtConfig)]),
    ("pt_step", C.c_int32, [_vp, C.c_int32, C.POINTER(C.c_int32)]),
    ("pt_read", C.c_int32, [_vp, _vp, C.c_int32, _vp]),
    ("pt_end", C.c_int32, [_vp, C.POINTER(PtStats)]),
    ("pt_shard_tiles", C.c_int32, [C.c_int32, C.c_int32, C.POINTER(PtShard), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("pt_render_tiles_device", C.c_int32, [_vp, C.POINTER(PtScene), C.POINTER(PtConfig), C.POINTER(PtShard), _vp, _vp,
                                            _vp, C.POINTER(PtStats)]),
    ("pt_untile_device", C.c_int32, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp,
                                      _vp]),
    ("pt_post_process", C.c_int32, [_vp, C.POINTER(PtPostConfig), _vp, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32]),
    ("pt_debug_profile", C.c_int32, [_vp, C.POINTER(C.c_uint64), C.c_int32]),
    ("pt_debug_scan_mismatches", C.c_int64, [_vp]),
    ("pt_debug_div_selftest", C.c_int64, [_vp, C.c_int32, C.c_uint64]),
    ("pt_debug_bvh_check", C.c_int32, [C.POINTER(PtScene), C.POINTER(C.c_int32)]),
    ("pt_debug_gather_mode", C.c_int32, [_vp]),
]


class PtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("ptcore error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Loads libptcore.so; raises if it was not built (run path_trace_golang_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -m path_trace_golang_amd.build` "
                              "(there is no fallback renderer)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)  # AttributeError if the symbol is not exported
            f.restype = res
            f.argtypes = args
        if L.pt_abi_version() != PT_ABI_VERSION:
            raise ImportError("libptcore.so ABI %d != binding %d" % (L.pt_abi_version(), PT_ABI_VERSION))
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != PT_OK:
        raise PtError(rc, (load().pt_last_error() or b"").decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int32(0)
    rc = load().pt_device_count(C.byref(n))
    return n.value if rc == PT_OK else 0


class Context:
    """pt_ctx owner."""

    def __init__(self, devices=None, ndev: int = 1):
        L = load()
        self._h = _vp()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*devices)
            check(L.pt_create(arr, len(devices), C.byref(self._h)))
        else:
            check(L.pt_create(None, ndev, C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            load().pt_destroy(self._h)
            self._h = _vp()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
