"""HIP backend: the plug-in that takes the place of the reference's
internal/engine/gpu package behind engine.RenderInto.

    gpu.Render(sc *scene.Scene, cfg gpu.RenderConfig, img *image.RGBA, progress func()) error
                                                   (/root/reference/internal/engine/gpu/gpu.go:2534)

`render` has that shape: it flattens the scene into the C ABI's plain structs,
renders through libptcore.so on the MI355X and fills `img` (H x W x 4 uint8, row 0 on
top) in place.  Errors raise (the Go binding returns them as `error`); there is no
CPU fallback here -- in the Go integration the fallback stays the reference's own CPU
engine (renderer.go:257-262).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from . import capi
from . import scene as scn

_MAT = {scn.MATERIAL_LAMBERT: capi.PT_MAT_LAMBERT, scn.MATERIAL_METAL: capi.PT_MAT_METAL,
        scn.MATERIAL_DIELECTRIC: capi.PT_MAT_DIELECTRIC, scn.MATERIAL_EMISSIVE: capi.PT_MAT_EMISSIVE,
        scn.MATERIAL_MIRROR: capi.PT_MAT_MIRROR}
_OBJ = {scn.OBJECT_SPHERE: capi.PT_OBJ_SPHERE, scn.OBJECT_PLANE: capi.PT_OBJ_PLANE, scn.OBJECT_BOX: capi.PT_OBJ_BOX,
        scn.OBJECT_SPHERE_LIGHT: capi.PT_OBJ_SPHERE_LIGHT}


@dataclass
class RenderConfig:  # gpu.RenderConfig, gpu.go:227-232 (+ the stream seed)
    width: int = 0
    height: int = 0
    samples_per_px: int = 0
    max_depth: int = 0
    seed: int = 1
    spp_chunk: int = 0
    flags: int = 0


class FlatScene:
    """pt_scene plus the arrays it points to (kept alive together)."""

    def __init__(self, sc: scn.Scene):
        nm, no = len(sc.materials), len(sc.objects)
        self.materials = (capi.PtMaterial * max(1, nm))()
        ids = {}
        for i, m in enumerate(sc.materials):
            pm = self.materials[i]
            pm.type = _MAT.get(m.type, capi.PT_MAT_LAMBERT)  # default branch, materials.go:51-53
            pm.albedo[:] = m.albedo.as_list()
            pm.rough = m.rough
            pm.ior = m.ior
            pm.emit[:] = m.emit.as_list()
            pm.power = m.power
            pm.absorption[:] = m.absorption.as_list()
            pm.smoothness = m.smoothness
            ids[m.id] = i  # later duplicates replace earlier ones, objects.go:227-229
        self.objects = (capi.PtObject * max(1, no))()
        for i, o in enumerate(sc.objects):
            po = self.objects[i]
            po.type = _OBJ.get(o.type, capi.PT_OBJ_UNKNOWN)
            po.material = ids.get(o.material_id, -1)
            po.position[:] = o.position.as_list()
            po.size[:] = o.size.as_list()
        s = capi.PtScene()
        cam = sc.camera
        s.camera.position[:] = cam.position.as_list()
        s.camera.target[:] = cam.target.as_list()
        s.camera.up[:] = cam.up.as_list()
        s.camera.fov = cam.fov
        s.camera.aperture = cam.aperture
        s.camera.focus_dist = cam.focus_dist
        s.camera.aspect_ratio = cam.aspect_ratio
        s.sky.background[:] = sc.background.as_list()
        if sc.sky is not None:
            s.sky.kind = (capi.PT_SKY_GRADIENT if sc.sky.type == "gradient"
                          else capi.PT_SKY_SOLID if sc.sky.type == "solid" else capi.PT_SKY_BACKGROUND)
            s.sky.color[:] = sc.sky.color.as_list()
            s.sky.horizon[:] = sc.sky.horizon.as_list()
            s.sky.zenith[:] = sc.sky.zenith.as_list()
        else:
            s.sky.kind = capi.PT_SKY_BACKGROUND
        s.num_materials = nm
        s.num_objects = no
        s.materials = C.cast(self.materials, C.POINTER(capi.PtMaterial))
        s.objects = C.cast(self.objects, C.POINTER(capi.PtObject))
        self.c = s


def pt_config(cfg: RenderConfig) -> capi.PtConfig:
    return capi.PtConfig(cfg.width, cfg.height, cfg.samples_per_px, cfg.max_depth, cfg.seed & 0xFFFFFFFFFFFFFFFF,
                         cfg.spp_chunk, cfg.flags)


_default_ctx: Optional[capi.Context] = None
_default_devices = None


def set_devices(devices) -> None:
    """Devices (HIP ordinals) used by `render`; like the reference's process-wide GL worker."""
    global _default_ctx, _default_devices
    if _default_ctx is not None:
        _default_ctx.close()
        _default_ctx = None
    _default_devices = list(devices) if devices is not None else None


def context() -> capi.Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = capi.Context(devices=_default_devices) if _default_devices else capi.Context(ndev=1)
    return _default_ctx


def _ptr(a: Optional[np.ndarray]):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def render(sc, cfg: RenderConfig, img: np.ndarray, progress: Optional[Callable[[], None]] = None,
           accum: Optional[np.ndarray] = None, nseg: Optional[np.ndarray] = None,
           ndraw: Optional[np.ndarray] = None, ctx: Optional[capi.Context] = None) -> dict:
    """Fills img (uint8 [H, W, 4], C-contiguous rows; row stride may exceed 4*W).

    With `progress`, samples are added in ~10 steps and progress() is called after each
    (the cadence of gpu.go:2209-2212, :2229) and once at the end (gpu.go:2523-2525).
    Returns the pt_stats of the frame as a dict.
    """
    L = capi.load()
    ctx = ctx or context()
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 4:
        raise ValueError("img must be uint8 [H, W, 4]")
    if img.shape[0] != cfg.height or img.shape[1] != cfg.width:
        # renderIntoCPU returns silently on a size mismatch (renderer.go:46-49)
        return {}
    if img.strides[2] != 1 or img.strides[1] != 4:
        raise ValueError("img rows must be contiguous RGBA")
    flat = sc if isinstance(sc, FlatScene) else FlatScene(sc)  # callers that render one scene repeatedly flatten it once
    pc = pt_config(cfg)
    st = capi.PtStats()
    if accum is not None and (accum.dtype != np.float64 or accum.shape != (cfg.height, cfg.width, 3)
                              or not accum.flags.c_contiguous):
        raise ValueError("accum must be contiguous float64 [H, W, 3]")
    for a in (nseg, ndraw):
        if a is not None and (a.dtype != np.uint32 or a.shape != (cfg.height, cfg.width) or not a.flags.c_contiguous):
            raise ValueError("nseg/ndraw must be contiguous uint32 [H, W]")
    stride = int(img.strides[0])
    if progress is None:
        capi.check(L.pt_render(ctx.handle, C.byref(flat.c), C.byref(pc), _ptr(img), stride, _ptr(accum), _ptr(nseg),
                               _ptr(ndraw), C.byref(st)))
        return st.as_dict()
    if nseg is not None or ndraw is not None:
        raise ValueError("per-pixel stats are only available without a progress callback")
    capi.check(L.pt_begin(ctx.handle, C.byref(flat.c), C.byref(pc)))
    try:
        step = max(1, cfg.samples_per_px // 10)
        done = C.c_int32(0)
        while done.value < cfg.samples_per_px:
            capi.check(L.pt_step(ctx.handle, step, C.byref(done)))
            capi.check(L.pt_read(ctx.handle, _ptr(img), stride, _ptr(accum)))
            progress()
        if cfg.samples_per_px <= 0:
            capi.check(L.pt_read(ctx.handle, _ptr(img), stride, _ptr(accum)))
    finally:
        rc = L.pt_end(ctx.handle, C.byref(st))
    capi.check(rc)
    progress()
    return st.as_dict()


@dataclass
class PostConfig:
    """Post-process passes of the reference's OpenGL backend (gpu.go:22-47, :2309-2520); all off by default
    because they are not part of the CPU engine's image."""
    tonemap: bool = False
    denoise: bool = False
    sigma_s: float = 1.0
    sigma_r: float = 0.15
    smooth: bool = False
    smooth_radius: int = 2
    smooth_strength: float = 0.5

    @classmethod
    def from_env(cls, environ=None) -> "PostConfig":
        """The reference's switches: PATHTRACER_GPU_DENOISE (default on there), _SIGMA_S, _SIGMA_R (gpu.go:77-95),
        PATHTRACER_GPU_SMOOTH (default off), _RADIUS, _STRENGTH (gpu.go:140-175); tone mapping always on."""
        import os

        env = os.environ if environ is None else environ
        cfg = cls(tonemap=True, denoise=True)
        v = env.get("PATHTRACER_GPU_DENOISE", "").lower()
        if v in ("0", "false", "off", "no"):
            cfg.denoise = False
        for key, attr in (("PATHTRACER_GPU_DENOISE_SIGMA_S", "sigma_s"), ("PATHTRACER_GPU_DENOISE_SIGMA_R", "sigma_r")):
            try:
                f = float(env[key])
                if f > 0:
                    setattr(cfg, attr, f)
            except (KeyError, ValueError):
                pass
        v = env.get("PATHTRACER_GPU_SMOOTH", "").lower()
        if v in ("1", "true", "on", "yes"):
            cfg.smooth = True
        try:
            cfg.smooth_radius = max(1, min(5, int(env["PATHTRACER_GPU_SMOOTH_RADIUS"])))
        except (KeyError, ValueError):
            pass
        try:
            cfg.smooth_strength = max(0.0, min(1.0, float(env["PATHTRACER_GPU_SMOOTH_STRENGTH"])))
        except (KeyError, ValueError):
            pass
        return cfg


def post_process(img: np.ndarray, post: PostConfig, accum: Optional[np.ndarray] = None, samples_per_px: int = 1,
                 ctx: Optional[capi.Context] = None) -> None:
    """Applies the selected passes to img (uint8 [H, W, 4]) in place on the GPU."""
    L = capi.load()
    ctx = ctx or context()
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 4 or img.strides[2] != 1 or img.strides[1] != 4:
        raise ValueError("img must be uint8 [H, W, 4] with contiguous RGBA rows")
    h, w = img.shape[0], img.shape[1]
    if accum is not None and (accum.dtype != np.float64 or accum.shape != (h, w, 3) or not accum.flags.c_contiguous):
        raise ValueError("accum must be contiguous float64 [H, W, 3]")
    pc = capi.PtPostConfig(int(post.tonemap), int(post.denoise), post.sigma_s, post.sigma_r, int(post.smooth),
                           post.smooth_radius, post.smooth_strength)
    capi.check(L.pt_post_process(ctx.handle, C.byref(pc), _ptr(accum), samples_per_px, _ptr(img), int(img.strides[0]), w, h))
