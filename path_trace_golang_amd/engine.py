"""Python mirror of the reference's internal/engine public surface for the hot path.

Names follow /root/reference/internal/engine: RenderConfig, Render, RenderInto
(renderer.go:17-41), Backend / SetBackend / GetBackend (backend.go:5-28),
RenderScene, RenderSettingsForMode, SavePNG (util.go:13-55).  Only the GPU branch
of RenderInto exists here, and it goes to the MI355X core (hip.render) instead of the
OpenGL package.  The CPU branch is the reference's own Go code and is not part of
this package: selecting it raises instead of silently rendering somewhere else.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from . import hip
from . import scene as scn


class Backend(enum.IntEnum):  # backend.go:5-10
    CPU = 0
    GPU = 1


BackendCPU = Backend.CPU
BackendGPU = Backend.GPU

# The reference starts on BackendCPU (backend.go:12); this package has only the GPU branch.
_current_backend = Backend.GPU


def set_backend(b) -> None:
    """SetBackend, backend.go:16-23: unknown values select the CPU backend."""
    global _current_backend
    try:
        _current_backend = Backend(int(b))
    except ValueError:
        _current_backend = Backend.CPU


def get_backend() -> Backend:
    return _current_backend


@dataclass
class RenderConfig:  # renderer.go:17-22
    width: int = 0
    height: int = 0
    samples_per_px: int = 0
    max_depth: int = 0
    seed: int = 1  # stream seed (the reference seeds from the clock, random.go:14-16)


def new_image(w: int, h: int) -> np.ndarray:
    """image.NewRGBA(image.Rect(0, 0, w, h)): zeroed [h, w, 4] uint8."""
    return np.zeros((h, w, 4), np.uint8)


def render_into(sc: scn.Scene, cfg: RenderConfig, img: np.ndarray,
                progress: Optional[Callable[[], None]] = None) -> dict:
    """RenderInto, renderer.go:34-41."""
    if get_backend() != Backend.GPU:
        raise NotImplementedError(
            "BackendCPU is the reference's Go renderer (renderIntoCPU) and is not shipped here; "
            "this package implements only the GPU branch of RenderInto")
    gcfg = hip.RenderConfig(cfg.width, cfg.height, cfg.samples_per_px, cfg.max_depth, cfg.seed)
    return hip.render(sc, gcfg, img, progress)


def render(sc: scn.Scene, cfg: RenderConfig) -> np.ndarray:
    """Render, renderer.go:25-29."""
    img = new_image(cfg.width, cfg.height)
    render_into(sc, cfg, img, None)
    return img


def render_scene(sc: scn.Scene, settings: scn.RenderSettings, seed: int = 1) -> np.ndarray:
    """RenderScene, util.go:13-22."""
    return render(sc, RenderConfig(settings.width, settings.height, settings.samples_per_px, settings.max_depth, seed))


def render_settings_for_mode(mode: str) -> scn.RenderSettings:
    """RenderSettingsForMode, util.go:25-42."""
    if mode == "final":
        return scn.RenderSettings(1920, 1080, 1000, 80)
    return scn.RenderSettings(400, 225, 20, 20)


def render_settings_for_scene(sc: scn.Scene, mode: str) -> scn.RenderSettings:
    """The editor's scene-settings override, internal/ui/app.go:60-75: the mode preset, replaced by the scene's
    width x height when both are > 0 (and only then by its samples_per_px / max_depth when > 0); a "final"
    render then takes 4x the samples and 2x the depth.  cmd/render itself ignores scene.settings (main.go:52)."""
    s = render_settings_for_mode(mode)
    st = sc.settings
    if st.width > 0 and st.height > 0:
        s.width, s.height = st.width, st.height
        if st.samples_per_px > 0:
            s.samples_per_px = st.samples_per_px
        if st.max_depth > 0:
            s.max_depth = st.max_depth
    if mode == "final":
        s.samples_per_px *= 4
        s.max_depth *= 2
    return s


def save_png(path: str, img: np.ndarray) -> None:
    """SavePNG, util.go:45-55: 8-bit RGBA PNG."""
    from PIL import Image

    try:
        Image.fromarray(np.ascontiguousarray(img), "RGBA").save(path, format="PNG")
    except OSError as e:
        raise OSError("create png: %s" % e) from e
