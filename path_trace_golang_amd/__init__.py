"""MI355X-native path-tracing core: drop-in for the render loop of
MarkJulian19/path_trace_golang's internal/engine.

    capi    ctypes binding of include/ptcore.h (libptcore.so: HIP kernels + C ABI)
    hip     the backend plug-in with the shape of the reference's gpu.Render
    scene   mirror of internal/scene (model + JSON Load/Save)
    engine  mirror of internal/engine's public surface (RenderInto, Backend, ...)
    build   hipcc / g++ build recipes
"""
__all__ = ["capi", "hip", "scene", "engine", "build"]
