"""Build recipes for the native parts (hipcc cross-compiles gfx950 without a GPU).

    python -m path_trace_golang_amd.build            # everything
    python -m path_trace_golang_amd.build core       # libptcore.so only
(ASan + UBSan builds of the host-side code and the CPU suite against them: tools/sanitize.py.)

Artifacts are written in-tree (git-ignored, but shipped to the GPU box by gpurun):
    path_trace_golang_amd/libptcore.so   C ABI + gfx950 kernels        (csrc/ptcore.hip)
    path_trace_golang_amd/libpthost.so   C++ mirror of the Go host layer (csrc/host/*.cpp)
    path_trace_golang_amd/render         CLI twin of cmd/render
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
ROOT = os.path.dirname(PKG)

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXX = os.environ.get("CXX") or shutil.which("g++") or "g++"

# -ffp-contract=off: the reference's Go code never fuses multiply-add on amd64, and the
# kernels must round exactly like it.  No fast-math anywhere.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
             "-Wall", "-Wextra"]
CXX_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-Wall", "-Wextra"]


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))


def build_core(force: bool = False) -> str:
    out = os.path.join(PKG, "libptcore.so")
    # every header and kernel source next to ptcore.hip (pt_kernels.h, pt_device.h, pt_math.h, pt_bvh.h, ...)
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "ptcore.h"))
    if force or _newer(out, srcs):
        _run([HIPCC, *HIP_FLAGS, "-shared", os.path.join(CSRC, "ptcore.hip"), "-o", out])
    return out


def build_host(force: bool = False) -> list[str]:
    hdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hdir):
        return []
    lib = os.path.join(PKG, "libpthost.so")
    exe = os.path.join(PKG, "render")
    lib_srcs = [os.path.join(hdir, f) for f in ("json.cpp", "scene.cpp", "engine.cpp", "png.cpp", "capi.cpp")]
    hdrs = [os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".hpp")]
    hdrs.append(os.path.join(ROOT, "include", "ptcore.h"))
    if force or _newer(lib, lib_srcs + hdrs):
        _run([CXX, *CXX_FLAGS, "-shared", *lib_srcs, "-o", lib, "-ldl", "-lpthread"])
    main = os.path.join(hdir, "render_main.cpp")
    if force or _newer(exe, lib_srcs + hdrs + [main]):
        _run([CXX, *CXX_FLAGS, main, *lib_srcs, "-o", exe, "-ldl", "-lpthread"])
    return [lib, exe]


def build_tools(force: bool = False) -> list[str]:
    """Stand-alone gfx950 measurement programs under tools/ (not part of the library)."""
    src = os.path.join(ROOT, "tools", "valu_floor.hip")
    out = os.path.join(ROOT, "tools", "valu_floor")
    if not os.path.exists(src):
        return []
    if force or _newer(out, [src]):
        _run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", src, "-o", out])
    return [out]


def build_all(force: bool = False) -> list[str]:
    return [build_core(force), *build_host(force), *build_tools(force)]


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    force = "--force" in sys.argv
    outs = build_core(force) if what == "core" else build_all(force)
    print(outs)
