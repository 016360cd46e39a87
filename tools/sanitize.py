#!/usr/bin/env python3
"""ASan + UBSan builds of everything that runs on the host, and the CPU suite against them (CPU box only: the pool offers no
GPU sanitizer).

    python tools/sanitize.py build     # build/sanitize/libptcore_asan.so, build/sanitize/libpthost_asan.so, oracle/libptoracle_asan.so
    python tools/sanitize.py test      # ... and `pytest -m "not gpu"` against them (what tests/test_sanitize_cpu.py runs, opt-in)

    libptcore_asan.so   libptcore with its HOST side instrumented (-fno-gpu-sanitize: the kernels are built as always): scene
                        conversion, broad-phase records, BVH and core-twin builder, shard API, argument checks
    libpthost_asan.so   the C++ mirror of the Go host layer (JSON, scene, engine, PNG)
    libptoracle_asan.so the CPU checker (oracle/Makefile)

The three libraries are swapped in through PTCORE_LIB / PTHOST_LIB / PTORACLE_LIB with the clang ASan runtime preloaded into
the Python process."""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_trace_golang_amd.build import CSRC, HIPCC, _newer, _run  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def asan_runtime() -> str:
    return subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()


def build(force: bool = False) -> dict:
    out = os.path.join(ROOT, "build", "sanitize")
    os.makedirs(out, exist_ok=True)
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
    core = os.path.join(out, "libptcore_asan.so")
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    if force or _newer(core, srcs):
        _run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", *san, "-fno-gpu-sanitize",
              "-shared-libsan", "-shared", os.path.join(CSRC, "ptcore.hip"), "-o", core])
    hdir = os.path.join(CSRC, "host")
    host = os.path.join(out, "libpthost_asan.so")
    lib_srcs = [os.path.join(hdir, f) for f in ("json.cpp", "scene.cpp", "engine.cpp", "png.cpp", "capi.cpp")]
    if force or _newer(host, lib_srcs):
        _run([CLANG, "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-Wall", "-Wextra", *san, "-shared-libsan", "-shared",
              *lib_srcs, "-o", host, "-ldl", "-lpthread"])
    ora = os.path.join(ROOT, "oracle", "libptoracle_asan.so")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "CC=" + CLANG.replace("clang++", "clang"), "SANFLAGS=-shared-libsan",
                    "libptoracle_asan.so"], check=True, capture_output=True)
    return {"PTCORE_LIB": core, "PTHOST_LIB": host, "PTORACLE_LIB": ora, "LD_PRELOAD": asan_runtime()}


def test(extra=None) -> int:
    env = dict(os.environ)
    env.update(build())
    env["ASAN_OPTIONS"] = "detect_leaks=0:verify_asan_link_order=0"  # (CPython leaks by design; the preload order is ours)
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["PT_SANITIZE_CHILD"] = "1"
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", *(extra or [])]
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "test"
    if what == "build":
        print(build("--force" in sys.argv))
    else:
        sys.exit(test([a for a in sys.argv[2:] if a != "--force"]))
