"""C-ABI surface on CPU: libptcore.so loads, exports every symbol include/ptcore.h declares, and
fails loudly (no fallback) when there is no HIP device.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptcore.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from path_trace_golang_amd import build, capi

    build.build_core()
    lib = capi.load()
    declared = _declared_symbols()
    assert "pt_render" in declared and "pt_render_tiles_device" in declared and len(declared) >= 12
    bound = {name for name, _, _ in capi.SYMBOLS}
    for sym in declared:
        assert hasattr(lib, sym), sym + " is declared in ptcore.h but not exported by libptcore.so"
        assert sym in bound, sym + " has no ctypes prototype in capi.SYMBOLS"
    assert lib.pt_abi_version() == capi.PT_ABI_VERSION


def test_struct_layouts_match_header():
    from path_trace_golang_amd import capi

    # sizes implied by the C declarations (LP64, natural alignment)
    assert C.sizeof(capi.PtMaterial) == 8 + 8 * (3 + 1 + 1 + 3 + 1 + 3 + 1)
    assert C.sizeof(capi.PtObject) == 8 + 48
    assert C.sizeof(capi.PtCamera) == 13 * 8
    assert C.sizeof(capi.PtSky) == 8 + 12 * 8
    assert C.sizeof(capi.PtScene) == 13 * 8 + 104 + 8 + 16
    assert C.sizeof(capi.PtConfig) == 32
    assert C.sizeof(capi.PtStats) == 4 * 8 + 4 * 8 + 4 * 4 + 8 * 8 + 8 + 2 * 8 + 2 * 4 + 4 * 8 + 8  # ABI 2: + glass_ms, trace_split_ms, two launch counts, four path-queue counters; ABI 3: + shader_clock_mhz


def test_no_device_is_an_error_not_a_fallback():
    from path_trace_golang_amd import capi

    if capi.device_count() > 0:
        pytest.skip("a HIP device is present; the no-device contract is checked in the build container")
    lib = capi.load()
    n = C.c_int32(-1)
    assert lib.pt_device_count(C.byref(n)) == capi.PT_ERR_NO_DEVICE and n.value == 0
    h = C.c_void_p()
    assert lib.pt_create(None, 1, C.byref(h)) == capi.PT_ERR_NO_DEVICE and not h.value
    assert b"no HIP device" in lib.pt_last_error()
    with pytest.raises(capi.PtError):
        capi.Context(ndev=1)


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under path_trace_golang_amd/ or include/ may reference it
    bad = []
    for base in ("path_trace_golang_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")):
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"\boracle\b|libptoracle|pt_oracle|\bora_", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_shard_tile_counts_agree_with_python_tiling():
    from path_trace_golang_amd import capi, tiling

    lib = capi.load()
    for (w, h, n) in [(1920, 1080, 8), (80, 70, 2), (33, 31, 3), (3840, 2160, 8), (400, 225, 1)]:
        for k in range(n):
            sh = capi.PtShard(k, n)
            a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
            assert lib.pt_shard_tiles(w, h, C.byref(sh), C.byref(a), C.byref(b), C.byref(c)) == 0
            assert a.value == len(tiling.shard_tiles(w, h, k, n)) and (b.value, c.value) == tiling.tile_grid(w, h)
    bad = capi.PtShard(3, 2)
    assert lib.pt_shard_tiles(64, 64, C.byref(bad), None, None, None) == capi.PT_ERR_INVALID


def test_bvh_builder_invariants_on_cpu():
    # host-only: the 4-wide hierarchy used beyond the candidate bitmasks reaches every finite object and every
    # node exactly once, keeps each object inside its slot's box, nests child boxes in their parent's, uses at
    # most four slots per node and needs a traversal stack within the kernel's limit
    import ctypes as C

    from conftest import SCENE_NAMES, scene_path
    from path_trace_golang_amd import capi, hip, scene, synth

    lib = capi.load()
    cases = [synth.make_scene(n, 3) for n in (1, 2, 5, 64, 65, 500, 5000)] + [scene.load(scene_path(s)) for s in SCENE_NAMES]
    # geometrically spaced spheres make SAH peel off one object per level: the builder must bound the depth
    cases.append(scene.Scene.decode({
        "camera": {"position": {"x": 0, "y": 1, "z": 6}, "target": {"x": 0, "y": 1, "z": 0}, "up": {"x": 0, "y": 1, "z": 0}, "fov": 50},
        "materials": [{"id": "d", "type": "lambert", "albedo": {"r": 0.5, "g": 0.5, "b": 0.5}}],
        "objects": [{"type": "sphere", "position": {"x": 1e-100 * 2.0 ** k, "y": 0, "z": 0}, "size": {"x": 1e-103 * 2.0 ** k, "y": 0, "z": 0},
                     "material_id": "d"} for k in range(900)]}))
    for sc in cases:
        flat = hip.FlatScene(sc)
        out = (C.c_int32 * 8)()
        assert lib.pt_debug_bvh_check(C.byref(flat.c), out) == 0
        nodes, objs, depth, slots, bad, outside, nested, planes = list(out)
        finite = sum(1 for o in sc.objects if o.type in ("sphere", "sphere_light", "box"))
        assert objs == finite and planes == sum(1 for o in sc.objects if o.type == "plane")
        assert bad == 0 and outside == 0 and nested == 0
        assert slots <= 4 and depth <= 64 and nodes >= 1


def test_synthetic_scene_is_reproducible_and_in_schema(tmp_path):
    from path_trace_golang_amd import scene, synth

    a, b, c = synth.make_scene(200, 5), synth.make_scene(200, 5), synth.make_scene(200, 6)
    assert a == b and a != c
    assert sum(1 for o in a.objects if o.type != "plane") == 200
    p = str(tmp_path / "synth.json")
    scene.save(p, a)
    assert scene.load(p) == a  # same JSON schema as the reference's scene files


def test_header_is_plain_c_and_matches_the_ctypes_mirror(tmp_path):
    # cgo compiles include/ptcore.h as C: build a C99 consumer with gcc (pedantic, warnings are errors), link it
    # against libptcore.so, and compare the struct sizes it sees with the ctypes mirror used by the tests
    import ctypes as C
    import subprocess

    from path_trace_golang_amd import build, capi

    lib = build.build_core()
    src = tmp_path / "consumer.c"
    src.write_text(r'''
#include <stdio.h>
#include "ptcore.h"
int main(void) {
    /* every entry point a cgo binding may name must be declared with a C prototype */
    typedef void (*fn_t)(void);
    fn_t fns[] = {(fn_t)pt_abi_version, (fn_t)pt_last_error, (fn_t)pt_device_count, (fn_t)pt_create, (fn_t)pt_destroy,
                  (fn_t)pt_render, (fn_t)pt_begin, (fn_t)pt_step, (fn_t)pt_read, (fn_t)pt_end, (fn_t)pt_shard_tiles,
                  (fn_t)pt_render_tiles_device, (fn_t)pt_untile_device, (fn_t)pt_post_process};
    pt_ctx *ctx = 0;
    int rc = pt_create(0, 1, &ctx);
    printf("%d %d %d %d %d %d %d %d %d %d\n", (int)pt_abi_version(), (int)sizeof(pt_material), (int)sizeof(pt_object),
           (int)sizeof(pt_camera), (int)sizeof(pt_sky), (int)sizeof(pt_scene), (int)sizeof(pt_config),
           (int)sizeof(pt_shard), (int)sizeof(pt_stats), (int)sizeof(pt_post_config));
    printf("%d %d\n", rc, (int)(sizeof fns / sizeof fns[0]));
    if (rc == PT_OK) pt_destroy(ctx); else printf("%s\n", pt_last_error());
    return 0;
}
''')
    exe = tmp_path / "consumer"
    libdir = os.path.dirname(lib)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe), "-L", libdir, "-lptcore", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = [int(x) for x in out[0].split()]
    want = [capi.load().pt_abi_version()] + [C.sizeof(t) for t in (capi.PtMaterial, capi.PtObject, capi.PtCamera, capi.PtSky,
                                                                   capi.PtScene, capi.PtConfig, capi.PtShard, capi.PtStats,
                                                                   capi.PtPostConfig)]
    assert got == want
    rc = int(out[1].split()[0])
    assert rc in (capi.PT_OK, capi.PT_ERR_NO_DEVICE)  # no GPU in the CPU container: an error code and a message, not an abort


def test_go_binding_names_match_the_header():
    # the cgo sources cannot be compiled here (no Go toolchain); at least every C field, constant and function
    # they name must exist in include/ptcore.h (cgo spells the C field `type` as `_type`)
    text = open(os.path.join(ROOT, "include", "ptcore.h")).read()
    structs = {}
    for m in re.finditer(r"typedef struct (?:\w+ )?\{(.*?)\} (\w+);", text, re.S):
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        structs[m.group(2)] = set(re.findall(r"(\w+)(?:\[[^\]]*\])*\s*;", body))
    names = set(re.findall(r"\b(pt_\w+|PT_\w+)\b", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    go = "".join(open(os.path.join(ROOT, "go", *p)).read() for p in (("internal", "engine", "hip", "hip.go"), ("cmd", "render", "main.go")))
    for struct, pat in (("pt_material", r"ms\[i\]\.(\w+)"), ("pt_object", r"os_\[i\]\.(\w+)"), ("pt_camera", r"cs\.camera\.(\w+)"),
                        ("pt_sky", r"cs\.sky\.(\w+)")):
        used = {x.lstrip("_") for x in re.findall(pat, go)}
        assert used and used <= structs[struct], (struct, used - structs[struct])
    cfg_fields = set(re.findall(r"(\w+):\s*C\.", re.findall(r"C\.pt_config\{(.*?)\}", go, re.S)[0]))
    assert cfg_fields and cfg_fields <= structs["pt_config"]
    used = set(re.findall(r"C\.(pt_\w+|PT_\w+)", go))
    assert used <= names, used - names
