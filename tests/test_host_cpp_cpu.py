"""The C++ mirror of the Go host layer (csrc/host): scene JSON model, flattening, PNG, CLI flags.
CPU only; the GPU leg of the CLI is in test_host_cpp_gpu.py."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SCENE_NAMES, scene_path


@pytest.fixture(scope="module")
def host():
    from path_trace_golang_amd import build, capi

    build.build_all()
    L = C.CDLL(os.environ.get("PTHOST_LIB") or os.path.join(ROOT, "path_trace_golang_amd", "libpthost.so"))  # PTHOST_LIB: a sanitizer build
    L.pth_last_error.restype = C.c_char_p
    L.pth_scene_load.restype = C.c_void_p
    L.pth_scene_load.argtypes = [C.c_char_p]
    L.pth_scene_decode.restype = C.c_void_p
    L.pth_scene_decode.argtypes = [C.c_char_p]
    L.pth_scene_free.argtypes = [C.c_void_p]
    L.pth_scene_flat.restype = C.POINTER(capi.PtScene)
    L.pth_scene_flat.argtypes = [C.c_void_p]
    L.pth_scene_has_sky.argtypes = [C.c_void_p]
    L.pth_scene_has_fog.argtypes = [C.c_void_p]
    L.pth_scene_encode.restype = C.c_char_p
    L.pth_scene_encode.argtypes = [C.c_void_p]
    L.pth_scene_save.argtypes = [C.c_void_p, C.c_char_p]
    L.pth_settings_for_mode.argtypes = [C.c_char_p, C.POINTER(C.c_int32)]
    L.pth_settings_for_scene.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32)]
    L.pth_save_png.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.pth_render_into.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    return L


def _same_flat(a, b):
    assert a.num_materials == b.num_materials and a.num_objects == b.num_objects
    for i in range(a.num_materials):
        x, y = a.materials[i], b.materials[i]
        assert (x.type, list(x.albedo), x.rough, x.ior, list(x.emit), x.power, list(x.absorption), x.smoothness) == \
               (y.type, list(y.albedo), y.rough, y.ior, list(y.emit), y.power, list(y.absorption), y.smoothness)
    for i in range(a.num_objects):
        x, y = a.objects[i], b.objects[i]
        assert (x.type, x.material, list(x.position), list(x.size)) == (y.type, y.material, list(y.position), list(y.size))
    for f in ("position", "target", "up"):
        assert list(getattr(a.camera, f)) == list(getattr(b.camera, f))
    assert (a.camera.fov, a.camera.aperture, a.camera.focus_dist, a.camera.aspect_ratio) == \
           (b.camera.fov, b.camera.aperture, b.camera.focus_dist, b.camera.aspect_ratio)
    assert a.sky.kind == b.sky.kind
    for f in ("background", "color", "horizon", "zenith"):
        assert list(getattr(a.sky, f)) == list(getattr(b.sky, f))


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_cpp_loader_equals_python_loader(host, name):
    from path_trace_golang_amd import hip, scene

    h = host.pth_scene_load(scene_path(name).encode())
    assert h, host.pth_last_error()
    try:
        _same_flat(host.pth_scene_flat(h).contents, hip.FlatScene(scene.load(scene_path(name))).c)
        # Save -> Load round trip is lossless and matches the Python writer's document
        text = host.pth_scene_encode(h).decode()
        assert text.startswith("{\n  \"name\"") and text.endswith("}\n")
        assert json.loads(text) == scene.load(scene_path(name)).encode()
        h2 = host.pth_scene_decode(text.encode())
        _same_flat(host.pth_scene_flat(h2).contents, host.pth_scene_flat(h).contents)
        host.pth_scene_free(h2)
    finally:
        host.pth_scene_free(h)


def test_cpp_decoder_rules(host):
    doc = {"NAME": "x", "camera": {"FOV": 12.5}, "sky": None, "objects": [{"type": "cube", "material_id": "q"}],
           "materials": [{"id": "q", "type": "metal"}, {"id": "q", "type": "mirror", "rough": 1e-7}], "junk": [1, {"a": "é\n"}]}
    h = host.pth_scene_decode(json.dumps(doc).encode())
    assert h, host.pth_last_error()
    f = host.pth_scene_flat(h).contents
    assert f.camera.fov == 12.5 and host.pth_scene_has_sky(h) == 0 and host.pth_scene_has_fog(h) == 0
    assert f.objects[0].type == -1 and f.objects[0].material == 1 and f.materials[1].rough == 1e-7
    host.pth_scene_free(h)
    assert not host.pth_scene_decode(b"{ \"a\": }") and b"decode scene" in host.pth_last_error()
    assert not host.pth_scene_load(b"/nonexistent/s.json") and b"open scene" in host.pth_last_error()
    assert not host.pth_scene_decode(b"[1,2]") and b"decode scene" in host.pth_last_error()


def test_cpp_number_formatting_roundtrip(host):
    vals = [0, 1, -1, 0.1, 1.7777778, 1e-7, 123456789.125, 2.5e21, -0.9520649081431036, 1e20, 5e-324, 255.999]
    doc = {"materials": [{"id": "m", "rough": v} for v in vals]}
    h = host.pth_scene_decode(json.dumps(doc).encode())
    back = json.loads(host.pth_scene_encode(h).decode())
    assert [m["rough"] for m in back["materials"]] == [float(v) for v in vals]
    host.pth_scene_free(h)


def test_settings_presets_and_backend_switch(host):
    out = (C.c_int32 * 4)()
    host.pth_settings_for_mode(b"final", out)
    assert list(out) == [1920, 1080, 1000, 80]
    host.pth_settings_for_mode(b"whatever", out)
    assert list(out) == [400, 225, 20, 20]
    host.pth_set_backend(0)
    assert host.pth_get_backend() == 0
    host.pth_set_backend(42)
    assert host.pth_get_backend() == 0  # unknown -> CPU (backend.go:16-23)
    h = host.pth_scene_load(scene_path("example_simple").encode())
    buf = np.zeros((4, 4, 4), np.uint8)
    rc = host.pth_render_into(h, 4, 4, 1, 1, 1, buf.ctypes.data_as(C.c_void_p), 4, 4, 16, None)
    assert rc == 1 and b"BackendCPU" in host.pth_last_error()  # the CPU branch is not shipped: loud error
    host.pth_set_backend(1)
    assert host.pth_get_backend() == 1
    host.pth_scene_free(h)


def test_png_encoder(host, tmp_path):
    from PIL import Image

    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    img[..., 3] = 255
    p = str(tmp_path / "a.png")
    assert host.pth_save_png(p.encode(), img.ctypes.data_as(C.c_void_p), 53, 37, 53 * 4) == 0
    im = Image.open(p)
    assert im.mode == "RGB"  # opaque image.RGBA -> 8-bit truecolour, like Go's image/png
    assert np.array_equal(np.array(im), img[..., :3])
    img[5, 7, 3] = 9
    assert host.pth_save_png(p.encode(), img.ctypes.data_as(C.c_void_p), 53, 37, 53 * 4) == 0
    im = Image.open(p)
    assert im.mode == "RGBA" and np.array_equal(np.array(im), img)
    big = rng.integers(0, 256, (300, 200, 4), dtype=np.uint8)  # > 64 KiB: several stored deflate blocks
    big[..., 3] = 255
    assert host.pth_save_png(p.encode(), big.ctypes.data_as(C.c_void_p), 200, 300, 800) == 0
    assert np.array_equal(np.array(Image.open(p)), big[..., :3])
    assert host.pth_save_png(str(tmp_path / "no" / "a.png").encode(), big.ctypes.data_as(C.c_void_p), 200, 300, 800) == 1
    assert b"create png" in host.pth_last_error()


def _cli(*args):
    exe = os.path.join(ROOT, "path_trace_golang_amd", "render")
    return subprocess.run([exe, *args], capture_output=True, text=True, cwd=ROOT, timeout=120)


def test_cli_flag_surface(host):
    r = _cli("-nope")
    assert r.returncode == 2 and "flag provided but not defined: -nope" in r.stderr and "Usage of render" in r.stderr
    r = _cli("-scene")
    assert r.returncode == 2 and "flag needs an argument: -scene" in r.stderr
    r = _cli("-gpu=maybe")
    assert r.returncode == 2 and "invalid boolean value" in r.stderr
    r = _cli("-width", "x")
    assert r.returncode == 2 and "invalid value" in r.stderr
    r = _cli("-h")
    assert r.returncode == 0 and "-headless" in r.stderr and "-mode string" in r.stderr
    # defaults of cmd/render/main.go:17-21 are echoed by the flags log line
    r = _cli("--headless=true", "-scene=/nonexistent.json", "-gpu")
    assert r.returncode == 1 and "pathtracer: starting main()" in r.stderr
    assert "flags: scene=/nonexistent.json mode=preview headless=true out=output.png" in r.stderr
    assert "headless render error: load scene: open scene" in r.stderr
    r = _cli()  # no -headless: the reference would open the UI
    assert r.returncode == 1 and "ui error" in r.stderr


def test_json_decoder_survives_mutated_input(host):
    # 3000 mutations of the scene files (truncation, byte flips, deep nesting, huge numbers): the C++ decoder must
    # return a scene or an error, never crash; whatever it accepts must flatten and re-encode
    import ctypes as C
    import random

    lib = host
    rnd = random.Random(4242)
    texts = [open(scene_path(n), "rb").read() for n in SCENE_NAMES]
    extra = [b"", b"{", b"[" * 5000, b"{\"objects\":" + b"[" * 3000, b"{\"camera\":{\"fov\":1e999}}", b"{\"objects\":[{\"type\":7}]}",
             b"\"\\ud800\"", b"{\"materials\":[{\"id\":\"" + b"x" * 100000 + b"\"}]}", b"nul", b"{\"a\":1,}", b"\xff\xfe{}"]
    accepted = 0
    for i in range(3000):
        if i < len(extra):
            t = extra[i]
        else:
            t = bytearray(rnd.choice(texts))
            for _ in range(rnd.randint(1, 6)):
                op = rnd.random()
                pos = rnd.randrange(len(t)) if t else 0
                if op < 0.3 and t:
                    t[pos] = rnd.randrange(256)
                elif op < 0.5:
                    del t[pos:pos + rnd.randint(1, 40)]
                elif op < 0.7:
                    t[pos:pos] = rnd.choice([b"{", b"}", b"[", b"]", b",", b":", b"\"", b"-", b"1e400", b"\\u12", b"null", b"\x00"])
                elif op < 0.8:
                    t = t[:pos]
                else:
                    t[pos:pos] = t[max(0, pos - 30):pos]
            t = bytes(t)
        h = lib.pth_scene_decode(t.replace(b"\x00", b" "))
        if h:
            accepted += 1
            flat = lib.pth_scene_flat(h)
            assert flat
            enc = lib.pth_scene_encode(h)
            assert enc is not None
            lib.pth_scene_free(h)
        else:
            assert lib.pth_last_error()
    assert accepted > 100  # many mutations only touch numbers or unknown keys


BAD_TYPES = ['{"objects":[{"type":7}]}', '{"camera":{"fov":"x"}}', '{"camera":{"fov":1e999}}', '{"objects":{}}',
             '{"materials":[{"id":5}]}', '{"camera":{"position":[1,2,3]}}', '{"settings":{"width":1.0}}',
             '{"settings":{"width":1e3}}', '{"settings":{"width":true}}', '{"camera":{"fov":true}}', '{"sky":5}',
             '{"fog":{"affect_sky":1}}', '{"objects":[3]}', '{"name":[]}', '{"background":"red"}']
GOOD_TYPES = ['{"sky":null,"objects":null,"camera":null,"name":null,"settings":null}', '{"objects":[null]}',
              '{"junk":1e999,"camera":{"FOV":2,"extra":[1,"a"]}}', '{"settings":{"width":-3}}', '{"camera":{"fov":1e-999}}']


def test_type_mismatches_fail_the_load_like_json_unmarshal(host):
    # encoding/json reports a value of the wrong JSON type (or a float literal out of range) as an error and
    # scene.Load fails (io.go:17-19); null and unknown keys are fine.  Same in the C++ and the Python decoder.
    from path_trace_golang_amd import scene

    for t in BAD_TYPES:
        assert not host.pth_scene_decode(t.encode()), t
        assert b"decode scene: json:" in host.pth_last_error(), (t, host.pth_last_error())
        with pytest.raises(ValueError, match="decode scene: json:"):
            scene.Scene.decode(json.loads(t))
    for t in GOOD_TYPES:
        h = host.pth_scene_decode(t.encode())
        assert h, (t, host.pth_last_error())
        host.pth_scene_free(h)
        scene.Scene.decode(json.loads(t))


def test_save_text_is_encoding_json_text_in_both_mirrors(host):
    # scene.Save (io.go:25-38) = json.Encoder with SetIndent("", "  "): ES6-style floats (exponent form below 1e-6 and
    # from 1e21, "e-7" not "e-07", integers without ".0", "-0"), HTML escaping, nil slices as null.  The C++ and the
    # Python writer must produce the same bytes; known spellings are checked literally.
    from path_trace_golang_amd import scene

    vals = {0.0: "0", 1.0: "1", 0.1: "0.1", 1e-6: "0.000001", 1e-7: "1e-7", 9.5e-10: "9.5e-10", 1.5e-12: "1.5e-12",
            1e20: "100000000000000000000", 1e21: "1e+21", 2.5e21: "2.5e+21", 1e100: "1e+100", 5e-324: "5e-324",
            123456789.125: "123456789.125", -0.9520649081431036: "-0.9520649081431036",
            1.7976931348623157e308: "1.7976931348623157e+308"}
    name = "a<b>&c" + chr(0x2028) + chr(8) + chr(12) + chr(1) + chr(0xe9) + '"' + chr(92)
    doc = {"name": name, "objects": [], "materials": [{"id": "m%d" % i, "rough": v} for i, v in enumerate(vals)]}
    text = json.dumps(doc)
    h = host.pth_scene_decode(text.encode())
    assert h, host.pth_last_error()
    cpp = host.pth_scene_encode(h).decode()
    host.pth_scene_free(h)
    py = scene.dumps(scene.Scene.decode(json.loads(text)))
    assert cpp == py
    for v, want in vals.items():
        assert '"rough": %s,' % want in cpp, (v, want)
    want_name = '"name": "a' + chr(92) + 'u003cb' + chr(92) + 'u003e' + chr(92) + 'u0026c' + chr(92) + 'u2028' + chr(92) + 'b' + \
                chr(92) + 'f' + chr(92) + 'u0001' + chr(0xe9) + chr(92) + '"' + chr(92) + chr(92) + '",'
    assert want_name in cpp
    assert '"objects": [],' in cpp and cpp.endswith("}\n")
    h = host.pth_scene_decode(b'{"camera":{"fov":-0.0}}')
    enc = host.pth_scene_encode(h).decode()
    assert '"objects": null,\n  "materials": null,' in enc and '"fov": -0,' in enc
    host.pth_scene_free(h)
    for n in SCENE_NAMES:  # the reference's own files through both writers
        raw = open(scene_path(n)).read()
        h = host.pth_scene_decode(raw.encode())
        assert host.pth_scene_encode(h).decode() == scene.dumps(scene.Scene.decode(json.loads(raw)))
        host.pth_scene_free(h)


def test_scene_settings_override_follows_the_editor_rule(host, tmp_path):
    """internal/ui/app.go:60-75 in the C++ and the Python mirror: the scene's width x height replace the preset only when
    both are > 0, samples / depth only inside that branch and only when > 0; "final" then takes 4x samples, 2x depth."""
    import json

    from path_trace_golang_amd import engine, scene

    # (settings block in the file) -> expected (preview, final)
    cases = [
        ("example_simple", (400, 225, 20, 10), (400, 225, 80, 20)),          # 400x225, 20 spp, depth 10 in the file
        ("gpu_showcase", (800, 450, 1, 12), (800, 450, 4, 24)),               # 800x450, 1 spp, depth 12
        ("metal_glass_room", (400, 225, 20, 20), (1920, 1080, 4000, 160)),    # all zero: the preset stays (final: x4, x2)
    ]
    out = (C.c_int32 * 4)()
    for name, prev, fin in cases:
        h = host.pth_scene_load(scene_path(name).encode())
        sc = scene.load(scene_path(name))
        for mode, want in (("preview", prev), ("final", fin)):
            host.pth_settings_for_scene(h, mode.encode(), out)
            assert tuple(out) == want, (name, mode)
            s = engine.render_settings_for_scene(sc, mode)
            assert (s.width, s.height, s.samples_per_px, s.max_depth) == want, (name, mode)
        host.pth_scene_free(h)
    # width without height: nothing of the block applies, not even samples / depth
    doc = json.load(open(scene_path("example_simple")))
    doc["settings"] = {"width": 640, "height": 0, "samples_per_px": 7, "max_depth": 3}
    p = tmp_path / "half.json"
    p.write_text(json.dumps(doc))
    h = host.pth_scene_load(str(p).encode())
    host.pth_settings_for_scene(h, b"preview", out)
    assert tuple(out) == (400, 225, 20, 20)
    host.pth_scene_free(h)
    s = engine.render_settings_for_scene(scene.load(str(p)), "preview")
    assert (s.width, s.height, s.samples_per_px, s.max_depth) == (400, 225, 20, 20)
    # both sizes set, samples zero: size and depth from the file, samples from the preset
    doc["settings"] = {"width": 64, "height": 36, "samples_per_px": 0, "max_depth": 5}
    p.write_text(json.dumps(doc))
    h = host.pth_scene_load(str(p).encode())
    host.pth_settings_for_scene(h, b"final", out)
    assert tuple(out) == (64, 36, 4000, 10)
    host.pth_scene_free(h)


def test_bench_refuses_a_world_size_that_does_not_match_gpus():
    """bench.py --gpus N must be started with exactly N ranks: any mismatch is an error, before any GPU work."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "--nproc-per-node 2" in r.stderr
