"""Python mirrors of internal/scene and internal/engine (host logic, no GPU)."""
import json
import os

import numpy as np
import pytest

from conftest import SCENE_NAMES, scene_path


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_load_counts_and_roundtrip(tmp_path, name):
    from path_trace_golang_amd import scene

    sc = scene.load(scene_path(name))
    doc = json.load(open(scene_path(name)))
    assert len(sc.objects) == len(doc["objects"]) and len(sc.materials) == len(doc["materials"])
    assert sc.camera.aspect_ratio == 1.7777778 and sc.camera.up.as_list() == [0, 1, 0]
    out = str(tmp_path / "s.json")
    scene.save(out, sc)
    again = scene.load(out)
    assert again == sc
    text = open(out).read()
    assert text.startswith("{\n  \"name\"") and text.endswith("}\n")  # two-space indent + newline (io.go:31-33)


def test_go_json_decoding_rules(tmp_path):
    from path_trace_golang_amd import scene

    p = tmp_path / "x.json"
    p.write_text(json.dumps({"Name": "n", "CAMERA": {"FOV": 33}, "objects": None, "sky": None, "extra": 1,
                             "materials": [{"ID": "a", "Type": "metal", "Smoothness": 0.5}]}))
    sc = scene.load(str(p))
    assert sc.name == "n" and sc.camera.fov == 33.0 and sc.objects == [] and sc.sky is None and sc.fog is None
    assert sc.materials[0].id == "a" and sc.materials[0].smoothness == 0.5 and sc.materials[0].rough == 0.0
    with pytest.raises(OSError, match="open scene"):
        scene.load(str(tmp_path / "missing.json"))
    p.write_text("{ not json")
    with pytest.raises(ValueError, match="decode scene"):
        scene.load(str(p))


def test_flat_scene_matches_oracle_harness():
    # the product's flattening (hip.FlatScene) and the test harness' independent flattening (oracle/ora.py)
    # must describe the same scene field by field
    from oracle import ora
    from path_trace_golang_amd import hip, scene

    for name in SCENE_NAMES:
        f = hip.FlatScene(scene.load(scene_path(name))).c
        o = ora.Scene.load(scene_path(name)).c
        assert f.num_materials == o.nmaterials and f.num_objects == o.nobjects
        for i in range(f.num_materials):
            a, b = f.materials[i], o.materials[i]
            assert (a.type, list(a.albedo), a.rough, a.ior, list(a.emit), a.power, list(a.absorption), a.smoothness) == \
                   (b.type, list(b.albedo), b.rough, b.ior, list(b.emit), b.power, list(b.absorption), b.smoothness)
        for i in range(f.num_objects):
            a, b = f.objects[i], o.objects[i]
            assert (a.type, a.material, list(a.position), list(a.size)) == (b.type, b.material, list(b.position), list(b.size))
        assert list(f.camera.position) == list(o.camera.position) and f.camera.fov == o.camera.fov
        assert f.sky.kind == o.sky.sky_type and list(f.sky.horizon) == list(o.sky.horizon)


def test_duplicate_and_missing_material_ids():
    from path_trace_golang_amd import hip, scene

    sc = scene.Scene.decode({"materials": [{"id": "m", "type": "lambert"}, {"id": "m", "type": "mirror"}],
                             "objects": [{"type": "sphere", "material_id": "m"}, {"type": "box", "material_id": "zz"},
                                         {"type": "teapot", "material_id": "m"}]})
    f = hip.FlatScene(sc).c
    assert f.objects[0].material == 1          # the later duplicate wins (objects.go:227-229)
    assert f.objects[1].material == -1         # unknown id -> zero material
    assert f.objects[2].type == -1             # unknown type is marked and skipped by the core


def test_engine_surface():
    from path_trace_golang_amd import engine, scene

    assert engine.render_settings_for_mode("final") == scene.RenderSettings(1920, 1080, 1000, 80)
    assert engine.render_settings_for_mode("preview") == scene.RenderSettings(400, 225, 20, 20)
    assert engine.render_settings_for_mode("anything") == scene.RenderSettings(400, 225, 20, 20)
    engine.set_backend(engine.BackendCPU)
    assert engine.get_backend() == engine.BackendCPU
    engine.set_backend(7)  # unknown -> CPU (backend.go:16-23)
    assert engine.get_backend() == engine.BackendCPU
    with pytest.raises(NotImplementedError):
        engine.render_into(scene.Scene(), engine.RenderConfig(4, 4, 1, 1), engine.new_image(4, 4))
    engine.set_backend(engine.BackendGPU)
    assert engine.get_backend() == engine.BackendGPU
    img = engine.new_image(5, 3)
    assert img.shape == (3, 5, 4) and img.dtype == np.uint8 and not img.any()


def test_save_png(tmp_path):
    from PIL import Image

    from path_trace_golang_amd import engine

    img = engine.new_image(6, 4)
    img[..., 0] = 200
    img[..., 3] = 255
    p = str(tmp_path / "o.png")
    engine.save_png(p, img)
    back = np.array(Image.open(p))
    assert back.shape == (4, 6, 4) and np.array_equal(back, img)
    with pytest.raises(OSError, match="create png"):
        engine.save_png(str(tmp_path / "nodir" / "o.png"), img)


def test_load_reads_one_json_value_like_the_go_decoder(tmp_path):
    # scene.Load uses json.NewDecoder(f).Decode: data after the first value is not looked at (io.go:17)
    from path_trace_golang_amd import scene

    p = tmp_path / "s.json"
    p.write_text('  {"name": "n", "camera": {"fov": 33}}\n{"name": "second"} trailing')
    sc = scene.load(str(p))
    assert sc.name == "n" and sc.camera.fov == 33
    p.write_text("")
    with pytest.raises(ValueError, match="decode scene"):
        scene.load(str(p))
