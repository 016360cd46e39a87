#include "scene.hpp"

#include <cerrno>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "json.hpp"

namespace pthost {
namespace scene {

namespace {

using json::Value;

Vec3 vec3_of(const Value *v) {
    Vec3 r;
    if (v) { r.X = v->number("x"); r.Y = v->number("y"); r.Z = v->number("z"); }
    return r;
}
Color color_of(const Value *v) {
    Color c;
    if (v) { c.R = v->number("r"); c.G = v->number("g"); c.B = v->number("b"); }
    return c;
}

void put(json::Writer &w, const char *k, const Vec3 &v) {
    w.key(k); w.begin_object();
    w.key("x"); w.value(v.X); w.key("y"); w.value(v.Y); w.key("z"); w.value(v.Z);
    w.end_object();
}
void put(json::Writer &w, const char *k, const Color &c) {
    w.key(k); w.begin_object();
    w.key("r"); w.value(c.R); w.key("g"); w.value(c.G); w.key("b"); w.value(c.B);
    w.end_object();
}

}  // namespace

std::unique_ptr<Scene> Decode(const std::string &text) {
    json::ValuePtr root;
    try {
        root = json::parse(text);
    } catch (const std::exception &e) {
        throw std::runtime_error(std::string("decode scene: ") + e.what());
    }
    if (root->kind != Value::Object) throw std::runtime_error("decode scene: top-level JSON value is not an object");
    auto sc = std::make_unique<Scene>();
    // every typed read below throws on a value of the wrong JSON type: scene.Load fails on json.Unmarshal's
    // UnmarshalTypeError the same way (io.go:17-19)
    try {
        // a null array element leaves the zero struct; anything but an object is a type error
        auto element = [](const json::ValuePtr &e, const char *field) -> const Value * {
            if (e->kind == Value::Null) return nullptr;
            if (e->kind != Value::Object)
                throw std::runtime_error(std::string("json: cannot unmarshal ") + Value::kind_name(e->kind) + " into field " + field +
                                         " of type struct");
            return e.get();
        };
        sc->Name = root->string("name");
        if (const Value *c = root->object("camera")) {
            sc->Cam.Position = vec3_of(c->object("position"));
            sc->Cam.Target = vec3_of(c->object("target"));
            sc->Cam.Up = vec3_of(c->object("up"));
            sc->Cam.FOV = c->number("fov");
            sc->Cam.Aperture = c->number("aperture");
            sc->Cam.FocusDist = c->number("focus_dist");
            sc->Cam.AspectRatio = c->number("aspect_ratio");
        }
        if (const Value *a = root->array("objects")) {
            sc->ObjectsNil = false;
            for (const auto &el : a->arr) {
                Object o;
                if (const Value *e = element(el, "objects")) {
                    o.ID = e->string("id");
                    o.Type = e->string("type");
                    o.Position = vec3_of(e->object("position"));
                    o.Size = vec3_of(e->object("size"));
                    o.MaterialID = e->string("material_id");
                }
                sc->Objects.push_back(o);
            }
        }
        if (const Value *a = root->array("materials")) {
            sc->MaterialsNil = false;
            for (const auto &el : a->arr) {
                Material m;
                if (const Value *e = element(el, "materials")) {
                    m.ID = e->string("id");
                    m.Type = e->string("type");
                    m.Albedo = color_of(e->object("albedo"));
                    m.Rough = e->number("rough");
                    m.IOR = e->number("ior");
                    m.Emit = color_of(e->object("emit"));
                    m.Power = e->number("power");
                    m.Absorption = color_of(e->object("absorption"));
                    m.Smoothness = e->number("smoothness");
                    m.Reflectivity = e->number("reflectivity");
                    m.Tint = color_of(e->object("tint"));
                    m.AbsorptionScale = e->number("absorption_scale");
                }
                sc->Materials.push_back(m);
            }
        }
        if (const Value *s = root->object("settings")) {
            sc->Settings.Width = (int)s->integer("width");
            sc->Settings.Height = (int)s->integer("height");
            sc->Settings.SamplesPerPx = (int)s->integer("samples_per_px");
            sc->Settings.MaxDepth = (int)s->integer("max_depth");
        }
        sc->Background = color_of(root->object("background"));
        if (const Value *s = root->object("sky")) {
            sc->SkyPtr = std::make_unique<Sky>();
            sc->SkyPtr->Type = s->string("type");
            sc->SkyPtr->Col = color_of(s->object("color"));
            sc->SkyPtr->Horizon = color_of(s->object("horizon"));
            sc->SkyPtr->Zenith = color_of(s->object("zenith"));
        }
        if (const Value *f = root->object("fog")) {
            sc->FogPtr = std::make_unique<Fog>();
            Fog &g = *sc->FogPtr;
            g.Density = f->number("density");
            g.Col = color_of(f->object("color"));
            g.Scatter = f->number("scatter");
            g.SigmaS = f->number("sigma_s");
            g.SigmaA = f->number("sigma_a");
            g.G = f->number("g");
            g.HeteroStrength = f->number("hetero_strength");
            g.NoiseScale = f->number("noise_scale");
            g.NoiseOctaves = (int)f->integer("noise_octaves");
            g.AffectSky = f->boolean("affect_sky");
            g.GPUVolumetric = f->boolean("gpu_volumetric");
        }
    } catch (const std::exception &e) {
        throw std::runtime_error(std::string("decode scene: ") + e.what());
    }
    return sc;
}

std::unique_ptr<Scene> Load(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("open scene: open " + path + ": " + std::strerror(errno));
    std::stringstream ss;
    ss << f.rdbuf();
    return Decode(ss.str());
}

std::string Encode(const Scene &sc) {
    json::Writer w;
    try {
        w.begin_object();
        w.key("name"); w.value(sc.Name);
        w.key("camera"); w.begin_object();
        put(w, "position", sc.Cam.Position); put(w, "target", sc.Cam.Target); put(w, "up", sc.Cam.Up);
        w.key("fov"); w.value(sc.Cam.FOV);
        w.key("aperture"); w.value(sc.Cam.Aperture);
        w.key("focus_dist"); w.value(sc.Cam.FocusDist);
        w.key("aspect_ratio"); w.value(sc.Cam.AspectRatio);
        w.end_object();
        w.key("objects");
        if (sc.ObjectsNil && sc.Objects.empty()) w.null(); else {
        w.begin_array();
        for (const Object &o : sc.Objects) {
            w.begin_object();
            w.key("id"); w.value(o.ID);
            w.key("type"); w.value(o.Type);
            put(w, "position", o.Position); put(w, "size", o.Size);
            w.key("material_id"); w.value(o.MaterialID);
            w.end_object();
        }
        w.end_array();
        }
        w.key("materials");
        if (sc.MaterialsNil && sc.Materials.empty()) w.null(); else {
        w.begin_array();
        for (const Material &m : sc.Materials) {
            w.begin_object();
            w.key("id"); w.value(m.ID);
            w.key("type"); w.value(m.Type);
            put(w, "albedo", m.Albedo);
            w.key("rough"); w.value(m.Rough);
            w.key("ior"); w.value(m.IOR);
            put(w, "emit", m.Emit);
            w.key("power"); w.value(m.Power);
            put(w, "absorption", m.Absorption);
            w.key("smoothness"); w.value(m.Smoothness);
            w.key("reflectivity"); w.value(m.Reflectivity);
            put(w, "tint", m.Tint);
            w.key("absorption_scale"); w.value(m.AbsorptionScale);
            w.end_object();
        }
        w.end_array();
        }
        w.key("settings"); w.begin_object();
        w.key("width"); w.value((long long)sc.Settings.Width);
        w.key("height"); w.value((long long)sc.Settings.Height);
        w.key("samples_per_px"); w.value((long long)sc.Settings.SamplesPerPx);
        w.key("max_depth"); w.value((long long)sc.Settings.MaxDepth);
        w.end_object();
        put(w, "background", sc.Background);
        w.key("sky");
        if (sc.SkyPtr) {
            w.begin_object();
            w.key("type"); w.value(sc.SkyPtr->Type);
            put(w, "color", sc.SkyPtr->Col); put(w, "horizon", sc.SkyPtr->Horizon); put(w, "zenith", sc.SkyPtr->Zenith);
            w.end_object();
        } else {
            w.null();
        }
        if (sc.FogPtr) {  // `json:"fog,omitempty"`
            const Fog &g = *sc.FogPtr;
            w.key("fog"); w.begin_object();
            w.key("density"); w.value(g.Density);
            put(w, "color", g.Col);
            w.key("scatter"); w.value(g.Scatter);
            w.key("sigma_s"); w.value(g.SigmaS);
            w.key("sigma_a"); w.value(g.SigmaA);
            w.key("g"); w.value(g.G);
            w.key("hetero_strength"); w.value(g.HeteroStrength);
            w.key("noise_scale"); w.value(g.NoiseScale);
            w.key("noise_octaves"); w.value((long long)g.NoiseOctaves);
            w.key("affect_sky"); w.value(g.AffectSky);
            w.key("gpu_volumetric"); w.value(g.GPUVolumetric);
            w.end_object();
        }
        w.end_object();
    } catch (const std::exception &e) {
        throw std::runtime_error(std::string("encode scene: ") + e.what());
    }
    return w.str() + "\n";
}

void Save(const std::string &path, const Scene &sc) {
    std::string text = Encode(sc);
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("create scene: open " + path + ": " + std::strerror(errno));
    f << text;
    if (!f) throw std::runtime_error("encode scene: write failed");
}

}  // namespace scene
}  // namespace pthost
