#include "engine.hpp"

#include <dlfcn.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <stdexcept>

#include "../../../include/ptcore.h"
#include "png.hpp"

namespace pthost {
namespace engine {

RGBA NewRGBA(int width, int height) {
    RGBA img;
    img.Width = width;
    img.Height = height;
    img.Stride = 4 * width;
    img.Pix.assign((size_t)img.Stride * (size_t)(height > 0 ? height : 0), 0);
    return img;
}

namespace {
// The reference starts on BackendCPU (backend.go:12); this host layer only has the GPU branch.
Backend g_backend = BackendGPU;
}  // namespace

void SetBackend(int b) {
    switch (b) {
        case BackendCPU:
        case BackendGPU:
            g_backend = (Backend)b;
            break;
        default:
            g_backend = BackendCPU;
    }
}
Backend GetBackend() { return g_backend; }

// ---------------------------------------------------------------- libptcore binding (dlopen)
namespace {

struct Core {
    void *handle = nullptr;
    decltype(&pt_abi_version) abi_version = nullptr;
    decltype(&pt_last_error) last_error = nullptr;
    decltype(&pt_create) create = nullptr;
    decltype(&pt_destroy) destroy = nullptr;
    decltype(&pt_render) render = nullptr;
    decltype(&pt_begin) begin = nullptr;
    decltype(&pt_step) step = nullptr;
    decltype(&pt_read) read = nullptr;
    decltype(&pt_end) end = nullptr;
    std::string error;  // sticky load error, like the reference's cached GL init failure (gpu.go:279-286)
};

Core g_core;
pt_ctx *g_ctx = nullptr;
std::vector<int> g_devices;
std::mutex g_mu;  // requests are serialised, like the reference's single GL worker (gpu.go:2534-2546)

std::string self_dir() {
    Dl_info info;
    if (dladdr((void *)&self_dir, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t k = p.find_last_of('/');
        if (k != std::string::npos) return p.substr(0, k);
    }
    return ".";
}

template <typename F>
bool sym(void *h, const char *name, F &out, std::string &err) {
    out = reinterpret_cast<F>(dlsym(h, name));
    if (!out) { err = std::string("libptcore.so lacks symbol ") + name; return false; }
    return true;
}

bool load_core() {
    if (g_core.handle) return true;
    if (!g_core.error.empty()) return false;
    std::vector<std::string> cands;
    if (const char *e = std::getenv("PTCORE_LIB")) cands.push_back(e);
    cands.push_back(self_dir() + "/libptcore.so");
    cands.push_back("libptcore.so");
    std::string errs;
    void *h = nullptr;
    for (const std::string &c : cands) {
        h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        errs += std::string(dlerror() ? dlerror() : "dlopen failed") + "; ";
    }
    if (!h) { g_core.error = "cannot load libptcore.so (" + errs + ")"; return false; }
    std::string err;
    Core c;
    c.handle = h;
    if (!(sym(h, "pt_abi_version", c.abi_version, err) && sym(h, "pt_last_error", c.last_error, err) &&
          sym(h, "pt_create", c.create, err) && sym(h, "pt_destroy", c.destroy, err) && sym(h, "pt_render", c.render, err) &&
          sym(h, "pt_begin", c.begin, err) && sym(h, "pt_step", c.step, err) && sym(h, "pt_read", c.read, err) &&
          sym(h, "pt_end", c.end, err))) {
        g_core.error = err;
        dlclose(h);
        return false;
    }
    if (c.abi_version() != PT_ABI_VERSION) {
        g_core.error = "libptcore.so ABI version mismatch";
        dlclose(h);
        return false;
    }
    g_core = c;
    return true;
}

struct Flat {
    std::vector<pt_material> materials;
    std::vector<pt_object> objects;
    pt_scene sc;
};

int material_type(const std::string &t) {
    if (t == scene::MaterialMetal) return PT_MAT_METAL;
    if (t == scene::MaterialDielectric) return PT_MAT_DIELECTRIC;
    if (t == scene::MaterialEmissive) return PT_MAT_EMISSIVE;
    if (t == scene::MaterialMirror) return PT_MAT_MIRROR;
    return PT_MAT_LAMBERT;  // convertMaterial's default branch (materials.go:51-53)
}
int object_type(const std::string &t) {
    if (t == scene::ObjectSphere) return PT_OBJ_SPHERE;
    if (t == scene::ObjectPlane) return PT_OBJ_PLANE;
    if (t == scene::ObjectBox) return PT_OBJ_BOX;
    if (t == scene::ObjectSphereLight) return PT_OBJ_SPHERE_LIGHT;
    return PT_OBJ_UNKNOWN;
}
void set3(double *d, const scene::Vec3 &v) { d[0] = v.X; d[1] = v.Y; d[2] = v.Z; }
void set3(double *d, const scene::Color &c) { d[0] = c.R; d[1] = c.G; d[2] = c.B; }

}  // namespace

// Flattens scene.Scene into the C ABI's plain structs.  Exposed for tests through capi.cpp.
void FlattenScene(const scene::Scene &sc, std::vector<pt_material> &materials, std::vector<pt_object> &objects,
                  pt_scene &out) {
    materials.clear();
    objects.clear();
    std::map<std::string, int> ids;
    for (size_t i = 0; i < sc.Materials.size(); i++) {
        const scene::Material &m = sc.Materials[i];
        pt_material pm;
        std::memset(&pm, 0, sizeof pm);
        pm.type = material_type(m.Type);
        set3(pm.albedo, m.Albedo);
        pm.rough = m.Rough;
        pm.ior = m.IOR;
        set3(pm.emit, m.Emit);
        pm.power = m.Power;
        set3(pm.absorption, m.Absorption);
        pm.smoothness = m.Smoothness;
        materials.push_back(pm);
        ids[m.ID] = (int)i;  // later duplicates replace earlier ones (objects.go:227-229)
    }
    for (const scene::Object &o : sc.Objects) {
        pt_object po;
        std::memset(&po, 0, sizeof po);
        po.type = object_type(o.Type);
        auto it = ids.find(o.MaterialID);
        po.material = it == ids.end() ? -1 : it->second;
        set3(po.position, o.Position);
        set3(po.size, o.Size);
        objects.push_back(po);
    }
    std::memset(&out, 0, sizeof out);
    set3(out.camera.position, sc.Cam.Position);
    set3(out.camera.target, sc.Cam.Target);
    set3(out.camera.up, sc.Cam.Up);
    out.camera.fov = sc.Cam.FOV;
    out.camera.aperture = sc.Cam.Aperture;
    out.camera.focus_dist = sc.Cam.FocusDist;
    out.camera.aspect_ratio = sc.Cam.AspectRatio;
    set3(out.sky.background, sc.Background);
    out.sky.kind = PT_SKY_BACKGROUND;
    if (sc.SkyPtr) {
        if (sc.SkyPtr->Type == "gradient") out.sky.kind = PT_SKY_GRADIENT;
        else if (sc.SkyPtr->Type == "solid") out.sky.kind = PT_SKY_SOLID;
        set3(out.sky.color, sc.SkyPtr->Col);
        set3(out.sky.horizon, sc.SkyPtr->Horizon);
        set3(out.sky.zenith, sc.SkyPtr->Zenith);
    }
    out.num_materials = (int32_t)materials.size();
    out.num_objects = (int32_t)objects.size();
    out.materials = materials.empty() ? nullptr : materials.data();
    out.objects = objects.empty() ? nullptr : objects.data();
}

namespace hip {

void SetDevices(const std::vector<int> &ordinals) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx && g_core.handle) { g_core.destroy(g_ctx); g_ctx = nullptr; }
    g_devices = ordinals;
}

void Shutdown() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx && g_core.handle) g_core.destroy(g_ctx);
    g_ctx = nullptr;
}

std::string Render(const scene::Scene &sc, const RenderConfig &cfg, RGBA &img, const std::function<void()> &progress,
                   Stats *stats) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (img.Width != cfg.Width || img.Height != cfg.Height) return "";  // renderIntoCPU: silently do nothing (renderer.go:46-49)
    if (!load_core()) return g_core.error;
    if (!g_ctx) {
        int32_t n = g_devices.empty() ? 1 : (int32_t)g_devices.size();
        std::vector<int32_t> ords(g_devices.begin(), g_devices.end());
        if (g_core.create(ords.empty() ? nullptr : ords.data(), n, &g_ctx) != PT_OK)
            return std::string("pt_create: ") + g_core.last_error();
    }
    Flat flat;
    FlattenScene(sc, flat.materials, flat.objects, flat.sc);
    pt_config pc;
    std::memset(&pc, 0, sizeof pc);
    pc.width = cfg.Width;
    pc.height = cfg.Height;
    pc.samples_per_px = cfg.SamplesPerPx;
    pc.max_depth = cfg.MaxDepth;
    pc.seed = cfg.Seed;
    pt_stats st;
    std::memset(&st, 0, sizeof st);
    std::string err;
    if (!progress) {
        if (g_core.render(g_ctx, &flat.sc, &pc, img.Pix.data(), img.Stride, nullptr, nullptr, nullptr, &st) != PT_OK)
            err = std::string("pt_render: ") + g_core.last_error();
    } else {
        if (g_core.begin(g_ctx, &flat.sc, &pc) != PT_OK) return std::string("pt_begin: ") + g_core.last_error();
        // refresh the preview every spp/10 samples and once at the end (gpu.go:2209-2212, :2229, :2523-2525)
        const int32_t step = cfg.SamplesPerPx / 10 > 0 ? cfg.SamplesPerPx / 10 : 1;
        int32_t done = 0;
        while (err.empty() && done < cfg.SamplesPerPx) {
            if (g_core.step(g_ctx, step, &done) != PT_OK) { err = std::string("pt_step: ") + g_core.last_error(); break; }
            if (g_core.read(g_ctx, img.Pix.data(), img.Stride, nullptr) != PT_OK) { err = std::string("pt_read: ") + g_core.last_error(); break; }
            progress();
        }
        if (err.empty() && cfg.SamplesPerPx <= 0 && g_core.read(g_ctx, img.Pix.data(), img.Stride, nullptr) != PT_OK)
            err = std::string("pt_read: ") + g_core.last_error();
        if (g_core.end(g_ctx, &st) != PT_OK && err.empty()) err = std::string("pt_end: ") + g_core.last_error();
        if (err.empty()) progress();
    }
    if (stats && err.empty()) {
        stats->samples = st.samples; stats->segments = st.segments; stats->exit_scans = st.exit_scans; stats->draws = st.draws;
        stats->seconds = st.seconds; stats->trace_ms = st.trace_ms; stats->resolve_ms = st.resolve_ms;
        stats->device_ms = st.device_ms; stats->num_devices = st.num_devices; stats->spp_chunk = st.spp_chunk;
    }
    return err;
}

}  // namespace hip

void RenderInto(const scene::Scene &sc, const RenderConfig &cfg, RGBA &img, const std::function<void()> &progress,
                Stats *stats) {
    if (GetBackend() != BackendGPU)
        throw std::runtime_error("BackendCPU is the reference's Go renderer (renderIntoCPU) and is not part of this host layer");
    std::string err = hip::Render(sc, cfg, img, progress, stats);
    if (!err.empty()) {
        // the reference prints this and falls back to its CPU renderer (renderer.go:257-262); that fallback
        // lives in the reference, so here the error is surfaced
        std::fprintf(stderr, "GPU render error: %s\n", err.c_str());
        throw std::runtime_error("GPU render error: " + err);
    }
}

RGBA Render(const scene::Scene &sc, const RenderConfig &cfg) {
    RGBA img = NewRGBA(cfg.Width, cfg.Height);
    RenderInto(sc, cfg, img, nullptr);
    return img;
}

RGBA RenderScene(const scene::Scene &sc, const scene::RenderSettings &s, uint64_t seed) {
    RenderConfig cfg;
    cfg.Width = s.Width;
    cfg.Height = s.Height;
    cfg.SamplesPerPx = s.SamplesPerPx;
    cfg.MaxDepth = s.MaxDepth;
    cfg.Seed = seed;
    return Render(sc, cfg);
}

scene::RenderSettings RenderSettingsForMode(const std::string &mode) {
    scene::RenderSettings s;
    if (mode == "final") { s.Width = 1920; s.Height = 1080; s.SamplesPerPx = 1000; s.MaxDepth = 80; }
    else { s.Width = 400; s.Height = 225; s.SamplesPerPx = 20; s.MaxDepth = 20; }
    return s;
}

scene::RenderSettings RenderSettingsForScene(const scene::Scene &sc, const std::string &mode) {
    scene::RenderSettings s = RenderSettingsForMode(mode);  // baseSettings, app.go:60
    if (sc.Settings.Width > 0 && sc.Settings.Height > 0) {  // app.go:61-70
        s.Width = sc.Settings.Width;
        s.Height = sc.Settings.Height;
        if (sc.Settings.SamplesPerPx > 0) s.SamplesPerPx = sc.Settings.SamplesPerPx;
        if (sc.Settings.MaxDepth > 0) s.MaxDepth = sc.Settings.MaxDepth;
    }
    if (mode == "final") {  // finalSettings, app.go:72-75
        s.SamplesPerPx *= 4;
        s.MaxDepth *= 2;
    }
    return s;
}

void SavePNG(const std::string &path, const RGBA &img) {
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("create png: open " + path + ": " + std::strerror(errno));
    std::vector<uint8_t> data = EncodePNG(img.Pix.data(), img.Width, img.Height, img.Stride);
    f.write(reinterpret_cast<const char *>(data.data()), (std::streamsize)data.size());
    if (!f) throw std::runtime_error("encode png: write failed");
}

}  // namespace engine
}  // namespace pthost
