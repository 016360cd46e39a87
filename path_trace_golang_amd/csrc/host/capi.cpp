// capi.cpp -- small C surface over the C++ host mirror so that the Python tests can drive it
// (ctypes).  Not part of the rendering boundary (that is include/ptcore.h).
#include <cstring>
#include <string>

#include "../../../include/ptcore.h"
#include "engine.hpp"
#include "scene.hpp"

namespace pthost {
namespace engine {
void FlattenScene(const scene::Scene &sc, std::vector<pt_material> &materials, std::vector<pt_object> &objects,
                  pt_scene &out);
}
}  // namespace pthost

namespace {
thread_local std::string g_err;
struct Handle {
    std::unique_ptr<pthost::scene::Scene> sc;
    std::vector<pt_material> materials;
    std::vector<pt_object> objects;
    pt_scene flat;
    std::string text;
};
}  // namespace

extern "C" {

const char *pth_last_error(void) { return g_err.c_str(); }

void *pth_scene_load(const char *path) {
    try {
        auto h = new Handle();
        h->sc = pthost::scene::Load(path);
        pthost::engine::FlattenScene(*h->sc, h->materials, h->objects, h->flat);
        return h;
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

void *pth_scene_decode(const char *json_text) {
    try {
        auto h = new Handle();
        h->sc = pthost::scene::Decode(json_text);
        pthost::engine::FlattenScene(*h->sc, h->materials, h->objects, h->flat);
        return h;
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

void pth_scene_free(void *h) { delete static_cast<Handle *>(h); }

const pt_scene *pth_scene_flat(void *h) { return &static_cast<Handle *>(h)->flat; }

int pth_scene_has_sky(void *h) { return static_cast<Handle *>(h)->sc->SkyPtr ? 1 : 0; }
int pth_scene_has_fog(void *h) { return static_cast<Handle *>(h)->sc->FogPtr ? 1 : 0; }

// scene.Save to a string; the pointer stays valid until the next call on the same handle
const char *pth_scene_encode(void *h) {
    try {
        Handle *hd = static_cast<Handle *>(h);
        hd->text = pthost::scene::Encode(*hd->sc);
        return hd->text.c_str();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

int pth_scene_save(void *h, const char *path) {
    try {
        pthost::scene::Save(path, *static_cast<Handle *>(h)->sc);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

void pth_set_backend(int b) { pthost::engine::SetBackend(b); }
int pth_get_backend(void) { return (int)pthost::engine::GetBackend(); }

void pth_settings_for_mode(const char *mode, int32_t out[4]) {
    auto s = pthost::engine::RenderSettingsForMode(mode);
    out[0] = s.Width; out[1] = s.Height; out[2] = s.SamplesPerPx; out[3] = s.MaxDepth;
}

// internal/ui/app.go:60-75 (what the CLI twin applies under -scene-settings)
void pth_settings_for_scene(void *h, const char *mode, int32_t out[4]) {
    auto s = pthost::engine::RenderSettingsForScene(*static_cast<Handle *>(h)->sc, mode);
    out[0] = s.Width; out[1] = s.Height; out[2] = s.SamplesPerPx; out[3] = s.MaxDepth;
}

// engine.RenderInto into caller memory; progress may be NULL. Returns 0 or 1 (pth_last_error()).
int pth_render_into(void *h, int32_t width, int32_t height, int32_t spp, int32_t depth, uint64_t seed, uint8_t *pix,
                    int32_t img_width, int32_t img_height, int32_t stride, void (*progress)(void)) {
    try {
        pthost::engine::RGBA img;
        img.Width = img_width;
        img.Height = img_height;
        img.Stride = stride;
        img.Pix.assign(pix, pix + (size_t)stride * (size_t)img_height);
        pthost::engine::RenderConfig cfg;
        cfg.Width = width; cfg.Height = height; cfg.SamplesPerPx = spp; cfg.MaxDepth = depth; cfg.Seed = seed;
        std::function<void()> cb;
        if (progress) cb = [progress]() { progress(); };
        pthost::engine::RenderInto(*static_cast<Handle *>(h)->sc, cfg, img, cb);
        std::memcpy(pix, img.Pix.data(), img.Pix.size());
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

int pth_save_png(const char *path, const uint8_t *pix, int32_t width, int32_t height, int32_t stride) {
    try {
        pthost::engine::RGBA img;
        img.Width = width; img.Height = height; img.Stride = stride;
        img.Pix.assign(pix, pix + (size_t)stride * (size_t)height);
        pthost::engine::SavePNG(path, img);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return 1;
    }
}

void pth_shutdown(void) { pthost::engine::hip::Shutdown(); }

}  // extern "C"
