// engine.hpp -- C++ mirror of the reference's internal/engine public surface for the render path
// (/root/reference/internal/engine/renderer.go:17-41, backend.go:5-28, util.go:13-55) and of the
// backend plug-in shape gpu.Render (/root/reference/internal/engine/gpu/gpu.go:2534).
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "scene.hpp"

namespace pthost {
namespace engine {

// image.RGBA: Pix holds Height rows of Stride bytes, 4 bytes per pixel, row 0 on top.
struct RGBA {
    int Width = 0, Height = 0, Stride = 0;
    std::vector<uint8_t> Pix;
};
RGBA NewRGBA(int width, int height);  // image.NewRGBA(image.Rect(0, 0, w, h))

struct RenderConfig {  // renderer.go:17-22 (+ the stream seed; the reference seeds from the clock)
    int Width = 0, Height = 0, SamplesPerPx = 0, MaxDepth = 0;
    uint64_t Seed = 1;
};

enum Backend { BackendCPU = 0, BackendGPU = 1 };  // backend.go:7-10
void SetBackend(int b);                           // backend.go:16-23: unknown values select the CPU backend
Backend GetBackend();

struct Stats {
    uint64_t samples = 0, segments = 0, exit_scans = 0, draws = 0;
    double seconds = 0, trace_ms = 0, resolve_ms = 0, device_ms = 0;
    int num_devices = 0, spp_chunk = 0;
};

namespace hip {
// The MI355X backend: same shape as gpu.Render(sc, cfg, img, progress) error.
// Returns "" on success, the error text otherwise.  Never renders on the CPU.
std::string Render(const scene::Scene &sc, const RenderConfig &cfg, RGBA &img, const std::function<void()> &progress,
                   Stats *stats = nullptr);
void SetDevices(const std::vector<int> &ordinals);  // HIP ordinals used by Render (default: device 0)
void Shutdown();                                    // releases the process-wide context
}  // namespace hip

// RenderInto (renderer.go:34-41).  BackendGPU goes to hip::Render.  The CPU branch is the reference's
// own Go renderer and is not shipped: selecting it, or a GPU failure (where the reference falls back
// to it, renderer.go:257-262), throws std::runtime_error here.
void RenderInto(const scene::Scene &sc, const RenderConfig &cfg, RGBA &img, const std::function<void()> &progress,
                Stats *stats = nullptr);
RGBA Render(const scene::Scene &sc, const RenderConfig &cfg);                       // renderer.go:25-29
RGBA RenderScene(const scene::Scene &sc, const scene::RenderSettings &settings, uint64_t seed = 1);  // util.go:13-22
scene::RenderSettings RenderSettingsForMode(const std::string &mode);               // util.go:25-42
// The editor's "scene settings override" (internal/ui/app.go:60-75): the mode preset, replaced by the scene's own
// width x height when BOTH are > 0 (and, only then, by its samples_per_px / max_depth when > 0); a "final" render
// then takes four times the samples and twice the depth (app.go:72-75).  cmd/render ignores scene.settings
// (main.go:52); the CLI twin applies this rule only under -scene-settings.
scene::RenderSettings RenderSettingsForScene(const scene::Scene &sc, const std::string &mode);
void SavePNG(const std::string &path, const RGBA &img);                             // util.go:45-55; throws "create png: ..."

}  // namespace engine
}  // namespace pthost
