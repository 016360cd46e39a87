// render_main.cpp -- CLI twin of the reference's cmd/render (/root/reference/cmd/render/main.go:14-63)
// for machines without a Go toolchain.  Same five flags with the same defaults and Go's flag syntax
// (-flag value, -flag=value, --flag, boolean -flag / -flag=false); the interactive UI is not part of
// this build, so running without -headless is an error instead of opening a window.
//
// Additive flags (the reference CLI cannot express the benchmark configurations, main.go:52 and
// util.go:25-42 hard-wire two presets): -width -height -spp -depth override the mode preset,
// -seed selects the sample streams (also env PATHTRACER_SEED), -devices N uses N GPUs (0..N-1),
// -scene-settings applies the editor's scene-settings override (internal/ui/app.go:60-75) before them;
// without it scene.settings is ignored exactly as main.go:52 does.
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>

#include "engine.hpp"
#include "scene.hpp"

using namespace pthost;

namespace {

void logf(const char *fmt, ...) {  // log.Printf: "2006/01/02 15:04:05 msg" on stderr
    std::time_t t = std::time(nullptr);
    std::tm tm;
    localtime_r(&t, &tm);
    char stamp[32];
    std::strftime(stamp, sizeof stamp, "%Y/%m/%d %H:%M:%S", &tm);
    std::fprintf(stderr, "%s ", stamp);
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fputc('\n', stderr);
}

struct Flags {
    std::string scene = "scenes/example_simple.json";
    std::string mode = "preview";
    bool gpu = false;
    bool headless = false;
    bool scene_settings = false;
    std::string out = "output.png";
    int width = 0, height = 0, spp = -1, depth = -1, devices = 1;
    unsigned long long seed = 1;
};

void usage() {
    std::fprintf(stderr,
                 "Usage of render:\n"
                 "  -depth int\n    \tmax path depth (default: the mode preset)\n"
                 "  -devices int\n    \tnumber of GPUs to tile the image over (default 1)\n"
                 "  -gpu\n    \tuse GPU backend for rendering (if available)\n"
                 "  -headless\n    \trender without UI and save PNG\n"
                 "  -height int\n    \timage height (default: the mode preset)\n"
                 "  -mode string\n    \trender mode: preview or final (default \"preview\")\n"
                 "  -out string\n    \toutput PNG file for headless render (default \"output.png\")\n"
                 "  -scene string\n    \tpath to scene JSON file (default \"scenes/example_simple.json\")\n"
                 "  -scene-settings\n    \tlet the scene file's settings block override the mode preset (the editor's rule); with -mode final this also\n"
                 "    \tapplies the editor's Final button (x4 samples, x2 depth, internal/ui/app.go:73-75), which in the editor is\n"
                 "    \tindependent of the preset\n"
                 "  -seed uint\n    \tsample-stream seed (default 1, or PATHTRACER_SEED)\n"
                 "  -spp int\n    \tsamples per pixel (default: the mode preset)\n"
                 "  -width int\n    \timage width (default: the mode preset)\n");
}

bool parse_bool(const std::string &v, bool &out) {
    if (v == "1" || v == "t" || v == "T" || v == "true" || v == "TRUE" || v == "True") { out = true; return true; }
    if (v == "0" || v == "f" || v == "F" || v == "false" || v == "FALSE" || v == "False") { out = false; return true; }
    return false;
}

// Go's flag package rules
int parse(int argc, char **argv, Flags &f) {
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() < 2 || a[0] != '-') break;  // first non-flag argument ends parsing
        if (a == "--") break;
        size_t dash = a[1] == '-' ? 2 : 1;
        std::string name = a.substr(dash), val;
        bool has_val = false;
        size_t eq = name.find('=');
        if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
        if (name == "h" || name == "help") { usage(); return 0; }
        if (name == "gpu" || name == "headless" || name == "scene-settings") {
            bool b = true;
            if (has_val && !parse_bool(val, b)) {
                std::fprintf(stderr, "invalid boolean value \"%s\" for -%s: parse error\n", val.c_str(), name.c_str());
                usage();
                return 2;
            }
            (name == "gpu" ? f.gpu : name == "headless" ? f.headless : f.scene_settings) = b;
            continue;
        }
        static const char *known[] = {"scene", "mode", "out", "width", "height", "spp", "depth", "seed", "devices"};
        bool ok = false;
        for (const char *k : known) ok = ok || name == k;
        if (!ok) {
            std::fprintf(stderr, "flag provided but not defined: -%s\n", name.c_str());
            usage();
            return 2;
        }
        if (!has_val) {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "flag needs an argument: -%s\n", name.c_str());
                usage();
                return 2;
            }
            val = argv[++i];
        }
        if (name == "scene") f.scene = val;
        else if (name == "mode") f.mode = val;
        else if (name == "out") f.out = val;
        else {
            char *end = nullptr;
            errno = 0;
            long long n = std::strtoll(val.c_str(), &end, 10);
            if (errno || end == val.c_str() || *end) {
                std::fprintf(stderr, "invalid value \"%s\" for flag -%s: parse error\n", val.c_str(), name.c_str());
                usage();
                return 2;
            }
            if (name == "width") f.width = (int)n;
            else if (name == "height") f.height = (int)n;
            else if (name == "spp") f.spp = (int)n;
            else if (name == "depth") f.depth = (int)n;
            else if (name == "devices") f.devices = (int)n;
            else f.seed = (unsigned long long)n;
        }
    }
    return -1;
}

// renderHeadless, main.go:46-63
int render_headless(const Flags &f) {
    std::unique_ptr<scene::Scene> sc;
    try {
        sc = scene::Load(f.scene);
    } catch (const std::exception &e) {
        logf("headless render error: load scene: %s", e.what());
        return 1;
    }
    // scene.settings is ignored, like main.go:52, unless -scene-settings asks for the editor's rule (ui/app.go:60-75)
    scene::RenderSettings s = f.scene_settings ? engine::RenderSettingsForScene(*sc, f.mode) : engine::RenderSettingsForMode(f.mode);
    if (f.width > 0) s.Width = f.width;
    if (f.height > 0) s.Height = f.height;
    if (f.spp >= 0) s.SamplesPerPx = f.spp;
    if (f.depth >= 0) s.MaxDepth = f.depth;
    try {
        if (f.devices > 1) {
            std::vector<int> ords;
            for (int i = 0; i < f.devices; i++) ords.push_back(i);
            engine::hip::SetDevices(ords);
        }
        engine::RenderConfig cfg;
        cfg.Width = s.Width; cfg.Height = s.Height; cfg.SamplesPerPx = s.SamplesPerPx; cfg.MaxDepth = s.MaxDepth;
        cfg.Seed = f.seed;
        engine::RGBA img = engine::NewRGBA(cfg.Width, cfg.Height);
        engine::Stats st;
        auto t0 = std::chrono::steady_clock::now();
        engine::RenderInto(*sc, cfg, img, nullptr, &st);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        logf("rendered %dx%d, %d spp, depth %d on %d GPU(s) in %.3f s: %.1f M segments/s, %.1f M samples/s", cfg.Width,
             cfg.Height, cfg.SamplesPerPx, cfg.MaxDepth, st.num_devices, dt, st.segments / dt / 1e6, st.samples / dt / 1e6);
        engine::SavePNG(f.out, img);
    } catch (const std::exception &e) {
        logf("headless render error: %s", e.what());
        engine::hip::Shutdown();
        return 1;
    }
    engine::hip::Shutdown();
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    logf("pathtracer: starting main()");
    Flags f;
    if (const char *e = std::getenv("PATHTRACER_SEED")) f.seed = std::strtoull(e, nullptr, 10);
    int rc = parse(argc, argv, f);
    if (rc >= 0) return rc;
    logf("flags: scene=%s mode=%s headless=%s out=%s", f.scene.c_str(), f.mode.c_str(), f.headless ? "true" : "false",
         f.out.c_str());
    // main.go:26-30: -gpu selects BackendGPU, otherwise BackendCPU.  Only the GPU branch exists in this
    // build, so without -gpu the render fails with a clear message instead of using a different engine.
    engine::SetBackend(f.gpu ? engine::BackendGPU : engine::BackendCPU);
    if (f.headless) return render_headless(f);
    logf("ui error: the interactive UI (internal/ui) is not part of this build; use -headless");
    return 1;
}
