/*
 * pt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, FP64, no FMA contraction) of the reference's Go CPU
 * engine, package internal/engine of MarkJulian19/path_trace_golang.  It is the
 * parity checker for the HIP path and the timed "port" CPU baseline of bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it; the product (path_trace_golang_amd/, include/) never does.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden images
 * or known-answer vectors, its Go toolchain is absent here, and its RNG is
 * time-seeded (internal/engine/random.go:14-16), so no reference output exists
 * to pin this restatement against.  It is pinned instead by hand-derived KATs
 * and closed-form scenes (tests/test_oracle_*.py); see DESIGN.md.
 *
 * The structs below carry exactly the scene.Scene fields the CPU engine reads
 * (internal/scene/scene.go:9-158); convertMaterial and sceneToWorld are
 * restated inside the oracle, not in the harness.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scene.Material (scene.go:41-63), fields read by convertMaterial (materials.go:28-55) */
typedef struct {
    int32_t type;        /* 0 lambert (also any unknown string), 1 metal, 2 dielectric, 3 emissive, 4 mirror */
    int32_t _pad;
    double albedo[3];
    double rough;
    double ior;
    double emit[3];
    double power;
    double absorption[3];
    double smoothness;
} ora_material;

/* scene.Object (scene.go:76-84) */
typedef struct {
    int32_t type;        /* 0 sphere, 1 plane, 2 box, 3 sphere_light, -1 unknown (skipped, objects.go:237-266) */
    int32_t material;    /* index of the LAST material with that id, -1 = id not present -> zero material (objects.go:233) */
    double position[3];
    double size[3];
} ora_object;

/* scene.Camera (scene.go:24-32) */
typedef struct {
    double position[3], target[3], up[3];
    double fov, aperture, focus_dist, aspect_ratio;
} ora_camera;

/* scene.Sky + scene.Background as the closure of renderer.go:56-92 sees them */
typedef struct {
    int32_t sky_type;    /* 0: sky == nil or type not in {gradient, solid} -> background; 1 gradient; 2 solid */
    int32_t _pad;
    double background[3], color[3], horizon[3], zenith[3];
} ora_sky;

typedef struct {
    ora_camera camera;
    ora_sky sky;
    int32_t nmaterials, nobjects;
    const ora_material *materials;
    const ora_object *objects;
} ora_scene;

typedef struct {
    int32_t width, height, spp, max_depth;
    uint64_t seed;
    int32_t workers;     /* 0 = PATHTRACER_WORKERS or online CPUs (renderer.go:117-129) */
    int32_t _pad;
} ora_config;

typedef struct {
    uint64_t samples;    /* W*H*spp */
    uint64_t segments;   /* rayColorOpt activations with depth > 0 (closest-hit scans) */
    uint64_t exit_scans; /* dielectric exit searches (renderer.go:316-371) */
    uint64_t draws;      /* RNG draws */
    double seconds;      /* wall time of the pixel loop */
    int32_t workers;
    int32_t _pad;
} ora_stats;

/*
 * Renders like renderIntoCPU (renderer.go:44-246).
 *   rgba    : H rows of `stride` bytes, row 0 = top (may be NULL)
 *   accum   : W*H*3 doubles, per pixel the raw sum over samples of the sample
 *             radiance, before the 1/spp scale (may be NULL)
 *   nseg    : W*H uint32, per pixel segment count (may be NULL)
 *   ndraw   : W*H uint32, per pixel RNG draw count (may be NULL)
 * Returns 0.  If x0/y0/x1/y1 window is non-empty only those pixels are done.
 */
int ora_render(const ora_scene *sc, const ora_config *cfg, uint8_t *rgba, int32_t stride,
               double *accum, uint32_t *nseg, uint32_t *ndraw, ora_stats *stats);

/* Same, restricted to pixel window [x0,x1) x [y0,y1) of the full W x H frame;
 * buffers are still full-frame sized and only the window is written. */
int ora_render_window(const ora_scene *sc, const ora_config *cfg, int32_t x0, int32_t y0, int32_t x1,
                      int32_t y1, uint8_t *rgba, int32_t stride, double *accum, uint32_t *nseg,
                      uint32_t *ndraw, ora_stats *stats);

/* One sample's radiance for pixel (x, y), sample s: the value renderer.go:186 adds. */
void ora_sample(const ora_scene *sc, const ora_config *cfg, int32_t x, int32_t y, int32_t s,
                double out_rgb[3], uint32_t *nseg, uint32_t *ndraw);

/* ---- unit-level entry points for KATs ---- */
double ora_sin(double x);   /* Go math.Sin (Cephes), pure-Go path */
double ora_cos(double x);
double ora_tan(double x);
double ora_exp(double x);   /* Go math.Exp, pure-Go (FreeBSD) path */
double ora_pow(double x, double y); /* Go math.Pow; only integral y >= 0 supported here */
double ora_min(double a, double b); /* Go math.Min special cases */
double ora_max(double a, double b);
uint64_t ora_stream_init(uint64_t seed, uint64_t pixel, uint64_t sample);
double ora_stream_next(uint64_t *state);

/* hit tests: obj = {kind 0 sphere / 1 plane / 2 box, a[3], b[3], radius};
 * out = {t, p[3], normal[3], frontFace}. Returns 1 on hit. */
int ora_hit(int32_t kind, const double a[3], const double b[3], double radius, const double orig[3],
            const double dir[3], double tmin, double tmax, double out[8]);
/* converted material: out = {typ, albedo[3], rough, ior, emit[3], absorption[3]} (12 doubles) */
void ora_convert_material(const ora_material *m, double out[12]);
/* camera: out = origin[3], lowerLeft[3], horizontal[3], vertical[3], u[3], v[3], w[3], lensRadius (22) */
void ora_camera_setup(const ora_camera *c, int32_t width, int32_t height, double out[22]);
/* ---- post-process passes of the reference's GPU backend, CPU restatement ----
 * internal/engine/gpu/gpu.go:22-47 (acesTonemap), :2309-2350 (tone-map + gamma + round), :2355-2439
 * (3x3 bilateral on the 8-bit image), :2444-2520 (box blur blended with the original). */
typedef struct {
    int32_t tonemap;        /* 1: rgba = round(sqrt(aces(float32(accum/spp))) * 255), needs accum */
    int32_t denoise;        /* 1: bilateral 3x3, skipped unless w > 2 && h > 2 (gpu.go:2358) */
    double sigma_s, sigma_r;
    int32_t smooth;         /* 1: box blur, radius clamped to 1..5, strength to 0..1 */
    int32_t smooth_radius;
    double smooth_strength;
} ora_post_config;
void ora_post_process(const ora_post_config *cfg, const double *accum, int32_t spp, uint8_t *rgba, int32_t stride,
                      int32_t width, int32_t height);
float ora_aces_tonemap(float x);

/* pixel finish (renderer.go:190-221): raw sum -> 3 bytes */
void ora_finish_pixel(const double sum[3], int32_t spp, uint8_t out[3]);

#ifdef __cplusplus
}
#endif
#endif
