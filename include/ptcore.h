/*
 * ptcore.h -- C ABI of libptcore.so, the MI355X (gfx950) path-tracing core.
 *
 * Drop-in boundary for ONE path of MarkJulian19/path_trace_golang: the per-pixel
 * Monte-Carlo render loop of internal/engine.  The entry points are what a Go
 * backend package binds through cgo in place of the reference's OpenGL backend
 *
 *     gpu.Render(sc *scene.Scene, cfg gpu.RenderConfig, img *image.RGBA,
 *                progress func()) error          internal/engine/gpu/gpu.go:2534
 *
 * which engine.RenderInto dispatches to (internal/engine/renderer.go:34-41,
 * :250-263).  Plain pointers and sizes only; no C++ or torch types.  The cgo
 * binding is shown in INTEGRATION.md; a C++ mirror of the Go host layer lives in
 * path_trace_golang_amd/csrc/host/ for machines without a Go toolchain.
 *
 * Semantics are those of the reference CPU engine (renderIntoCPU,
 * renderer.go:44-246), not of its GLSL backend: FP64 arithmetic, linear object
 * scan, sqrt gamma, uint8(v*255.999).  The one deliberate difference is the RNG:
 * the reference seeds math/rand from the clock (random.go:14-16); here each
 * (seed, pixel, sample) owns a counter-based stream, so a render is reproducible
 * and independent of device count.
 *
 * Threading: a pt_ctx is not re-entrant (the reference serialises GPU renders
 * through one goroutine too, gpu.go:250-297, :2534-2546).  Different contexts
 * may be used from different threads.  No pointer passed in is retained after a
 * call returns (cgo pointer rule).
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 3

/* status codes; pt_last_error() holds the text (thread-local) */
enum {
    PT_OK = 0,
    PT_ERR_INVALID = 1,   /* bad argument / inconsistent scene */
    PT_ERR_NO_DEVICE = 2, /* no usable HIP device */
    PT_ERR_HIP = 3,       /* HIP runtime error */
    PT_ERR_NOMEM = 4,
    PT_ERR_STATE = 5      /* call out of sequence (pt_step without pt_begin, ...) */
};

/* scene.MaterialType strings (internal/scene/scene.go:33-40) in materialType
 * order (internal/engine/materials.go:11-17).  Any other string is PT_MAT_LAMBERT:
 * convertMaterial's default branch (materials.go:51-53). */
enum { PT_MAT_LAMBERT = 0, PT_MAT_METAL = 1, PT_MAT_DIELECTRIC = 2, PT_MAT_EMISSIVE = 3, PT_MAT_MIRROR = 4 };

/* scene.ObjectType strings (scene.go:66-73).  Any other string is PT_OBJ_UNKNOWN
 * and is skipped like sceneToWorld does (internal/engine/objects.go:237-266). */
enum { PT_OBJ_UNKNOWN = -1, PT_OBJ_SPHERE = 0, PT_OBJ_PLANE = 1, PT_OBJ_BOX = 2, PT_OBJ_SPHERE_LIGHT = 3 };

/* Sky selection of the closure at renderer.go:56-92. */
enum {
    PT_SKY_BACKGROUND = 0, /* scene.Sky == nil, or Sky.Type not "gradient"/"solid": scene.Background */
    PT_SKY_GRADIENT = 1,
    PT_SKY_SOLID = 2
};

/* scene.Material (scene.go:41-63): the fields convertMaterial reads
 * (materials.go:28-55).  Reflectivity, Tint, AbsorptionScale are ignored by the
 * CPU engine and do not cross the boundary. */
typedef struct pt_material {
    int32_t type; /* PT_MAT_* */
    int32_t reserved;
    double albedo[3];
    double rough;
    double ior;
    double emit[3];
    double power;
    double absorption[3];
    double smoothness;
} pt_material;

/* scene.Object (scene.go:76-84). `material` is the index into pt_scene.materials
 * of the material whose ID equals Object.MaterialID (the LAST one if IDs repeat:
 * the map assignment at objects.go:227-229), or -1 when no material has that ID
 * (zero material = black lambert, objects.go:233). */
typedef struct pt_object {
    int32_t type; /* PT_OBJ_* */
    int32_t material;
    double position[3];
    double size[3]; /* sphere radius = size[0]; box full extents; plane ignores it */
} pt_object;

/* scene.Camera (scene.go:24-32) */
typedef struct pt_camera {
    double position[3];
    double target[3];
    double up[3];
    double fov;
    double aperture;
    double focus_dist;
    double aspect_ratio;
} pt_camera;

/* scene.Sky (scene.go:135-140) + scene.Background (scene.go:150) */
typedef struct pt_sky {
    int32_t kind; /* PT_SKY_* */
    int32_t reserved;
    double background[3];
    double color[3];
    double horizon[3];
    double zenith[3];
} pt_sky;

typedef struct pt_scene {
    pt_camera camera;
    pt_sky sky;
    int32_t num_materials;
    int32_t num_objects;
    const pt_material *materials;
    const pt_object *objects;
} pt_scene;

/* engine.RenderConfig (renderer.go:17-22) + the stream seed. */
typedef struct pt_config {
    int32_t width;
    int32_t height;
    int32_t samples_per_px;
    int32_t max_depth;
    uint64_t seed;
    int32_t spp_chunk; /* samples per pixel per device pass; 0 = choose from the buffer budget */
    int32_t flags;     /* PT_FLAG_* */
} pt_config;

enum {
    PT_FLAG_NONE = 0,
    PT_FLAG_PIXEL_STATS = 1 /* also produce per-pixel segment / RNG-draw counts (slower; parity debugging) */
};

/* Which 32x32 tiles (renderer.go:132, row-major tile index t = ty*ntx + tx) this
 * call renders: t = index, index+count, index+2*count, ...  {0,1} = whole frame. */
typedef struct pt_shard {
    int32_t index;
    int32_t count;
} pt_shard;

typedef struct pt_stats {
    uint64_t samples;    /* primary samples traced */
    uint64_t segments;   /* closest-hit scans = rayColorOpt activations with depth > 0 (renderer.go:286-302) */
    uint64_t exit_scans; /* dielectric exit searches (renderer.go:316-371) */
    uint64_t draws;      /* RNG draws */
    double seconds;      /* host wall time of the call (upload + kernels + download) */
    double trace_ms;     /* device time inside the trace kernel(s) alone, HIP events around each launch */
    double resolve_ms;   /* device time inside the resolve kernel(s) */
    double device_ms;    /* device time first launch -> last launch complete */
    int32_t trace_launches;
    int32_t resolve_launches;
    int32_t spp_chunk;   /* chunk actually used */
    int32_t num_devices;
    double per_device_ms[8];
    double raygen_ms;    /* device time inside the ray-generation kernel(s) */
    /* split passes (ABI 2): dielectric hits leave the trace kernel through a path-state queue and are shaded by glass_kernel */
    double glass_ms;       /* device time inside glass_kernel */
    double trace_split_ms; /* the part of trace_ms spent in the split form of the trace kernel (the dominant launches) */
    int32_t glass_launches;
    int32_t trace_split_launches;
    uint64_t glass_events;    /* paths parked in the glass queue by the split trace passes (= dielectric closest hits there) */
    uint64_t continuations;   /* paths glass_kernel handed back through the continuation queue */
    uint64_t split_cont_in;   /* continuation entries taken up by split trace passes (the rest finish in the all-in-one pass) */
    uint64_t split_finished;  /* paths that ended inside a split trace pass */
    /* ABI 3: the shader clock the trace kernels actually ran at: one wave per launch reads the shader-cycle counter
     * (s_memtime) and the 100 MHz reference counter (s_memrealtime) when it starts and when it retires; the ratio of
     * the sums over the frame's trace launches (0 when no trace launch was observed) */
    double shader_clock_mhz;
} pt_stats;

typedef struct pt_ctx pt_ctx;

int32_t pt_abi_version(void);

/* Text of the last failure on the calling thread ("" if none). Never NULL. */
const char *pt_last_error(void);

/* Number of visible HIP devices (0 and PT_ERR_NO_DEVICE if none). */
int32_t pt_device_count(int32_t *count);

/* Creates a context on `ndev` devices (`devices` = HIP ordinals; NULL = 0..ndev-1).
 * Replaces the lazily created GL worker of gpu.go:266-297.  Like the reference,
 * an init failure is an error return, never an abort. */
int32_t pt_create(const int32_t *devices, int32_t ndev, pt_ctx **out);
void pt_destroy(pt_ctx *ctx);

/*
 * Blocking whole-frame render into caller memory: the body of gpu.Render.
 *   rgba   : `height` rows of `stride` bytes (image.RGBA.Pix / .Stride), row 0 = top,
 *            A = 255, written completely before return (renderer.go:106-112, :218-221)
 *   accum  : optional width*height*3 doubles, per pixel the raw sum over samples of
 *            the sample radiance (before the 1/spp scale of renderer.go:190-192)
 *   nseg / ndraw : optional width*height uint32 (need PT_FLAG_PIXEL_STATS)
 * With several devices in the context the frame is split over interleaved 32x32
 * tiles and gathered on devices[0]; pixels do not depend on the device count.
 */
int32_t pt_render(pt_ctx *ctx, const pt_scene *scene, const pt_config *cfg, uint8_t *rgba, int32_t stride,
                  double *accum, uint32_t *nseg, uint32_t *ndraw, pt_stats *stats);

/*
 * Progressive form (the interactive contract of gpu.go:2209-2290: preview
 * refreshes while samples accumulate).  pt_begin uploads the scene; each pt_step
 * adds up to `nspp` samples per pixel and returns the total done so far; pt_read
 * resolves the current estimate (normalised by the samples done) without ending
 * the render; pt_end releases the frame and reports totals.  The Go wrapper calls
 * progress() between steps, so no C -> Go callback is needed.
 */
int32_t pt_begin(pt_ctx *ctx, const pt_scene *scene, const pt_config *cfg);
int32_t pt_step(pt_ctx *ctx, int32_t nspp, int32_t *done_spp);
int32_t pt_read(pt_ctx *ctx, uint8_t *rgba, int32_t stride, double *accum);
int32_t pt_end(pt_ctx *ctx, pt_stats *stats);

/*
 * Device-resident form for one-process-per-GPU hosts (bench.py, torch.distributed):
 * renders the tiles of `shard` on the context's first device, asynchronously on
 * `stream` (a hipStream_t; NULL = the context's own stream) and leaves
 *   d_tiles_rgba  : [ntiles_local][32][32][4] uint8 (device), tile-major
 *   d_tiles_accum : optional [ntiles_local][32][32][3] double (device)
 * for the caller to gather.  Pixels outside the frame in edge tiles are zero.
 * pt_shard_tiles() gives ntiles_local.  Stats are filled after the stream drains
 * only if `stats` is non-NULL (that makes the call blocking).
 */
int32_t pt_shard_tiles(int32_t width, int32_t height, const pt_shard *shard, int32_t *ntiles_local,
                       int32_t *ntiles_x, int32_t *ntiles_y);
int32_t pt_render_tiles_device(pt_ctx *ctx, const pt_scene *scene, const pt_config *cfg, const pt_shard *shard,
                               void *d_tiles_rgba, void *d_tiles_accum, void *stream, pt_stats *stats);

/* Scatters gathered tile buffers into a row-major frame on the device: d_rgba =
 * height rows of `stride` bytes (stride % 4 == 0); d_accum optional width*height*3
 * doubles.  The gathered buffer holds shard 0's tiles, then shard 1's, ...; shard k
 * starts at tile k*shard_stride_tiles, or right after shard k-1 when
 * shard_stride_tiles == 0 (compact).  A fixed stride is what an equal-sized
 * collective gather (ncclGather / torch.distributed.gather) produces. */
int32_t pt_untile_device(pt_ctx *ctx, int32_t width, int32_t height, int32_t shard_count, int32_t shard_stride_tiles,
                         const void *d_tiles_rgba, const void *d_tiles_accum, void *d_rgba, int32_t stride,
                         void *d_accum, void *stream);

/*
 * Optional post-process passes of the reference's OpenGL backend, applied after a render:
 *   tonemap : rgba = uint8(sqrt(aces(float32(accum/spp)))*255 + 0.5)   acesTonemap internal/engine/gpu/gpu.go:22-47,
 *             the loop at gpu.go:2309-2350 -- replaces the CPU engine's finish; needs `accum`
 *   denoise : 3x3 bilateral filter on the 8-bit image, gpu.go:2355-2439 (defaults sigma_s 1.0, sigma_r 0.15,
 *             gpu.go:77-95; skipped unless width > 2 && height > 2)
 *   smooth  : box blur of radius 1..5 blended by strength 0..1, gpu.go:2444-2520 (defaults 2 and 0.5, gpu.go:140-175)
 * None of them is part of the CPU engine's image; they exist so that a user of the reference's -gpu look can have
 * it.  rgba is `height` rows of `stride` bytes in host memory, updated in place; accum (width*height*3 doubles, the
 * raw sums pt_render returns) may be NULL when tonemap is 0.
 */
typedef struct pt_post_config {
    int32_t tonemap;
    int32_t denoise;
    double sigma_s;
    double sigma_r;
    int32_t smooth;
    int32_t smooth_radius;
    double smooth_strength;
} pt_post_config;
int32_t pt_post_process(pt_ctx *ctx, const pt_post_config *post, const double *accum, int32_t samples_per_px, uint8_t *rgba,
                        int32_t stride, int32_t width, int32_t height);

/*
 * Diagnostics only (not part of the rendering boundary): with PTCORE_PROFILE=1 in the
 * environment at pt_create, the trace kernel runs a build that counts, per code
 * section, wave executions, active lanes and shader-clock cycles.  Copies up to n
 * values of the last collected frame ([section][executions, lanes, cycles]) and
 * returns the number available (negative status is impossible: errors return PT_ERR_*
 * as positive codes <= 5, counts are >= 6).
 */
int32_t pt_debug_profile(pt_ctx *ctx, uint64_t *out, int32_t n);

/* Diagnostics only: with PTCORE_SCAN=verify at pt_create the trace kernel runs BOTH closest-hit
 * strategies (the FP32-culled one and the plain every-object FP64 scan) for every segment, renders
 * with the plain one and counts the segments on which the two disagree.  Returns that count,
 * accumulated over every frame finished in this process (0 = the culling never changed a result). */
int64_t pt_debug_scan_mismatches(pt_ctx *ctx);

/* Diagnostics only: the narrow phase and Russian roulette divide several numerators by one denominator and share the
 * reciprocal refinement of the IEEE division between them (div_shared in csrc/pt_kernels.h).  Compares that form with
 * the plain division, bit for bit, on `millions` x 10^6 random and patterned operand pairs on the GPU and returns the
 * number of pairs that differ (must be 0), or -PT_ERR_* on failure. */
int64_t pt_debug_div_selftest(pt_ctx *ctx, int32_t millions, uint64_t seed);

/* Diagnostics only, no GPU needed: builds the bounding-volume hierarchy used for scenes with more
 * than 128 spheres or 128 boxes and checks its invariants.  out = {nodes, objects in the tree, depth, most slots
 * used by a node (<= 4), objects or nodes not reached exactly once, objects not inside their slot's box,
 * child boxes not inside the parent's box, planes kept outside the tree}. */
int32_t pt_debug_bvh_check(const pt_scene *scene, int32_t out[8]);

/* Diagnostics only: how a context that holds several devices collects the tiles of a frame on devices[0]: 0 = one
 * hipMemcpyPeerAsync per peer (the default), 1 = grouped ncclSend / ncclRecv through librccl.so (PTCORE_GATHER=rccl at
 * pt_create; the library is dlopen'ed then and only then). */
int32_t pt_debug_gather_mode(pt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PTCORE_H */
