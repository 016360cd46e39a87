// ptcore.hip -- libptcore.so: C ABI (include/ptcore.h) over the gfx950 kernels.
//
// Host responsibilities, all restating the set-up half of the reference CPU engine:
//   * convertMaterial      internal/engine/materials.go:28-55
//   * sceneToWorld         internal/engine/objects.go:225-269
//   * newCamera            internal/engine/camera.go:19-58
//   * sky closure select   internal/engine/renderer.go:56-92
//   * frame constants      internal/engine/renderer.go:95-98
//   * 32x32 tile grid      internal/engine/renderer.go:132-157 (here: the multi-GPU shard unit)
// then per chunk of samples: reset queue -> trace_kernel -> resolve_kernel.
// There is no CPU rendering path in this library: without a HIP device every entry
// point fails with PT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl.so is loaded on request (PTCORE_GATHER=rccl), never linked

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptcore.h"
#include "pt_bvh.h"
#include "pt_device.h"
#include "pt_kernels.h"
#include "pt_wavefront.h"
#include "pt_walk32.h"
#include "pt_primary.h"
#include "pt_math.h"

using namespace ptd;

namespace {

thread_local std::string g_last_error;
unsigned long long g_profile_scratch[3 * ptk::SEC_COUNT] = {};
unsigned long long g_mismatches = 0;
unsigned long long g_mismatch_sample[10] = {};  // SCAN_VERIFY disagreements of the last collected frame

int32_t fail(int32_t code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(PT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
};

// One device's share of a frame.
struct Device {
    int ordinal = 0;
    int num_cu = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // own_stream or the caller's
    DevBuf<DevObj> objs;
    DevBuf<DevMat> mats;
    DevBuf<BroadSphere> bsph;
    DevBuf<BroadBox> bbox;
    DevBuf<int32_t> plane_idx;
    size_t bytes_cap = 0;             // bytes of per-pass buffers (jobs + path-state queues) that fitted after an allocation failed (0: never failed)
    DevBuf<BvhNode> bvh_nodes;
    DevBuf<BvhObj> bvh_objs;
    DevBuf<BvhNode> bvh_cores;
    DevBuf<double> L;
    DevBuf<double> ray;
    DevBuf<unsigned long long> ray_rng;
    DevBuf<uint16_t> ray_ndraw;
    DevBuf<uint32_t> job_seg, job_draw;
    DevBuf<double> acc;
    DevBuf<uint32_t> acc_seg, acc_draw;
    DevBuf<uint8_t> tiles_rgba;
    DevBuf<double> tiles_accum;
    DevBuf<uint32_t> tiles_seg, tiles_draw;
    DevBuf<unsigned int> queue;       // [0] item cursor of the running trace pass, [1] glass count, [2],[3] continuation counts (ping-pong), [4] always 0
    DevBuf<BroadSphere> bsph_diel;    // broad-phase records of the dielectric objects only
    DevBuf<BroadBox> bbox_diel;
    // path-state queues of the split passes (one entry per job of a chunk at most)
    DevBuf<double> gq_d, cq_d;
    DevBuf<unsigned long long> gq_rs, cq_rs;
    DevBuf<uint32_t> gq_u32, cq_u32;  // job, depth, best, hit, jseg, jdraw planes
    DevBuf<double> xq_d;              // third queue of the wavefront form (gq = exit queue, cq / xq = the two path queues)
    DevBuf<unsigned long long> xq_rs;
    DevBuf<uint32_t> xq_u32;
    int blocks_per_cu_wf = 0;
    int blocks_per_cu_walk = 0;        // wf_walk32_kernel (PTCORE_PIPELINE=walk32)
    DevBuf<uint32_t> cand_ids, cand_n, slow_list;  // walk32: candidate lists of the running level, entries left to the FP64 traversal
    DevBuf<uint32_t> wf_perm, wf_key, wf_bins;  // ray sorting of the wavefront form
    size_t q_cap = 0;
    DevBuf<unsigned long long> counters;
    DevBuf<unsigned long long> prof;
    std::vector<EventPair> ev_trace, ev_resolve, ev_raygen, ev_glass;
    size_t n_trace = 0, n_resolve = 0, n_raygen = 0, n_glass = 0;
    std::vector<char> trace_is_split;  // per trace launch of the frame: the split form?
    hipEvent_t ev_first = nullptr, ev_last = nullptr;
    bool first_recorded = false;
    unsigned long long pass_log_prev[24] = {};  // PTCORE_DEBUG_PASS_LOG: the counters after the previous pass of this frame
    // frame state
    pt_shard shard{0, 1};
    int32_t nlocal = 0;
    uint32_t nslots = 0;
    bool acc_started = false;
    int blocks_per_cu = 0, blocks_per_cu_split = 0, blocks_per_cu_glass = 0, blocks_per_cu_primary = 0;
    uint64_t scene_gen = 0;  // SceneData generation resident on this device (0 = none)
};

struct Frame {
    bool open = false;
    pt_config cfg{};
    DevFrame F{};
    DevCamera cam{};
    DevSky sky{};
    int32_t nobj = 0, nmat = 0;
    int32_t ntx = 0, nty = 0;
    int32_t done_spp = 0;
    uint32_t chunk = 0;
    bool stats_on = false;
    size_t lds_bytes = 0;
    size_t glass_lds_bytes = 0;
    bool wavefront = false;  // the wavefront form (pt_wavefront.h) instead of the all-in-one loop
    bool walk32 = false;     // ... with its traversal pass split into the FP32 walk and the exact pass of pt_walk32.h
    size_t shade_lds_bytes = 0;
    int split_rounds = 0;  // trace + glass pass pairs before the all-in-one pass (0: all-in-one only)
    size_t budget_bytes = 0;  // job-buffer budget of this frame (pt_ctx::l_budget_bytes, or the grown one)
    int tail_form = 0;     // ptk::FORM_* of the pass behind the split rounds (FORM_NESTED for the bitmask scans, see pt_kernels.h)
    bool has_glass = false;  // some object is dielectric
    bool primary_pass = false;  // BVH scans: the first segment of every path by primary_bvh_kernel (pt_primary.h), the rest through the continuation queue
    int scan = 0;  // ptk::SCAN_* used for this frame
    std::chrono::steady_clock::time_point t0;
};

// Everything derived from the scene alone (world, broad-phase records, hierarchies).  Kept across
// frames: an unchanged scene (progressive previews, benchmark loops, camera-only edits do not count:
// the camera is not part of it) is neither rebuilt nor uploaded again.
struct SceneData {
    bool valid = false;
    int scan_req = -2;
    uint64_t gen = 0;
    std::vector<DevObj> world;
    std::vector<DevMat> mats;
    std::vector<BroadSphere> bsph;
    std::vector<BroadBox> bbox;
    std::vector<BroadSphere> bsph_diel;
    std::vector<BroadBox> bbox_diel;
    std::vector<pt_material> raw_mats;  // the caller's arrays as last seen (change detection without converting)
    std::vector<pt_object> raw_objs;
    std::vector<int32_t> plane_idx;
    std::vector<BvhNode> bvh_nodes;
    std::vector<BvhObj> bvh_objs;
    std::vector<BvhNode> bvh_cores;   // core twins of bvh_nodes (the FP32 walk's certain bounds)
    int bvh_depth = 0;
    int bvh_stack_need = 0;
    bool has_glass = false;           // some object is dielectric
    size_t lds_bytes = 0;
    size_t glass_lds_bytes = 0;
    int scan = 0;
    DevFrame Fs{};  // the scene-dependent fields of DevFrame
};

}  // namespace

struct pt_ctx {
    std::vector<Device> devs;
    Frame frame;
    SceneData sd;
    // device-0 gather / frame buffers for the host-memory entry points
    DevBuf<uint8_t> g_tiles_rgba;
    DevBuf<double> g_tiles_accum;
    DevBuf<uint32_t> g_tiles_seg, g_tiles_draw;
    DevBuf<uint8_t> f_rgba;
    DevBuf<double> f_accum;
    DevBuf<uint32_t> f_seg, f_draw;
    size_t l_budget_bytes = (size_t)48 << 30;  // per-chunk job buffers (radiance, primary rays, path-state queues): a sixth of the 288 GB
    // A context that renders frame after frame of one shape (a UI, an animation: gpu.go:2534-2546 is called once per frame) grows its
    // job buffers by itself from the second such frame on, to the size bench.py asks for explicitly (4 instead of 13 passes per
    // 1080p x 1024-spp frame: -4.5 % time per frame), provided the device has that much free: a one-shot render keeps the small set-up.
    // Off when PTCORE_L_BUDGET_MB names a size, or with PTCORE_AUTO_GROW=0.
    size_t grown_budget_bytes = (size_t)160 << 30;
    bool auto_grow = true;
    int32_t last_w = 0, last_h = 0, last_spp = 0;  // the shape of the previous frame of this context
    bool grown = false;
    int pipeline = -1;     // PTCORE_PIPELINE=mega|wavefront|walk32 (default: by scene, see frame_open)
    int wf_min_lanes = 40; // PTCORE_WF_MIN_LANES: the walk loop of a traversal pass is left for a refill below this many walking lanes
    int wf_sort = 0;       // PTCORE_WF_SORT=1: reorder the paths of a level by direction octant and origin cell
    int split_rounds = 2;  // PTCORE_SPLIT_ROUNDS: trace + glass pass pairs per chunk before the all-in-one pass (bitmask scan only)
    bool primary_coop = true;  // PTCORE_PRIMARY=lane: BVH scenes without the wave-cooperative primary pass (the round-3 loop)
    bool tail_nested = true;  // PTCORE_TAIL=trip: the pass behind the split rounds in the round-1 form (exit search = the lane's next trip) instead of FORM_NESTED
    // PTCORE_GATHER=rccl: the tiles of a frame reach devices[0] through RCCL (grouped ncclSend / ncclRecv over one communicator per
    // device, ncclCommInitAll) instead of hipMemcpyPeerAsync.  librccl.so is dlopen'ed then and only then.
    struct Rccl {
        void *lib = nullptr;
        std::vector<ncclComm_t> comms;
        decltype(&ncclCommInitAll) CommInitAll = nullptr;
        decltype(&ncclCommDestroy) CommDestroy = nullptr;
        decltype(&ncclGroupStart) GroupStart = nullptr;
        decltype(&ncclGroupEnd) GroupEnd = nullptr;
        decltype(&ncclSend) Send = nullptr;
        decltype(&ncclRecv) Recv = nullptr;
        decltype(&ncclGetErrorString) GetErrorString = nullptr;
        uint64_t gathers = 0;  // frames gathered through it
    } rccl;
    uint32_t claim = 0;   // jobs per queue claim; 0 = by pass shape (dev_step), PTCORE_CLAIM forces one
    int max_blocks_per_cu = 8;
    int scan_mode = -1;  // -1 = choose by scene size; PTCORE_SCAN=uniform|broad|verify|bvh|verify_bvh forces one
    unsigned long long last_mismatches = 0;
    bool profile_sections = false;  // PTCORE_PROFILE=1: diagnostic kernel build with per-section counters
    unsigned long long last_profile[3 * ptk::SEC_COUNT] = {};
};

namespace {

// ---------------------------------------------------------------- scene conversion

double clampd(double x, double lo, double hi) {  // materials.go:57-65
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

DevMat convert_material(const pt_material &m) {  // materials.go:28-55
    DevMat r;
    std::memset(&r, 0, sizeof r);
    switch (m.type) {
        case PT_MAT_METAL: {
            double rough = m.rough;
            if (m.smoothness > 0) rough = 1.0 - clampd(m.smoothness, 0, 1);
            r.typ = MAT_METAL;
            for (int i = 0; i < 3; i++) r.albedo[i] = m.albedo[i];
            r.rough = clampd(rough, 0, 1);
            break;
        }
        case PT_MAT_DIELECTRIC: {
            double ior = m.ior;
            if (ior == 0) ior = 1.5;
            r.typ = MAT_DIELECTRIC;
            for (int i = 0; i < 3; i++) { r.albedo[i] = m.albedo[i]; r.absorption[i] = m.absorption[i]; }
            r.ior = ior;
            // reflectance's r0 for both faces (materials.go:183, :226-229): the reference recomputes these on every hit
            // from the same operands; done once here with the same IEEE operations (this file is built with
            // -ffp-contract=off like the kernels)
            r.inv_ior = 1.0 / ior;
            {
                double a = (1 - r.inv_ior) / (1 + r.inv_ior);
                r.r0_front = a * a;
                double b = (1 - ior) / (1 + ior);
                r.r0_back = b * b;
            }
            break;
        }
        case PT_MAT_EMISSIVE:
            r.typ = MAT_EMISSIVE;
            for (int i = 0; i < 3; i++) r.emit[i] = m.emit[i] * m.power;
            break;
        case PT_MAT_MIRROR:
            r.typ = MAT_MIRROR;
            for (int i = 0; i < 3; i++) r.albedo[i] = m.albedo[i];
            break;
        default:
            r.typ = MAT_LAMBERT;
            for (int i = 0; i < 3; i++) r.albedo[i] = m.albedo[i];
            r.rough = clampd(m.rough, 0, 1);
            break;
    }
    r.absorbs = (r.absorption[0] > 0 || r.absorption[1] > 0 || r.absorption[2] > 0) ? 1 : 0;
    r.rough_sq = r.rough * r.rough;
    return r;
}

// objects.go:225-269; materials are converted once and indexed (the zero material
// of a missing id is slot num_materials)
void scene_to_world(const pt_scene &sc, std::vector<DevObj> &world, std::vector<DevMat> &mats) {
    mats.clear();
    for (int i = 0; i < sc.num_materials; i++) mats.push_back(convert_material(sc.materials[i]));
    DevMat zero;
    std::memset(&zero, 0, sizeof zero);
    mats.push_back(zero);
    world.clear();
    for (int i = 0; i < sc.num_objects; i++) {
        const pt_object &o = sc.objects[i];
        DevObj d;
        std::memset(&d, 0, sizeof d);
        d.mat = (o.material >= 0 && o.material < sc.num_materials) ? o.material : sc.num_materials;
        int kind;
        switch (o.type) {
            case PT_OBJ_SPHERE:
            case PT_OBJ_SPHERE_LIGHT:
                kind = KIND_SPHERE;
                for (int k = 0; k < 3; k++) d.a[k] = o.position[k];
                d.radius = o.size[0];
                d.radius_sq = d.radius * d.radius;
                d.inv_radius = 1.0 / d.radius;
                break;
            case PT_OBJ_PLANE:
                kind = KIND_PLANE;
                for (int k = 0; k < 3; k++) d.a[k] = o.position[k];
                d.b[0] = 0; d.b[1] = 1; d.b[2] = 0;
                break;
            case PT_OBJ_BOX:
                kind = KIND_BOX;
                for (int k = 0; k < 3; k++) {
                    d.a[k] = o.position[k] - o.size[k] * 0.5;
                    d.b[k] = o.position[k] + o.size[k] * 0.5;
                }
                break;
            default:
                continue;  // unknown types are skipped
        }
        d.kind = kind | (mats[(size_t)d.mat].typ == MAT_DIELECTRIC ? 0x100 : 0);
        world.push_back(d);
    }
}

struct H3 {
    double x, y, z;
};
H3 h3(const double *p) { return H3{p[0], p[1], p[2]}; }
H3 sub(H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; }
H3 mul(H3 a, double t) { return H3{a.x * t, a.y * t, a.z * t}; }
H3 divs(H3 a, double t) { double inv = 1.0 / t; return H3{a.x * inv, a.y * inv, a.z * inv}; }
double dot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
H3 cross(H3 a, H3 b) { return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
double length(H3 a) { return ptm::f_sqrt(dot(a, a)); }
H3 unit(H3 a) {
    double l = length(a);
    if (l == 0) return a;
    return divs(a, l);
}
void put(double *d, H3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

DevCamera new_camera(const pt_camera &c, int width, int height) {  // camera.go:19-58
    DevCamera cam;
    double aspect = (double)width / (double)height;
    if (c.aspect_ratio != 0) aspect = c.aspect_ratio;
    const double theta = c.fov * 3.141592653589793 / 180;
    const double h = ptm::go_tan(theta / 2);
    const double viewportHeight = 2.0 * h;
    const double viewportWidth = aspect * viewportHeight;
    const H3 origin = h3(c.position), target = h3(c.target), up = h3(c.up);
    const H3 w = unit(sub(origin, target));
    const H3 u = unit(cross(up, w));
    const H3 v = cross(w, u);
    double focusDist = c.focus_dist;
    if (focusDist == 0) focusDist = length(sub(origin, target));
    const H3 horizontal = mul(u, viewportWidth * focusDist);
    const H3 vertical = mul(v, viewportHeight * focusDist);
    const H3 llc = sub(sub(sub(origin, divs(horizontal, 2)), divs(vertical, 2)), mul(w, focusDist));
    put(cam.origin, origin);
    put(cam.lower_left, llc);
    put(cam.horizontal, horizontal);
    put(cam.vertical, vertical);
    put(cam.u, u);
    put(cam.v, v);
    cam.lens_radius = c.aperture / 2;
    return cam;
}

DevSky make_sky(const pt_sky &s) {  // renderer.go:56-92
    DevSky d;
    std::memset(&d, 0, sizeof d);
    d.kind = s.kind == PT_SKY_GRADIENT ? 1 : (s.kind == PT_SKY_SOLID ? 2 : 0);
    const double *c0 = d.kind == 1 ? s.horizon : (d.kind == 2 ? s.color : s.background);
    for (int i = 0; i < 3; i++) { d.c0[i] = c0[i]; d.c1[i] = s.zenith[i]; }
    return d;
}

float round_up_f(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, INFINITY);
    return f;
}
float round_down_f(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -INFINITY);
    return f;
}

// Conservative FP32 bounds for the broad phase.  B bounds every finite object's coordinates;
// everything is inflated by m = B * 2^-12 (two orders of magnitude above the worst FP32 rounding of
// the broad-phase arithmetic for ray origins inside [-4B, 4B]^3), and rounded outward.
void build_broad(const std::vector<DevObj> &world, SceneData &fr) {
    DevFrame &F = fr.Fs;
    fr.bsph.clear();
    fr.bbox.clear();
    fr.bsph_diel.clear();
    fr.bbox_diel.clear();
    fr.plane_idx.clear();
    F.sph_all = F.box_all = F.sph_diel = F.box_diel = 0;
    double B = 1.0;
    for (const DevObj &o : world) {
        const int kind = o.kind & 0xff;
        if (kind == KIND_SPHERE) {
            for (int k = 0; k < 3; k++) B = std::max(B, std::fabs(o.a[k]) + std::fabs(o.radius));
        } else if (kind == KIND_BOX) {
            for (int k = 0; k < 3; k++) B = std::max(B, std::max(std::fabs(o.a[k]), std::fabs(o.b[k])));
        }
    }
    if (!(B < 1e30)) B = INFINITY;  // absurd or non-finite geometry: every object stays a candidate
    const double m = B * (1.0 / 4096.0);
    for (size_t i = 0; i < world.size(); i++) {
        const DevObj &o = world[i];
        const int kind = o.kind & 0xff;
        if (kind == KIND_SPHERE) {
            BroadSphere s;
            std::memset(&s, 0, sizeof s);
            s.cx = (float)o.a[0]; s.cy = (float)o.a[1]; s.cz = (float)o.a[2];
            const double rm = std::fabs(o.radius) + m;
            s.rm2 = round_up_f(rm * rm);
            if (!(s.rm2 == s.rm2)) s.rm2 = INFINITY;
            s.index = (int32_t)i;
            s.diel = (o.kind & 0x100) ? 1 : 0;
            fr.bsph.push_back(s);
            if (o.kind & 0x100) fr.bsph_diel.push_back(s);
        } else if (kind == KIND_BOX) {
            BroadBox b;
            std::memset(&b, 0, sizeof b);
            for (int k = 0; k < 3; k++) {
                const double lo = std::min(o.a[k], o.b[k]) - m, hi = std::max(o.a[k], o.b[k]) + m;
                const float c = (float)(0.5 * lo + 0.5 * hi);
                if (lo == lo && hi == hi && std::isfinite(c)) {
                    b.c[k] = c;
                    b.h[k] = round_up_f(std::max((double)c - lo, hi - (double)c));  // [c - h, c + h] holds [lo, hi]
                    if (!(b.h[k] == b.h[k])) b.h[k] = INFINITY;
                } else {  // absurd or non-finite bounds: this slab constrains nothing
                    b.c[k] = 0.0f;
                    b.h[k] = INFINITY;
                }
            }
            b.index = (int32_t)i;
            b.diel = (o.kind & 0x100) ? 1 : 0;
            fr.bbox.push_back(b);
            if (o.kind & 0x100) fr.bbox_diel.push_back(b);
        } else {
            fr.plane_idx.push_back((int32_t)i);
        }
    }
    F.n_bsph = (int32_t)fr.bsph.size();
    F.n_bbox = (int32_t)fr.bbox.size();
    // the dielectric records among the first 32 of each kind, in the bit order of the kernels' candidate masks (push_keep_bit)
    for (int i = 0; i < F.n_bsph && i < 32 && F.n_bsph <= 32; i++)
        if (fr.bsph[(size_t)i].diel) F.sph_diel |= 1u << ptk::pt_record_slot(i, F.n_bsph);
    for (int i = 0; i < F.n_bbox && i < 32 && F.n_bbox <= 32; i++)
        if (fr.bbox[(size_t)i].diel) F.box_diel |= 1u << ptk::pt_record_slot(i, F.n_bbox);
    F.n_plane = (int32_t)fr.plane_idx.size();
    F.planes_y = 1;
    for (int32_t i : fr.plane_idx) {
        const DevObj &o = world[(size_t)i];
        if (!(o.b[0] == 0.0 && o.b[1] == 1.0 && o.b[2] == 0.0) || std::signbit(o.b[0]) || std::signbit(o.b[2])) F.planes_y = 0;
        // plane_exact_y drops the terms (px - ox) * 0 and (pz - oz) * 0 of objects.go:107: they are zeros only for a finite point
        // (inf * 0 = NaN makes the reference's t a NaN, which its range test accepts)
        if (!(std::isfinite(o.a[0]) && std::isfinite(o.a[1]) && std::isfinite(o.a[2]))) F.planes_y = 0;
    }
    F.plane0_index = -1;
    F.plane0_y = 0.0;
    F.plane0_kind = 0;
    if (F.planes_y && fr.plane_idx.size() == 1) {
        const DevObj &o = world[(size_t)fr.plane_idx[0]];
        F.plane0_index = fr.plane_idx[0];
        F.plane0_y = o.a[1];
        F.plane0_kind = o.kind;
    }
    F.n_dsph = (int32_t)fr.bsph_diel.size();
    F.n_dbox = (int32_t)fr.bbox_diel.size();
    F.broad_ok = (fr.bsph.size() <= 32 && fr.bbox.size() <= 32) ? 1 : (fr.bsph.size() <= 128 && fr.bbox.size() <= 128) ? 2 : 0;
    F.sph_all = fr.bsph.size() >= 32 ? 0xffffffffu : ((1u << fr.bsph.size()) - 1u);
    F.box_all = fr.bbox.size() >= 32 ? 0xffffffffu : ((1u << fr.bbox.size()) - 1u);
    F.origin_bound = (float)std::min(4.0 * B, 3.0e38);
    F.scene_bound = B * (1.0 + 1.0 / 512.0);  // the inflation is B/4096
    F.clip_bound = B * 3.5;
    F.margin = m;
}

int32_t tiles_of_shard(int32_t ntiles, const pt_shard &sh) {
    if (sh.index >= ntiles) return 0;
    return (ntiles - sh.index + sh.count - 1) / sh.count;
}

int32_t validate(const pt_scene *scene, const pt_config *cfg) {
    if (!scene || !cfg) return fail(PT_ERR_INVALID, "null scene or config");
    if (cfg->width <= 0 || cfg->height <= 0) return fail(PT_ERR_INVALID, "width and height must be positive");
    if (cfg->samples_per_px < 0) return fail(PT_ERR_INVALID, "samples_per_px must be >= 0");
    if ((int64_t)cfg->width * cfg->height > (int64_t)1 << 28) return fail(PT_ERR_INVALID, "frame too large");
    // pixel slots are 32-bit: 1024 per 32x32 tile, also for the nearly empty tiles of a 1-pixel-wide frame
    if ((int64_t)((cfg->width + 31) / 32) * ((cfg->height + 31) / 32) > (int64_t)1 << 21)
        return fail(PT_ERR_INVALID, "frame too large (more than 2^21 tiles)");
    if (scene->num_materials < 0 || scene->num_objects < 0) return fail(PT_ERR_INVALID, "negative scene counts");
    if (scene->num_materials > 0 && !scene->materials) return fail(PT_ERR_INVALID, "materials is null");
    if (scene->num_objects > 0 && !scene->objects) return fail(PT_ERR_INVALID, "objects is null");
    return PT_OK;
}

// ---------------------------------------------------------------- per-device frame

using TraceFn = void (*)(const TraceArgs);

// The shipping instantiations are <false,false,*,*>; STATS adds per-pixel counters, PROF the section profile.
// form (pt_kernels.h): FORM_SPLIT = dielectric hits leave for the glass queue, FORM_NESTED = the pass behind the split rounds (both: the
// bitmask scans of the reference-sized scenes only), FORM_ALL_IN_ONE = everything else.
TraceFn pick_trace(bool stats, bool prof, int scan, int form = ptk::FORM_ALL_IN_ONE) {
    using namespace ptk;
    if (prof) {
        if (scan == SCAN_UNIFORM) return trace_kernel<false, true, SCAN_UNIFORM, FORM_ALL_IN_ONE>;
        if (scan == SCAN_BVH || scan == SCAN_VERIFY_BVH) return trace_kernel<false, true, SCAN_BVH, FORM_ALL_IN_ONE>;
        if (scan == SCAN_BROAD_WIDE || scan == SCAN_VERIFY_WIDE) return trace_kernel<false, true, SCAN_BROAD_WIDE, FORM_ALL_IN_ONE>;
        return form == FORM_SPLIT ? trace_kernel<false, true, SCAN_BROAD, FORM_SPLIT>
               : form == FORM_NESTED ? trace_kernel<false, true, SCAN_BROAD, FORM_NESTED> : trace_kernel<false, true, SCAN_BROAD, FORM_ALL_IN_ONE>;
    }
#define PT_PICK(SCAN_, FORM_) (stats ? trace_kernel<true, false, SCAN_, FORM_> : trace_kernel<false, false, SCAN_, FORM_>)
    if (form != FORM_ALL_IN_ONE) {  // bitmask scans only (trace_form() never asks otherwise)
        const bool nest = form == FORM_NESTED;
        if (scan == SCAN_VERIFY) return nest ? PT_PICK(SCAN_VERIFY, FORM_NESTED) : PT_PICK(SCAN_VERIFY, FORM_SPLIT);
        if (scan == SCAN_BROAD_WIDE) return nest ? PT_PICK(SCAN_BROAD_WIDE, FORM_NESTED) : PT_PICK(SCAN_BROAD_WIDE, FORM_SPLIT);
        if (scan == SCAN_VERIFY_WIDE) return nest ? PT_PICK(SCAN_VERIFY_WIDE, FORM_NESTED) : PT_PICK(SCAN_VERIFY_WIDE, FORM_SPLIT);
        return nest ? PT_PICK(SCAN_BROAD, FORM_NESTED) : PT_PICK(SCAN_BROAD, FORM_SPLIT);
    }
    switch (scan) {
        case SCAN_BROAD: return PT_PICK(SCAN_BROAD, FORM_ALL_IN_ONE);
        case SCAN_VERIFY: return PT_PICK(SCAN_VERIFY, FORM_ALL_IN_ONE);
        case SCAN_BROAD_WIDE: return PT_PICK(SCAN_BROAD_WIDE, FORM_ALL_IN_ONE);
        case SCAN_VERIFY_WIDE: return PT_PICK(SCAN_VERIFY_WIDE, FORM_ALL_IN_ONE);
        case SCAN_BVH: return PT_PICK(SCAN_BVH, FORM_ALL_IN_ONE);
        case SCAN_VERIFY_BVH: return PT_PICK(SCAN_VERIFY_BVH, FORM_ALL_IN_ONE);
        default: return PT_PICK(SCAN_UNIFORM, FORM_ALL_IN_ONE);
    }
#undef PT_PICK
}

TraceFn pick_primary(bool stats, int scan) {
    using namespace ptk;
    if (scan == SCAN_VERIFY_BVH) return stats ? primary_bvh_kernel<true, true> : primary_bvh_kernel<false, true>;
    return stats ? primary_bvh_kernel<true, false> : primary_bvh_kernel<false, false>;
}

using GlassFn = void (*)(const ptk::GlassArgs);
GlassFn pick_glass(bool stats, int scan) {
    using namespace ptk;
    if (scan == SCAN_VERIFY) return stats ? glass_kernel<true, true, false> : glass_kernel<false, true, false>;
    if (scan == SCAN_BROAD_WIDE) return stats ? glass_kernel<true, false, true> : glass_kernel<false, false, true>;
    if (scan == SCAN_VERIFY_WIDE) return stats ? glass_kernel<true, true, true> : glass_kernel<false, true, true>;
    return stats ? glass_kernel<true, false, false> : glass_kernel<false, false, false>;
}

int32_t dev_events(Device &d, std::vector<EventPair> &v, size_t need) {
    while (v.size() < need) {
        EventPair e;
        HIP_TRY(hipEventCreate(&e.a));
        HIP_TRY(hipEventCreate(&e.b));
        v.push_back(e);
    }
    (void)d;
    return PT_OK;
}

#define PT_GLASS_MAX_BLOCKS_PER_CU 8

size_t walk32_lds_bytes() { return 0; }  // the stacks are static LDS of the kernel

// Slots a path-state queue needs beyond one per job of the pass: every wave of a pass that appends to it reserves slots in
// windows (one atomic per window, see trace_kernel) and may leave its last window partly empty -- fewer than one window per
// wave and pass.  Derived from the very grids the launches use (dev_step, dev_step_wavefront), none of which is wider
// than one block per PT_BLOCK items of the pass:
//   split form      glass queue <- trace_kernel<split> (num_cu x blocks_per_cu_split blocks, windows of PT_QUEUE_BLOCK)
//                   continuation queue <- glass_kernel (num_cu x blocks_per_cu_glass blocks, windows of PT_CONT_BLOCK)
//   wavefront form  path queue <- wf_shade_kernel AND wf_exit_kernel of one level (num_cu x PT_WF_PASS_BLOCKS_PER_CU blocks each)
// every window priced at PT_CONT_BLOCK (static_assert in pt_kernels.h: PT_QUEUE_BLOCK <= PT_CONT_BLOCK).
#define PT_WF_PASS_BLOCKS_PER_CU 4
size_t queue_slack(const pt_ctx *ctx, const Device &d, size_t njobs_max) {
    const Frame &fr = ctx->frame;
    const size_t grid_cap = std::max<size_t>(1, (njobs_max + PT_BLOCK - 1) / PT_BLOCK);
    size_t writer_blocks;
    if (fr.wavefront)
        writer_blocks = 2 * std::min<size_t>((size_t)d.num_cu * PT_WF_PASS_BLOCKS_PER_CU, grid_cap);
    else if (fr.primary_pass)  // continuation queue <- primary_bvh_kernel (num_cu x blocks_per_cu_primary blocks, windows of PT_CONT_BLOCK)
        writer_blocks = std::min<size_t>((size_t)d.num_cu * (size_t)d.blocks_per_cu_primary, grid_cap);
    else
        writer_blocks = std::min<size_t>((size_t)d.num_cu * (size_t)std::max(d.blocks_per_cu_split, d.blocks_per_cu_glass), grid_cap);
    return writer_blocks * (PT_BLOCK / PT_WAVE) * PT_CONT_BLOCK;
}

// bytes of job buffers a device holds right now
size_t dev_held_bytes(const Device &d) {
    return d.L.cap * sizeof(double) + d.ray.cap * sizeof(double) + d.ray_rng.cap * 8 + d.ray_ndraw.cap * 2 + (d.job_seg.cap + d.job_draw.cap) * 4 +
           (d.gq_d.cap + d.cq_d.cap + d.xq_d.cap) * sizeof(double) + (d.gq_rs.cap + d.cq_rs.cap + d.xq_rs.cap) * 8 +
           (d.gq_u32.cap + d.cq_u32.cap + d.xq_u32.cap + d.wf_perm.cap + d.wf_key.cap + d.cand_ids.cap + d.cand_n.cap + d.slow_list.cap) * 4;
}

// the samples per pass were chosen from the buffer budget (not forced by pt_config.spp_chunk)
bool cfg_chunk_free(const Frame &fr) { return fr.cfg.spp_chunk <= 0; }

// equal passes: ceil(spp / chunk) passes of ceil(spp / passes) samples instead of full passes and a short last one
void balance_chunk(Frame &fr) {
    const uint32_t spp = (uint32_t)std::max(1, fr.cfg.samples_per_px);
    if (fr.chunk >= spp) { fr.chunk = spp; return; }
    const uint32_t passes = (spp + fr.chunk - 1) / fr.chunk;
    fr.chunk = (spp + passes - 1) / passes;
}

int32_t dev_begin(pt_ctx *ctx, Device &d, const pt_shard &shard, hipStream_t stream) {
    Frame &fr = ctx->frame;
    const SceneData &sd = ctx->sd;
    const std::vector<DevObj> &world = sd.world;
    const std::vector<DevMat> &mats = sd.mats;
    HIP_TRY(hipSetDevice(d.ordinal));
    d.stream = stream ? stream : d.own_stream;
    d.shard = shard;
    d.nlocal = tiles_of_shard(fr.ntx * fr.nty, shard);
    d.nslots = (uint32_t)d.nlocal * 1024u;
    d.acc_started = false;
    d.n_trace = d.n_resolve = d.n_raygen = d.n_glass = 0;
    d.first_recorded = false;
    std::memset(d.pass_log_prev, 0, sizeof d.pass_log_prev);  // the device counters are cleared below, once per frame
    if (d.scene_gen != sd.gen) {
        HIP_TRY(d.objs.reserve(std::max<size_t>(1, world.size())));
        HIP_TRY(d.mats.reserve(mats.size()));
        if (!world.empty())
            HIP_TRY(hipMemcpyAsync(d.objs.p, world.data(), world.size() * sizeof(DevObj), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(d.mats.p, mats.data(), mats.size() * sizeof(DevMat), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(d.bsph.reserve(std::max<size_t>(1, sd.bsph.size())));
        HIP_TRY(d.bbox.reserve(std::max<size_t>(1, sd.bbox.size())));
        HIP_TRY(d.plane_idx.reserve(std::max<size_t>(1, sd.plane_idx.size())));
        if (!sd.bsph.empty())
            HIP_TRY(hipMemcpyAsync(d.bsph.p, sd.bsph.data(), sd.bsph.size() * sizeof(BroadSphere), hipMemcpyHostToDevice, d.stream));
        if (!sd.bbox.empty())
            HIP_TRY(hipMemcpyAsync(d.bbox.p, sd.bbox.data(), sd.bbox.size() * sizeof(BroadBox), hipMemcpyHostToDevice, d.stream));
        if (!sd.plane_idx.empty())
            HIP_TRY(hipMemcpyAsync(d.plane_idx.p, sd.plane_idx.data(), sd.plane_idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(d.bsph_diel.reserve(std::max<size_t>(1, sd.bsph_diel.size())));
        HIP_TRY(d.bbox_diel.reserve(std::max<size_t>(1, sd.bbox_diel.size())));
        if (!sd.bsph_diel.empty())
            HIP_TRY(hipMemcpyAsync(d.bsph_diel.p, sd.bsph_diel.data(), sd.bsph_diel.size() * sizeof(BroadSphere), hipMemcpyHostToDevice, d.stream));
        if (!sd.bbox_diel.empty())
            HIP_TRY(hipMemcpyAsync(d.bbox_diel.p, sd.bbox_diel.data(), sd.bbox_diel.size() * sizeof(BroadBox), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(d.bvh_nodes.reserve(std::max<size_t>(1, sd.bvh_nodes.size())));
        HIP_TRY(d.bvh_objs.reserve(std::max<size_t>(1, sd.bvh_objs.size())));
        if (!sd.bvh_nodes.empty())
            HIP_TRY(hipMemcpyAsync(d.bvh_nodes.p, sd.bvh_nodes.data(), sd.bvh_nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice, d.stream));
        if (!sd.bvh_objs.empty())
            HIP_TRY(hipMemcpyAsync(d.bvh_objs.p, sd.bvh_objs.data(), sd.bvh_objs.size() * sizeof(BvhObj), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(d.bvh_cores.reserve(std::max<size_t>(1, sd.bvh_cores.size())));
        if (!sd.bvh_cores.empty())
            HIP_TRY(hipMemcpyAsync(d.bvh_cores.p, sd.bvh_cores.data(), sd.bvh_cores.size() * sizeof(BvhNode), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream));
        d.scene_gen = sd.gen;
    }
    HIP_TRY(d.queue.reserve(8));
    HIP_TRY(d.counters.reserve(48));
    HIP_TRY(hipMemsetAsync(d.counters.p, 0, 48 * sizeof(unsigned long long), d.stream));
    if (ctx->profile_sections) {
        HIP_TRY(d.prof.reserve(3 * ptk::SEC_COUNT));
        HIP_TRY(hipMemsetAsync(d.prof.p, 0, 3 * ptk::SEC_COUNT * sizeof(unsigned long long), d.stream));
    }
    const size_t ns = std::max<uint32_t>(1, d.nslots);
    HIP_TRY(d.acc.reserve(3 * ns));
    if (fr.stats_on) {
        HIP_TRY(d.acc_seg.reserve(ns));
        HIP_TRY(d.acc_draw.reserve(ns));
    }
    if (!d.ev_first) {
        HIP_TRY(hipEventCreate(&d.ev_first));
        HIP_TRY(hipEventCreate(&d.ev_last));
    }
    // occupancy of the kernels of this frame for the scene's LDS footprint (before the buffers: the queue slack depends on it)
    const size_t lds = fr.lds_bytes;
    int nb = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pick_trace(fr.stats_on, ctx->profile_sections, fr.scan, fr.tail_form), PT_BLOCK, lds));
    d.blocks_per_cu = std::max(1, std::min(nb, ctx->max_blocks_per_cu));
    d.blocks_per_cu_split = d.blocks_per_cu;
    d.blocks_per_cu_glass = 1;
    if (fr.wavefront) {
        const bool bvh = fr.scan == ptk::SCAN_BVH || fr.scan == ptk::SCAN_VERIFY_BVH;
        if (bvh) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ptk::wf_traverse_kernel<0, false>, PT_BLOCK, lds));
        else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ptk::wf_scan_flat_kernel<0, false>, PT_BLOCK, lds));
        d.blocks_per_cu_wf = std::max(1, std::min(nb, ctx->max_blocks_per_cu));
        if (fr.walk32) {
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ptk::wf_walk32_kernel<0>, PT_BLOCK, walk32_lds_bytes()));
            d.blocks_per_cu_walk = std::max(1, std::min(nb, ctx->max_blocks_per_cu));
            if (std::getenv("PTCORE_VERBOSE")) std::fprintf(stderr, "ptcore: walk32: %d blocks per CU for the FP32 walk, %d for the FP64 traversal of the slow list\n", d.blocks_per_cu_walk, d.blocks_per_cu_wf);
        }
    }
    if (fr.primary_pass) {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pick_primary(fr.stats_on, fr.scan), PT_BLOCK, 0));
        d.blocks_per_cu_primary = std::max(1, std::min(nb, 8));
    }
    if (fr.split_rounds > 0) {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pick_trace(fr.stats_on, ctx->profile_sections, fr.scan, ptk::FORM_SPLIT), PT_BLOCK, lds));
        d.blocks_per_cu_split = std::max(1, std::min(nb, ctx->max_blocks_per_cu));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pick_glass(fr.stats_on, fr.scan), PT_BLOCK, fr.glass_lds_bytes));
        d.blocks_per_cu_glass = std::max(1, std::min(nb, PT_GLASS_MAX_BLOCKS_PER_CU));
    }
    // per-pass buffers: 90 B per job (radiance record, primary ray, stream state, draw count; + 8 B with pixel stats) and, for the
    // forms that park paths in HBM, two or three path-state queues
    const bool queues = fr.wavefront || (fr.split_rounds > 0 && fr.has_glass) || fr.primary_pass;
    const size_t nqueues = !queues ? 0 : fr.wavefront ? 3 : fr.primary_pass ? 1 : 2;  // (the primary pass of the BVH path fills the continuation queue only)
    const size_t qplanes = fr.stats_on ? 6 : 4;
    const size_t qentry = 10 * sizeof(double) + sizeof(unsigned long long) + qplanes * sizeof(uint32_t);
    const size_t job_bytes = 4 * sizeof(double) + 6 * sizeof(double) + sizeof(unsigned long long) + sizeof(uint16_t) +
                             (fr.stats_on ? 2 * sizeof(uint32_t) : 0);
    auto queue_cap = [&](size_t njobs_max) { return njobs_max + queue_slack(ctx, d, njobs_max); };
    auto need_bytes = [&](uint32_t chunk) {
        const size_t nj = (size_t)ns * chunk;
        return nj * job_bytes + nqueues * queue_cap(nj) * qentry + (fr.wavefront && ctx->wf_sort ? 2 * queue_cap(nj) * sizeof(uint32_t) : 0) +
               (fr.walk32 ? (PT_CAND_MAX + 2) * queue_cap(nj) * sizeof(uint32_t) : 0);
    };
    auto held_bytes = [&]() { return dev_held_bytes(d); };
    // The budget covers everything a pass holds, the window slack of the queues included: when the queues push the total over it,
    // the samples per pass shrink (a frame is cut into more passes; pixels do not depend on that).
    if (cfg_chunk_free(fr) && need_bytes(fr.chunk) > fr.budget_bytes) {
        const size_t fixed = need_bytes(1) > (size_t)ns * (job_bytes + nqueues * qentry) ? need_bytes(1) - (size_t)ns * (job_bytes + nqueues * qentry) : 0;
        const size_t per = (size_t)ns * (job_bytes + nqueues * qentry);
        fr.chunk = (uint32_t)std::max<size_t>(1, fr.budget_bytes > fixed ? (fr.budget_bytes - fixed) / per : 1);
        while (fr.chunk > 1 && need_bytes(fr.chunk) > fr.budget_bytes) fr.chunk -= std::max(1u, fr.chunk / 64u);
        balance_chunk(fr);
    }
    // A device that once could not give the budget keeps the size that fitted (trying the full budget again on every frame costs
    // seconds of hipMalloc / hipFree per frame when two processes share one GPU) -- until the device shows room for it again.
    if (d.bytes_cap && need_bytes(fr.chunk) > d.bytes_cap) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + held_bytes() >= need_bytes(fr.chunk) + need_bytes(fr.chunk) / 16) {
            d.bytes_cap = 0;  // whatever took the memory is gone: back to the full size
        } else {
            while (fr.chunk > 1 && need_bytes(fr.chunk) > d.bytes_cap) fr.chunk = std::max<uint32_t>(1, fr.chunk / 2);
        }
    }
    // when the device cannot give that much right now the chunk is halved until it can
    // (the ordered accumulation makes the pixels independent of the chunk size)
    for (;;) {
        const size_t njobs_max = (size_t)ns * fr.chunk;
        hipError_t e = d.L.reserve(4 * njobs_max);
        if (e == hipSuccess) e = d.ray.reserve(6 * njobs_max);
        if (e == hipSuccess) e = d.ray_rng.reserve(njobs_max);
        if (e == hipSuccess) e = d.ray_ndraw.reserve(njobs_max);
        if (e == hipSuccess && fr.stats_on) e = d.job_seg.reserve(njobs_max);
        if (e == hipSuccess && fr.stats_on) e = d.job_draw.reserve(njobs_max);
        if (queues) {  // path-state queues: 10 doubles + stream state + job, depth, hit object, answer (+ 2 counters) per entry
            // one entry per job at most, plus the slots the waves of the writing passes reserve in windows and may leave empty
            size_t qcap = queue_cap(njobs_max);
            // PTCORE_DEBUG_QUEUE_CAP=<entries> (tests only): queues too small for the frame, to show that an overflow fails the
            // frame with PT_ERR_STATE instead of writing outside them
            if (const char *dbg = std::getenv("PTCORE_DEBUG_QUEUE_CAP")) qcap = (size_t)std::max(64L, std::atol(dbg));
            if (!fr.primary_pass) {
                if (e == hipSuccess) e = d.gq_d.reserve(10 * qcap);
                if (e == hipSuccess) e = d.gq_rs.reserve(qcap);
                if (e == hipSuccess) e = d.gq_u32.reserve(qplanes * qcap);
            }
            if (e == hipSuccess) e = d.cq_d.reserve(10 * qcap);
            if (e == hipSuccess) e = d.cq_rs.reserve(qcap);
            if (e == hipSuccess) e = d.cq_u32.reserve(qplanes * qcap);
            if (fr.wavefront) {
                if (e == hipSuccess) e = d.xq_d.reserve(10 * qcap);
                if (e == hipSuccess) e = d.xq_rs.reserve(qcap);
                if (e == hipSuccess) e = d.xq_u32.reserve(qplanes * qcap);
                if (ctx->wf_sort) {
                    if (e == hipSuccess) e = d.wf_perm.reserve(qcap);
                    if (e == hipSuccess) e = d.wf_key.reserve(qcap);
                    if (e == hipSuccess) e = d.wf_bins.reserve(PT_WF_BINS + 8);
                }
                if (fr.walk32) {
                    if (e == hipSuccess) e = d.cand_ids.reserve((size_t)PT_CAND_MAX * qcap);
                    if (e == hipSuccess) e = d.cand_n.reserve(qcap);
                    if (e == hipSuccess) e = d.slow_list.reserve(qcap);
                }
            }
            d.q_cap = qcap;
        }
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || fr.chunk <= 1) return fail(PT_ERR_HIP, std::string("job buffers: ") + hipGetErrorString(e));
        d.L.release(); d.ray.release(); d.ray_rng.release(); d.ray_ndraw.release(); d.job_seg.release(); d.job_draw.release();
        d.gq_d.release(); d.cq_d.release(); d.gq_rs.release(); d.cq_rs.release(); d.gq_u32.release(); d.cq_u32.release();
        d.xq_d.release(); d.xq_rs.release(); d.xq_u32.release(); d.wf_perm.release(); d.wf_key.release();
        d.cand_ids.release(); d.cand_n.release(); d.slow_list.release();
        fr.chunk = std::max<uint32_t>(1, fr.chunk / 2);
        d.bytes_cap = need_bytes(fr.chunk);
        if (std::getenv("PTCORE_VERBOSE")) std::fprintf(stderr, "ptcore: device %d is short of memory, samples per pass reduced to %u\n", d.ordinal, fr.chunk);
    }
    return PT_OK;
}

// One chunk in the wavefront form (pt_wavefront.h): primary rays are in the ray buffers (raygen_kernel ran); queue words:
// [0] item cursor of the running traversal pass, [1] [2] entries of the two path queues, [3] entries of the exit queue.
int32_t dev_step_wavefront(pt_ctx *ctx, Device &d, const DevFrame &F, const TraceBuffers &B) {
    Frame &fr = ctx->frame;
    const bool bvh = fr.scan == ptk::SCAN_BVH || fr.scan == ptk::SCAN_VERIFY_BVH;
    const bool verify = fr.scan == ptk::SCAN_VERIFY_BVH || fr.scan == ptk::SCAN_VERIFY;
    const bool stats = fr.stats_on;
    unsigned int *qw = d.queue.p;
    const size_t cap = d.q_cap;
    auto bind = [&](PathQueue &q, DevBuf<double> &qd, DevBuf<unsigned long long> &qrs, DevBuf<uint32_t> &qu, unsigned int *count) {
        std::memset(&q, 0, sizeof q);
        q.d = qd.p;
        q.rs = qrs.p;
        q.job = qu.p;
        q.depth = reinterpret_cast<int32_t *>(qu.p + cap);
        q.best = reinterpret_cast<int32_t *>(qu.p + 2 * cap);
        q.hit = reinterpret_cast<int32_t *>(qu.p + 3 * cap);
        q.jseg = stats ? qu.p + 4 * cap : nullptr;
        q.jdraw = stats ? qu.p + 5 * cap : nullptr;
        q.count = count;
        q.cap = (uint32_t)cap;
    };
    PathQueue qa, qb, qe;
    bind(qa, d.cq_d, d.cq_rs, d.cq_u32, qw + 1);
    bind(qb, d.xq_d, d.xq_rs, d.xq_u32, qw + 2);
    bind(qe, d.gq_d, d.gq_rs, d.gq_u32, qw + 3);
    ptk::WfArgs A;
    std::memset(&A, 0, sizeof A);
    A.F = F;
    A.F.fresh = F.njobs;
    A.F.bvh_min_lanes = ctx->wf_min_lanes;
    A.sky = fr.sky;
    A.B = B;
    A.cursor = qw;
    const size_t lds_scan = fr.lds_bytes, lds_shade = fr.shade_lds_bytes, lds_mat = (size_t)F.nmat * sizeof(DevMat);
    const uint32_t blocks_all = (F.njobs + PT_BLOCK - 1) / PT_BLOCK;
    const uint32_t grid_scan = std::max(1u, std::min((uint32_t)(d.num_cu * d.blocks_per_cu_wf), blocks_all));
    // shading passes: PT_WF_PASS_BLOCKS_PER_CU blocks per CU.  Their waves append to the next level's queue in windows of
    // PT_CONT_BLOCK slots (one atomic per window: a pass this short cannot afford more on one address); shade + exit pass
    // together leave at most 2 x grid_pass x 4 waves x PT_CONT_BLOCK slots empty, which is what queue_slack() allocates.
    const uint32_t grid_pass = std::max(1u, std::min((uint32_t)(d.num_cu * PT_WF_PASS_BLOCKS_PER_CU), blocks_all));
    const int levels = std::max(0, fr.cfg.max_depth);
    if (int32_t rc = dev_events(d, d.ev_trace, d.n_trace + 4 * (size_t)levels + 1)) return rc;
    if (int32_t rc = dev_events(d, d.ev_glass, d.n_glass + 3 * (size_t)levels + 2)) return rc;
    auto timed = [&](std::vector<EventPair> &v, size_t &n, auto &&launch) -> int32_t {
        EventPair &e = v[n++];
        HIP_TRY(hipEventRecord(e.a, d.stream));
        launch();
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e.b, d.stream));
        return PT_OK;
    };
    // walk32 (pt_walk32.h): the FP32 walk lists candidates, the FP64 traversal answers the few entries the walk hands over
    const bool walk32 = fr.walk32 && bvh;
    const bool walk_diag = std::getenv("PTCORE_WALK_STATS") != nullptr;
    auto walk_args = [&]() {
        ptk::Walk32Args K;
        std::memset(&K, 0, sizeof K);
        K.W = A;
        K.cand_ids = d.cand_ids.p;
        K.cand_n = d.cand_n.p;
        K.slow_list = d.slow_list.p;
        K.slow_count = qw + 6;
        K.cores = d.bvh_cores.p;
        K.min_lanes = ctx->wf_min_lanes;
        K.diag = d.counters.p + 24;
        return K;
    };
    const uint32_t grid_walk = std::max(1u, std::min((uint32_t)(d.num_cu * std::max(1, d.blocks_per_cu_walk)), blocks_all));
    auto scan_pass = [&](int mode) -> int32_t {
        HIP_TRY(hipMemsetAsync(qw, 0, sizeof(unsigned int), d.stream));
        if (d.trace_is_split.size() <= d.n_trace) d.trace_is_split.resize(d.n_trace + 1);
        d.trace_is_split[d.n_trace] = 0;
        if (walk32) {
            HIP_TRY(hipMemsetAsync(qw + 6, 0, sizeof(unsigned int), d.stream));
            const ptk::Walk32Args K = walk_args();
            if (int32_t rc = timed(d.ev_trace, d.n_trace, [&] {
                    if (walk_diag) {
                        if (mode == 0) hipLaunchKernelGGL((ptk::wf_walk32_kernel<0, true>), dim3(grid_walk), dim3(PT_BLOCK), walk32_lds_bytes(), d.stream, K);
                        else hipLaunchKernelGGL((ptk::wf_walk32_kernel<1, true>), dim3(grid_walk), dim3(PT_BLOCK), walk32_lds_bytes(), d.stream, K);
                    } else if (mode == 0) hipLaunchKernelGGL((ptk::wf_walk32_kernel<0>), dim3(grid_walk), dim3(PT_BLOCK), walk32_lds_bytes(), d.stream, K);
                    else hipLaunchKernelGGL((ptk::wf_walk32_kernel<1>), dim3(grid_walk), dim3(PT_BLOCK), walk32_lds_bytes(), d.stream, K);
                }))
                return rc;
            // the entries the walk handed over, through the FP64 traversal (their number lives on the device)
            HIP_TRY(hipMemsetAsync(qw, 0, sizeof(unsigned int), d.stream));
            if (int32_t rc = dev_events(d, d.ev_trace, d.n_trace + 1)) return rc;
            if (d.trace_is_split.size() <= d.n_trace) d.trace_is_split.resize(d.n_trace + 1);
            d.trace_is_split[d.n_trace] = 0;
            ptk::WfArgs S = A;
            S.perm = d.slow_list.p;
            S.n_sorted = qw + 6;
            const uint32_t grid_slow = std::max(1u, std::min(grid_scan, (uint32_t)d.num_cu));
            return timed(d.ev_trace, d.n_trace, [&] {
                if (mode == 0) hipLaunchKernelGGL((ptk::wf_traverse_kernel<0, false>), dim3(grid_slow), dim3(PT_BLOCK), lds_scan, d.stream, S);
                else hipLaunchKernelGGL((ptk::wf_traverse_kernel<1, false>), dim3(grid_slow), dim3(PT_BLOCK), lds_scan, d.stream, S);
            });
        }
        return timed(d.ev_trace, d.n_trace, [&] {
            if (bvh) {
                if (mode == 0) {
                    if (verify) hipLaunchKernelGGL((ptk::wf_traverse_kernel<0, true>), dim3(grid_scan), dim3(PT_BLOCK), lds_scan, d.stream, A);
                    else hipLaunchKernelGGL((ptk::wf_traverse_kernel<0, false>), dim3(grid_scan), dim3(PT_BLOCK), lds_scan, d.stream, A);
                } else {
                    if (verify) hipLaunchKernelGGL((ptk::wf_traverse_kernel<1, true>), dim3(grid_scan), dim3(PT_BLOCK), lds_scan, d.stream, A);
                    else hipLaunchKernelGGL((ptk::wf_traverse_kernel<1, false>), dim3(grid_scan), dim3(PT_BLOCK), lds_scan, d.stream, A);
                }
            } else {
                if (mode == 0) {
                    if (verify) hipLaunchKernelGGL((ptk::wf_scan_flat_kernel<0, true>), dim3(grid_scan), dim3(PT_BLOCK), lds_scan, d.stream, A);
                    else hipLaunchKernelGGL((ptk::wf_scan_flat_kernel<0, false>), dim3(grid_scan), dim3(PT_BLOCK), lds_scan, d.stream, A);
                } else {
                    if (verify) hipLaunchKernelGGL((ptk::wf_scan_flat_kernel<1, true>), dim3(grid_scan), dim3(PT_BLOCK), lds_scan, d.stream, A);
                    else hipLaunchKernelGGL((ptk::wf_scan_flat_kernel<1, false>), dim3(grid_scan), dim3(PT_BLOCK), lds_scan, d.stream, A);
                }
            }
        });
    };
    // fresh jobs -> queue A
    A.qin = qa;
    A.qout = qb;
    A.qexit = qe;
    if (int32_t rc = timed(d.ev_glass, d.n_glass, [&] {
            if (stats) hipLaunchKernelGGL(ptk::wf_init_kernel<true>, dim3(std::min(blocks_all, (uint32_t)d.num_cu * 8u)), dim3(PT_BLOCK), 0, d.stream, A);
            else hipLaunchKernelGGL(ptk::wf_init_kernel<false>, dim3(std::min(blocks_all, (uint32_t)d.num_cu * 8u)), dim3(PT_BLOCK), 0, d.stream, A);
        }))
        return rc;
    PathQueue cur_in = qa, cur_out = qb;
    for (int level = 0; level < levels; level++) {
        A.qin = cur_in;
        A.qout = cur_out;
        A.qexit = qe;
        A.perm = nullptr;
        if (ctx->wf_sort && level >= 1 && bvh) {
            // primary rays leave raygen_kernel in pixel order, which is coherent; from the first bounce on the queue order
            // means nothing, and the traversal pass takes the rays by direction octant and cell of the origin instead
            A.bin_count = d.wf_bins.p;
            A.bin_key = d.wf_key.p;
            HIP_TRY(hipMemsetAsync(d.wf_bins.p, 0, (PT_WF_BINS + 1) * sizeof(uint32_t), d.stream));
            if (int32_t rc = timed(d.ev_glass, d.n_glass, [&] {
                    hipLaunchKernelGGL(ptk::wf_bin_count_kernel, dim3(grid_pass), dim3(PT_BLOCK), 0, d.stream, A);
                    hipLaunchKernelGGL(ptk::wf_bin_scan_kernel, dim3(1), dim3(1024), 0, d.stream, A);
                    hipLaunchKernelGGL(ptk::wf_bin_scatter_kernel, dim3(grid_pass), dim3(PT_BLOCK), 0, d.stream, A, d.wf_perm.p);
                }))
                return rc;
            A.perm = d.wf_perm.p;
            A.n_sorted = d.wf_bins.p + PT_WF_BINS;
        }
        if (int32_t rc = scan_pass(0)) return rc;
        A.perm = nullptr;
        HIP_TRY(hipMemsetAsync(cur_out.count, 0, sizeof(unsigned int), d.stream));
        HIP_TRY(hipMemsetAsync(qe.count, 0, sizeof(unsigned int), d.stream));
        if (int32_t rc = timed(d.ev_glass, d.n_glass, [&] {
                if (walk32) {
                    const ptk::Walk32Args K = walk_args();
                    if (stats) {
                        if (verify) hipLaunchKernelGGL((ptk::wf_shade32_kernel<true, true>), dim3(grid_pass), dim3(PT_BLOCK), lds_mat, d.stream, K);
                        else hipLaunchKernelGGL((ptk::wf_shade32_kernel<true, false>), dim3(grid_pass), dim3(PT_BLOCK), lds_mat, d.stream, K);
                    } else {
                        if (verify) hipLaunchKernelGGL((ptk::wf_shade32_kernel<false, true>), dim3(grid_pass), dim3(PT_BLOCK), lds_mat, d.stream, K);
                        else hipLaunchKernelGGL((ptk::wf_shade32_kernel<false, false>), dim3(grid_pass), dim3(PT_BLOCK), lds_mat, d.stream, K);
                    }
                } else if (stats) hipLaunchKernelGGL(ptk::wf_shade_kernel<true>, dim3(grid_pass), dim3(PT_BLOCK), lds_shade, d.stream, A);
                else hipLaunchKernelGGL(ptk::wf_shade_kernel<false>, dim3(grid_pass), dim3(PT_BLOCK), lds_shade, d.stream, A);
            }))
            return rc;
        if (fr.has_glass) {
            A.qin = qe;
            if (int32_t rc = scan_pass(1)) return rc;
            if (int32_t rc = timed(d.ev_glass, d.n_glass, [&] {
                    if (walk32) {
                        const ptk::Walk32Args K = walk_args();
                        if (stats) {
                            if (verify) hipLaunchKernelGGL((ptk::wf_exit32_kernel<true, true>), dim3(grid_pass), dim3(PT_BLOCK), lds_mat, d.stream, K);
                            else hipLaunchKernelGGL((ptk::wf_exit32_kernel<true, false>), dim3(grid_pass), dim3(PT_BLOCK), lds_mat, d.stream, K);
                        } else {
                            if (verify) hipLaunchKernelGGL((ptk::wf_exit32_kernel<false, true>), dim3(grid_pass), dim3(PT_BLOCK), lds_mat, d.stream, K);
                            else hipLaunchKernelGGL((ptk::wf_exit32_kernel<false, false>), dim3(grid_pass), dim3(PT_BLOCK), lds_mat, d.stream, K);
                        }
                    } else if (stats) hipLaunchKernelGGL(ptk::wf_exit_kernel<true>, dim3(grid_pass), dim3(PT_BLOCK), lds_mat, d.stream, A);
                    else hipLaunchKernelGGL(ptk::wf_exit_kernel<false>, dim3(grid_pass), dim3(PT_BLOCK), lds_mat, d.stream, A);
                }))
                return rc;
        }
        std::swap(cur_in, cur_out);
        // deep presets (the "final" mode asks for 80 levels): stop as soon as no path is left
        if (level >= 8 && level % 4 == 0 && level + 1 < levels) {
            unsigned int left = 0;
            HIP_TRY(hipMemcpyAsync(&left, cur_in.count, sizeof left, hipMemcpyDeviceToHost, d.stream));
            HIP_TRY(hipStreamSynchronize(d.stream));
            if (left == 0) break;
        }
    }
    return PT_OK;
}

// Adds samples [s0, s0+S) to this device's running sums.
int32_t dev_step(pt_ctx *ctx, Device &d, uint32_t s0, uint32_t S) {
    Frame &fr = ctx->frame;
    if (d.nlocal == 0 || S == 0) return PT_OK;
    HIP_TRY(hipSetDevice(d.ordinal));
    DevFrame F = fr.F;
    F.shard_index = d.shard.index;
    F.shard_count = d.shard.count;
    F.nlocal = d.nlocal;
    F.s0 = s0;
    F.S = S;
    F.njobs = d.nslots * S;
    // jobs a wave claims per pop of the item cursor: 256, and 512 for the bitmask scans once a pass holds 128 samples per pixel
    // or more (same-box sweeps, profiles/r02_claim_sweep.txt: C4 at 265 spp per pass 664 / 653 / 654 / 659 / 673 ms for 256 / 512 /
    // 1024 / 2048 / 4096, at 79 spp per pass 677 / 675 / 678 ms; the BVH path loses 3 % at 512 and 7 % at 1024)
    {
        const bool bvh_scan = fr.scan == ptk::SCAN_BVH || fr.scan == ptk::SCAN_VERIFY_BVH;
        F.claim = ctx->claim ? ctx->claim : (!bvh_scan && S >= 128u) ? 512u : 256u;
    }
    TraceBuffers B;
    B.objs = d.objs.p;
    B.mats = d.mats.p;
    B.bsph = d.bsph.p;
    B.bbox = d.bbox.p;
    B.plane_idx = d.plane_idx.p;
    B.bvh_nodes = d.bvh_nodes.p;
    B.bvh_objs = d.bvh_objs.p;
    B.bvh_cores = d.bvh_cores.p;
    B.L = d.L.p;
    B.ray = d.ray.p;
    B.ray_rng = d.ray_rng.p;
    B.ray_ndraw = d.ray_ndraw.p;
    B.job_seg = fr.stats_on ? d.job_seg.p : nullptr;
    B.job_draw = fr.stats_on ? d.job_draw.p : nullptr;
    B.queue = d.queue.p;
    B.counters = d.counters.p;
    B.prof = ctx->profile_sections ? d.prof.p : nullptr;
    const bool timeline = !ctx->profile_sections && std::getenv("PTCORE_DEBUG_TIMELINE") != nullptr;  // diagnostics: when each wave of a trace launch retires
    if (timeline) {
        HIP_TRY(d.prof.reserve(65536));
        B.prof = d.prof.p;
    }

    const int rounds = fr.split_rounds;
    if (int32_t rc = dev_events(d, d.ev_trace, d.n_trace + (size_t)rounds + 2)) return rc;
    if (int32_t rc = dev_events(d, d.ev_glass, d.n_glass + (size_t)rounds + 1)) return rc;
    if (int32_t rc = dev_events(d, d.ev_resolve, d.n_resolve + 1)) return rc;
    if (!d.first_recorded) {
        HIP_TRY(hipEventRecord(d.ev_first, d.stream));
        d.first_recorded = true;
    }
    // queue words: [0] item cursor of the running trace pass, [1] glass entries, [2],[3] continuation entries (one is
    // read by a trace pass while glass_kernel fills the other), [4] stays 0 (a first pass starts from no continuations)
    unsigned int *qw = d.queue.p;
    B.cont_in = qw + 4;
    B.bsph_diel = d.bsph_diel.p;
    B.bbox_diel = d.bbox_diel.p;
    std::memset(&B.glass, 0, sizeof B.glass);
    std::memset(&B.cont, 0, sizeof B.cont);
    if (rounds > 0 || fr.primary_pass) {
        const size_t cap = d.q_cap;
        auto bind = [&](PathQueue &q, DevBuf<double> &qd, DevBuf<unsigned long long> &qrs, DevBuf<uint32_t> &qu) {
            q.d = qd.p;
            q.rs = qrs.p;
            q.job = qu.p;
            q.depth = reinterpret_cast<int32_t *>(qu.p + cap);
            q.best = reinterpret_cast<int32_t *>(qu.p + 2 * cap);
            q.hit = reinterpret_cast<int32_t *>(qu.p + 3 * cap);
            q.jseg = fr.stats_on ? qu.p + 4 * cap : nullptr;
            q.jdraw = fr.stats_on ? qu.p + 5 * cap : nullptr;
            q.cap = (uint32_t)cap;
        };
        if (rounds > 0) bind(B.glass, d.gq_d, d.gq_rs, d.gq_u32);
        bind(B.cont, d.cq_d, d.cq_rs, d.cq_u32);
        B.glass.count = qw + 1;
    }
    {  // ray generation, then the trace passes (which also handle max_depth <= 0: black samples, camera draws counted)
        HIP_TRY(hipMemsetAsync(qw, 0, 8 * sizeof(unsigned int), d.stream));
        const size_t lds = fr.lds_bytes;
        const uint32_t waves_needed = (F.njobs + 63u) / 64u;
        if (int32_t rc = dev_events(d, d.ev_raygen, d.n_raygen + 1)) return rc;
        EventPair &eg = d.ev_raygen[d.n_raygen++];
        HIP_TRY(hipEventRecord(eg.a, d.stream));
        const char *rg_form = std::getenv("PTCORE_RAYGEN");  // A/B: "column" = round 2's walk down a lane's column, "simple" = one job per lane
        const bool rg_simple = std::getenv("PTCORE_RAYGEN_SIMPLE") || (rg_form && !std::strcmp(rg_form, "simple"));
        if (fr.cam.lens_radius > 0 && !rg_simple && !(rg_form && !std::strcmp(rg_form, "column")))  // thin lens: the rejection loop over a wave's pool of jobs
            hipLaunchKernelGGL(ptk::raygen_lens_pool_kernel, dim3((F.njobs + PT_BLOCK * PT_RG_POOL_ROWS - 1) / (PT_BLOCK * PT_RG_POOL_ROWS)), dim3(PT_BLOCK), 0,
                               d.stream, F, fr.cam, d.ray.p, d.ray_rng.p, d.ray_ndraw.p);
        else if (fr.cam.lens_radius > 0 && !rg_simple)
            hipLaunchKernelGGL(ptk::raygen_lens_kernel, dim3((F.njobs + PT_BLOCK * PT_RG_ROWS - 1) / (PT_BLOCK * PT_RG_ROWS)), dim3(PT_BLOCK), 0,
                               d.stream, F, fr.cam, d.ray.p, d.ray_rng.p, d.ray_ndraw.p);
        else
            hipLaunchKernelGGL(ptk::raygen_kernel, dim3((F.njobs + PT_BLOCK - 1) / PT_BLOCK), dim3(PT_BLOCK), 0, d.stream, F, fr.cam,
                               d.ray.p, d.ray_rng.p, d.ray_ndraw.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(eg.b, d.stream));
        if (fr.wavefront) {
            if (int32_t rc = dev_step_wavefront(ctx, d, F, B)) return rc;
        } else {
        const bool pass_log = std::getenv("PTCORE_DEBUG_PASS_LOG") != nullptr;
        auto launch_trace = [&](bool split, bool first) -> int32_t {
            TraceArgs A;
            A.F = F;
            A.F.fresh = first ? F.njobs : 0u;
            A.sky = fr.sky;
            A.B = B;
            uint32_t grid = (uint32_t)(d.num_cu * (split ? d.blocks_per_cu_split : d.blocks_per_cu));
            // (later passes: the item count lives on the device; no pass holds more items than the chunk has jobs, which is also
            // the bound queue_slack() assumes)
            grid = std::max(1u, std::min(grid, (waves_needed + 3u) / 4u));
            if (d.trace_is_split.size() <= d.n_trace) d.trace_is_split.resize(d.n_trace + 1);
            d.trace_is_split[d.n_trace] = split ? 1 : 0;
            EventPair &e = d.ev_trace[d.n_trace++];
            HIP_TRY(hipEventRecord(e.a, d.stream));
            hipLaunchKernelGGL(pick_trace(fr.stats_on, ctx->profile_sections, fr.scan, split ? (int)ptk::FORM_SPLIT : fr.tail_form),
                               dim3(grid), dim3(PT_BLOCK), lds, d.stream, A);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(e.b, d.stream));
            if (timeline) {
                HIP_TRY(hipStreamSynchronize(d.stream));
                const uint32_t nw = std::min(grid * 4u, 65536u);
                std::vector<unsigned long long> t(nw);
                HIP_TRY(hipMemcpy(t.data(), d.prof.p, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::sort(t.begin(), t.end());
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
                auto back = [&](double q) { return (double)(t[nw - 1] - t[(size_t)(q * (nw - 1))]) * 1e-5; };  // ms before the last wave
                std::fprintf(stderr, "ptcore timeline: %s form %d  %.3f ms, %u waves; before the last wave retired: first %.3f ms, 1 %% %.3f, 10 %% %.3f, 25 %% %.3f, 50 %% %.3f, 75 %% %.3f, 90 %% %.3f, 99 %% %.3f; mean %.3f ms\n",
                             split ? "trace<split>" : "trace<tail>", split ? 1 : fr.tail_form, ms, nw, back(0.0), back(0.01), back(0.10), back(0.25), back(0.50), back(0.75), back(0.90), back(0.99),
                             [&] { double a = 0; for (auto v : t) a += (double)(t[nw - 1] - v); return a / nw * 1e-5; }());
            }
            if (pass_log) {  // PTCORE_DEBUG_PASS_LOG=1 (diagnostics): what every trace pass did; serialises the stream
                HIP_TRY(hipStreamSynchronize(d.stream));
                unsigned long long c[24];
                HIP_TRY(hipMemcpy(c, d.counters.p, sizeof c, hipMemcpyDeviceToHost));
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
                unsigned long long *prev = d.pass_log_prev;
                std::fprintf(stderr, "ptcore pass: %s grid %u  %.3f ms  segments +%llu  exit scans +%llu  parked +%llu  ended here +%llu  continuations in +%llu  -> %.1f Mseg/s\n",
                             split ? "trace<split>" : "trace<all-in-one>", grid, ms, c[0] - prev[0], c[1] - prev[1], c[5] - prev[5], c[18] - prev[18],
                             c[7] - prev[7], (double)(c[0] - prev[0]) / (ms * 1e3));
                std::memcpy(prev, c, sizeof c);
            }
            return PT_OK;
        };
        if (fr.primary_pass) {
            // BVH scenes: the first segment of every path by the wave-cooperative kernel (pt_primary.h); what goes on -- and what that
            // kernel is not meant for, unshaded -- reaches the per-lane loop through the continuation queue
            B.cont_in = qw + 2;
            B.cont.count = qw + 2;
            TraceArgs A;
            A.F = F;
            A.F.fresh = 0u;
            A.sky = fr.sky;
            A.B = B;
            const uint32_t pgrid = std::max(1u, std::min((uint32_t)(d.num_cu * d.blocks_per_cu_primary), (F.njobs + PT_BLOCK - 1) / PT_BLOCK));  // same bound as queue_slack()
            if (d.trace_is_split.size() <= d.n_trace) d.trace_is_split.resize(d.n_trace + 1);
            d.trace_is_split[d.n_trace] = 0;
            EventPair &e = d.ev_trace[d.n_trace++];
            HIP_TRY(hipEventRecord(e.a, d.stream));
            hipLaunchKernelGGL(pick_primary(fr.stats_on, fr.scan), dim3(pgrid), dim3(PT_BLOCK), 0, d.stream, A);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(e.b, d.stream));
            if (pass_log) {
                HIP_TRY(hipStreamSynchronize(d.stream));
                unsigned long long c[48];
                HIP_TRY(hipMemcpy(c, d.counters.p, sizeof c, hipMemcpyDeviceToHost));
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
                std::fprintf(stderr, "ptcore pass: primary<wave> grid %u  %.3f ms  segments shaded %llu  handed on %llu  wave-level node visits %llu over %llu blocks of 64 jobs (%llu more handed over unwalked)\n",
                             pgrid, ms, c[0] - d.pass_log_prev[0], c[6], c[40], c[41], c[42]);
                std::memcpy(d.pass_log_prev, c, sizeof d.pass_log_prev);
            }
            if (int32_t rc = launch_trace(false, false)) return rc;
        } else if (rounds == 0) {
            if (int32_t rc = launch_trace(false, true)) return rc;
        } else {
            // Split passes: trace (dielectric hits -> glass queue), glass (-> continuation queue), `rounds` times; what is
            // still under way then (paths with more than `rounds` dielectric bounces) finishes in the all-in-one form.
            const uint32_t glass_grid = std::max(1u, std::min((uint32_t)(d.num_cu * d.blocks_per_cu_glass), (F.njobs + PT_BLOCK - 1) / PT_BLOCK));  // same bound as queue_slack()
            for (int r = 0; r < rounds; r++) {
                if (r > 0) HIP_TRY(hipMemsetAsync(qw, 0, 2 * sizeof(unsigned int), d.stream));  // cursor and glass count
                unsigned int *c_in = r == 0 ? qw + 4 : qw + 2 + (r & 1), *c_out = qw + 2 + ((r + 1) & 1);
                HIP_TRY(hipMemsetAsync(c_out, 0, sizeof(unsigned int), d.stream));
                B.cont_in = c_in;
                B.cont.count = c_out;
                if (int32_t rc = launch_trace(true, r == 0)) return rc;
                if (!fr.has_glass) break;  // nothing can have entered the glass queue: the frame is done
                EventPair &e = d.ev_glass[d.n_glass++];
                HIP_TRY(hipEventRecord(e.a, d.stream));
                {
                    ptk::GlassArgs GA;
                    GA.F = F;
                    GA.B = B;
                    hipLaunchKernelGGL(pick_glass(fr.stats_on, fr.scan), dim3(glass_grid), dim3(PT_BLOCK), fr.glass_lds_bytes, d.stream, GA);
                }
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(e.b, d.stream));
            }
            if (fr.has_glass) {
                HIP_TRY(hipMemsetAsync(qw, 0, sizeof(unsigned int), d.stream));
                B.cont_in = qw + 2 + (rounds & 1);
                B.cont.count = qw + 5;  // unused by the all-in-one form
                if (int32_t rc = launch_trace(false, false)) return rc;
            }
        }
        }
    }
    ptk::ResolveArgs R;
    std::memset(&R, 0, sizeof R);
    R.L = d.L.p;
    R.job_seg = B.job_seg;
    R.job_draw = B.job_draw;
    R.acc = d.acc.p;
    R.acc_seg = fr.stats_on ? d.acc_seg.p : nullptr;
    R.acc_draw = fr.stats_on ? d.acc_draw.p : nullptr;
    R.nslots = d.nslots;
    R.njobs = F.njobs;
    R.S = S;
    R.first = d.acc_started ? 0 : 1;
    R.finish = 0;
    R.have_chunk = 1;
    R.inv_samples = 0;
    R.width = fr.cfg.width;
    R.height = fr.cfg.height;
    R.ntx = fr.ntx;
    R.shard_index = d.shard.index;
    R.shard_count = d.shard.count;
    EventPair &e = d.ev_resolve[d.n_resolve++];
    HIP_TRY(hipEventRecord(e.a, d.stream));
    hipLaunchKernelGGL(ptk::resolve_kernel, dim3((d.nslots + PT_BLOCK - 1) / PT_BLOCK), dim3(PT_BLOCK), 0, d.stream, R);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e.b, d.stream));
    d.acc_started = true;
    return PT_OK;
}

// Writes the current estimate of this device's tiles (normalised by spp_done).
int32_t dev_finish(pt_ctx *ctx, Device &d, int32_t spp_done, uint8_t *tiles_rgba, double *tiles_accum,
                   uint32_t *tiles_seg, uint32_t *tiles_draw) {
    Frame &fr = ctx->frame;
    if (d.nlocal == 0) return PT_OK;
    HIP_TRY(hipSetDevice(d.ordinal));
    ptk::ResolveArgs R;
    std::memset(&R, 0, sizeof R);
    R.acc = d.acc.p;
    R.acc_seg = fr.stats_on ? d.acc_seg.p : nullptr;
    R.acc_draw = fr.stats_on ? d.acc_draw.p : nullptr;
    R.tiles_rgba = tiles_rgba;
    R.tiles_accum = tiles_accum;
    R.tiles_seg = fr.stats_on ? tiles_seg : nullptr;
    R.tiles_draw = fr.stats_on ? tiles_draw : nullptr;
    R.nslots = d.nslots;
    R.first = d.acc_started ? 0 : 1;
    R.finish = 1;
    R.have_chunk = 0;
    R.inv_samples = 1.0 / (double)spp_done;  // renderer.go:97
    R.width = fr.cfg.width;
    R.height = fr.cfg.height;
    R.ntx = fr.ntx;
    R.shard_index = d.shard.index;
    R.shard_count = d.shard.count;
    if (int32_t rc = dev_events(d, d.ev_resolve, d.n_resolve + 1)) return rc;
    EventPair &e = d.ev_resolve[d.n_resolve++];
    HIP_TRY(hipEventRecord(e.a, d.stream));
    hipLaunchKernelGGL(ptk::resolve_kernel, dim3((d.nslots + PT_BLOCK - 1) / PT_BLOCK), dim3(PT_BLOCK), 0, d.stream, R);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e.b, d.stream));
    HIP_TRY(hipEventRecord(d.ev_last, d.stream));
    return PT_OK;
}

int32_t dev_collect(Device &d, pt_stats *st, int slot) {
    HIP_TRY(hipSetDevice(d.ordinal));
    HIP_TRY(hipStreamSynchronize(d.stream));
    if (d.nlocal == 0) return PT_OK;
    unsigned long long c[48] = {};
    HIP_TRY(hipMemcpy(c, d.counters.p, sizeof c, hipMemcpyDeviceToHost));
    if (std::getenv("PTCORE_WALK_STATS") && c[24 + 2])
        std::fprintf(stderr, "ptcore walk32: rays walked %llu (+%llu missing the scene cube, %llu handed over before the walk, %llu of them far AND through the cube, widest inflation %llu margins), "
                             "node visits %llu, core visits %llu, wave iterations %llu (%.1f lanes per iteration), refills %llu, candidates %llu, handed over for > %d candidates %llu, for stack depth %llu\n",
                     c[24 + 8], c[24 + 9], c[24 + 5], c[24 + 10], c[24 + 11], c[24 + 0], c[24 + 1], c[24 + 2], (double)(c[24] + c[25]) / (double)c[24 + 2], c[24 + 3], c[24 + 4], PT_CAND_MAX,
                     c[24 + 6], c[24 + 7]);
    if (c[19]) return fail(PT_ERR_STATE, "internal: a path-state queue overflowed (" + std::to_string(c[19]) + " paths lost); the frame is invalid");
    g_mismatches += c[4];
    if ((c[20] || c[21] || c[23]) && std::getenv("PTCORE_VERBOSE"))
        std::fprintf(stderr, "ptcore: BVH path: %llu wave-trips through the plain every-object scan (rays with non-finite or absurd components), "
                             "%llu through the careful traversal (far-away rays), %llu lane-trips with bounds the FP32 tests were not analysed for, "
                             "%llu rays scanned by a whole wave (bounds as wide as the scene)\n", c[20], c[21], c[22], c[23]);
    if (c[4]) std::memcpy(g_mismatch_sample, c + 8, sizeof g_mismatch_sample);
    st->segments += c[0];
    st->exit_scans += c[1];
    st->draws += c[2];
    st->samples += c[3];
    if (c[47]) st->shader_clock_mhz = (double)c[46] / ((double)c[47] / 100.0);  // cycles / (ticks of the 100 MHz counter) = MHz (device 0 of the last collected)
    st->glass_events += c[5];
    st->continuations += c[6];
    st->split_cont_in += c[7];
    st->split_finished += c[18];
    if (d.prof.p && slot == 0)
        HIP_TRY(hipMemcpy(g_profile_scratch, d.prof.p, sizeof g_profile_scratch, hipMemcpyDeviceToHost));
    double tr = 0, rs = 0, trs = 0, gl = 0;
    for (size_t i = 0; i < d.n_trace; i++) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, d.ev_trace[i].a, d.ev_trace[i].b));
        tr += ms;
        if (i < d.trace_is_split.size() && d.trace_is_split[i]) trs += ms;
    }
    for (size_t i = 0; i < d.n_glass; i++) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, d.ev_glass[i].a, d.ev_glass[i].b));
        gl += ms;
    }
    st->glass_ms = std::max(st->glass_ms, gl);
    st->trace_split_ms = std::max(st->trace_split_ms, trs);
    st->glass_launches += (int32_t)d.n_glass;
    for (size_t i = 0; i < d.n_trace && i < d.trace_is_split.size(); i++) st->trace_split_launches += d.trace_is_split[i] ? 1 : 0;
    for (size_t i = 0; i < d.n_resolve; i++) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, d.ev_resolve[i].a, d.ev_resolve[i].b));
        rs += ms;
    }
    double rg = 0;
    for (size_t i = 0; i < d.n_raygen; i++) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, d.ev_raygen[i].a, d.ev_raygen[i].b));
        rg += ms;
    }
    st->raygen_ms = std::max(st->raygen_ms, rg);
    float span = 0;
    if (d.first_recorded) HIP_TRY(hipEventElapsedTime(&span, d.ev_first, d.ev_last));
    st->trace_ms = std::max(st->trace_ms, tr);
    st->resolve_ms = std::max(st->resolve_ms, rs);
    st->device_ms = std::max(st->device_ms, (double)span);
    st->trace_launches += (int32_t)d.n_trace;
    st->resolve_launches += (int32_t)d.n_resolve;
    if (slot < 8) st->per_device_ms[slot] = span;
    return PT_OK;
}

// (Re)builds ctx->sd when the scene or the requested scan strategy changed.
int32_t scene_prepare(pt_ctx *ctx, const pt_scene *scene) {
    SceneData &sd = ctx->sd;
    // fast path for a scene that has not changed since the last frame (progressive previews, benchmark loops): compare the
    // caller's arrays with the copy kept from then; converting a 10^6-object world just to find it unchanged cost 25 ms a frame
    {
        const size_t nm = (size_t)std::max(0, scene->num_materials), no = (size_t)std::max(0, scene->num_objects);
        const bool raw_same = sd.valid && sd.scan_req == ctx->scan_mode && sd.raw_mats.size() == nm && sd.raw_objs.size() == no &&
                              (nm == 0 || std::memcmp(sd.raw_mats.data(), scene->materials, nm * sizeof(pt_material)) == 0) &&
                              (no == 0 || std::memcmp(sd.raw_objs.data(), scene->objects, no * sizeof(pt_object)) == 0);
        if (raw_same) return PT_OK;
        sd.raw_mats.assign(scene->materials, scene->materials + nm);
        sd.raw_objs.assign(scene->objects, scene->objects + no);
    }
    std::vector<DevObj> world;
    std::vector<DevMat> mats;
    scene_to_world(*scene, world, mats);
    const bool same = sd.valid && sd.scan_req == ctx->scan_mode && world.size() == sd.world.size() &&
                      mats.size() == sd.mats.size() &&
                      (world.empty() || std::memcmp(world.data(), sd.world.data(), world.size() * sizeof(DevObj)) == 0) &&
                      std::memcmp(mats.data(), sd.mats.data(), mats.size() * sizeof(DevMat)) == 0;
    if (same) return PT_OK;
    sd.valid = false;
    sd.world = std::move(world);
    sd.mats = std::move(mats);
    sd.scan_req = ctx->scan_mode;
    DevFrame &F = sd.Fs;
    std::memset(&F, 0, sizeof F);
    F.nobj = (int32_t)sd.world.size();
    F.nmat = (int32_t)sd.mats.size();
    build_broad(sd.world, sd);
    // closest-hit strategy: at most 32 spheres and 32 boxes -> candidate bitmasks; at most 128 of each -> the same in
    // groups of 32; more -> BVH.  PTCORE_SCAN overrides.
    int scan = ctx->scan_mode;
    // the LDS copy of the world (objects + materials + record indices) must leave room for five blocks per CU
    const size_t world_lds = sd.world.size() * sizeof(DevObj) + sd.mats.size() * sizeof(DevMat) + (sd.bsph.size() + sd.bbox.size() + sd.bsph_diel.size() + sd.bbox_diel.size()) * sizeof(int);
    const bool wide_ok = F.broad_ok == 2 && world_lds <= 30 * 1024;
    if (scan < 0) scan = F.broad_ok == 1 ? ptk::SCAN_BROAD : wide_ok ? ptk::SCAN_BROAD_WIDE : ptk::SCAN_BVH;
    if ((scan == ptk::SCAN_BROAD || scan == ptk::SCAN_VERIFY) && F.broad_ok != 1) {
        const bool verify = scan == ptk::SCAN_VERIFY;
        scan = wide_ok ? (verify ? ptk::SCAN_VERIFY_WIDE : ptk::SCAN_BROAD_WIDE) : (verify ? ptk::SCAN_VERIFY_BVH : ptk::SCAN_BVH);
    }
    if ((scan == ptk::SCAN_BROAD_WIDE || scan == ptk::SCAN_VERIFY_WIDE) && !wide_ok && F.broad_ok != 1)
        scan = scan == ptk::SCAN_VERIFY_WIDE ? ptk::SCAN_VERIFY_BVH : ptk::SCAN_BVH;
    // A NaN or an infinity in the geometry (reachable through the C ABI, not through the JSON loader) makes the reference's own
    // tests return NaN parameters, which its range tests accept (NaN compares false, objects.go:56-60, :110) and which then
    // poison `closest` for every later object of the loop.  Only the loop itself reproduces that: such a world takes the
    // plain object-by-object scan, whatever was asked for.
    // The same for coordinates or sizes of 1e37 and more: every culled strategy keeps FP32 bounds of the objects, which must stay
    // finite UPPER bounds (the hierarchy rounds its half extents up to a multiple of 256 ulp and has no room to do so within a
    // factor 34 of FLT_MAX).
    for (const DevObj &o : sd.world) {
        bool fin = std::fabs(o.radius) < 1e37;
        for (int k = 0; k < 3; k++) fin = fin && std::fabs(o.a[k]) < 1e37 && std::fabs(o.b[k]) < 1e37;
        if (!fin) scan = ptk::SCAN_UNIFORM;
    }
    sd.scan = scan;
    const bool big = scan == ptk::SCAN_BVH || scan == ptk::SCAN_VERIFY_BVH;
    sd.bvh_nodes.clear();
    sd.bvh_objs.clear();
    sd.bvh_cores.clear();
    F.bvh_root = F.bvh_root_exit = -1;
    const std::vector<DevObj> &w = sd.world;
    if (big) {
        std::vector<int32_t> finite;
        double Bnd = 1.0;
        for (size_t i = 0; i < w.size(); i++) {
            const int kind = w[i].kind & 0xff;
            if (kind == KIND_PLANE) continue;
            finite.push_back((int32_t)i);
            const ptbvh::Aabb bb = ptbvh::object_bounds(w[i]);
            for (int k = 0; k < 3; k++) Bnd = std::max(Bnd, std::max(std::fabs(bb.lo[k]), std::fabs(bb.hi[k])));
        }
        if (!(Bnd < 1e30)) Bnd = INFINITY;
        // two hierarchies: every finite object (closest-hit scans) and the dielectric ones only (exit
        // searches accept nothing else, renderer.go:333, and would otherwise walk the whole line of sight)
        std::vector<int32_t> glass;
        for (int32_t i : finite)
            if (w[(size_t)i].kind & 0x100) glass.push_back(i);
        const double margin = Bnd * (1.0 / 4096.0);
        ptbvh::Built built = ptbvh::build(w, finite, margin);
        ptbvh::Built builtd = ptbvh::build(w, glass, margin);
        // a tree too deep for the per-lane stack (adversarial spacing) is rebuilt with fewer SAH levels
        for (int lv = 16; built.stack_need >= PT_BVH_STACK && lv >= 0; lv -= 16) built = ptbvh::build(w, finite, margin, lv);
        for (int lv = 16; builtd.stack_need >= PT_BVH_STACK && lv >= 0; lv -= 16) builtd = ptbvh::build(w, glass, margin, lv);
        sd.bvh_depth = std::max(built.depth, builtd.depth);
        sd.bvh_stack_need = std::max(built.stack_need, builtd.stack_need);
        if (sd.bvh_stack_need >= PT_BVH_STACK)
            return fail(PT_ERR_INVALID, "BVH deeper than the traversal stack");
        // core twins (the FP32 walk of pt_walk32.h): inside-the-object boxes of the main tree's object slots; the dielectric
        // tree's twins are empty (an exit search takes no FP32 bound)
        // Only PTCORE_PIPELINE=walk32 reads them (and the bits 20-23 build_cores sets in the nodes' meta): the default loop neither
        // builds nor uploads 128 B per node for nothing.
        if (ctx->pipeline == 2) {
            sd.bvh_cores = ptbvh::build_cores(built, w, margin);
            sd.bvh_cores.resize(built.nodes.size() + builtd.nodes.size());
            for (size_t q = built.nodes.size(); q < sd.bvh_cores.size(); q++) {
                std::memset(&sd.bvh_cores[q], 0, sizeof(BvhNode));
            }
        }
        const int32_t node_off = (int32_t)built.nodes.size(), obj_off = (int32_t)built.order.size();
        F.bvh_main_nodes = node_off;
        sd.bvh_nodes = std::move(built.nodes);
        for (BvhNode nd : builtd.nodes) {  // the dielectric tree follows the main one in both arrays
            nd.node_base += node_off;
            nd.obj_base += obj_off;
            sd.bvh_nodes.push_back(nd);
        }
        // The walk of scan_bvh reads the first 96 bytes of a node only (six 16-byte requests per lane and visit instead of seven: the
        // CU's address unit, not the caches, is what its visits wait for -- DESIGN 10.3): node_base, obj_base and meta ride in the LOW BYTES
        // of the twelve half extents, which are rounded up to a multiple of 256 ulp first (they only ever had to be upper bounds).
        // The fields themselves stay where they were for everything else that reads a node (host tools, the walk32 form).
        for (BvhNode &nd : sd.bvh_nodes) {
            const uint32_t payload[3] = {(uint32_t)nd.node_base, (uint32_t)nd.obj_base, nd.meta & 0xffffffu};
            for (int k = 0; k < 3; k++)
                for (int sl = 0; sl < 4; sl++) {
                    uint32_t b;
                    std::memcpy(&b, &nd.h[k][sl], 4);
                    if ((b & 0x7f800000u) == 0x7f800000u || (b >> 31)) b = 0x7f7fff00u;  // inf (unbounded slab) / NaN: the largest finite float
                    else b = (b + 0xffu) & ~0xffu;                                          // up to the next multiple of 256 ulp
                    if (b >= 0x7f800000u) b = 0x7f7fff00u;
                    b |= (payload[k] >> (8 * sl)) & 0xffu;
                    std::memcpy(&nd.h[k][sl], &b, 4);
                }
        }
        sd.bvh_objs.resize(built.order.size() + builtd.order.size());
        for (size_t k = 0; k < sd.bvh_objs.size(); k++) {
            const int32_t oi = k < built.order.size() ? built.order[k] : builtd.order[k - built.order.size()];
            std::memset(&sd.bvh_objs[k], 0, sizeof(BvhObj));
            sd.bvh_objs[k].o = w[(size_t)oi];
            sd.bvh_objs[k].index = oi;
        }
        F.bvh_root_exit = builtd.nodes.empty() ? -1 : node_off;
        F.bvh_root = sd.bvh_nodes.empty() || built.order.empty() ? -1 : 0;
    }
    F.n_bvh_nodes = (int32_t)sd.bvh_nodes.size();
    F.n_bvh_objs = (int32_t)sd.bvh_objs.size();
    F.world_in_lds = big ? 0 : 1;
    // LDS plan of the BVH path: per-lane stacks sized by the tree depth, and the first (top-level)
    // nodes of the main tree in what is left of a 40 KiB budget (4 blocks of 256 threads per CU)
    F.bvh_stack = big ? std::max(4, ((sd.bvh_stack_need + 1 + 3) / 4) * 4) : 0;
    const size_t stack_bytes = (size_t)F.bvh_stack * PT_BLOCK * sizeof(int);
    F.bvh_lds_nodes = 0;
    F.bvh_min_lanes = 24;
    F.debug_drop = 0;  // PTCORE_DEBUG_DROP=<mask>: the verify instantiations of the bitmask scans lose these candidate bits
    if (const char *e = std::getenv("PTCORE_DEBUG_DROP")) F.debug_drop = (uint32_t)std::strtoul(e, nullptr, 0);
    if (const char *e = std::getenv("PTCORE_BVH_MIN_LANES")) F.bvh_min_lanes = std::max(0, std::min(64, std::atoi(e)));
    F.bvh_leaf_single = 1;
    if (const char *e = std::getenv("PTCORE_BVH_LEAF_SINGLE")) F.bvh_leaf_single = std::atoi(e) != 0;
    F.bvh_node_min = 16;
    if (const char *e = std::getenv("PTCORE_BVH_NODE_MIN")) F.bvh_node_min = std::max(0, std::min(65, std::atoi(e)));
    size_t lds_budget = (size_t)(160 * 1024 / PT_BVH_WAVES);
    if (const char *e = std::getenv("PTCORE_BVH_LDS_BUDGET")) lds_budget = (size_t)std::max(0, std::min(160 * 1024, std::atoi(e)));
    if (big && F.bvh_root == 0 && stack_bytes < lds_budget)
        F.bvh_lds_nodes = (int32_t)std::min<size_t>((lds_budget - stack_bytes) / sizeof(BvhNode), (size_t)F.bvh_main_nodes);
    sd.lds_bytes = big ? stack_bytes + (size_t)F.bvh_lds_nodes * sizeof(BvhNode)
                       : (size_t)F.nobj * sizeof(DevObj) + (size_t)F.nmat * sizeof(DevMat) +
                             (size_t)(sd.bsph.size() + sd.bbox.size() + sd.bsph_diel.size() + sd.bbox_diel.size()) * sizeof(int);  // (+ the dielectric-only tables of FORM_NESTED)
    // single-group scans keep the records' objects a second time, in record order (trace_kernel: lds_rec)
    if (scan == ptk::SCAN_BROAD || scan == ptk::SCAN_VERIFY)
        sd.lds_bytes = ((sd.lds_bytes + 15) & ~(size_t)15) + (sd.bsph.size() + sd.bbox.size() + sd.bsph_diel.size() + sd.bbox_diel.size()) * sizeof(DevObj);
    sd.glass_lds_bytes = (size_t)F.nobj * sizeof(DevObj) + (size_t)F.nmat * sizeof(DevMat) +
                         (size_t)(sd.bsph_diel.size() + sd.bbox_diel.size()) * sizeof(int);
    if (scan == ptk::SCAN_BROAD || scan == ptk::SCAN_VERIFY)  // glass_kernel<*, *, false>: the dielectric records' objects in record order
        sd.glass_lds_bytes = ((sd.glass_lds_bytes + 15) & ~(size_t)15) + (sd.bsph_diel.size() + sd.bbox_diel.size()) * sizeof(DevObj);
    // PTCORE_BVH_LDS_PAD=<bytes>: unused LDS on top of the BVH plan (occupancy experiments: 4 blocks per CU fit 40 KiB each)
    if (const char *e = std::getenv("PTCORE_BVH_LDS_PAD"))
        if (big) sd.lds_bytes += (size_t)std::max(0, std::min(120 * 1024, std::atoi(e)));
    if (big && std::getenv("PTCORE_VERBOSE"))
        std::fprintf(stderr, "ptcore: BVH %d nodes (%d wide levels), %d objects, stack %d entries per lane, %d nodes in LDS, %zu B of LDS per block\n",
                     F.n_bvh_nodes, sd.bvh_depth, F.n_bvh_objs, F.bvh_stack, F.bvh_lds_nodes, sd.lds_bytes);
    if (sd.lds_bytes > 160 * 1024)
        return fail(PT_ERR_INVALID, "scene does not fit the 160 KiB LDS of a CU with this scan strategy (use the BVH: unset PTCORE_SCAN)");
    sd.has_glass = false;
    for (const DevObj &o : sd.world) sd.has_glass = sd.has_glass || (o.kind & 0x100);
    sd.gen++;
    sd.valid = true;
    return PT_OK;
}

int32_t frame_open(pt_ctx *ctx, const pt_scene *scene, const pt_config *cfg, uint32_t max_slots) {
    if (int32_t rc = scene_prepare(ctx, scene)) return rc;
    const SceneData &sd = ctx->sd;
    Frame &fr = ctx->frame;
    fr = Frame();
    fr.cfg = *cfg;
    fr.nobj = sd.Fs.nobj;
    fr.nmat = sd.Fs.nmat;
    fr.scan = sd.scan;
    fr.lds_bytes = sd.lds_bytes;
    fr.glass_lds_bytes = sd.glass_lds_bytes;
    fr.has_glass = sd.has_glass;
    // split passes: the bitmask scan of reference-sized scenes; a path has at most max_depth dielectric bounces
    fr.split_rounds = 0;
    if ((sd.scan == ptk::SCAN_BROAD || sd.scan == ptk::SCAN_VERIFY || sd.scan == ptk::SCAN_BROAD_WIDE || sd.scan == ptk::SCAN_VERIFY_WIDE) &&
        cfg->max_depth > 0)
        fr.split_rounds = fr.has_glass ? std::max(0, std::min(ctx->split_rounds, cfg->max_depth)) : (ctx->split_rounds > 0 ? 1 : 0);
    {
        const bool bitmask = sd.scan == ptk::SCAN_BROAD || sd.scan == ptk::SCAN_VERIFY || sd.scan == ptk::SCAN_BROAD_WIDE || sd.scan == ptk::SCAN_VERIFY_WIDE;
        fr.tail_form = (bitmask && fr.has_glass && ctx->tail_nested && !ctx->profile_sections) ? ptk::FORM_NESTED : ptk::FORM_ALL_IN_ONE;
    }
    {  // the wavefront form: on request (PTCORE_PIPELINE=wavefront), for the scans that have a pass form (bitmask, BVH).
       // Measured slower than the all-in-one loop in every regime, so it is the A/B, not the default.
        const bool bvh = sd.scan == ptk::SCAN_BVH || sd.scan == ptk::SCAN_VERIFY_BVH;
        const bool flat = sd.scan == ptk::SCAN_BROAD || sd.scan == ptk::SCAN_VERIFY;
        fr.wavefront = !ctx->profile_sections && ((ctx->pipeline == 1 && (bvh || flat)) || (ctx->pipeline == 2 && bvh));
        fr.walk32 = fr.wavefront && ctx->pipeline == 2;
        if (fr.wavefront) { fr.split_rounds = 0; fr.tail_form = ptk::FORM_ALL_IN_ONE; }
        fr.primary_pass = bvh && !fr.wavefront && !ctx->profile_sections && ctx->primary_coop && sd.Fs.bvh_root >= 0;
        fr.shade_lds_bytes = (size_t)sd.Fs.nmat * sizeof(DevMat) + (sd.Fs.world_in_lds ? (size_t)sd.Fs.nobj * sizeof(DevObj) : 0);
    }
    fr.cam = new_camera(scene->camera, cfg->width, cfg->height);
    fr.sky = make_sky(scene->sky);
    fr.ntx = (cfg->width + 31) / 32;
    fr.nty = (cfg->height + 31) / 32;
    fr.stats_on = (cfg->flags & PT_FLAG_PIXEL_STATS) != 0;
    DevFrame &F = fr.F;
    F = sd.Fs;
    F.width = cfg->width;
    F.height = cfg->height;
    F.max_depth = cfg->max_depth;
    F.ntx = fr.ntx;
    F.nty = fr.nty;
    F.seed_key = ptm::seed_key(cfg->seed);
    F.inv_width = 1.0 / (double)(cfg->width - 1);
    F.inv_height = 1.0 / (double)(cfg->height - 1);
    F.height_m1 = (double)(cfg->height - 1);
    // chunk of samples per pass: bounded by the L budget and by 2^31 jobs
    uint32_t chunk = cfg->spp_chunk > 0 ? (uint32_t)cfg->spp_chunk : 0;
    const uint32_t slots = std::max(1u, max_slots);
    // per job: 32 B radiance record + 58 B primary ray, and with split passes two path-state queues of 100 B per entry
    const size_t job_bytes = 90 + (fr.wavefront ? 330 + (fr.walk32 ? 4 * (PT_CAND_MAX + 2) : 0) : fr.split_rounds > 0 && fr.has_glass ? 220 : fr.primary_pass ? 110 : 0);
    fr.budget_bytes = ctx->l_budget_bytes;
    if (ctx->auto_grow && ctx->grown_budget_bytes > fr.budget_bytes && !ctx->devs.empty()) {
        const bool same_shape = cfg->width == ctx->last_w && cfg->height == ctx->last_h && cfg->samples_per_px == ctx->last_spp;
        if (same_shape && !ctx->grown && (size_t)slots * job_bytes * (size_t)std::max(1, cfg->samples_per_px) > fr.budget_bytes) {
            // the second frame of this shape, and it takes more than one pass: is the room there?  (free memory + what this context
            // already holds must cover the grown size with a tenth to spare, on the first device -- the others are its twins)
            size_t free_b = 0, total_b = 0;
            if (hipSetDevice(ctx->devs[0].ordinal) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
                free_b + dev_held_bytes(ctx->devs[0]) >= ctx->grown_budget_bytes + ctx->grown_budget_bytes / 10)
                ctx->grown = true;
            (void)hipGetLastError();
        }
        if (ctx->grown && same_shape) fr.budget_bytes = ctx->grown_budget_bytes;
    }
    ctx->last_w = cfg->width; ctx->last_h = cfg->height; ctx->last_spp = cfg->samples_per_px;
    if (chunk == 0) chunk = (uint32_t)std::max<size_t>(1, fr.budget_bytes / ((size_t)slots * job_bytes));
    chunk = std::min<uint32_t>(chunk, (uint32_t)std::max(1, cfg->samples_per_px));
    chunk = std::min<uint32_t>(chunk, std::max(1u, 0x7fffffffu / slots));
    fr.chunk = chunk;
    if (cfg->spp_chunk <= 0) balance_chunk(fr);
    fr.done_spp = 0;
    fr.t0 = std::chrono::steady_clock::now();
    fr.open = true;
    return PT_OK;
}

void fill_stats_common(pt_ctx *ctx, pt_stats *st) {
    Frame &fr = ctx->frame;
    st->spp_chunk = (int32_t)fr.chunk;
    st->num_devices = (int32_t)ctx->devs.size();
    st->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - fr.t0).count();
}

}  // namespace

// ==================================================================== C ABI

namespace {

// Loads librccl.so and makes one communicator per device of the context (one process, ncclCommInitAll).  RCCL refuses a
// device list that names one GPU twice -- the "virtual devices" of the tests -- and so does this.
int32_t rccl_open(pt_ctx *ctx) {
    pt_ctx::Rccl &R = ctx->rccl;
    const char *path = std::getenv("PTCORE_RCCL_LIB");
    // A process that already has an RCCL (PyTorch brings its own, soname librccl.so.1) must not get a second copy of it: ask for
    // the loaded one first.
    void *lib = path ? dlopen(path, RTLD_NOW | RTLD_LOCAL) : dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!lib && !path) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib && !path) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib && !path) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(PT_ERR_HIP, std::string("PTCORE_GATHER=rccl: cannot load librccl.so: ") + dlerror());
#define PT_RCCL_SYM(field, name)                                                              \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(lib, name));                          \
    if (!R.field) return fail(PT_ERR_HIP, std::string("PTCORE_GATHER=rccl: librccl.so has no ") + name);
    PT_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    PT_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    PT_RCCL_SYM(GroupStart, "ncclGroupStart")
    PT_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    PT_RCCL_SYM(Send, "ncclSend")
    PT_RCCL_SYM(Recv, "ncclRecv")
    PT_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef PT_RCCL_SYM
    std::vector<int> devs;
    for (const Device &d : ctx->devs) devs.push_back(d.ordinal);
    R.comms.assign(devs.size(), nullptr);
    const ncclResult_t r = R.CommInitAll(R.comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) {
        R.comms.clear();
        return fail(PT_ERR_HIP, std::string("PTCORE_GATHER=rccl: ncclCommInitAll: ") + R.GetErrorString(r));
    }
    R.lib = lib;
    return PT_OK;
}

#define RCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        const ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return fail(PT_ERR_HIP, std::string(#expr ": ") + ctx->rccl.GetErrorString(r_)); \
    } while (0)

}  // namespace

extern "C" {

int32_t pt_abi_version(void) { return PT_ABI_VERSION; }

const char *pt_last_error(void) { return g_last_error.c_str(); }

int32_t pt_device_count(int32_t *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    if (e != hipSuccess || n <= 0) return fail(PT_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    return PT_OK;
}

int32_t pt_create(const int32_t *devices, int32_t ndev, pt_ctx **out) {
    if (!out) return fail(PT_ERR_INVALID, "out is null");
    *out = nullptr;
    if (ndev <= 0 || ndev > 64) return fail(PT_ERR_INVALID, "ndev must be in 1..64");
    int32_t n = 0;
    if (int32_t rc = pt_device_count(&n)) return rc;
    pt_ctx *ctx = new pt_ctx();
    if (const char *e = std::getenv("PTCORE_L_BUDGET_MB")) {
        long mb = std::atol(e);
        if (mb > 0) { ctx->l_budget_bytes = (size_t)mb << 20; ctx->auto_grow = false; }
    }
    if (const char *e = std::getenv("PTCORE_AUTO_GROW")) ctx->auto_grow = std::atoi(e) != 0;
    if (const char *e = std::getenv("PTCORE_CLAIM")) {
        long c = std::atol(e);
        if (c >= 64 && c % 64 == 0) ctx->claim = (uint32_t)c;
    }
    if (const char *e = std::getenv("PTCORE_SCAN")) {
        if (!std::strcmp(e, "uniform")) ctx->scan_mode = ptk::SCAN_UNIFORM;
        else if (!std::strcmp(e, "verify")) ctx->scan_mode = ptk::SCAN_VERIFY;
        else if (!std::strcmp(e, "wide")) ctx->scan_mode = ptk::SCAN_BROAD_WIDE;
        else if (!std::strcmp(e, "verify_wide")) ctx->scan_mode = ptk::SCAN_VERIFY_WIDE;
        else if (!std::strcmp(e, "bvh")) ctx->scan_mode = ptk::SCAN_BVH;
        else if (!std::strcmp(e, "verify_bvh")) ctx->scan_mode = ptk::SCAN_VERIFY_BVH;
        else ctx->scan_mode = -1;
    }
    if (const char *e = std::getenv("PTCORE_PROFILE")) ctx->profile_sections = std::atoi(e) != 0;
    if (const char *e = std::getenv("PTCORE_SPLIT_ROUNDS")) ctx->split_rounds = std::max(0, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("PTCORE_TAIL")) ctx->tail_nested = std::strcmp(e, "trip") != 0;
    if (const char *e = std::getenv("PTCORE_PRIMARY")) ctx->primary_coop = std::strcmp(e, "lane") != 0;
    if (const char *e = std::getenv("PTCORE_PIPELINE")) ctx->pipeline = !std::strcmp(e, "wavefront") ? 1 : !std::strcmp(e, "walk32") ? 2 : !std::strcmp(e, "mega") ? 0 : -1;
    if (const char *e = std::getenv("PTCORE_WF_MIN_LANES")) ctx->wf_min_lanes = std::max(1, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("PTCORE_WF_SORT")) ctx->wf_sort = std::atoi(e);
    if (const char *e = std::getenv("PTCORE_BLOCKS_PER_CU")) {
        long c = std::atol(e);
        if (c >= 1 && c <= 8) ctx->max_blocks_per_cu = (int)c;
    }
    ctx->devs.resize((size_t)ndev);
    for (int i = 0; i < ndev; i++) {
        Device &d = ctx->devs[(size_t)i];
        d.ordinal = devices ? devices[i] : i;
        if (d.ordinal < 0 || d.ordinal >= n) {
            pt_destroy(ctx);
            return fail(PT_ERR_INVALID, "device ordinal out of range");
        }
        hipError_t e = hipSetDevice(d.ordinal);
        hipDeviceProp_t prop;
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, d.ordinal);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.own_stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            pt_destroy(ctx);
            return fail(PT_ERR_HIP, std::string("device init: ") + hipGetErrorString(e));
        }
        d.num_cu = prop.multiProcessorCount;
        d.stream = d.own_stream;
    }
    // tile gather goes device -> devices[0] by peer DMA over xGMI: enable direct access where the
    // topology offers it (without it hipMemcpyPeerAsync still works, staged through the host)
    for (int i = 1; i < ndev; i++) {
        const int a = ctx->devs[(size_t)i].ordinal, b = ctx->devs[0].ordinal;
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can && hipSetDevice(a) == hipSuccess) {
            hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess) (void)hipGetLastError();  // already enabled or refused: the copy path copes
        }
    }
    if (const char *e = std::getenv("PTCORE_GATHER")) {
        if (!std::strcmp(e, "rccl")) {
            if (int32_t rc = rccl_open(ctx)) {
                pt_destroy(ctx);
                return rc;
            }
        } else if (std::strcmp(e, "peer") != 0) {
            pt_destroy(ctx);
            return fail(PT_ERR_INVALID, "PTCORE_GATHER must be peer or rccl");
        }
    }
    *out = ctx;
    return PT_OK;
}

int32_t pt_debug_gather_mode(pt_ctx *ctx) { return ctx && ctx->rccl.lib ? 1 : 0; }

void pt_destroy(pt_ctx *ctx) {
    if (!ctx) return;
    if (ctx->rccl.lib) {
        for (size_t i = 0; i < ctx->rccl.comms.size(); i++)
            if (ctx->rccl.comms[i] && hipSetDevice(ctx->devs[i].ordinal) == hipSuccess) (void)ctx->rccl.CommDestroy(ctx->rccl.comms[i]);
        ctx->rccl.comms.clear();
        // (the library stays mapped: RCCL keeps threads and device state of its own that a dlclose would pull away under them)
        ctx->rccl.lib = nullptr;
    }
    for (Device &d : ctx->devs) {
        if (hipSetDevice(d.ordinal) != hipSuccess) continue;
        if (d.own_stream) (void)hipStreamSynchronize(d.own_stream);
        d.ray.release(); d.ray_rng.release(); d.ray_ndraw.release();
        d.objs.release(); d.mats.release(); d.bsph.release(); d.bbox.release(); d.plane_idx.release(); d.bvh_nodes.release(); d.bvh_objs.release(); d.bvh_cores.release(); d.L.release(); d.job_seg.release(); d.job_draw.release();
        d.prof.release();
        d.bsph_diel.release(); d.bbox_diel.release();
        d.gq_d.release(); d.cq_d.release(); d.gq_rs.release(); d.cq_rs.release(); d.gq_u32.release(); d.cq_u32.release();
        d.xq_d.release(); d.xq_rs.release(); d.xq_u32.release();
        d.wf_perm.release(); d.wf_key.release(); d.wf_bins.release();
        d.cand_ids.release(); d.cand_n.release(); d.slow_list.release();
        for (EventPair &e : d.ev_glass) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        d.acc.release(); d.acc_seg.release(); d.acc_draw.release(); d.tiles_rgba.release();
        d.tiles_accum.release(); d.tiles_seg.release(); d.tiles_draw.release(); d.queue.release();
        d.counters.release();
        for (EventPair &e : d.ev_trace) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (EventPair &e : d.ev_resolve) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (EventPair &e : d.ev_raygen) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        if (d.ev_first) (void)hipEventDestroy(d.ev_first);
        if (d.ev_last) (void)hipEventDestroy(d.ev_last);
        if (d.own_stream) (void)hipStreamDestroy(d.own_stream);
    }
    if (!ctx->devs.empty() && hipSetDevice(ctx->devs[0].ordinal) == hipSuccess) {
        ctx->g_tiles_rgba.release(); ctx->g_tiles_accum.release(); ctx->g_tiles_seg.release();
        ctx->g_tiles_draw.release(); ctx->f_rgba.release(); ctx->f_accum.release(); ctx->f_seg.release();
        ctx->f_draw.release();
    }
    delete ctx;
}

int32_t pt_post_process(pt_ctx *ctx, const pt_post_config *post, const double *accum, int32_t spp, uint8_t *rgba, int32_t stride,
                        int32_t width, int32_t height) {
    if (!ctx || !post || !rgba) return fail(PT_ERR_INVALID, "null argument");
    if (width <= 0 || height <= 0 || stride < width * 4) return fail(PT_ERR_INVALID, "bad frame size or stride");
    if (post->tonemap && !accum) return fail(PT_ERR_INVALID, "tonemap needs the accum buffer");
    if (ctx->frame.open) return fail(PT_ERR_STATE, "a frame is open");
    Device &d = ctx->devs[0];
    HIP_TRY(hipSetDevice(d.ordinal));
    const size_t npix = (size_t)width * (size_t)height;
    HIP_TRY(ctx->f_rgba.reserve(npix * 4));
    HIP_TRY(ctx->g_tiles_rgba.reserve(npix * 4));  // second frame-sized byte buffer (ping-pong)
    uint8_t *cur = ctx->f_rgba.p, *other = ctx->g_tiles_rgba.p;
    hipStream_t s = d.own_stream;
    const unsigned grid = (unsigned)((npix + PT_BLOCK - 1) / PT_BLOCK);
    if (post->tonemap) {
        HIP_TRY(ctx->f_accum.reserve(npix * 3));
        HIP_TRY(hipMemcpyAsync(ctx->f_accum.p, accum, npix * 3 * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(ptk::post_tonemap_kernel, dim3(grid), dim3(PT_BLOCK), 0, s, ctx->f_accum.p, spp, cur, (int32_t)npix);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemcpy2DAsync(cur, (size_t)width * 4, rgba, (size_t)stride, (size_t)width * 4, (size_t)height,
                                 hipMemcpyHostToDevice, s));
    }
    if (post->denoise && width > 2 && height > 2) {
        const double ss = post->sigma_s > 0 ? post->sigma_s : 1.0, sr = post->sigma_r > 0 ? post->sigma_r : 0.15;
        hipLaunchKernelGGL(ptk::post_bilateral_kernel, dim3(grid), dim3(PT_BLOCK), 0, s, cur, other, width, height, 2 * ss * ss,
                           2 * sr * sr);
        HIP_TRY(hipGetLastError());
        std::swap(cur, other);
    }
    if (post->smooth && width > 2 && height > 2 && post->smooth_radius > 0 && post->smooth_strength > 0) {
        const int32_t rad = std::max(1, std::min(5, post->smooth_radius));
        const double str = std::max(0.0, std::min(1.0, post->smooth_strength));
        hipLaunchKernelGGL(ptk::post_smooth_kernel, dim3(grid), dim3(PT_BLOCK), 0, s, cur, other, width, height, rad, str);
        HIP_TRY(hipGetLastError());
        std::swap(cur, other);
    }
    HIP_TRY(hipMemcpy2DAsync(rgba, (size_t)stride, cur, (size_t)width * 4, (size_t)width * 4, (size_t)height, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return PT_OK;
}

int32_t pt_debug_profile(pt_ctx *ctx, uint64_t *out, int32_t n) {
    if (!ctx || !out) return fail(PT_ERR_INVALID, "null argument");
    if (!ctx->profile_sections) return fail(PT_ERR_STATE, "set PTCORE_PROFILE=1 before pt_create");
    for (int32_t i = 0; i < n && i < 3 * ptk::SEC_COUNT; i++) out[i] = g_profile_scratch[i];
    return 3 * ptk::SEC_COUNT;
}

// Host-only: builds the BVH of `scene` exactly as a render would and checks its invariants.
// out = {nodes, objects, depth, most slots used in a node, objects (or nodes) not reached exactly once,
// objects outside their slot's box, child boxes not inside the parent's box (or malformed slots), plane count}.
int32_t pt_debug_bvh_check(const pt_scene *scene, int32_t out[8]) {
    if (!scene || !out) return fail(PT_ERR_INVALID, "null argument");
    if (scene->num_materials < 0 || scene->num_objects < 0) return fail(PT_ERR_INVALID, "negative scene counts");
    std::vector<DevObj> world;
    std::vector<DevMat> mats;
    scene_to_world(*scene, world, mats);
    std::vector<int32_t> finite;
    double Bnd = 1.0;
    int planes = 0;
    for (size_t i = 0; i < world.size(); i++) {
        if ((world[i].kind & 0xff) == KIND_PLANE) { planes++; continue; }
        finite.push_back((int32_t)i);
        const ptbvh::Aabb bb = ptbvh::object_bounds(world[i]);
        for (int k = 0; k < 3; k++) Bnd = std::max(Bnd, std::max(std::fabs(bb.lo[k]), std::fabs(bb.hi[k])));
    }
    const double margin = Bnd * (1.0 / 4096.0);
    ptbvh::Built b = ptbvh::build(world, finite, margin);
    for (int lv = 16; b.stack_need >= PT_BVH_STACK && lv >= 0; lv -= 16) b = ptbvh::build(world, finite, margin, lv);
    std::vector<int> seen(world.size(), 0);
    int widest = 0, outside = 0, nested = 0;
    struct Item { int32_t node; double lo[3], hi[3]; };
    std::vector<Item> st;
    if (!b.nodes.empty()) {
        Item r;
        r.node = 0;
        for (int k = 0; k < 3; k++) { r.lo[k] = -INFINITY; r.hi[k] = INFINITY; }
        st.push_back(r);
    }
    std::vector<int> visits(b.nodes.size(), 0);
    while (!st.empty()) {
        const Item it = st.back();
        st.pop_back();
        const BvhNode &nd = b.nodes[(size_t)it.node];
        visits[(size_t)it.node]++;
        const uint32_t intm = (nd.meta >> 8) & 0xfu, objm = (nd.meta >> 12) & 0xfu;
        if (intm & objm) nested++;  // a slot is one or the other
        widest = std::max(widest, __builtin_popcount(intm | objm));
        for (int s = 0; s < 4; s++) {
            if (!((intm | objm) & (1u << s))) continue;
            const int rank = (int)((nd.meta >> (2 * s)) & 3u);
            // (centre / half-extent boxes are rounded one by one: a child's may stick out of its parent's by a few ulps, which
            // the walk does not care about -- every box holds its own content, that is all it relies on)
            for (int k = 0; k < 3; k++) {
                const double slo = bvh_slot_lo(nd, k, s), shi = bvh_slot_hi(nd, k, s);
                // (centre / half-extent boxes are rounded one by one: a child's may stick out of its parent's by a few ulps, which
                // the walk does not care about -- every box holds its own content, that is all it relies on)
                const double tol = 8.0 * 1.1920929e-7 * std::max(std::fabs(slo), std::fabs(shi));
                if (slo < it.lo[k] - tol || shi > it.hi[k] + tol) nested++;
            }
            if (intm & (1u << s)) {
                Item c;
                c.node = bvh_node_base(nd) + rank;
                for (int k = 0; k < 3; k++) { c.lo[k] = bvh_slot_lo(nd, k, s); c.hi[k] = bvh_slot_hi(nd, k, s); }
                if (c.node <= it.node || c.node >= (int32_t)b.nodes.size()) { nested++; continue; }
                st.push_back(c);
            } else {
                const int32_t slot = bvh_obj_base(nd) + rank;
                if (slot < 0 || slot >= (int32_t)b.order.size()) { outside++; continue; }
                const int32_t oi = b.order[(size_t)slot];
                seen[(size_t)oi]++;
                const ptbvh::Aabb bb = ptbvh::object_bounds(world[(size_t)oi]);
                for (int a = 0; a < 3; a++)
                    if (bvh_slot_lo(nd, a, s) > bb.lo[a] - margin * 0.999 || bvh_slot_hi(nd, a, s) < bb.hi[a] + margin * 0.999) { outside++; break; }
            }
        }
    }
    int bad = 0;
    for (int32_t i : finite)
        if (seen[(size_t)i] != 1) bad++;
    for (int v : visits)
        if (v != 1) bad++;
    const int largest = widest;
    if (b.stack_need >= PT_BVH_STACK) bad++;
    out[0] = (int32_t)b.nodes.size();
    out[1] = (int32_t)b.order.size();
    out[2] = b.depth;
    out[3] = largest;
    out[4] = bad;
    out[5] = outside;
    out[6] = nested;
    out[7] = planes;
    return PT_OK;
}

// Runs div_selftest_kernel on device 0 of the context: `millions` x 10^6 operand pairs; returns the number of pairs
// whose shared-reciprocal quotient differs from the IEEE division (must be 0), or a negative PT_ERR_* code.
int64_t pt_debug_div_selftest(pt_ctx *ctx, int32_t millions, uint64_t seed) {
    if (!ctx || millions <= 0) return -(int64_t)fail(PT_ERR_INVALID, "null context or no work");
    Device &d = ctx->devs[0];
    if (hipSetDevice(d.ordinal) != hipSuccess) return -(int64_t)fail(PT_ERR_HIP, "hipSetDevice");
    unsigned long long *out = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&out), 8) != hipSuccess) return -(int64_t)fail(PT_ERR_HIP, "hipMalloc");
    (void)hipMemsetAsync(out, 0, 8, d.own_stream);
    const uint32_t per_thread = 1000;
    const uint32_t blocks = (uint32_t)(((uint64_t)millions * 1000000ull + (uint64_t)per_thread * PT_BLOCK - 1) / ((uint64_t)per_thread * PT_BLOCK));
    hipLaunchKernelGGL(ptk::div_selftest_kernel, dim3(blocks), dim3(PT_BLOCK), 0, d.own_stream, (unsigned long long)seed, per_thread, out);
    unsigned long long bad = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, out, 8, hipMemcpyDeviceToHost, d.own_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d.own_stream);
    (void)hipFree(out);
    if (e != hipSuccess) return -(int64_t)fail(PT_ERR_HIP, std::string("div selftest: ") + hipGetErrorString(e));
    return (int64_t)bad;
}

int64_t pt_debug_scan_mismatches(pt_ctx *ctx) {
    (void)ctx;
    if (g_mismatches && std::getenv("PTCORE_VERBOSE")) {
        const unsigned long long *s = g_mismatch_sample;
        double v[8];
        for (int i = 0; i < 2; i++) std::memcpy(&v[i], &s[1 + i], 8);
        for (int i = 0; i < 6; i++) std::memcpy(&v[2 + i], &s[4 + i], 8);
        std::fprintf(stderr, "scan mismatch sample: culled best %d t %.17g | plain best %d t %.17g | mode %llu | o %.17g %.17g %.17g d %.17g %.17g %.17g\n",
                     (int)(s[0] >> 32), v[0], (int)(uint32_t)s[0], v[1], s[3], v[2], v[3], v[4], v[5], v[6], v[7]);
    }
    return (int64_t)g_mismatches;
}

int32_t pt_shard_tiles(int32_t width, int32_t height, const pt_shard *shard, int32_t *ntiles_local, int32_t *ntiles_x,
                       int32_t *ntiles_y) {
    if (width <= 0 || height <= 0) return fail(PT_ERR_INVALID, "width and height must be positive");
    pt_shard sh = shard ? *shard : pt_shard{0, 1};
    if (sh.count <= 0 || sh.index < 0 || sh.index >= sh.count) return fail(PT_ERR_INVALID, "bad shard");
    const int32_t ntx = (width + 31) / 32, nty = (height + 31) / 32;
    if (ntiles_x) *ntiles_x = ntx;
    if (ntiles_y) *ntiles_y = nty;
    if (ntiles_local) *ntiles_local = tiles_of_shard(ntx * nty, sh);
    return PT_OK;
}

int32_t pt_begin(pt_ctx *ctx, const pt_scene *scene, const pt_config *cfg) {
    if (!ctx) return fail(PT_ERR_INVALID, "ctx is null");
    if (int32_t rc = validate(scene, cfg)) return rc;
    if (ctx->frame.open) return fail(PT_ERR_STATE, "pt_begin: a frame is already open");
    const int32_t ndev = (int32_t)ctx->devs.size();
    const int32_t ntiles = ((cfg->width + 31) / 32) * ((cfg->height + 31) / 32);
    const uint32_t max_slots = (uint32_t)tiles_of_shard(ntiles, pt_shard{0, ndev}) * 1024u;
    if (int32_t rc = frame_open(ctx, scene, cfg, max_slots)) return rc;
    for (int32_t i = 0; i < ndev; i++) {
        if (int32_t rc = dev_begin(ctx, ctx->devs[(size_t)i], pt_shard{i, ndev}, nullptr)) {
            ctx->frame.open = false;
            return rc;
        }
    }
    return PT_OK;
}

int32_t pt_step(pt_ctx *ctx, int32_t nspp, int32_t *done_spp) {
    if (!ctx || !ctx->frame.open) return fail(PT_ERR_STATE, "pt_step without pt_begin");
    Frame &fr = ctx->frame;
    int32_t left = fr.cfg.samples_per_px - fr.done_spp;
    int32_t todo = std::max(0, std::min(nspp, left));
    while (todo > 0) {
        const uint32_t S = std::min<uint32_t>((uint32_t)todo, fr.chunk);
        for (Device &d : ctx->devs)
            if (int32_t rc = dev_step(ctx, d, (uint32_t)fr.done_spp, S)) return rc;
        fr.done_spp += (int32_t)S;
        todo -= (int32_t)S;
    }
    for (Device &d : ctx->devs) {
        HIP_TRY(hipSetDevice(d.ordinal));
        HIP_TRY(hipStreamSynchronize(d.stream));
    }
    if (done_spp) *done_spp = fr.done_spp;
    return PT_OK;
}

// gathers every device's tiles on device 0, untiles, copies to host
static int32_t read_frame(pt_ctx *ctx, uint8_t *rgba, int32_t stride, double *accum, uint32_t *nseg, uint32_t *ndraw) {
    Frame &fr = ctx->frame;
    const int32_t W = fr.cfg.width, H = fr.cfg.height;
    if (rgba && stride < W * 4) return fail(PT_ERR_INVALID, "stride smaller than 4*width");
    const int32_t ndev = (int32_t)ctx->devs.size();
    const size_t ntiles = (size_t)fr.ntx * (size_t)fr.nty;
    const bool want_stats = fr.stats_on && (nseg || ndraw);
    Device &d0 = ctx->devs[0];
    HIP_TRY(hipSetDevice(d0.ordinal));
    HIP_TRY(ctx->g_tiles_rgba.reserve(ntiles * 4096));
    if (accum) HIP_TRY(ctx->g_tiles_accum.reserve(ntiles * 1024 * 3));
    if (want_stats) {
        HIP_TRY(ctx->g_tiles_seg.reserve(ntiles * 1024));
        HIP_TRY(ctx->g_tiles_draw.reserve(ntiles * 1024));
    }
    const int32_t spp_done = fr.done_spp;
    size_t before = 0;
    for (int32_t i = 0; i < ndev; i++) {
        Device &d = ctx->devs[(size_t)i];
        if (d.nlocal == 0) continue;
        const size_t nt = (size_t)d.nlocal;
        if (i == 0 && !ctx->rccl.lib) {
            // device 0 resolves straight into the gather buffer
            if (int32_t rc = dev_finish(ctx, d, spp_done, ctx->g_tiles_rgba.p + before * 4096,
                                        accum ? ctx->g_tiles_accum.p + before * 3072 : nullptr,
                                        want_stats ? ctx->g_tiles_seg.p + before * 1024 : nullptr,
                                        want_stats ? ctx->g_tiles_draw.p + before * 1024 : nullptr))
                return rc;
        } else {
            HIP_TRY(hipSetDevice(d.ordinal));
            HIP_TRY(d.tiles_rgba.reserve(nt * 4096));
            if (accum) HIP_TRY(d.tiles_accum.reserve(nt * 3072));
            if (want_stats) {
                HIP_TRY(d.tiles_seg.reserve(nt * 1024));
                HIP_TRY(d.tiles_draw.reserve(nt * 1024));
            }
            if (int32_t rc = dev_finish(ctx, d, spp_done, d.tiles_rgba.p, accum ? d.tiles_accum.p : nullptr,
                                        want_stats ? d.tiles_seg.p : nullptr, want_stats ? d.tiles_draw.p : nullptr))
                return rc;
            if (ctx->rccl.lib) {  // the exchange itself follows the loop, all devices in one RCCL group
                before += nt;
                continue;
            }
            // gather over xGMI: peer DMA into device 0's buffer, ordered on the source stream
            HIP_TRY(hipMemcpyPeerAsync(ctx->g_tiles_rgba.p + before * 4096, d0.ordinal, d.tiles_rgba.p, d.ordinal,
                                       nt * 4096, d.stream));
            if (accum)
                HIP_TRY(hipMemcpyPeerAsync(ctx->g_tiles_accum.p + before * 3072, d0.ordinal, d.tiles_accum.p, d.ordinal,
                                           nt * 3072 * sizeof(double), d.stream));
            if (want_stats) {
                HIP_TRY(hipMemcpyPeerAsync(ctx->g_tiles_seg.p + before * 1024, d0.ordinal, d.tiles_seg.p, d.ordinal,
                                           nt * 1024 * sizeof(uint32_t), d.stream));
                HIP_TRY(hipMemcpyPeerAsync(ctx->g_tiles_draw.p + before * 1024, d0.ordinal, d.tiles_draw.p, d.ordinal,
                                           nt * 1024 * sizeof(uint32_t), d.stream));
            }
        }
        before += nt;
    }
    if (ctx->rccl.lib) {
        // RCCL gather of the per-tile framebuffers: every device (devices[0] too: its own share travels the same way) sends its
        // tiles to rank 0 on its stream, rank 0 posts the matching receives on its stream -- one group, so that the sends and
        // receives of this one process pair up without deadlock (ncclGather is this same pattern; the counts differ per rank here)
        const pt_ctx::Rccl &R = ctx->rccl;
        RCCL_TRY(R.GroupStart());
        size_t off = 0;
        for (int32_t i = 0; i < ndev; i++) {
            Device &d = ctx->devs[(size_t)i];
            if (d.nlocal == 0) continue;
            const size_t nt = (size_t)d.nlocal;
            RCCL_TRY(R.Send(d.tiles_rgba.p, nt * 4096, ncclUint8, 0, R.comms[(size_t)i], d.stream));
            RCCL_TRY(R.Recv(ctx->g_tiles_rgba.p + off * 4096, nt * 4096, ncclUint8, i, R.comms[0], d0.stream));
            if (accum) {
                RCCL_TRY(R.Send(d.tiles_accum.p, nt * 3072, ncclDouble, 0, R.comms[(size_t)i], d.stream));
                RCCL_TRY(R.Recv(ctx->g_tiles_accum.p + off * 3072, nt * 3072, ncclDouble, i, R.comms[0], d0.stream));
            }
            if (want_stats) {
                RCCL_TRY(R.Send(d.tiles_seg.p, nt * 1024, ncclUint32, 0, R.comms[(size_t)i], d.stream));
                RCCL_TRY(R.Recv(ctx->g_tiles_seg.p + off * 1024, nt * 1024, ncclUint32, i, R.comms[0], d0.stream));
                RCCL_TRY(R.Send(d.tiles_draw.p, nt * 1024, ncclUint32, 0, R.comms[(size_t)i], d.stream));
                RCCL_TRY(R.Recv(ctx->g_tiles_draw.p + off * 1024, nt * 1024, ncclUint32, i, R.comms[0], d0.stream));
            }
            off += nt;
        }
        RCCL_TRY(R.GroupEnd());
        ctx->rccl.gathers++;
    }
    for (int32_t i = 1; i < ndev; i++) {
        HIP_TRY(hipSetDevice(ctx->devs[(size_t)i].ordinal));
        HIP_TRY(hipStreamSynchronize(ctx->devs[(size_t)i].stream));
    }
    HIP_TRY(hipSetDevice(d0.ordinal));
    HIP_TRY(ctx->f_rgba.reserve((size_t)W * H * 4));
    if (accum) HIP_TRY(ctx->f_accum.reserve((size_t)W * H * 3));
    if (want_stats) {
        HIP_TRY(ctx->f_seg.reserve((size_t)W * H));
        HIP_TRY(ctx->f_draw.reserve((size_t)W * H));
    }
    ptk::UntileArgs U;
    std::memset(&U, 0, sizeof U);
    U.tiles_rgba = ctx->g_tiles_rgba.p;
    U.tiles_accum = accum ? ctx->g_tiles_accum.p : nullptr;
    U.tiles_u32a = want_stats ? ctx->g_tiles_seg.p : nullptr;
    U.tiles_u32b = want_stats ? ctx->g_tiles_draw.p : nullptr;
    U.rgba = ctx->f_rgba.p;
    U.accum = accum ? ctx->f_accum.p : nullptr;
    U.u32a = want_stats ? ctx->f_seg.p : nullptr;
    U.u32b = want_stats ? ctx->f_draw.p : nullptr;
    U.width = W; U.height = H; U.ntx = fr.ntx; U.nty = fr.nty; U.stride = W * 4; U.shard_count = ndev;
    hipLaunchKernelGGL(ptk::untile_kernel, dim3((unsigned)fr.ntx, (unsigned)fr.nty, 4), dim3(PT_BLOCK), 0, d0.stream, U);
    HIP_TRY(hipGetLastError());
    if (rgba)
        HIP_TRY(hipMemcpy2DAsync(rgba, (size_t)stride, ctx->f_rgba.p, (size_t)W * 4, (size_t)W * 4, (size_t)H,
                                 hipMemcpyDeviceToHost, d0.stream));
    if (accum)
        HIP_TRY(hipMemcpyAsync(accum, ctx->f_accum.p, (size_t)W * H * 3 * sizeof(double), hipMemcpyDeviceToHost, d0.stream));
    if (want_stats && nseg)
        HIP_TRY(hipMemcpyAsync(nseg, ctx->f_seg.p, (size_t)W * H * sizeof(uint32_t), hipMemcpyDeviceToHost, d0.stream));
    if (want_stats && ndraw)
        HIP_TRY(hipMemcpyAsync(ndraw, ctx->f_draw.p, (size_t)W * H * sizeof(uint32_t), hipMemcpyDeviceToHost, d0.stream));
    HIP_TRY(hipStreamSynchronize(d0.stream));
    return PT_OK;
}

int32_t pt_read(pt_ctx *ctx, uint8_t *rgba, int32_t stride, double *accum) {
    if (!ctx || !ctx->frame.open) return fail(PT_ERR_STATE, "pt_read without pt_begin");
    return read_frame(ctx, rgba, stride, accum, nullptr, nullptr);
}

int32_t pt_end(pt_ctx *ctx, pt_stats *stats) {
    if (!ctx || !ctx->frame.open) return fail(PT_ERR_STATE, "pt_end without pt_begin");
    pt_stats st;
    std::memset(&st, 0, sizeof st);
    int32_t rc = PT_OK;
    int slot = 0;
    for (Device &d : ctx->devs) {
        if (!d.first_recorded || d.nlocal == 0) { slot++; continue; }
        // make sure ev_last exists even if pt_read was never called
        if (hipSetDevice(d.ordinal) == hipSuccess) (void)hipEventRecord(d.ev_last, d.stream);
        if (int32_t r = dev_collect(d, &st, slot)) rc = r;
        slot++;
    }
    fill_stats_common(ctx, &st);
    ctx->frame.open = false;
    if (stats) *stats = st;
    return rc;
}

int32_t pt_render(pt_ctx *ctx, const pt_scene *scene, const pt_config *cfg, uint8_t *rgba, int32_t stride, double *accum,
                  uint32_t *nseg, uint32_t *ndraw, pt_stats *stats) {
    if (!ctx) return fail(PT_ERR_INVALID, "ctx is null");
    if ((nseg || ndraw) && cfg && !(cfg->flags & PT_FLAG_PIXEL_STATS))
        return fail(PT_ERR_INVALID, "nseg/ndraw need PT_FLAG_PIXEL_STATS");
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    if (int32_t rc = pt_begin(ctx, scene, cfg)) return rc;
    const auto t1 = clk::now();
    int32_t rc = pt_step(ctx, cfg->samples_per_px, nullptr);
    const auto t2 = clk::now();
    if (rc == PT_OK) rc = read_frame(ctx, rgba, stride, accum, nseg, ndraw);
    const auto t3 = clk::now();
    pt_stats st;
    int32_t rc2 = pt_end(ctx, &st);
    if (std::getenv("PTCORE_VERBOSE")) {
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "ptcore: pt_render host phases: begin %.2f ms, step %.2f ms, read %.2f ms, end %.2f ms\n", ms(t0, t1), ms(t1, t2),
                     ms(t2, t3), ms(t3, clk::now()));
    }
    if (stats) *stats = st;
    return rc != PT_OK ? rc : rc2;
}

int32_t pt_render_tiles_device(pt_ctx *ctx, const pt_scene *scene, const pt_config *cfg, const pt_shard *shard,
                               void *d_tiles_rgba, void *d_tiles_accum, void *stream, pt_stats *stats) {
    if (!ctx) return fail(PT_ERR_INVALID, "ctx is null");
    if (int32_t rc = validate(scene, cfg)) return rc;
    if (ctx->frame.open) return fail(PT_ERR_STATE, "a frame is already open");
    if (!d_tiles_rgba) return fail(PT_ERR_INVALID, "d_tiles_rgba is null");
    pt_shard sh = shard ? *shard : pt_shard{0, 1};
    if (sh.count <= 0 || sh.index < 0 || sh.index >= sh.count) return fail(PT_ERR_INVALID, "bad shard");
    if (cfg->flags & PT_FLAG_PIXEL_STATS) return fail(PT_ERR_INVALID, "pixel stats are not available on the device entry point");
    const int32_t ntiles = ((cfg->width + 31) / 32) * ((cfg->height + 31) / 32);
    const uint32_t slots = (uint32_t)tiles_of_shard(ntiles, sh) * 1024u;
    if (int32_t rc = frame_open(ctx, scene, cfg, slots)) return rc;
    Frame &fr = ctx->frame;
    Device &d = ctx->devs[0];
    int32_t rc = dev_begin(ctx, d, sh, static_cast<hipStream_t>(stream));
    for (int32_t s = 0; rc == PT_OK && s < cfg->samples_per_px;) {
        const uint32_t S = std::min<uint32_t>((uint32_t)(cfg->samples_per_px - s), fr.chunk);
        rc = dev_step(ctx, d, (uint32_t)s, S);
        s += (int32_t)S;
    }
    fr.done_spp = cfg->samples_per_px;
    if (rc == PT_OK)
        rc = dev_finish(ctx, d, cfg->samples_per_px, static_cast<uint8_t *>(d_tiles_rgba), static_cast<double *>(d_tiles_accum),
                        nullptr, nullptr);
    if (rc == PT_OK && stats) {
        pt_stats st;
        std::memset(&st, 0, sizeof st);
        rc = dev_collect(d, &st, 0);
        fill_stats_common(ctx, &st);
        st.num_devices = 1;
        *stats = st;
    }
    fr.open = false;
    return rc;
}

int32_t pt_untile_device(pt_ctx *ctx, int32_t width, int32_t height, int32_t shard_count, int32_t shard_stride_tiles,
                         const void *d_tiles_rgba, const void *d_tiles_accum, void *d_rgba, int32_t stride, void *d_accum,
                         void *stream) {
    if (!ctx) return fail(PT_ERR_INVALID, "ctx is null");
    if (width <= 0 || height <= 0 || shard_count <= 0) return fail(PT_ERR_INVALID, "bad frame or shard count");
    {
        const int32_t nt = ((width + 31) / 32) * ((height + 31) / 32);
        if (shard_stride_tiles != 0 && shard_stride_tiles < tiles_of_shard(nt, pt_shard{0, shard_count}))
            return fail(PT_ERR_INVALID, "shard_stride_tiles smaller than the largest shard");
    }
    if (d_rgba && (stride < width * 4 || stride % 4 != 0)) return fail(PT_ERR_INVALID, "stride must be a multiple of 4 and >= 4*width");
    if (d_rgba && !d_tiles_rgba) return fail(PT_ERR_INVALID, "d_tiles_rgba is null");
    if (d_accum && !d_tiles_accum) return fail(PT_ERR_INVALID, "d_tiles_accum is null");
    Device &d = ctx->devs[0];
    HIP_TRY(hipSetDevice(d.ordinal));
    ptk::UntileArgs U;
    std::memset(&U, 0, sizeof U);
    U.tiles_rgba = static_cast<const uint8_t *>(d_tiles_rgba);
    U.tiles_accum = d_accum ? static_cast<const double *>(d_tiles_accum) : nullptr;
    U.rgba = static_cast<uint8_t *>(d_rgba);
    U.accum = static_cast<double *>(d_accum);
    U.width = width; U.height = height;
    U.ntx = (width + 31) / 32; U.nty = (height + 31) / 32;
    U.stride = stride; U.shard_count = shard_count; U.shard_stride_tiles = shard_stride_tiles;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : d.own_stream;
    hipLaunchKernelGGL(ptk::untile_kernel, dim3((unsigned)U.ntx, (unsigned)U.nty, 4), dim3(PT_BLOCK), 0, s, U);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

}  // extern "C"
