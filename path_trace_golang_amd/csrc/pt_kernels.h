// pt_kernels.h -- gfx950 kernels of libptcore (included by ptcore.hip only).
//
//   trace_kernel    persistent waves; every lane owns one path at a time and pulls the
//                   next (pixel, sample) job from a wave-level cursor that is refilled
//                   from one global queue (ballot + mbcnt prefix rank).  The closest-hit scan over the
//                   world (renderer.go:297-302; objects.go:37-222), shading
//                   (materials.go:74-224), the dielectric exit search (renderer.go:316-371)
//                   and Russian roulette (renderer.go:375-393) run in one loop whose
//                   scan section is executed by all 64 lanes together: the exit search
//                   of a glass hit is just that lane's next trip through the same scan.
//                   Two scan strategies produce the same winner bit for bit:
//                     SCAN_UNIFORM  every lane tests every object in FP64 (wave-uniform index,
//                                   scalar loads);
//                     SCAN_BROAD    an FP32 broad phase over conservative (inflated) bounds
//                                   builds a per-lane candidate bitmask, then each lane runs
//                                   the exact FP64 tests only on its own candidates (per-lane
//                                   look-ups in the LDS copy of the world).  Needs <= 32 spheres and <= 32 boxes.
//                     SCAN_BROAD_WIDE  the same in groups of 32 records, up to 128 spheres and 128 boxes.
//                     SCAN_BVH      larger scenes: per-lane traversal of a 4-wide BVH whose FP32
//                                   slot boxes are conservative (same inflation as the broad phase);
//                                   a slot is an internal node or one object, the exact FP64 test
//                                   runs on the objects whose own box is pierced; nodes and objects
//                                   are read from HBM/L2 (top of the tree from LDS), the stack is in LDS.
//                     SCAN_VERIFY*  run a culled strategy AND the plain scan, count disagreements.
//                   Split form (SPLIT): a dielectric hit is not shaded in the loop; the path is parked in a path-state
//                   queue in HBM and comes back through a continuation queue (see glass_kernel); the scan is then
//                   compiled for closest hits only.
//   glass_kernel    the dielectric bounce of every parked path, all lanes on the same branch: scatter
//                   (materials.go:162-200), exit search over the dielectric objects only (renderer.go:316-349), its
//                   epilogue (renderer.go:352-370), roulette (renderer.go:375-403); survivors -> continuation queue.
//   raygen_kernel / raygen_lens_kernel   one thread per job: stream init, pixel jitter and camera.getRay
//                   (camera.go:60-74, renderer.go:181-184) as a coherent pre-pass of every chunk; with a thin lens a
//                   lane walks a column of four jobs so that a wave does not wait for its unluckiest rejection loop.
//   resolve_kernel  per pixel slot, adds the chunk's sample radiances IN SAMPLE ORDER
//                   to the running sum (renderer.go:186), and on request finishes the
//                   pixel: 1/spp, sqrt gamma, *255.999, clamp, truncate (renderer.go:190-221).
//   untile_kernel   tile-major -> row-major frame.
//
// FP64 throughout; built with -ffp-contract=off so every product and sum rounds
// exactly as the reference's Go code does on amd64.
#pragma once

#include <hip/hip_runtime.h>

#include "pt_device.h"
#include "pt_math.h"

namespace ptk {

using namespace ptd;

typedef float v2f __attribute__((ext_vector_type(2)));  // operand pair of the packed FP32 instructions (v_pk_fma_f32)

#define PT_WAVE 64
#define PT_BLOCK 256
#define PT_QUEUE_BLOCK 256u     // glass-queue slots a wave of trace_kernel reserves per atomic
#define PT_CONT_BLOCK 1024u     // continuation slots a wave of glass_kernel reserves per atomic
#ifndef PT_BVH_WAVES
#define PT_BVH_WAVES 4  // blocks of 256 threads per CU (= waves per SIMD) the BVH kernels are compiled for
#endif
#ifndef PT_FLAT_WAVES
#define PT_FLAT_WAVES 5  // waves per SIMD the flat-scan trace kernels and glass_kernel are compiled for (A/B builds: -DPT_FLAT_WAVES=4|6)
#endif
#ifndef PT_SPLIT_WAVES
#define PT_SPLIT_WAVES 6  // ... and the headline kernel (split form of the bitmask scan) for: since the candidate masks are built by
                          // push_keep_bit it needs 87 registers; held to 80 it spills two and is 2.2 % faster at six waves than at five
                          // (profiles/r03_occ_c4.txt, second block; before that change six waves lost 2.3 %)
#endif
#ifndef PT_BROAD_UNROLL
#define PT_BROAD_UNROLL 4  // records per turn of the broad-phase loops of the single-group scan: all their scalar loads are issued before the first
                           // record is used.  Same box, C4: 530.6 / 526.8 / 524.4 ms per frame for 1 / 2 / 4 (profiles/r04_broad_unroll_ab.txt)
#endif
#ifndef PT_CLIP32
#define PT_CLIP32 1  // bitmask scans: the scene-cube clip in FP32 with an explicit error term (A/B: -DPT_CLIP32=0)
#endif
#ifndef PT_PLANE0
#define PT_PLANE0 1  // single-group scan: the scene's one plane from the argument block (A/B: -DPT_PLANE0=0)
#endif
#ifndef PT_NESTED_WAVES
#define PT_NESTED_WAVES PT_FLAT_WAVES  // ... and the pass behind the split rounds (FORM_NESTED)
#endif
#define PT_HOLE 0xffffffffu     // job id of a reserved but unused queue slot
// The host sizes every path-state queue as (entries a pass can append) + (waves of the widest writer grid) x (the larger
// window): queue_slack() in ptcore.hip.  What that arithmetic relies on:
static_assert(PT_QUEUE_BLOCK >= PT_WAVE && PT_CONT_BLOCK >= PT_WAVE, "a window must hold one push of a whole wave");
static_assert(PT_QUEUE_BLOCK <= PT_CONT_BLOCK, "queue_slack() prices every window at PT_CONT_BLOCK slots");
static_assert(PT_BLOCK % PT_WAVE == 0, "whole waves per block");

__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Outward normal of object `o` at surface point p (objects.go:70-73 sphere, :252 plane,
// :186-217 box nearest face with strict '<' in the order -x,+x,-y,+y,-z,+z).
template <typename ObjRef>
__device__ __forceinline__ void outward_normal(const ObjRef &o, int kind, double px, double py, double pz, double &nx,
                                               double &ny, double &nz) {
    if (kind == KIND_SPHERE) {
        nx = (px - o.a[0]) * o.inv_radius;
        ny = (py - o.a[1]) * o.inv_radius;
        nz = (pz - o.a[2]) * o.inv_radius;
    } else if (kind == KIND_PLANE) {
        nx = o.b[0];
        ny = o.b[1];
        nz = o.b[2];
    } else {
        double dxMin = px - o.a[0], dxMax = o.b[0] - px;
        double dyMin = py - o.a[1], dyMax = o.b[1] - py;
        double dzMin = pz - o.a[2], dzMax = o.b[2] - pz;
        double minDist = dxMin;
        nx = -1; ny = 0; nz = 0;
        if (dxMax < minDist) { minDist = dxMax; nx = 1; ny = 0; nz = 0; }
        if (dyMin < minDist) { minDist = dyMin; nx = 0; ny = -1; nz = 0; }
        if (dyMax < minDist) { minDist = dyMax; nx = 0; ny = 1; nz = 0; }
        if (dzMin < minDist) { minDist = dzMin; nx = 0; ny = 0; nz = -1; }
        if (dzMax < minDist) { nx = 0; ny = 0; nz = 1; }
    }
}

// reflectVec, math.go:39-46
__device__ __forceinline__ void reflect_vec(double vx, double vy, double vz, double nx, double ny, double nz, double &rx,
                                            double &ry, double &rz) {
    double dot = vx * nx + vy * ny + vz * nz;
    rx = vx - nx * 2 * dot;
    ry = vy - ny * 2 * dot;
    rz = vz - nz * 2 * dot;
}


// Section ids of the diagnostic build (PROF = true): per section the kernel counts wave
// executions, active lanes and shader-clock cycles (leader lane only).  The shipping
// instantiations have PROF = false and contain none of this.
enum { SEC_ITER = 0, SEC_RAYGEN /* loading the pre-generated ray */, SEC_HIST0 /* BVH histogram words */, SEC_SCAN, SEC_HIST1, SEC_HIST2,
       SEC_HITREC, SEC_COSINE,
       SEC_DIEL, SEC_EXITPOST, SEC_RR, SEC_FINISH, SEC_SKY, SEC_UNITDIR, SEC_BROAD, SEC_NSPH, SEC_NBOX, SEC_PLANE, SEC_COUNT };


enum { SCAN_UNIFORM = 0, SCAN_BROAD = 1, SCAN_VERIFY = 2, SCAN_BVH = 3, SCAN_VERIFY_BVH = 4, SCAN_BROAD_WIDE = 5, SCAN_VERIFY_WIDE = 6 };
#define PT_BVH_STACK 96  // upper bound of the per-lane stack (sized per scene from the tree)

// Diagnostic hooks handed to the scan routines (all no-ops unless PROF).
struct ProfHooks {
    uint32_t *exec;
    uint32_t *lanes;
    unsigned long long *cyc;
    uint32_t lane;
    unsigned long long *dbg;  // counters + 8 (diagnostic sample slot)
};
#define PH_BEGIN(id)                                                              \
    unsigned long long pht_##id = 0;                                              \
    bool phl_##id = false;                                                        \
    if (PROF) {                                                                   \
        const uint64_t m_ = __ballot(1);                                          \
        phl_##id = ph.lane == (uint32_t)(__ffsll((long long)m_) - 1);             \
        ph.lanes[id]++;                                                           \
        if (phl_##id) { ph.exec[id]++; pht_##id = __builtin_amdgcn_s_memtime(); } \
    }
#define PH_END(id) \
    if (PROF && phl_##id) ph.cyc[id] += __builtin_amdgcn_s_memtime() - pht_##id;


struct RayD {
    double ox, oy, oz, dx, dy, dz;
};

// math.Max / math.Min (materials.go:185, math.go:49, renderer.go:378,384) as ONE v_max_f64 / v_min_f64 plus the NaN rule.
// The instruction already does what Go's special cases ask for -- an infinity wins, max(+0, -0) = +0, min(+0, -0) = -0
// (CDNA ISA, V_MAX_F64 / V_MIN_F64) -- except for a NaN operand, which it drops and Go propagates (unless the other one is
// the winning infinity).  ptm::go_max / go_min spell the special cases out (15 instructions each); compared bit for bit by
// pt_debug_div_selftest, whose first run found the infinity-before-NaN order.
__device__ __forceinline__ double dev_go_max(double x, double y) {
    const double m = __builtin_fmax(x, y);
    // Go tests for +Inf before it tests for NaN: Max(+Inf, NaN) = +Inf -- which is what the instruction, dropping the NaN, returned
    return ((x != x || y != y) && !(m == ptm::inf_pos())) ? ptm::qnan() : m;
}
__device__ __forceinline__ double dev_go_min(double x, double y) {
    const double m = __builtin_fmin(x, y);
    return ((x != x || y != y) && !(m == -ptm::inf_pos())) ? ptm::qnan() : m;
}

// IEEE division n / d with the reciprocal work shared between several numerators.  The compiler's f64 division is
//   ds = div_scale(d), ns = div_scale(n); y = rcp(ds) refined by two Newton steps; q0 = ns*y; r = fma(-ds, q0, ns);
//   q = div_fmas(r, y, q0); div_fixup(q, d, n)
// where div_scale / div_fmas / div_fixup only act on operands near the ends of the exponent range, zeros, infinities
// and NaNs (CDNA ISA, V_DIV_SCALE_F64: exponent difference >= 768, denormal operand, reciprocal or quotient, numerator
// below 2^-970).  Away from those the quotient is exactly fma(fma(-d, n*y, n), y, n*y) with a y that depends on d alone --
// computed once per denominator here (v_rcp_f64 is 16 cycles, each fma 4).  The guard: with the denominator within
// 2^+-340 (the callers' `tame` rays, or a roulette probability) and the numerator within 2^+-300, none of the special
// cases can arise; a zero numerator and anything else are treated apart.  Checked bit for bit against `/`
// on 4*10^9 random and patterned operand pairs by pt_debug_div_selftest (tests/test_div_shared_gpu.py).
__device__ __forceinline__ double div_recip(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}
__device__ __forceinline__ double div_shared(double n, double d, double y) {
    const double q = n * y;
    const double r = __builtin_fma(-d, q, n);
    double f = __builtin_fma(r, y, q);
    const uint32_t ex = ((uint32_t)(ptm::to_bits(n) >> 52)) & 0x7ffu;
    if (!(ex - (1023u - 300u) <= 600u)) {  // numerator outside 2^+-300: rare
        // a zero keeps the sign the product n*y gave it (the residual step would turn -0 / d into +0); everything
        // else out here (huge, tiny, denormal, infinite, NaN) takes the plain division
        if (n == 0) f = q;
        else f = n / d;
    }
    return f;
}

// Exact sphere test, objects.go:37-61: first root (near, then far) that lies in [tmin, tmax].
__device__ __forceinline__ bool sphere_exact(double cx, double cy, double cz, double radius_sq, const RayD &r, double a,
                                             double tmin, double tmax, double &t) {
    const double ocx = r.ox - cx, ocy = r.oy - cy, ocz = r.oz - cz;
    const double halfB = ocx * r.dx + ocy * r.dy + ocz * r.dz;
    const double ocLenSq = ocx * ocx + ocy * ocy + ocz * ocz;
    const double c = ocLenSq - radius_sq;
    const double disc = halfB * halfB - a * c;
    bool valid = false;
    if (!(disc < 0)) {
        const double sq = ptm::f_sqrt(disc);
        double root = (-halfB - sq) / a;
        valid = true;
        if (root < tmin || root > tmax) {
            root = (-halfB + sq) / a;
            if (root < tmin || root > tmax) valid = false;
        }
        t = root;
    }
    return valid;
}

// The same with the two divisions by a = |d|^2 sharing ya = div_recip(a) (a within 2^+-340: the caller's `tame` test).
template <bool OUTLINE_SQRT = true>
__device__ __forceinline__ bool sphere_exact_shared(double cx, double cy, double cz, double radius_sq, const RayD &r, double a, double ya,
                                                    double tmin, double tmax, double &t) {
    const double ocx = r.ox - cx, ocy = r.oy - cy, ocz = r.oz - cz;
    const double halfB = ocx * r.dx + ocy * r.dy + ocz * r.dz;
    const double ocLenSq = ocx * ocx + ocy * ocy + ocz * ocz;
    const double c = ocLenSq - radius_sq;
    const double disc = halfB * halfB - a * c;
    bool valid = false;
    if (!(disc < 0)) {
        const double sq = ptm::f_sqrt<OUTLINE_SQRT>(disc);
        double root = div_shared(-halfB - sq, a, ya);
        valid = true;
        if (root < tmin || root > tmax) {
            root = div_shared(-halfB + sq, a, ya);
            if (root < tmin || root > tmax) valid = false;
        }
        t = root;
    }
    return valid;
}

// Exact slab test, objects.go:141-179.  t0 only grows and t1 only shrinks, so the per-axis early
// return of objects.go:176 equals the single test after the third axis.
// MINMAX: the updates as v_max_f64 / v_min_f64 (one 4-cycle instruction instead of a compare and two selects).  Equal to
// the compares whenever t0 and t1 start as numbers: a NaN slab parameter (0 * inf) is skipped by both forms, and a zero
// of either sign can only ever sit in t1, where only `t1 <= t0` with t0 >= tmin > 0 looks at it.  The plain scan keeps the
// compare form: there tmax can be the NaN root of a degenerate sphere, which `if (tf < t1)` leaves in place.
template <bool MINMAX = false>
__device__ __forceinline__ bool box_exact(double ax, double ay, double az, double bx, double by, double bz, const RayD &r,
                                          double ivx, double ivy, double ivz, double tmin, double tmax, double &t) {
    double t0 = tmin, t1 = tmax;
    if (MINMAX) {
        double tn = (ax - r.ox) * ivx, tf = (bx - r.ox) * ivx;
        if (ivx < 0) { const double s = tn; tn = tf; tf = s; }
        t0 = __builtin_fmax(t0, tn);
        t1 = __builtin_fmin(t1, tf);
        tn = (ay - r.oy) * ivy; tf = (by - r.oy) * ivy;
        if (ivy < 0) { const double s = tn; tn = tf; tf = s; }
        t0 = __builtin_fmax(t0, tn);
        t1 = __builtin_fmin(t1, tf);
        tn = (az - r.oz) * ivz; tf = (bz - r.oz) * ivz;
        if (ivz < 0) { const double s = tn; tn = tf; tf = s; }
        t0 = __builtin_fmax(t0, tn);
        t1 = __builtin_fmin(t1, tf);
        t = t0;
        return !(t1 <= t0);
    }
    double tn = (ax - r.ox) * ivx, tf = (bx - r.ox) * ivx;
    if (ivx < 0) { const double s = tn; tn = tf; tf = s; }
    if (tn > t0) t0 = tn;
    if (tf < t1) t1 = tf;
    tn = (ay - r.oy) * ivy; tf = (by - r.oy) * ivy;
    if (ivy < 0) { const double s = tn; tn = tf; tf = s; }
    if (tn > t0) t0 = tn;
    if (tf < t1) t1 = tf;
    tn = (az - r.oz) * ivz; tf = (bz - r.oz) * ivz;
    if (ivz < 0) { const double s = tn; tn = tf; tf = s; }
    if (tn > t0) t0 = tn;
    if (tf < t1) t1 = tf;
    t = t0;
    return !(t1 <= t0);
}

// Exact plane test, objects.go:98-112.
__device__ __forceinline__ bool plane_exact(double px, double py, double pz, double nx, double ny, double nz, const RayD &r,
                                            double tmin, double tmax, double &t) {
    const double denom = nx * r.dx + ny * r.dy + nz * r.dz;
    if (ptm::f_abs(denom) < 1e-6) return false;
    t = ((px - r.ox) * nx + (py - r.oy) * ny + (pz - r.oz) * nz) / denom;
    return !(t < tmin || t > tmax);
}

// The same for the only planes the engine builds, normal (0, 1, 0) (objects.go:252), and a ray with finite components
// (the culled scans' `tame` rays): n.d = (0*dx + 1*dy) + 0*dz is dy itself unless dy is a zero, which the 1e-6 test rejects
// whatever its sign, and the numerator is (py - oy) itself unless that is a zero, whose quotient falls below tMin whatever
// its sign.  Same decisions, same t, ten FP64 operations fewer per plane and scan.
__device__ __forceinline__ bool plane_exact_y(double py, const RayD &r, double tmin, double tmax, double &t) {
    const double denom = r.dy;
    if (ptm::f_abs(denom) < 1e-6) return false;
    t = (py - r.oy) / denom;
    return !(t < tmin || t > tmax);
}

// renderer.go:333-347 without the `t < exitT` part: glass back face at a sane distance from the
// entry point (which is the ray origin during an exit search).
template <typename ObjRef>
__device__ __forceinline__ bool exit_candidate_ok(const ObjRef &o, int kind, const RayD &r, double t) {
    const double px = r.ox + r.dx * t, py = r.oy + r.dy * t, pz = r.oz + r.dz * t;
    double nx, ny, nz;
    outward_normal(o, kind, px, py, pz, nx, ny, nz);
    const bool ff = (r.dx * nx + r.dy * ny + r.dz * nz) < 0;
    if (ff) return false;
    const double ex = px - r.ox, ey = py - r.oy, ez = pz - r.oz;
    const double distSq = ex * ex + ey * ey + ez * ez;
    return distSq > 1e-8 && distSq < 1000.0;
}

// The reference's closest-hit loop (renderer.go:297-302) and exit search (renderer.go:329-349),
// object by object in file order, every lane on the same object.
template <typename ObjPtr>
__device__ __forceinline__ void scan_uniform(const DevFrame &F, ObjPtr g_obj, const RayD &r, int mode, int &best,
                                             double &tmax) {
    const double tmin = mode ? 0.0001 : 0.001;  // renderer.go:322 / :292
    tmax = ptm::max_float64();
    best = -1;
    const double a = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;      // objects.go:43, same for every sphere
    const double ivx = 1 / r.dx, ivy = 1 / r.dy, ivz = 1 / r.dz;  // objects.go:149,154,159
    for (int i = 0; i < F.nobj; i++) {
        const auto &o = g_obj[i];
        const int kind = o.kind & 0xff;
        const bool diel = (o.kind & 0x100) != 0;
        if (mode != 0 && !diel) continue;  // only glass can end an exit search (renderer.go:333)
        bool valid;
        double t = 0;
        if (kind == KIND_SPHERE) valid = sphere_exact(o.a[0], o.a[1], o.a[2], o.radius_sq, r, a, tmin, tmax, t);
        else if (kind == KIND_BOX) valid = box_exact(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, ivx, ivy, ivz, tmin, tmax, t);
        else valid = plane_exact(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, tmin, tmax, t);
        if (valid) {
            if (mode == 0) {
                best = i;
                tmax = t;
            } else if (t < tmax && exit_candidate_ok(o, kind, r, t)) {
                best = i;
                tmax = t;
            }
        }
    }
}


// A ray against the cube [-Bs, Bs]^3 that holds every finite object, in FP64.
//   inside   : the origin is within 3.5 scene sizes (F.clip_bound): ts = 0, nothing else to do -- the FP32 tests are
//              conservative from any origin within 4 B, and most rays start inside the scene
//   miss     : the ray never is inside the cube at a parameter >= tmin  -> no sphere or box can be hit
//   te, ts   : entry parameter; the FP32 tests run from the entry point o + d*ts (parameters relative to ts)
//   far      : the origin is so far out (more than ~2000 scene sizes) that the REFERENCE's own FP64 sphere
//              test loses its meaning there: halfB*halfB - a*c cancels and carries an error of a few
//              2^-53 a |oc|^2, so it reports "hits" on spheres the line passes within
//              delta ~ 3e-8 |oc| of.  Culling by the true geometry would drop those bit-exact artefacts.
//              The bitmask strategy sends such rays through the plain every-object scan; the BVH widens
//              every bound by `infl` = 4 delta for them (and the cube here by the same), which keeps every
//              object the reference's test can accept.  The slab test of a box has no such cancellation.
// v_min/v_max skip the NaN of a 0 * inf slab boundary, which leaves that slab unconstrained: conservative.
struct Clip {
    double ts, te, infl;
    bool miss, far;
};
__device__ __forceinline__ Clip clip_ray(const DevFrame &F, const RayD &r, double tmin) {
    Clip c{0.0, 0.0, 0.0, false, false};
    double Bs = F.scene_bound;
    const double Cb = F.clip_bound;
    if (!(ptm::f_abs(r.ox) <= Cb && ptm::f_abs(r.oy) <= Cb && ptm::f_abs(r.oz) <= Cb)) {
        // v_rcp_f64 (relative error < 2^-26) instead of three IEEE divisions: the entry point may be off by
        // ~1.5e-8 * reach <= 3e-5 B, and the cube keeps B/512 - m = 1.7e-3 B of clearance around every bound
        const double ix = __builtin_amdgcn_rcp(r.dx), iy = __builtin_amdgcn_rcp(r.dy), iz = __builtin_amdgcn_rcp(r.dz);
        double te, tx;
        auto cube = [&]() {
            const double x0 = (-Bs - r.ox) * ix, x1 = (Bs - r.ox) * ix;
            const double y0 = (-Bs - r.oy) * iy, y1 = (Bs - r.oy) * iy;
            const double z0 = (-Bs - r.oz) * iz, z1 = (Bs - r.oz) * iz;
            te = __builtin_fmax(__builtin_fmax(__builtin_fmin(x0, x1), __builtin_fmin(y0, y1)), __builtin_fmin(z0, z1));
            tx = __builtin_fmin(__builtin_fmin(__builtin_fmax(x0, x1), __builtin_fmax(y0, y1)), __builtin_fmax(z0, z1));
        };
        cube();
        const double reach = ptm::f_abs(r.ox) + ptm::f_abs(r.oy) + ptm::f_abs(r.oz) +
                             (ptm::f_abs(r.dx) + ptm::f_abs(r.dy) + ptm::f_abs(r.dz)) * ptm::f_abs(te);
        // delta must stay far inside the margin m that every bound already has
        c.far = !(reach * 3.0e-8 <= F.margin * 0.25);
        if (c.far) {
            c.infl = reach * 1.2e-7;
            Bs += c.infl;
            cube();
        }
        c.miss = te > tx || tx < tmin;
        c.te = te;
        c.ts = (te > 0 && !c.miss) ? te : 0;
    }
    return c;
}

// Order-free statement of the sequential winner.  Closest-hit mode: smallest t; on an exact tie a
// sphere/plane beats a box (their range test is inclusive, objects.go:56-60, :110, the box's is
// exclusive, :176), among spheres/planes the higher index wins, among boxes the lower.  Exit mode:
// smallest t, lower index on a tie (`tempRec.t < exitT`, renderer.go:333).
__device__ __forceinline__ bool wins(int mode, bool is_box, int i, double t, int best, bool best_is_box, double tmax) {
    if (t < tmax) return true;
    if (!(t == tmax)) return false;
    if (best < 0) return !is_box;  // tmax is still MaxFloat64: inclusive tests accept t == MaxFloat64, the box does not
    if (mode != 0) return i < best;
    if (is_box) return best_is_box && i < best;  // boxes: the earlier one keeps an exact tie, and never beats a sphere/plane
    return best_is_box || i > best;
}

// The closest hit / exit search of ONE ray by the whole wave: lane l tests objects l, l + 64, ... of the world (every object,
// the reference's FP64 tests), then the lanes' winners are merged by `wins`, the order-free statement of the sequential loop
// (valid for a ray whose every t is a number: the callers' `tame` rays).  For the rays no hierarchy helps with: a ray from so
// far away that the reference's own sphere discriminant cancels (clip_ray: `far`) must keep every object within `infl` of
// its line, and when that tube is as wide as the scene a per-lane walk degenerates into one lane testing every object alone --
// ONE such ray in a frame of 2 * 10^8 took 100 ms at 10^5 objects (profiles/r03_fat_ray.txt).  Here it costs
// nobj / 64 tests per lane.  All 64 lanes must call this together (wave-uniform ray and mode).
// (Not inlined: inside the persistent loop its registers cost the BVH kernel 27 more spilled VGPRs; as a function the loop
// only pays the call sequence on the rare path.)
struct LinearHit {
    int best;
    double tmax;
};
__device__ __attribute__((noinline)) LinearHit scan_linear_wave(int nobj, const DevObj *__restrict__ objs, double rox, double roy, double roz, double rdx,
                                                                double rdy, double rdz, int mode, uint32_t lane) {
    const RayD r{rox, roy, roz, rdx, rdy, rdz};
    int best;
    double tmax;
    const double tmin = mode ? 0.0001 : 0.001;
    tmax = ptm::max_float64();
    best = -1;
    bool best_is_box = false;
    const double a = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
    const double ivx = 1 / r.dx, ivy = 1 / r.dy, ivz = 1 / r.dz;
    for (int i = (int)lane; i < nobj; i += PT_WAVE) {
        const DevObj &o = objs[i];
        const int kind = o.kind & 0xff;
        if (mode != 0 && !(o.kind & 0x100)) continue;  // only glass can end an exit search (renderer.go:333)
        double t = 0;
        bool valid;
        const bool is_box = kind == KIND_BOX;
        if (kind == KIND_SPHERE) valid = sphere_exact(o.a[0], o.a[1], o.a[2], o.radius_sq, r, a, tmin, tmax, t);
        else if (is_box) valid = box_exact(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, ivx, ivy, ivz, tmin, ptm::max_float64(), t);
        else valid = plane_exact(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, tmin, tmax, t);
        if (valid && wins(mode, is_box, i, t, best, best_is_box, tmax) && (mode == 0 || exit_candidate_ok(o, kind, r, t))) {
            best = i;
            tmax = t;
            best_is_box = is_box;
        }
    }
    // merge: after six exchanges every lane holds the wave's winner
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int ob = __shfl_xor(best, off, 64);
        const double ot = __shfl_xor(tmax, off, 64);
        const int obox = __shfl_xor((int)best_is_box, off, 64);
        if (ob >= 0 && wins(mode, obox != 0, ob, ot, best, best_is_box, tmax)) {
            best = ob;
            tmax = ot;
            best_is_box = obox != 0;
        }
    }
    return LinearHit{best, tmax};
}

// The record lists one bitmask scan runs over: every finite object (closest-hit scans, and exit searches of the
// all-in-one kernel, which mask the dielectric ones) or the dielectric objects only (exit searches of glass_kernel).
template <typename SphPtr, typename BoxPtr, bool RECORD_ORDER = false>
struct BroadLists {
    static constexpr bool RO = RECORD_ORDER;
    SphPtr bs;             // sphere records (scalar loads)
    BoxPtr bb;             // box records
    int n_bsph, n_bbox;    // <= 32 each
    uint32_t sph_all, box_all;    // (1 << n) - 1
    uint32_t sph_diel, box_diel;  // records whose object is dielectric
    const int *kidx_s;     // LDS: record -> object index
    const int *kidx_b;
    // LDS, single-group scans of trace_kernel (round 4): the objects again IN RECORD ORDER (slot of push_keep_bit), file index in bits 16-31
    // of `kind`: a narrow-phase round reads its object at ctz(mask) * 80 instead of going through kidx first (one dependent LDS read and
    // three instructions fewer per round); used when RECORD_ORDER
    const DevObj *rs = nullptr;
    const DevObj *rb = nullptr;
};

// Slab parameters of one inflated box from its centre c and half extent h (both rounded so that [c-h, c+h] holds the
// inflated box): t = (c -+ h - o) / d = c*iv - o*iv -+ h*|iv|, i.e. three fma per axis and no min/max to order the
// two slab planes (v_min/v_max_f32 cost a gfx950 SIMD 4 cycles per wave, an fma pair packs into one 4-cycle v_pk_fma:
// profiles/r02_valu_floor.json).  A zero direction component gives iv = inf and NaN or +-inf parameters, which
// v_max3/v_min3 skip or which bound nothing: that slab then constrains nothing (conservative).
#define PT_BOX_SLABS(bx, tn, tf)                                                                                       \
    const float tcx_ = __builtin_fmaf(bx.c[0], ivxf, noxf), tcy_ = __builtin_fmaf(bx.c[1], ivyf, noyf),                \
                tcz_ = __builtin_fmaf(bx.c[2], ivzf, nozf);                                                            \
    const float tn = pt_vmax3(__builtin_fmaf(-bx.h[0], aivxf, tcx_), __builtin_fmaf(-bx.h[1], aivyf, tcy_),            \
                              pt_vmax(__builtin_fmaf(-bx.h[2], aivzf, tcz_), tminf));                                  \
    const float tf = pt_vmin3(__builtin_fmaf(bx.h[0], aivxf, tcx_), __builtin_fmaf(bx.h[1], aivyf, tcy_), __builtin_fmaf(bx.h[2], aivzf, tcz_));

// v_max / v_max3 / v_min3 as instructions (NaN operands are skipped, like the builtins' lowering).  PT_BOX_SLABS uses them: through
// __builtin_fmaxf the compiler quiets tminf again on every turn of the record loop (a v_max_f32 x, x per box: 0.5 % of a C4 frame).
__device__ __forceinline__ float pt_vmax(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float pt_vmin(float a, float b) {
    float d;
    asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// One v_pk_fma_f32 over two slots with the ray constant taken from ONE half of a register pair (op_sel picks the half for both
// results; SELC_LO / SELC_HI pick the halves of the addend: 0,1 = the pair as it is).
#define PT_PK_FMA(dst, a, b, c, SEL_B, SELC_LO, SELC_HI, NEG_A)                                                                \
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0," #SEL_B "," #SELC_LO "] op_sel_hi:[1," #SEL_B "," #SELC_HI "] neg_lo:[" #NEG_A \
        ",0,0] neg_hi:[" #NEG_A ",0,0]"                                                                                        \
        : "=v"(dst)                                                                                                            \
        : "v"(a), "v"(b), "v"(c))
__device__ __forceinline__ float pt_vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float pt_vmin3(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// clip_ray in FP32 with an explicit error term, for the bitmask scans: they never need `infl` (a far ray
// takes the reference's plain loop there), and everything else clip_ray returns only has to err on the safe side:
//   * a slab parameter (+-Bs - o) * (1 / d) in FP32 carries a relative error below 3.6e-7 of (|o| + Bs) |1 / d| (two conversions, one
//     subtraction, v_rcp_f32 at 1 ulp, one product); E = the largest of the three per-axis bounds at 5e-7;
//   * miss is said only when te - E > tx + E (or tx + E < tmin): a ray wrongly called a hit is merely scanned; one wrongly called a miss would
//     have to pass within E |d| <= 5e-7 reach of the cube, which keeps B / 512 - m = 1.7e-3 B of clearance around every bound, and reach
//     is below 2 035 B whenever the ray is not `far` (5e-7 * 2 035 B = 1.0e-3 B);
//   * ts (where the FP32 tests start) and te (compared with tmax) are LOWER bounds of the entry parameter: starting a little early is
//     always safe, the point stays within 1.1e-3 B of the cube, far inside the 4 B the FP32 bounds were analysed for;
//   * far is decided on a reach rounded up.
// 25 FP32 instructions (three v_rcp_f32 the box records need anyway) where clip_ray takes 50 FP64 ones and three v_rcp_f64.
__device__ __forceinline__ Clip clip_ray32(const DevFrame &F, const RayD &r, double tmin) {
    Clip c{0.0, 0.0, 0.0, false, false};
    const double Cb = F.clip_bound;
    if (!(ptm::f_abs(r.ox) <= Cb && ptm::f_abs(r.oy) <= Cb && ptm::f_abs(r.oz) <= Cb)) {
        const float Bf = (float)F.scene_bound * 1.0000002f;  // >= Bs
        const float fox = (float)r.ox, foy = (float)r.oy, foz = (float)r.oz;
        const float fdx = (float)r.dx, fdy = (float)r.dy, fdz = (float)r.dz;
        const float ix = __builtin_amdgcn_rcpf(fdx), iy = __builtin_amdgcn_rcpf(fdy), iz = __builtin_amdgcn_rcpf(fdz);
        const float x0 = (-Bf - fox) * ix, x1 = (Bf - fox) * ix;
        const float y0 = (-Bf - foy) * iy, y1 = (Bf - foy) * iy;
        const float z0 = (-Bf - foz) * iz, z1 = (Bf - foz) * iz;
        // (v_min / v_max skip the NaN of a 0 * inf slab boundary: that slab then constrains nothing, as in clip_ray)
        const float te = pt_vmax3(pt_vmin(x0, x1), pt_vmin(y0, y1), pt_vmin(z0, z1));
        const float tx = pt_vmin3(pt_vmax(x0, x1), pt_vmax(y0, y1), pt_vmax(z0, z1));
        const float E = 5.0e-7f * pt_vmax3((__builtin_fabsf(fox) + Bf) * __builtin_fabsf(ix), (__builtin_fabsf(foy) + Bf) * __builtin_fabsf(iy),
                                           (__builtin_fabsf(foz) + Bf) * __builtin_fabsf(iz));
        const float reach = (__builtin_fabsf(fox) + __builtin_fabsf(foy) + __builtin_fabsf(foz) +
                             (__builtin_fabsf(fdx) + __builtin_fabsf(fdy) + __builtin_fabsf(fdz)) * __builtin_fabsf(te)) * 1.00001f;
        // (NaN or infinite quantities land on `far`: the plain loop)
        c.far = !(reach * 3.0e-8f <= (float)(F.margin * 0.2499));
        const float te_lo = te - E, tx_hi = tx + E;
        c.miss = te_lo > tx_hi || tx_hi < (float)tmin * 0.99f;
        if (c.far) c.miss = false;
        c.te = (double)te_lo;
        c.ts = (te_lo > 0.0f && !c.miss) ? (double)te_lo : 0.0;
    }
    return c;
}

// Candidate masks are built by shifting: mask = 2 * mask + keep, keep = !(a < b) -- one compare and ONE v_addc_co_u32 (the
// compare's lane mask is the carry-in) where `mask |= keep ? bit : 0` cost a v_mov, a v_cndmask and a v_or per record (8 against 4
// SIMD-cycles per record and wave, 26 records on C4).  NaN keeps the record (nlt), as before.  The k-th of n records ends up
// in bit n - 1 - k; pt_record_slot() is the same map for the tables and masks that go with it.
__device__ __forceinline__ void push_keep_bit(uint32_t &mask, float a, float b) {
    asm("v_cmp_nlt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(a), "v"(b) : "vcc");
}
// table / mask position of record i of n, taken in groups of 32: its group's base + the bit push_keep_bit leaves it in
__host__ __device__ __forceinline__ int pt_record_slot(int i, int n) {
    const int g = i & ~31, cnt = (n - g) < 32 ? (n - g) : 32;
    return g + (cnt - 1 - (i - g));
}

// The scene's one plane (normal (0, 1, 0)) as the scans take it from the argument block: DevFrame::plane0_*, read by the caller right before
// the scan (from the kernarg segment, where the kernel re-reads such things) so that it does not sit in scalar registers through the loops.
struct Plane0 {
    double y;
    int32_t index;  // < 0: no such plane, the scan goes through plane_idx[] -> objs[]
    int32_t kind;
};

// Broad phase in FP32 over inflated bounds + exact FP64 narrow phase over the survivors.
// MODE: 0 closest hit, 1 exit search, -1 decided per lane by `mode_rt`.
template <bool PROF, bool DBG, int MODE, typename ObjPtr, typename BLT, typename IdxPtr>
__device__ __forceinline__ void scan_broad_narrow(const DevFrame &F, ObjPtr g_obj, const BLT &BL, IdxPtr g_pl,
                                                  const DevObj *s_obj, const RayD &r, const Clip &clip,
                                                  int mode_rt, int &best, double &tmax, const ProfHooks &ph, const Plane0 P0 = Plane0{0.0, -1, 0}) {
    const int mode = MODE < 0 ? mode_rt : MODE;
    const double tmin = mode ? 0.0001 : 0.001;
    tmax = ptm::max_float64();
    best = -1;
    bool best_is_box = false;
    const double a = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
    struct SphRec { float cx, cy, cz, rm2; };
    // (requesting the first four sphere records here, ahead of the plane test and the FP32 set-up, was measured: +1.2 % time -- the sixteen
    // scalar registers they hold meanwhile cost more than the wait they save; profiles/r04_plane0_ab.txt)

    // ---- planes: infinite, always tested exactly (wave-uniform, scalar loads)
    PH_BEGIN(SEC_PLANE)
#if PT_PLANE0
    if (P0.index >= 0) {
        // the one plane of the scene, normal (0, 1, 0), straight from the argument block (see DevFrame::plane0_y)
        const int i = P0.index;
        const bool pdiel = (P0.kind & 0x100) != 0;
        if (!(MODE == 1 && !pdiel)) {
            double t = 0;
            bool acc = plane_exact_y(P0.y, r, tmin, tmax, t);
            if (MODE < 0 && mode != 0 && !pdiel) acc = false;
            if (acc) {
                if (mode == 0 ? wins(0, false, i, t, best, best_is_box, tmax)
                              : (wins(1, false, i, t, best, best_is_box, tmax) && exit_candidate_ok(g_obj[i], KIND_PLANE, r, t))) {
                    best = i;
                    tmax = t;
                    best_is_box = false;
                }
            }
        }
    } else
#endif
    for (int k = 0; k < F.n_plane; k++) {
        const int i = g_pl[k];
        const auto &o = g_obj[i];
        if (MODE == 1 && !(o.kind & 0x100)) continue;  // wave-uniform
        double t = 0;
        bool acc = F.planes_y ? plane_exact_y(o.a[1], r, tmin, tmax, t)
                              : plane_exact(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, tmin, tmax, t);
        if (MODE < 0 && mode != 0 && !(o.kind & 0x100)) acc = false;
        if (acc) {
            if (mode == 0 ? wins(0, false, i, t, best, best_is_box, tmax)
                          : (wins(1, false, i, t, best, best_is_box, tmax) && exit_candidate_ok(o, KIND_PLANE, r, t))) {
                best = i;
                tmax = t;
                best_is_box = false;
            }
        }
    }

    PH_END(SEC_PLANE)
    // ---- broad phase
    PH_BEGIN(SEC_BROAD)
    // rays from outside the scene cube start their FP32 tests at the entry point (clip_ray)
    const double ts = clip.ts;
    const bool outside_all = clip.miss || clip.te > tmax;
    const float fox = (float)(r.ox + r.dx * ts), foy = (float)(r.oy + r.dy * ts), foz = (float)(r.oz + r.dz * ts);
    const float fdx = (float)r.dx, fdy = (float)r.dy, fdz = (float)r.dz;
    const float fa = __builtin_fmaf(fdx, fdx, __builtin_fmaf(fdy, fdy, fdz * fdz));
    // written so that NaN lands on "keep everything"
    const bool trust = (fa > 1e-30f) && (fa < 1e30f) && (__builtin_fabsf(fox) <= F.origin_bound) &&
                       (__builtin_fabsf(foy) <= F.origin_bound) && (__builtin_fabsf(foz) <= F.origin_bound);
    float tminf = (float)(tmin - ts);  // FP32 parameters are relative to the entry point
    tminf -= __builtin_fabsf(tminf) * 1e-2f + 1e-6f;  // a little below tMin - ts
    // The sphere test runs on the UNIT direction u = d / |d| (|u| = 1 +- 3e-7) and in distances along the ray: the closest
    // approach lies at distance -oc.u, tMin at tminf |d|.  Two multiplications per record less than with d and parameters
    // (-b / a and w^2 a); the error this adds to |q|^2 is ~ 5e-6 B r against the 2 r m = 5e-4 B r the inflated radius carries.
    const float inv_len = __builtin_amdgcn_rsqf(fa);
    const float ux = fdx * inv_len, uy = fdy * inv_len, uz = fdz * inv_len;
    const float tmin_d = tminf * (fa * inv_len);
    // candidate masks: the k-th of n sphere records in bit n - 1 - k of `cs`, likewise the boxes in `cb` (<= 32 of each; push_keep_bit)
    uint32_t cs = 0, cb = 0;
    // (two records per turn, both fetched before the first is used: the scalar loads of the second hide behind the first one's
    // 18 vector instructions; the compiler does not unroll across the asm of push_keep_bit by itself)
    // (Round 4 also tried |q|^2 = |oc|^2 - sd^2 with an explicit error term on the culling side of the compare -- 16 instead of 18 vector
    // instructions per record: trace passes 441.0 -> 440.1 ms per C4 frame, i.e. nothing; profiles/r04_broad_unroll_ab.txt.  Not kept.)
    auto sphere_record = [&](const auto &s) {
        const float ocx = fox - s.cx, ocy = foy - s.cy, ocz = foz - s.cz;
        const float sd = __builtin_fmaf(ocx, ux, __builtin_fmaf(ocy, uy, ocz * uz));  // minus the distance of closest approach
        const float qx = __builtin_fmaf(-sd, ux, ocx), qy = __builtin_fmaf(-sd, uy, ocy), qz = __builtin_fmaf(-sd, uz, ocz);
        const float d2 = __builtin_fmaf(qx, qx, __builtin_fmaf(qy, qy, qz * qz));
        const float rem = s.rm2 - d2;  // >= 0: the line passes within the inflated radius
        const float w = __builtin_fmaxf(tmin_d + sd, 0.0f);  // > 0: closest approach lies before tMin
        // outside (rem < 0) or wholly behind (w^2 > rem) in one compare, as w^2 >= 0; NaN keeps the sphere
        push_keep_bit(cs, rem, w * w);
    };
    {
        int k = 0;
        typedef SphRec Rec;
#if PT_BROAD_UNROLL >= 4
        for (; k + 3 < BL.n_bsph; k += 4) {
            const auto &a0 = BL.bs[k];
            const auto &a1 = BL.bs[k + 1];
            const auto &a2 = BL.bs[k + 2];
            const auto &a3 = BL.bs[k + 3];
            const Rec r0{a0.cx, a0.cy, a0.cz, a0.rm2}, r1{a1.cx, a1.cy, a1.cz, a1.rm2}, r2{a2.cx, a2.cy, a2.cz, a2.rm2}, r3{a3.cx, a3.cy, a3.cz, a3.rm2};
            sphere_record(r0);
            sphere_record(r1);
            sphere_record(r2);
            sphere_record(r3);
        }
#endif
        for (; k + 1 < BL.n_bsph; k += 2) {
            const auto &a0 = BL.bs[k];
            const auto &a1 = BL.bs[k + 1];
            const Rec r0{a0.cx, a0.cy, a0.cz, a0.rm2}, r1{a1.cx, a1.cy, a1.cz, a1.rm2};
            sphere_record(r0);
            sphere_record(r1);
        }
        if (k < BL.n_bsph) sphere_record(BL.bs[k]);
    }
    const float ivxf = __builtin_amdgcn_rcpf(fdx), ivyf = __builtin_amdgcn_rcpf(fdy), ivzf = __builtin_amdgcn_rcpf(fdz);
    const float aivxf = __builtin_fabsf(ivxf), aivyf = __builtin_fabsf(ivyf), aivzf = __builtin_fabsf(ivzf);
    // -o/d: the extra rounding (~2^-24 |o| in distance) is far inside the margin, and inf - inf = NaN for a zero
    // direction component leaves that slab unconstrained
    const float noxf = -fox * ivxf, noyf = -foy * ivyf, nozf = -foz * ivzf;
    auto box_record = [&](const auto &bx) {
        PT_BOX_SLABS(bx, t0, t1)
        push_keep_bit(cb, t1, t0);
    };
    {
        int k = 0;
        struct Rec { float c[3], h[3]; };
#if PT_BROAD_UNROLL >= 4
        for (; k + 3 < BL.n_bbox; k += 4) {
            const auto &a0 = BL.bb[k];
            const auto &a1 = BL.bb[k + 1];
            const auto &a2 = BL.bb[k + 2];
            const auto &a3 = BL.bb[k + 3];
            const Rec b0{{a0.c[0], a0.c[1], a0.c[2]}, {a0.h[0], a0.h[1], a0.h[2]}};
            const Rec b1{{a1.c[0], a1.c[1], a1.c[2]}, {a1.h[0], a1.h[1], a1.h[2]}};
            const Rec b2{{a2.c[0], a2.c[1], a2.c[2]}, {a2.h[0], a2.h[1], a2.h[2]}};
            const Rec b3{{a3.c[0], a3.c[1], a3.c[2]}, {a3.h[0], a3.h[1], a3.h[2]}};
            box_record(b0);
            box_record(b1);
            box_record(b2);
            box_record(b3);
        }
#endif
        for (; k + 1 < BL.n_bbox; k += 2) {
            const auto &a0 = BL.bb[k];
            const auto &a1 = BL.bb[k + 1];
            const Rec b0{{a0.c[0], a0.c[1], a0.c[2]}, {a0.h[0], a0.h[1], a0.h[2]}};
            const Rec b1{{a1.c[0], a1.c[1], a1.c[2]}, {a1.h[0], a1.h[1], a1.h[2]}};
            box_record(b0);
            box_record(b1);
        }
        if (k < BL.n_bbox) box_record(BL.bb[k]);
    }
    if (!trust) { cs = BL.sph_all; cb = BL.box_all; }
    if (outside_all) { cs = 0; cb = 0; }
    if (MODE < 0) {
        if (mode != 0) { cs &= BL.sph_diel; cb &= BL.box_diel; }
    } else if (MODE == 1) {
        cs &= BL.sph_diel;
        cb &= BL.box_diel;
    }
    if (DBG) { cs &= ~F.debug_drop; cb &= ~F.debug_drop; }
    PH_END(SEC_BROAD)

    // ---- narrow phase: spheres, then boxes, each lane on its own candidates (index order)
    uint32_t ms = cs;
    const double ya = div_recip(a);  // both roots of every sphere divide by a
    while (__ballot(ms != 0) != 0) {
        if (ms != 0) {
            PH_BEGIN(SEC_NSPH)
            const int slot_ = __builtin_ctz(ms);
            ms &= ms - 1;
            const DevObj &o = BLT::RO ? BL.rs[slot_] : s_obj[BL.kidx_s[BLT::RO ? 0 : slot_]];
            const int i = BLT::RO ? (int)((uint32_t)o.kind >> 16) : BL.kidx_s[BLT::RO ? 0 : slot_];  // record -> object index
            double t = 0;
            bool acc = sphere_exact_shared(o.a[0], o.a[1], o.a[2], o.radius_sq, r, a, ya, tmin, tmax, t) &&
                       wins(mode, false, i, t, best, best_is_box, tmax);
            if (MODE != 0 && acc && mode != 0) acc = exit_candidate_ok(o, KIND_SPHERE, r, t);
            // selects, not branches: the update is four v_cndmask
            best = acc ? i : best;
            tmax = acc ? t : tmax;
            best_is_box = acc ? false : best_is_box;
            PH_END(SEC_NSPH)
        }
    }
    uint32_t mb = cb;
    if (__ballot(mb != 0) != 0) {
        const double ivx = 1 / r.dx, ivy = 1 / r.dy, ivz = 1 / r.dz;
        while (__ballot(mb != 0) != 0) {
            if (mb != 0) {
                PH_BEGIN(SEC_NBOX)
                const int slot_ = __builtin_ctz(mb);
                mb &= mb - 1;
                const DevObj &o = BLT::RO ? BL.rb[slot_] : s_obj[BL.kidx_b[BLT::RO ? 0 : slot_]];
                const int i = BLT::RO ? (int)((uint32_t)o.kind >> 16) : BL.kidx_b[BLT::RO ? 0 : slot_];
                double t = 0;
                // the range is left open at the top here: `wins` compares t with tmax (strictly for a box)
                bool acc = box_exact<true>(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, ivx, ivy, ivz, tmin, ptm::max_float64(), t) &&
                           wins(mode, true, i, t, best, best_is_box, tmax);
                if (MODE != 0 && acc && mode != 0) acc = exit_candidate_ok(o, KIND_BOX, r, t);
                best = acc ? i : best;
                tmax = acc ? t : tmax;
                best_is_box = acc ? true : best_is_box;
                PH_END(SEC_NBOX)
            }
        }
    }
}

// The same broad / narrow scan for up to 128 spheres and 128 boxes: the records are taken in groups of 32 (one
// candidate mask at a time, so no more registers than the single-group version), the dielectric mask of a
// group is collected from the records on the scalar unit.  Between ~33 and ~200 objects this linear scan at
// full lanes beats the hierarchy, whose walks diverge.
template <bool PROF, bool DBG, int MODE, typename ObjPtr, typename BLT, typename IdxPtr>
__device__ __forceinline__ void scan_broad_narrow_wide(const DevFrame &F, ObjPtr g_obj, const BLT &BL, IdxPtr g_pl,
                                                       const DevObj *s_obj, const RayD &r, const Clip &clip,
                                                       int mode_rt, int &best, double &tmax, const ProfHooks &ph, const Plane0 P0 = Plane0{0.0, -1, 0}) {
    const int mode = MODE < 0 ? mode_rt : MODE;
    const double tmin = mode ? 0.0001 : 0.001;
    tmax = ptm::max_float64();
    best = -1;
    bool best_is_box = false;
    const double a = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
    const double ya = div_recip(a);

    PH_BEGIN(SEC_PLANE)
#if PT_PLANE0
    if (P0.index >= 0) {  // the scene's one plane from the argument block (see scan_broad_narrow)
        const int i = P0.index;
        double t = 0;
        if (!(mode != 0 && !(P0.kind & 0x100)) && plane_exact_y(P0.y, r, tmin, tmax, t)) {
            if (mode == 0 ? wins(0, false, i, t, best, best_is_box, tmax)
                          : (wins(1, false, i, t, best, best_is_box, tmax) && exit_candidate_ok(g_obj[i], KIND_PLANE, r, t))) {
                best = i;
                tmax = t;
                best_is_box = false;
            }
        }
    } else
#endif
    for (int k = 0; k < F.n_plane; k++) {
        const int i = g_pl[k];
        const auto &o = g_obj[i];
        if (mode != 0 && !(o.kind & 0x100)) continue;
        double t = 0;
        if (F.planes_y ? plane_exact_y(o.a[1], r, tmin, tmax, t)
                       : plane_exact(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, tmin, tmax, t)) {
            if (mode == 0 ? wins(0, false, i, t, best, best_is_box, tmax)
                          : (wins(1, false, i, t, best, best_is_box, tmax) && exit_candidate_ok(o, KIND_PLANE, r, t))) {
                best = i;
                tmax = t;
                best_is_box = false;
            }
        }
    }
    PH_END(SEC_PLANE)

    const double ts = clip.ts;
    const bool outside_all = clip.miss || clip.te > tmax;
    const float fox = (float)(r.ox + r.dx * ts), foy = (float)(r.oy + r.dy * ts), foz = (float)(r.oz + r.dz * ts);
    const float fdx = (float)r.dx, fdy = (float)r.dy, fdz = (float)r.dz;
    const float fa = __builtin_fmaf(fdx, fdx, __builtin_fmaf(fdy, fdy, fdz * fdz));
    const bool trust = (fa > 1e-30f) && (fa < 1e30f) && (__builtin_fabsf(fox) <= F.origin_bound) &&
                       (__builtin_fabsf(foy) <= F.origin_bound) && (__builtin_fabsf(foz) <= F.origin_bound);
    float tminf = (float)(tmin - ts);
    tminf -= __builtin_fabsf(tminf) * 1e-2f + 1e-6f;
    const float inv_a = __builtin_amdgcn_rcpf(fa);
    const float ivxf = __builtin_amdgcn_rcpf(fdx), ivyf = __builtin_amdgcn_rcpf(fdy), ivzf = __builtin_amdgcn_rcpf(fdz);
    const float aivxf = __builtin_fabsf(ivxf), aivyf = __builtin_fabsf(ivyf), aivzf = __builtin_fabsf(ivzf);
    const float noxf = -fox * ivxf, noyf = -foy * ivyf, nozf = -foz * ivzf;

    for (int base = 0; base < BL.n_bsph; base += 32) {
        const int cnt = BL.n_bsph - base < 32 ? BL.n_bsph - base : 32;
        PH_BEGIN(SEC_BROAD)
        uint32_t cs = 0, diel = 0;
        struct SRec { float cx, cy, cz, rm2; int32_t diel; };
        auto sphere_record = [&](const SRec &s) {
            const float ocx = fox - s.cx, ocy = foy - s.cy, ocz = foz - s.cz;
            const float b = __builtin_fmaf(ocx, fdx, __builtin_fmaf(ocy, fdy, ocz * fdz));
            const float tca = -b * inv_a;
            const float qx = __builtin_fmaf(fdx, tca, ocx), qy = __builtin_fmaf(fdy, tca, ocy), qz = __builtin_fmaf(fdz, tca, ocz);
            const float d2 = __builtin_fmaf(qx, qx, __builtin_fmaf(qy, qy, qz * qz));
            const float rem = s.rm2 - d2;
            const float w = __builtin_fmaxf(tminf - tca, 0.0f);
            push_keep_bit(cs, rem, w * w * fa);
            diel = (diel << 1) | (s.diel ? 1u : 0u);  // wave-uniform: scalar unit (same bit order as cs)
        };
        {   // four records per turn, their scalar loads issued first (see scan_broad_narrow)
            int k = 0;
            for (; k + 3 < cnt; k += 4) {
                SRec r[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { const auto &a = BL.bs[base + k + q]; r[q] = SRec{a.cx, a.cy, a.cz, a.rm2, a.diel}; }
#pragma unroll
                for (int q = 0; q < 4; q++) sphere_record(r[q]);
            }
            for (; k < cnt; k++) { const auto &a = BL.bs[base + k]; sphere_record(SRec{a.cx, a.cy, a.cz, a.rm2, a.diel}); }
        }
        if (!trust) cs = cnt == 32 ? 0xffffffffu : ((1u << cnt) - 1u);
        if (outside_all) cs = 0;
        if (mode != 0) cs &= diel;
        if (DBG) cs &= ~F.debug_drop;
        PH_END(SEC_BROAD)
        while (__ballot(cs != 0) != 0) {
            if (cs != 0) {
                PH_BEGIN(SEC_NSPH)
                const int i = BL.kidx_s[base + __builtin_ctz(cs)];
                cs &= cs - 1;
                const DevObj &o = s_obj[i];
                double t = 0;
                bool acc = sphere_exact_shared(o.a[0], o.a[1], o.a[2], o.radius_sq, r, a, ya, tmin, tmax, t) &&
                           wins(mode, false, i, t, best, best_is_box, tmax);
                if (acc && mode != 0) acc = exit_candidate_ok(o, KIND_SPHERE, r, t);
                best = acc ? i : best;
                tmax = acc ? t : tmax;
                best_is_box = acc ? false : best_is_box;
                PH_END(SEC_NSPH)
            }
        }
    }
    if (BL.n_bbox > 0) {
        const double ivx = 1 / r.dx, ivy = 1 / r.dy, ivz = 1 / r.dz;
        for (int base = 0; base < BL.n_bbox; base += 32) {
            const int cnt = BL.n_bbox - base < 32 ? BL.n_bbox - base : 32;
            PH_BEGIN(SEC_BROAD)
            uint32_t cb = 0, diel = 0;
            struct BRec { float c[3], h[3]; int32_t diel; };
            auto box_record = [&](const BRec &bx) {
                PT_BOX_SLABS(bx, t0, t1)
                push_keep_bit(cb, t1, t0);
                diel = (diel << 1) | (bx.diel ? 1u : 0u);
            };
            {
                int k = 0;
                for (; k + 3 < cnt; k += 4) {
                    BRec r[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) { const auto &a = BL.bb[base + k + q]; r[q] = BRec{{a.c[0], a.c[1], a.c[2]}, {a.h[0], a.h[1], a.h[2]}, a.diel}; }
#pragma unroll
                    for (int q = 0; q < 4; q++) box_record(r[q]);
                }
                for (; k < cnt; k++) { const auto &a = BL.bb[base + k]; box_record(BRec{{a.c[0], a.c[1], a.c[2]}, {a.h[0], a.h[1], a.h[2]}, a.diel}); }
            }
            if (!trust) cb = cnt == 32 ? 0xffffffffu : ((1u << cnt) - 1u);
            if (outside_all) cb = 0;
            if (mode != 0) cb &= diel;
            if (DBG) cb &= ~F.debug_drop;
            PH_END(SEC_BROAD)
            while (__ballot(cb != 0) != 0) {
                if (cb != 0) {
                    PH_BEGIN(SEC_NBOX)
                    const int i = BL.kidx_b[base + __builtin_ctz(cb)];
                    cb &= cb - 1;
                    const DevObj &o = s_obj[i];
                    double t = 0;
                    bool acc = box_exact<true>(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, ivx, ivy, ivz, tmin, ptm::max_float64(), t) &&
                               wins(mode, true, i, t, best, best_is_box, tmax);
                    if (acc && mode != 0) acc = exit_candidate_ok(o, KIND_BOX, r, t);
                    best = acc ? i : best;
                    tmax = acc ? t : tmax;
                    best_is_box = acc ? true : best_is_box;
                    PH_END(SEC_NBOX)
                }
            }
        }
    }
}

// A node as the FP32 walk uses it: the slot boxes as centre and half extent, two slots per register pair (seven 16-byte loads).
struct NodeQ {
    v2f cA[3], cB[3], hA[3], hB[3];  // [axis]: slots (0, 1) and (2, 3)
    int nbase, obase;
    uint32_t meta;
};
__device__ __forceinline__ NodeQ load_node(const BvhNode *__restrict__ p) {
    const float4 *p4 = reinterpret_cast<const float4 *>(p);
    NodeQ n;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float4 c = p4[k], h = p4[3 + k];
        n.cA[k] = v2f{c.x, c.y};
        n.cB[k] = v2f{c.z, c.w};
        n.hA[k] = v2f{h.x, h.y};
        n.hB[k] = v2f{h.z, h.w};
    }
    const int4 m4 = reinterpret_cast<const int4 *>(p)[6];
    n.nbase = m4.x;
    n.obase = m4.y;
    n.meta = (uint32_t)m4.z;
    return n;
}

// True when the FP32 quantities scan_bvh derives from the ray are inside the range its bounds were analysed
// for (direction length and re-based origin); a little stricter than the FP32 test of the careful
// instantiation, so a wave that passes here needs no per-node check.
__device__ __forceinline__ bool bvh_ray_trusted(const DevFrame &F, const RayD &r, const Clip &clip, double a) {
    const double lim = 0.99 * (double)F.origin_bound;
    return (a > 1e-29) && (a < 1e29) && (ptm::f_abs(r.ox + r.dx * clip.ts) <= lim) &&
           (ptm::f_abs(r.oy + r.dy * clip.ts) <= lim) && (ptm::f_abs(r.oz + r.dz * clip.ts) <= lim);
}

// Closest hit / exit search through the BVH.  Traversal order and culling only decide which
// objects get the exact test; `wins` makes the result independent of that order.
// CAREFUL: some lane of the wave carries clip.infl > 0 (see clip_ray; every node and object bound is then
// widened by the lane's own infl) or a ray the FP32 bounds were not analysed for (it visits every node).
// The caller guarantees bvh_ray_trusted() for every lane of a wave it sends to the other instantiation.
//
// A traversal can be left unfinished: lanes of one wave need very different numbers of node visits, and a wave
// that waits for its longest walk runs at a fifth of its lanes.  Once fewer than F.bvh_min_lanes lanes are still
// walking (and at least one lane of this trip is done), the loop ends; the stragglers keep their state
// (TravState + their LDS stack column), skip shading, and carry on in the wave's next trip next to the fresh
// scans of the other lanes.  Returns true when this lane's scan is complete (best / tmax valid).
struct TravState {
    int cur = -1;        // node to visit next
    int sp = 0;
    int best = -1;
    double tmax = 0;
    bool best_is_box = false;
    bool live = false;   // a traversal of the lane's current ray is in progress
};
template <bool PROF, bool CAREFUL, typename ObjPtr, typename IdxPtr>
__device__ __forceinline__ bool scan_bvh(const DevFrame &F, ObjPtr g_obj, IdxPtr g_pl, const BvhNode *__restrict__ nodes,
                                         const BvhNode *lds_nodes /* nodes[0 .. F.bvh_lds_nodes) staged in LDS */,
                                         const BvhObj *__restrict__ bobjs, int *stack /* this lane's column, stride PT_BLOCK */,
                                         const RayD &r, const Clip &clip, int mode, TravState &S, int &best, double &tmax,
                                         const ProfHooks &ph) {
    const double tmin = mode ? 0.0001 : 0.001;
    const bool resume = S.live;
    S.live = false;
    tmax = resume ? S.tmax : ptm::max_float64();
    best = resume ? S.best : -1;
    bool best_is_box = resume ? S.best_is_box : false;
    const double a = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
    const double ya = div_recip(a);

    if (!resume) {
    PH_BEGIN(SEC_PLANE)
    for (int k = 0; k < F.n_plane; k++) {
        const int i = g_pl[k];
        const auto &o = g_obj[i];
        if (mode != 0 && !(o.kind & 0x100)) continue;
        double t = 0;
        if (F.planes_y ? plane_exact_y(o.a[1], r, tmin, tmax, t)
                       : plane_exact(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, tmin, tmax, t)) {
            if (mode == 0 ? wins(0, false, i, t, best, best_is_box, tmax)
                          : (wins(1, false, i, t, best, best_is_box, tmax) && exit_candidate_ok(o, KIND_PLANE, r, t))) {
                best = i;
                tmax = t;
                best_is_box = false;
            }
        }
    }
    PH_END(SEC_PLANE)
    }
    const int root = mode ? F.bvh_root_exit : F.bvh_root;
    if (root < 0) return true;

    const double ivx = 1 / r.dx, ivy = 1 / r.dy, ivz = 1 / r.dz;
    // rays from outside the scene cube: a miss ends the scan, the others start their FP32 node tests at the
    // entry point (clip_ray); parameters below are relative to ts
    if (!resume && (clip.miss || clip.te > tmax)) return true;
    const double ts = clip.ts;
    const float fox = (float)(r.ox + r.dx * ts), foy = (float)(r.oy + r.dy * ts), foz = (float)(r.oz + r.dz * ts);
    const float fdx = (float)r.dx, fdy = (float)r.dy, fdz = (float)r.dz;
    const float fa = __builtin_fmaf(fdx, fdx, __builtin_fmaf(fdy, fdy, fdz * fdz));
    const bool trust = !CAREFUL || ((fa > 1e-30f) && (fa < 1e30f) && (__builtin_fabsf(fox) <= F.origin_bound) &&
                                    (__builtin_fabsf(foy) <= F.origin_bound) && (__builtin_fabsf(foz) <= F.origin_bound));
    // node parameters are relative to the re-based origin: t' = t - ts
    float tminf = (float)((mode ? 0.0001 : 0.001) - ts);
    tminf -= __builtin_fabsf(tminf) * 1e-2f + 1e-6f;  // a little below tMin - ts
    // A re-based ray starts on the scene cube, which keeps B/512 - m of clearance around every bound: nothing
    // begins before t' = 0.  With that every entry parameter below is >= 0 and its bit pattern orders like it.
    tminf = __builtin_fmaxf(tminf, 0.0f);
    const float ivxf = __builtin_amdgcn_rcpf(fdx), ivyf = __builtin_amdgcn_rcpf(fdy), ivzf = __builtin_amdgcn_rcpf(fdz);
    // slab parameters as bound * (1/d) - o/d (see scan_broad_narrow)
    const float noxf = -fox * ivxf, noyf = -foy * ivyf, nozf = -foz * ivzf;
    // widening of a slab by infl in parameter units (a little more than infl / |d|; inf or NaN for a zero
    // component, which unconstrains the slab)
    const float inflf = CAREFUL ? (float)clip.infl * 1.0001f : 0.0f;
    const float exf = inflf * __builtin_fabsf(ivxf), eyf = inflf * __builtin_fabsf(ivyf), ezf = inflf * __builtin_fabsf(ivzf);
    // the per-ray constants two to a register pair, the half picked per instruction by op_sel (PT_PK_FMA): 5 pairs (6 with the
    // widening) where the compiler's own splat handling gives each constant a pair of its own (9 / 12 pairs)
    const v2f iv_xy = {ivxf, ivyf}, ivz_aivx = {ivzf, __builtin_fabsf(ivxf)}, aiv_yz = {__builtin_fabsf(ivyf), __builtin_fabsf(ivzf)};
    const v2f no_xy = {noxf, noyf}, noz_ex = {nozf, exf}, ey_ez = {eyf, ezf};
    float tmaxf = (float)(tmax - ts);
    tmaxf += __builtin_fabsf(tmaxf) * 4.8e-7f;  // >= tmax - ts (MaxFloat64 becomes +inf)
    int sp = resume ? S.sp : 0;
    int cur = resume ? S.cur : root;  // node to visit next, -1 when this lane has none left
    uint32_t pend = 0;         // slots of the last visited node whose object still awaits its exact test
    uint32_t pend_meta = 0;
    int pend_base = 0;
    uint32_t n_leaf = 0;  // PROF only: object batches this lane went through
    PH_BEGIN(SEC_BROAD)
    const int n_start = __popcll(__ballot(1));
    const bool single = F.bvh_leaf_single != 0;  // one exact test per waiting lane and pass (lanes that reach objects meanwhile join the next pass)
    for (;;) {
        const int walking = __popcll(__ballot(cur >= 0));
        const bool waiting = __ballot(pend != 0) != 0;
        // stragglers carry on in the next trip (with no test pending: what waits is tested before the loop is left)
        // (Ending a trip after a fixed number of visits instead -- so that lanes that finish early wait for a bounded time -- was
        // measured in round 3 and lost: 8 / 12 / 16 / 24 / 32 visits per trip 80.9 / 71.0 / 67.0 / 63.8 / 62.6 ms against 62.7 at 10^5
        // objects, profiles/r03_visits_sweep.txt: every trip pays the set-up, shading and refill code for the whole wave.)
        const bool leaving = walking < F.bvh_min_lanes && walking < n_start;
        if (!waiting && (walking == 0 || leaving)) break;
        // ---- walk internal nodes: a lane that reaches a node with pierced object slots waits (pend != 0) for the exact
        // tests below.  The walk goes on as long as enough lanes are still walking; once fewer than F.bvh_node_min are and
        // some lane waits, the waiting lanes get their tests first and rejoin the walk (waiting for the LAST lane to
        // find its objects left the visit code running at a quarter of its lanes).
        for (;;) {
            const bool want = cur >= 0 && pend == 0;
            const uint64_t wm = __ballot(want);
            if (wm == 0 || leaving) break;
            if ((int)__popcll(wm) < F.bvh_node_min && __ballot(pend != 0) != 0) break;
            if (want) {
            PH_BEGIN(SEC_COSINE)  // BVH runs: wave-level executions, lanes and cycles of the node visit
            if (PROF) {
                ph.lanes[SEC_NBOX]++;  // node visits (lane count)
                if (cur < F.bvh_lds_nodes) ph.exec[SEC_DIEL]++;       // ... served by the LDS copy of the top of the tree
                if (cur < 4 * F.bvh_lds_nodes) ph.lanes[SEC_DIEL]++;  // ... that a four times larger copy would serve
            }
            BvhNode nd;
            if (cur < F.bvh_lds_nodes) nd = lds_nodes[cur];  // top of the tree: LDS packet
            else nd = nodes[cur];                            // below: HBM / L2
            float t0[4];
            uint32_t hb = 0;
            // slab parameters of the four slots from centre and half extent, two slots per packed instruction:
            // tc = c*iv - o*iv, tn = tc - h*|iv|, tf = tc + h*|iv| (three v_pk_fma_f32 per axis and pair; the lo / hi form
            // cost two v_fma_f32 and a 4-cycle v_min / v_max pair per axis and SLOT)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const v2f cx = {nd.c[0][2 * p], nd.c[0][2 * p + 1]}, cy = {nd.c[1][2 * p], nd.c[1][2 * p + 1]},
                          cz = {nd.c[2][2 * p], nd.c[2][2 * p + 1]};
                const v2f hx = {nd.h[0][2 * p], nd.h[0][2 * p + 1]}, hy = {nd.h[1][2 * p], nd.h[1][2 * p + 1]},
                          hz = {nd.h[2][2 * p], nd.h[2][2 * p + 1]};
                v2f tcx, tcy, tcz;
                PT_PK_FMA(tcx, cx, iv_xy, no_xy, 0, 0, 0, 0);
                PT_PK_FMA(tcy, cy, iv_xy, no_xy, 1, 1, 1, 0);
                PT_PK_FMA(tcz, cz, ivz_aivx, noz_ex, 0, 0, 0, 0);
                v2f nx_, ny_, nz_, fx_, fy_, fz_;
                if (CAREFUL) {  // every bound widened by the lane's own inflation (in parameter units: exf, eyf, ezf)
                    v2f thx, thy, thz;
                    PT_PK_FMA(thx, hx, ivz_aivx, noz_ex, 1, 1, 1, 0);
                    PT_PK_FMA(thy, hy, aiv_yz, ey_ez, 0, 0, 0, 0);
                    PT_PK_FMA(thz, hz, aiv_yz, ey_ez, 1, 1, 1, 0);
                    nx_ = tcx - thx; ny_ = tcy - thy; nz_ = tcz - thz;
                    fx_ = tcx + thx; fy_ = tcy + thy; fz_ = tcz + thz;
                } else {
                    PT_PK_FMA(nx_, hx, ivz_aivx, tcx, 1, 0, 1, 1);
                    PT_PK_FMA(ny_, hy, aiv_yz, tcy, 0, 0, 1, 1);
                    PT_PK_FMA(nz_, hz, aiv_yz, tcz, 1, 0, 1, 1);
                    PT_PK_FMA(fx_, hx, ivz_aivx, tcx, 1, 0, 1, 0);
                    PT_PK_FMA(fy_, hy, aiv_yz, tcy, 0, 0, 1, 0);
                    PT_PK_FMA(fz_, hz, aiv_yz, tcz, 1, 0, 1, 0);
                }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int s = 2 * p + q;
                    // v_min/v_max return the other operand for a NaN: a NaN slab (0 * inf) constrains nothing
                    // (as instructions: the builtins would first quiet every operand that comes out of an asm statement)
                    t0[s] = pt_vmax3(nx_[q], ny_[q], pt_vmax(nz_[q], tminf));
                    const float t1 = pt_vmin3(fx_[q], fy_[q], pt_vmin(fz_[q], tmaxf));
                    hb |= (t1 < t0[s]) ? 0u : (1u << s);
                }
            }
            if (CAREFUL && !trust) {  // rays outside the analysed range visit everything
                hb = 0xfu;
                t0[0] = t0[1] = t0[2] = t0[3] = 0.0f;
            }
            // node_base, obj_base and meta from the low bytes of the half extents (ptcore.hip puts them there): the record's last 32
            // bytes are never requested
#define PT_LOWBYTES(a, n4) \
    (__builtin_amdgcn_perm(__float_as_uint(nd.h[a][1]), __float_as_uint(nd.h[a][0]), 0x0c0c0400u) | \
     ((n4) ? __builtin_amdgcn_perm(__float_as_uint(nd.h[a][3]), __float_as_uint(nd.h[a][2]), 0x04000c0cu) \
           : __builtin_amdgcn_perm(0u, __float_as_uint(nd.h[a][2]), 0x0c000c0cu)))
            const uint32_t meta = PT_LOWBYTES(2, false);
            const uint32_t oh = hb & (meta >> 12) & 0xfu;  // object children pierced
            // internal children nearest first: sort keys = entry parameter (>= 0, so its bits order like the
            // value) with 2*slot in the low three bits; 0xffffffff = not a candidate
            uint32_t k0, k1, k2, k3;
            {
                uint32_t key[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const uint32_t bits = (__float_as_uint(t0[s]) & ~7u) | (uint32_t)(2 * s);
                    // a candidate only when the slot is pierced AND an internal node (meta bits 8-11)
                    key[s] = ((hb & (meta >> 8)) & (1u << s)) ? bits : 0xffffffffu;
                }
                // 5-exchange network
                uint32_t a0 = key[0] < key[1] ? key[0] : key[1], a1 = key[0] < key[1] ? key[1] : key[0];
                uint32_t a2 = key[2] < key[3] ? key[2] : key[3], a3 = key[2] < key[3] ? key[3] : key[2];
                k0 = a0 < a2 ? a0 : a2;
                const uint32_t m0 = a0 < a2 ? a2 : a0;
                k3 = a1 < a3 ? a3 : a1;
                const uint32_t m1 = a1 < a3 ? a1 : a3;
                k1 = m0 < m1 ? m0 : m1;
                k2 = m0 < m1 ? m1 : m0;
            }
            const int nbase = (int)PT_LOWBYTES(0, true);
#define PT_CHILD(k) (nbase + (int)((meta >> ((k) & 7u)) & 3u))
            if (k1 != 0xffffffffu) {  // sorted: k2 and k3 can only be candidates when k1 is
                if (k2 != 0xffffffffu) {
                    if (k3 != 0xffffffffu) { stack[sp * PT_BLOCK] = PT_CHILD(k3); sp++; }
                    stack[sp * PT_BLOCK] = PT_CHILD(k2);
                    sp++;
                }
                stack[sp * PT_BLOCK] = PT_CHILD(k1);
                sp++;
            }
            if (k0 != 0xffffffffu) {
                cur = PT_CHILD(k0);
            } else if (sp > 0) {
                sp--;
                cur = stack[sp * PT_BLOCK];
            } else {
                cur = -1;
            }
#undef PT_CHILD
            if (oh != 0) {
                pend = oh;
                pend_meta = meta;
                pend_base = (int)PT_LOWBYTES(1, true);
            }
#undef PT_LOWBYTES
            PH_END(SEC_COSINE)
            }
        }
        // ---- exact tests of the objects gathered at the last node
        if (pend != 0) {
            PH_BEGIN(SEC_NSPH)
            if (PROF) n_leaf++;
            do {
                PH_BEGIN(SEC_UNITDIR)  // BVH runs: wave-level executions and lanes of one exact test (cycles: up to the test itself)
                PH_END(SEC_UNITDIR)
                const uint32_t s = (uint32_t)__builtin_ctz(pend);
                pend &= pend - 1;
                const BvhObj &bo = bobjs[pend_base + (int)((pend_meta >> (2u * s)) & 3u)];
                if (PROF) ph.exec[SEC_NBOX]++;  // exact object tests (lane count)
                const int kind = bo.o.kind & 0xff;
                if (mode != 0 && !(bo.o.kind & 0x100)) continue;
                const int i = bo.index;
                double t = 0;
                bool valid;
                const bool is_box = kind == KIND_BOX;
                if (is_box)
                    valid = box_exact<true>(bo.o.a[0], bo.o.a[1], bo.o.a[2], bo.o.b[0], bo.o.b[1], bo.o.b[2], r, ivx, ivy, ivz, tmin,
                                      ptm::max_float64(), t);
                else
                    valid = sphere_exact_shared<false>(bo.o.a[0], bo.o.a[1], bo.o.a[2], bo.o.radius_sq, r, a, ya, tmin, tmax, t);
                if (valid && wins(mode, is_box, i, t, best, best_is_box, tmax) &&
                    (mode == 0 || exit_candidate_ok(bo.o, kind, r, t))) {
                    best = i;
                    tmax = t;
                    best_is_box = is_box;
                    tmaxf = (float)(tmax - ts);
                    tmaxf += __builtin_fabsf(tmaxf) * 4.8e-7f;
                }
            } while (pend != 0 && (!single || leaving));
            PH_END(SEC_NSPH)
        }
    }
    PH_END(SEC_BROAD)
    if (cur >= 0) {  // unfinished: keep the walk for the next trip
        S.live = true;
        S.cur = cur;
        S.sp = sp;
        S.best = best;
        S.tmax = tmax;
        S.best_is_box = best_is_box;
        return false;
    }
    if (PROF) {
        // histogram of object batches per scan: bins <4, <16, <64, <256, <1024, >=1024 (closest-hit scans in
        // the `exec` counters of three otherwise unused section ids and their `cyc` words, exit searches in `lanes`)
        const int bin = n_leaf < 4 ? 0 : n_leaf < 16 ? 1 : n_leaf < 64 ? 2 : n_leaf < 256 ? 3 : n_leaf < 1024 ? 4 : 5;
        if (bin == 5 && ph.dbg) {  // sample one very long traversal
            ph.dbg[-4] = 1;
            ph.dbg[0] = n_leaf;
            ph.dbg[1] = ptm::to_bits(tmax);
            ph.dbg[2] = ptm::to_bits((double)trust);
            ph.dbg[3] = (unsigned long long)mode;
            ph.dbg[4] = ptm::to_bits(r.ox); ph.dbg[5] = ptm::to_bits(r.oy); ph.dbg[6] = ptm::to_bits(r.oz);
            ph.dbg[7] = ptm::to_bits(r.dx); ph.dbg[8] = ptm::to_bits(r.dy); ph.dbg[9] = ptm::to_bits(r.dz);
        }
        const int id = bin < 2 ? SEC_HIST0 : bin < 4 ? SEC_HIST1 : SEC_HIST2;
        if (mode == 0) {
            if (bin & 1) ph.cyc[id]++;
            else ph.exec[id]++;
        } else if (!(bin & 1)) {
            ph.lanes[id]++;
        }
    }
    return true;
}

// One radiance record: r, g, b and a pad that makes it one whole 32-byte sector per path ending.  (Round 4 measured the alternatives on C4, same box: packed 24-byte
// records take resolve_kernel from 11.5 to 8.7 ms per frame and cost the trace passes 2 - 4 ms -- two stores per ending, records across sectors --,
// profiles/r04_radiance_stride_ab.txt; the non-temporal hint on these stores, on the ray planes and on the queue entries -- all written once and read once,
// gigabytes apart -- stays within the noise, profiles/r04_stream_nt_ab.txt.)
__device__ __forceinline__ void store_radiance(double *L, size_t job, double x, double y, double z) {
    reinterpret_cast<double4 *>(L)[job] = make_double4(x, y, z, 0.0);
}

// Ray generation pre-pass: one thread per job of the chunk, all lanes busy and neighbouring
// lanes on neighbouring pixels.  Per job: stream init, u then v (renderer.go:182-183),
// camera.getRay with the lens rejection loop (camera.go:60-74, math.go:74-84).  Writes the primary
// ray (6 doubles, SoA), the stream state after the draws, and the number of draws used (0xffff marks a
// job whose pixel lies outside the frame).
__global__ __launch_bounds__(PT_BLOCK) void raygen_kernel(const DevFrame F, const DevCamera cam, double *__restrict__ ray,
                                                            unsigned long long *__restrict__ ray_rng,
                                                            uint16_t *__restrict__ ray_ndraw) {
    const uint32_t myjob = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (myjob >= F.njobs) return;
    // job -> (tile, sub-block, sample, pixel)
    const uint32_t p = myjob & 63u;
    const uint32_t q = __builtin_amdgcn_readfirstlane(myjob >> 6);  // the wave's row of 64 jobs: tile, sub-block and sample are wave-uniform (scalar unit)
    const uint32_t blk = q / F.S;
    const uint32_t sl = q - blk * F.S;
    const uint32_t lt = blk >> 4, sb = blk & 15u;
    const uint32_t t = (uint32_t)F.shard_index + lt * (uint32_t)F.shard_count;
    const uint32_t ty = t / (uint32_t)F.ntx, tx = t - ty * (uint32_t)F.ntx;
    const uint32_t x = tx * 32u + (sb & 3u) * 8u + (p & 7u);
    const uint32_t y = ty * 32u + (sb >> 2) * 8u + (p >> 3);
    if (!(x < (uint32_t)F.width && y < (uint32_t)F.height)) {
        ray_ndraw[myjob] = 0xffffu;
        return;
    }
    uint32_t nd = 0;
    const uint64_t pixel = (uint64_t)y * (uint64_t)(uint32_t)F.width + (uint64_t)x;
    uint64_t rs = ptm::stream_init(F.seed_key, pixel, (uint64_t)(F.s0 + sl));
#define RG_DRAW(var) const double var = ptm::stream_next(rs); nd++;
    RG_DRAW(xi_u)
    RG_DRAW(xi_v)
    const double u = ((double)x + xi_u) * F.inv_width;
    const double vv = ((F.height_m1 - (double)y) + xi_v) * F.inv_height;
    const double tx_ = cam.lower_left[0] + cam.horizontal[0] * u;
    const double ty_ = cam.lower_left[1] + cam.horizontal[1] * u;
    const double tz_ = cam.lower_left[2] + cam.horizontal[2] * u;
    const double ax = tx_ + cam.vertical[0] * vv;
    const double ay = ty_ + cam.vertical[1] * vv;
    const double az = tz_ + cam.vertical[2] * vv;
    double ox, oy, oz, dx, dy, dz;
    if (cam.lens_radius > 0) {
        double rx, ry, rz;
        for (;;) {  // randomInUnitSphere
            RG_DRAW(d0)
            RG_DRAW(d1)
            RG_DRAW(d2)
            rx = d0 * 2 - 1;
            ry = d1 * 2 - 1;
            rz = d2 * 2 - 1;
            const double lenSq = rx * rx + ry * ry + rz * rz;
            if (lenSq >= 1.0) continue;
            break;
        }
        rx = rx * cam.lens_radius;
        ry = ry * cam.lens_radius;
        const double offx = cam.u[0] * rx + cam.v[0] * ry;
        const double offy = cam.u[1] * rx + cam.v[1] * ry;
        const double offz = cam.u[2] * rx + cam.v[2] * ry;
        ox = cam.origin[0] + offx;
        oy = cam.origin[1] + offy;
        oz = cam.origin[2] + offz;
        dx = (ax - cam.origin[0]) - offx;
        dy = (ay - cam.origin[1]) - offy;
        dz = (az - cam.origin[2]) - offz;
    } else {
        ox = cam.origin[0];
        oy = cam.origin[1];
        oz = cam.origin[2];
        dx = ax - cam.origin[0];
        dy = ay - cam.origin[1];
        dz = az - cam.origin[2];
    }
#undef RG_DRAW
    const size_t nj = F.njobs;
    ray[myjob] = ox;
    ray[nj + myjob] = oy;
    ray[2 * nj + myjob] = oz;
    ray[3 * nj + myjob] = dx;
    ray[4 * nj + myjob] = dy;
    ray[5 * nj + myjob] = dz;
    ray_rng[myjob] = rs;
    ray_ndraw[myjob] = (uint16_t)(nd < 0xfffeu ? nd : 0xfffeu);
}

// Ray generation for a thin-lens camera (lens_radius > 0).  The lens sample is a rejection loop (randomInUnitSphere,
// math.go:74-84: 1.9 attempts on average, the unluckiest of 64 lanes needs 6-7), and in raygen_kernel a wave waits for its
// unluckiest lane on every job: 2.75 ms of a 4.4 ms launch.  Here a wave owns PT_RG_ROWS rows of 64 jobs and a lane walks
// down its column: as soon as one job's sample is accepted it starts on the job below, so the wave waits for the largest
// SUM of attempts over a column (~14 for four rows instead of 4 x 6.5).  Three phases, all per-job data through LDS so
// that every global store is a full coalesced row: (A) lockstep: stream init, u, v, the point on the focus plane;
// (B) the rejection walk; (C) lockstep: lens offset, ray, stores.  Same draws in the same order per job as
// raygen_kernel (renderer.go:182-183, camera.go:60-74).
#ifndef PT_RG_ROWS
#define PT_RG_ROWS 2  // (4 in round 2; with the cheaper sample stream of round 3 the kernel is bound by its stores, and two rows = 26 KB of LDS = six
                      // blocks per CU beat four rows' better balance: 26.0 against 28.2 ms per C4 frame, 3 rows 26.9; profiles/r03_raygen_rows_ab.txt)
#endif
__global__ __launch_bounds__(PT_BLOCK) void raygen_lens_kernel(const DevFrame F, const DevCamera cam, double *__restrict__ ray,
                                                                 unsigned long long *__restrict__ ray_rng,
                                                                 uint16_t *__restrict__ ray_ndraw) {
    // per job, plane by plane (lane-minor: conflict-free): the point on the focus plane, the accepted lens sample, the
    // stream state and the draws so far (0xffffffff: pixel outside the frame, or no such job); 13 KB per row
    __shared__ double s_ax[PT_RG_ROWS][PT_BLOCK], s_ay[PT_RG_ROWS][PT_BLOCK], s_az[PT_RG_ROWS][PT_BLOCK];
    __shared__ double s_rx[PT_RG_ROWS][PT_BLOCK], s_ry[PT_RG_ROWS][PT_BLOCK];
    __shared__ unsigned long long s_rs[PT_RG_ROWS][PT_BLOCK];
    __shared__ uint32_t s_nd[PT_RG_ROWS][PT_BLOCK];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (PT_WAVE - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * PT_BLOCK + tid) >> 6);
    const uint32_t row0 = wave * PT_RG_ROWS;  // rows of 64 consecutive jobs
    // ---- A: everything up to the lens sample, row by row
#pragma unroll
    for (int k = 0; k < PT_RG_ROWS; k++) {
        const uint32_t myjob = (row0 + (uint32_t)k) * 64u + lane;
        uint32_t nd0 = 0xffffffffu;
        if (myjob < F.njobs) {
            const uint32_t p = lane;
            const uint32_t q = row0 + (uint32_t)k;
            const uint32_t blk = q / F.S;
            const uint32_t sl = q - blk * F.S;
            const uint32_t lt = blk >> 4, sb = blk & 15u;
            const uint32_t t = (uint32_t)F.shard_index + lt * (uint32_t)F.shard_count;
            const uint32_t ty = t / (uint32_t)F.ntx, tx = t - ty * (uint32_t)F.ntx;
            const uint32_t x = tx * 32u + (sb & 3u) * 8u + (p & 7u);
            const uint32_t y = ty * 32u + (sb >> 2) * 8u + (p >> 3);
            if (x < (uint32_t)F.width && y < (uint32_t)F.height) {
                const uint64_t pixel = (uint64_t)y * (uint64_t)(uint32_t)F.width + (uint64_t)x;
                uint64_t rs = ptm::stream_init(F.seed_key, pixel, (uint64_t)(F.s0 + sl));
                const double xi_u = ptm::stream_next(rs);
                const double xi_v = ptm::stream_next(rs);
                const double u = ((double)x + xi_u) * F.inv_width;
                const double vv = ((F.height_m1 - (double)y) + xi_v) * F.inv_height;
                const double tx_ = cam.lower_left[0] + cam.horizontal[0] * u;
                const double ty_ = cam.lower_left[1] + cam.horizontal[1] * u;
                const double tz_ = cam.lower_left[2] + cam.horizontal[2] * u;
                s_ax[k][tid] = tx_ + cam.vertical[0] * vv;
                s_ay[k][tid] = ty_ + cam.vertical[1] * vv;
                s_az[k][tid] = tz_ + cam.vertical[2] * vv;
                s_rs[k][tid] = rs;
                nd0 = 2;
            }
        }
        s_nd[k][tid] = nd0;
    }
    // ---- B: the rejection walk down the lane's column (only this lane touches its column: no barrier needed)
    {
        int k = 0;
        while (k < PT_RG_ROWS && s_nd[k][tid] == 0xffffffffu) k++;
        uint64_t rs = k < PT_RG_ROWS ? s_rs[k][tid] : 0;
        uint32_t nd = 2;
        while (k < PT_RG_ROWS) {
            const double d0 = ptm::stream_next(rs), d1 = ptm::stream_next(rs), d2 = ptm::stream_next(rs);
            nd += 3;
            const double rx = d0 * 2 - 1, ry = d1 * 2 - 1, rz = d2 * 2 - 1;
            const double lenSq = rx * rx + ry * ry + rz * rz;
            if (!(lenSq >= 1.0)) {  // accepted (randomInUnitSphere keeps drawing while lenSq >= 1)
                s_rx[k][tid] = rx;
                s_ry[k][tid] = ry;
                s_rs[k][tid] = rs;
                s_nd[k][tid] = nd;
                k++;
                while (k < PT_RG_ROWS && s_nd[k][tid] == 0xffffffffu) k++;
                if (k < PT_RG_ROWS) rs = s_rs[k][tid];
                nd = 2;
            }
        }
    }
    // ---- C: lens offset, ray, coalesced stores
    const size_t nj = F.njobs;
#pragma unroll
    for (int k = 0; k < PT_RG_ROWS; k++) {
        const uint32_t myjob = (row0 + (uint32_t)k) * 64u + lane;
        if (myjob >= F.njobs) continue;
        const uint32_t nd = s_nd[k][tid];
        if (nd == 0xffffffffu) {
            ray_ndraw[myjob] = 0xffffu;
            continue;
        }
        const double rx = s_rx[k][tid] * cam.lens_radius;
        const double ry = s_ry[k][tid] * cam.lens_radius;
        const double offx = cam.u[0] * rx + cam.v[0] * ry;
        const double offy = cam.u[1] * rx + cam.v[1] * ry;
        const double offz = cam.u[2] * rx + cam.v[2] * ry;
        ray[myjob] = cam.origin[0] + offx;
        ray[nj + myjob] = cam.origin[1] + offy;
        ray[2 * nj + myjob] = cam.origin[2] + offz;
        ray[3 * nj + myjob] = (s_ax[k][tid] - cam.origin[0]) - offx;
        ray[4 * nj + myjob] = (s_ay[k][tid] - cam.origin[1]) - offy;
        ray[5 * nj + myjob] = (s_az[k][tid] - cam.origin[2]) - offz;
        ray_rng[myjob] = s_rs[k][tid];
        ray_ndraw[myjob] = (uint16_t)(nd < 0xfffeu ? nd : 0xfffeu);
    }
}

// The same with the wave's jobs as ONE pool for the rejection walk (round 4).  Walking down a column, a wave waits for the largest
// sum of attempts over its columns -- ten wave-level attempts for two rows where the jobs need 3.8 on average.  Here the lane whose
// sample was accepted takes the next job of the pool nobody has started (ballot + mbcnt rank past a wave-uniform cursor), so the
// wave's attempts are the pool's total over 64 lanes plus one job's tail: 6 per two rows with four rows in the pool.  The point on
// the focus plane stays in the registers of the lane that computes it in (A) and uses it in (C); LDS holds what crosses lanes --
// stream state, accepted sample, draw count (26 B per job: four rows = 26 KB per block, six blocks per CU as before).
#ifndef PT_RG_POOL_ROWS
#define PT_RG_POOL_ROWS 4
#endif
__global__ __launch_bounds__(PT_BLOCK) void raygen_lens_pool_kernel(const DevFrame F, const DevCamera cam, double *__restrict__ ray,
                                                                      unsigned long long *__restrict__ ray_rng,
                                                                      uint16_t *__restrict__ ray_ndraw) {
    constexpr uint32_t R = PT_RG_POOL_ROWS, NJ = R * PT_WAVE;  // jobs in a wave's pool
    __shared__ unsigned long long s_rs[PT_BLOCK / PT_WAVE][NJ];
    __shared__ double s_rx[PT_BLOCK / PT_WAVE][NJ], s_ry[PT_BLOCK / PT_WAVE][NJ];
    __shared__ uint16_t s_nd[PT_BLOCK / PT_WAVE][NJ];  // draws so far, saturating at 0xfffe; 0xffff: pixel outside the frame, or no such job
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (PT_WAVE - 1), wib = tid >> 6;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * PT_BLOCK + tid) >> 6);  // (said so: a row's tile, sub-block and sample -- two integer divisions -- are then the scalar unit's)
    const uint32_t row0 = wave * R;  // rows of 64 consecutive jobs
    unsigned long long *const p_rs = s_rs[wib];
    double *const p_rx = s_rx[wib], *const p_ry = s_ry[wib];
    uint16_t *const p_nd = s_nd[wib];
    // ---- A: everything up to the lens sample, row by row (lockstep)
    double ax[R], ay[R], az[R];
    const uint32_t blk0 = row0 / F.S, sl0 = row0 - blk0 * F.S;  // one division per wave, the rows after the first follow by carry
#pragma unroll
    for (uint32_t k = 0; k < R; k++) {
        const uint32_t myjob = (row0 + k) * 64u + lane;
        uint16_t nd0 = 0xffffu;
        ax[k] = 0; ay[k] = 0; az[k] = 0;
        if (myjob < F.njobs) {
            const uint32_t p = lane;
            uint32_t sl = sl0 + k, blk = blk0;  // row0 + k = blk * S + sl
            while (sl >= F.S) { sl -= F.S; blk++; }
            const uint32_t lt = blk >> 4, sb = blk & 15u;
            const uint32_t t = (uint32_t)F.shard_index + lt * (uint32_t)F.shard_count;
            const uint32_t ty = t / (uint32_t)F.ntx, tx = t - ty * (uint32_t)F.ntx;
            const uint32_t x = tx * 32u + (sb & 3u) * 8u + (p & 7u);
            const uint32_t y = ty * 32u + (sb >> 2) * 8u + (p >> 3);
            if (x < (uint32_t)F.width && y < (uint32_t)F.height) {
                const uint64_t pixel = (uint64_t)y * (uint64_t)(uint32_t)F.width + (uint64_t)x;
                uint64_t rs = ptm::stream_init(F.seed_key, pixel, (uint64_t)(F.s0 + sl));
                const double xi_u = ptm::stream_next(rs);
                const double xi_v = ptm::stream_next(rs);
                const double u = ((double)x + xi_u) * F.inv_width;
                const double vv = ((F.height_m1 - (double)y) + xi_v) * F.inv_height;
                const double tx_ = cam.lower_left[0] + cam.horizontal[0] * u;
                const double ty_ = cam.lower_left[1] + cam.horizontal[1] * u;
                const double tz_ = cam.lower_left[2] + cam.horizontal[2] * u;
                ax[k] = tx_ + cam.vertical[0] * vv;
                ay[k] = ty_ + cam.vertical[1] * vv;
                az[k] = tz_ + cam.vertical[2] * vv;
                p_rs[k * 64u + lane] = rs;
                nd0 = 2;
            }
        }
        p_nd[k * 64u + lane] = nd0;
    }
    // (only this wave touches its pool, and a wave's LDS operations complete in order: a fence for the compiler, no barrier)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // ---- B: the rejection walk over the pool
    {
        uint32_t j = lane;   // the job this lane works on
        uint32_t next = 64;  // first job of the pool nobody has started (wave-uniform)
        uint32_t nd = p_nd[j];
        uint64_t rs = nd != 0xffffu ? p_rs[j] : 0;
        for (;;) {
            const bool live = j < NJ;
            if (__ballot(live) == 0) break;
            bool done = false;
            if (live) {
                done = nd == 0xffffu;  // nothing to draw for this job
                if (!done) {
                    const double d0 = ptm::stream_next(rs), d1 = ptm::stream_next(rs), d2 = ptm::stream_next(rs);
                    nd = nd + 3u < 0xfffeu ? nd + 3u : 0xfffeu;
                    const double rx = d0 * 2 - 1, ry = d1 * 2 - 1, rz = d2 * 2 - 1;
                    const double lenSq = rx * rx + ry * ry + rz * rz;
                    if (!(lenSq >= 1.0)) {  // accepted (randomInUnitSphere keeps drawing while lenSq >= 1)
                        p_rx[j] = rx;
                        p_ry[j] = ry;
                        p_rs[j] = rs;
                        p_nd[j] = (uint16_t)nd;
                        done = true;
                    }
                }
            }
            const uint64_t m = __ballot(done);
            if (m) {
                if (done) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    j = next + rank;
                    if (j < NJ) {
                        nd = p_nd[j];
                        if (nd != 0xffffu) rs = p_rs[j];
                    }
                }
                next += (uint32_t)__popcll(m);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // ---- C: lens offset, ray, coalesced stores (lockstep)
    const size_t nj = F.njobs;
#pragma unroll
    for (uint32_t k = 0; k < R; k++) {
        const uint32_t myjob = (row0 + k) * 64u + lane;
        if (myjob >= F.njobs) continue;
        const uint32_t nd = p_nd[k * 64u + lane];
        if (nd == 0xffffu) {
            ray_ndraw[myjob] = 0xffffu;
            continue;
        }
        const double rx = p_rx[k * 64u + lane] * cam.lens_radius;
        const double ry = p_ry[k * 64u + lane] * cam.lens_radius;
        const double offx = cam.u[0] * rx + cam.v[0] * ry;
        const double offy = cam.u[1] * rx + cam.v[1] * ry;
        const double offz = cam.u[2] * rx + cam.v[2] * ry;
        ray[myjob] = cam.origin[0] + offx;
        ray[nj + myjob] = cam.origin[1] + offy;
        ray[2 * nj + myjob] = cam.origin[2] + offz;
        ray[3 * nj + myjob] = (ax[k] - cam.origin[0]) - offx;
        ray[4 * nj + myjob] = (ay[k] - cam.origin[1]) - offy;
        ray[5 * nj + myjob] = (az[k] - cam.origin[2]) - offz;
        ray_rng[myjob] = p_rs[k * 64u + lane];
        ray_ndraw[myjob] = (uint16_t)nd;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Shading pieces shared by trace_kernel (all-in-one form) and glass_kernel.  Every expression is the reference's,
// in the reference's association order.

// One draw of the sample stream (random.go:27-34) with the bookkeeping both kernels keep.
template <bool STATS>
__device__ __forceinline__ double draw_next(uint64_t &rs, uint32_t &c_draw, uint32_t &j_draw) {
    c_draw++;
    if (STATS) j_draw++;
    return ptm::stream_next(rs);
}

// material.scatter, dielectric branch (materials.go:162-200) on the unit incoming direction u, the face normal n and
// the mirror direction rf = reflectVec(u, n).  1/ior and r0 come precomputed per material (DevMat): the same IEEE
// operations on the same operands the reference repeats on every hit.
template <bool STATS>
__device__ __forceinline__ void dielectric_scatter(const DevMat &m, bool ff, double ux, double uy, double uz, double nx, double ny,
                                                   double nz, double rfx, double rfy, double rfz, uint64_t &rs, uint32_t &c_draw,
                                                   uint32_t &j_draw, double &ndx, double &ndy, double &ndz) {
    ndx = rfx; ndy = rfy; ndz = rfz;
    const double ratio = ff ? m.inv_ior : m.ior;
    const double cosTheta = dev_go_min(-(ux * nx + uy * ny + uz * nz), 1.0);
    const double sinTheta = ptm::f_sqrt(1.0 - cosTheta * cosTheta);
    const bool cannot = ratio * sinTheta > 1.0;
    const double r0 = ff ? m.r0_front : m.r0_back;
    const double reflectProb = r0 + (1 - r0) * ptm::go_pow5(1 - cosTheta);
    bool reflects = cannot;
    if (!cannot) {  // Go's || short-circuit: the draw happens only here
        const double xi = draw_next<STATS>(rs, c_draw, j_draw);
        reflects = reflectProb > xi;
    }
    if (!reflects) {  // refractVec, math.go:48-64
        const double ct = dev_go_min(-ux * nx - uy * ny - uz * nz, 1.0);
        double qx = ux + nx * ct, qy = uy + ny * ct, qz = uz + nz * ct;
        qx *= ratio; qy *= ratio; qz *= ratio;
        const double perpLenSq = qx * qx + qy * qy + qz * qz;
        const double par = -ptm::f_sqrt(ptm::f_abs(1.0 - perpLenSq));
        ndx = qx + nx * par; ndy = qy + ny * par; ndz = qz + nz * par;
    }
}

// After an exit search (renderer.go:352-370): Beer-Lambert attenuation over the distance travelled inside and the
// origin moved to the exit point; the entry hit point is the ray origin.
__device__ __forceinline__ void exit_post(const DevMat &m, int best, double tmax, double &ox, double &oy, double &oz, double dx,
                                          double dy, double dz, double &attx, double &atty, double &attz) {
    if (best >= 0) {
        const double px = ox + dx * tmax, py = oy + dy * tmax, pz = oz + dz * tmax;
        const double ex = px - ox, ey = py - oy, ez = pz - oz;
        const double distance = ptm::f_sqrt(ex * ex + ey * ey + ez * ez);
        if (m.absorbs) {
            attx = ptm::go_exp(-m.absorption[0] * distance);
            atty = ptm::go_exp(-m.absorption[1] * distance);
            attz = ptm::go_exp(-m.absorption[2] * distance);
        }
        ox = px; oy = py; oz = pz;
    }
}

// Russian roulette on the last three levels and the step to the next level (renderer.go:375-403 with
// renderer.go:287-289 of the callee).  Returns true when the path ends here (it contributes `emitted` = 0).
template <bool STATS>
__device__ __forceinline__ bool roulette_advance(int &depth, double attx, double atty, double attz, double &Tx, double &Ty, double &Tz,
                                                 uint64_t &rs, uint32_t &c_draw, uint32_t &j_draw) {
    bool finished = false;
    if (depth <= 3) {  // renderer.go:375-393
        const double maxAtt = dev_go_max(attx, dev_go_max(atty, attz));
        if (maxAtt < 1e-6) {
            finished = true;
        } else {
            const double rrProb = dev_go_min(maxAtt, 0.95);
            const double xi = draw_next<STATS>(rs, c_draw, j_draw);
            if (xi > rrProb) {
                finished = true;
            } else {  // three divisions by the same probability (1e-6 <= rrProb <= 0.95)
                const double yp = div_recip(rrProb);
                attx = div_shared(attx, rrProb, yp);
                atty = div_shared(atty, rrProb, yp);
                attz = div_shared(attz, rrProb, yp);
            }
        }
    }
    if (!finished) {
        Tx *= attx; Ty *= atty; Tz *= attz;
        depth--;
        if (depth <= 0) finished = true;  // renderer.go:287-289 contributes zero
    }
    return finished;
}

// What the surface does with the path at its closest hit (renderer.go:308-319): the hit record of the winner
// (objects.go:63-88, :114-132, :181-221), `emitted` and material.scatter (materials.go:67-224).
//   finished     the path ends here with radiance `term` (an emissive surface: its emission; a failed scatter: 0)
//   exit_search  (GLASS only) a dielectric front face: the way out must be found before roulette (renderer.go:316-319),
//                exit_mat = the material
//   otherwise    o, d hold the scattered ray, att its attenuation: Russian roulette comes next
// GLASS = false: the caller never passes a dielectric hit (split passes park those for glass_kernel).
template <bool STATS, bool GLASS>
__device__ __forceinline__ void shade_hit(const DevObj &o, const DevMat *s_mat, double tmax, double &ox, double &oy, double &oz, double &dx,
                                          double &dy, double &dz, uint64_t &rs, uint32_t &c_draw, uint32_t &j_draw, bool &finished,
                                          double &termx, double &termy, double &termz, double &attx, double &atty, double &attz,
                                          bool &exit_search, int &exit_mat) {
            const int kind = o.kind & 0xff;
        const double px = ox + dx * tmax, py = oy + dy * tmax, pz = oz + dz * tmax;
        double nx, ny, nz;
        outward_normal(o, kind, px, py, pz, nx, ny, nz);
        const bool ff = (dx * nx + dy * ny + dz * nz) < 0;
        if (!ff) { nx = -nx; ny = -ny; nz = -nz; }
        const int mi = o.mat;
        const DevMat &m = s_mat[mi];
        const int typ = m.typ;
        if (typ == MAT_EMISSIVE) {  // materials.go:67-72, :202-203
            finished = true;
            termx = m.emit[0]; termy = m.emit[1]; termz = m.emit[2];
        } else {
            // unit direction for the specular kinds (materials.go:102-109, :175-182, :207-214)
            double ux = 0, uy = 0, uz = 0, rfx = 0, rfy = 0, rfz = 0;
            bool zero_dir = false;
            if (typ != MAT_LAMBERT) {
                const double dirLen = ptm::f_sqrt(dx * dx + dy * dy + dz * dz);
                if (dirLen == 0) {
                    zero_dir = true;
                } else {
                    const double invLen = 1.0 / dirLen;
                    ux = dx * invLen; uy = dy * invLen; uz = dz * invLen;
                    reflect_vec(ux, uy, uz, nx, ny, nz, rfx, rfy, rfz);
                }
            }
            if (zero_dir) {
                finished = true;  // scatter fails -> emitted (0)
            } else {
                double ndx = rfx, ndy = rfy, ndz = rfz;  // mirror / smooth metal / reflecting glass
                const bool cosine = (typ == MAT_LAMBERT) || (typ == MAT_METAL && m.rough > 1e-6);
                if (cosine) {
                    // randomCosineDirection, math.go:94-131, about the normal (lambert)
                    // or about the mirror direction (rough metal, materials.go:119)
                    const double wx = (typ == MAT_LAMBERT) ? nx : rfx;
                    const double wy = (typ == MAT_LAMBERT) ? ny : rfy;
                    const double wz = (typ == MAT_LAMBERT) ? nz : rfz;
                    const double r1 = draw_next<STATS>(rs, c_draw, j_draw);
                    const double r2 = draw_next<STATS>(rs, c_draw, j_draw);
                    const double phi = 6.283185307179586 * r1;
                    const double cosTheta = ptm::f_sqrt(r2);
                    const double sinTheta = ptm::f_sqrt(1.0 - r2);
                    const bool xmajor = ptm::f_abs(wx) > 0.9;
                    const double hx = xmajor ? 0.0 : 1.0, hy = xmajor ? 1.0 : 0.0, hz = 0.0;
                    // vVec = unit(w x h), uVec = vVec x w
                    double cx = wy * hz - wz * hy;
                    double cy = wz * hx - wx * hz;
                    double cz = wx * hy - wy * hx;
                    const double cl = ptm::f_sqrt(cx * cx + cy * cy + cz * cz);
                    if (cl != 0) {
                        const double inv = 1.0 / cl;
                        cx = cx * inv; cy = cy * inv; cz = cz * inv;
                    }
                    const double bx = cy * wz - cz * wy;
                    const double by = cz * wx - cx * wz;
                    const double bz = cx * wy - cy * wx;
                    double sn, cs;
                    ptm::sincos_pos(phi, &sn, &cs);
                    const double lx = sinTheta * cs, ly = sinTheta * sn, lz = cosTheta;
                    double sx = lx * bx + ly * cx + lz * wx;
                    double sy = lx * by + ly * cy + lz * wy;
                    double sz = lx * bz + ly * cz + lz * wz;
                    if (typ == MAT_LAMBERT) {
                        if (m.rough > 1e-6) {  // materials.go:84-91
                            double qx, qy, qz;
                            for (;;) {
                                const double d0 = draw_next<STATS>(rs, c_draw, j_draw);
                                const double d1 = draw_next<STATS>(rs, c_draw, j_draw);
                                const double d2 = draw_next<STATS>(rs, c_draw, j_draw);
                                qx = d0 * 2 - 1; qy = d1 * 2 - 1; qz = d2 * 2 - 1;
                                if (qx * qx + qy * qy + qz * qz >= 1.0) continue;
                                break;
                            }
                            sx += qx * m.rough * 0.1;
                            sy += qy * m.rough * 0.1;
                            sz += qz * m.rough * 0.1;
                            const double l = ptm::f_sqrt(sx * sx + sy * sy + sz * sz);
                            if (l != 0) {
                                const double inv = 1.0 / l;
                                sx = sx * inv; sy = sy * inv; sz = sz * inv;
                            }
                        }
                        ndx = sx; ndy = sy; ndz = sz;
                    } else {
                        // materials.go:121-148
                        const double alpha = m.rough_sq;
                        double mx = rfx * (1.0 - alpha) + sx * alpha;
                        double my = rfy * (1.0 - alpha) + sy * alpha;
                        double mz = rfz * (1.0 - alpha) + sz * alpha;
                        const double lenSq = mx * mx + my * my + mz * mz;
                        if (lenSq < 1e-8) {
                            mx = rfx; my = rfy; mz = rfz;
                        } else {
                            const double inv = 1.0 / ptm::f_sqrt(lenSq);
                            mx *= inv; my *= inv; mz *= inv;
                        }
                        const double dot = mx * nx + my * ny + mz * nz;
                        if (dot <= 0) { mx = rfx; my = rfy; mz = rfz; }
                        ndx = mx; ndy = my; ndz = mz;
                    }
                }
                if (GLASS && typ == MAT_DIELECTRIC) {  // materials.go:162-200
                    dielectric_scatter<STATS>(m, ff, ux, uy, uz, nx, ny, nz, rfx, rfy, rfz, rs, c_draw, j_draw, ndx, ndy, ndz);
                } else {
                    attx = m.albedo[0]; atty = m.albedo[1]; attz = m.albedo[2];
                }
                // scattered ray starts at the hit point (no offset)
                ox = px; oy = py; oz = pz;
                dx = ndx; dy = ndy; dz = ndz;
                if (GLASS && typ == MAT_DIELECTRIC && ff) {
                    exit_search = true;  // renderer.go:316-319: find the way out before roulette
                    exit_mat = mi;
                }
            }
        }
}

// Hides where a (wave-uniform) pointer came from, so that what is read through it is re-read at the point of use
// instead of being kept in scalar registers from the kernel's prologue on.
template <typename P>
__device__ __forceinline__ P pt_launder(P p) {
    asm volatile("" : "+s"(p));
    return p;
}

// SPLIT: dielectric hits are not shaded here.  The lane parks the path in the glass queue (HBM) and takes the next
// job; glass_kernel handles them together, and the paths that go on come back through the continuation queue.
// The trace loop then never runs the dielectric branch, the exit search or its epilogue at 14 % of its lanes,
// and no lane spends a whole trip through the scan on an exit search.
// (launch bounds: the flat scans are asked to fit five waves per SIMD = 96 VGPRs, the BVH forms four = 128; so is the all-in-one form
// of the grouped scan, which at 96 registers spilled 19 - 34 of them to scratch and only ever runs as the tail pass of a chunk, over
// the few paths with more dielectric bounces than split rounds; the diagnostic form is not held to anything)
// FORM: 0 = all-in-one, a dielectric exit search is the lane's next trip through the scan (every strategy);
//       1 = split (above); 2 = all-in-one with the exit search NESTED: right after the shading of a dielectric front face, over the
//       dielectric objects' own records only -- the form of the pass behind the split rounds, where nearly every lane's hit is glass
//       (paths that creep through glass boxes): half of that pass's lane-trips were exit searches that each paid a whole
//       trip through the broad phase over EVERY record (bitmask scans only).
enum { FORM_ALL_IN_ONE = 0, FORM_SPLIT = 1, FORM_NESTED = 2 };
template <bool STATS, bool PROF, int SCAN, int FORM>
__global__ __launch_bounds__(PT_BLOCK, PROF ? 1
                                       : (SCAN == SCAN_BVH || SCAN == SCAN_VERIFY_BVH) ? PT_BVH_WAVES
                                       : ((SCAN == SCAN_BROAD_WIDE || SCAN == SCAN_VERIFY_WIDE) && FORM != FORM_SPLIT && PT_FLAT_WAVES > 4) ? 4
                                       : (SCAN == SCAN_BROAD && FORM == FORM_SPLIT && !STATS)                                                  ? PT_SPLIT_WAVES
                                       : (SCAN == SCAN_BROAD && FORM == FORM_NESTED && !STATS)                                                 ? PT_NESTED_WAVES
                                                                                                                               : PT_FLAT_WAVES) void trace_kernel(const TraceArgs A) {
    constexpr bool SPLIT = FORM == FORM_SPLIT;
    constexpr bool NEST = FORM == FORM_NESTED;
    static_assert(!NEST || SCAN == SCAN_BROAD || SCAN == SCAN_VERIFY || SCAN == SCAN_BROAD_WIDE || SCAN == SCAN_VERIFY_WIDE,
                  "the nested exit search exists for the bitmask scans");
    extern __shared__ __align__(16) unsigned char smem[];
    const DevFrame &F = A.F;
    const TraceBuffers &B = A.B;  // set-up and epilogue only; the loop reads the argument block through KA
    // The loop needs some 40 scalars of the argument block once per trip (sky colours, queue and buffer pointers) next
    // to the ones its scan loops use all the time.  Kept in SGPRs for the whole loop they do not fit, and the compiler
    // parks them in VGPR lanes: every v_readlane / v_writelane is a 4-cycle VALU instruction in a VALU-bound kernel.
    // KA re-derives the pointer to the argument block (kernarg segment, constant address space) through an opaque
    // asm, so those fields are re-read by scalar loads where they are used (SMEM issue, no VALU slot).
    typedef const TraceArgs __attribute__((address_space(4))) *ConstArgsPtr;
    const ConstArgsPtr ka = (ConstArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
#define KA (pt_launder(ka))
    constexpr bool BIG = (SCAN == SCAN_BVH || SCAN == SCAN_VERIFY_BVH);
    // small scenes: the world is staged in LDS (the winner look-up after the scan is per lane);
    // BVH scenes: LDS holds the traversal stacks and the world is read from HBM/L2
    DevObj *lds_obj = reinterpret_cast<DevObj *>(smem);
    DevMat *lds_mat = reinterpret_cast<DevMat *>(smem + (size_t)F.nobj * sizeof(DevObj));
    int *lds_stack = reinterpret_cast<int *>(smem);
    BvhNode *lds_nodes = reinterpret_cast<BvhNode *>(smem + (size_t)F.bvh_stack * PT_BLOCK * sizeof(int));
    if (BIG) {
        const uint64_t *gsrc = reinterpret_cast<const uint64_t *>(B.bvh_nodes);
        uint64_t *ldst = reinterpret_cast<uint64_t *>(lds_nodes);
        const int nw = F.bvh_lds_nodes * (int)(sizeof(BvhNode) / 8);
        for (int i = threadIdx.x; i < nw; i += PT_BLOCK) ldst[i] = gsrc[i];
        __syncthreads();
    }
    // object index of every broad-phase record (spheres, then boxes), for the per-lane narrow phase
    int *lds_kidx = reinterpret_cast<int *>(smem + (size_t)F.nobj * sizeof(DevObj) + (size_t)F.nmat * sizeof(DevMat));
    // single-group scans: the records' objects once more, in record order (behind the index tables, 16-byte aligned; ptcore.hip sizes the LDS for it)
    constexpr bool RO = (SCAN == SCAN_BROAD || SCAN == SCAN_VERIFY);
    DevObj *lds_rec = reinterpret_cast<DevObj *>(smem + (((size_t)F.nobj * sizeof(DevObj) + (size_t)F.nmat * sizeof(DevMat) +
                                                            (size_t)(F.n_bsph + F.n_bbox + F.n_dsph + F.n_dbox) * sizeof(int) + 15) & ~(size_t)15));
    if (!BIG) {
        const uint64_t *g0 = reinterpret_cast<const uint64_t *>(B.objs);
        uint64_t *l0 = reinterpret_cast<uint64_t *>(lds_obj);
        const int n0 = F.nobj * (int)(sizeof(DevObj) / 8);
        for (int i = threadIdx.x; i < n0; i += PT_BLOCK) l0[i] = g0[i];
        const uint64_t *g1 = reinterpret_cast<const uint64_t *>(B.mats);
        uint64_t *l1 = reinterpret_cast<uint64_t *>(lds_mat);
        const int n1 = F.nmat * (int)(sizeof(DevMat) / 8);
        for (int i = threadIdx.x; i < n1; i += PT_BLOCK) l1[i] = g1[i];
        if (SCAN == SCAN_BROAD || SCAN == SCAN_VERIFY || SCAN == SCAN_BROAD_WIDE || SCAN == SCAN_VERIFY_WIDE) {
            for (int i = threadIdx.x; i < F.n_bsph; i += PT_BLOCK) lds_kidx[pt_record_slot(i, F.n_bsph)] = B.bsph[i].index;
            for (int i = threadIdx.x; i < F.n_bbox; i += PT_BLOCK) lds_kidx[F.n_bsph + pt_record_slot(i, F.n_bbox)] = B.bbox[i].index;
            if (NEST) {  // record -> object of the dielectric-only lists (behind the tables of the full lists)
                int *kd = lds_kidx + F.n_bsph + F.n_bbox;
                for (int i = threadIdx.x; i < F.n_dsph; i += PT_BLOCK) kd[pt_record_slot(i, F.n_dsph)] = B.bsph_diel[i].index;
                for (int i = threadIdx.x; i < F.n_dbox; i += PT_BLOCK) kd[F.n_dsph + pt_record_slot(i, F.n_dbox)] = B.bbox_diel[i].index;
            }
        }
        if (RO) {  // the objects of the records again, in record order, file index in bits 16-31 of `kind` (BroadLists::rs / rb)
            uint64_t *lr = reinterpret_cast<uint64_t *>(lds_rec);
            const int nr = F.n_bsph + F.n_bbox, W8 = (int)(sizeof(DevObj) / 8);
            for (int q = threadIdx.x; q < nr * W8; q += PT_BLOCK) {
                const int rcd = q / W8, wd = q - rcd * W8;
                const bool sph = rcd < F.n_bsph;
                const int k = sph ? rcd : rcd - F.n_bsph;
                const int oi = sph ? B.bsph[k].index : B.bbox[k].index;
                const int slot = sph ? pt_record_slot(k, F.n_bsph) : F.n_bsph + pt_record_slot(k, F.n_bbox);
                uint64_t v = g0[(size_t)oi * W8 + wd];
                if (wd == W8 - 1) v = (v & ~0xffff0000ull) | ((uint64_t)(uint32_t)oi << 16);  // last word: kind (low half), mat (high half)
                lr[(size_t)slot * W8 + wd] = v;
            }
            if (NEST) {  // ... and those of the dielectric-only records of the nested exit search, behind them
                uint64_t *ld = reinterpret_cast<uint64_t *>(lds_rec + nr);
                const int nd = F.n_dsph + F.n_dbox;
                for (int q = threadIdx.x; q < nd * W8; q += PT_BLOCK) {
                    const int rcd = q / W8, wd = q - rcd * W8;
                    const bool sph = rcd < F.n_dsph;
                    const int k = sph ? rcd : rcd - F.n_dsph;
                    const int oi = sph ? B.bsph_diel[k].index : B.bbox_diel[k].index;
                    const int slot = sph ? pt_record_slot(k, F.n_dsph) : F.n_dsph + pt_record_slot(k, F.n_dbox);
                    uint64_t v = g0[(size_t)oi * W8 + wd];
                    if (wd == W8 - 1) v = (v & ~0xffff0000ull) | ((uint64_t)(uint32_t)oi << 16);
                    ld[(size_t)slot * W8 + wd] = v;
                }
            }
        }
        __syncthreads();
    }
    const DevObj *const s_obj = BIG ? B.objs : lds_obj;
    const DevMat *const s_mat = BIG ? B.mats : lds_mat;

    // The world is immutable for the whole launch: read it through the constant address
    // space so the wave-uniform scan index turns into scalar (SGPR) loads.
    typedef const DevObj __attribute__((address_space(4))) *ConstObjPtr;
    const ConstObjPtr g_obj = (ConstObjPtr)(B.objs);
    typedef const BroadSphere __attribute__((address_space(4))) *ConstSphPtr;
    typedef const BroadBox __attribute__((address_space(4))) *ConstBoxPtr;
    typedef const int32_t __attribute__((address_space(4))) *ConstIdxPtr;
    const ConstSphPtr g_bs = (ConstSphPtr)(B.bsph);
    const ConstBoxPtr g_bb = (ConstBoxPtr)(B.bbox);
    const ConstIdxPtr g_pl = (ConstIdxPtr)(B.plane_idx);
    const BroadLists<ConstSphPtr, ConstBoxPtr, RO> BL{g_bs, g_bb, F.n_bsph, F.n_bbox, F.sph_all, F.box_all, F.sph_diel, F.box_diel,
                                                      lds_kidx, lds_kidx + F.n_bsph, lds_rec, lds_rec + F.n_bsph};
    const uint32_t lane = threadIdx.x & (PT_WAVE - 1);
    // the shader clock this launch runs at (pt_stats.shader_clock_mhz): the first wave of the launch notes the shader-cycle and the
    // 100 MHz reference counters now and again when it retires (in memory, not in registers: the loop has none to spare)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        B.counters[44] = __builtin_amdgcn_s_memtime();
        B.counters[45] = __builtin_amdgcn_s_memrealtime();
    }
    const BvhNode *const bvh_nodes_ = B.bvh_nodes;
    const BvhObj *const bvh_objs_ = B.bvh_objs;

    // work items of this pass: continuation entries [0, n_cont), then fresh jobs [n_cont, n_cont + F.fresh)
    // (only the scans that have a split form ever see continuation entries)
    // (... and, since round 4, the BVH scans: primary_bvh_kernel -- pt_primary.h -- hands every path over through the queue)
    constexpr bool CONT = (SCAN == SCAN_BROAD || SCAN == SCAN_VERIFY || SCAN == SCAN_BROAD_WIDE || SCAN == SCAN_VERIFY_WIDE || BIG);
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const uint32_t n_cont_raw = CONT ? *(ConstU32Ptr)(B.cont_in) : 0u;
    const uint32_t n_cont = n_cont_raw < B.cont.cap ? n_cont_raw : B.cont.cap;
    const uint32_t n_items = n_cont + F.fresh;

    // per-lane path state
    bool active = false;
    int mode = 0;  // 0: closest-hit scan, 1: dielectric exit search (all-in-one form only)
    int depth = 0;
    int exit_mat = 0;
    uint32_t job = 0;
    uint64_t rs = 0;
    double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    double Tx = 1, Ty = 1, Tz = 1;
    uint32_t c_seg = 0, c_exit = 0, c_draw = 0, c_samples = 0;
    uint32_t j_seg = 0, j_draw = 0;
    uint32_t c_mismatch = 0;  // SCAN_VERIFY only
    uint32_t c_glass = 0, c_fin = 0, c_contin = 0;  // SPLIT only: paths parked in the glass queue / ended here / taken from the continuation queue
    TravState trav;           // BVH strategies only: an unfinished traversal of this lane's ray

    // wave-uniform item cursor
    uint32_t cur = 0, end = 0;
    bool exhausted = false;
    // wave-uniform window of glass-queue slots this wave has reserved (SPLIT).  One atomic on the queue's counter per
    // PT_QUEUE_BLOCK slots: a push per wave and trip on ONE address would be ~20 M same-address atomics per launch,
    // which that address cannot serve (measured: the pass ran 2.5x slower).  Slots a wave has reserved but not filled
    // when it retires are marked as holes (job = PT_HOLE) and skipped by glass_kernel.
    uint32_t g_cur = 0, g_end = 0;

    // diagnostic counters (registers; only materialised when PROF)
    uint32_t p_exec[SEC_COUNT], p_lanes[SEC_COUNT];
    unsigned long long p_cyc[SEC_COUNT];
    if (PROF) {
#pragma unroll
        for (int i = 0; i < SEC_COUNT; i++) { p_exec[i] = 0; p_lanes[i] = 0; p_cyc[i] = 0; }
    }
#define SEC_BEGIN(id)                                                                 \
    unsigned long long t_##id = 0;                                                    \
    bool lead_##id = false;                                                           \
    if (PROF) {                                                                       \
        const uint64_t m_ = __ballot(1);                                              \
        lead_##id = lane == (uint32_t)(__ffsll((long long)m_) - 1);                   \
        p_lanes[id]++;                                                                \
        if (lead_##id) { p_exec[id]++; t_##id = __builtin_amdgcn_s_memtime(); }       \
    }
#define SEC_END(id) \
    if (PROF && lead_##id) p_cyc[id] += __builtin_amdgcn_s_memtime() - t_##id;

#define PT_DRAW(var) double var = draw_next<STATS>(rs, c_draw, j_draw);

    for (;;) {
        SEC_BEGIN(SEC_ITER)
        // ------------------------------------------------------------ regeneration
        const uint64_t need = __ballot(!active);
        // (Refilling only once N lanes are idle -- VERDICT r03 item 3 -- was swept in round 4: N = 1 / 2 / 4 the same, 8 / 16 / 32 slower by 0.5 / 3 / 10 % on C4 and
        // 0.8 / 3.7 / 11 % on C3, profiles/r04_refill_min.txt: a scan costs the wave the same whatever the number of its lanes that take part, so an idle lane is a lost segment.)
        if (need != 0) {
            if (cur >= end && !exhausted) {
                // (Round 4 tried guided claims -- a claim's size falling with what is left of the queue, 1 / (4 x waves) of it down to one
                // row of 64, so that the waves of a launch end together: C4 533.3 against 529.8 ms per frame at 4 passes, 559.1 against
                // 551.2 at 13, profiles/r04_guided_claims_ab.txt.  A claim is an atomic round trip during which the wave's idle lanes
                // wait; many small ones cost more than the even finish gives.  Round 2 had the same answer for a fixed tail of 64-job claims.)
                const uint32_t want = F.claim;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(KA->B.queue, want);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= n_items) {
                    exhausted = true;
                } else {
                    cur = base;
                    end = (n_items - base < want) ? n_items : base + want;
                }
            }
            const uint32_t avail = end - cur;
            const uint32_t rank = lane_rank(need);
            const bool take = !active && rank < avail;
            const uint32_t nneed = (uint32_t)__popcll(need);
            const uint32_t item = cur + rank;
            const uint32_t cur0 = cur;  // (wave-uniform: the claimed range's addresses are a scalar base + the lane's rank)
            cur += nneed < avail ? nneed : avail;

            if (CONT && take && item < n_cont) {
                // a path that left glass_kernel: its whole state comes from the continuation queue
                const auto cq = &KA->B.cont;
                const size_t qc = cq->cap;
                // (addresses as the scalar base of the wave's claim + rank, see the fresh rays below)
                const uint32_t rk = rank & 63u;
                const size_t c0 = cur0;
                job = (cq->job + c0)[rk];
                active = job != PT_HOLE;  // a slot some wave of glass_kernel reserved and did not fill
                if (SPLIT && active) c_contin++;
                mode = 0;
                depth = (cq->depth + c0)[rk];
                const double *q0 = cq->d + c0;
                ox = q0[rk];
                oy = (q0 + qc)[rk];
                oz = (q0 + 2 * qc)[rk];
                dx = (q0 + 3 * qc)[rk];
                dy = (q0 + 4 * qc)[rk];
                dz = (q0 + 5 * qc)[rk];
                Tx = (q0 + 6 * qc)[rk];
                Ty = (q0 + 7 * qc)[rk];
                Tz = (q0 + 8 * qc)[rk];
                rs = (cq->rs + c0)[rk];
                if (STATS) { j_seg = cq->jseg[item]; j_draw = cq->jdraw[item]; }
            } else if (take) {
                // the primary ray of this job was generated by raygen_kernel (coherent pre-pass)
                const uint32_t myjob = item - n_cont;
                const auto kb = &KA->B;
                // addresses as (scalar base of the wave's claim) + rank: no 64-bit vector arithmetic per plane
                const uint32_t rk = rank & 63u;  // (rank < 64: said so, the offset fits the 32-bit lane offset of a scalar-base load)
                const int64_t jb = (int64_t)cur0 - (int64_t)n_cont;  // job of rank 0 (below zero while the claim still holds continuations)
                const uint32_t nd = (kb->ray_ndraw + jb)[rk];
                // the ray is fetched before nd is looked at (an out-of-frame job's slots exist too, their content is not used): one
                // round trip to memory per refill instead of two in a row
                // (assigned here, not under the test: the compiler would sink the loads back below it)
                // (Round 4 also tried to have these lines in L2 by then -- a wave requesting the rays of its claim 64 - 128 jobs ahead of its
                // refills, 29 fire-and-forget `global_load_lds` per 64 jobs: trace passes 436 -> 445 ms per C4 frame, profiles/r04_refill_prefetch_ab.txt;
                // the other five waves of the SIMD cover this round trip already, and the prefetch costs a uniform branch and address arithmetic on every refill.)
                const size_t nj = F.njobs;
                const double *r0 = kb->ray + jb;
                ox = r0[rk];
                oy = (r0 + nj)[rk];
                oz = (r0 + 2 * nj)[rk];
                dx = (r0 + 3 * nj)[rk];
                dy = (r0 + 4 * nj)[rk];
                dz = (r0 + 5 * nj)[rk];
                rs = (kb->ray_rng + jb)[rk];
                if (nd != 0xffffu && F.max_depth <= 0) {
                    // rayColorOpt returns black before any scan (renderer.go:287-289); the camera draws happened
                    c_samples++;
                    c_draw += nd;
                    store_radiance(kb->L, myjob, 0.0, 0.0, 0.0);
                    if (STATS) {
                        kb->job_seg[myjob] = 0;
                        kb->job_draw[myjob] = nd;
                    }
                } else if (nd != 0xffffu) {  // 0xffff: the job's pixel lies outside the frame (edge tile)
                    SEC_BEGIN(SEC_RAYGEN)
                    job = myjob;
                    active = true;
                    mode = 0;
                    depth = F.max_depth;
                    Tx = 1; Ty = 1; Tz = 1;
                    c_samples++;
                    c_draw += nd;
                    if (STATS) { j_seg = 0; j_draw = nd; }
                    SEC_END(SEC_RAYGEN)
                }
            }
            if (__ballot(active) == 0) {
                if (exhausted) break;  // queue drained and every lane idle
                // every job taken so far lay outside the frame (edge tiles: up to 256*S such jobs in a row):
                // the queue still holds work, go round again.  (Leaving here instead lost the last tiles of a
                // frame once more such claims than resident waves piled up at the end of the queue.)
                continue;
            }
        }

        bool finished = false;
        bool to_glass = false;  // SPLIT: the hit is a dielectric, the path leaves for the glass queue
        double termx = 0, termy = 0, termz = 0;

        int best = -1;
        double tmax = 0;
        bool scanned = true;  // false: this lane's BVH walk goes on in the next trip, nothing to shade yet
        bool fat = false;     // BVH path: a far-away ray whose widened bounds would take in a sixteenth of the scene or more
        if (active) {
            // -------------------------------------------------------- scan
            SEC_BEGIN(SEC_SCAN)
            const RayD ray{ox, oy, oz, dx, dy, dz};
            const ProfHooks ph{p_exec, p_lanes, p_cyc, lane, (PROF && BIG) ? B.counters + 8 : nullptr};
            if (SCAN == SCAN_UNIFORM) {
                scan_uniform(F, g_obj, ray, mode, best, tmax);
            } else {
                // The culled strategies assume every candidate t is a number and that the reference's FP64 tests
                // mean what the geometry says.  Rays with non-finite or absurd components (a 1-pixel-wide frame
                // divides by W-1 = 0, renderer.go:95) break the first; a wave holding one takes the plain sequential
                // scan, which IS the reference's loop.  Rays from astronomically far away (clip_ray) break the
                // second: the bitmask strategy treats them the same way, the BVH widens its bounds for them.
                const double a_ = dx * dx + dy * dy + dz * dz;
                constexpr bool WIDE = (SCAN == SCAN_BROAD_WIDE || SCAN == SCAN_VERIFY_WIDE);
                constexpr bool BITMASK = (SCAN == SCAN_BROAD || SCAN == SCAN_VERIFY || WIDE);
                const Clip clip = (BITMASK && PT_CLIP32) ? clip_ray32(F, ray, mode ? 0.0001 : 0.001) : clip_ray(F, ray, mode ? 0.0001 : 0.001);
                const bool tame = (a_ >= 1e-100) && (a_ <= 1e100) && (ptm::f_abs(ox) <= 1e100) &&
                                  (ptm::f_abs(oy) <= 1e100) && (ptm::f_abs(oz) <= 1e100) && !(BITMASK && clip.far);
                const bool plain = __ballot(!tame) != 0;
                constexpr bool VERIFY = (SCAN == SCAN_VERIFY || SCAN == SCAN_VERIFY_BVH || SCAN == SCAN_VERIFY_WIDE);
                if (plain) {
                    scan_uniform(F, g_obj, ray, mode, best, tmax);
                    trav.live = false;  // a complete answer: whatever walk was pending is obsolete
                } else {
                    if (WIDE)
                        scan_broad_narrow_wide<PROF, VERIFY, (SPLIT || NEST) ? 0 : -1>(F, g_obj, BL, g_pl, s_obj, ray, clip, mode, best, tmax, ph,
                                                                                       Plane0{KA->F.plane0_y, KA->F.plane0_index, KA->F.plane0_kind});
                    else if (BITMASK)
                        scan_broad_narrow<PROF, VERIFY, (SPLIT || NEST) ? 0 : -1>(F, g_obj, BL, g_pl, s_obj, ray, clip, mode, best, tmax, ph,
                                                                                  Plane0{KA->F.plane0_y, KA->F.plane0_index, KA->F.plane0_kind});
                    else if ((fat = !trav.live && clip.far && !clip.miss && clip.infl * 16.0 > F.scene_bound))
                        scanned = false;  // no walk for this one: the whole wave scans the world for it, below
                    else if (__ballot(clip.far || !bvh_ray_trusted(F, ray, clip, a_)) != 0)
                        scanned = scan_bvh<PROF, true>(F, g_obj, g_pl, bvh_nodes_, lds_nodes, bvh_objs_, lds_stack + threadIdx.x,
                                                       ray, clip, mode, trav, best, tmax, ph);
                    else
                        scanned = scan_bvh<PROF, false>(F, g_obj, g_pl, bvh_nodes_, lds_nodes, bvh_objs_, lds_stack + threadIdx.x,
                                                        ray, clip, mode, trav, best, tmax, ph);
                    if (VERIFY && scanned) {
                        int best2;
                        double tmax2;
                        scan_uniform(F, g_obj, ray, mode, best2, tmax2);
                        if (best != best2 || (best >= 0 && !(tmax == tmax2))) {
                            c_mismatch++;
                            unsigned long long *dbg = KA->B.counters + 8;  // one disagreeing segment (racy, any one will do)
                            dbg[0] = ((unsigned long long)(uint32_t)best << 32) | (uint32_t)best2;
                            dbg[1] = ptm::to_bits(tmax);
                            dbg[2] = ptm::to_bits(tmax2);
                            dbg[3] = (unsigned long long)mode;
                            dbg[4] = ptm::to_bits(ox); dbg[5] = ptm::to_bits(oy); dbg[6] = ptm::to_bits(oz);
                            dbg[7] = ptm::to_bits(dx); dbg[8] = ptm::to_bits(dy); dbg[9] = ptm::to_bits(dz);
                        }
                        best = best2;
                        tmax = tmax2;
                    }
                }
            }
            SEC_END(SEC_SCAN)
        }
        if (BIG) {  // vanishingly rare (one ray in 2 * 10^8 on the synthetic scenes): usually all this costs is the ballot
            uint64_t fm = __ballot(fat);
            while (fm != 0) {
                const int src = __ffsll((long long)fm) - 1;
                fm &= fm - 1;
                const LinearHit fh = scan_linear_wave(F.nobj, KA->B.objs, __shfl(ox, src, 64), __shfl(oy, src, 64), __shfl(oz, src, 64), __shfl(dx, src, 64),
                                                      __shfl(dy, src, 64), __shfl(dz, src, 64), __shfl(mode, src, 64), lane);
                if ((int)lane == src) {
                    best = fh.best;
                    tmax = fh.tmax;
                    scanned = true;
                    if (SCAN == SCAN_VERIFY_BVH) {  // the verify instantiation checks this path like the walks above
                        int best2;
                        double tmax2;
                        scan_uniform(F, g_obj, RayD{ox, oy, oz, dx, dy, dz}, mode, best2, tmax2);
                        if (best != best2 || (best >= 0 && !(tmax == tmax2))) c_mismatch++;
                        best = best2;
                        tmax = tmax2;
                    }
                }
                if (lane == 0) atomicAdd(KA->B.counters + 23, 1ull);
            }
        }
        if (active && scanned) {
            // -------------------------------------------------------- shade
            bool do_rr = false;
            bool nest_exit = false;  // NEST: a dielectric front face was shaded, its exit search comes now
            double attx = 1, atty = 1, attz = 1;
            if (SPLIT || NEST || mode == 0) {
                c_seg++;
                if (STATS) j_seg++;
                if (best < 0) {
                    // sky closure, renderer.go:56-92
                    SEC_BEGIN(SEC_SKY)
                    finished = true;
                    const auto sky = &KA->sky;
                    if (sky->kind == 1) {
                        const double dirLen = ptm::f_sqrt(dx * dx + dy * dy + dz * dz);
                        if (dirLen == 0) {
                            termx = sky->c0[0]; termy = sky->c0[1]; termz = sky->c0[2];
                        } else {
                            double tt = (dy / dirLen + 1.0) * 0.5;
                            if (tt < 0) tt = 0;
                            if (tt > 1) tt = 1;
                            termx = sky->c0[0] * (1 - tt) + sky->c1[0] * tt;
                            termy = sky->c0[1] * (1 - tt) + sky->c1[1] * tt;
                            termz = sky->c0[2] * (1 - tt) + sky->c1[2] * tt;
                        }
                    } else {
                        termx = sky->c0[0]; termy = sky->c0[1]; termz = sky->c0[2];
                    }
                    SEC_END(SEC_SKY)
                } else if (SPLIT && (s_obj[best].kind & 0x100)) {
                    to_glass = true;  // dielectric: shaded by glass_kernel
                } else {
                    SEC_BEGIN(SEC_HITREC)
                    bool exit_search = false;
                    shade_hit<STATS, !SPLIT>(s_obj[best], s_mat, tmax, ox, oy, oz, dx, dy, dz, rs, c_draw, j_draw, finished, termx, termy,
                                             termz, attx, atty, attz, exit_search, exit_mat);
                    if (exit_search) {
                        c_exit++;
                        if (NEST) nest_exit = true;
                        else mode = 1;
                    } else if (!finished) {
                        do_rr = true;
                    }
                    SEC_END(SEC_HITREC)
                }
            } else {
                // exit search done (renderer.go:352-370); the hit point of the entry is the ray origin
                SEC_BEGIN(SEC_EXITPOST)
                exit_post(s_mat[exit_mat], best, tmax, ox, oy, oz, dx, dy, dz, attx, atty, attz);
                mode = 0;
                do_rr = true;
                SEC_END(SEC_EXITPOST)
            }

            if (NEST && nest_exit) {
                // ---------------------------------------------------- the way out of the glass, before roulette (renderer.go:316-371):
                // the scattered ray against the dielectric objects' own records (as glass_kernel does for the split rounds)
                SEC_BEGIN(SEC_EXITPOST)
                const RayD eray{ox, oy, oz, dx, dy, dz};
                const ProfHooks eph{p_exec, p_lanes, p_cyc, lane, nullptr};
                int ebest = -1;
                double etmax = 0;
                const double ea = dx * dx + dy * dy + dz * dz;
                const Clip eclip = PT_CLIP32 ? clip_ray32(F, eray, 0.0001) : clip_ray(F, eray, 0.0001);
                // same guards as the main scan: untamed or far-away rays take the reference's plain loop
                const bool etame = (ea >= 1e-100) && (ea <= 1e100) && (ptm::f_abs(ox) <= 1e100) && (ptm::f_abs(oy) <= 1e100) &&
                                   (ptm::f_abs(oz) <= 1e100) && !eclip.far;
                if (__ballot(!etame) != 0) {
                    scan_uniform(F, g_obj, eray, 1, ebest, etmax);
                } else {
                    constexpr bool WIDE_ = (SCAN == SCAN_BROAD_WIDE || SCAN == SCAN_VERIFY_WIDE);
                    constexpr bool VERIFY_ = (SCAN == SCAN_VERIFY || SCAN == SCAN_VERIFY_WIDE);
                    const uint32_t all_s = F.n_dsph >= 32 ? 0xffffffffu : ((1u << F.n_dsph) - 1u), all_b = F.n_dbox >= 32 ? 0xffffffffu : ((1u << F.n_dbox) - 1u);
                    const BroadLists<ConstSphPtr, ConstBoxPtr, RO> BLd{(ConstSphPtr)KA->B.bsph_diel, (ConstBoxPtr)KA->B.bbox_diel, F.n_dsph, F.n_dbox, all_s, all_b, all_s, all_b,
                                                                       lds_kidx + F.n_bsph + F.n_bbox, lds_kidx + F.n_bsph + F.n_bbox + F.n_dsph,
                                                                       lds_rec + F.n_bsph + F.n_bbox, lds_rec + F.n_bsph + F.n_bbox + F.n_dsph};
                    if (WIDE_) scan_broad_narrow_wide<PROF, VERIFY_, 1>(F, g_obj, BLd, g_pl, s_obj, eray, eclip, 1, ebest, etmax, eph,
                                                                        Plane0{KA->F.plane0_y, KA->F.plane0_index, KA->F.plane0_kind});
                    else scan_broad_narrow<PROF, VERIFY_, 1>(F, g_obj, BLd, g_pl, s_obj, eray, eclip, 1, ebest, etmax, eph,
                                                             Plane0{KA->F.plane0_y, KA->F.plane0_index, KA->F.plane0_kind});
                    if (VERIFY_) {
                        int best2;
                        double tmax2;
                        scan_uniform(F, g_obj, eray, 1, best2, tmax2);
                        if (ebest != best2 || (ebest >= 0 && !(etmax == tmax2))) c_mismatch++;
                        ebest = best2;
                        etmax = tmax2;
                    }
                }
                exit_post(s_mat[exit_mat], ebest, etmax, ox, oy, oz, dx, dy, dz, attx, atty, attz);
                do_rr = true;
                SEC_END(SEC_EXITPOST)
            }

            // ------------------------------------------------------------ roulette + advance
            if (do_rr) {
                SEC_BEGIN(SEC_RR)
                finished = roulette_advance<STATS>(depth, attx, atty, attz, Tx, Ty, Tz, rs, c_draw, j_draw);
                SEC_END(SEC_RR)
            }

            if (finished) {
                SEC_BEGIN(SEC_FINISH)
                // one whole 32-byte record per job: lanes finish at different times, so a [3][njobs] layout
                // would dirty three partly written sectors per job
                const auto kb = &KA->B;
                // (requesting the pointer ahead of the shading code instead of here was measured: nothing, profiles/r04_plane0_ab.txt)
                store_radiance(kb->L, job, Tx * termx, Ty * termy, Tz * termz);
                if (STATS) {
                    kb->job_seg[job] = j_seg;
                    kb->job_draw[job] = j_draw;
                }
                active = false;
                if (SPLIT) c_fin++;
                SEC_END(SEC_FINISH)
            }
        }
        if (SPLIT) {
            // ---------------------------------------------------------------- dielectric hits leave for the glass queue
            const uint64_t pm = __ballot(to_glass);
            if (pm != 0) {
                const auto gq = &KA->B.glass;
                const uint32_t np = (uint32_t)__popcll(pm), room = g_end - g_cur;
                uint32_t nbase = 0;
                if (np > room) {  // the first `room` lanes fill the old window, the others start a new one
                    if (lane == 0) nbase = atomicAdd(gq->count, (uint32_t)PT_QUEUE_BLOCK);
                    nbase = __builtin_amdgcn_readfirstlane(nbase);
                }
                const uint32_t rank = lane_rank(pm);
                const uint32_t slot = rank < room ? g_cur + rank : nbase + (rank - room);
                const uint32_t g_cur0 = g_cur;
                if (np > room) {
                    g_cur = nbase + (np - room);
                    g_end = nbase + PT_QUEUE_BLOCK;
                } else {
                    g_cur += np;
                }
                if (to_glass && slot >= gq->cap) {  // cannot happen (queues hold every job plus every window): fail loudly, write nothing
                    atomicAdd(KA->B.counters + 19, 1ull);
                    active = false;
                } else if (to_glass) {
                    const size_t qc = gq->cap;
                    // entry `idx` past slot `base` of every plane
                    auto store_entry = [&](size_t base, uint32_t idx) {
                        double *q0 = gq->d + base;
                        q0[idx] = ox;
                        (q0 + qc)[idx] = oy;
                        (q0 + 2 * qc)[idx] = oz;
                        (q0 + 3 * qc)[idx] = dx;
                        (q0 + 4 * qc)[idx] = dy;
                        (q0 + 5 * qc)[idx] = dz;
                        (q0 + 6 * qc)[idx] = Tx;
                        (q0 + 7 * qc)[idx] = Ty;
                        (q0 + 8 * qc)[idx] = Tz;
                        (q0 + 9 * qc)[idx] = tmax;
                        (gq->rs + base)[idx] = rs;
                        (gq->job + base)[idx] = job;
                        (gq->depth + base)[idx] = depth;
                        (gq->best + base)[idx] = best;
                        if (STATS) { (gq->jseg + base)[idx] = j_seg; (gq->jdraw + base)[idx] = j_draw; }
                    };
                    // nearly every push fits the wave's window: the addresses are then a scalar base (the window cursor) + the lane's
                    // rank, which costs no 64-bit vector arithmetic per plane (14 planes); a push that straddles two windows goes by slot
                    if (np <= room) store_entry(g_cur0, rank & 63u);
                    else store_entry(0, slot);
                    active = false;
                    c_glass++;
                }
            }
        }
        SEC_END(SEC_ITER)
    }
    if (SPLIT) {  // what is left of this wave's window stays empty
        for (uint32_t s = g_cur + lane; s < g_end && s < B.glass.cap; s += PT_WAVE) B.glass.job[s] = PT_HOLE;
    }
#undef KA
#undef PT_DRAW
#undef SEC_BEGIN
#undef SEC_END
    if (!PROF) {
        // PTCORE_DEBUG_TIMELINE (diagnostics): when every wave of the launch retired, in ticks of the 100 MHz counter (the host prints
        // how far apart the waves of a launch finish; B.prof holds one word per wave of the widest launch then).  The pointer is re-read
        // from the kernarg segment here: kept in scalar registers through the loop it cost the headline kernel three more spilled SGPRs.
        unsigned long long *tl = pt_launder(ka)->B.prof;
        if (tl != nullptr && lane == 0) tl[(blockIdx.x * PT_BLOCK + threadIdx.x) >> 6] = __builtin_amdgcn_s_memrealtime();
    }
    if (PROF) {
#pragma unroll
        for (int i = 0; i < SEC_COUNT; i++) {
            atomicAdd(&B.prof[3 * i], (unsigned long long)p_exec[i]);
            atomicAdd(&B.prof[3 * i + 1], (unsigned long long)p_lanes[i]);
            atomicAdd(&B.prof[3 * i + 2], p_cyc[i]);
        }
    }

    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&B.counters[46], t1 - __hip_atomic_load(&B.counters[44], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        atomicAdd(&B.counters[47], r1 - __hip_atomic_load(&B.counters[45], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    // counters: one atomic per wave
    const uint32_t w_seg = wave_sum(c_seg), w_exit = wave_sum(c_exit), w_draw = wave_sum(c_draw),
                   w_samples = wave_sum(c_samples);
    if (lane == 0) {
        atomicAdd(&B.counters[0], (unsigned long long)w_seg);
        atomicAdd(&B.counters[1], (unsigned long long)w_exit);
        atomicAdd(&B.counters[2], (unsigned long long)w_draw);
        atomicAdd(&B.counters[3], (unsigned long long)w_samples);
    }
    if (SCAN == SCAN_VERIFY || SCAN == SCAN_VERIFY_BVH || SCAN == SCAN_VERIFY_WIDE) {
        const uint32_t w_mis = wave_sum(c_mismatch);
        if (lane == 0 && w_mis) atomicAdd(&B.counters[4], (unsigned long long)w_mis);
    }
    if (SPLIT) {
        const uint32_t w_glass = wave_sum(c_glass), w_fin = wave_sum(c_fin), w_contin = wave_sum(c_contin);
        if (lane == 0) {
            if (w_glass) atomicAdd(&B.counters[5], (unsigned long long)w_glass);
            if (w_fin) atomicAdd(&B.counters[18], (unsigned long long)w_fin);
            if (w_contin) atomicAdd(&B.counters[7], (unsigned long long)w_contin);
        }
    }
}

// The dielectric bounce of every path in the glass queue, all lanes on the same branch: hit record
// (objects.go:63-88, :181-221), material.scatter for glass (materials.go:162-200), the exit search over the
// dielectric objects (renderer.go:316-349) with its epilogue (renderer.go:352-370), Russian roulette and the step
// to the next level (renderer.go:375-403).  Paths that go on are appended to the continuation queue, paths that end
// write their radiance record (always zero here: glass emits nothing).  One entry per lane, grid-stride.
// VERIFY: the exit search is also done by the plain object-by-object loop and disagreements are counted.
// WIDE: more than 32 spheres or boxes in the scene: the exit search takes its (dielectric-only) records in groups of 32.
struct GlassArgs {  // the argument block of glass_kernel (one by-value kernel argument, i.e. the kernarg segment)
    DevFrame F;
    TraceBuffers B;
};
template <bool STATS, bool VERIFY, bool WIDE>
__global__ __launch_bounds__(PT_BLOCK, (WIDE && PT_FLAT_WAVES > 4) ? 4 : PT_FLAT_WAVES) void glass_kernel(const GlassArgs A) {  // (the grouped form spills 6 - 17 VGPRs at 96)
    extern __shared__ __align__(16) unsigned char smem[];
    const DevFrame &F = A.F;
    const TraceBuffers &B = A.B;  // set-up and epilogue only
    // The loop reads two dozen queue pointers once per entry; kept in SGPRs for the whole loop they do not fit next to what the
    // exit search holds, and the compiler parked 60 - 79 of them in VGPR lanes (v_readlane / v_writelane: 4-cycle VALU
    // instructions).  As in trace_kernel they are re-read from the kernarg segment where they are used (KB).
    typedef const GlassArgs __attribute__((address_space(4))) *ConstArgsPtr;
    const ConstArgsPtr ka = (ConstArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
#define KB (&pt_launder(ka)->B)
    DevObj *lds_obj = reinterpret_cast<DevObj *>(smem);
    DevMat *lds_mat = reinterpret_cast<DevMat *>(smem + (size_t)F.nobj * sizeof(DevObj));
    int *lds_kidx = reinterpret_cast<int *>(smem + (size_t)F.nobj * sizeof(DevObj) + (size_t)F.nmat * sizeof(DevMat));
    DevObj *lds_rec = reinterpret_cast<DevObj *>(smem + (((size_t)F.nobj * sizeof(DevObj) + (size_t)F.nmat * sizeof(DevMat) +
                                                            (size_t)(F.n_dsph + F.n_dbox) * sizeof(int) + 15) & ~(size_t)15));
    {
        const uint64_t *g0 = reinterpret_cast<const uint64_t *>(B.objs);
        uint64_t *l0 = reinterpret_cast<uint64_t *>(lds_obj);
        const int n0 = F.nobj * (int)(sizeof(DevObj) / 8);
        for (int i = threadIdx.x; i < n0; i += PT_BLOCK) l0[i] = g0[i];
        const uint64_t *g1 = reinterpret_cast<const uint64_t *>(B.mats);
        uint64_t *l1 = reinterpret_cast<uint64_t *>(lds_mat);
        const int n1 = F.nmat * (int)(sizeof(DevMat) / 8);
        for (int i = threadIdx.x; i < n1; i += PT_BLOCK) l1[i] = g1[i];
        for (int i = threadIdx.x; i < F.n_dsph; i += PT_BLOCK) lds_kidx[pt_record_slot(i, F.n_dsph)] = B.bsph_diel[i].index;
        for (int i = threadIdx.x; i < F.n_dbox; i += PT_BLOCK) lds_kidx[F.n_dsph + pt_record_slot(i, F.n_dbox)] = B.bbox_diel[i].index;
        if (!WIDE) {  // the dielectric records' objects again, in record order (BroadLists::rs / rb, see trace_kernel)
            uint64_t *ld = reinterpret_cast<uint64_t *>(lds_rec);
            const int nd = F.n_dsph + F.n_dbox, W8 = (int)(sizeof(DevObj) / 8);
            for (int q = threadIdx.x; q < nd * W8; q += PT_BLOCK) {
                const int rcd = q / W8, wd = q - rcd * W8;
                const bool sph = rcd < F.n_dsph;
                const int k = sph ? rcd : rcd - F.n_dsph;
                const int oi = sph ? B.bsph_diel[k].index : B.bbox_diel[k].index;
                const int slot = sph ? pt_record_slot(k, F.n_dsph) : F.n_dsph + pt_record_slot(k, F.n_dbox);
                uint64_t v = g0[(size_t)oi * W8 + wd];
                if (wd == W8 - 1) v = (v & ~0xffff0000ull) | ((uint64_t)(uint32_t)oi << 16);
                ld[(size_t)slot * W8 + wd] = v;
            }
        }
        __syncthreads();
    }
    typedef const DevObj __attribute__((address_space(4))) *ConstObjPtr;
    typedef const BroadSphere __attribute__((address_space(4))) *ConstSphPtr;
    typedef const BroadBox __attribute__((address_space(4))) *ConstBoxPtr;
    typedef const int32_t __attribute__((address_space(4))) *ConstIdxPtr;
    const ConstObjPtr g_obj = (ConstObjPtr)(B.objs);
    const ConstIdxPtr g_pl = (ConstIdxPtr)(B.plane_idx);
    const uint32_t all_s = F.n_dsph >= 32 ? 0xffffffffu : ((1u << F.n_dsph) - 1u), all_b = F.n_dbox >= 32 ? 0xffffffffu : ((1u << F.n_dbox) - 1u);
    const BroadLists<ConstSphPtr, ConstBoxPtr, !WIDE> BL{(ConstSphPtr)B.bsph_diel, (ConstBoxPtr)B.bbox_diel, F.n_dsph, F.n_dbox, all_s, all_b,
                                                         all_s, all_b, lds_kidx, lds_kidx + F.n_dsph, lds_rec, lds_rec + F.n_dsph};
    const uint32_t lane = threadIdx.x & (PT_WAVE - 1);
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const uint32_t n_raw = *(ConstU32Ptr)(B.glass.count);
    const uint32_t n = n_raw < B.glass.cap ? n_raw : B.glass.cap;
    uint32_t c_exit = 0, c_draw = 0, c_mismatch = 0, c_cont = 0;
    const size_t qg = B.glass.cap, qc = B.cont.cap;
    uint32_t q_cur = 0, q_end = 0;  // this wave's window of continuation slots (see trace_kernel: one atomic per window)

    // (the wave's first entry is wave-uniform: said so, every plane of the entry is then read at a scalar base + the lane's offset
    // instead of a 64-bit vector address per plane -- 14 planes in, 12 out per entry; 8 spilled SGPRs instead of 19.  No change in
    // time: profiles/r04_glass_ab.txt)
    for (uint32_t i0 = __builtin_amdgcn_readfirstlane(blockIdx.x * PT_BLOCK + (threadIdx.x & ~(PT_WAVE - 1u))); i0 < n; i0 += gridDim.x * PT_BLOCK) {
        const uint32_t i = i0 + lane;
        const auto gq = &KB->glass;
        const bool live = i < n && (gq->job + i0)[lane] != PT_HOLE;
        bool go_on = false;
        double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, Tx = 0, Ty = 0, Tz = 0;
        uint64_t rs = 0;
        uint32_t job = 0, j_seg = 0, j_draw = 0;
        int depth = 0;
        if (live) {
            const double *q0 = gq->d + i0;
            ox = q0[lane]; oy = (q0 + qg)[lane]; oz = (q0 + 2 * qg)[lane];
            dx = (q0 + 3 * qg)[lane]; dy = (q0 + 4 * qg)[lane]; dz = (q0 + 5 * qg)[lane];
            Tx = (q0 + 6 * qg)[lane]; Ty = (q0 + 7 * qg)[lane]; Tz = (q0 + 8 * qg)[lane];
            const double tmax = (q0 + 9 * qg)[lane];
            rs = (gq->rs + i0)[lane];
            job = (gq->job + i0)[lane];
            depth = (gq->depth + i0)[lane];
            const int best = (gq->best + i0)[lane];
            if (STATS) { j_seg = (gq->jseg + i0)[lane]; j_draw = (gq->jdraw + i0)[lane]; }

            // hit record of the winner
            const DevObj &o = lds_obj[best];
            const int kind = o.kind & 0xff;
            const double px = ox + dx * tmax, py = oy + dy * tmax, pz = oz + dz * tmax;
            double nx, ny, nz;
            outward_normal(o, kind, px, py, pz, nx, ny, nz);
            const bool ff = (dx * nx + dy * ny + dz * nz) < 0;
            if (!ff) { nx = -nx; ny = -ny; nz = -nz; }
            const int mi = o.mat;
            const DevMat &m = lds_mat[mi];
            bool finished = false;
            // materials.go:175-182
            const double dirLen = ptm::f_sqrt(dx * dx + dy * dy + dz * dz);
            if (dirLen == 0) {
                finished = true;  // scatter fails -> emitted (0)
            } else {
                const double invLen = 1.0 / dirLen;
                const double ux = dx * invLen, uy = dy * invLen, uz = dz * invLen;
                double rfx, rfy, rfz;
                reflect_vec(ux, uy, uz, nx, ny, nz, rfx, rfy, rfz);
                double ndx, ndy, ndz;
                dielectric_scatter<STATS>(m, ff, ux, uy, uz, nx, ny, nz, rfx, rfy, rfz, rs, c_draw, j_draw, ndx, ndy, ndz);
                // scattered ray starts at the hit point (no offset)
                ox = px; oy = py; oz = pz;
                dx = ndx; dy = ndy; dz = ndz;
                double attx = 1, atty = 1, attz = 1;
                if (ff) {  // renderer.go:316-371: the way out, before roulette
                    c_exit++;
                    const RayD ray{ox, oy, oz, dx, dy, dz};
                    const ProfHooks ph{nullptr, nullptr, nullptr, lane, nullptr};
                    int ebest = -1;
                    double etmax = 0;
                    // same guards as the trace kernel: untamed or far-away rays take the reference's plain loop
                    const double a_ = dx * dx + dy * dy + dz * dz;
                    const Clip clip = PT_CLIP32 ? clip_ray32(F, ray, 0.0001) : clip_ray(F, ray, 0.0001);
                    const bool tame = (a_ >= 1e-100) && (a_ <= 1e100) && (ptm::f_abs(ox) <= 1e100) && (ptm::f_abs(oy) <= 1e100) &&
                                      (ptm::f_abs(oz) <= 1e100) && !clip.far;
                    if (__ballot(!tame) != 0) {
                        scan_uniform(F, g_obj, ray, 1, ebest, etmax);
                    } else {
                        if (WIDE) scan_broad_narrow_wide<false, VERIFY, 1>(F, g_obj, BL, g_pl, lds_obj, ray, clip, 1, ebest, etmax, ph,
                                                                           Plane0{pt_launder(ka)->F.plane0_y, pt_launder(ka)->F.plane0_index, pt_launder(ka)->F.plane0_kind});
                        else scan_broad_narrow<false, VERIFY, 1>(F, g_obj, BL, g_pl, lds_obj, ray, clip, 1, ebest, etmax, ph,
                                                                 Plane0{pt_launder(ka)->F.plane0_y, pt_launder(ka)->F.plane0_index, pt_launder(ka)->F.plane0_kind});
                        if (VERIFY) {
                            int best2;
                            double tmax2;
                            scan_uniform(F, g_obj, ray, 1, best2, tmax2);
                            if (ebest != best2 || (ebest >= 0 && !(etmax == tmax2))) {
                                c_mismatch++;
                                unsigned long long *dbg = KB->counters + 8;
                                dbg[0] = ((unsigned long long)(uint32_t)ebest << 32) | (uint32_t)best2;
                                dbg[1] = ptm::to_bits(etmax);
                                dbg[2] = ptm::to_bits(tmax2);
                                dbg[3] = 1ull;
                                dbg[4] = ptm::to_bits(ox); dbg[5] = ptm::to_bits(oy); dbg[6] = ptm::to_bits(oz);
                                dbg[7] = ptm::to_bits(dx); dbg[8] = ptm::to_bits(dy); dbg[9] = ptm::to_bits(dz);
                            }
                            ebest = best2;
                            etmax = tmax2;
                        }
                    }
                    exit_post(m, ebest, etmax, ox, oy, oz, dx, dy, dz, attx, atty, attz);
                }
                finished = roulette_advance<STATS>(depth, attx, atty, attz, Tx, Ty, Tz, rs, c_draw, j_draw);
            }
            if (finished) {
                store_radiance(KB->L, job, Tx * 0.0, Ty * 0.0, Tz * 0.0);
                if (STATS) {
                    KB->job_seg[job] = j_seg;
                    KB->job_draw[job] = j_draw;
                }
            } else {
                go_on = true;
            }
        }
        const uint64_t pm = __ballot(go_on);
        if (pm != 0) {
            const uint32_t np = (uint32_t)__popcll(pm), room = q_end - q_cur;
            uint32_t nbase = 0;
            if (np > room) {
                if (lane == 0) nbase = atomicAdd(KB->cont.count, (uint32_t)PT_CONT_BLOCK);
                nbase = __builtin_amdgcn_readfirstlane(nbase);
            }
            const uint32_t rank = lane_rank(pm);
            const uint32_t slot = rank < room ? q_cur + rank : nbase + (rank - room);
            const uint32_t q_cur0 = q_cur;
            if (np > room) {
                q_cur = nbase + (np - room);
                q_end = nbase + PT_CONT_BLOCK;
            } else {
                q_cur += np;
            }
            if (go_on && slot >= B.cont.cap) {
                atomicAdd(KB->counters + 19, 1ull);  // cannot happen; never write outside the queue
            } else if (go_on) {
                const auto cq = &KB->cont;
                // entry `idx` past slot `base` of every plane (as trace_kernel's push: nearly every push fits the wave's window, and
                // its addresses are then the window cursor, a scalar, + the lane's rank)
                auto store_entry = [&](size_t base, uint32_t idx) {
                    double *q0 = cq->d + base;
                    q0[idx] = ox;
                    (q0 + qc)[idx] = oy;
                    (q0 + 2 * qc)[idx] = oz;
                    (q0 + 3 * qc)[idx] = dx;
                    (q0 + 4 * qc)[idx] = dy;
                    (q0 + 5 * qc)[idx] = dz;
                    (q0 + 6 * qc)[idx] = Tx;
                    (q0 + 7 * qc)[idx] = Ty;
                    (q0 + 8 * qc)[idx] = Tz;
                    (cq->rs + base)[idx] = rs;
                    (cq->job + base)[idx] = job;
                    (cq->depth + base)[idx] = depth;
                    if (STATS) { (cq->jseg + base)[idx] = j_seg; (cq->jdraw + base)[idx] = j_draw; }
                };
                if (np <= room) store_entry(q_cur0, rank & 63u);
                else store_entry(0, slot);
                c_cont++;
            }
        }
    }
    for (uint32_t s = q_cur + lane; s < q_end && s < B.cont.cap; s += PT_WAVE) B.cont.job[s] = PT_HOLE;  // the rest of the window stays empty
    const uint32_t w_exit = wave_sum(c_exit), w_draw = wave_sum(c_draw), w_cont = wave_sum(c_cont);
    if (lane == 0) {
        if (w_exit) atomicAdd(&B.counters[1], (unsigned long long)w_exit);
        if (w_draw) atomicAdd(&B.counters[2], (unsigned long long)w_draw);
        if (w_cont) atomicAdd(&B.counters[6], (unsigned long long)w_cont);
    }
#undef KB
    if (VERIFY) {
        const uint32_t w_mis = wave_sum(c_mismatch);
        if (lane == 0 && w_mis) atomicAdd(&B.counters[4], (unsigned long long)w_mis);
    }
}

// uint8(v) of renderer.go:218-220 after the clamp of :200-217; NaN -> 0 as on amd64.
__device__ __forceinline__ uint32_t quantise(double v) {
    if (v < 0) v = 0;
    else if (v > 255.999) v = 255.999;
    if (v != v) return 0;
    return (uint32_t)v;
}

struct ResolveArgs {
    const double *L;         // [njobs][4]: r, g, b, 0
    const uint32_t *job_seg; // [njobs] or null
    const uint32_t *job_draw;
    double *acc;             // [3][nslots]
    uint32_t *acc_seg;       // [nslots] or null
    uint32_t *acc_draw;
    uint8_t *tiles_rgba;     // [nlocal][32][32][4] or null (no finish)
    double *tiles_accum;     // [nlocal][32][32][3] or null
    uint32_t *tiles_seg;     // [nlocal][32][32] or null
    uint32_t *tiles_draw;
    uint32_t nslots;         // nlocal*1024
    uint32_t njobs;
    uint32_t S;
    int32_t first;           // 1: the running sum starts at zero
    int32_t finish;          // 1: write tiles_rgba / tiles_accum
    int32_t have_chunk;      // 0: no chunk to add (pure finish, pt_read)
    double inv_samples;      // 1/spp_done
    int32_t width, height, ntx, shard_index, shard_count;
};

__global__ __launch_bounds__(PT_BLOCK) void resolve_kernel(const ResolveArgs R) {
    const uint32_t slot = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (slot >= R.nslots) return;
    const uint32_t blk = slot >> 6, p = slot & 63u;
    const uint32_t lt = blk >> 4, sb = blk & 15u;
    const uint32_t t = (uint32_t)R.shard_index + lt * (uint32_t)R.shard_count;
    const uint32_t ty = t / (uint32_t)R.ntx, tx = t - ty * (uint32_t)R.ntx;
    const uint32_t lx = (sb & 3u) * 8u + (p & 7u), ly = (sb >> 2) * 8u + (p >> 3);
    const uint32_t x = tx * 32u + lx, y = ty * 32u + ly;
    const bool inside = x < (uint32_t)R.width && y < (uint32_t)R.height;

    double cx = 0, cy = 0, cz = 0;
    uint32_t nseg = 0, ndraw = 0;
    if (inside) {
        if (!R.first) {
            cx = R.acc[slot];
            cy = R.acc[(size_t)R.nslots + slot];
            cz = R.acc[2 * (size_t)R.nslots + slot];
            if (R.acc_seg) { nseg = R.acc_seg[slot]; ndraw = R.acc_draw[slot]; }
        }
        if (R.have_chunk) {
            const size_t base = (size_t)blk * R.S * 64u + p;
            for (uint32_t s = 0; s < R.S; s++) {  // col = col.add(sample), renderer.go:186, in sample order
                const size_t j = base + (size_t)s * 64u;
                const double4 l = reinterpret_cast<const double4 *>(R.L)[j];
                cx += l.x;
                cy += l.y;
                cz += l.z;
                if (R.job_seg) { nseg += R.job_seg[j]; ndraw += R.job_draw[j]; }
            }
            R.acc[slot] = cx;
            R.acc[(size_t)R.nslots + slot] = cy;
            R.acc[2 * (size_t)R.nslots + slot] = cz;
            if (R.acc_seg) { R.acc_seg[slot] = nseg; R.acc_draw[slot] = ndraw; }
        }
    }
    if (R.finish) {
        const size_t pix = (size_t)lt * 1024u + ly * 32u + lx;
        if (R.tiles_rgba) {
            uint32_t packed = 0;
            if (inside) {
                // renderer.go:190-221
                const double r = ptm::f_sqrt(cx * R.inv_samples) * 255.999;
                const double g = ptm::f_sqrt(cy * R.inv_samples) * 255.999;
                const double b = ptm::f_sqrt(cz * R.inv_samples) * 255.999;
                packed = quantise(r) | (quantise(g) << 8) | (quantise(b) << 16) | (255u << 24);
            }
            reinterpret_cast<uint32_t *>(R.tiles_rgba)[pix] = packed;
        }
        if (R.tiles_accum) {
            R.tiles_accum[3 * pix] = inside ? cx : 0.0;
            R.tiles_accum[3 * pix + 1] = inside ? cy : 0.0;
            R.tiles_accum[3 * pix + 2] = inside ? cz : 0.0;
        }
        if (R.tiles_seg) {
            R.tiles_seg[pix] = inside ? nseg : 0u;
            R.tiles_draw[pix] = inside ? ndraw : 0u;
        }
    }
}

struct UntileArgs {
    const uint8_t *tiles_rgba;   // concatenated per shard: shard k holds its tiles in local order
    const double *tiles_accum;   // or null
    const uint32_t *tiles_u32a;  // optional per-pixel u32 planes in tile order (stats)
    const uint32_t *tiles_u32b;
    uint8_t *rgba;               // row-major frame, `stride` bytes per row (or null)
    double *accum;               // width*height*3 (or null)
    uint32_t *u32a, *u32b;       // width*height (or null)
    int32_t width, height, ntx, nty, stride, shard_count;
    int32_t shard_stride_tiles;  // 0: shards are packed back to back
};

__global__ __launch_bounds__(PT_BLOCK) void untile_kernel(const UntileArgs U) {
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u);
    const uint32_t y = blockIdx.y * 32u + (threadIdx.x >> 5) + blockIdx.z * 8u;
    if (x >= (uint32_t)U.width || y >= (uint32_t)U.height) return;
    const uint32_t tx = x >> 5, ty = y >> 5;
    const uint32_t t = ty * (uint32_t)U.ntx + tx;
    const uint32_t ntiles = (uint32_t)U.ntx * (uint32_t)U.nty;
    const uint32_t k = t % (uint32_t)U.shard_count, lt = t / (uint32_t)U.shard_count;
    // tiles owned by shards 0..k-1
    const uint32_t q = ntiles / (uint32_t)U.shard_count, r = ntiles % (uint32_t)U.shard_count;
    const uint32_t before = U.shard_stride_tiles ? k * (uint32_t)U.shard_stride_tiles : k * q + (k < r ? k : r);
    const size_t pix = ((size_t)before + lt) * 1024u + (y & 31u) * 32u + (x & 31u);
    if (U.rgba)
        *reinterpret_cast<uint32_t *>(U.rgba + (size_t)y * (size_t)U.stride + (size_t)x * 4u) =
            reinterpret_cast<const uint32_t *>(U.tiles_rgba)[pix];
    const size_t o = (size_t)y * (size_t)U.width + x;
    if (U.accum) {
        U.accum[3 * o] = U.tiles_accum[3 * pix];
        U.accum[3 * o + 1] = U.tiles_accum[3 * pix + 1];
        U.accum[3 * o + 2] = U.tiles_accum[3 * pix + 2];
    }
    if (U.u32a) U.u32a[o] = U.tiles_u32a[pix];
    if (U.u32b) U.u32b[o] = U.tiles_u32b[pix];
}

// Self-test of div_shared against the compiler's IEEE division: `per_thread` operand pairs per thread, exponents drawn
// over the guarded ranges (denominator 2^+-340, numerator 2^+-300, every 16th numerator outside them or zero to take
// the fall-back), mantissas random or at their extremes.  out[0] += pairs whose bits differ.
__global__ __launch_bounds__(PT_BLOCK) void div_selftest_kernel(unsigned long long seed, uint32_t per_thread, unsigned long long *out) {
    uint64_t s = ptm::mix64(seed + (uint64_t)(blockIdx.x * PT_BLOCK + threadIdx.x) * PTM_GOLDEN);
    uint32_t bad = 0;
    for (uint32_t k = 0; k < per_thread; k++) {
        s += PTM_GOLDEN;
        const uint64_t h0 = ptm::mix64(s), h1 = ptm::mix64(s ^ 0x5851f42d4c957f2dULL), h2 = ptm::mix64(s + 0x14057b7ef767814fULL);
        const uint32_t sel = (uint32_t)(h2 >> 40) & 15u;
        uint64_t md = h0 & 0xfffffffffffffULL, mn = h1 & 0xfffffffffffffULL;
        if (((h2 >> 8) & 7u) == 0) md = 0;
        if (((h2 >> 11) & 7u) == 0) md = 0xfffffffffffffULL;
        if (((h2 >> 14) & 7u) == 0) mn = 0;
        if (((h2 >> 17) & 7u) == 0) mn = 0xfffffffffffffULL;
        const uint64_t ed = 1023u - 340u + (uint32_t)((h2 >> 20) % 681u);
        uint64_t en = 1023u - 300u + (uint32_t)((h2 >> 44) % 601u);
        if (sel == 0) en = (h1 >> 52) & 0x7ffu;  // anything, also denormals, infinities and NaNs: the guard must send them to `/`
        const double d = ptm::from_bits(((h0 >> 63) << 63) | (ed << 52) | md);
        double n = ptm::from_bits(((h1 >> 63) << 63) | (en << 52) | mn);
        if (sel == 1) n = (h1 >> 63) ? -0.0 : 0.0;
        const double q0 = n / d;
        const double q1 = div_shared(n, d, div_recip(d));
        const bool same = ptm::to_bits(q0) == ptm::to_bits(q1) || (q0 != q0 && q1 != q1);
        bad += same ? 0u : 1u;
        // ptm::f_sqrt (the refinement without the range scaling) against the compiler's IEEE square root: the numerator as it is
        // (either sign, any exponent every 16th time), its magnitude, and a magnitude pushed towards the scaling threshold 2^-767
        {
            const double x0 = n, x1 = ptm::f_abs(n), x2 = ptm::from_bits(((uint64_t)(200u + (uint32_t)((h2 >> 30) % 120u)) << 52) | mn);
            const double r0 = ptm::f_sqrt(x0), r1 = ptm::f_sqrt(x1), r2 = ptm::f_sqrt(x2);
            const double e0 = __builtin_sqrt(x0), e1 = __builtin_sqrt(x1), e2 = __builtin_sqrt(x2);
            const bool same_sq = (ptm::to_bits(r0) == ptm::to_bits(e0) || (r0 != r0 && e0 != e0)) && ptm::to_bits(r1) == ptm::to_bits(e1) &&
                                 ptm::to_bits(r2) == ptm::to_bits(e2);
            bad += same_sq ? 0u : 1u;
        }
        // math.Max / math.Min: the one-instruction forms against the spelled-out ones, on the same operands and on the
        // special values (zeros of both signs, infinities, NaN, 1, denormals) in every combination over the run
        const uint64_t sp[8] = {0x0ULL, 0x8000000000000000ULL, 0x7ff0000000000000ULL, 0xfff0000000000000ULL,
                                0x7ff8000000000001ULL, 0x3ff0000000000000ULL, 0x0000000000000001ULL, 0x800fffffffffffffULL};
        double mx = n, my = d;
        if (((h0 >> 3) & 3u) == 0) mx = ptm::from_bits(sp[(h0 >> 5) & 7u]);
        if (((h1 >> 3) & 3u) == 0) my = ptm::from_bits(sp[(h1 >> 5) & 7u]);
        const double a0 = ptm::go_max(mx, my), a1 = dev_go_max(mx, my), b0 = ptm::go_min(mx, my), b1 = dev_go_min(mx, my);
        const bool same_mm = (ptm::to_bits(a0) == ptm::to_bits(a1) || (a0 != a0 && a1 != a1)) &&
                             (ptm::to_bits(b0) == ptm::to_bits(b1) || (b0 != b0 && b1 != b1));
        bad += same_mm ? 0u : 1u;
    }
    const uint32_t w = wave_sum(bad);
    if ((threadIdx.x & (PT_WAVE - 1)) == 0 && w) atomicAdd(out, (unsigned long long)w);
}

// ---------------------------------------------------------------------------------------------
// Optional post-process passes of the reference's GPU backend.  They are NOT part of
// the CPU engine's look and are off unless asked for.  All three are per-pixel, bandwidth-bound passes
// over a tightly packed RGBA8 frame (row stride 4*width).

// acesTonemap, gpu.go:22-47
__device__ __forceinline__ float aces_tonemap(float x) {
    if (x <= 0) return 0;
    const double y = (double)x;
    const double num = y * (2.51 * y + 0.03);
    const double den = y * (2.43 * y + 0.59) + 0.14;
    if (den <= 0) return 0;
    double r = num / den;
    if (r < 0) r = 0;
    else if (r > 1) r = 1;
    return (float)r;
}

// gpu.go:2309-2350: clamp, ACES, sqrt gamma, uint8(g*255.0 + 0.5) in float32 arithmetic
__global__ __launch_bounds__(PT_BLOCK) void post_tonemap_kernel(const double *__restrict__ accum, int32_t spp,
                                                                  uint8_t *__restrict__ rgba, int32_t npix) {
    const int32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= npix) return;
    uint32_t packed = 255u << 24;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float lin = (float)(accum[(size_t)i * 3 + c] / (double)spp);
        if (lin < 0) lin = 0;
        const float tm = aces_tonemap(lin);
        float g = (float)ptm::f_sqrt((double)tm);
        if (g > 1) g = 1;
        float v = g * 255.0f;
        v = v + 0.5f;
        packed |= ((uint32_t)v & 0xffu) << (8 * c);
    }
    reinterpret_cast<uint32_t *>(rgba)[i] = packed;
}

// gpu.go:2355-2439: 3x3 bilateral filter on the 8-bit image (spatial sigma_s, range sigma_r in sRGB 0..1)
__global__ __launch_bounds__(PT_BLOCK) void post_bilateral_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int32_t w,
                                                                    int32_t h, double two_sigma_s2, double two_sigma_r2) {
    const int32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= w * h) return;
    const int32_t x = i % w, y = i / w;
    const uint32_t cpk = reinterpret_cast<const uint32_t *>(src)[i];
    const double cr = (double)(cpk & 0xffu) / 255.0, cg = (double)((cpk >> 8) & 0xffu) / 255.0,
                 cb = (double)((cpk >> 16) & 0xffu) / 255.0;
    double sumR = 0, sumG = 0, sumB = 0, sumW = 0;
    for (int ky = -1; ky <= 1; ky++) {
        const int ny = y + ky;
        if (ny < 0 || ny >= h) continue;
        for (int kx = -1; kx <= 1; kx++) {
            const int nx = x + kx;
            if (nx < 0 || nx >= w) continue;
            const uint32_t npk = reinterpret_cast<const uint32_t *>(src)[(size_t)ny * w + nx];
            const double nr = (double)(npk & 0xffu) / 255.0, ng = (double)((npk >> 8) & 0xffu) / 255.0,
                         nb = (double)((npk >> 16) & 0xffu) / 255.0;
            const double ds2 = (double)(kx * kx + ky * ky);
            const double dr = cr - nr, dg = cg - ng, dbb = cb - nb;
            const double dr2 = dr * dr + dg * dg + dbb * dbb;
            const double ws = ptm::go_exp(-ds2 / two_sigma_s2);
            const double wr = ptm::go_exp(-dr2 / two_sigma_r2);
            const double wgt = ws * wr;
            sumW += wgt;
            sumR += nr * wgt;
            sumG += ng * wgt;
            sumB += nb * wgt;
        }
    }
    uint32_t out = cpk;
    if (sumW > 0) {
        double v[3] = {sumR / sumW, sumG / sumW, sumB / sumW};
        out = 255u << 24;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double q = v[c];
            if (q < 0) q = 0;
            else if (q > 1) q = 1;
            out |= ((uint32_t)(q * 255.0 + 0.5) & 0xffu) << (8 * c);
        }
    }
    reinterpret_cast<uint32_t *>(dst)[i] = out;
}

// gpu.go:2444-2520: box average of radius 1..5 blended with the original by `strength`
__global__ __launch_bounds__(PT_BLOCK) void post_smooth_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int32_t w,
                                                                 int32_t h, int32_t rad, double str) {
    const int32_t i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= w * h) return;
    const int32_t x = i % w, y = i / w;
    double sumR = 0, sumG = 0, sumB = 0, count = 0;
    for (int ky = -rad; ky <= rad; ky++) {
        const int ny = y + ky;
        if (ny < 0 || ny >= h) continue;
        for (int kx = -rad; kx <= rad; kx++) {
            const int nx = x + kx;
            if (nx < 0 || nx >= w) continue;
            const uint32_t npk = reinterpret_cast<const uint32_t *>(src)[(size_t)ny * w + nx];
            sumR += (double)(npk & 0xffu);
            sumG += (double)((npk >> 8) & 0xffu);
            sumB += (double)((npk >> 16) & 0xffu);
            count++;
        }
    }
    const uint32_t cpk = reinterpret_cast<const uint32_t *>(src)[i];
    const double avg[3] = {sumR / count, sumG / count, sumB / count};
    uint32_t out = 255u << 24;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double o = (1 - str) * (double)((cpk >> (8 * c)) & 0xffu) + str * avg[c];
        if (o < 0) o = 0;
        else if (o > 255) o = 255;
        out |= ((uint32_t)(o + 0.5) & 0xffu) << (8 * c);
    }
    reinterpret_cast<uint32_t *>(dst)[i] = out;
}

}  // namespace ptk
