// pt_walk32.h -- the BVH walk taken out of the FP64 kernels (included by ptcore.hip after pt_wavefront.h).
//
// PTCORE_PIPELINE=walk32: the wavefront form of pt_wavefront.h with its traversal pass cut in two.
//
//   wf_walk32_kernel   walks the hierarchy in FP32 ONLY.  It never decides a hit: for every path of the level it lists the
//                      objects whose own (inflated, outward-rounded) FP32 box the ray pierces before `tmaxf` -- the
//                      candidates -- and shrinks `tmaxf` only by bounds that are CERTAIN: a ray that passes through the
//                      middle of an object (its CORE: a box shrunk by the margin on every side, the cube inscribed in a
//                      sphere shrunk by the margin; ptbvh::object_core) is hit by the reference's FP64 test no later than
//                      where it enters that core.  The cores live in a twin of every node (ptbvh::build_cores), so a core
//                      visit is the same instructions as a node visit.  No FP64 ray, reciprocals or path state are alive
//                      during the walk, so the kernel needs half the registers of scan_bvh and runs at twice the waves per
//                      SIMD; no lane ever waits for an exact test.  (Measured: slower than the FP64 loop all the same --
//                      more visits, a heavier visit, and the passes around it)
//   wf_shade32_kernel  the exact pass, one path per lane: the reference's FP64 tests (objects.go:37-61, :141-179) on the
//                      listed candidates only, the planes, `wins` (the order-free statement of renderer.go:297-302), then
//                      the shading of wf_shade_kernel.  wf_exit32_kernel likewise for the exit searches (renderer.go:329-349).
//   the slow list      rays the FP32 bounds were not analysed for (origins thousands of scene sizes away, where the
//                      reference's own sphere discriminant cancels; non-finite or absurd components), rays with more than
//                      PT_CAND_MAX candidates and walks deeper than the LDS stack are not walked here: their queue entries are
//                      listed, and wf_traverse_kernel (the FP64 traversal of pt_wavefront.h, with its widened bounds) answers them.
//
// Why the result is the reference's: the exact pass takes the reference's decision on every object the sequential loop
// could have accepted as the closest so far.  An object is skipped only if its box was never pierced before tmaxf, and
// tmaxf >= t_winner always: it only ever shrinks to the entry parameter of a core that lies inside an object by the
// margin m = B / 4096 (B bounds the scene), two orders of magnitude more than the FP32 rounding of the slab arithmetic
// (<= 1.2e-6 B for origins within 4 B, the same analysis as the broad phase), so the FP64 test of that object
// reports a hit at or before it, and every object that could tie with the winner has its inflated box entered m earlier.
#pragma once

#include "pt_wavefront.h"

namespace ptk {

#define PT_CAND_MAX 8                 // candidates listed per ray; more: the slow list
#define PT_CAND_EXACT 0x80000000u     // cand_n: the entry was answered by wf_traverse_kernel (hit / tmax planes hold the exact answer)
#define PT_WALK_STACK 16              // LDS stack entries per lane (16 KiB per block: 8 blocks per CU); a walk that needs more goes to the slow list
#ifndef PT_WALK_WAVES
#define PT_WALK_WAVES 8               // waves per SIMD the walk is compiled for (64 VGPRs)
#endif
#define PT_WALK_CORE 0x40000000       // `cur` bit: the visit is to the node's core twin (bvh_cores), not to the node

struct Walk32Args {
    WfArgs W;
    const BvhNode *cores;   // [n_bvh_nodes] core twin of every node: slot s holds a box INSIDE the object of slot s (by the margin), or nothing
    uint32_t *cand_ids;     // [cap][PT_CAND_MAX] indices into bvh_objs
    uint32_t *cand_n;       // [cap] number of candidates of the entry, or PT_CAND_EXACT
    uint32_t *slow_list;    // [cap] entries left to wf_traverse_kernel
    uint32_t *slow_count;   // number of entries in slow_list
    int32_t min_lanes;      // the visit loop refills its idle lanes below this many walking ones
    unsigned long long *diag;  // [16] diagnostic counters (DIAG instantiation only)
};

// One v_pk_fma_f32 over two node slots with the ray constant taken from ONE half of a register pair (op_sel picks the
// half for both results): nine per-ray constants live in five pairs instead of nine (the compiler's own splat handling
// gives every constant a pair of its own, 18 registers in scan_bvh).
// (PT_PK_FMA: pt_kernels.h)
// v_max3 / v_min3 on the packed results (as instructions: through the builtins the compiler first quiets every operand
// that comes out of an asm statement, one v_max_f32 x, x each).  Like v_max / v_min they return the other operand for a
// NaN: a NaN slab parameter (0 * inf) constrains nothing.
__device__ __forceinline__ float pt_max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float pt_min3(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// The FP32 walk.  MODE 0: closest-hit candidates over the main tree, MODE 1: exit-search candidates over the dielectric tree.
//
// One loop iteration = one visit for every walking lane, the same instructions for all of them:
//   node visit  slab test of the four slots (FP32 boxes, inflated and rounded outward); pierced object slots are listed as
//               candidates (a store each); pierced internal slots are ordered nearest first, the nearest is visited next,
//               the others are pushed.  If an object slot was pierced, the lane's next visit is to the node's CORE TWIN.
//   core visit  the twin holds, slot by slot, a box that lies INSIDE the object of that slot by the margin (a box shrunk
//               by m on every side; the cube inscribed in a sphere shrunk by m).  A core pierced after tMin is a certain hit
//               of the reference's FP64 test, no later than where the ray enters the core: tmaxf shrinks to that.  (For a
//               sphere only when the ray starts outside its box: from inside the reference may return the far root.)
// Exit searches (MODE 1) list candidates only: what ends an exit search also depends on the face and the distance
// (renderer.go:333-347), which no FP32 bound decides.
// DIAG (PTCORE_WALK_STATS=1): counts visits, refills, candidates and hand-overs into K.diag (see walk32_diag_names in ptcore.hip).
template <int MODE, bool DIAG = false>
__global__ __launch_bounds__(PT_BLOCK, PT_WALK_WAVES) void wf_walk32_kernel(const Walk32Args K) {
    __shared__ int s_stack[PT_WALK_STACK * PT_BLOCK];
    const WfArgs &A = K.W;
    const DevFrame &F = A.F;
    const TraceBuffers &B = A.B;
    const PathQueue &Q = A.qin;
    int *const stack = s_stack + threadIdx.x;  // this lane's column, stride PT_BLOCK
    typedef const DevObj __attribute__((address_space(4))) *ConstObjPtr;
    typedef const int32_t __attribute__((address_space(4))) *ConstIdxPtr;
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const ConstObjPtr g_obj = (ConstObjPtr)(B.objs);
    const ConstIdxPtr g_pl = (ConstIdxPtr)(B.plane_idx);
    const BvhNode *__restrict__ const nodes = B.bvh_nodes;
    const BvhNode *__restrict__ const cores = K.cores;
    const uint32_t lane = threadIdx.x & (PT_WAVE - 1);
    const uint32_t n_count = *(ConstU32Ptr)(Q.count);
    const uint32_t n_items = n_count < Q.cap ? n_count : Q.cap;
    const size_t qc = Q.cap;
    const int root = MODE ? F.bvh_root_exit : F.bvh_root;
    const double tmin = MODE ? 0.0001 : 0.001;

    // per-lane walk state: FP32 only
    int cur = -1;          // node to visit next (| PT_WALK_CORE: its core twin); -1: this lane has no ray
    int sp = 0;
    uint32_t e = 0;        // queue entry of the ray
    uint32_t cnt = 0;      // candidates listed so far
    uint32_t core_ok = 0;  // core visit: the slots whose core may bound the ray (boxes; spheres the ray starts outside of)
    v2f iv_xy = {0, 0}, iv_z_aiv_x = {0, 0}, aiv_yz = {0, 0}, no_xy = {0, 0}, no_z_ = {0, 0};
    float tminf = 0, tmaxf = 0, tmin_hi = 0;
    // wave-uniform item cursor
    uint32_t q_cur = 0, q_end = 0;
    bool exhausted = false;
    uint32_t d_node = 0, d_core = 0, d_iter = 0, d_refill = 0, d_cand = 0, d_far = 0, d_over = 0, d_deep = 0, d_rays = 0, d_miss = 0;  // DIAG

    for (;;) {
        const int walking = __popcll(__ballot(cur >= 0));
        if (DIAG && lane == 0) d_iter++;
        if (walking < K.min_lanes) {
            if (DIAG && lane == 0) d_refill++;
            // ------------------------------------------------------------ refill the idle lanes
            if (!exhausted) {
                const uint64_t need = __ballot(cur < 0);
                if (q_cur >= q_end) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(A.cursor, F.claim);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= n_items) {
                        exhausted = true;
                    } else {
                        q_cur = base;
                        q_end = (n_items - base < F.claim) ? n_items : base + F.claim;
                    }
                }
                const uint32_t avail = q_end - q_cur;
                const uint32_t rank = lane_rank(need);
                const bool take = cur < 0 && rank < avail;
                const uint32_t nneed = (uint32_t)__popcll(need);
                const uint32_t item = q_cur + rank;
                q_cur += nneed < avail ? nneed : avail;
                if (take && Q.job[item] != PT_HOLE) {
                    e = item;
                    cnt = 0;
                    sp = 0;
                    // ---- FP64, transient: the ray, its clip against the scene cube, the planes' certain bounds
                    const RayD r{Q.d[e], Q.d[qc + e], Q.d[2 * qc + e], Q.d[3 * qc + e], Q.d[4 * qc + e], Q.d[5 * qc + e]};
                    const double a_ = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
                    const Clip clip = clip_ray(F, r, tmin);
                    const bool tame = (a_ >= 1e-100) && (a_ <= 1e100) && (ptm::f_abs(r.ox) <= 1e100) && (ptm::f_abs(r.oy) <= 1e100) &&
                                      (ptm::f_abs(r.oz) <= 1e100);
                    if (!tame || clip.far || !bvh_ray_trusted(F, r, clip, a_)) {
                        // not a ray the FP32 bounds were analysed for: the FP64 traversal answers it
                        K.slow_list[atomicAdd(K.slow_count, 1u)] = e;
                        K.cand_n[e] = PT_CAND_EXACT;
                        if (DIAG) {
                            d_far++;
                            if (clip.far && !clip.miss) {
                                atomicAdd(K.diag + 10, 1ull);  // far AND through the scene cube: the fat rays of the FP64 traversal
                                atomicMax(K.diag + 11, (unsigned long long)(clip.infl / F.margin));
                            }
                        }
                    } else if (root < 0 || clip.miss) {
                        if (DIAG) d_miss++;
                        K.cand_n[e] = 0;  // no finite object can be hit: the planes are the exact pass's business
                    } else {
                        const double ts = clip.ts;
                        const float fox = (float)(r.ox + r.dx * ts), foy = (float)(r.oy + r.dy * ts), foz = (float)(r.oz + r.dz * ts);
                        const float ivx = __builtin_amdgcn_rcpf((float)r.dx), ivy = __builtin_amdgcn_rcpf((float)r.dy),
                                    ivz = __builtin_amdgcn_rcpf((float)r.dz);
                        iv_xy = v2f{ivx, ivy};
                        iv_z_aiv_x = v2f{ivz, __builtin_fabsf(ivx)};
                        aiv_yz = v2f{__builtin_fabsf(ivy), __builtin_fabsf(ivz)};
                        // -o/d: the extra rounding is far inside the margin, and inf - inf = NaN for a zero direction component
                        // leaves that slab unconstrained (see scan_broad_narrow)
                        no_xy = v2f{-fox * ivx, -foy * ivy};
                        no_z_ = v2f{-foz * ivz, 0.0f};
                        // node parameters are relative to the re-based origin: t' = t - ts (see scan_bvh)
                        const float tl = (float)(tmin - ts);
                        tminf = __builtin_fmaxf(tl - (__builtin_fabsf(tl) * 1e-2f + 1e-6f), 0.0f);  // a little below tMin - ts
                        tmin_hi = __builtin_fmaxf(tl, 0.0f) * 1.01f + 1e-6f;                        // a little above
                        float tmx = __builtin_inff();
                        if (MODE == 0 && F.planes_y) {
                            // a plane with the normal (0, 1, 0) is certainly hit at t = (py - oy) / dy when |dy| is clear of the
                            // 1e-6 rejection (objects.go:101-103) and t is clear of tMin: an upper bound of the winner's t
                            for (int k = 0; k < F.n_plane; k++) {
                                const auto &o = g_obj[g_pl[k]];
                                const double tp = (o.a[1] - r.oy) * __builtin_amdgcn_rcp(r.dy);  // relative error < 2^-26
                                if (ptm::f_abs(r.dy) >= 1.001e-6 && tp > tmin * 1.001) {
                                    float tpf = (float)(tp - ts);
                                    tpf += __builtin_fabsf(tpf) * 1e-6f + 1e-6f;
                                    tmx = __builtin_fminf(tmx, tpf);
                                }
                            }
                        }
                        tmaxf = tmx;
                        cur = root;
                        if (DIAG) d_rays++;
                    }
                }
            }
            if (__ballot(cur >= 0) == 0) {
                if (exhausted) break;
                continue;  // holes, misses and hand-overs only: claim again
            }
        }
        if (cur >= 0) {
            // ---------------------------------------------------------------- one visit
            const bool core_visit = (cur & PT_WALK_CORE) != 0;
            const int node = cur & (PT_WALK_CORE - 1);
            if (DIAG) { if (core_visit) d_core++; else d_node++; }
            const NodeQ nd = load_node(core_visit ? &cores[node] : &nodes[node]);
            const int nbase = nd.nbase, obase = nd.obase;
            const uint32_t meta = nd.meta;
            // slab parameters of the four slots, two per packed instruction: tc = c*iv - o*iv, tn = tc - h*|iv|, tf = tc + h*|iv|
            // (the third operand of the tn / tf instructions is the whole pair tc: lo for the lo result, hi for the hi result)
            v2f nxA, nxB, fxA, fxB, nyA, nyB, fyA, fyB, nzA, nzB, fzA, fzB;
#define PT_W32_AXIS(AX, IVPAIR, IVSEL, NOPAIR, NOSEL, AIVPAIR, AIVSEL, nA, nB, fA, fB)                                   \
    {                                                                                                                  \
        v2f tcA, tcB;                                                                                                  \
        PT_PK_FMA(tcA, nd.cA[AX], IVPAIR, NOPAIR, IVSEL, NOSEL, NOSEL, 0);                                             \
        PT_PK_FMA(tcB, nd.cB[AX], IVPAIR, NOPAIR, IVSEL, NOSEL, NOSEL, 0);                                             \
        PT_PK_FMA(nA, nd.hA[AX], AIVPAIR, tcA, AIVSEL, 0, 1, 1);                                                       \
        PT_PK_FMA(nB, nd.hB[AX], AIVPAIR, tcB, AIVSEL, 0, 1, 1);                                                       \
        PT_PK_FMA(fA, nd.hA[AX], AIVPAIR, tcA, AIVSEL, 0, 1, 0);                                                       \
        PT_PK_FMA(fB, nd.hB[AX], AIVPAIR, tcB, AIVSEL, 0, 1, 0);                                                       \
    }
            PT_W32_AXIS(0, iv_xy, 0, no_xy, 0, iv_z_aiv_x, 1, nxA, nxB, fxA, fxB)
            PT_W32_AXIS(1, iv_xy, 1, no_xy, 1, aiv_yz, 0, nyA, nyB, fyA, fyB)
            PT_W32_AXIS(2, iv_z_aiv_x, 0, no_z_, 0, aiv_yz, 1, nzA, nzB, fzA, fzB)
#undef PT_W32_AXIS
            float t0[4], t1[4];
            t0[0] = pt_max3(nxA.x, nyA.x, nzA.x); t0[1] = pt_max3(nxA.y, nyA.y, nzA.y);
            t0[2] = pt_max3(nxB.x, nyB.x, nzB.x); t0[3] = pt_max3(nxB.y, nyB.y, nzB.y);
            t1[0] = pt_min3(fxA.x, fyA.x, fzA.x); t1[1] = pt_min3(fxA.y, fyA.y, fzA.y);
            t1[2] = pt_min3(fxB.x, fyB.x, fzB.x); t1[3] = pt_min3(fxB.y, fyB.y, fzB.y);
            uint32_t hb = 0, outside = 0;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                outside |= (t0[s] > tmin_hi) ? (1u << s) : 0u;  // the ray starts before the slot's box (NaN: no)
                t0[s] = pt_max3(t0[s], tminf, tminf);
                hb |= (pt_min3(t1[s], tmaxf, tmaxf) < t0[s]) ? 0u : (1u << s);
            }
            if (core_visit) {
                // a pierced core, left after tMin: the object is hit no later than where the ray enters the core
                float th = __builtin_inff();
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const bool hit = ((hb & core_ok & (meta >> 12)) >> s) & 1u;
                    const float en = __builtin_fmaxf(t0[s], tmin_hi);
                    th = (hit && t1[s] > tmin_hi) ? __builtin_fminf(th, en) : th;
                }
                th += __builtin_fabsf(th) * 2e-6f;
                tmaxf = __builtin_fminf(tmaxf, th);
                hb = 0;  // a twin has no children
            }
            // ---- object slots whose box is pierced: candidates
            uint32_t oh = core_visit ? 0u : (hb & (meta >> 12) & 0xfu);
            const uint32_t want_core = MODE == 0 ? (oh & ((meta >> 16) | outside) & (meta >> 20)) : 0u;  // meta 16-19: the slot's object is a box, 20-23: it has a core
            while (oh != 0) {
                const uint32_t s = (uint32_t)__builtin_ctz(oh);
                oh &= oh - 1;
                if (cnt < PT_CAND_MAX) K.cand_ids[(size_t)e * PT_CAND_MAX + cnt] = (uint32_t)(obase + (int)((meta >> (2u * s)) & 3u));
                cnt++;
                if (DIAG) d_cand++;
            }
            // ---- internal children nearest first (the sorting network of scan_bvh)
            uint32_t k0, k1, k2, k3;
            {
                uint32_t key[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const uint32_t bits = (__float_as_uint(t0[s]) & ~7u) | (uint32_t)(2 * s);
                    key[s] = ((hb & (meta >> 8)) & (1u << s)) ? bits : 0xffffffffu;  // pierced AND an internal node
                }
                uint32_t a0 = key[0] < key[1] ? key[0] : key[1], a1 = key[0] < key[1] ? key[1] : key[0];
                uint32_t a2 = key[2] < key[3] ? key[2] : key[3], a3 = key[2] < key[3] ? key[3] : key[2];
                k0 = a0 < a2 ? a0 : a2;
                const uint32_t m0 = a0 < a2 ? a2 : a0;
                k3 = a1 < a3 ? a3 : a1;
                const uint32_t m1 = a1 < a3 ? a1 : a3;
                k1 = m0 < m1 ? m0 : m1;
                k2 = m0 < m1 ? m1 : m0;
            }
#define PT_CHILD(k) (nbase + (int)((meta >> ((k) & 7u)) & 3u))
            bool deep = false;
            if (want_core != 0) {  // the core twin first (it may cull what follows); every child waits on the stack
                const uint32_t tmp = k0; k0 = 0xffffffffu;
                if (tmp != 0xffffffffu) {
                    if (sp + 4 > PT_WALK_STACK) deep = true;
                    else {
                        if (k3 != 0xffffffffu) { stack[sp * PT_BLOCK] = PT_CHILD(k3); sp++; }
                        if (k2 != 0xffffffffu) { stack[sp * PT_BLOCK] = PT_CHILD(k2); sp++; }
                        if (k1 != 0xffffffffu) { stack[sp * PT_BLOCK] = PT_CHILD(k1); sp++; }
                        stack[sp * PT_BLOCK] = PT_CHILD(tmp); sp++;
                    }
                }
                core_ok = want_core;
                cur = node | PT_WALK_CORE;
            } else {
                if (k1 != 0xffffffffu) {  // sorted: k2 and k3 can only be candidates when k1 is
                    if (sp + 3 > PT_WALK_STACK) {
                        deep = true;  // not enough stack left in LDS: the FP64 traversal (with its full stack) takes the ray
                    } else {
                        if (k2 != 0xffffffffu) {
                            if (k3 != 0xffffffffu) { stack[sp * PT_BLOCK] = PT_CHILD(k3); sp++; }
                            stack[sp * PT_BLOCK] = PT_CHILD(k2);
                            sp++;
                        }
                        stack[sp * PT_BLOCK] = PT_CHILD(k1);
                        sp++;
                    }
                }
                if (k0 != 0xffffffffu) {
                    cur = PT_CHILD(k0);
                } else if (sp > 0) {
                    sp--;
                    cur = stack[sp * PT_BLOCK];
                } else {
                    cur = -1;
                }
            }
#undef PT_CHILD
            if (DIAG && (deep || cnt > PT_CAND_MAX)) { if (deep) d_deep++; else d_over++; }
            if (deep || cnt > PT_CAND_MAX) {  // more than the list holds, or too deep: hand the ray over
                K.slow_list[atomicAdd(K.slow_count, 1u)] = e;
                K.cand_n[e] = PT_CAND_EXACT;
                cur = -1;
            } else if (cur < 0) {
                K.cand_n[e] = cnt;  // walk complete
            }
        }
    }
    if (DIAG) {
        const uint32_t v[10] = {d_node, d_core, d_iter, d_refill, d_cand, d_far, d_over, d_deep, d_rays, d_miss};
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const uint32_t w = wave_sum(v[i]);
            if (lane == 0 && w) atomicAdd(K.diag + i, (unsigned long long)w);
        }
    }
}

// (A stack entry may have been pushed before tmaxf shrank: it is visited anyway and its slots are culled there.  Keeping the
// entry parameters on the stack to drop such entries when popped was measured in round 2 and lost.)

// The exact FP64 tests of one path's candidates: planes, then the listed objects; `best` / `tmax` as scan_bvh leaves them.
template <int MODE, typename ObjPtr, typename IdxPtr>
__device__ __forceinline__ void exact_candidates(const DevFrame &F, ObjPtr g_obj, IdxPtr g_pl, const BvhObj *__restrict__ bobjs,
                                                 const uint32_t *__restrict__ ids, uint32_t n, const RayD &r, int &best, double &tmax) {
    const double tmin = MODE ? 0.0001 : 0.001;
    tmax = ptm::max_float64();
    best = -1;
    bool best_is_box = false;
    for (int k = 0; k < F.n_plane; k++) {
        const int i = g_pl[k];
        const auto &o = g_obj[i];
        if (MODE != 0 && !(o.kind & 0x100)) continue;
        double t = 0;
        if (F.planes_y ? plane_exact_y(o.a[1], r, tmin, tmax, t)
                       : plane_exact(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, tmin, tmax, t)) {
            if (MODE == 0 ? wins(0, false, i, t, best, best_is_box, tmax)
                          : (wins(1, false, i, t, best, best_is_box, tmax) && exit_candidate_ok(o, KIND_PLANE, r, t))) {
                best = i;
                tmax = t;
                best_is_box = false;
            }
        }
    }
    if (__ballot(n != 0) == 0) return;
    const double a = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
    const double ya = div_recip(a);
    const double ivx = 1 / r.dx, ivy = 1 / r.dy, ivz = 1 / r.dz;
    for (uint32_t k = 0; __ballot(k < n) != 0; k++) {
        if (k < n) {
            const BvhObj &bo = bobjs[ids[k]];
            const int kind = bo.o.kind & 0xff;
            const int i = bo.index;
            double t = 0;
            bool valid;
            const bool is_box = kind == KIND_BOX;
            if (is_box)
                valid = box_exact<true>(bo.o.a[0], bo.o.a[1], bo.o.a[2], bo.o.b[0], bo.o.b[1], bo.o.b[2], r, ivx, ivy, ivz, tmin,
                                        ptm::max_float64(), t);
            else
                valid = sphere_exact_shared(bo.o.a[0], bo.o.a[1], bo.o.a[2], bo.o.radius_sq, r, a, ya, tmin, tmax, t);
            if (MODE != 0 && !(bo.o.kind & 0x100)) valid = false;  // only glass can end an exit search (renderer.go:333)
            if (valid && wins(MODE, is_box, i, t, best, best_is_box, tmax) && (MODE == 0 || exit_candidate_ok(bo.o, kind, r, t))) {
                best = i;
                tmax = t;
                best_is_box = is_box;
            }
        }
    }
}

// Runs the exact pass for queue entry i: MODE 0 closest hit, MODE 1 exit search.  VERIFY: the reference's own loop over every
// object as well; disagreements are counted (counters[4]) and the loop's answer is used.
template <int MODE, bool VERIFY, typename ObjPtr, typename IdxPtr>
__device__ __forceinline__ void exact_answer(const Walk32Args &K, ObjPtr g_obj, IdxPtr g_pl, uint32_t i, const RayD &r, int &best, double &tmax,
                                             uint32_t &c_mismatch) {
    const WfArgs &A = K.W;
    const PathQueue &Q = A.qin;
    const uint32_t n = K.cand_n[i];
    if (n == PT_CAND_EXACT) {  // answered by the FP64 traversal
        best = Q.hit[i];
        tmax = Q.d[9 * (size_t)Q.cap + i];
    }
    // (lanes answered already sit the tests out with n = 0: the planes are part of the FP64 traversal's answer)
    int b2 = -1;
    double t2 = 0;
    exact_candidates<MODE>(A.F, g_obj, g_pl, A.B.bvh_objs, K.cand_ids + (size_t)i * PT_CAND_MAX, n == PT_CAND_EXACT ? 0u : n, r, b2, t2);
    if (n != PT_CAND_EXACT) {
        best = b2;
        tmax = t2;
    }
    if (VERIFY) {
        int b3;
        double t3;
        scan_uniform(A.F, g_obj, r, MODE, b3, t3);
        if (best != b3 || (best >= 0 && !(tmax == t3))) {
            c_mismatch++;
            unsigned long long *dbg = A.B.counters + 8;
            dbg[0] = ((unsigned long long)(uint32_t)best << 32) | (uint32_t)b3;
            dbg[1] = ptm::to_bits(tmax);
            dbg[2] = ptm::to_bits(t3);
            dbg[3] = (unsigned long long)MODE | ((unsigned long long)n << 8);
            dbg[4] = ptm::to_bits(r.ox); dbg[5] = ptm::to_bits(r.oy); dbg[6] = ptm::to_bits(r.oz);
            dbg[7] = ptm::to_bits(r.dx); dbg[8] = ptm::to_bits(r.dy); dbg[9] = ptm::to_bits(r.dz);
        }
        best = b3;
        tmax = t3;
    }
}

// Exact tests + shading of every path in qin after its walk (wf_shade_kernel with the exact pass in front).
template <bool STATS, bool VERIFY>
__global__ __launch_bounds__(PT_BLOCK) void wf_shade32_kernel(const Walk32Args K) {
    extern __shared__ __align__(16) unsigned char smem[];
    const WfArgs &A = K.W;
    const DevFrame &F = A.F;
    const TraceBuffers &B = A.B;
    const PathQueue &Q = A.qin;
    DevMat *lds_mat = reinterpret_cast<DevMat *>(smem);
    {
        const uint64_t *g1 = reinterpret_cast<const uint64_t *>(B.mats);
        uint64_t *l1 = reinterpret_cast<uint64_t *>(lds_mat);
        const int n1 = F.nmat * (int)(sizeof(DevMat) / 8);
        for (int i = threadIdx.x; i < n1; i += PT_BLOCK) l1[i] = g1[i];
        __syncthreads();
    }
    typedef const DevObj __attribute__((address_space(4))) *ConstObjPtr;
    typedef const int32_t __attribute__((address_space(4))) *ConstIdxPtr;
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const ConstObjPtr g_obj = (ConstObjPtr)(B.objs);
    const ConstIdxPtr g_pl = (ConstIdxPtr)(B.plane_idx);
    const uint32_t lane = threadIdx.x & (PT_WAVE - 1);
    const uint32_t n_count = *(ConstU32Ptr)(Q.count);
    const uint32_t n = n_count < Q.cap ? n_count : Q.cap;
    const size_t qc = Q.cap;
    uint32_t c_seg = 0, c_draw = 0, c_exit = 0, c_mismatch = 0;
    QueueWindow w_out, w_exit;

    for (uint32_t i0 = blockIdx.x * PT_BLOCK + (threadIdx.x & ~(PT_WAVE - 1u)); i0 < n; i0 += gridDim.x * PT_BLOCK) {
        const uint32_t i = i0 + lane;
        bool go_on = false, to_exit = false;
        double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, Tx = 0, Ty = 0, Tz = 0;
        uint64_t rs = 0;
        uint32_t job = PT_HOLE, j_seg = 0, j_draw = 0;
        int depth = 0, exit_mat = 0;
        if (i < n) job = Q.job[i];
        if (job != PT_HOLE) {
            ox = Q.d[i]; oy = Q.d[qc + i]; oz = Q.d[2 * qc + i];
            dx = Q.d[3 * qc + i]; dy = Q.d[4 * qc + i]; dz = Q.d[5 * qc + i];
            int best = -1;
            double tmax = 0;
            exact_answer<0, VERIFY>(K, g_obj, g_pl, i, RayD{ox, oy, oz, dx, dy, dz}, best, tmax, c_mismatch);
            Tx = Q.d[6 * qc + i]; Ty = Q.d[7 * qc + i]; Tz = Q.d[8 * qc + i];
            rs = Q.rs[i];
            depth = Q.depth[i];
            if (STATS) { j_seg = Q.jseg[i]; j_draw = Q.jdraw[i]; }
            c_seg++;
            if (STATS) j_seg++;
            bool finished = false;
            double termx = 0, termy = 0, termz = 0, attx = 1, atty = 1, attz = 1;
            if (best < 0) {
                // sky closure, renderer.go:56-92
                finished = true;
                const DevSky &sky = A.sky;
                if (sky.kind == 1) {
                    const double dirLen = ptm::f_sqrt(dx * dx + dy * dy + dz * dz);
                    if (dirLen == 0) {
                        termx = sky.c0[0]; termy = sky.c0[1]; termz = sky.c0[2];
                    } else {
                        double tt = (dy / dirLen + 1.0) * 0.5;
                        if (tt < 0) tt = 0;
                        if (tt > 1) tt = 1;
                        termx = sky.c0[0] * (1 - tt) + sky.c1[0] * tt;
                        termy = sky.c0[1] * (1 - tt) + sky.c1[1] * tt;
                        termz = sky.c0[2] * (1 - tt) + sky.c1[2] * tt;
                    }
                } else {
                    termx = sky.c0[0]; termy = sky.c0[1]; termz = sky.c0[2];
                }
            } else {
                bool exit_search = false;
                shade_hit<STATS, true>(B.objs[best], lds_mat, tmax, ox, oy, oz, dx, dy, dz, rs, c_draw, j_draw, finished, termx, termy, termz, attx,
                                       atty, attz, exit_search, exit_mat);
                if (exit_search) {
                    to_exit = true;
                    c_exit++;
                } else if (!finished) {
                    finished = roulette_advance<STATS>(depth, attx, atty, attz, Tx, Ty, Tz, rs, c_draw, j_draw);
                    go_on = !finished;
                }
            }
            if (finished) {
                ptk::store_radiance(B.L, job, Tx * termx, Ty * termy, Tz * termz);
                if (STATS) { B.job_seg[job] = j_seg; B.job_draw[job] = j_draw; }
            }
        }
        const uint32_t s_out = window_push(w_out, A.qout.count, go_on, lane, PT_CONT_BLOCK);
        if (go_on) queue_store(A.qout, s_out, ox, oy, oz, dx, dy, dz, Tx, Ty, Tz, rs, job, depth, -1, j_seg, j_draw, STATS, B.counters + 19);
        const uint32_t s_ex = window_push(w_exit, A.qexit.count, to_exit, lane, PT_QUEUE_BLOCK);
        if (to_exit) queue_store(A.qexit, s_ex, ox, oy, oz, dx, dy, dz, Tx, Ty, Tz, rs, job, depth, exit_mat, j_seg, j_draw, STATS, B.counters + 19);
    }
    window_close(w_out, A.qout, lane);
    window_close(w_exit, A.qexit, lane);
    const uint32_t w_seg = wave_sum(c_seg), w_draw = wave_sum(c_draw), w_ex = wave_sum(c_exit);
    if (lane == 0) {
        if (w_seg) atomicAdd(&B.counters[0], (unsigned long long)w_seg);
        if (w_ex) atomicAdd(&B.counters[1], (unsigned long long)w_ex);
        if (w_draw) atomicAdd(&B.counters[2], (unsigned long long)w_draw);
    }
    if (VERIFY) {
        const uint32_t w_mis = wave_sum(c_mismatch);
        if (lane == 0 && w_mis) atomicAdd(&B.counters[4], (unsigned long long)w_mis);
    }
}

// Exact exit searches + their epilogue (wf_exit_kernel with the exact pass in front).
template <bool STATS, bool VERIFY>
__global__ __launch_bounds__(PT_BLOCK) void wf_exit32_kernel(const Walk32Args K) {
    extern __shared__ __align__(16) unsigned char smem[];
    const WfArgs &A = K.W;
    const DevFrame &F = A.F;
    const TraceBuffers &B = A.B;
    const PathQueue &Q = A.qin;
    DevMat *lds_mat = reinterpret_cast<DevMat *>(smem);
    {
        const uint64_t *g1 = reinterpret_cast<const uint64_t *>(B.mats);
        uint64_t *l1 = reinterpret_cast<uint64_t *>(lds_mat);
        const int n1 = F.nmat * (int)(sizeof(DevMat) / 8);
        for (int i = threadIdx.x; i < n1; i += PT_BLOCK) l1[i] = g1[i];
        __syncthreads();
    }
    typedef const DevObj __attribute__((address_space(4))) *ConstObjPtr;
    typedef const int32_t __attribute__((address_space(4))) *ConstIdxPtr;
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const ConstObjPtr g_obj = (ConstObjPtr)(B.objs);
    const ConstIdxPtr g_pl = (ConstIdxPtr)(B.plane_idx);
    const uint32_t lane = threadIdx.x & (PT_WAVE - 1);
    const uint32_t n_count = *(ConstU32Ptr)(Q.count);
    const uint32_t n = n_count < Q.cap ? n_count : Q.cap;
    const size_t qc = Q.cap;
    uint32_t c_draw = 0, c_mismatch = 0;
    QueueWindow w_out;
    for (uint32_t i0 = blockIdx.x * PT_BLOCK + (threadIdx.x & ~(PT_WAVE - 1u)); i0 < n; i0 += gridDim.x * PT_BLOCK) {
        const uint32_t i = i0 + lane;
        bool go_on = false;
        double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, Tx = 0, Ty = 0, Tz = 0;
        uint64_t rs = 0;
        uint32_t job = PT_HOLE, j_seg = 0, j_draw = 0;
        int depth = 0;
        if (i < n) job = Q.job[i];
        if (job != PT_HOLE) {
            ox = Q.d[i]; oy = Q.d[qc + i]; oz = Q.d[2 * qc + i];
            dx = Q.d[3 * qc + i]; dy = Q.d[4 * qc + i]; dz = Q.d[5 * qc + i];
            int ebest = -1;
            double tmax = 0;
            exact_answer<1, VERIFY>(K, g_obj, g_pl, i, RayD{ox, oy, oz, dx, dy, dz}, ebest, tmax, c_mismatch);
            Tx = Q.d[6 * qc + i]; Ty = Q.d[7 * qc + i]; Tz = Q.d[8 * qc + i];
            rs = Q.rs[i];
            depth = Q.depth[i];
            const int exit_mat = Q.best[i];
            if (STATS) { j_seg = Q.jseg[i]; j_draw = Q.jdraw[i]; }
            double attx = 1, atty = 1, attz = 1;
            exit_post(lds_mat[exit_mat], ebest, tmax, ox, oy, oz, dx, dy, dz, attx, atty, attz);
            const bool finished = roulette_advance<STATS>(depth, attx, atty, attz, Tx, Ty, Tz, rs, c_draw, j_draw);
            if (finished) {
                ptk::store_radiance(B.L, job, Tx * 0.0, Ty * 0.0, Tz * 0.0);
                if (STATS) { B.job_seg[job] = j_seg; B.job_draw[job] = j_draw; }
            } else {
                go_on = true;
            }
        }
        const uint32_t s_out = window_push(w_out, A.qout.count, go_on, lane, PT_CONT_BLOCK);
        if (go_on) queue_store(A.qout, s_out, ox, oy, oz, dx, dy, dz, Tx, Ty, Tz, rs, job, depth, -1, j_seg, j_draw, STATS, B.counters + 19);
    }
    window_close(w_out, A.qout, lane);
    const uint32_t w_draw = wave_sum(c_draw);
    if (lane == 0 && w_draw) atomicAdd(&B.counters[2], (unsigned long long)w_draw);
    if (VERIFY) {
        const uint32_t w_mis = wave_sum(c_mismatch);
        if (lane == 0 && w_mis) atomicAdd(&B.counters[4], (unsigned long long)w_mis);
    }
}

}  // namespace ptk
