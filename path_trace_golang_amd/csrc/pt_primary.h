// pt_primary.h -- the first segment of every path of a BVH scene, traversed by the WAVE instead of by the lane (round 4).
//
// A trace wave of the BVH path walks 64 unrelated rays, one stack per lane: divergent node fetches (six 16-byte requests per
// lane and visit), a fifth to a half of the lanes waiting for the slowest walks of the wave.  The primary rays of
// a pass are the one population for which none of that is necessary: 64 consecutive jobs are one sample of one 8 x 8 pixel
// block (pt_device.h), i.e. 64 rays from (nearly) one point through (nearly) one direction.  primary_bvh_kernel gives each
// such wave ONE stack:
//   * a node is fetched once per wave by scalar loads (wave-uniform address, no vector memory request at all) and its four
//     slot boxes feed the lanes' FP32 slab tests as scalar operands;
//   * the wave descends into a child when ANY lane's ray pierces it before that lane's own tmax (ballot), nearest first
//     by the entry parameter of the first lane that pierces it, the others pushed on the wave's stack (LDS, 96 words);
//   * an object slot pierced by any lane is tested at once: the 96-byte record arrives by scalar loads, the kind is
//     wave-uniform (no divergence between sphere and box code), the lanes whose own slab test passed run the reference's
//     exact FP64 test (objects.go:37-61, :141-179) and take the winner by `wins` (order-free statement of the loop,
//     renderer.go:297-302), each culling against its own tmax from then on.
// Every lane therefore tests at least the objects its own per-lane walk would have tested (those whose inflated box its ray
// pierces before its current tmax) -- the result is the sequential loop's winner, bit for bit; `verify_bvh` checks it.
//
// The kernel then shades the hit (renderer.go:304-403: sky, emitted, scatter, roulette) with the code the trace loop uses
// and appends the paths that go on to the continuation queue (the split passes' PathQueue, windows of PT_CONT_BLOCK slots per
// atomic), from which trace_kernel<.., SCAN_BVH, ..> takes them like fresh jobs; paths that end write their radiance record.
// Whatever this kernel is not meant for goes to the queue UNSHADED, as the primary ray itself at full depth, and trace_kernel
// does what it always did: a wave that holds a ray with non-finite or absurd components, one that starts outside 3.5 scene
// sizes (clip / far-origin logic of clip_ray) or outside the range the FP32 bounds were analysed for; and dielectric hits
// (their exit search walks the second tree; one object in ten of the synthetic scenes).
#pragma once

#include "pt_kernels.h"

namespace ptk {

__device__ __forceinline__ void sky_radiance(const DevSky &sky, double dx, double dy, double dz, double &termx, double &termy, double &termz) {
    // sky closure, renderer.go:56-92 (the expressions of trace_kernel's sky branch)
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
}

#ifndef PT_PRIMARY_WAVES
#define PT_PRIMARY_WAVES 6  // blocks of 256 threads per CU the kernel is compiled for (84 VGPRs)
#endif

template <bool STATS, bool VERIFY>
__global__ __launch_bounds__(PT_BLOCK, PT_PRIMARY_WAVES) void primary_bvh_kernel(const TraceArgs A) {
    __shared__ int wstack[PT_BLOCK / PT_WAVE][PT_BVH_STACK];
    const DevFrame &F = A.F;
    const TraceBuffers &B = A.B;
    typedef const BvhNode __attribute__((address_space(4))) *ConstNodePtr;
    typedef const BvhObj __attribute__((address_space(4))) *ConstBObjPtr;
    typedef const DevObj __attribute__((address_space(4))) *ConstObjPtr;
    typedef const int32_t __attribute__((address_space(4))) *ConstIdxPtr;
    const ConstNodePtr nodes = (ConstNodePtr)B.bvh_nodes;
    const ConstBObjPtr bobjs = (ConstBObjPtr)B.bvh_objs;
    const ConstObjPtr g_obj = (ConstObjPtr)B.objs;
    const ConstIdxPtr g_pl = (ConstIdxPtr)B.plane_idx;
    const uint32_t lane = threadIdx.x & (PT_WAVE - 1);
    int *const wst = wstack[threadIdx.x >> 6];
    const uint32_t wave0 = (blockIdx.x * PT_BLOCK + threadIdx.x) >> 6, nwaves = (gridDim.x * PT_BLOCK) >> 6;
    const size_t nj = F.njobs, qc = B.cont.cap;
    uint32_t q_cur = 0, q_end = 0;  // this wave's window of continuation slots (one atomic per PT_CONT_BLOCK slots, see trace_kernel)
    uint32_t c_seg = 0, c_draw = 0, c_samples = 0, c_mismatch = 0, c_cont = 0;
    uint32_t c_visits = 0, c_coop = 0, c_odd = 0;  // wave-uniform (diagnostics)
    const double tmin = 0.001;  // renderer.go:292

    for (uint32_t wj = wave0; (size_t)wj * PT_WAVE < nj; wj += nwaves) {  // njobs is a multiple of 64: a wave's jobs all exist
        const uint32_t job = wj * PT_WAVE + lane;
        const uint32_t nd = B.ray_ndraw[job];
        const bool have = nd != 0xffffu;  // 0xffff: the job's pixel lies outside the frame (edge tile)
        double ox = B.ray[job], oy = B.ray[nj + job], oz = B.ray[2 * nj + job];
        double dx = B.ray[3 * nj + job], dy = B.ray[4 * nj + job], dz = B.ray[5 * nj + job];
        uint64_t rs = B.ray_rng[job];
        uint32_t j_seg = 0, j_draw = nd;
        if (have) { c_samples++; c_draw += nd; }
        if (F.max_depth <= 0) {  // rayColorOpt returns black before any scan (renderer.go:287-289); the camera draws happened
            if (have) {
                store_radiance(B.L, job, 0.0, 0.0, 0.0);
                if (STATS) { B.job_seg[job] = 0; B.job_draw[job] = nd; }
            }
            continue;
        }
        // ------------------------------------------------------------ is this a wave for the shared walk?
        const RayD r{ox, oy, oz, dx, dy, dz};
        const double a = dx * dx + dy * dy + dz * dz;
        const double Cb = F.clip_bound;
        const bool tame = (a >= 1e-100) && (a <= 1e100) && (ptm::f_abs(ox) <= Cb) && (ptm::f_abs(oy) <= Cb) && (ptm::f_abs(oz) <= Cb);
        const Clip noclip{0.0, 0.0, 0.0, false, false};  // origins within 3.5 scene sizes are scanned from where they are (clip_ray: `inside`)
        const bool odd = have && !(tame && bvh_ray_trusted(F, r, noclip, a));
        const bool coop = __ballot(odd) == 0;
        if (coop) c_coop++;
        else c_odd++;
        int best = -1;
        double tmax = ptm::max_float64();
        if (coop && have) {
            bool best_is_box = false;
            // ---- planes: infinite, always tested exactly (scan_bvh does the same)
            for (int k = 0; k < F.n_plane; k++) {
                const int i = g_pl[k];
                const auto &o = g_obj[i];
                double t = 0;
                if (F.planes_y ? plane_exact_y(o.a[1], r, tmin, tmax, t)
                               : plane_exact(o.a[0], o.a[1], o.a[2], o.b[0], o.b[1], o.b[2], r, tmin, tmax, t)) {
                    if (wins(0, false, i, t, best, best_is_box, tmax)) {
                        best = i;
                        tmax = t;
                        best_is_box = false;
                    }
                }
            }
            if (F.bvh_root >= 0) {
                const double ya = div_recip(a);
                const double ivx = 1 / dx, ivy = 1 / dy, ivz = 1 / dz;  // objects.go:149,154,159
                // FP32 quantities of the node tests, as scan_bvh derives them (ts = 0: the ray starts inside the clip bound)
                const float fox = (float)ox, foy = (float)oy, foz = (float)oz;
                const float fdx = (float)dx, fdy = (float)dy, fdz = (float)dz;
                float tminf = (float)tmin;
                tminf -= __builtin_fabsf(tminf) * 1e-2f + 1e-6f;
                tminf = __builtin_fmaxf(tminf, 0.0f);
                const float ivxf = __builtin_amdgcn_rcpf(fdx), ivyf = __builtin_amdgcn_rcpf(fdy), ivzf = __builtin_amdgcn_rcpf(fdz);
                const float aivxf = __builtin_fabsf(ivxf), aivyf = __builtin_fabsf(ivyf), aivzf = __builtin_fabsf(ivzf);
                const float noxf = -fox * ivxf, noyf = -foy * ivyf, nozf = -foz * ivzf;
                float tmaxf = (float)tmax;
                tmaxf += __builtin_fabsf(tmaxf) * 4.8e-7f;  // >= tmax (MaxFloat64 becomes +inf)
                int sp = 0;
                int cur = F.bvh_root;
                while (cur >= 0) {
                    c_visits++;
                    const auto &nd_ = nodes[cur];  // wave-uniform: scalar loads
                    const uint32_t meta = nd_.meta;
                    const int nbase = nd_.node_base, obase = nd_.obj_base;
                    uint32_t key[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const bool is_node = (meta >> (8 + s)) & 1u, is_obj = (meta >> (12 + s)) & 1u;
                        if (!(is_node || is_obj)) continue;  // empty slot (wave-uniform)
                        // slab parameters from centre and half extent (see PT_BOX_SLABS); the half extents carry the embedded index
                        // bytes of the per-lane walk in their low mantissa bits: still upper bounds
                        const float tcx = __builtin_fmaf(nd_.c[0][s], ivxf, noxf), tcy = __builtin_fmaf(nd_.c[1][s], ivyf, noyf),
                                    tcz = __builtin_fmaf(nd_.c[2][s], ivzf, nozf);
                        const float t0 = pt_vmax3(__builtin_fmaf(-nd_.h[0][s], aivxf, tcx), __builtin_fmaf(-nd_.h[1][s], aivyf, tcy),
                                                  pt_vmax(__builtin_fmaf(-nd_.h[2][s], aivzf, tcz), tminf));
                        const float t1 = pt_vmin3(__builtin_fmaf(nd_.h[0][s], aivxf, tcx), __builtin_fmaf(nd_.h[1][s], aivyf, tcy),
                                                  pt_vmin(__builtin_fmaf(nd_.h[2][s], aivzf, tcz), tmaxf));
                        const bool pierced = !(t1 < t0);  // NaN slabs constrain nothing, like the per-lane walk
                        const uint64_t pm = __ballot(pierced);
                        if (pm == 0) continue;
                        if (is_node) {
                            // entry parameter of the first lane that pierces the slot (>= 0: its bits order like the value), slot in the low bits
                            const int src = __ffsll((long long)pm) - 1;
                            const uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(t0), src);
                            key[s] = (bits & ~3u) | (uint32_t)s;
                        } else if (pierced) {
                            // the slot's object, for every lane whose own slab test passed: the reference's exact test
                            const auto &bo = bobjs[obase + (int)((meta >> (2 * s)) & 3u)];
                            const int i = bo.index;
                            const bool is_box = (bo.o.kind & 0xff) == KIND_BOX;  // wave-uniform
                            double t = 0;
                            bool valid;
                            if (is_box)
                                valid = box_exact<true>(bo.o.a[0], bo.o.a[1], bo.o.a[2], bo.o.b[0], bo.o.b[1], bo.o.b[2], r, ivx, ivy, ivz, tmin,
                                                        ptm::max_float64(), t);
                            else
                                valid = sphere_exact_shared<false>(bo.o.a[0], bo.o.a[1], bo.o.a[2], bo.o.radius_sq, r, a, ya, tmin, tmax, t);
                            if (valid && wins(0, is_box, i, t, best, best_is_box, tmax)) {
                                best = i;
                                tmax = t;
                                best_is_box = is_box;
                                tmaxf = (float)tmax;
                                tmaxf += __builtin_fabsf(tmaxf) * 4.8e-7f;
                            }
                        }
                    }
                    // internal children nearest first (wave-uniform keys: scalar unit); 0xffffffff = not a candidate
                    uint32_t k0, k1, k2, k3;
                    {
                        const uint32_t a0 = key[0] < key[1] ? key[0] : key[1], a1 = key[0] < key[1] ? key[1] : key[0];
                        const uint32_t a2 = key[2] < key[3] ? key[2] : key[3], a3 = key[2] < key[3] ? key[3] : key[2];
                        k0 = a0 < a2 ? a0 : a2;
                        const uint32_t m0 = a0 < a2 ? a2 : a0;
                        k3 = a1 < a3 ? a3 : a1;
                        const uint32_t m1 = a1 < a3 ? a1 : a3;
                        k1 = m0 < m1 ? m0 : m1;
                        k2 = m0 < m1 ? m1 : m0;
                    }
#define PT_CHILD(k) (nbase + (int)((meta >> (2u * ((k) & 3u))) & 3u))
                    if (k1 != 0xffffffffu) {
                        if (k2 != 0xffffffffu) {
                            if (k3 != 0xffffffffu) { wst[sp] = PT_CHILD(k3); sp++; }
                            wst[sp] = PT_CHILD(k2);
                            sp++;
                        }
                        wst[sp] = PT_CHILD(k1);
                        sp++;
                    }
                    if (k0 != 0xffffffffu) {
                        cur = PT_CHILD(k0);
                    } else if (sp > 0) {
                        sp--;
                        cur = __builtin_amdgcn_readfirstlane(wst[sp]);
                    } else {
                        cur = -1;
                    }
#undef PT_CHILD
                }
            }
            if (VERIFY) {
                int best2;
                double tmax2;
                scan_uniform(F, g_obj, r, 0, best2, tmax2);
                if (best != best2 || (best >= 0 && !(tmax == tmax2))) c_mismatch++;
                best = best2;
                tmax = tmax2;
            }
        }
        // ------------------------------------------------------------ shade (renderer.go:304-403), or hand the ray over unshaded
        bool go_on = false;
        int depth = F.max_depth;
        double Tx = 1, Ty = 1, Tz = 1;
        if (have) {
            const bool hit_glass = coop && best >= 0 && (B.objs[best].kind & 0x100);
            if (!coop || hit_glass) {
                go_on = true;  // the primary ray itself, at full depth: trace_kernel scans and shades it
            } else {
                c_seg++;
                j_seg = 1;
                bool finished = false;
                double termx = 0, termy = 0, termz = 0;
                if (best < 0) {
                    finished = true;
                    sky_radiance(A.sky, dx, dy, dz, termx, termy, termz);
                } else {
                    double attx = 1, atty = 1, attz = 1;
                    bool exit_search = false;
                    int exit_mat = 0;
                    uint32_t jd = j_draw;
                    shade_hit<true, false>(B.objs[best], B.mats, tmax, ox, oy, oz, dx, dy, dz, rs, c_draw, jd, finished, termx, termy, termz, attx,
                                           atty, attz, exit_search, exit_mat);
                    if (!finished) finished = roulette_advance<true>(depth, attx, atty, attz, Tx, Ty, Tz, rs, c_draw, jd);
                    j_draw = jd;
                }
                if (finished) {
                    store_radiance(B.L, job, Tx * termx, Ty * termy, Tz * termz);
                    if (STATS) { B.job_seg[job] = j_seg; B.job_draw[job] = j_draw; }
                } else {
                    go_on = true;
                }
            }
        }
        // ------------------------------------------------------------ survivors -> continuation queue (as glass_kernel does)
        const uint64_t pm = __ballot(go_on);
        if (pm != 0) {
            const uint32_t np = (uint32_t)__popcll(pm), room = q_end - q_cur;
            uint32_t nbase = 0;
            if (np > room) {
                if (lane == 0) nbase = atomicAdd(B.cont.count, (uint32_t)PT_CONT_BLOCK);
                nbase = __builtin_amdgcn_readfirstlane(nbase);
            }
            const uint32_t rank = lane_rank(pm);
            const uint32_t slot = rank < room ? q_cur + rank : nbase + (rank - room);
            if (np > room) {
                q_cur = nbase + (np - room);
                q_end = nbase + PT_CONT_BLOCK;
            } else {
                q_cur += np;
            }
            if (go_on && slot >= B.cont.cap) {
                atomicAdd(B.counters + 19, 1ull);  // cannot happen (the queue holds every job plus every window); never write outside it
            } else if (go_on) {
                B.cont.d[slot] = ox;
                B.cont.d[qc + slot] = oy;
                B.cont.d[2 * qc + slot] = oz;
                B.cont.d[3 * qc + slot] = dx;
                B.cont.d[4 * qc + slot] = dy;
                B.cont.d[5 * qc + slot] = dz;
                B.cont.d[6 * qc + slot] = Tx;
                B.cont.d[7 * qc + slot] = Ty;
                B.cont.d[8 * qc + slot] = Tz;
                B.cont.rs[slot] = rs;
                B.cont.job[slot] = job;
                B.cont.depth[slot] = depth;
                if (STATS) { B.cont.jseg[slot] = j_seg; B.cont.jdraw[slot] = j_draw; }
                c_cont++;
            }
        }
    }
    for (uint32_t s = q_cur + lane; s < q_end && s < B.cont.cap; s += PT_WAVE) B.cont.job[s] = PT_HOLE;  // the rest of the window stays empty
    const uint32_t w_seg = wave_sum(c_seg), w_draw = wave_sum(c_draw), w_samples = wave_sum(c_samples), w_cont = wave_sum(c_cont);
    if (lane == 0) {
        if (w_seg) atomicAdd(&B.counters[0], (unsigned long long)w_seg);
        if (w_draw) atomicAdd(&B.counters[2], (unsigned long long)w_draw);
        if (w_samples) atomicAdd(&B.counters[3], (unsigned long long)w_samples);
        if (w_cont) atomicAdd(&B.counters[6], (unsigned long long)w_cont);
        if (c_visits) atomicAdd(&B.counters[40], (unsigned long long)c_visits);  // wave-level node visits (diagnostics)
        if (c_coop) atomicAdd(&B.counters[41], (unsigned long long)c_coop);      // blocks of 64 jobs walked by the wave
        if (c_odd) atomicAdd(&B.counters[42], (unsigned long long)c_odd);        // ... handed over unshaded
    }
    if (VERIFY) {
        const uint32_t w_mis = wave_sum(c_mismatch);
        if (lane == 0 && w_mis) atomicAdd(&B.counters[4], (unsigned long long)w_mis);
    }
}

}  // namespace ptk
