// pt_wavefront.h -- the wavefront form of the render loop (included by ptcore.hip after pt_kernels.h).
//
// Every path of a chunk lives in a path-state queue in HBM (PathQueue, pt_device.h) and the loop of
// rayColorOpt (renderer.go:286-404) is cut into passes over those queues, one level of the recursion at a time:
//
//   wf_init_kernel       fresh jobs (primary rays of raygen_kernel) -> queue A
//   per level:
//     wf_traverse_kernel<0>  closest hit of every path in A        (renderer.go:297-302)
//     wf_shade_kernel        sky / emitted / scatter / roulette       (renderer.go:303-319, :375-403): paths that go on
//                            are appended to queue B, dielectric front-face hits to the exit queue E
//     wf_traverse_kernel<1>  exit search of every path in E        (renderer.go:321-349)
//     wf_exit_kernel         Beer-Lambert + origin move + roulette    (renderer.go:352-403): survivors -> B
//     (optional) wf_bin_*    B is reordered by direction octant and cell of the origin, so that a wave's 64 rays walk
//                            the hierarchy together
//     A <-> B
//
// Why: in the all-in-one loop a wave does one scan per trip for all its lanes and then shades; in a scene with 10^5
// objects the traversals of one wave differ by an order of magnitude in length, and the wave waits for its longest
// (20 % of lanes busy, profiles/r01_n3_summary.json).  Here a lane that has finished its traversal writes the hit
// and takes the next ray of the queue at once (ballot + mbcnt refill, as for jobs in trace_kernel), so the walk loop
// always runs on full waves, and shading runs as its own pass at one path per lane.
// The price is HBM traffic: 48 B in + 12 B out per traversal, 100 B in + up to 100 B out per shaded path.
#pragma once

#include "pt_kernels.h"

namespace ptk {

struct WfArgs {
    DevFrame F;
    DevSky sky;
    TraceBuffers B;
    PathQueue qin;    // paths of this level
    PathQueue qout;   // paths that go on to the next level
    PathQueue qexit;  // exit searches of this level
    unsigned int *cursor;  // item cursor of the running traversal pass
    // ray sorting (optional): the traversal pass takes its rays in the order of `perm`, which lists the entries of qin by
    // direction octant and cell of the origin (counting sort: wf_bin_count / wf_bin_scan / wf_bin_scatter)
    const uint32_t *perm;      // [n_sorted] entry indices, or null: queue order
    const uint32_t *n_sorted;  // number of entries in perm (holes are left out)
    uint32_t *bin_count;       // [PT_WF_BINS + 1] histogram, then exclusive offsets
    uint32_t *bin_key;         // [cap] key of every entry (PT_HOLE for a hole)
};

#define PT_WF_GRID 8u                                        // cells per axis over the scene cube
#define PT_WF_BINS (8u * PT_WF_GRID * PT_WF_GRID * PT_WF_GRID)  // direction octant x cell = 4096

// 3-bit Morton interleave of a cell coordinate
__device__ __forceinline__ uint32_t wf_spread3(uint32_t v) {
    v &= 7u;
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4);
}
// direction octant (3 bits, high) and Morton cell of the origin inside the scene cube (9 bits): rays of one bin start
// in the same eighth of each axis and head the same way, so they meet the same upper nodes of the hierarchy
__device__ __forceinline__ uint32_t wf_ray_key(const DevFrame &F, double ox, double oy, double oz, double dx, double dy, double dz) {
    const float inv = (float)(0.5 * PT_WF_GRID) / (float)F.scene_bound;
    const float half = 0.5f * PT_WF_GRID;
    const float fx = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf((float)ox, inv, half), 0.0f), (float)(PT_WF_GRID - 1u));
    const float fy = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf((float)oy, inv, half), 0.0f), (float)(PT_WF_GRID - 1u));
    const float fz = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf((float)oz, inv, half), 0.0f), (float)(PT_WF_GRID - 1u));
    const uint32_t cell = wf_spread3((uint32_t)fx) | (wf_spread3((uint32_t)fy) << 1) | (wf_spread3((uint32_t)fz) << 2);
    const uint32_t oct = (dx < 0 ? 1u : 0u) | (dy < 0 ? 2u : 0u) | (dz < 0 ? 4u : 0u);
    return (oct << 9) | cell;
}

// Counting sort of the entries of A.qin by key.  Every block owns one contiguous slice of the queue in passes 1 and 3
// and keeps its histogram in LDS, so the global counters see one atomic per block and bin instead of one per ray
// (a single address serves ~10^8 returning atomics a second on this chip: a popular bin would take longer than the
// traversal it is meant to speed up).
__device__ __forceinline__ void wf_slice(uint32_t n, uint32_t &lo, uint32_t &hi) {
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    lo = blockIdx.x * per < n ? blockIdx.x * per : n;
    hi = lo + per < n ? lo + per : n;
}
// pass 1: key of every entry (PT_HOLE for a hole) and the histogram of the keys
__global__ __launch_bounds__(PT_BLOCK) void wf_bin_count_kernel(const WfArgs A) {
    __shared__ uint32_t hist[PT_WF_BINS];
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const PathQueue &Q = A.qin;
    const uint32_t n_count = *(ConstU32Ptr)(Q.count);
    const uint32_t n = n_count < Q.cap ? n_count : Q.cap;
    const size_t qc = Q.cap;
    for (uint32_t k = threadIdx.x; k < PT_WF_BINS; k += PT_BLOCK) hist[k] = 0;
    __syncthreads();
    uint32_t lo, hi;
    wf_slice(n, lo, hi);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += PT_BLOCK) {
        uint32_t key = PT_HOLE;
        if (Q.job[i] != PT_HOLE) {
            key = wf_ray_key(A.F, Q.d[i], Q.d[qc + i], Q.d[2 * qc + i], Q.d[3 * qc + i], Q.d[4 * qc + i], Q.d[5 * qc + i]);
            atomicAdd(&hist[key], 1u);
        }
        A.bin_key[i] = key;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < PT_WF_BINS; k += PT_BLOCK)
        if (hist[k]) atomicAdd(&A.bin_count[k], hist[k]);
}
// pass 2: exclusive prefix sum of the histogram (one block), total -> bin_count[PT_WF_BINS]
__global__ __launch_bounds__(1024) void wf_bin_scan_kernel(const WfArgs A) {
    __shared__ uint32_t part[1024];
    constexpr uint32_t PER = PT_WF_BINS / 1024u;
    uint32_t loc[PER];
    uint32_t s = 0;
    for (uint32_t k = 0; k < PER; k++) {
        loc[k] = s;
        s += A.bin_count[threadIdx.x * PER + k];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {  // Hillis-Steele over the 1024 partial sums
        const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t base = part[threadIdx.x] - s;
    for (uint32_t k = 0; k < PER; k++) A.bin_count[threadIdx.x * PER + k] = base + loc[k];
    if (threadIdx.x == 1023u) A.bin_count[PT_WF_BINS] = part[1023];
}
// pass 3: the block reserves room for its slice in every bin (one atomic per bin), its entries take the places
__global__ __launch_bounds__(PT_BLOCK) void wf_bin_scatter_kernel(const WfArgs A, uint32_t *perm) {
    __shared__ uint32_t hist[PT_WF_BINS];
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const uint32_t n_count = *(ConstU32Ptr)(A.qin.count);
    const uint32_t n = n_count < A.qin.cap ? n_count : A.qin.cap;
    for (uint32_t k = threadIdx.x; k < PT_WF_BINS; k += PT_BLOCK) hist[k] = 0;
    __syncthreads();
    uint32_t lo, hi;
    wf_slice(n, lo, hi);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += PT_BLOCK) {
        const uint32_t key = A.bin_key[i];
        if (key != PT_HOLE) atomicAdd(&hist[key], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < PT_WF_BINS; k += PT_BLOCK)
        if (hist[k]) hist[k] = atomicAdd(&A.bin_count[k], hist[k]);  // from here on: next free place of the bin for this block
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += PT_BLOCK) {
        const uint32_t key = A.bin_key[i];
        if (key != PT_HOLE) perm[atomicAdd(&hist[key], 1u)] = i;
    }
}

// One wave's window into a queue it appends to (block reservation: see trace_kernel's glass queue).
struct QueueWindow {
    uint32_t cur = 0, end = 0;
};
// Slots for the lanes with `push` set; lanes without get an unspecified value.
__device__ __forceinline__ uint32_t window_push(QueueWindow &w, uint32_t *count, bool push, uint32_t lane, uint32_t block) {
    const uint64_t pm = __ballot(push);
    if (pm == 0) return 0;
    const uint32_t np = (uint32_t)__popcll(pm), room = w.end - w.cur;
    uint32_t nbase = 0;
    if (np > room) {
        if (lane == 0) nbase = atomicAdd(count, block);
        nbase = __builtin_amdgcn_readfirstlane(nbase);
    }
    const uint32_t rank = lane_rank(pm);
    const uint32_t slot = rank < room ? w.cur + rank : nbase + (rank - room);
    if (np > room) {
        w.cur = nbase + (np - room);
        w.end = nbase + block;
    } else {
        w.cur += np;
    }
    return slot;
}
__device__ __forceinline__ void window_close(const QueueWindow &w, const PathQueue &q, uint32_t lane) {
    for (uint32_t s = w.cur + lane; s < w.end && s < q.cap; s += PT_WAVE) q.job[s] = PT_HOLE;
}

__device__ __forceinline__ void queue_store(const PathQueue &q, uint32_t slot, double ox, double oy, double oz, double dx, double dy,
                                            double dz, double Tx, double Ty, double Tz, uint64_t rs, uint32_t job, int depth, int best,
                                            uint32_t j_seg, uint32_t j_draw, bool stats, unsigned long long *overflow) {
    const size_t qc = q.cap;
    if (slot >= q.cap) {  // cannot happen (the host sizes the queues for every path plus every window); never write outside,
        atomicAdd(overflow, 1ull);  // and make the frame fail instead of losing a path quietly
        return;
    }
    q.d[slot] = ox; q.d[qc + slot] = oy; q.d[2 * qc + slot] = oz;
    q.d[3 * qc + slot] = dx; q.d[4 * qc + slot] = dy; q.d[5 * qc + slot] = dz;
    q.d[6 * qc + slot] = Tx; q.d[7 * qc + slot] = Ty; q.d[8 * qc + slot] = Tz;
    q.rs[slot] = rs;
    q.job[slot] = job;
    q.depth[slot] = depth;
    q.best[slot] = best;
    if (stats) { q.jseg[slot] = j_seg; q.jdraw[slot] = j_draw; }
}

// Fresh jobs -> queue entries (entry i = job i; a job whose pixel lies outside the frame is a hole).
template <bool STATS>
__global__ __launch_bounds__(PT_BLOCK) void wf_init_kernel(const WfArgs A) {
    const DevFrame &F = A.F;
    const TraceBuffers &B = A.B;
    uint32_t c_samples = 0, c_draw = 0;
    // grid-stride: the two counters below get one atomic per wave of a few thousand, not one per 64 jobs
    for (uint32_t i = blockIdx.x * PT_BLOCK + threadIdx.x; i < F.njobs; i += gridDim.x * PT_BLOCK) {
        const uint32_t nd = B.ray_ndraw[i];
        uint32_t job = PT_HOLE;
        if (nd != 0xffffu) {
            c_samples++;
            c_draw += nd;
            if (F.max_depth <= 0) {  // rayColorOpt returns black before any scan (renderer.go:287-289)
                ptk::store_radiance(B.L, i, 0.0, 0.0, 0.0);
                if (STATS) { B.job_seg[i] = 0; B.job_draw[i] = nd; }
            } else {
                job = i;
                const size_t nj = F.njobs;
                queue_store(A.qin, i, B.ray[i], B.ray[nj + i], B.ray[2 * nj + i], B.ray[3 * nj + i], B.ray[4 * nj + i], B.ray[5 * nj + i], 1.0,
                            1.0, 1.0, B.ray_rng[i], i, F.max_depth, -1, 0u, nd, STATS, B.counters + 19);
            }
        }
        if (job == PT_HOLE && i < A.qin.cap) A.qin.job[i] = PT_HOLE;
    }
    const uint32_t w_s = wave_sum(c_samples), w_d = wave_sum(c_draw);
    if ((threadIdx.x & (PT_WAVE - 1)) == 0) {
        if (w_d) atomicAdd(&B.counters[2], (unsigned long long)w_d);
        if (w_s) atomicAdd(&B.counters[3], (unsigned long long)w_s);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *A.qin.count = F.njobs;
}

// Closest hit (MODE 0) or exit search (MODE 1) of every path in A.qin: best and tmax are written into the entry.
// Persistent waves; a lane that has its answer takes the next entry at once, so the walk loop of scan_bvh always
// starts on a full wave (it is left, as in trace_kernel, once fewer than F.bvh_min_lanes lanes are still walking).
template <int MODE, bool VERIFY>
__global__ __launch_bounds__(PT_BLOCK, PT_BVH_WAVES) void wf_traverse_kernel(const WfArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const DevFrame &F = A.F;
    const TraceBuffers &B = A.B;
    const PathQueue &Q = A.qin;
    int *lds_stack = reinterpret_cast<int *>(smem);
    BvhNode *lds_nodes = reinterpret_cast<BvhNode *>(smem + (size_t)F.bvh_stack * PT_BLOCK * sizeof(int));
    {
        const uint64_t *gsrc = reinterpret_cast<const uint64_t *>(B.bvh_nodes);
        uint64_t *ldst = reinterpret_cast<uint64_t *>(lds_nodes);
        const int nw = F.bvh_lds_nodes * (int)(sizeof(BvhNode) / 8);
        for (int i = threadIdx.x; i < nw; i += PT_BLOCK) ldst[i] = gsrc[i];
        __syncthreads();
    }
    typedef const DevObj __attribute__((address_space(4))) *ConstObjPtr;
    typedef const int32_t __attribute__((address_space(4))) *ConstIdxPtr;
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const ConstObjPtr g_obj = (ConstObjPtr)(B.objs);
    const ConstIdxPtr g_pl = (ConstIdxPtr)(B.plane_idx);
    const uint32_t lane = threadIdx.x & (PT_WAVE - 1);
    const uint32_t n_count = *(ConstU32Ptr)(Q.count);
    const uint32_t n_items = A.perm ? *(ConstU32Ptr)(A.n_sorted) : (n_count < Q.cap ? n_count : Q.cap);
    const size_t qc = Q.cap;

    bool have = false;
    uint32_t idx = 0;
    double ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    TravState trav;
    uint32_t cur = 0, end = 0;
    bool exhausted = false;
    uint32_t c_mismatch = 0;
    const ProfHooks ph{nullptr, nullptr, nullptr, lane, nullptr};

    for (;;) {
        const uint64_t need = __ballot(!have);
        if (need != 0) {
            if (cur >= end && !exhausted) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(A.cursor, F.claim);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= n_items) {
                    exhausted = true;
                } else {
                    cur = base;
                    end = (n_items - base < F.claim) ? n_items : base + F.claim;
                }
            }
            const uint32_t avail = end - cur;
            const uint32_t rank = lane_rank(need);
            const bool take = !have && rank < avail;
            const uint32_t nneed = (uint32_t)__popcll(need);
            const uint32_t item = cur + rank;
            cur += nneed < avail ? nneed : avail;
            if (take) {
                const uint32_t e = A.perm ? A.perm[item] : item;  // sorted order lists no holes
                if (A.perm || Q.job[e] != PT_HOLE) {
                    idx = e;
                    ox = Q.d[e]; oy = Q.d[qc + e]; oz = Q.d[2 * qc + e];
                    dx = Q.d[3 * qc + e]; dy = Q.d[4 * qc + e]; dz = Q.d[5 * qc + e];
                    have = true;
                    trav = TravState();
                }
            }
            if (__ballot(have) == 0) {
                if (exhausted) break;
                continue;
            }
        }
        int best = -1;
        double tmax = 0;
        bool scanned = true;
        bool fat = false;  // a far-away ray with bounds as wide as the scene: the whole wave scans the world for it (see trace_kernel)
        if (have) {
            const RayD ray{ox, oy, oz, dx, dy, dz};
            const double a_ = dx * dx + dy * dy + dz * dz;
            const Clip clip = clip_ray(F, ray, MODE ? 0.0001 : 0.001);
            const bool tame = (a_ >= 1e-100) && (a_ <= 1e100) && (ptm::f_abs(ox) <= 1e100) && (ptm::f_abs(oy) <= 1e100) &&
                              (ptm::f_abs(oz) <= 1e100);
            if (__ballot(!tame) != 0) {  // rays with non-finite or absurd components: the reference's own loop
                scan_uniform(F, g_obj, ray, MODE, best, tmax);
                trav.live = false;
            } else {
                if ((fat = !trav.live && clip.far && !clip.miss && clip.infl * 16.0 > F.scene_bound))
                    scanned = false;
                else if (__ballot(clip.far || !bvh_ray_trusted(F, ray, clip, a_)) != 0)
                    scanned = scan_bvh<false, true>(F, g_obj, g_pl, B.bvh_nodes, lds_nodes, B.bvh_objs, lds_stack + threadIdx.x, ray, clip, MODE,
                                                    trav, best, tmax, ph);
                else
                    scanned = scan_bvh<false, false>(F, g_obj, g_pl, B.bvh_nodes, lds_nodes, B.bvh_objs, lds_stack + threadIdx.x, ray, clip, MODE,
                                                     trav, best, tmax, ph);
                if (VERIFY && scanned) {
                    int best2;
                    double tmax2;
                    scan_uniform(F, g_obj, ray, MODE, best2, tmax2);
                    if (best != best2 || (best >= 0 && !(tmax == tmax2))) {
                        c_mismatch++;
                        unsigned long long *dbg = B.counters + 8;
                        dbg[0] = ((unsigned long long)(uint32_t)best << 32) | (uint32_t)best2;
                        dbg[1] = ptm::to_bits(tmax);
                        dbg[2] = ptm::to_bits(tmax2);
                        dbg[3] = (unsigned long long)MODE;
                        dbg[4] = ptm::to_bits(ox); dbg[5] = ptm::to_bits(oy); dbg[6] = ptm::to_bits(oz);
                        dbg[7] = ptm::to_bits(dx); dbg[8] = ptm::to_bits(dy); dbg[9] = ptm::to_bits(dz);
                    }
                    best = best2;
                    tmax = tmax2;
                }
            }
        }
        {
            uint64_t fm = __ballot(fat);
            while (fm != 0) {
                const int src = __ffsll((long long)fm) - 1;
                fm &= fm - 1;
                const LinearHit fh = scan_linear_wave(F.nobj, B.objs, __shfl(ox, src, 64), __shfl(oy, src, 64), __shfl(oz, src, 64), __shfl(dx, src, 64),
                                                      __shfl(dy, src, 64), __shfl(dz, src, 64), MODE, lane);
                if ((int)lane == src) {
                    best = fh.best;
                    tmax = fh.tmax;
                    scanned = true;
                    if (VERIFY) {  // the verify instantiation checks this path like the walks above
                        int best2;
                        double tmax2;
                        scan_uniform(F, g_obj, RayD{ox, oy, oz, dx, dy, dz}, MODE, best2, tmax2);
                        if (best != best2 || (best >= 0 && !(tmax == tmax2))) c_mismatch++;
                        best = best2;
                        tmax = tmax2;
                    }
                }
                if (lane == 0) atomicAdd(B.counters + 23, 1ull);
            }
        }
        if (have && scanned) {
            Q.hit[idx] = best;  // (`best` of an exit-queue entry keeps the glass material)
            Q.d[9 * qc + idx] = tmax;
            have = false;
        }
    }
    if (VERIFY) {
        const uint32_t w_mis = wave_sum(c_mismatch);
        if (lane == 0 && w_mis) atomicAdd(&B.counters[4], (unsigned long long)w_mis);
    }
}

// The flat scans as a pass: one path per lane (bitmask scan over <= 32 + 32 records; the wavefront form of the
// reference-sized scenes, kept as the A/B of the all-in-one loop).
template <int MODE, bool VERIFY>
__global__ __launch_bounds__(PT_BLOCK) void wf_scan_flat_kernel(const WfArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const DevFrame &F = A.F;
    const TraceBuffers &B = A.B;
    const PathQueue &Q = A.qin;
    DevObj *lds_obj = reinterpret_cast<DevObj *>(smem);
    int *lds_kidx = reinterpret_cast<int *>(smem + (size_t)F.nobj * sizeof(DevObj));
    {
        const uint64_t *g0 = reinterpret_cast<const uint64_t *>(B.objs);
        uint64_t *l0 = reinterpret_cast<uint64_t *>(lds_obj);
        const int n0 = F.nobj * (int)(sizeof(DevObj) / 8);
        for (int i = threadIdx.x; i < n0; i += PT_BLOCK) l0[i] = g0[i];
        for (int i = threadIdx.x; i < F.n_bsph; i += PT_BLOCK) lds_kidx[pt_record_slot(i, F.n_bsph)] = B.bsph[i].index;
        for (int i = threadIdx.x; i < F.n_bbox; i += PT_BLOCK) lds_kidx[F.n_bsph + pt_record_slot(i, F.n_bbox)] = B.bbox[i].index;
        __syncthreads();
    }
    typedef const DevObj __attribute__((address_space(4))) *ConstObjPtr;
    typedef const BroadSphere __attribute__((address_space(4))) *ConstSphPtr;
    typedef const BroadBox __attribute__((address_space(4))) *ConstBoxPtr;
    typedef const int32_t __attribute__((address_space(4))) *ConstIdxPtr;
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const ConstObjPtr g_obj = (ConstObjPtr)(B.objs);
    const ConstIdxPtr g_pl = (ConstIdxPtr)(B.plane_idx);
    const BroadLists<ConstSphPtr, ConstBoxPtr> BL{(ConstSphPtr)B.bsph, (ConstBoxPtr)B.bbox, F.n_bsph, F.n_bbox, F.sph_all, F.box_all,
                                                  F.sph_diel, F.box_diel, lds_kidx, lds_kidx + F.n_bsph};
    const uint32_t lane = threadIdx.x & (PT_WAVE - 1);
    const uint32_t n_count = *(ConstU32Ptr)(Q.count);
    const uint32_t n = n_count < Q.cap ? n_count : Q.cap;
    const size_t qc = Q.cap;
    uint32_t c_mismatch = 0;
    const ProfHooks ph{nullptr, nullptr, nullptr, lane, nullptr};
    for (uint32_t i0 = blockIdx.x * PT_BLOCK + (threadIdx.x & ~(PT_WAVE - 1u)); i0 < n; i0 += gridDim.x * PT_BLOCK) {
        const uint32_t i = i0 + lane;
        if (i < n && Q.job[i] != PT_HOLE) {
            const RayD ray{Q.d[i], Q.d[qc + i], Q.d[2 * qc + i], Q.d[3 * qc + i], Q.d[4 * qc + i], Q.d[5 * qc + i]};
            int best = -1;
            double tmax = 0;
            const double a_ = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
            const Clip clip = clip_ray(F, ray, MODE ? 0.0001 : 0.001);
            const bool tame = (a_ >= 1e-100) && (a_ <= 1e100) && (ptm::f_abs(ray.ox) <= 1e100) && (ptm::f_abs(ray.oy) <= 1e100) &&
                              (ptm::f_abs(ray.oz) <= 1e100) && !clip.far;
            if (__ballot(!tame) != 0) {
                scan_uniform(F, g_obj, ray, MODE, best, tmax);
            } else {
                scan_broad_narrow<false, VERIFY, MODE>(F, g_obj, BL, g_pl, lds_obj, ray, clip, MODE, best, tmax, ph);
                if (VERIFY) {
                    int best2;
                    double tmax2;
                    scan_uniform(F, g_obj, ray, MODE, best2, tmax2);
                    if (best != best2 || (best >= 0 && !(tmax == tmax2))) c_mismatch++;
                    best = best2;
                    tmax = tmax2;
                }
            }
            Q.hit[i] = best;
            Q.d[9 * qc + i] = tmax;
        }
    }
    if (VERIFY) {
        const uint32_t w_mis = wave_sum(c_mismatch);
        if (lane == 0 && w_mis) atomicAdd(&B.counters[4], (unsigned long long)w_mis);
    }
}

// Shading of every path in A.qin after its closest-hit pass.
template <bool STATS>
__global__ __launch_bounds__(PT_BLOCK) void wf_shade_kernel(const WfArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const DevFrame &F = A.F;
    const TraceBuffers &B = A.B;
    const PathQueue &Q = A.qin;
    // small scenes: world and materials in LDS; BVH scenes: materials only (objects come from HBM / L2)
    const bool world_in_lds = F.world_in_lds != 0;
    DevObj *lds_obj = reinterpret_cast<DevObj *>(smem);
    DevMat *lds_mat = reinterpret_cast<DevMat *>(smem + (world_in_lds ? (size_t)F.nobj * sizeof(DevObj) : 0));
    {
        if (world_in_lds) {
            const uint64_t *g0 = reinterpret_cast<const uint64_t *>(B.objs);
            uint64_t *l0 = reinterpret_cast<uint64_t *>(lds_obj);
            const int n0 = F.nobj * (int)(sizeof(DevObj) / 8);
            for (int i = threadIdx.x; i < n0; i += PT_BLOCK) l0[i] = g0[i];
        }
        const uint64_t *g1 = reinterpret_cast<const uint64_t *>(B.mats);
        uint64_t *l1 = reinterpret_cast<uint64_t *>(lds_mat);
        const int n1 = F.nmat * (int)(sizeof(DevMat) / 8);
        for (int i = threadIdx.x; i < n1; i += PT_BLOCK) l1[i] = g1[i];
        __syncthreads();
    }
    const DevObj *const s_obj = world_in_lds ? lds_obj : B.objs;
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const uint32_t lane = threadIdx.x & (PT_WAVE - 1);
    const uint32_t n_count = *(ConstU32Ptr)(Q.count);
    const uint32_t n = n_count < Q.cap ? n_count : Q.cap;
    const size_t qc = Q.cap;
    uint32_t c_seg = 0, c_draw = 0, c_exit = 0;
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
            Tx = Q.d[6 * qc + i]; Ty = Q.d[7 * qc + i]; Tz = Q.d[8 * qc + i];
            const double tmax = Q.d[9 * qc + i];
            rs = Q.rs[i];
            depth = Q.depth[i];
            const int best = Q.hit[i];
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
                shade_hit<STATS, true>(s_obj[best], lds_mat, tmax, ox, oy, oz, dx, dy, dz, rs, c_draw, j_draw, finished, termx, termy, termz, attx,
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
}

// After the exit searches of a level (renderer.go:352-403): every entry of A.qin (the exit queue; `best` = the glass
// material, `hit` / tmax = the answer of wf_traverse_kernel<1>) gets its attenuation, its origin moved to the exit
// point and its roulette; survivors are appended to A.qout.
template <bool STATS>
__global__ __launch_bounds__(PT_BLOCK) void wf_exit_kernel(const WfArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
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
    typedef const uint32_t __attribute__((address_space(4))) *ConstU32Ptr;
    const uint32_t lane = threadIdx.x & (PT_WAVE - 1);
    const uint32_t n_count = *(ConstU32Ptr)(Q.count);
    const uint32_t n = n_count < Q.cap ? n_count : Q.cap;
    const size_t qc = Q.cap;
    uint32_t c_draw = 0;
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
            Tx = Q.d[6 * qc + i]; Ty = Q.d[7 * qc + i]; Tz = Q.d[8 * qc + i];
            const double tmax = Q.d[9 * qc + i];
            rs = Q.rs[i];
            depth = Q.depth[i];
            const int exit_mat = Q.best[i];
            const int ebest = Q.hit[i];
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
}

}  // namespace ptk
