// pt_device.h -- device-side scene layout shared by the host set-up code and the
// gfx950 kernels of libptcore.
//
// HBM layout (all per context, per device):
//   DevObj  objs[nobj]     80 B each   world after sceneToWorld (objects.go:225-269), file order
//   DevMat  mats[nmat+1]  104 B each   materials after convertMaterial (materials.go:28-55);
//                                      slot nmat is the zero material of a missing id
//   double  L[3][njobs]                per-job sample radiance of the current spp chunk (SoA)
//   double  acc[3][nslots]             per-pixel-slot running sum over samples (SoA)
//   uint8   tiles[ntiles][32][32][4]   resolved RGBA8, tile-major
// A "slot" is one pixel position inside an owned 32x32 tile (renderer.go:132): tile lt,
// 8x8 sub-block sb (row-major 4x4 inside the tile), pixel p (row-major 8x8):
//   slot = (lt*16 + sb)*64 + p.
// A "job" is one (slot, sample) pair of the current chunk of S samples:
//   job  = ((lt*16 + sb)*S + (s - s0))*64 + p
// so 64 consecutive jobs are one sample of one 8x8 block (coherent primary rays,
// coalesced L traffic) and a wave that claims a job range stays inside one block.
#pragma once

#include <stdint.h>

namespace ptd {

enum { KIND_SPHERE = 0, KIND_PLANE = 1, KIND_BOX = 2 };
enum { MAT_LAMBERT = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_EMISSIVE = 3, MAT_MIRROR = 4 };

struct alignas(16) DevObj {
    double a[3];       // sphere centre | plane point | box min
    double b[3];       // -             | plane normal | box max
    double radius;     // sphere
    double radius_sq;  // radius*radius (objects.go:46), same product computed once on the host
    double inv_radius; // 1.0/radius (objects.go:70)
    int32_t kind;      // KIND_* in the low byte; bit 8 set when the material is dielectric
    int32_t mat;       // index into mats[]
};
static_assert(sizeof(DevObj) == 80, "DevObj layout");

struct alignas(8) DevMat {
    double albedo[3];
    double rough;
    double ior;
    double emit[3];
    double absorption[3];
    int32_t typ;
    int32_t absorbs;   // absorption.x > 0 || .y > 0 || .z > 0 (renderer.go:360)
    double rough_sq;   // rough*rough (materials.go:121)
    // dielectric only, the same IEEE operations the reference performs on every hit, done once on the host:
    double inv_ior;    // 1.0 / ior                                (materials.go:183, front face)
    double r0_front;   // ((1 - 1/ior) / (1 + 1/ior))^2            (materials.go:227-228 with ratio = 1/ior)
    double r0_back;    // ((1 - ior) / (1 + ior))^2                (ratio = ior)
};
static_assert(sizeof(DevMat) == 128, "DevMat layout");

// Broad-phase records (FP32, read with scalar loads): conservative bounds of the finite
// objects, inflated by `m = 2^-12 * scene bound` and rounded outward, so that an FP32 test
// can only ever keep too many candidates, never drop one the FP64 test would accept.
struct alignas(16) BroadSphere {
    float cx, cy, cz;
    float rm2;         // (radius + m)^2, rounded up
    int32_t index;     // object index = bit in the candidate mask
    int32_t diel;      // 1: the object is dielectric (grouped scan: the exit-search mask is built on the fly)
    int32_t pad[2];
};
static_assert(sizeof(BroadSphere) == 32, "BroadSphere layout");
struct alignas(16) BroadBox {
    float c[3];        // centre of [min - m, max + m]
    float h[3];        // half extent, rounded up so that [c - h, c + h] holds [min - m, max + m]
    int32_t index;
    int32_t diel;      // 1: the object is dielectric
};
static_assert(sizeof(BroadBox) == 32, "BroadBox layout");

// BVH node for scenes beyond the candidate bitmasks: four slots, each an internal node, one object or
// empty, with the slot's FP32 box (inflated, rounded outward; as centre and half extent) stored slot-minor so one
// 16-byte load brings the same number of all four.  Nodes are numbered breadth-first: the internal children of a
// node are node_base + rank, its object children bvh_objs[obj_base + rank].
//   meta bits 0-7: rank of slot s within its kind at bits [2s, 2s+2); 8-11: slot is an internal node; 12-15: slot is an
//   object; 16-19: the slot's object is a box (else a sphere); 20-23: the slot's object has a certain core in the node's
//   twin (pt_bvh.h build_cores, pt_walk32.h).
// (Round 3 also built and measured a 48-byte node -- the slot boxes on an 8-bit grid of the node's own, three 16-byte loads
// per visit instead of seven: profiles/r03_qnode_ab.txt.  The walks are bound by vector-instruction issue, not by the
// loads: the 35 instructions that decode such a node cost more than its four saved loads give back.)
struct alignas(128) BvhNode {   // one 128-byte cache line per node (108 bytes used; scan_bvh reads the first 96 only: ptcore.hip embeds the three words below in the low bytes of h)
    float c[3][4];    // [axis][slot] centre of the slot's inflated box
    float h[3][4];    // half extent, rounded up so that [c - h, c + h] holds it (the slab test is then three fma per axis and slot
                      // pair, no min / max to order the planes: see PT_BOX_SLABS in pt_kernels.h)
    int32_t node_base;
    int32_t obj_base;
    uint32_t meta;
    int32_t pad[5];
};
static_assert(sizeof(BvhNode) == 128, "BvhNode layout");

inline int32_t bvh_node_base(const BvhNode &n) { return n.node_base; }
inline int32_t bvh_obj_base(const BvhNode &n) { return n.obj_base; }
inline double bvh_slot_lo(const BvhNode &n, int k, int s) { return (double)n.c[k][s] - (double)n.h[k][s]; }
inline double bvh_slot_hi(const BvhNode &n, int k, int s) { return (double)n.c[k][s] + (double)n.h[k][s]; }

// Object as stored in node order for the BVH path: the 80-byte DevObj plus its index in file order
// (tie rules and the winner look-up use the original index).
struct alignas(16) BvhObj {
    DevObj o;
    int32_t index;
    int32_t pad[3];
};
static_assert(sizeof(BvhObj) == 96, "BvhObj layout");

struct DevCamera {     // camera.go:9-17 after newCamera (camera.go:19-58)
    double origin[3];
    double lower_left[3];
    double horizontal[3];
    double vertical[3];
    double u[3];
    double v[3];
    double lens_radius;
};

struct DevSky {
    int32_t kind;      // PT_SKY_*
    int32_t pad;
    double c0[3];      // gradient: horizon; solid: colour; background: background
    double c1[3];      // gradient: zenith
};

struct DevFrame {
    int32_t width, height;
    int32_t max_depth;
    int32_t nobj;
    int32_t nmat;        // materials incl. the zero material
    int32_t ntx, nty;    // 32x32 tiles of the frame
    int32_t shard_index, shard_count;
    int32_t nlocal;      // tiles owned by this shard
    uint32_t s0;         // first sample index of the chunk
    uint32_t S;          // samples per pixel in the chunk
    uint32_t njobs;      // nlocal*16*S*64 (also the plane stride of the primary-ray buffers)
    uint32_t fresh;      // fresh jobs this trace pass takes (njobs for a first pass, 0 for a pass over continuations only)
    int32_t n_dsph, n_dbox;  // dielectric-only broad-phase records (bsph_diel / bbox_diel)
    uint32_t claim;      // jobs a wave claims per queue pop (multiple of 64)
    int32_t n_bsph, n_bbox, n_plane;  // broad-phase record counts; planes are always tested exactly
    int32_t n_bvh_nodes, n_bvh_objs;  // BVH path (scenes beyond 128 spheres / 128 boxes)
    int32_t world_in_lds;             // 1: DevObj/DevMat copies are staged in LDS (small scenes)
    int32_t bvh_root;                 // root node of the hierarchy over every finite object, -1 if none
    int32_t bvh_root_exit;            // root of the hierarchy over dielectric objects only, -1 if none
    int32_t bvh_main_nodes;           // nodes of the main tree (breadth-first order)
    int32_t bvh_stack;                // traversal stack entries per lane (LDS)
    int32_t bvh_lds_nodes;            // top-level nodes of the main tree staged in LDS
    int32_t bvh_min_lanes;            // a traversal loop with fewer lanes still walking leaves them for the next trip
    int32_t bvh_node_min;             // the node walk yields to the exact tests of waiting lanes below this many walking lanes
    int32_t bvh_leaf_single;          // 1: one exact test per waiting lane and pass, then back to the walk
    int32_t planes_y;                 // 1: every plane has the normal (0, 1, 0) (all the engine ever builds): plane_exact_y applies
    uint32_t debug_drop;              // verify instantiations only (PTCORE_DEBUG_DROP): candidate bits cleared on purpose, so that
                                      // the disagreement counter can be shown to move
    int32_t broad_ok;    // 1: at most 32 sphere records and 32 box records -> candidate-bitmask scan usable;
                         // 2: at most 128 of each -> the same in groups of 32 (SCAN_BROAD_WIDE)
    uint32_t sph_all, box_all;    // (1 << n_bsph) - 1, (1 << n_bbox) - 1
    uint32_t sph_diel, box_diel;  // records whose object is dielectric (exit searches)
    float origin_bound;  // rays whose origin leaves [-origin_bound, origin_bound]^3 keep every candidate
    float pad_f;
    double scene_bound;  // Bs: every finite object (inflated) lies inside [-Bs, Bs]^3
    double clip_bound;   // 3.5 B: rays that start inside [-clip_bound, clip_bound]^3 are scanned from their origin, the others
                         // are clipped against the scene cube first (the FP32 bounds were analysed for origins within 4 B)
    double margin;       // m = B/4096: inflation of every FP32 bound
    uint64_t seed_key;   // ptm::seed_key(seed)
    double plane0_y;     // planes_y and n_plane == 1 (every scene file of the reference): the plane's point.y, its object index and kind --
    int32_t plane0_index;  // the scans then take the plane from the argument block instead of through plane_idx[] -> objs[] (two dependent
    int32_t plane0_kind;   // scalar loads at the head of every scan); plane0_index < 0: no such plane
    double inv_width;    // 1/(W-1)  renderer.go:95
    double inv_height;   // 1/(H-1)  renderer.go:96
    double height_m1;    // H-1      renderer.go:98
};

// Path states parked in HBM between passes (SoA, `cap` entries per plane; entries are appended with one
// wave-aggregated atomic per push, so the lanes of a push write consecutive slots).
//   glass queue        paths whose closest hit is a dielectric: the incoming ray, the hit (object index, t), throughput,
//                      stream state, depth.  glass_kernel scatters them (materials.go:162-200), runs the exit search
//                      (renderer.go:316-371) and Russian roulette, all lanes on the same branch.
//   continuation queue paths that go on after their dielectric bounce; the next trace pass takes them like fresh jobs.
struct PathQueue {
    double *d;                  // [10][cap]: ox oy oz dx dy dz Tx Ty Tz tmax
    unsigned long long *rs;     // [cap] stream state
    uint32_t *job;              // [cap]
    int32_t *depth;             // [cap] remaining depth (renderer.go:286 counts down)
    int32_t *best;              // [cap] glass queue: object hit; exit queue of the wavefront form: the glass material
    int32_t *hit;               // [cap] wavefront form: answer of the traversal pass (object index or -1; its t goes to plane 9 of d)
    uint32_t *jseg, *jdraw;     // [cap] per-job counters so far (PT_FLAG_PIXEL_STATS) or null
    uint32_t *count;            // entries appended so far
    uint32_t cap;
    uint32_t pad;
};

struct TraceBuffers {
    const DevObj *objs;
    const DevMat *mats;
    const BroadSphere *bsph;
    const BroadBox *bbox;
    const int32_t *plane_idx;
    const BvhNode *bvh_nodes;
    const BvhObj *bvh_objs;
    const BvhNode *bvh_cores;  // core twins of bvh_nodes (walk32 only)
    const double *ray;    // [6][njobs] primary rays of the chunk (raygen_kernel)
    const unsigned long long *ray_rng;  // [njobs] stream state after the camera draws
    const uint16_t *ray_ndraw;          // [njobs] draws used by ray generation; 0xffff = pixel outside the frame
    double *L;            // [njobs][4]: r, g, b, 0 (one 32-byte record per job)
    uint32_t *job_seg;    // [njobs] or null (PT_FLAG_PIXEL_STATS)
    uint32_t *job_draw;   // [njobs] or null
    unsigned int *queue;  // job queue head
    unsigned long long *counters;  // [4]: segments, exit_scans, draws, samples
    unsigned long long *prof;      // diagnostic build only: [SEC_COUNT][3] executions, lanes, cycles
    PathQueue glass;               // split passes: dielectric hits leave the trace kernel here
    PathQueue cont;                // continuation entries: read by trace_kernel (cont_in of them), written by glass_kernel
    const uint32_t *cont_in;       // number of continuation entries this trace pass starts from (device word; 0 for a first pass)
    const BroadSphere *bsph_diel;  // broad-phase records of the dielectric objects only (exit searches of glass_kernel)
    const BroadBox *bbox_diel;
};

// The argument block of trace_kernel (one by-value kernel argument, i.e. the kernarg segment).
struct TraceArgs {
    DevFrame F;
    DevSky sky;
    TraceBuffers B;
};

}  // namespace ptd
