// pt_bvh.h -- host-side BVH builder for scenes beyond the candidate bitmasks (more than 128 spheres or boxes).
//
// The reference scans every object for every ray segment (renderer.go:297-302).  Its winner is
// an order-free function of the per-object hit distances (see `wins` in pt_kernels.h), so any
// structure that never skips an object the exact test would accept returns the same winner.  The
// hierarchy stores FP32 boxes inflated by the same margin as the flat broad phase and rounded
// outward; the exact FP64 tests run only on objects whose own FP32 box the ray pierces.
//
// Build: binned-SAH binary tree down to single objects, then collapsed into 4-wide nodes (the
// child with the largest box is replaced by its two children until the node has four).  A slot
// of a wide node is an internal node, one object, or empty.  Nodes are numbered breadth-first, so
// the internal children of a node are consecutive (node_base + rank), its object children are
// consecutive in the object array (obj_base + rank), and the top of the tree is one prefix that
// the kernel stages in LDS.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "pt_device.h"

namespace ptbvh {

using namespace ptd;

constexpr int WIDTH = 4;

struct Aabb {
    double lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    }
    void grow(const Aabb &o) {
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); }
    }
    double area() const {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0;
        return 2 * (dx * dy + dy * dz + dz * dx);
    }
};

inline Aabb object_bounds(const DevObj &o) {
    Aabb b;
    const int kind = o.kind & 0xff;
    if (kind == KIND_SPHERE) {
        const double r = std::fabs(o.radius);
        for (int k = 0; k < 3; k++) { b.lo[k] = o.a[k] - r; b.hi[k] = o.a[k] + r; }
    } else {
        for (int k = 0; k < 3; k++) { b.lo[k] = std::min(o.a[k], o.b[k]); b.hi[k] = std::max(o.a[k], o.b[k]); }
    }
    for (int k = 0; k < 3; k++) {  // non-finite geometry: unbounded box, always descended into
        if (!(b.lo[k] == b.lo[k]) || !(b.hi[k] == b.hi[k])) { b.lo[k] = -INFINITY; b.hi[k] = INFINITY; }
    }
    return b;
}

struct Built {
    std::vector<BvhNode> nodes;      // nodes[0] is the root (present whenever there is an object)
    std::vector<int32_t> order;      // object slot -> index into the world array
    int depth = 0;                   // levels of wide nodes
    int stack_need = 0;              // most internal-node entries a traversal can have pushed at once
};

namespace detail {

inline float down(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -INFINITY);
    return f;
}
inline float up(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, INFINITY);
    return f;
}

struct BinNode {
    Aabb box;
    int32_t left = -1, right = -1;  // both -1: leaf
    int32_t obj = -1;               // leaf: world index
};

struct Builder {
    const std::vector<Aabb> &bounds;
    std::vector<double> cx[3];
    std::vector<int32_t> idx;       // working permutation of the object list
    std::vector<BinNode> bin;
    double margin;

    // chooses the split position inside [a, b): binned SAH (32 bins: 5 % fewer node visits per ray than 16 on the synthetic scenes, tools/bvh_quality.cpp) over each centroid axis, the cheapest of the
    // three wins; median of the widest axis when no bin boundary separates the objects
    int split(int a, int b) {
        Aabb cb;
        cb.reset();
        for (int i = a; i < b; i++)
            for (int k = 0; k < 3; k++) {
                const double c = cx[k][(size_t)idx[(size_t)i]];
                cb.lo[k] = std::min(cb.lo[k], c);
                cb.hi[k] = std::max(cb.hi[k], c);
            }
        int wide = 0;
        double ext = -1;
        for (int k = 0; k < 3; k++) {
            const double e = cb.hi[k] - cb.lo[k];
            if (e == e && e > ext && std::isfinite(e)) { ext = e; wide = k; }
        }
        const int mid = (a + b) / 2;
        if (!(ext > 0)) return mid;  // all centroids coincide (or are not finite): any split is as good
        constexpr int NB = 32;
        double best = INFINITY;
        int best_bin = -1, best_axis = -1;
        double best_scale = 0;
        for (int axis = 0; axis < 3; axis++) {
            const double e = cb.hi[axis] - cb.lo[axis];
            if (!(e > 0) || !std::isfinite(e)) continue;
            Aabb bb[NB];
            int cnt[NB];
            for (int i = 0; i < NB; i++) { bb[i].reset(); cnt[i] = 0; }
            const double scale = NB / e;
            for (int i = a; i < b; i++) {
                const int32_t o = idx[(size_t)i];
                const int q = std::max(0, std::min(NB - 1, (int)((cx[axis][(size_t)o] - cb.lo[axis]) * scale)));
                bb[q].grow(bounds[(size_t)o]);
                cnt[q]++;
            }
            Aabb right[NB];
            int rc[NB];
            Aabb acc;
            acc.reset();
            int n = 0;
            for (int i = NB - 1; i >= 0; i--) { acc.grow(bb[i]); n += cnt[i]; right[i] = acc; rc[i] = n; }
            acc.reset();
            n = 0;
            for (int i = 0; i + 1 < NB; i++) {
                acc.grow(bb[i]);
                n += cnt[i];
                if (n == 0 || rc[i + 1] == 0) continue;
                const double cost = acc.area() * n + right[i + 1].area() * rc[i + 1];
                if (cost < best) { best = cost; best_bin = i; best_axis = axis; best_scale = scale; }
            }
        }
        int m = mid;
        if (best_bin >= 0 && std::isfinite(best)) {
            const double lo = cb.lo[best_axis];
            auto it = std::partition(idx.begin() + a, idx.begin() + b, [&](int32_t o) {
                return std::max(0, std::min(NB - 1, (int)((cx[best_axis][(size_t)o] - lo) * best_scale))) <= best_bin;
            });
            m = (int)(it - idx.begin());
        }
        if (m <= a || m >= b) {
            std::nth_element(idx.begin() + a, idx.begin() + mid, idx.begin() + b,
                             [&](int32_t u, int32_t v) { return cx[wide][(size_t)u] < cx[wide][(size_t)v]; });
            m = mid;
        }
        return m;
    }

    // binary tree over idx[a, b); returns its node.  Below `sah_levels` levels the split is the median of the
    // object list (whatever SAH would like): a scene that makes SAH peel off one object per level
    // (geometrically spaced objects) must not drive the recursion or the traversal stack to depth n.
    int sah_levels = 40;
    int32_t build(int a, int b, int level = 0) {
        const int32_t me = (int32_t)bin.size();
        bin.emplace_back();
        if (b - a == 1) {
            bin[(size_t)me].obj = idx[(size_t)a];
            bin[(size_t)me].box = bounds[(size_t)idx[(size_t)a]];
            return me;
        }
        const int m = level < sah_levels ? split(a, b) : (a + b) / 2;
        const int32_t l = build(a, m, level + 1);
        const int32_t r = build(m, b, level + 1);
        BinNode &nd = bin[(size_t)me];
        nd.left = l;
        nd.right = r;
        nd.box = bin[(size_t)l].box;
        nd.box.grow(bin[(size_t)r].box);
        return me;
    }

    // the (up to WIDTH) binary nodes that become the slots of the wide node made from binary node b
    int gather(int32_t b, int32_t out[WIDTH]) const {
        int n = 0;
        if (bin[(size_t)b].obj >= 0) {  // a tree of one object: the root holds it in slot 0
            out[n++] = b;
            return n;
        }
        out[n++] = bin[(size_t)b].left;
        out[n++] = bin[(size_t)b].right;
        while (n < WIDTH) {
            int pick = -1;
            double area = -1;
            for (int k = 0; k < n; k++) {
                const BinNode &c = bin[(size_t)out[k]];
                if (c.obj >= 0) continue;
                double ar = c.box.area();
                if (!(ar == ar)) ar = INFINITY;  // unbounded boxes first
                if (ar > area) { area = ar; pick = k; }
            }
            if (pick < 0) break;
            const BinNode &c = bin[(size_t)out[pick]];
            out[pick] = c.left;
            out[n++] = c.right;
        }
        return n;
    }
};

}  // namespace detail

// `finite` lists the world indices of the spheres and boxes (planes stay outside the tree).
inline Built build(const std::vector<DevObj> &world, const std::vector<int32_t> &finite, double margin, int sah_levels = 40) {
    Built out;
    const int n = (int)finite.size();
    if (n == 0) return out;  // no finite objects: the kernel skips the traversal
    std::vector<Aabb> bounds(world.size());
    detail::Builder bl{bounds, {}, finite, {}, margin};
    for (int k = 0; k < 3; k++) bl.cx[k].assign(world.size(), 0.0);
    for (int32_t i : finite) {
        bounds[(size_t)i] = object_bounds(world[(size_t)i]);
        for (int k = 0; k < 3; k++) {
            double c = 0.5 * (bounds[(size_t)i].lo[k] + bounds[(size_t)i].hi[k]);
            if (!std::isfinite(c)) c = 0;
            bl.cx[k][(size_t)i] = c;
        }
    }
    bl.sah_levels = sah_levels;
    bl.bin.reserve((size_t)2 * n);
    const int32_t root = bl.build(0, n);

    // breadth-first collapse: queue entry = binary node that becomes wide node number q
    std::vector<int32_t> queue{root};
    std::vector<int32_t> level{1};
    out.order.reserve((size_t)n);
    for (size_t q = 0; q < queue.size(); q++) {
        int32_t slot[WIDTH];
        const int ns = bl.gather(queue[q], slot);
        BvhNode nd;
        std::memset(&nd, 0, sizeof nd);
        nd.node_base = (int32_t)queue.size();
        nd.obj_base = (int32_t)out.order.size();
        uint32_t ranks = 0, intm = 0, objm = 0, boxm = 0;
        int ni = 0, no = 0;
        for (int s = 0; s < WIDTH; s++) {
            if (s >= ns) {  // empty slot: never flagged in the masks; the box is a far-away point
                for (int k = 0; k < 3; k++) { nd.c[k][s] = 3.0e38f; nd.h[k][s] = 0.0f; }
                continue;
            }
            const detail::BinNode &c = bl.bin[(size_t)slot[s]];
            for (int k = 0; k < 3; k++) {
                const double lo = c.box.lo[k] - margin, hi = c.box.hi[k] + margin;
                const float cf = (float)(0.5 * lo + 0.5 * hi);
                if (lo == lo && hi == hi && std::isfinite(cf)) {
                    nd.c[k][s] = cf;
                    nd.h[k][s] = detail::up(std::max((double)cf - lo, hi - (double)cf));  // [c - h, c + h] holds [lo, hi]
                    if (!(nd.h[k][s] == nd.h[k][s])) nd.h[k][s] = INFINITY;
                } else {  // absurd or non-finite bounds: the slab constrains nothing
                    nd.c[k][s] = 0.0f;
                    nd.h[k][s] = INFINITY;
                }
            }
            if (c.obj >= 0) {
                ranks |= (uint32_t)no << (2 * s);
                objm |= 1u << s;
                if ((world[(size_t)c.obj].kind & 0xff) == KIND_BOX) boxm |= 1u << s;
                out.order.push_back(c.obj);
                no++;
            } else {
                ranks |= (uint32_t)ni << (2 * s);
                intm |= 1u << s;
                queue.push_back(slot[s]);
                level.push_back(level[q] + 1);
                ni++;
            }
        }
        nd.meta = ranks | (intm << 8) | (objm << 12) | (boxm << 16);
        out.nodes.push_back(nd);
        out.depth = std::max(out.depth, level[q]);
    }
    // deepest stack: a node with k internal children leaves at most k-1 entries below the subtree being walked
    std::vector<int32_t> need(out.nodes.size(), 0);
    for (size_t q = out.nodes.size(); q-- > 0;) {
        const BvhNode &nd = out.nodes[q];
        const int k = __builtin_popcount((nd.meta >> 8) & 0xfu);
        int deepest = 0;
        for (int c = 0; c < k; c++) deepest = std::max(deepest, need[(size_t)(nd.node_base + c)]);
        need[q] = k > 0 ? (k - 1) + deepest : 0;
    }
    out.stack_need = need[0];
    return out;
}

// The certain core of an object (pt_walk32.h): a box every point of which lies inside the object by the margin m.  A ray
// that passes through it is hit by the reference's FP64 test of the object (objects.go:37-61, :141-179) no later than where
// it enters the core: the FP32 slab arithmetic of the walk is off by <= 1.2e-6 B in position, two orders of magnitude less
// than m = B / 4096.
//   box     [min + m, max - m]; only a proper box (min < max on every axis: anything else the slab test never hits)
//   sphere  the cube of half side (|r| - m) / sqrt(3) about the centre (the reference squares the radius: its sign is irrelevant)
// Returns false when the object has no core (thinner than 2 m, degenerate, not finite).
inline bool object_core(const DevObj &o, double m, double lo[3], double hi[3]) {
    if (!(m > 0) || !std::isfinite(m)) return false;
    const int kind = o.kind & 0xff;
    if (kind == KIND_SPHERE) {
        const double r = std::fabs(o.radius);
        const double q = (r - m) * 0.57735026 * (1.0 - 1e-9);
        if (!(q > 0.25 * m) || !std::isfinite(q)) return false;
        for (int k = 0; k < 3; k++) { lo[k] = o.a[k] - q; hi[k] = o.a[k] + q; }
    } else if (kind == KIND_BOX) {
        for (int k = 0; k < 3; k++) {
            if (!(o.a[k] < o.b[k])) return false;
            lo[k] = o.a[k] + m;
            hi[k] = o.b[k] - m;
            if (!(hi[k] - lo[k] > 0.5 * m)) return false;
        }
    } else {
        return false;
    }
    for (int k = 0; k < 3; k++)
        if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) return false;
    return true;
}

// Core twins of the nodes of `b`: twin q holds, in slot s, a box INSIDE the core of the object in slot s of node q, as FP32
// centre / half extent rounded INWARD (meta bits 12-15: the slot has a core; no internal slots).  Also marks in the NODE's
// meta (bits 20-23) which object slots have a core.
inline std::vector<BvhNode> build_cores(Built &b, const std::vector<DevObj> &world, double margin) {
    std::vector<BvhNode> cores(b.nodes.size());
    for (size_t q = 0; q < b.nodes.size(); q++) {
        BvhNode &nd = b.nodes[q];
        BvhNode tw;
        std::memset(&tw, 0, sizeof tw);
        uint32_t has = 0;
        for (int s = 0; s < WIDTH; s++) {
            for (int k = 0; k < 3; k++) { tw.c[k][s] = 3.0e38f; tw.h[k][s] = 0.0f; }
            if (!((nd.meta >> (12 + s)) & 1u)) continue;
            const int32_t oi = b.order[(size_t)(nd.obj_base + (int)((nd.meta >> (2 * s)) & 3u))];
            double lo[3], hi[3];
            if (!object_core(world[(size_t)oi], margin, lo, hi)) continue;
            float c[3], h[3];
            bool ok = true;
            for (int k = 0; k < 3; k++) {
                const float cf = (float)(0.5 * lo[k] + 0.5 * hi[k]);
                const double g = std::min((double)cf - lo[k], hi[k] - (double)cf);  // [cf - g, cf + g] lies inside [lo, hi]
                // cf - hf and cf + hf are formed in FP32 by the walk (c*iv - h*|iv|): keep two more ulps of the larger magnitude clear
                const float hf = detail::down(g - 2.0 * 1.1920929e-7 * (std::fabs((double)cf) + std::fabs(g)));
                ok = ok && hf > 0 && std::isfinite(cf) && std::isfinite(hf);
                c[k] = cf;
                h[k] = hf;
            }
            if (!ok) continue;
            for (int k = 0; k < 3; k++) { tw.c[k][s] = c[k]; tw.h[k][s] = h[k]; }
            has |= 1u << s;
        }
        tw.meta = has << 12;
        nd.meta = (nd.meta & ~(0xfu << 20)) | (has << 20);
        cores[q] = tw;
    }
    return cores;
}

}  // namespace ptbvh
