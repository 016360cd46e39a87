// pt_bvh.h -- host-side BVH builder for scenes with more than 64 finite objects.
//
// The reference scans every object for every ray segment (renderer.go:297-302).  Its winner is
// an order-free function of the per-object hit distances (see `wins` in pt_kernels.h), so any
// structure that never skips an object the exact test would accept returns the same winner.  The
// hierarchy stores FP32 boxes inflated by the same margin as the flat broad phase and rounded
// outward; the exact FP64 tests run only at the leaves.
//
// Layout: binary tree, each 64-byte node carries BOTH children's boxes (one node fetch decides
// both descents); leaves hold up to LEAF_MAX objects, which are stored contiguously in leaf order
// (80-byte DevObj + original index) so a leaf is one contiguous read.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "pt_device.h"

namespace ptbvh {

using namespace ptd;

constexpr int LEAF_MAX = 4;

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
    std::vector<BvhNode> nodes;      // nodes[0] is the root (present even for 1 object)
    std::vector<int32_t> order;      // leaf order -> index into the world array
    int depth = 0;
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

struct Builder {
    const std::vector<Aabb> &bounds;
    std::vector<double> cx[3];
    std::vector<int32_t> &idx;
    std::vector<BvhNode> &nodes;
    double margin;
    int max_depth = 0;

    Aabb range_bounds(int a, int b) const {
        Aabb r;
        r.reset();
        for (int i = a; i < b; i++) r.grow(bounds[(size_t)idx[(size_t)i]]);
        return r;
    }
    void put_box(float *lo, float *hi, const Aabb &b) const {
        for (int k = 0; k < 3; k++) { lo[k] = down(b.lo[k] - margin); hi[k] = up(b.hi[k] + margin); }
    }
    static int32_t leaf_code(int first, int count) { return ~(int32_t)((uint32_t)first | ((uint32_t)(count - 1) << 28)); }

    // chooses the split position inside [a, b): binned SAH over the largest centroid axis, median fallback
    int split(int a, int b) {
        Aabb cb;
        cb.reset();
        for (int i = a; i < b; i++)
            for (int k = 0; k < 3; k++) {
                const double c = cx[k][(size_t)idx[(size_t)i]];
                cb.lo[k] = std::min(cb.lo[k], c);
                cb.hi[k] = std::max(cb.hi[k], c);
            }
        int axis = 0;
        double ext = -1;
        for (int k = 0; k < 3; k++) {
            const double e = cb.hi[k] - cb.lo[k];
            if (e == e && e > ext && std::isfinite(e)) { ext = e; axis = k; }
        }
        const int mid = (a + b) / 2;
        auto by_axis = [&](int32_t u, int32_t v) { return cx[axis][(size_t)u] < cx[axis][(size_t)v]; };
        if (!(ext > 0)) {
            return mid;  // all centroids coincide (or are not finite): any split is as good
        }
        constexpr int NB = 16;
        Aabb bb[NB];
        int cnt[NB];
        for (int i = 0; i < NB; i++) { bb[i].reset(); cnt[i] = 0; }
        const double scale = NB / ext;
        for (int i = a; i < b; i++) {
            const int32_t o = idx[(size_t)i];
            int bin = (int)((cx[axis][(size_t)o] - cb.lo[axis]) * scale);
            bin = std::max(0, std::min(NB - 1, bin));
            bb[bin].grow(bounds[(size_t)o]);
            cnt[bin]++;
        }
        double best = INFINITY;
        int best_bin = -1;
        Aabb left[NB], right[NB];
        int lc[NB], rc[NB];
        Aabb acc;
        acc.reset();
        int n = 0;
        for (int i = 0; i < NB; i++) { acc.grow(bb[i]); n += cnt[i]; left[i] = acc; lc[i] = n; }
        acc.reset();
        n = 0;
        for (int i = NB - 1; i >= 0; i--) { acc.grow(bb[i]); n += cnt[i]; right[i] = acc; rc[i] = n; }
        for (int i = 0; i + 1 < NB; i++) {
            if (lc[i] == 0 || rc[i + 1] == 0) continue;
            const double cost = left[i].area() * lc[i] + right[i + 1].area() * rc[i + 1];
            if (cost < best) { best = cost; best_bin = i; }
        }
        if (best_bin < 0 || !std::isfinite(best)) {
            std::nth_element(idx.begin() + a, idx.begin() + mid, idx.begin() + b, by_axis);
            return mid;
        }
        auto it = std::partition(idx.begin() + a, idx.begin() + b, [&](int32_t o) {
            int bin = (int)((cx[axis][(size_t)o] - cb.lo[axis]) * scale);
            bin = std::max(0, std::min(NB - 1, bin));
            return bin <= best_bin;
        });
        int m = (int)(it - idx.begin());
        if (m <= a || m >= b) {
            std::nth_element(idx.begin() + a, idx.begin() + mid, idx.begin() + b, by_axis);
            m = mid;
        }
        return m;
    }

    // returns the child code of the subtree over idx[a, b)
    int32_t build(int a, int b, int depth) {
        max_depth = std::max(max_depth, depth);
        const int n = b - a;
        if (n <= LEAF_MAX) return leaf_code(a, n);
        const int m = split(a, b);
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        {
            BvhNode nd;
            std::memset(&nd, 0, sizeof nd);
            put_box(nd.lo0, nd.hi0, range_bounds(a, m));
            put_box(nd.lo1, nd.hi1, range_bounds(m, b));
            nodes[(size_t)me] = nd;
        }
        const int32_t c0 = build(a, m, depth + 1);
        const int32_t c1 = build(m, b, depth + 1);
        nodes[(size_t)me].c0 = c0;
        nodes[(size_t)me].c1 = c1;
        return me;
    }
};

}  // namespace detail

// `finite` lists the world indices of the spheres and boxes (planes stay outside the tree).
inline Built build(const std::vector<DevObj> &world, const std::vector<int32_t> &finite, double margin) {
    Built out;
    out.order = finite;
    std::vector<Aabb> bounds(world.size());
    detail::Builder bl{bounds, {}, out.order, out.nodes, margin};
    for (int k = 0; k < 3; k++) bl.cx[k].assign(world.size(), 0.0);
    for (int32_t i : finite) {
        bounds[(size_t)i] = object_bounds(world[(size_t)i]);
        for (int k = 0; k < 3; k++) {
            double c = 0.5 * (bounds[(size_t)i].lo[k] + bounds[(size_t)i].hi[k]);
            if (!std::isfinite(c)) c = 0;
            bl.cx[k][(size_t)i] = c;
        }
    }
    const int n = (int)finite.size();
    if (n == 0) return out;  // no finite objects: the kernel skips the traversal
    if (n <= LEAF_MAX) {
        // too few objects for a split by the builder: one root over two leaves (the same leaf twice
        // when there is a single object; testing an object twice cannot change the winner)
        const int m = n > 1 ? n / 2 : 1;
        BvhNode nd;
        std::memset(&nd, 0, sizeof nd);
        bl.put_box(nd.lo0, nd.hi0, bl.range_bounds(0, m));
        nd.c0 = detail::Builder::leaf_code(0, m);
        if (n > 1) {
            bl.put_box(nd.lo1, nd.hi1, bl.range_bounds(m, n));
            nd.c1 = detail::Builder::leaf_code(m, n - m);
        } else {
            bl.put_box(nd.lo1, nd.hi1, bl.range_bounds(0, 1));
            nd.c1 = detail::Builder::leaf_code(0, 1);
        }
        out.nodes.push_back(nd);
        out.depth = 1;
        return out;
    }
    bl.build(0, n, 1);
    out.depth = bl.max_depth;
    return out;
}

}  // namespace ptbvh
