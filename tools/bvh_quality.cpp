// bvh_quality.cpp -- host-only measure of the tree pt_bvh.h builds: node visits and exact tests per closest-hit
// query over random rays in a synthetic scene of the shape of path_trace_golang_amd/synth.py (no GPU needed).
//   g++ -O2 -std=c++17 -Ipath_trace_golang_amd/csrc tools/bvh_quality.cpp -o /tmp/bvh_quality && /tmp/bvh_quality 100000
// The walk follows scan_bvh's order (internal children nearest first, the rest stacked far to near, objects of a
// node tested exactly before the walk goes on); it is a yardstick for builder changes, not a model of the kernel.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "pt_bvh.h"

using namespace ptd;

static bool hit_sphere(const DevObj &o, const double *ro, const double *rd, double tmin, double tmax, double &t) {
    const double ocx = ro[0] - o.a[0], ocy = ro[1] - o.a[1], ocz = ro[2] - o.a[2];
    const double a = rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2];
    const double hb = ocx * rd[0] + ocy * rd[1] + ocz * rd[2];
    const double c = ocx * ocx + ocy * ocy + ocz * ocz - o.radius_sq;
    const double disc = hb * hb - a * c;
    if (disc < 0) return false;
    const double sq = std::sqrt(disc);
    double r = (-hb - sq) / a;
    if (r < tmin || r > tmax) {
        r = (-hb + sq) / a;
        if (r < tmin || r > tmax) return false;
    }
    t = r;
    return true;
}

static bool hit_box(const DevObj &o, const double *ro, const double *rd, double tmin, double tmax, double &t) {
    double t0 = tmin, t1 = tmax;
    for (int k = 0; k < 3; k++) {
        const double iv = 1.0 / rd[k];
        double n = (o.a[k] - ro[k]) * iv, f = (o.b[k] - ro[k]) * iv;
        if (iv < 0) std::swap(n, f);
        if (n > t0) t0 = n;
        if (f < t1) t1 = f;
        if (t1 <= t0) return false;
    }
    t = t0;
    return true;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 100000;
    const int nrays = argc > 2 ? std::atoi(argv[2]) : 200000;
    const double room = 40.0, h = room * 0.5;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<DevObj> world;
    auto add_box = [&](double px, double py, double pz, double sx, double sy, double sz) {
        DevObj o{};
        o.kind = KIND_BOX;
        o.a[0] = px - sx / 2; o.a[1] = py - sy / 2; o.a[2] = pz - sz / 2;
        o.b[0] = px + sx / 2; o.b[1] = py + sy / 2; o.b[2] = pz + sz / 2;
        world.push_back(o);
    };
    add_box(0, h, -h, room, room, 0.5); add_box(-h, h, 0, 0.5, room, room); add_box(h, h, 0, 0.5, room, room); add_box(0, room, 0, room, 0.5, room);
    const double base = std::max(0.05, std::min(1.5, 0.22 * room / std::cbrt((double)std::max(1, n - 4))));
    for (int i = 0; i < n - 4; i++) {
        const double px = -h * 0.9 + U(rng) * 1.8 * h, py = base + U(rng) * (room * 0.8 - base), pz = -h * 0.9 + U(rng) * 1.8 * h;
        const double s = base * (0.5 + U(rng));
        if (i % 23 == 0 || U(rng) < 0.6) {
            DevObj o{};
            o.kind = KIND_SPHERE;
            o.a[0] = px; o.a[1] = py; o.a[2] = pz;
            o.radius = (i % 23 == 0) ? s * 0.6 : s;
            o.radius_sq = o.radius * o.radius;
            o.inv_radius = 1.0 / o.radius;
            world.push_back(o);
        } else {
            add_box(px, py, pz, s * (0.6 + 1.2 * U(rng)), s * (0.6 + 1.2 * U(rng)), s * (0.6 + 1.2 * U(rng)));
        }
    }
    std::vector<int32_t> finite(world.size());
    for (size_t i = 0; i < world.size(); i++) finite[i] = (int32_t)i;
    const auto t0 = std::chrono::steady_clock::now();
    ptbvh::Built b = ptbvh::build(world, finite, 1e-4);
    const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    size_t slots = 0;
    for (const BvhNode &nd : b.nodes) slots += (size_t)__builtin_popcount((nd.meta >> 8) & 0xffu);
    std::printf("objects %zu  nodes %zu  wide levels %d  stack %d  slots used per node %.2f  build %.2f s\n", world.size(), b.nodes.size(), b.depth,
                b.stack_need, (double)slots / (double)b.nodes.size(), build_s);

    unsigned long long visits = 0, tests = 0, hits = 0, maxv = 0;
    std::vector<int> stack(256);
    for (int r = 0; r < nrays; r++) {
        double ro[3] = {-h * 0.95 + U(rng) * 1.9 * h, 0.05 + U(rng) * (room - 0.5), -h * 0.95 + U(rng) * 1.9 * h};
        double rd[3];
        for (;;) {
            rd[0] = 2 * U(rng) - 1; rd[1] = 2 * U(rng) - 1; rd[2] = 2 * U(rng) - 1;
            const double l = rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2];
            if (l > 1e-3 && l <= 1) break;
        }
        const double iv[3] = {1 / rd[0], 1 / rd[1], 1 / rd[2]};
        double tmax = 1e300;
        const double tmin = 1e-3;
        int sp = 0, cur = 0;
        unsigned long long v = 0;
        while (cur >= 0) {
            const BvhNode &nd = b.nodes[(size_t)cur];
            v++;
            double key[4];
            int kid[4], nk = 0;
            for (int s = 0; s < 4; s++) {
                if (!((nd.meta >> (8 + s)) & 1u) && !((nd.meta >> (12 + s)) & 1u)) continue;
                double a0 = tmin, a1 = tmax;
                for (int k = 0; k < 3; k++) {
                    const double tc = (0.5 * (bvh_slot_lo(nd, k, s) + bvh_slot_hi(nd, k, s)) - ro[k]) * iv[k], th = 0.5 * (bvh_slot_hi(nd, k, s) - bvh_slot_lo(nd, k, s)) * std::fabs(iv[k]);
                    a0 = std::max(a0, tc - th);
                    a1 = std::min(a1, tc + th);
                }
                if (a1 < a0) continue;
                const int rank = (int)((nd.meta >> (2 * s)) & 3u);
                if ((nd.meta >> (12 + s)) & 1u) {
                    const DevObj &o = world[(size_t)b.order[(size_t)(bvh_obj_base(nd) + rank)]];
                    tests++;
                    double t;
                    if ((o.kind == KIND_SPHERE) ? hit_sphere(o, ro, rd, tmin, tmax, t) : hit_box(o, ro, rd, tmin, tmax, t)) tmax = t;
                } else {
                    key[nk] = a0;
                    kid[nk++] = bvh_node_base(nd) + rank;
                }
            }
            for (int i = 1; i < nk; i++)  // nearest first
                for (int j = i; j > 0 && key[j] < key[j - 1]; j--) { std::swap(key[j], key[j - 1]); std::swap(kid[j], kid[j - 1]); }
            for (int i = nk - 1; i >= 1; i--) { stack[(size_t)sp++] = kid[i]; stack[(size_t)sp++] = 0; reinterpret_cast<float &>(stack[(size_t)sp - 1]) = (float)key[i]; }
            cur = -1;
            if (nk > 0 && key[0] <= tmax) cur = kid[0];
            while (cur < 0 && sp > 0) {
                const float k0 = reinterpret_cast<float &>(stack[(size_t)sp - 1]);
                sp -= 2;
                (void)k0;  // the kernel keeps no entry parameter on its stack: a stale entry costs one visit
                cur = stack[(size_t)sp];
            }
        }
        visits += v;
        if (v > maxv) maxv = v;
        if (tmax < 1e299) hits++;
    }
    std::printf("rays %d  node visits per ray %.2f (max %llu)  exact tests per ray %.2f  hit rate %.3f\n", nrays, (double)visits / nrays, maxv,
                (double)tests / nrays, (double)hits / nrays);
    return 0;
}
