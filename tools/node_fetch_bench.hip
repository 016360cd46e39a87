// node_fetch_bench.hip -- how many 128-byte "node visits" per second can the memory system serve when every lane of a wave
// chases its own chain of nodes (the access pattern of the BVH walk: seven 16-byte loads of one 128-byte record per lane and
// visit, the next record's index taken from the record just loaded)?
//   hipcc --offload-arch=gfx950 -O3 tools/node_fetch_bench.hip -o tools/node_fetch_bench && tools/node_fetch_bench
// Prints visits/s for tables of several sizes (L2-resident ... far beyond L2), 4 and 8 waves per SIMD, 7 or 3 loads per visit,
// and with a quarter of the lanes idle (as in the walk).  No arithmetic worth the name: what is measured is the fetch path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct alignas(128) Node { uint32_t w[32]; };
// SCATTER (round 4): the next index also depends on the step and the thread, so that the lanes do not all fall into the one
// short cycle a fixed random mapping of the table has (chase<*, false> on a table of any size ends up L2-resident after ~1000
// steps: profiles/r04_n3_summary.json, fetch_size_control) -- every visit is then an independent random record of the table.
template <int LOADS, bool SCATTER = false>
__global__ __launch_bounds__(256) void chase(const Node *__restrict__ t, uint32_t mask, int steps, int active_lanes, uint32_t *out) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
    uint32_t acc = 0;
    if ((int)(threadIdx.x & 63) < active_lanes) {
        for (int s = 0; s < steps; s++) {
            const uint4 *p = reinterpret_cast<const uint4 *>(&t[idx]);
            uint4 a = p[0], b = p[1], c = p[2];
            uint32_t x = a.x ^ b.y ^ c.z;
            if (LOADS == 7) {
                uint4 d = p[3], e = p[4], f = p[5], g = p[6];
                x ^= d.w ^ e.x ^ f.y ^ g.z;
            }
            acc += x;
            idx = (a.x + (x & 1u)) & mask;  // the next node comes out of this one
            if (SCATTER) idx = (idx ^ ((uint32_t)s * 0x9E3779B9u) ^ (tid * 0x85EBCA6Bu)) & mask;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    uint32_t *out; CHECK(hipMalloc(&out, 4));
    printf("{\"device\": \"%s\", \"cus\": %d, \"results\": [\n", prop.name, ncu);
    bool first = true;
    for (size_t mb : {1, 7, 16, 64, 512}) {
        size_t n = 1; while (n * 128 < mb * 1024 * 1024) n <<= 1;
        std::vector<Node> h(n);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < n; i++) for (int k = 0; k < 32; k++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].w[k] = (uint32_t)(s >> 16); }
        Node *d; CHECK(hipMalloc(&d, n * sizeof(Node)));
        CHECK(hipMemcpy(d, h.data(), n * sizeof(Node), hipMemcpyHostToDevice));
        for (int waves : {4, 8}) for (int loads : {7, 3}) for (int lanes : {64, 32}) {
            const int blocks = ncu * waves, steps = 2000;
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(a));
                if (loads == 7) hipLaunchKernelGGL(chase<7>, dim3(blocks), dim3(256), 0, 0, d, (uint32_t)(n - 1), steps, lanes, out);
                else hipLaunchKernelGGL(chase<3>, dim3(blocks), dim3(256), 0, 0, d, (uint32_t)(n - 1), steps, lanes, out);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            const double visits = (double)blocks * 4 * lanes * steps;
            printf("%s  {\"table_mib\": %zu, \"waves_per_simd\": %d, \"loads_per_visit\": %d, \"lanes\": %d, \"ms\": %.3f, \"g_visits_per_s\": %.1f, \"wave_visit_ns\": %.0f}",
                   first ? "" : ",\n", n * 128 >> 20, waves, loads, lanes, ms, visits / ms * 1e-6, ms * 1e6 / steps);
            first = false;
        }
        CHECK(hipFree(d));
    }
    if (getenv("NODE_FETCH_SCATTER")) {  // independent random records of a table far beyond L2 and the Infinity Cache
        size_t n = (size_t)1 << 24;  // 2 GiB
        std::vector<Node> h(n);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < n; i++) for (int k = 0; k < 32; k += 4) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].w[k] = (uint32_t)(s >> 16); }
        Node *d; CHECK(hipMalloc(&d, n * sizeof(Node)));
        CHECK(hipMemcpy(d, h.data(), n * sizeof(Node), hipMemcpyHostToDevice));
        for (int loads : {7, 3}) {
            const int waves = 4, lanes = 64, blocks = ncu * waves, steps = 500;
            hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(a));
                if (loads == 7) hipLaunchKernelGGL((chase<7, true>), dim3(blocks), dim3(256), 0, 0, d, (uint32_t)(n - 1), steps, lanes, out);
                else hipLaunchKernelGGL((chase<3, true>), dim3(blocks), dim3(256), 0, 0, d, (uint32_t)(n - 1), steps, lanes, out);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            const double visits = (double)blocks * 4 * lanes * steps;
            printf(",\n  {\"scatter\": true, \"table_mib\": %zu, \"waves_per_simd\": %d, \"loads_per_visit\": %d, \"lanes\": %d, \"steps\": %d, \"record_visits\": %.0f, \"ms\": %.3f, \"g_visits_per_s\": %.2f, \"gb_per_s_at_128B_per_visit\": %.0f}",
                   n * 128 >> 20, waves, loads, lanes, steps, visits, ms, visits / ms * 1e-6, visits * 128.0 / ms * 1e-6);
        }
        CHECK(hipFree(d));
    }
    printf("\n]}\n");
    return 0;
}
