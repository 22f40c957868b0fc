// gather_ceiling.hip — measured ceiling for the aligner's gather kernels (k_map, k_seed): dependent random 64-byte line
// loads per second on one MI355X, as a function of resident wavefronts per SIMD, chains per lane and working-set size.
//
// Every lane walks `chains` independent pointer chases; one step loads a whole 64-B line (4 x 16 B, like one BOSS block)
// at an index derived from the previous line's contents, so successive loads of a chain are data-dependent — the access
// pattern of BOSS::fwd / map_to_edges.  Working sets: the size of the bench graph's block array (~104 MB, fits the
// 256 MiB Infinity Cache) and the size of blocks + 15-mer suffix-range table (~9 GB, DRAM).
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_ceiling tools/gather_ceiling.hip && ./gather_ceiling > profiles/rNN_gather_ceiling.json
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__global__ void k_fill(uint4 *buf, uint64_t n16) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        buf[i] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)(x * 3), (uint32_t)((x * 5) >> 32));
    }
}

template <int CHAINS>
__global__ void __launch_bounds__(64) k_chase(const uint4 *buf, uint64_t n_lines, uint32_t steps, uint64_t *sink) {
    uint64_t idx[CHAINS];
    const uint64_t tid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) idx[c] = ((tid * CHAINS + c) * 0x9E3779B97F4A7C15ull) % n_lines;
    uint64_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            const uint4 *p = buf + idx[c] * 4;
            uint4 a = p[0], b = p[1], cc = p[2], d = p[3];
            uint64_t v = ((uint64_t)(a.x ^ b.y ^ cc.z ^ d.w) << 32) | (a.y + b.z + cc.w + d.x);
            acc += v;
            // the next index depends on the loaded data AND on the step and the chain: a map idx -> idx alone (round 2) sends every
            // chain into the same few short cycles of the random mapping after ~sqrt(n) steps — for the 104 MB set that is
            // ~1000 steps, after which the "working set" was a few thousand lines sitting in L1 / L2 (the 200 G lines/s of
            // profiles/r02_gather_ceiling.json); the 9 GB set (tail ~8000 steps) was not affected
            idx[c] = ((v + s * 0x9E3779B97F4A7C15ull + (tid * CHAINS + c)) ^ (idx[c] * 0xD6E8FEB86659FD93ull)) % n_lines;
        }
    }
    if (acc == 0x1234567) sink[0] = acc;
}

template <int CHAINS>
static double run(const uint4 *buf, uint64_t n_lines, int waves_per_simd, uint32_t steps, uint64_t *sink, int n_cu) {
    const uint32_t blocks = (uint32_t)n_cu * 4u * (uint32_t)waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_chase<CHAINS><<<blocks, 64>>>(buf, n_lines, steps / 8 + 1, sink);      // warm-up
    CHECK(hipEventRecord(e0, 0));
    k_chase<CHAINS><<<blocks, 64>>>(buf, n_lines, steps, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double lines = (double)blocks * 64.0 * CHAINS * steps;
    return lines / (ms * 1e-3);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    uint64_t *sink;
    CHECK(hipMalloc(&sink, 8));
    const uint64_t sets[2] = { 104ull << 20, 9ull << 30 };
    const char *names[2] = { "104MB (block array, Infinity-Cache resident)", "9GB (blocks + 15-mer table, DRAM)" };
    printf("{\"device\": \"%s\", \"cus\": %d, \"unit\": \"1e9 dependent 64-B lines/s\", \"sets\": [\n", prop.name, n_cu);
    for (int si = 0; si < 2; ++si) {
        const uint64_t bytes = sets[si], n_lines = bytes / 64;
        uint4 *buf;
        CHECK(hipMalloc(&buf, bytes));
        k_fill<<<n_cu * 8, 256>>>(buf, bytes / 16);
        CHECK(hipDeviceSynchronize());
        printf("  {\"working_set\": \"%s\", \"rows\": [\n", names[si]);
        const int wps[6] = { 1, 2, 4, 6, 8, 8 };
        for (int wi = 0; wi < 5; ++wi) {
            const int w = wps[wi];
            const uint32_t steps = 2000;
            double r1 = run<1>(buf, n_lines, w, steps, sink, n_cu);
            double r2 = run<2>(buf, n_lines, w, steps, sink, n_cu);
            double r4 = run<4>(buf, n_lines, w, steps / 2, sink, n_cu);
            printf("    {\"waves_per_simd\": %d, \"chains1\": %.2f, \"chains2\": %.2f, \"chains4\": %.2f, \"GBps_chains4\": %.0f}%s\n",
                   w, r1 / 1e9, r2 / 1e9, r4 / 1e9, r4 * 64 / 1e9, wi < 4 ? "," : "");
        }
        printf("  ]}%s\n", si == 0 ? "," : "");
        CHECK(hipFree(buf));
    }
    printf("]}\n");
    return 0;
}
