// fetch_calibrate.hip — what do rocprofv3's FETCH_SIZE / WRITE_SIZE count for the access patterns of the aligner's kernels?
// Every kernel below moves a KNOWN number of bytes; tools/fetch_calibrate.py runs this program under
// `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and divides the counters by the known bytes.
// Patterns: dependent random 64-byte line gathers (4 x 16 B per lane, like one BOSS block) over a DRAM-sized and an
// Infinity-Cache-sized set, a streaming read, random 64-byte line stores, a streaming write.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calibrate tools/fetch_calibrate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
__device__ __forceinline__ uint64_t chase(const uint4 *buf, uint64_t n_lines, uint32_t steps, uint64_t idx, uint64_t salt) {
    uint64_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        const uint4 *p = buf + idx * 4;
        uint4 a = p[0], b = p[1], cc = p[2], d = p[3];
        uint64_t v = ((uint64_t)(a.x ^ b.y ^ cc.z ^ d.w) << 32) | (a.y + b.z + cc.w + d.x);
        acc += v;
        idx = ((v + s * 0x9E3779B97F4A7C15ull + salt) ^ (idx * 0xD6E8FEB86659FD93ull)) % n_lines;      // (no short cycles: see gather_ceiling.hip)
    }
    return acc;
}
__global__ void __launch_bounds__(64) k_gather64_dram(const uint4 *buf, uint64_t n_lines, uint32_t steps, uint64_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    uint64_t acc = chase(buf, n_lines, steps, (tid * 0x9E3779B97F4A7C15ull) % n_lines, tid);
    if (acc == 0x1234567) sink[0] = acc;
}
__global__ void __launch_bounds__(64) k_gather64_cache(const uint4 *buf, uint64_t n_lines, uint32_t steps, uint64_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    uint64_t acc = chase(buf, n_lines, steps, (tid * 0x9E3779B97F4A7C15ull) % n_lines, tid);
    if (acc == 0x1234567) sink[0] = acc;
}
__global__ void k_stream_read(const uint4 *buf, uint64_t n16, uint64_t *sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n16; i += stride) { uint4 v = buf[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x1234567) sink[0] = acc;
}
__global__ void __launch_bounds__(64) k_scatter64(uint4 *buf, uint64_t n_lines, uint32_t steps) {
    const uint64_t tid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    uint64_t idx = (tid * 0x9E3779B97F4A7C15ull) % n_lines;
    for (uint32_t s = 0; s < steps; ++s) {
        uint4 *p = buf + idx * 4;
        const uint4 v = make_uint4((uint32_t)idx, s, (uint32_t)tid, 7u);
        p[0] = v; p[1] = v; p[2] = v; p[3] = v;
        idx = (idx * 0xD6E8FEB86659FD93ull + 0x632BE59BD9B4E019ull + s) % n_lines;
    }
}
__global__ void k_stream_write(uint4 *buf, uint64_t n16) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) buf[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    uint64_t *sink;
    CHECK(hipMalloc(&sink, 8));
    const uint64_t big = 9ull << 30, small = 104ull << 20;
    uint4 *bufb, *bufs;
    CHECK(hipMalloc(&bufb, big));
    CHECK(hipMalloc(&bufs, small));
    k_fill<<<n_cu * 8, 256>>>(bufb, big / 16);
    k_fill<<<n_cu * 8, 256>>>(bufs, small / 16);
    CHECK(hipDeviceSynchronize());
    const uint32_t blocks = (uint32_t)n_cu * 4u * 4u, steps = 1000;
    const double lanes = (double)blocks * 64.0;
    printf("{\"device\": \"%s\", \"kernels\": {\n", prop.name);
    k_gather64_dram<<<blocks, 64>>>(bufb, big / 64, steps, sink);
    printf("  \"k_gather64_dram\": {\"pattern\": \"dependent random 64-B line loads, 9 GB set\", \"read_bytes\": %.0f, \"written_bytes\": 0},\n", lanes * steps * 64.0);
    k_gather64_cache<<<blocks, 64>>>(bufs, small / 64, steps, sink);
    printf("  \"k_gather64_cache\": {\"pattern\": \"dependent random 64-B line loads, 104 MB set (Infinity-Cache resident)\", \"read_bytes\": %.0f, \"written_bytes\": 0},\n", lanes * steps * 64.0);
    k_stream_read<<<n_cu * 8, 256>>>(bufb, (2ull << 30) / 16, sink);
    printf("  \"k_stream_read\": {\"pattern\": \"streaming 16-B loads, 2 GB\", \"read_bytes\": %.0f, \"written_bytes\": 0},\n", (double)(2ull << 30));
    k_scatter64<<<blocks, 64>>>(bufb, big / 64, steps);
    printf("  \"k_scatter64\": {\"pattern\": \"random 64-B line stores (4 x 16 B per lane), 9 GB set\", \"read_bytes\": 0, \"written_bytes\": %.0f},\n", lanes * steps * 64.0);
    k_stream_write<<<n_cu * 8, 256>>>(bufb, (2ull << 30) / 16);
    printf("  \"k_stream_write\": {\"pattern\": \"streaming 16-B stores, 2 GB\", \"read_bytes\": 0, \"written_bytes\": %.0f}\n", (double)(2ull << 30));
    CHECK(hipDeviceSynchronize());
    printf("}}\n");
    return 0;
}
