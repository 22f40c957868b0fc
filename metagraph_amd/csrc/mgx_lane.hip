// mgx_lane.hip — the thread-per-read instantiation of the aligner's wave program (see wave_lane.hpp).
// Separate translation unit: the same sources (dev_graph.hpp, align_core.hpp) are compiled here with a
// one-lane wave; symbols live in namespace mgx_lane so that both instantiations coexist in libmgx.so.
#define mgx mgx_lane
#include "wave_lane.hpp"
#include "align_core.hpp"

using namespace mgx_lane;

// each thread owns one read at a time and one arena slice; 64 reads share a hardware wavefront
__global__ void __launch_bounds__(64) k_align_lane(AlignParams P, uint32_t n_slots) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ int8_t sm_rows[6 * 128];
    for (int x = threadIdx.x; x < 6 * 128; x += blockDim.x) {
        uint32_t code = (uint32_t)(x >> 7);
        uint8_t row = code != 5 ? decode_code(code) : 0;
        sm_rows[x] = P.score_matrix[(uint32_t)(row & 127) * 128 + (x & 127)];
    }
    __syncthreads();
    if (slot >= n_slots) return;
    KernelStats acc;
    memset(&acc, 0, sizeof(acc));
    Wave w;
    for (;;) {
        uint64_t read = atomicAdd(P.read_cursor, 1ull);
        if (read >= P.n_reads) break;
        align_read(w, P, read, slot, &acc, nullptr, sm_rows, nullptr, 0);
    }
    atomicAdd(&P.stats->rank_lines, acc.rank_lines);
    atomicAdd(&P.stats->select_lines, acc.select_lines);
    atomicAdd(&P.stats->bit_lines, acc.bit_lines);
    atomicAdd(&P.stats->columns, acc.columns);
    atomicAdd(&P.stats->extensions, acc.extensions);
    atomicAdd(&P.stats->seeds, acc.seeds);
    atomicAdd(&P.stats->capacity_errors, acc.capacity_errors);
}

extern "C" int mgx_launch_align_lane(const void *params, uint32_t n_slots, void *stream) {
    const AlignParams &P = *static_cast<const AlignParams *>(params);
    uint32_t blocks = (n_slots + 63) / 64;
    k_align_lane<<<blocks, 64, 0, (hipStream_t)stream>>>(P, n_slots);
    return (int)hipGetLastError();
}
