// lane_column_bench.hip — what does a lane-per-read chain step cost on the MI355X?  (a measurement, not a product
// kernel.)  Every lane extends its own synthetic read: it calls lane_column() (metagraph_amd/csrc/lane_column.hpp, the function
// pinned cell by cell against the product's chain_step) once per column along a path that spells the read with a substitution
// every `mm` characters, updates the x-drop cut-off as DefaultColumnExtender::extend does, and writes the column's 64-byte slot
// (flag byte + 8-bit S offset per cell, four words of metadata) to its own slice of a slot array — the arithmetic and the
// stores of a chain step without its graph access and convergence table.  Output: columns per second for 1 .. 8 waves per SIMD,
// next to the product's rate (1.77 x 10^9 columns of the bench batch in 1.149 s = 1.54 x 10^9 columns/s, 66 % of it chain steps).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I metagraph_amd/csrc -o lane_column_bench tools/lane_column_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define MGX_HD __host__ __device__ __forceinline__
#define MGX_DEV __host__ __device__ __forceinline__
namespace mgx {
constexpr int32_t NINF = INT32_MIN + 100;
template <class T> MGX_DEV T imin(T a, T b) { return a < b ? a : b; }
template <class T> MGX_DEV T imax(T a, T b) { return a > b ? a : b; }
MGX_DEV int32_t iabs(int32_t a) { return a < 0 ? -a : a; }
MGX_DEV double fma_f64(double a, double b, double c) { return __builtin_fma(a, b, c); }
enum { CF_REAL = 1, CF_S_IS_E = 2, CF_E_EXT = 4, CF_MATCH = 8, CF_S_IS_F = 16, CF_F_EXT = 32, CF_SP_REAL = 64 };
}
#include "lane_column.hpp"

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

using namespace mgx;

constexpr int L = 150;

template <int WPS>
__global__ void __launch_bounds__(64, WPS) k_lane_chain(const uint8_t *reads, uint4 *slots, uint32_t slot_cols, unsigned long long *cursor,
                                                        uint64_t n_reads, int mm, unsigned long long *columns_done, int32_t *score_sum) {
    __shared__ int8_t rows[4 * 128];                 // score rows of A C G T: match 2, mismatch -3 (the CLI defaults)
    for (int x = threadIdx.x; x < 4 * 128; x += 64) rows[x] = ("ACGT"[x >> 7] == (char)(x & 127)) ? 2 : -3;
    __syncthreads();
    const uint32_t lane_slot = blockIdx.x * 64 + threadIdx.x;
    uint4 *my_slots = slots + (uint64_t)lane_slot * slot_cols * 4;
    unsigned long long cols = 0;
    int32_t ssum = 0;
    for (;;) {
        const unsigned long long r = atomicAdd(cursor, 1ull);
        if (r >= n_reads) break;
        const uint8_t *q = reads + r * 160;
        // root column (extend_begin): S[0] = 0, then the insertion run within the x-drop
        const int32_t go = -5, ge = -2, xdrop = 27;
        int32_t cutoff = -xdrop, best = 0, min_cell = 0;
        int32_t pS[LFW], pF[LFW];
        for (int x = 0; x < LFW; ++x) { pS[x] = NINF; pF[x] = NINF; }
        pS[0] = 0;
        int32_t p_org = 0, p_trim = 0, p_size = 1;
        {
            int32_t v = go, t = 1;
            while (v >= cutoff && t < LFW - 8) { pS[t] = v; v += ge; ++t; }
            p_size = t;
        }
        for (int col = 0; col < L; ++col) {
            LaneColumnIn in;
            in.band_given = 0; in.band_begin = in.band_prev_end = 0;
            in.p_org = p_org; in.p_trim = p_trim; in.p_size = p_size;
            in.xdrop_cutoff = cutoff; in.start = 0; in.window_size = L; in.qlen = L; in.go = go; in.ge = ge;
            in.next_offset = col + 1; in.score = 0; in.in_seed = false;
            in.best_score = best; in.min_cell_score = min_cell; in.rel_cutoff = 0.95;
            in.partial_sum_offset = 0; in.psum_lin = 2; in.psum = nullptr; in.seed_off = 0;
            in.q = q;
            uint8_t c = q[col];
            if (mm && (col % mm) == mm - 1) c = "ACGT"[((c >> 1) + 1) & 3];          // a substitution on the path
            const int code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
            in.row = rows + code * 128;
            LaneColumnOut out;
            const int rc = lane_column(in, pS, pF, out);
            if (rc != LC_OK) break;
            // the slot: 16 bytes of metadata, then (4 flag bytes, 4 S offsets) per four cells, 24 cells (the compact form)
            const int32_t base = out.max_val;
            uint32_t w[16];
            w[0] = (uint32_t)col; w[1] = (uint32_t)out.begin | ((uint32_t)out.size << 16); w[2] = (uint32_t)base; w[3] = (uint32_t)out.max_pos;
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                uint32_t fw = out.fw[g], sw = 0;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int x = 4 * g + s;
                    const int32_t d = pS[x] == NINF ? -128 : pS[x] - base;
                    sw |= ((uint32_t)d & 0xFF) << (8 * s);
                }
                w[4 + 2 * g] = fw; w[5 + 2 * g] = sw;
            }
            uint4 *dst = my_slots + (uint64_t)(col % slot_cols) * 4;
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]); dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
            dst[2] = make_uint4(w[8], w[9], w[10], w[11]); dst[3] = make_uint4(w[12], w[13], w[14], w[15]);
            // the extension's loop state (:671-700)
            if (out.max_val - cutoff > xdrop) cutoff = out.max_val - xdrop;
            best = imax(best, out.max_val);
            min_cell = out.min_cell_score;
            p_org = out.org; p_trim = out.begin; p_size = out.size;
            ++cols;
        }
        ssum += best;
    }
    atomicAdd(columns_done, cols);
    atomicAdd(score_sum, ssum);
}

template <int WPS>
static void run(const uint8_t *d_reads, uint4 *d_slots, uint32_t slot_cols, uint64_t n_reads, int mm, int n_cu, unsigned long long *d_ctr, int32_t *d_sum) {
    const uint32_t blocks = (uint32_t)n_cu * 4u * WPS;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best_rate = 0;
    unsigned long long cols = 0;
    int32_t ssum = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(d_ctr, 0, 16)); CHECK(hipMemset(d_sum, 0, 4));
        CHECK(hipEventRecord(e0, 0));
        k_lane_chain<WPS><<<blocks, 64>>>(d_reads, d_slots, slot_cols, d_ctr, n_reads, mm, d_ctr + 1, d_sum);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2];
        CHECK(hipMemcpy(h, d_ctr, 16, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&ssum, d_sum, 4, hipMemcpyDeviceToHost));
        cols = h[1];
        const double rate = (double)cols / (ms * 1e-3);
        if (rate > best_rate) best_rate = rate;
    }
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, (const void *)k_lane_chain<WPS>));
    printf("    {\"waves_per_simd\": %d, \"mismatch_every\": %d, \"columns\": %llu, \"columns_per_s\": %.4g, \"vgprs\": %d, \"scratch_bytes_per_lane\": %d, \"score_checksum\": %d}",
           WPS, mm, cols, best_rate, fa.numRegs, (int)fa.localSizeBytes, ssum);
}

int main(int argc, char **argv) {
    const uint64_t n_reads = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4000000ull;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::vector<uint8_t> reads(n_reads * 160);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (uint64_t i = 0; i < reads.size(); ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; reads[i] = "ACGT"[x & 3]; }
    uint8_t *d_reads; uint4 *d_slots; unsigned long long *d_ctr; int32_t *d_sum;
    const uint32_t slot_cols = 160;
    const uint64_t max_lanes = (uint64_t)n_cu * 4 * 8 * 64;
    CHECK(hipMalloc(&d_reads, reads.size()));
    CHECK(hipMemcpy(d_reads, reads.data(), reads.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_slots, max_lanes * slot_cols * 64));
    CHECK(hipMalloc(&d_ctr, 16)); CHECK(hipMalloc(&d_sum, 4));
    printf("{\"device\": \"%s\", \"cus\": %d, \"reads\": %llu, \"read_length\": %d, \"window_cells\": %d, \"what\": \"lane_column() per column + one 64-byte slot store, one lane per read\",\n  \"product_columns_per_s\": 1.54e9, \"rows\": [\n",
           prop.name, n_cu, (unsigned long long)n_reads, L, LFW);
    run<1>(d_reads, d_slots, slot_cols, n_reads, 50, n_cu, d_ctr, d_sum); printf(",\n");
    run<2>(d_reads, d_slots, slot_cols, n_reads, 50, n_cu, d_ctr, d_sum); printf(",\n");
    run<3>(d_reads, d_slots, slot_cols, n_reads, 50, n_cu, d_ctr, d_sum); printf(",\n");
    run<4>(d_reads, d_slots, slot_cols, n_reads, 50, n_cu, d_ctr, d_sum); printf(",\n");
    run<8>(d_reads, d_slots, slot_cols, n_reads, 50, n_cu, d_ctr, d_sum); printf(",\n");
    run<4>(d_reads, d_slots, slot_cols, n_reads, 0, n_cu, d_ctr, d_sum); printf("\n  ]}\n");
    return 0;
}
