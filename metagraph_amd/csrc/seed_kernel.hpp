// seed_kernel.hpp — the wave-per-read kernel around align_read (one 64-lane wavefront per read), instantiated for the
// seeding phase by mgx.hip (every graph but PRIMARY ones) and by mgx_primary.hip (-DMGX_WITH_PRIMARY=1).
#pragma once
#include "align_core.hpp"

// one wave per read, persistent over the batch; each wave owns one arena slice
#ifndef MGX_ALIGN_WAVES_PER_SIMD
#define MGX_ALIGN_WAVES_PER_SIMD 4
#endif
// WPS = waves per SIMD the register allocation targets: 4 for the kernels that extend; the seeding-only instantiation of
// short-read batches runs at 8 (64 VGPRs, seeding tables mostly in the arena) — a gather kernel gains more from the extra
// wavefronts than it loses to spills (measured 112 vs 117 ms per 2 M reads)
#ifndef MGX_SEED_WPS
#define MGX_SEED_WPS 8          // waves per SIMD of the short-read seeding instantiation
#endif
template <int PHASE, int WPS = MGX_ALIGN_WAVES_PER_SIMD>
__global__ void __launch_bounds__(64, WPS) k_align(mgx::AlignParams P, uint32_t lds_bytes) {
    const uint32_t slot = blockIdx.x;
    using namespace mgx;
    __shared__ Wave w;                // the wave's scalar state lives in LDS, not in registers
    // (a seeding-only instantiation needs neither the score rows nor, with the sdust counters in registers, more than the
    // interval lists of the scratch: 1280 bytes more of the CU's LDS for the per-read seeding tables of every wavefront)
    constexpr bool kExtends = (PHASE & PH_EXTEND) != 0;
    __shared__ __attribute__((aligned(16))) uint8_t sd_buf[kExtends || !MGX_HAS_REGTAB ? sizeof(SdustScratch) : SDUST_LDS_BYTES_REGTAB];
    __shared__ int8_t sm_rows[kExtends ? 6 * 128 : 16];
    SdustScratch &sd = *reinterpret_cast<SdustScratch *>(sd_buf);
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    KernelStats acc;
    memset(&acc, 0, sizeof(acc));
    if (kExtends) mgx::load_score_rows(P, sm_rows);
    const uint64_t n_items = P.n_items ? P.n_items : (P.n_items_ptr ? *P.n_items_ptr : P.n_reads);
    for (;;) {
        LV<uint64_t> rv;
        rv.v = 0;
        if (lane_id() == 0) rv.v = atomicAdd(P.read_cursor, 1ull);
        uint64_t item = wave_bcast(rv, 0);
        if (item >= n_items) break;
        uint64_t read = (PHASE == PH_EXTEND && P.order) ? P.order[item] : (PHASE == PH_SEED && P.seed_list) ? (uint64_t)P.seed_list[item] : item;
        const uint8_t *rec = nullptr;
        if (PHASE == PH_EXTEND && P.resume_in) {         // a later pass of the multi-pass extension: `read` is a retry position
            rec = P.resume_in + read * P.resume_rec_bytes;
            read = P.resume_reads[read];
        }
        align_read<PHASE>(w, P, read, slot, &acc, &sd, sm_rows, dyn_lds, lds_bytes, rec);
    }
    if (lane_id() == 0) {
        atomicAdd(&P.stats->rank_lines, acc.rank_lines);
        atomicAdd(&P.stats->select_lines, acc.select_lines);
        atomicAdd(&P.stats->bit_lines, acc.bit_lines);
        atomicAdd(&P.stats->columns, acc.columns);
        if (acc.fast_columns) atomicAdd(&P.stats->fast_columns, acc.fast_columns);
        atomicAdd(&P.stats->extensions, acc.extensions);
        atomicAdd(&P.stats->seeds, acc.seeds);
        atomicAdd(&P.stats->capacity_errors, acc.capacity_errors);
        if (acc.seed_lines) atomicAdd(&P.stats->seed_lines, acc.seed_lines);
        for (int x = 0; x < 8; ++x) { atomicAdd(&P.stats->cyc[x], acc.cyc[x]); atomicAdd(&P.stats->xcyc[x], acc.xcyc[x]); }
    }
}

