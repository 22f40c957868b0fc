// mgx_seedlane.hip — the lane-per-read seeding kernel (seed_lane.hpp): every lane of a wavefront seeds its own read, 64 reads
// per instruction, in front of the wave-per-read seeder (k_align<PH_SEED>, mgx.hip).  A lane finishes a read completely
// (header, seeds, work key) or lists it for the wave program, which then seeds it from scratch.
//
// Shape: persistent wavefronts; a lane takes the next read of the batch as soon as it is through with its own (one atomic per
// wavefront and round for the lanes that want one); the seed-stream space of the round's finished reads is handed out with
// one atomic per wavefront.  Per lane: two packed strands in LDS (word-major, conflict free), the sdust scan's counters
// bit-sliced in registers, the seeds in the wavefront's HBM scratch, interleaved over the lanes.
#include <hip/hip_runtime.h>

#define mgx mgx_seedlane_ns
#define MGX_NO_EXTEND
#include "wave.hpp"
#include "graph_build.hpp"
#include "seed_lane.hpp"

using namespace mgx;

#ifndef MGX_SEEDLANE_WAVES_PER_SIMD
#define MGX_SEEDLANE_WAVES_PER_SIMD 4
#endif

template <int QW>
__global__ void __launch_bounds__(64, MGX_SEEDLANE_WAVES_PER_SIMD) k_seed_lane(const SeedLaneParams *__restrict__ SPp) {
    const SeedLaneParams &SP = *SPp;
    const AlignParams &P = SP.P;
    __shared__ uint64_t s_qw[2 * QW][64];
    __shared__ uint32_t s_cnt[16][64];
    const int lane = (int)threadIdx.x;
    SeedLaneChip chip;
    chip.qw = &s_qw[0][lane]; chip.qstride = 64;
    chip.qwords = QW; chip.max_l = QW == SL_QWORDS_SHORT ? SL_SHORT_L : SL_MAX_L;
    chip.sbuf = SP.scratch + (uint64_t)blockIdx.x * seed_lane_wave_scratch_words(SP.max_entries, SP.max_pending) + (uint32_t)lane; chip.sstride = 64;
    chip.cnt = &s_cnt[0][lane]; chip.cntstride = 64;
    chip.max_entries = (int32_t)SP.max_entries; chip.max_pending = (int32_t)SP.max_pending; chip.second_pass = (int32_t)SP.second_pass;
    const uint64_t n_front = SP.second_pass ? (uint64_t)gld(SP.in_count) : 0, n_back = SP.second_pass && SP.in_count_back ? (uint64_t)gld(SP.in_count_back) : 0;
    const uint64_t n_items = SP.second_pass ? n_front + n_back : (P.n_items ? P.n_items : P.n_reads);
    uint32_t c_rank = 0, c_sel = 0, c_bit = 0, c_seeds = 0, c_done = 0;
    for (;;) {
        // one read per lane and round
        LV<uint64_t> bv;
        bv.v = 0;
        if (lane == 0) bv.v = atomicAdd(P.read_cursor, 64ull);
        const uint64_t base = wave_bcast(bv, 0);
        if (base >= n_items) break;
        const uint64_t item = base + (uint64_t)lane;
        const bool active = item < n_items;
        const uint64_t read = !active ? 0 : !SP.second_pass ? item
                              : (uint64_t)gld(SP.in_list + (item < n_front ? item : SP.list_len - 1 - (item - n_front)));
        chip.second_pass = !SP.second_pass ? 0 : item < n_front ? 1 : 2;
        SeedLaneOut out;
#if MGX_SL_TIMERS
        out.t0 = cycle_clock();
        for (int x = 0; x < 8; ++x) out.t[x] = 0;
#endif
        out.reason = 0; out.L = 0;
        out.n_seeds[0] = out.n_seeds[1] = 0;
        int rc = SL_BAIL;
        if (active) rc = seed_lane_read(P, read, chip, out);
        const bool done = active && rc == SL_DONE, bail = active && rc == SL_BAIL;
#if MGX_SL_TIMERS
        { LV<bool> anyv; anyv.v = true; (void)wave_ballot(anyv); }
        SL_T(7);
#endif
        // seed-stream space of the wavefront's finished reads: one atomic, a prefix sum over the lanes
        LV<int32_t> wv;
        wv.v = done ? out.n_seeds[0] + out.n_seeds[1] : 0;
        const LV<int32_t> pre = wave_prefix_sum_excl(wv);
        const int32_t total = wave_sum(wv);
        LV<uint64_t> sv;
        sv.v = 0;
        if (lane == 0 && total) sv.v = atomicAdd(P.seed_cursor, (unsigned long long)total);
        const uint64_t so = wave_bcast(sv, 0) + (uint64_t)pre.v;
        if (done) {
            seed_lane_publish(P, read, chip, out, so);
            c_seeds += (uint32_t)wv.v; ++c_done;
        }
#if MGX_SL_TIMERS
        SL_T(6);
        if (lane == 0 && SP.bail_hist) for (int x = 0; x < 8; ++x) atomicAdd(SP.bail_hist + 16 + x, (unsigned long long)out.t[x]);
#endif
        if (active) { c_rank += out.ctr.rank_lines; c_sel += out.ctr.select_lines; c_bit += out.ctr.bit_lines; }
        // the reads for the wave program
        for (int side = 0; side < 2; ++side) {
            LV<bool> bl;
            bl.v = bail && (side == 1) == (SP.bail_count_back != nullptr && out.reason != 4u);
            const uint64_t bm = wave_ballot(bl);
            if (!bm) continue;
            LV<uint64_t> pv;
            pv.v = 0;
            if (lane == 0) pv.v = atomicAdd(side ? SP.bail_count_back : SP.bail_count, (unsigned long long)popc64(bm));
            const uint64_t at = wave_bcast(pv, 0) + (uint64_t)popc64(bm & ((1ull << lane) - 1));
            if (bl.v) {
                gst(SP.bail_list + (side ? SP.list_len - 1 - at : at), (uint32_t)read);
                if (SP.bail_hist) atomicAdd(SP.bail_hist + (out.reason & 15u), 1ull);
            }
        }
    }
    LV<int32_t> v;
    v.v = (int32_t)c_rank; const int32_t rl = wave_sum(v);
    v.v = (int32_t)c_sel; const int32_t sl = wave_sum(v);
    v.v = (int32_t)c_bit; const int32_t bl = wave_sum(v);
    v.v = (int32_t)c_seeds; const int32_t ns = wave_sum(v);
    v.v = (int32_t)c_done; const int32_t nd = wave_sum(v);
    if (lane == 0) {
        atomicAdd(&P.stats->rank_lines, (unsigned long long)rl);
        atomicAdd(&P.stats->select_lines, (unsigned long long)sl);
        atomicAdd(&P.stats->bit_lines, (unsigned long long)bl);
        atomicAdd(&P.stats->seed_lines, (unsigned long long)rl + (unsigned long long)sl + (unsigned long long)bl);
        atomicAdd(&P.stats->seeds, (unsigned long long)ns);
        atomicAdd(SP.done_count, (unsigned long long)nd);
    }
}

// blocks = resident wavefronts (wavefront b owns SeedLaneParams::scratch + b * seed_lane_wave_scratch_words(...))
// long_reads: the build for reads of more than SL_SHORT_L characters (SeedLaneParams::long_reads says the same to the host side)
extern "C" int mgx_launch_seed_lane(const void *d_params, uint32_t blocks, int long_reads, void *stream) {
    if (long_reads) k_seed_lane<SL_QWORDS_LONG><<<blocks, 64, 0, (hipStream_t)stream>>>(static_cast<const SeedLaneParams *>(d_params));
    else k_seed_lane<SL_QWORDS_SHORT><<<blocks, 64, 0, (hipStream_t)stream>>>(static_cast<const SeedLaneParams *>(d_params));
    return (int)hipGetLastError();
}
extern "C" int mgx_seed_lane_waves_per_simd(void) { return MGX_SEEDLANE_WAVES_PER_SIMD; }
