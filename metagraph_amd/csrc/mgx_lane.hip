// mgx_lane.hip — the lane-per-read kernel (lane_read.hpp): every lane of a wavefront aligns its own read, 64 reads per
// instruction, in front of the 8-lane group kernel of mgx_grp.hip.  A lane finishes a "simple" read completely (result record
// + output stream) or lists it for the group kernel, which then aligns it from scratch; see lane_read.hpp for the contract.
//
// Shape: persistent wavefronts; a wavefront takes 64 consecutive items of the work-sorted order at a time (neighbours need about
// the same number of columns), runs lane_read() on them — the column loop is where the lanes spend their time; a lane whose read
// ends or bails early idles until its 63 wave-mates are through — hands out the output-stream words of the whole wavefront with
// ONE atomic, and writes results.  Per-lane state: the DP window (32 S + 32 F cells) and the bookkeeping of one extension in
// VGPRs; the packed query and the cold per-read state in LDS (word-major, conflict free); column slots, S rows and the CIGAR
// runs of the trace in the wavefront's HBM scratch, interleaved over its lanes (the lanes commit their columns in lock-step:
// full-line stores); the node table in a private slice per lane.
#include <hip/hip_runtime.h>

// Built twice: as is — reads of up to 256 characters, two wavefronts per SIMD — and with -DMGX_LANE_SHORT for batches whose
// longest read has at most 160 characters (the benchmark's 150-bp reads): seven packed words per strand instead of ten, which
// together with the CIGAR runs and six cold words that left the LDS in round 6 fits THREE wavefronts per SIMD into the LDS
// (52 rows x 256 B = 13 312 B per wavefront); the register budget of three (168) is the compiler's to meet.
#ifdef MGX_LANE_SHORT
#define mgx mgx_lane_short_ns
#define MGX_LANE_MAX_L 160
#ifndef MGX_LANE_WAVES_PER_SIMD
#define MGX_LANE_WAVES_PER_SIMD 3
#endif
#define k_lane k_lane_short
#define mgx_launch_lane mgx_launch_lane_short
#define mgx_lane_waves_per_simd mgx_lane_short_waves_per_simd
#else
#define mgx mgx_lane_ns
#endif
#include "wave.hpp"
#include "graph_build.hpp"
#include "lane_read.hpp"

using namespace mgx;

#ifndef MGX_LANE_WAVES_PER_SIMD
#define MGX_LANE_WAVES_PER_SIMD 2
#endif

__global__ void __launch_bounds__(64, MGX_LANE_WAVES_PER_SIMD) k_lane(const LaneParams *__restrict__ LPp) {
    // (the parameter block stays in memory: its ~90 uniform words are re-read with scalar loads where they are used instead of
    // occupying — and spilling — scalar registers across the column loop)
    const LaneParams &LP = *LPp;
    __shared__ uint64_t s_qw[LANE_QWORDS][64];
    __shared__ uint32_t s_cold[LANE_COLD_WORDS][64];
    const int lane = (int)threadIdx.x;
    // the wavefront's scratch from this lane's word on (layout: LaneParams — column slots and S rows interleaved over the lanes)
    uint8_t *scratch = LP.scratch + (uint64_t)blockIdx.x * LP.wave_stride + 4u * (uint32_t)lane;
    LaneChip chip;
    chip.lane = lane;
    chip.qw = &s_qw[0][lane]; chip.qstride = 64;
    chip.cold = &s_cold[0][lane]; chip.cstride = 64;
    const uint64_t n_items = LP.P.n_items ? LP.P.n_items : LP.P.n_reads;
    LaneCounters ctr;
    memset(&ctr, 0, sizeof(ctr));
#if MGX_LANE_TIMERS
    const uint64_t t_kernel = cycle_clock();
    uint64_t t_emit = 0;
#endif
    {
        uint32_t *rec = lane_record(LP, scratch, lane);
        for (int x = 26; x < 32; ++x) gst(rec + x, 0u);
    }
    // Every lane holds one read at a time: a new one from the sorted order, or — after LR_AGAIN — the same one for its backward
    // pass, while its wave-mates move on to new reads.
    uint64_t read = 0, item = 0;
    int pass = 0;
    bool holding = false;
    for (;;) {
        LV<bool> wantv;
        wantv.v = !holding;
        const uint64_t wm = wave_ballot(wantv);
        LV<uint64_t> bv;
        bv.v = 0;
        if (lane == 0 && wm) bv.v = atomicAdd(LP.P.read_cursor, (unsigned long long)popc64(wm));
        const uint64_t base = wave_bcast(bv, 0);
        if (!holding) {
            item = base + (uint64_t)popc64(wm & ((1ull << lane) - 1));
            pass = 0;
            if (item < n_items) { holding = true; read = LP.P.order ? (uint64_t)gld(LP.P.order + item) : item; }
        }
        LV<bool> actv;
        actv.v = holding;
        if (!wave_ballot(actv)) break;                       // nobody holds a read and the order is exhausted
        const bool active = holding;
        int rc = LR_DONE;
        LaneResult R;
        R.words = 0; R.have_aln = 0;
        if (active) rc = lane_read(LP, read, (uint32_t)item, pass, scratch, chip, ctr, R);
        if (active && rc == LR_AGAIN) pass = 1; else holding = false;
        const bool done = active && rc == LR_DONE, bail = active && rc == LR_BAIL;
        // output-stream words of the wavefront's finished reads: one atomic, a prefix sum over the lanes
        LV<int32_t> wv;
        wv.v = done ? (int32_t)R.words : 0;
        const LV<int32_t> pre = wave_prefix_sum_excl(wv);
        const int32_t total = wave_sum(wv);
        LV<uint64_t> sv;
        sv.v = 0;
        if (lane == 0 && total) sv.v = atomicAdd(LP.P.out_cursor, (unsigned long long)total);
        const uint64_t so = wave_bcast(sv, 0) + (uint64_t)pre.v;
#if MGX_LANE_TIMERS
        const uint64_t te0 = cycle_clock();
#endif
        if (done) {
            lane_emit(LP, read, scratch, chip, R, so);
            uint32_t *rec = lane_record(LP, scratch, lane);
            gst(rec + 29, gld(rec + 29) + 1u); gst(rec + 30, gld(rec + 30) + R.rr.n_extensions);
            gst(rec + 31, gld(rec + 31) + (uint32_t)(R.rr.status != ST_OK));
        }
#if MGX_LANE_TIMERS
        t_emit += cycle_clock() - te0;
#endif
        // the reads for the group kernel, in processing order
        LV<bool> bl;
        bl.v = bail;
        const uint64_t bm = wave_ballot(bl);
        if (bm) {
            LV<uint64_t> pv;
            pv.v = 0;
            if (lane == 0) pv.v = atomicAdd(LP.bail_count, (unsigned long long)popc64(bm));
            const uint64_t p0 = wave_bcast(pv, 0);
            if (bail) gst(LP.bail_list + p0 + (uint64_t)popc64(bm & ((1ull << lane) - 1)), (uint32_t)read);
            // ... and why, for the statistics: one atomic per distinct reason of the wavefront
            uint64_t left = bm;
            while (left) {
                LV<uint32_t> rv;
                rv.v = ctr.reason;
                const uint32_t code = wave_bcast(rv, ctz64(left)) & 31u;
                LV<bool> same;
                same.v = bail && (ctr.reason & 31u) == code;
                const uint64_t sm = wave_ballot(same);
                if (lane == 0) atomicAdd(LP.bail_hist + code, (unsigned long long)popc64(sm));
                left &= ~sm;
            }
        }
    }
    // counters: one atomic per wavefront and counter
    const uint32_t *rec = lane_record(LP, scratch, lane);
    LV<int32_t> v;
    v.v = (int32_t)gld(rec + 27); const int32_t rl = wave_sum(v);
    v.v = (int32_t)gld(rec + 28); const int32_t sl = wave_sum(v);
    v.v = (int32_t)gld(rec + 26); const int32_t cl = wave_sum(v);
    v.v = (int32_t)gld(rec + 29); const int32_t nd = wave_sum(v);
    v.v = (int32_t)gld(rec + 30); const int32_t ne = wave_sum(v);
    v.v = (int32_t)gld(rec + 31); const int32_t nc = wave_sum(v);
#if MGX_LANE_TIMERS
    // (bail_hist[32 .. 39]: the sections of lane_read(), [40] lane_emit, [41] the kernel: cycles of lane 0 of every wavefront)
    if (lane == 0) {
        for (int x = 0; x < 8; ++x) atomicAdd(LP.bail_hist + 32 + x, (unsigned long long)ctr.t[x]);
        atomicAdd(LP.bail_hist + 40, (unsigned long long)t_emit);
        atomicAdd(LP.bail_hist + 41, (unsigned long long)(cycle_clock() - t_kernel));
    }
#endif
    if (lane == 0) {
        atomicAdd(&LP.P.stats->rank_lines, (unsigned long long)rl);
        atomicAdd(&LP.P.stats->select_lines, (unsigned long long)sl);
        atomicAdd(&LP.P.stats->columns, (unsigned long long)cl);
        atomicAdd(&LP.P.stats->fast_columns, (unsigned long long)cl);
        atomicAdd(&LP.P.stats->lane_lines, (unsigned long long)rl + (unsigned long long)sl);
        atomicAdd(&LP.P.stats->lane_columns, (unsigned long long)cl);
        atomicAdd(&LP.P.stats->extensions, (unsigned long long)ne);
        atomicAdd(&LP.P.stats->capacity_errors, (unsigned long long)nc);
        atomicAdd(LP.done_count, (unsigned long long)nd);
    }
}

// blocks = resident wavefronts (wavefront b owns LaneParams::scratch + b * wave_stride)
// d_params: the LaneParams of this launch in device memory
extern "C" int mgx_launch_lane(const void *d_params, uint32_t blocks, void *stream) {
    k_lane<<<blocks, 64, 0, (hipStream_t)stream>>>(static_cast<const LaneParams *>(d_params));
    return (int)hipGetLastError();
}
extern "C" int mgx_lane_waves_per_simd(void) { return MGX_LANE_WAVES_PER_SIMD; }
