// lane_column.hpp — one column of a register-resident chain, computed by ONE lane (round 3 groundwork for the lane-per-read
// chain kernel of DESIGN.md §9.1; not part of any product kernel yet).
//
// chain_step() (align_core.hpp) spreads the window of a column over the 8 lanes of a group, four cells per lane, and pays for
// it with cross-lane scans and with per-read bookkeeping executed as full vector instructions for 8 reads.  lane_column() is
// the same arithmetic — DefaultColumnExtender::update_column (A/aligner_extender_methods.cpp:209-290) with its 4-wide
// overshoot and scalar tail, extend_ins_end (:293-328), the scan (:643-669) and the flag byte per cell that backtrack reads
// (ColSlot) — with the whole window of LFW cells in one lane, cell after cell: the E chain is a running scalar, nothing
// crosses lanes, and a wavefront computes 64 columns of 64 reads per pass.
//
// Pinned against chain_step() cell by cell: an -DMGX_LANE_CHECK build of the host model (8 lanes per read: FW == LFW) calls it
// on the inputs of every chain step of every extension the CPU suite and the fuzzing campaign run, and aborts on the first
// difference (tests/test_lane_column.py).  tools/lane_column_bench.hip measures it on the GPU.
#pragma once

namespace mgx {

// keeps the instruction scheduler from interleaving the eight four-cell blocks of the column pass: hoisting the profile scores and
// band tests of all 32 cells to the front of the pass costs ~100 live registers in the lane-per-read kernel
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_NO_SCHED_FENCE)
#define LANE_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define LANE_SCHED_FENCE() ((void)0)
#endif

// hides how a per-lane base was computed: the compiler otherwise folds x ge - bo ge back into (x - bo) ge, one multiplication per cell
#if defined(__HIP_DEVICE_COMPILE__)
#define LANE_COL_OPAQUE(v) asm volatile("" : "+v"(v))
#else
#define LANE_COL_OPAQUE(v) ((void)0)
#endif

#ifndef MGX_LFW
#define MGX_LFW 32
#endif
constexpr int LFW = MGX_LFW;                         // window cells held by a lane (32 == FW of the 8-lane groups, which the
                                                     // cross-check needs; the microbenchmark also measures 24, the compact slot's cells)

struct LaneColumnIn {
    int32_t p_org, p_trim, p_size;                   // the parent's window origin, trim and size (XState::f_org, f_trim, f_size)
    int32_t xdrop_cutoff, start, window_size, qlen, go, ge;
    int32_t next_offset, score;                      // the child's offset and edge score (call_outgoing)
    bool in_seed;
    int32_t best_score, min_cell_score;
    double rel_cutoff;
    int32_t partial_sum_offset, psum_lin;
    const int32_t *psum;                             // partial sums of the strand (used when psum_lin == 0)
    int32_t seed_off;
    int32_t band_given, band_begin, band_prev_end;   // != 0: the parent's band as computed earlier (the children of a fork share the
                                                     // band found with the cut-off at the time the parent was popped)
    const uint8_t *q;                                // the extender's query
    const int8_t *row;                               // score-matrix row of the child's character (128 entries)
};

struct LaneColumnOut {
    uint32_t fw[LFW / 4];                            // CF_* per cell, four cells per word (cell x: byte x & 3 of word x >> 2)
    int32_t begin, size, size0, pushes, org;
    int32_t max_val, max_pos, min_cell_score, converged;
    bool has_extension;
};

enum { LC_OK = 0, LC_EMPTY_BAND = 1, LC_FALLBACK = 2, LC_POP = 3 };

// profile_score_[c][start + a] (:38-59) of the child's character against the query, by window cell: prepare(ap0) is told the
// absolute position of cell 0 once the window's origin is known, at(x, ap) returns the score of cell x (ap == ap0 + x).
// This one reads the byte query and the score-matrix row (the 8-lane groups' data; the cross-check build); the lane-per-read
// kernel brings its own over the 2-bit packed strand (lane_read.hpp).
struct LaneProfBytes {
    static constexpr bool linear_psum_only = false;  // the caller may hand over a table of partial sums (LaneColumnIn::psum)
    const uint8_t *q; const int8_t *row; int32_t qlen;
    MGX_HD void prepare(int32_t) {}
    MGX_HD int32_t at(int, int32_t ap) const { return (ap >= 1 && ap <= qlen) ? (int32_t)row[q[ap - 1] & 127] : 0; }
};

// a + b, saturating (v_add_i32 ... clamp)
MGX_HD int32_t lane_add_sat(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_add_sat(a, b);
#else
    const int64_t t = (int64_t)a + (int64_t)b;
    return t < (int64_t)INT32_MIN ? INT32_MIN : (t > (int64_t)INT32_MAX ? INT32_MAX : (int32_t)t);
#endif
}

// band within the x-drop cut-off (:549-560): [begin, prev_end) in window positions; empty when prev_end <= begin.
// (Round 6: the cells at or above the cut-off are collected as one bit each, shifted in from the top cell down, and the two ends
// are the lowest and highest bit inside the column's range — three instructions per cell where the per-cell range test with its
// two selects and a min / max took ten; the or-of-selected-bits form of the same idea costs the allocator three spilled registers
// in the lane kernel: profiles/r06_ab12_lane_same_box.txt, r06_ab14_lane_band2.txt.)
MGX_HD void lane_band(const LaneColumnIn &in, const int32_t *S, int32_t &begin, int32_t &prev_end) {
    static_assert(LFW <= 32, "one bit per window cell");
    uint32_t m = 0;
#pragma unroll
    for (int x = LFW - 1; x >= 0; --x) m = (m << 1) | (S[x] >= in.xdrop_cutoff ? 1u : 0u);
    // the column's cells: j = x + p_org - p_trim in [0, p_size)
    const int32_t d = in.p_trim - in.p_org;
    const int32_t lo = imax(d, 0), hi = imin(d + in.p_size, LFW);
    const uint32_t range = hi > lo ? ((hi - lo >= 32 ? ~0u : (1u << (hi - lo)) - 1u) << lo) : 0u;
    m &= range;
    begin = INT32_MAX; prev_end = INT32_MIN;
    if (m) { begin = in.p_org + ctz64((uint64_t)m); prev_end = in.p_org + (64 - clz64((uint64_t)m)); }
}

// S, F: in = the parent's window (cell x is window position in.p_org + x), out = the child's (cell x is position out.org + x).
// LC_EMPTY_BAND: no parent cell within the x-drop (chain_step returns FR_END before anything else; window untouched).
// LC_FALLBACK: the column does not fit the window (untouched if the band itself does not fit; clobbered if the insertion
// run behind the column's end does not).  LC_POP: computed, but popped again (:646-653: below the cut-off or nothing left to
// gain).  LC_OK: S / F / out hold the column as chain_step would commit it.
template <class Prof>
MGX_HD int lane_column(const LaneColumnIn &in, int32_t *S, int32_t *F, LaneColumnOut &out, Prof &profile) {
    const int32_t go = in.go, ge = in.ge, score = in.score, cutoff = in.xdrop_cutoff;
    int32_t begin, prev_end;
    if (in.band_given) { begin = in.band_begin; prev_end = in.band_prev_end; }
    else lane_band(in, S, begin, prev_end);
    if (prev_end <= begin) return LC_EMPTY_BAND;
    const int32_t end = imin(prev_end, in.window_size) + 1;
    const int32_t size0 = end - begin;
    const int32_t max_size = in.window_size + 1 - begin;
    const int32_t n_prev = prev_end - begin, n_loop = (n_prev + 3) & ~3;
    const int32_t org = begin & ~3;
    if ((begin - org) + imax(n_loop, size0) > LFW) return LC_FALLBACK;
    profile.prepare(in.start + org);
    // the parent moves to the child's origin, four cells at a time (both origins are multiples of four); the cell just below
    // the new origin is kept: the first cell's match compares against it
    int32_t p_below = NINF;
    for (int32_t sh = org - in.p_org; sh >= 4; sh -= 4) {
        p_below = S[3];
#pragma unroll
        for (int x = 0; x < LFW - 4; ++x) { S[x] = S[x + 4]; F[x] = F[x + 4]; }
#pragma unroll
        for (int x = LFW - 4; x < LFW; ++x) { S[x] = NINF; F[x] = NINF; }
    }
    const bool tail = size0 > imax(1, n_prev);                    // scalar tail (:284-289)
    const int32_t diag_i = in.next_offset - (in.seed_off - 1);
    const int32_t extension_cutoff = (int32_t)fma_f64((double)in.best_score, in.rel_cutoff, (double)in.partial_sum_offset);
    const int32_t skip = begin ? 0 : 1;
    const int32_t bo = begin - org;                  // window cell of the column's cell 0 (0 .. 3)
    const bool has_del = in.next_offset > 1;
    int32_t sm1 = p_below;                           // parent S at a - 1
    int32_t run = INT32_MIN;                         // max over the cells so far of m + go - j ge
    int32_t e_next = NINF;                           // E[j] of the cell at hand as the recurrence gives it
    int32_t ce_prev = NINF;                          // E of the cell before, final
    bool pushing = false;                            // extend_ins_end (:293-328), decided at the column's last cell
    int32_t n_push = 0;
    int32_t mx = INT32_MIN, mn = INT32_MAX, key = INT32_MAX, conv = INT32_MIN;
    bool ext = false;
    // (round 6: everything of a cell that is linear in its index — jj ge, (jj + 1) ge, the insertion run behind the column's end,
    // the linear partial sum — is a per-lane base plus a multiple of a uniform: one addition with a scalar operand where the pass
    // spent three quarter-rate multiplications per cell)
    int32_t boge = bo * ge;                          // jj ge == x ge - boge
    int32_t go_boge = go + boge;                     // go - jj ge == go_boge - x ge
    int32_t ps0 = in.psum_lin ? (in.qlen - (in.start + org)) * in.psum_lin : 0;            // the linear partial sum under cell 0
    int32_t ins_base = 0;                            // ins_score - (bo + size0) ge: the run's cell t scores ins_base + x ge
    LANE_COL_OPAQUE(boge); LANE_COL_OPAQUE(go_boge); LANE_COL_OPAQUE(ps0);
    // cells from `hi` on hold nothing (beyond the overshoot of update_column and beyond the column's end): whole blocks of four
    // are skipped; the insertion run may move the end once, when the pass reaches the column's last cell
    int32_t hi = bo + imax(n_loop, size0);
    bool fallback = false;
#pragma unroll
    for (int b = 0; b < LFW / 4; ++b) {
        uint32_t fwb = 0;
        if (4 * b < hi) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int x = 4 * b + s4;
                const int32_t a = org + x, jj = x - bo;
                const int32_t ps = S[x], pf = F[x];          // the parent at a
                const int32_t ap = in.start + a;
                const int32_t prof = profile.at(x, ap);
                const int32_t mraw = sm1 + prof + score;
                const bool inl = (uint32_t)jj < (uint32_t)n_loop;
                const int32_t del = has_del ? imax(ps + go, pf + ge) + score : NINF;
                const int32_t match = jj >= 1 ? mraw : NINF;
                const int32_t m = imax(match, del);
                int32_t fv = inl ? del : NINF;
                run = inl ? imax(run, m + (go_boge - x * ge)) : run;
                // (E of the cell at hand: e_next is ninf wherever the cell before was not in the loop, i.e. outside 0 < jj <= n_loop,
                // and the recurrence's value — ninf extended — at jj == 0)
                int32_t ce = e_next;
                int32_t sv = imax(m, e_next);
                sv = (inl && sv > cutoff - 1) ? sv : NINF;
                // E[j + 1] = max(E[j] + ge, m[j] + go) in closed form over the column (E[0] = ninf extended j + 1 times, saturating)
                const int32_t from_e0 = lane_add_sat(NINF, (x + 1) * ge - boge);          // (dec < -100 ? INT32_MIN : ninf + dec, dec = (jj + 1) ge)
                e_next = inl ? imax(run + (x * ge - boge), from_e0) : NINF;
                if (jj == size0 - 1) {
                    if (tail) { const int32_t mt = imax(mraw, ce); if (mt >= cutoff) sv = mt; }
                    if (size0 < max_size) {
                        const int32_t ins = imax(sv + go, ce + ge);
                        if (ins >= cutoff) {
                            const int32_t room = max_size - (size0 + 1);
                            n_push = 1 + room;
                            if (ge != 0) n_push = 1 + imin(room, (int32_t)((uint32_t)(ins - cutoff) / (uint32_t)(-ge)));
                            // (no exit out of the unrolled pass: an exit edge makes the compiler keep two copies of the window and
                            // move one onto the other at every block; the pass runs on and its result is discarded)
                            fallback = bo + size0 + n_push > LFW;
                            n_push = fallback ? 0 : n_push;
                            pushing = !fallback;
                            ins_base = ins - (bo + size0) * ge;
                            LANE_COL_OPAQUE(ins_base);
                            hi = imax(hi, bo + size0 + n_push);
                        }
                    }
                }
                {
                    // behind the column's last cell: the insertion run, then nothing
                    const int32_t t = jj - size0;
                    const bool beyond = pushing && t >= 0;
                    const bool pc = beyond && t < n_push;
                    const int32_t v = ins_base + x * ge;                               // ins_score + t ge
                    sv = beyond ? (pc ? v : NINF) : sv;
                    ce = beyond ? (pc ? v : NINF) : ce;
                    fv = beyond ? NINF : fv;
                    const bool in_col = (uint32_t)jj < (uint32_t)size0 || pc;
                    // scan (:643-669) and what the convergence table takes for a node's first column (selects, not branches: every
                    // branch in the unrolled pass costs the compiler's exec-mask bookkeeping 32 times over)
                    const int32_t kk = (iabs(a - diag_i) << 12) | jj;
                    const int32_t svc = in_col ? sv : INT32_MIN;                       // a cell outside the column never wins
                    key = svc > mx ? kk : ((svc == mx && in_col) ? imin(key, kk) : key);
                    mx = imax(mx, svc);
                    mn = (in_col && sv != NINF) ? imin(mn, sv) : mn;
                    int32_t ps_here = ps0 - x * in.psum_lin;                          // (qlen - (start + a)) psum_lin
                    if (!Prof::linear_psum_only) { if (!in.psum_lin) ps_here = in_col ? in.psum[in.start + a] : 0; }
                    ext |= in_col && sv + ps_here >= extension_cutoff;
                    conv = (in_col && jj >= skip) ? imax(conv, sv) : conv;
                }
                // the flag byte (what backtrack compares, evaluated once: ColSlot)
                {
                    const int32_t ep = ce_prev;                  // (ninf at jj <= 0: no cell before it was in the loop)
                    const bool pin = a - 1 >= in.p_trim;
                    uint32_t fl = 0;
                    fl |= sv != NINF ? CF_REAL : 0;
                    fl |= sv == ce ? CF_S_IS_E : 0;
                    fl |= ce == ep + ge ? CF_E_EXT : 0;
                    fl |= (pin && sv == mraw) ? CF_MATCH : 0;
                    fl |= sv == fv ? CF_S_IS_F : 0;
                    fl |= fv == pf + score + ge ? CF_F_EXT : 0;
                    fl |= (pin && sm1 != NINF) ? CF_SP_REAL : 0;
                    fwb |= fl << (8 * s4);
                }
                S[x] = sv; F[x] = fv;
                sm1 = ps; ce_prev = ce;
            }
        } else {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) { S[4 * b + s4] = NINF; F[4 * b + s4] = NINF; }
        }
        out.fw[b] = fwb;
        LANE_SCHED_FENCE();
    }
    if (fallback) return LC_FALLBACK;
    const int32_t pushes = pushing ? n_push : 0;
    out.min_cell_score = imin(in.min_cell_score, mn);
    out.has_extension = in.in_seed || ext;
    out.max_val = mx;
    out.max_pos = begin + (key & 4095);
    out.begin = begin; out.size = size0 + pushes; out.size0 = size0; out.pushes = pushes; out.org = org;
    out.converged = conv;
    if ((!in.in_seed && mx < cutoff) || (!in.in_seed && !out.has_extension)) return LC_POP;
    return LC_OK;
}

// the byte-query form (LaneColumnIn::q / row)
MGX_HD int lane_column(const LaneColumnIn &in, int32_t *S, int32_t *F, LaneColumnOut &out) {
    LaneProfBytes pb;
    pb.q = in.q; pb.row = in.row; pb.qlen = in.qlen;
    return lane_column(in, S, F, out, pb);
}

} // namespace mgx
