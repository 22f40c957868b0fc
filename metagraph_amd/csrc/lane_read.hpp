// lane_read.hpp — ONE LANE PER READ: the whole alignment of a "simple" read by a single lane (round 4).
//
// Round 3 measured the 8-lane extension kernel at its issue wall: ~1 800 issued instructions per 8-read column step, >95 % of
// them per-read bookkeeping replicated across the 8 lanes of a group.  One lane per read makes every instruction serve 64
// reads.  lane_read() is the per-read program of DBGAligner<> (A/dbg_aligner.cpp:263-355,657-736) for the case almost every
// short read is: the first seed of the better strand extends forward along a non-branching path to the end of the read
// (every column a chain step, lane_column.hpp), its backtrack yields one alignment that starts at query position 0 (no backward
// pass), and every later seed dies on the convergence table (check_seed).  Reads without seeds finish here too.
//
// The contract with the 8-lane kernel is "complete or redo": a read either FINISHES here — result record and output stream
// written exactly as the group kernel would have written them — or BAILS at the first event outside the pattern (a fork, a node
// seen before, a band wider than the window, a deferred column, a dummy node, a backward pass, a live later seed, invalid
// characters, a capacity limit ...) without having written anything; bailed reads are listed and the group kernel aligns them
// from scratch.  Nothing is handed over mid-read, so results cannot depend on which kernel ran a read (tests run every read
// both ways).
//
// Label-aware alignment (LabeledAligner, A/aligner_labeled.cpp; round 6): the pattern is "one label all along".  The seeds of the
// read's strands carry exactly one label L, the same for all (filter_seeds :612-721 then keeps every seed with { L } or, below
// min_exact_match, none); every column's node has the row { L } (an only child inherits its parent's set unseen, flush() :81-137
// confirms it: checked here when the column is committed); at a fork the children without L are no children (:232-263); one
// backtrack reports the alignment with { L } (:304-359), which ends it (terminate_backtrack_start).  Anything else — a node with
// no or several labels, seeds of different labels — leaves the lane, as ever without having written anything.
//
// Restated (A/ = M/src/graph/alignment/), as far as the pattern reaches:
//   driver        A/dbg_aligner.cpp:360-384,657-755  (align_core, aln_both, align_both_directions) — flat_drive
//   extension     A/aligner_extender_methods.cpp:412-772 (extend), :209-328 (update_column, extend_ins_end: lane_column.hpp),
//                 :330-387 (call_outgoing), :66-156 (check_seed / update_seed_filter: first visit of a node only)
//   backtrack     A/aligner_extender_methods.cpp:774-1034
//   aggregator    A/aligner_aggregator.hpp:68-202 (one alignment)
// Every step mirrors the wave program of align_core.hpp (extend_begin / chain_step / bt_begin / bt_step / flat_drive), whose
// host model and GPU build are pinned to the oracle; the host model runs this function on every read of its test worlds
// (MGX_EMU_LANE=1) and the fuzzing campaign does the same.
#pragma once
#include "lane_types.hpp"

namespace mgx {

// per-lane views of the small on-chip arrays (LDS on the device, plain arrays in the host model)
struct LaneChip {
    int32_t lane;                        // the lane's index in its wavefront (its private scratch slice: lane_rest())
    uint64_t *qw; int32_t qstride;       // packed strand of the read: word i at qw[i * qstride]
    uint32_t *cold; int32_t cstride;     // the lane's cold state (below): word i at cold[i * cstride]
};

// Per-read state that is touched once per column or less — the best backtrack start, the parked child of a fork, the columns that
// stayed behind, the modelled capacities ... — lives in LDS, not in VGPRs: the DP window (64 registers) and the column pass need
// the register file (the kernel was at 250+ VGPRs and spilling with all of it in registers).  Accessed as plain variables through
// the macros below.  (Round 6: 44 -> 38 words, part of the LDS a third wavefront per SIMD needs — what is touched once or twice
// per pass — the read's length and seed count, the backward pass's clipping, the root's insertion run — moved to the lane's
// record in its scratch slice, arec() words 16 .. 19; the two line counters share a word, as do the two children's characters.)
enum { CD_B_SCORE, CD_B_NOD, CD_B_I, CD_B_POS, CD_T_SCORE, CD_T_NOD, CD_T_POS,
       CD_FA_ALIVE, CD_FA_CONV, CD_FA_MAX,
       CD_D_SCORE0, CD_D_SCORE1, CD_D_SCORE2, CD_D_MAX0, CD_D_MAX1, CD_D_MAX2,
       CD_KID_N0, CD_KID_C, CD_KID_N1, CD_KID_R0, CD_KID_R1,
       CD_TABLE_CAP, CD_TSB_LO, CD_TSB_HI, CD_CELL_TOP, CD_NCOLS, CD_F_NODE, CD_F_IDX, CD_F_MAX,
       CD_SEED_LEN, CD_SEED_OFF, CD_NODE0, CD_CTR, CD_FWD_N_NODES, CD_FWD_N_SEQ, CD_HMS, CD_REPLAY_TOP, CD_S8_FILTER,
       LANE_COLD_WORDS };
static_assert(LANE_MAX_DEFER == 3, "CD_D_* above");
#define LANE_CI(f) (*(int32_t *)(chip.cold + (f) * chip.cstride))
#define LANE_CU(f) (*(chip.cold + (f) * chip.cstride))
// CD_HMS: bit 0 = an alignment so far, bits 1-2 = where lane_emit() finds it (LANE_EMIT_*), bit 3 = the strand of pass 0,
// bit 4 = the replayed characters have all matched the query's so far (backward pass)
#define LANE_REPLAY_MATCHING() ((LANE_CI(CD_HMS) >> 4) & 1)
#define LANE_SET_REPLAY_MATCHING(v) (LANE_CI(CD_HMS) = (LANE_CI(CD_HMS) & ~16) | ((v) ? 16 : 0))
// bit 5 = a tie between columns beyond the query's end was resolved in the lane's own order (see the frontier): from then on the
// pass must not see anything the order of equal-score columns could change
#define LANE_TIE_MODE() ((LANE_CI(CD_HMS) >> 5) & 1)
#define LANE_SET_TIE_MODE(v) (LANE_CI(CD_HMS) = (LANE_CI(CD_HMS) & ~32) | ((v) ? 32 : 0))
// bit 6 = the replayed node's vector is kept exactly from here on (backward pass; a word of the lane's record in HBM until round 6:
// a store and a dependent load per replay column in the column loop's chain)
#define LANE_REPLAY_EXACT() ((LANE_CI(CD_HMS) >> 6) & 1)
#define LANE_SET_REPLAY_EXACT(v) (LANE_CI(CD_HMS) = (LANE_CI(CD_HMS) & ~64) | ((v) ? 64 : 0))
// bits 8 .. 31 = the read's label (label-aware alignment: the one label its seeds and every column of its extensions carry)
#define LANE_LABEL() ((uint32_t)LANE_CI(CD_HMS) >> 8)
#define LANE_HAVE_ALN() (LANE_CI(CD_HMS) & 1)
#define LANE_MODE() ((LANE_CI(CD_HMS) >> 1) & 3)
#define LANE_STRAND() ((LANE_CI(CD_HMS) >> 3) & 1)
#define LANE_SET_HAVE_ALN(v) (LANE_CI(CD_HMS) = (LANE_CI(CD_HMS) & ~1) | ((v) ? 1 : 0))
#define LANE_SET_MODE(v) (LANE_CI(CD_HMS) = (LANE_CI(CD_HMS) & ~6) | ((int32_t)(v) << 1))
// hides where a value came from: what is computed from it afterwards is computed again, not kept in registers across the column loop
#if defined(__HIP_DEVICE_COMPILE__)
#define LANE_OPAQUE(x) asm volatile("" : "+v"(x))
#define LANE_OPAQUE_PTR(p) asm volatile("" : "+v"(p))
#else
#define LANE_OPAQUE(x) ((void)0)
#define LANE_OPAQUE_PTR(p) ((void)0)
#endif

// profile scores over the packed strand (see LaneProfBytes)
struct LaneProfPacked {
    static constexpr bool linear_psum_only = true;   // the lane kernel takes reads without invalid characters only: psum_lin != 0
    const uint64_t *qw; int32_t qstride, qlen;
    uint32_t rowp;                       // the column character's row of LaneParams::t4
    uint64_t w;                          // codes of the query characters under cells 0 .. 31 (cell x: bits 2x, 2x + 1)
    MGX_HD void prepare(int32_t ap0) {
        const int32_t p = ap0 - 1;       // query index under cell 0 (-1: cell 0 lies before the query)
        if (p < 0) { w = qw[0] << 2; return; }
        const int32_t wi = p >> 5, sh = 2 * (p & 31);
        const uint64_t lo = qw[wi * qstride], hi = qw[(wi + 1) * qstride];
        w = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    }
    MGX_HD int32_t at(int x, int32_t ap) const {
        const uint32_t code = (uint32_t)(w >> (2 * x)) & 3u;
        const int32_t v = (int32_t)(int8_t)(rowp >> (8 * code));
        return (ap >= 1 && ap <= qlen) ? v : 0;
    }
};

// timing probes (never in a product build): MGX_LANE_PROBE bit 0 = no node-table probe / insert, bit 1 = no slot stores,
// bit 2 = the select of lane_children predicted from DevGraph::sel_anchor (global memory) instead of the last_hint fetch
#ifndef MGX_LANE_PROBE
#define MGX_LANE_PROBE 0
#endif
enum { LR_DONE = 0, LR_BAIL = 1, LR_AGAIN = 2 };      // LR_AGAIN: call again for the same read with pass = 1 (the backward pass)

// -DMGX_LANE_TIMERS (measurement builds): wave cycles by section of lane_read() — 0 read set-up, 1 the head's children (graph),
// 2 the column pass, 3 node table + commit + start cells, 4 frontier, 5 later seeds + trace, 6 result
#ifndef MGX_LANE_TIMERS
#define MGX_LANE_TIMERS 0
#endif
struct LaneCounters {
    uint32_t reason;                             // which test sent the read to the group kernel
#if MGX_LANE_TIMERS
    uint64_t t[8], t0;
#endif
};
#if MGX_LANE_TIMERS
#define LANE_T(i) do { const uint64_t t_ = cycle_clock(); ctr.t[i] += t_ - ctr.t0; ctr.t0 = t_; } while (0)
#else
#define LANE_T(i) ((void)0)
#endif
#define LANE_BAIL(code) do { ctr.reason = (code); return LR_BAIL; } while (0)

// the children of `v` on the reverse-complement view of the graph (RCDBG::call_outgoing_kmers, rc_dbg.hpp:88-99): the parents of
// v with the complement of the first character of their k-mers (BOSS::call_incoming_to_target boss.cpp:766-786 through
// NodeFirstCache, node_first_cache.cpp:38-52; dev_graph.hpp incoming() without its arrays); a '$' first character is dropped
// (aligner_extender_methods.cpp:381-384).  Same return convention as lane_children.
MGX_DEV int lane_parents(const DevGraph &g, uint32_t vv, uint32_t &n0, uint32_t &c0, uint32_t &n1, uint32_t &c1, const LaneChip &chip) {
    LineCtr lc = { 0, 0, 0 };
    const uint64_t v = vv;
    int n = 0;
    auto add = [&](uint64_t e) {
        const uint32_t cc = first_char(g, e, lc);
        if (cc == 0) return;
        if (n == 0) { n0 = (uint32_t)e; c0 = 5u - cc; } else if (n == 1) { n1 = (uint32_t)e; c1 = 5u - cc; }
        ++n;
    };
    const uint64_t x = bwd(g, v, lc);
    const uint32_t d = node_last_value(g, v);
    if (in_graph(g, x)) add(x);
    // edges after x labelled d + SIGMA, up to the next unflagged d
    uint64_t pos = x + 1;
    uint32_t bi = (uint32_t)(pos >> 6);
    while (pos <= g.n) {
        ++lc.rank_lines;
        const Block b = load_block(g, bi);
        const uint64_t from = ~(mask_upto((int)(pos & 63)) >> 1);          // bits >= pos & 63
        uint64_t cm = code_mask(b, d) & from;
        if (bi == g.n_blocks - 1 && ((g.n + 1) & 63)) cm &= mask_upto((int)(g.n & 63));
        const uint64_t stop = cm & ~b.pf;
        uint64_t flg = cm & b.pf;
        if (stop) flg &= mask_upto(ctz64(stop));
        while (flg) {
            const int j = ctz64(flg);
            flg &= flg - 1;
            const uint64_t e = ((uint64_t)bi << 6) + (uint32_t)j;
            if (in_graph(g, e)) add(e);
        }
        if (stop) break;
        ++bi;
        pos = (uint64_t)bi << 6;
    }
    LANE_CU(CD_CTR) += (lc.rank_lines + lc.bit_lines) | (lc.select_lines << 16);
    return n > 2 ? 3 : n;
}

// a global store the compiler will not merge with its neighbours
MGX_DEV void lane_store_single(uint32_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    *(volatile __attribute__((address_space(1))) uint32_t *)(uintptr_t)p = v;
#else
    *p = v;
#endif
}

// value of window cell x (dynamic) of a register window: a chain of selects, never an indexed access (the window must stay in
// registers)
MGX_HD int32_t lane_win_at(const int32_t *W, int32_t x) {
    int32_t v = NINF;
#pragma unroll
    for (int t = 0; t < LFW; ++t) v = x == t ? W[t] : v;
    return v;
}
MGX_HD uint32_t lane_flags_at(const uint32_t *fw, int32_t x) {
    uint32_t v = 0;
#pragma unroll
    for (int t = 0; t < LFW / 4; ++t) v = (x >> 2) == t ? fw[t] : v;
    return (uint32_t)x < (uint32_t)LFW ? (v >> (8 * (x & 3))) & 0xFFu : 0u;
}

MGX_HD uint32_t lane_hash(uint32_t key, uint32_t mask) {
    uint32_t h = (key ^ (key >> 15)) * 0x85EBCA6Bu;
    h ^= h >> 13;
    return h & mask;
}

// the children of `v` on the forward graph (DBGSuccinct::call_outgoing_kmers, dbg_succinct.cpp:110-139, minus the sentinel-
// labelled children the extender drops, aligner_extender_methods.cpp:381-384; dev_graph.hpp outgoing() without its arrays):
// returns their number — 3 stands for "more than two" — and the first two (node, label code) in edge order
MGX_DEV int lane_children(const DevGraph &g, uint32_t vv, uint32_t sel_r, uint32_t &n0, uint32_t &c0, uint32_t &r0,
                          uint32_t &n1, uint32_t &c1, uint32_t &r1, const LaneChip &chip) {
    // sel_r: NF[c] + rank_W(v, c) of v's own label c if the caller knows it (r0 / r1 of the call that enumerated v: a child's
    // edge lies in a block that call held, so the rank half of the child's fwd() costs it a few instructions and saves this call
    // the line of v's block), else 0.  r0 / r1: the same for the children returned here.
    LineCtr lc = { 0, 0, 0 };
    const uint64_t v = vv;
    int n = 0;
    Block tgt;
    uint64_t lst = 0;
    if (sel_r) {
        // (the rank_W primitive of this fwd() ran on the block the caller held: counted as the line the algorithm requires —
        // SURVEY 8(d) counts primitives — although no request leaves for it; the PMC traffic shows the saving)
        ++lc.rank_lines;
#if MGX_LANE_PROBE & 4
        lst = select_last_scan(g, sel_r, sel_predict(g.sel_anchor, g.sel_shift, sel_r), tgt, lc);
#else
        lst = select_last_blk(g, sel_r, tgt, lc);
#endif
    } else {
        ++lc.rank_lines;
        const Block cur = load_block(g, (uint32_t)(v >> 6));
        const uint32_t w = block_W(cur, (int)(v & 63));
        if (!(v > 1 && w == 0)) lst = fwd_from(g, v, cur, w % SIGMA, tgt, lc);
    }
    if (lst) {
        uint64_t first = pred_last_from(g, lst - 1, ((lst - 1) >> 6) == (lst >> 6) ? tgt : load_block(g, (uint32_t)((lst - 1) >> 6)), lc) + 1;
        if (first < 2) first = 2;
        Block b = tgt;
        uint32_t bi = (uint32_t)(lst >> 6);
        for (uint64_t i = first; i <= lst; ++i) {
            if ((uint32_t)(i >> 6) != bi) { bi = (uint32_t)(i >> 6); ++lc.rank_lines; b = load_block(g, bi); }
            const uint32_t c = block_W(b, (int)(i & 63)) % SIGMA;
            if (c != 0 && in_graph(g, i)) {
                const uint32_t r = nf_of(g, c) + block_rank_W(b, (int)(i & 63), c, bi == 0);
                if (n == 0) { n0 = (uint32_t)i; c0 = c; r0 = r; } else if (n == 1) { n1 = (uint32_t)i; c1 = c; r1 = r; }
                ++n;
            }
        }
    }
    LANE_CU(CD_CTR) += lc.rank_lines | (lc.select_lines << 16);
    return n > 2 ? 3 : n;
}

// where the alignment lane_emit() writes out lives: the column slots of the (only) extension, walked down the parent links; the
// path arrays the forward alignment was moved to when a backward pass followed; or the slots of the backward extension, whose
// reversal (Alignment::reverse_complement) is what a walk down the parent links yields anyway
enum { LANE_EMIT_SLOTS = 0, LANE_EMIT_ARRAYS = 1, LANE_EMIT_SLOTS_REVERSED = 2 };

// ---- the wavefront's scratch as one lane sees it (layout: LaneParams, lane_types.hpp) ----
// `il`: the interleaved part from this lane's word on (wave base + 4 * lane): word w of column c's slot, of its S row
MGX_HD uint32_t *lane_slot_word(const uint8_t *il, int32_t c, int w) {
    return (uint32_t *)il + ((uint32_t)c * LANE_SLOT_WORDS + (uint32_t)w) * LANE_WAVE;
}
MGX_HD uint8_t *lane_slot_flag(const uint8_t *il, int32_t c, int32_t x) {          // flag byte of window cell x
    return (uint8_t *)lane_slot_word(il, c, x >> 2) + (x & 3);
}
MGX_HD uint32_t *lane_s8_word(const uint8_t *il, uint32_t max_cols, int32_t c, int w) {
    return (uint32_t *)il + (max_cols * LANE_SLOT_WORDS + (uint32_t)c * LANE_S8_WORDS + (uint32_t)w) * LANE_WAVE;
}
MGX_HD uint8_t *lane_s8_byte(const uint8_t *il, uint32_t max_cols, int32_t c, int32_t x) {
    return (uint8_t *)lane_s8_word(il, max_cols, c, x >> 2) + (x & 3);
}
// CIGAR run i of the trace (last first), behind the S rows: written when a run ends, read when the alignment is written out — the
// 4 KB per wavefront they took in LDS until round 6 are part of what a third wavefront per SIMD needs
MGX_HD uint32_t *lane_run_word(const uint8_t *il, uint32_t max_cols, int32_t i) {
    return (uint32_t *)il + (max_cols * (LANE_SLOT_WORDS + LANE_S8_WORDS) + (uint32_t)i) * LANE_WAVE;
}
// the lane's private slice behind the interleaved part
MGX_HD uint8_t *lane_rest(const LaneParams &LP, const uint8_t *il, int32_t lane) {
    return (uint8_t *)il + (LP.wave_stride - (uint64_t)LANE_WAVE * LP.rest_stride) + (uint64_t)lane * (LP.rest_stride - 4);
}

// update_seed_filter (:100-156) for column c of a node whose columns all belong to one replay (the first node of a seed): merges
// the column's S row into the node's vector cv[query position] (range state in vr[0] = start, vr[1] = length; length 0 = none yet)
// exactly as chain_step does — a disjoint range is taken as it is (the gap reads ninf), an overlapping one cell by cell: a cell
// counts as improved when S > old * rel_score_cutoff — and returns the converged score (ninf: nothing improved).
MGX_DEV int32_t lane_merge_column(const uint8_t *il, uint32_t max_cols, int32_t *cv, uint32_t *vr, int32_t c, int32_t start, double rel) {
    const int32_t base = (int32_t)gld(lane_slot_word(il, c, 9));
    const uint32_t geom = gld(lane_slot_word(il, c, 10));
    const int32_t begin = (int32_t)(geom & 0xFFFF), size = (int32_t)((geom >> 16) & 0xFF), org = begin & ~3;
    const int32_t skip = begin ? 0 : 1;
    const int32_t cn = size - skip, query_start = start + begin - (begin ? 1 : 0);
    auto cell_S = [&](int32_t j) -> int32_t {                       // cell j of the column (window position begin + j)
        const int32_t d = (int32_t)(int8_t)gld(lane_s8_byte(il, max_cols, c, begin + j - org));
        return d == -128 ? NINF : base + d;
    };
    const int32_t vstart = (int32_t)gld(vr), vlen = (int32_t)gld(vr + 1);
    int32_t converged = NINF;
    if (vlen == 0) {
        for (int32_t j = skip; j < size; ++j) { const int32_t sv = cell_S(j); gst(cv + start + begin + j - 1, sv); converged = imax(converged, sv); }
        gst(vr, (uint32_t)query_start); gst(vr + 1, (uint32_t)cn);
        return converged;
    }
    const int32_t vlast = vstart + vlen - 1;
    if (query_start + cn <= vstart || query_start >= vstart + vlen) {
        for (int32_t j = skip; j < size; ++j) { const int32_t sv = cell_S(j); gst(cv + start + begin + j - 1, sv); converged = imax(converged, sv); }
        if (query_start + cn <= vstart) { for (int32_t p = query_start + cn; p < vstart; ++p) gst(cv + p, NINF); }
        else { for (int32_t p = vstart + vlen; p < query_start; ++p) gst(cv + p, NINF); }
    } else {
        for (int32_t j = skip; j < size; ++j) {
            const int32_t pos = start + begin + j - 1;
            const bool old = pos >= vstart && pos <= vlast;
            const int32_t sv = cell_S(j);
            const int32_t vv = old ? gld(cv + pos) : NINF;
            const bool up = (double)sv > (double)vv * rel;
            const int32_t nv = up ? imax(vv, sv) : NINF;
            if (up || !old) gst(cv + pos, nv);
            if (up) converged = imax(converged, nv);
        }
    }
    const int32_t nstart = imin(vstart, query_start), nend = imax(vstart + vlen, query_start + cn);
    gst(vr, (uint32_t)nstart); gst(vr + 1, (uint32_t)(nend - nstart));
    return converged;
}

// the lane's record in its scratch slice (lane_read's arec): words 26 .. 31 are counters that outlive a read — columns, rank-type
// lines, select-type lines, reads finished, extensions, capacity statuses
MGX_DEV uint32_t *lane_record(const LaneParams &LP, uint8_t *scratch, int32_t lane) {
    return (uint32_t *)(lane_rest(LP, scratch, lane) + (uint64_t)LP.hash_slots * 8) + 4 * LFW + 2 * LP.max_cols + LANE_MAX_RUNS;
}

// what lane_read() leaves for lane_emit(): the result record and where the alignment's pieces are
struct LaneResult {
    ReadResult rr;
    int32_t have_aln, mode;              // mode: where lane_emit() finds the alignment (LANE_EMIT_*)
    int32_t score, offset, clip, end_clip, n_runs, j_hi, n_nodes, n_seq, trim, strand;      // j_hi: last column of the path; trim: nodes trim_offset dropped
    uint32_t words;                      // words of the output stream the alignment takes
    uint32_t label;                      // label-aware: the alignment's one label
};

// One read.  `item`: its position in the launch (tags the node table).  scratch: the wavefront's LaneParams::scratch from this
// lane's word on (wave base + 4 * chip.lane).
// Returns LR_DONE (R filled: lane_emit() writes results[read] and the output stream) or LR_BAIL (nothing to write).
#define b_score LANE_CI(CD_B_SCORE)
#define b_nod LANE_CI(CD_B_NOD)
#define b_i LANE_CI(CD_B_I)
#define b_pos LANE_CI(CD_B_POS)
#define t_score LANE_CI(CD_T_SCORE)
#define t_nod LANE_CI(CD_T_NOD)
#define t_pos LANE_CI(CD_T_POS)
#define fa_alive LANE_CI(CD_FA_ALIVE)
#define fa_conv LANE_CI(CD_FA_CONV)
#define fa_max_val LANE_CI(CD_FA_MAX)
#define d_score(t) LANE_CI(CD_D_SCORE0 + (t))
#define d_max(t) LANE_CI(CD_D_MAX0 + (t))
#define kid_node0 LANE_CU(CD_KID_N0)
#define kid_codes LANE_CU(CD_KID_C)                  /* child 0's character in the low byte, child 1's above it */
#define kid_node1 LANE_CU(CD_KID_N1)
#define kid_rank0 LANE_CU(CD_KID_R0)
#define kid_rank1 LANE_CU(CD_KID_R1)
#define table_cap LANE_CU(CD_TABLE_CAP)
#define tsb_lo LANE_CU(CD_TSB_LO)
#define tsb_hi LANE_CU(CD_TSB_HI)
#define table_size_bytes() (((uint64_t)tsb_hi << 32) | tsb_lo)
#define cell_top LANE_CU(CD_CELL_TOP)
#define cols_done LANE_CI(CD_NCOLS)
#define f_node LANE_CU(CD_F_NODE)
#define f_idx LANE_CI(CD_F_IDX)
#define f_max_val LANE_CI(CD_F_MAX)
// pass: 0 = a new read; 1 = the backward pass of the read whose call with pass 0 returned LR_AGAIN (same lane, nothing in between).
MGX_DEV int lane_read(const LaneParams &LP, uint64_t read, const uint32_t item, const int pass, uint8_t *scratch, const LaneChip &chip,
                      LaneCounters &ctr, LaneResult &R) {
    const AlignParams &P = LP.P;
    const DevConfig &cfg = P.cfg;
    const DevLimits &lim = P.lim;
    const int32_t k = (int32_t)P.g.k;
    const int32_t go = cfg.gap_open, ge = cfg.gap_ext;
    const int32_t m = LP.self_score;
#if MGX_LANE_TIMERS
    ctr.t0 = cycle_clock();
#endif
    const bool have_rc = cfg.fwd_and_rc != 0;
    // The arrays of the scratch are addressed from its base WHERE THEY ARE USED: left to itself the compiler hoists every one of
    // these address computations out of the column loop and keeps a dozen 64-bit pointers in registers across it.
    auto sbase = [&]() -> uint8_t * { uint8_t *b = scratch; LANE_OPAQUE_PTR(b); return b; };
    auto slots = [&]() -> uint8_t * { return sbase(); };                             // (lane_slot_word / lane_s8_word)
    auto rest = [&]() -> uint8_t * { return lane_rest(LP, sbase(), chip.lane); };
    auto htab = [&]() -> uint64_t * { return (uint64_t *)rest(); };
    auto save_p = [&]() -> uint32_t * { return (uint32_t *)(rest() + (uint64_t)LP.hash_slots * 8); };
    auto save_a = [&]() -> uint32_t * { return save_p() + 2 * LFW; };      // (unused since the parked child lives in a frontier slot)
    auto pa_node = [&]() -> uint32_t * { return save_p() + 4 * LFW; };            // the forward alignment's nodes and character codes,
    auto pa_code = [&]() -> uint32_t * { return pa_node() + LP.max_cols; };       // in path order (the seed of the backward pass)
    auto runs_fwd = [&]() -> uint32_t * { return pa_code() + LP.max_cols; };      // ... and its CIGAR runs
    // the alignment found so far, as the output needs it (written once per pass, read at the end: memory, not registers):
    // [0 .. 8) score, offset, clip, end clip, runs, last column, nodes, characters; [8 .. 12) the forward alignment while the
    // backward pass runs: in the aggregator?, score, clip, end clip; [16 .. 20) the read's length, its seed count, the clipping of the backward pass's seed, the root's insertion run; [26 .. 32) counters
    auto arec = [&]() -> uint32_t * { return runs_fwd() + LANE_MAX_RUNS; };
    // a column that stays behind in the frontier: its window (S, F) and what makes it the head again — slot t of LANE_MAX_DEFER
    auto dsave = [&](int t) -> uint32_t * { return arec() + 32 + (LANE_MAX_L + 8) + t * LANE_DSLOT_WORDS; };
    // ---- the read and its seeds (flat_read_begin).  What is derived from the seed header here is derived again after the
    // extension, where the later seeds and the result need it: nothing of it is live across the column loop.
    int32_t L, n;
    if (!pass) {
        const uint64_t off = gld(P.offsets + read);
        L = (int32_t)(gld(P.offsets + read + 1) - off);
        if (L > (int32_t)lim.Lmax || L > LANE_MAX_L) LANE_BAIL(1);
        const SeedHdr *hp = P.seed_hdr + read;
        if (gld(&hp->status) != ST_OK) LANE_BAIL(2);
        uint32_t nm0 = gld(&hp->num_matching[0]), nm1 = gld(&hp->num_matching[1]);
        int32_t ns0 = (int32_t)gld(&hp->n_seeds[0]), ns1 = (int32_t)gld(&hp->n_seeds[1]);
        uint32_t label = 0;
        if (P.labeled) {
            // LabeledAligner::filter_seeds (aligner_labeled.cpp:612-721) for both strands, in the one-label case: every seed's
            // first node has the row { L }; L stays if the query positions under the seeds' first k-mers reach min_exact_match
            // (else the strand loses its seeds); num_matching is recounted over the seeds left (get_num_char_matches_in_seeds,
            // alignment.hpp:100-127 with its quirk: nothing behind the first sub-k seed counts)
            if (!(P.labeled & 2u)) LANE_BAIL(30);                    // (rows of dummy nodes would need the W test)
            bool have_label = false;
            const int32_t ns0_h = ns0;                                // (strand 1's seeds follow strand 0's in the stream)
            for (int s2 = 0; s2 < 2; ++s2) {
                const int32_t n_in = s2 ? ns1 : ns0;
                if (!n_in) continue;
                if (s2 && !have_rc) continue;
                const DevSeed *sd0 = P.seed_stream + gld(&hp->off) + (s2 ? ns0_h : 0);
                const uint32_t *rn = (s2 ? P.nodes_rc : P.nodes_fwd) + gld(P.node_begin + read);
                uint64_t cov[4] = { 0, 0, 0, 0 };
                uint32_t nm = 0;
                int32_t last_end = 0;
                bool counting = true;
                for (int32_t j = 0; j < n_in; ++j) {
                    const DevSeed *sj = sd0 + j;
                    const int32_t cl = (int32_t)gld(&sj->clipping), so = (int32_t)gld(&sj->offset), len = (int32_t)gld(&sj->length);
                    const uint32_t node0 = so == 0 ? gld(rn + cl) : gld(&sj->node);
                    uint64_t h = 0;
                    if (node0 && node0 <= P.g.n && (uint64_t)node0 - 1 < P.anno_rows) h = gld(P.anno_head + ((uint64_t)node0 - 1));
                    if ((h & 0xFFFF) != 1) LANE_BAIL(30);              // no label or several: the group kernel's label sets
                    const uint32_t lbl = (uint32_t)(h >> 16);
                    if (!have_label) { label = lbl; have_label = true; } else if (lbl != label) LANE_BAIL(30);
                    const int32_t lo = cl, hi = imin(cl + k - so, L);
#pragma unroll
                    for (int wq = 0; wq < 4; ++wq) {                       // bits [lo, hi) of the 256-position indicator
                        const int32_t a = imin(imax(lo - 64 * wq, 0), 64), b = imin(imax(hi - 64 * wq, 0), 64);
                        const uint64_t below_b = b >= 64 ? ~0ull : (1ull << b) - 1ull, below_a = a >= 64 ? ~0ull : (1ull << a) - 1ull;
                        cov[wq] |= below_b & ~below_a;
                    }
                    if (counting) {
                        const int32_t q_end = cl + len;
                        if (q_end > last_end) nm += (uint32_t)((q_end - cl) - (cl < last_end ? last_end - cl : 0));
                        last_end = q_end;
                        if (so) counting = false;
                    }
                }
                const uint32_t cnt = (uint32_t)(popc64(cov[0]) + popc64(cov[1]) + popc64(cov[2]) + popc64(cov[3]));
                const bool keep = !((double)cnt < cfg.min_exact_match * (double)L);
                if (s2) { ns1 = keep ? n_in : 0; nm1 = keep ? nm : 0u; } else { ns0 = keep ? n_in : 0; nm0 = keep ? nm : 0u; }
            }
            gst(arec() + 15, (uint32_t)ns0 | ((uint32_t)ns1 << 16));
            gst(arec() + 25, nm0 | (nm1 << 16));
        }
        // align_both_directions (:738-755): the strand with more matches; the other one only if it is within rel_score_cutoff
        const int first = nm0 >= nm1 ? 0 : 1;
        const int s = have_rc ? first : 0;
        if (have_rc) {
            const uint32_t m_first = first ? nm1 : nm0, m_second = first ? nm0 : nm1;
            const int32_t n_second = first ? ns0 : ns1;
            if ((double)m_second >= (double)m_first * cfg.rel_score_cutoff && n_second > 0) LANE_BAIL(3);     // a second strand to align
        }
        n = s ? ns1 : ns0;
        // the result: which alignment (LaneResult::mode) and its scalars
        LANE_CI(CD_HMS) = (s << 3) | (int32_t)(label << 8); gst(arec() + 16, (uint32_t)L); gst(arec() + 17, (uint32_t)n);
        cols_done = 0;
        LANE_CU(CD_CTR) = 0;
    } else {
        L = (int32_t)gld(arec() + 16); n = (int32_t)gld(arec() + 17);
    }
    const int32_t n_extensions = n > 0 ? pass + 1 : 0;
    if (n > 0) {
        if (n > LANE_MAX_SEEDS) LANE_BAIL(4);
        auto qcode = [&](int32_t qi) -> uint32_t { return (uint32_t)(chip.qw[(qi >> 5) * chip.qstride] >> (2 * (qi & 31))) & 3u; };
        // the strand of a pass: 2-bit packed by k_pack_reads; any character outside ACGT (psum_lin == 0) is not for this kernel
        auto load_strand = [&](int strand) -> bool {
            const uint64_t off = gld(P.offsets + read);
            const uint64_t wb = packed_word_begin(off, read);
            const int32_t nw = (L + 31) >> 5;
            uint32_t any_inv = 0;
            for (int32_t j = 0; j < LANE_QWORDS; ++j) {
                uint64_t v = 0;
                if (j < nw) { v = gld(LP.pk[strand] + wb + j); any_inv |= gld(LP.iv[strand] + wb + j); }
                chip.qw[j * chip.qstride] = v;
            }
            return any_inv == 0;
        };
        int32_t clipping;
        if (!pass) {
            const int s = LANE_STRAND();
            if (!load_strand(s)) LANE_BAIL(5);
            // ---- seed 0 (seedref_from_seed) ----
            const SeedHdr *hp = P.seed_hdr + read;
            const DevSeed *s0 = P.seed_stream + gld(&hp->off) + (s ? (int32_t)gld(&hp->n_seeds[0]) : 0);
            const uint32_t *rnodes = (s ? P.nodes_rc : P.nodes_fwd) + gld(P.node_begin + read);
            clipping = (int32_t)gld(&s0->clipping);
            const int32_t seed_len = (int32_t)gld(&s0->length), seed_off = (int32_t)gld(&s0->offset);
            const uint32_t node0 = seed_off == 0 ? gld(rnodes + clipping) : gld(&s0->node);
            if (node0 == 0) LANE_BAIL(6);
            LANE_CI(CD_SEED_LEN) = seed_len; LANE_CI(CD_SEED_OFF) = seed_off; LANE_CU(CD_NODE0) = node0;
            // Which columns will be asked for their S row: those of the later seeds' last nodes (check_seed below), as a 32-bit
            // filter over the node ids — a column whose node misses the filter keeps its slot only (one line written per column
            // instead of two; the trace needs flags and links, not scores).
            uint32_t flt = 0;
            if (n - 1 > LANE_S8_FILTER_SEEDS) flt = ~0u;
            else for (int32_t t = 1; t < n; ++t) {
                const DevSeed *sj = s0 + t;
                const int32_t cl = (int32_t)gld(&sj->clipping), so = (int32_t)gld(&sj->offset), nn = (int32_t)gld(&sj->n_nodes);
                const uint32_t ln = so == 0 ? gld(rnodes + cl + nn - 1) : gld(&sj->node);
                flt |= 1u << (lane_hash(ln, 31u));
            }
            LANE_CU(CD_S8_FILTER) = flt;
        } else {
            clipping = (int32_t)gld(arec() + 18);            // (the seed of the backward pass and its strand: set up when pass 0 ended)
        }
#define c_seed_len LANE_CI(CD_SEED_LEN)
#define c_seed_off LANE_CI(CD_SEED_OFF)
#define c_node0 LANE_CU(CD_NODE0)
#define c_fwd_n_nodes LANE_CI(CD_FWD_N_NODES)
#define c_fwd_n_seq LANE_CI(CD_FWD_N_SEQ)
        const int32_t xdrop = cfg.xdrop;
        const uint32_t tag = ((LP.tag_seed + item) * 0x9E3779B1u >> 12) | 1u;              // 20 bits, never 0
        const uint32_t hmask = LP.hash_slots - 1;
        // the forward alignment while the backward pass runs (pass 1), for the aggregator's choice at the end
        // Pass 0: the seed, forward on the graph.  Pass 1 (aln_both :683-736, when the forward alignment starts inside the query):
        // its reversal as the seed (force_fixed_seed), on the other strand of the query and the reverse-complement view of the
        // graph.  One call of this function per pass (the caller keeps the lane on its read), so that the column pass exists once
        // in the kernel and nothing of a pass is live across the other's column loop.
        do {
        // ---- extend_begin (:412-470): set_seed, the root column ----
        int32_t xdrop_cutoff = imax(-xdrop, NINF + 1);
        const int32_t start = clipping, window_size = L - start, qlen = L;
        const int32_t last_pos = window_size;
        // extend() / backtrack's min_path_score: aln_both passes max(0, min_cell_score) forward, get_min_path_score backward
        // (the aggregator holds the forward alignment by then); align_core without a reverse strand: get_min_path_score
        int32_t min_start_score = have_rc ? imax(0, cfg.min_cell_score) : imax(0, cfg.min_path_score);
        if (pass) {
            const int32_t fw_added = (int32_t)gld(arec() + 8), fw_score = (int32_t)gld(arec() + 9);
            // get_min_path_score (dbg_aligner.cpp:277-282): the aggregator's global cut-off — label-aware its cut-off for the seed's
            // labels (get_score_cutoff, aligner_aggregator.hpp:152-166): the full queue of L holds the forward alignment, whose score
            // is the label's cut-off and lies above the global one
            int32_t gcut = !fw_added ? NINF : (fw_score > 0 ? (int32_t)((double)fw_score * cfg.rel_score_cutoff) : fw_score);
            if (P.labeled && fw_added) gcut = imax(gcut, fw_score);
            min_start_score = imax(0, imax(cfg.min_path_score, gcut));
        }
        const uint32_t ptag = pass ? (((tag ^ 0x5A5A5u) & 0xFFFFFu) | 1u) : tag;
        int32_t tsize = 1;
        int32_t min_cell_score = 0, best_score = 0;
        int32_t S[LFW], F[LFW];
        int32_t f_org = 0, f_trim = 0, f_size, f_offset;
        {
            const int32_t sroot = (cfg.left_end_bonus && !clipping) ? cfg.left_end_bonus : 0;
            int32_t root_pushes = 0;
            const int32_t root_ins = imax(sroot + go, NINF + ge);
            if (1 < window_size + 1 && root_ins >= xdrop_cutoff) {
                int32_t n_push = 1;
                const int32_t room = window_size + 1 - 2;
                if (ge == 0) n_push += room;
                else { int32_t v = root_ins; while (n_push - 1 < room && v + ge >= xdrop_cutoff) { v += ge; ++n_push; } }
                root_pushes = n_push;
            }
            const int32_t root_size = 1 + root_pushes;
            gst(arec() + 19, (uint32_t)root_pushes);
            if ((uint64_t)rec_words((uint32_t)root_size + 8) > lim.cell_words) LANE_BAIL(8);
            // (the table of a strand's extender keeps its capacity between extensions; a read's two passes use two extenders)
            table_cap = 1;
            const uint64_t tsb0 = (uint64_t)136 * 1 + (uint64_t)(3 * ref_capacity(1, (uint32_t)root_pushes)) * 4;
            tsb_lo = (uint32_t)tsb0; tsb_hi = (uint32_t)(tsb0 >> 32);
            // the root leaves the frontier and enters the chain window (extend_step: fast_fits + fast_load)
            if (root_size + 3 > LFW) LANE_BAIL(9);
#pragma unroll
            for (int x = 0; x < LFW; ++x) {
                S[x] = x == 0 ? sroot : (x <= root_pushes ? root_ins + (x - 1) * ge : NINF);
                F[x] = NINF;
            }
            f_size = root_size;
            f_offset = c_seed_off - 1;
            f_max_val = sroot; f_idx = 0; f_node = c_node0;
            cell_top = rec_words((uint32_t)((root_size + 5 + 3) & ~3));
        }
        // backtrack start cells, collected while the columns are in registers (bt_begin :815-867)
        b_score = INT32_MIN; b_nod = INT32_MIN; b_i = 0; b_pos = 0;                 // the best (score, -off_diag, -i, pos)
        t_score = INT32_MIN; t_nod = 0; t_pos = 0;                                  // the head column's start cell if it were a tip
        bool tie_bad = false;                               // tie mode: a start cell that could compete with the best one
        auto cand = [&](int32_t sc, int32_t nod, int32_t i, int32_t pos) {
            const int32_t bs = b_score, bn = b_nod, bi = b_i, bp = b_pos;
            if (LANE_TIE_MODE() && (bs == INT32_MIN || sc >= bs)) tie_bad = true;
            const bool better = sc != bs ? sc > bs : (nod != bn ? nod > bn : (-i != -bi ? -i > -bi : pos > bp));
            if (bs == INT32_MIN || better) { b_score = sc; b_nod = nod; b_i = i; b_pos = pos; }
        };
        LaneProfPacked prof;
        prof.qw = chip.qw; prof.qstride = chip.qstride; prof.qlen = qlen; prof.rowp = 0; prof.w = 0;
        // ---- the extension (extend_step): chain steps along the head column, a general step where the graph forks ----
        // The frontier (:477-504) is the head column — the one whose window is in registers — plus the few columns that stayed
        // behind: the other child of a fork, kept as (converged score, column maximum).  A column that stays behind is only ever
        // popped to die here: when its turn comes, its cells must all lie below the x-drop cut-off the head has raised since
        // (band test :549-560) — a read where one would go on leaves this kernel.
#pragma unroll
        for (int t = 0; t < LANE_MAX_DEFER; ++t) { d_score(t) = INT32_MIN; d_max(t) = NINF; }
        // the children of the head that are being computed (call_outgoing :330-387): one, or the two of a fork
        kid_node0 = 0; kid_codes = 0; kid_node1 = 0; kid_rank0 = 0; kid_rank1 = 0;
        int n_kids = 0, kid = 0;
        // the first child of a fork, parked while the second is computed (its window in the lane's scratch)
        fa_alive = 0; fa_conv = 0; fa_max_val = 0;
        // (one dword per store, on purpose: merged into 16-byte stores they want four consecutive registers each, and pinning
        // the window to such tuples made the allocator spill 600+ registers around the column pass; forks are rare)
        auto win_save = [&](uint32_t *dst) {
#pragma unroll
            for (int x = 0; x < LFW; ++x) { lane_store_single(dst + x, (uint32_t)S[x]); lane_store_single(dst + LFW + x, (uint32_t)F[x]); }
        };
        auto win_load = [&](const uint32_t *src) {
#pragma unroll
            for (int x = 0; x < LFW; ++x) { S[x] = (int32_t)gld(src + x); F[x] = (int32_t)gld(src + LFW + x); }
        };
        // One loop iteration = one column: the (next) child of the head.  The window S / F is written at exactly two places — the
        // reload at the top (the parent again for the second child of a fork; the parked first child when it becomes the head)
        // and lane_column() — and the loop has one back edge: with the window assigned on several paths the register allocator
        // kept copies of it and spilled hundreds of registers.
        int32_t hd_begin = 0, hd_prev_end = 0;
#define replay_top LANE_CI(CD_REPLAY_TOP)
        replay_top = -1; LANE_SET_REPLAY_MATCHING(1); LANE_SET_TIE_MODE(0);
        bool reload = false, ext_over = false;
        int reload_slot = -1;                               // -1: the parent of a fork again; else the frontier slot whose column is the head now
        LANE_T(0);
        while (!ext_over) {
            LANE_T(4);
            if (reload) { win_load(reload_slot < 0 ? save_p() : dsave(reload_slot)); reload = false; }
            bool head_dead = false;
            if (n_kids == 0) {
                // a new head: early cut-offs when off the optimal path (:521-547), its band, its children
                bool stop_all = false;
                if (f_max_val < best_score) {
                    if ((double)tsize / (double)window_size >= cfg.max_nodes_per_seq_char) stop_all = true;
                    else if ((double)table_size_bytes() / 1000000.0 > cfg.max_ram_per_alignment) stop_all = true;
                }
                if (stop_all && LANE_TIE_MODE()) LANE_BAIL(27);                        // (which columns exist by now depends on the order)
                if (stop_all) {
                    ext_over = true;                                                   // (the frontier is dropped with it)
                } else {
                    // (the band is the head's, computed once with the cut-off at that time: the children of a fork share it, :549-560)
                    LaneColumnIn bin;
                    bin.p_org = f_org; bin.p_trim = f_trim; bin.p_size = f_size; bin.xdrop_cutoff = xdrop_cutoff;
                    lane_band(bin, S, hd_begin, hd_prev_end);
                    head_dead = hd_prev_end <= hd_begin;
                    if (!head_dead) {
                        const int32_t no = f_offset + 1, sp = no - c_seed_off;
                        const bool use_seed = sp >= 0 && sp < c_seed_len && (no < k || pass);
                        // (the replayed column's character and node are REQUESTED here and taken below, behind the enumeration of
                        // the wave-mates' heads: the branches of a wavefront run one after the other, and a load that is consumed
                        // inside its branch keeps the others waiting for its round trip)
                        uint32_t rp_code = 0, rp_node = 0;
                        if (use_seed && pass) {
                            // force_fixed_seed (:344-372): the reversed forward alignment, node by node — its spelling is
                            // the complement of the path's, read backwards (A <-> T, C <-> G: code 5 - c)
                            const int32_t fn = c_fwd_n_nodes, fq = c_fwd_n_seq;
                            rp_code = gld(pa_code() + (fq - 1 - sp));
                            rp_node = gld(pa_node() + (fn - 1 - imax(0, no - k + 1)));
                        }
                        if (!use_seed) {
                            uint32_t kn0 = 0, kc0 = 0, kn1 = 0, kc1 = 0, kr0 = 0, kr1 = 0;
                            // (the head is a child of the previous enumeration, or that one's rank says nothing about it)
                            const uint32_t hr = f_node == kid_node0 ? kid_rank0 : (f_node == kid_node1 ? kid_rank1 : 0u);
                            const int nc = pass ? lane_parents(P.g, f_node, kn0, kc0, kn1, kc1, chip)
                                                : lane_children(P.g, f_node, hr, kn0, kc0, kr0, kn1, kc1, kr1, chip);
                            if (nc > 2) LANE_BAIL(10);                                 // more than two children
                            int nc_l = nc;
                            if (P.labeled && nc == 2) {
                                // LabeledExtender::call_outgoing at a fork (aligner_labeled.cpp:232-263): the children that share
                                // a label with this column — here: whose row is { L } — are the children; a row with several labels
                                // would make a new set (the group kernel's business)
                                const uint32_t lbl = LANE_LABEL();
                                auto row_of = [&](uint32_t node) -> uint64_t {
                                    return (node && node <= P.g.n && (uint64_t)node - 1 < P.anno_rows) ? gld(P.anno_head + ((uint64_t)node - 1)) : 0ull;
                                };
                                const uint64_t h0 = row_of(kn0), h1 = row_of(kn1);
                                if ((h0 & 0xFFFF) > 1 || (h1 & 0xFFFF) > 1) LANE_BAIL(30);
                                const bool k0 = (h0 & 0xFFFF) == 1 && (uint32_t)(h0 >> 16) == lbl, k1 = (h1 & 0xFFFF) == 1 && (uint32_t)(h1 >> 16) == lbl;
                                if (!k0) { kn0 = kn1; kc0 = kc1; kr0 = kr1; }
                                nc_l = (k0 ? 1 : 0) + (k1 ? 1 : 0);
                                if (nc_l < 2) { kn1 = 0; kc1 = 0; kr1 = 0; }
                            }
                            kid_node0 = kn0; kid_codes = (kc0 & 0xFFu) | (kc1 << 8); kid_node1 = kn1; kid_rank0 = kr0; kid_rank1 = kr1;
                            if (nc_l == 0) {                                           // a tip: its start cell counts after all
                                if (t_score != INT32_MIN) cand(t_score, t_nod, f_idx, t_pos);
                                if (tie_bad) LANE_BAIL(27);
                                head_dead = true;
                            }
                            n_kids = nc_l;
                        } else {
                            if (!pass) {
                                kid_node0 = c_node0; kid_rank0 = 0; kid_codes = (kid_codes & ~0xFFu) | (qcode(clipping + sp) + 1);   // the seed's first node, its spelling
                            } else {
                                kid_codes = (kid_codes & ~0xFFu) | ((5u - rp_code) & 0xFFu);
                                kid_node0 = rp_node; kid_rank0 = 0;
                            }
                            n_kids = 1;
                        }
                        kid = 0;
                        fa_alive = 0;
                    }
                }
            }
            LANE_T(1);
            // the child computed in this iteration (if any)
            int32_t c_alive = 0, c_conv = NINF, c_org = 0, c_trim = 0, c_size = 0, c_max_val = 0, c_idx = 0;
            int32_t c_t_score = INT32_MIN, c_t_nod = 0, c_t_pos = 0;
            uint32_t c_node = 0;
            const int32_t next_offset = f_offset + 1;
            const bool compute = !ext_over && !head_dead && n_kids > 0;
            const bool forked = n_kids == 2;
            if (compute) {
                if (forked && kid == 0) win_save(save_p());                            // the parent is needed twice
                const uint32_t next = kid == 0 ? kid_node0 : kid_node1, ccode = kid == 0 ? (kid_codes & 0xFFu) : (kid_codes >> 8);
                const int32_t seed_off = c_seed_off;
                const int32_t seed_pos = next_offset - seed_off;
                const bool in_seed = seed_pos >= 0 && seed_pos < c_seed_len;
                LaneColumnIn in;
                in.p_org = f_org; in.p_trim = f_trim; in.p_size = f_size;
                in.xdrop_cutoff = xdrop_cutoff; in.start = start; in.window_size = window_size; in.qlen = qlen; in.go = go; in.ge = ge;
                const int32_t begin = hd_begin;
                in.band_given = 1; in.band_begin = hd_begin; in.band_prev_end = hd_prev_end;
                if (next == 0) LANE_BAIL(11);
                if (tsize >= (int32_t)LP.max_cols - 1 || tsize >= (int32_t)lim.max_columns - 1) LANE_BAIL(12);
                // (the cell arena of the group kernel: its capacity test, with room for every record a column that stays behind
                // could own there — a read near that limit is the group kernel's to report)
                if ((uint64_t)cell_top + (LANE_MAX_DEFER + 1) * rec_words((uint32_t)LFW) + rec_words((uint32_t)(window_size + 1 - begin + 8)) > lim.cell_words) LANE_BAIL(13);
                // The node table: first visits only (a node seen before would merge convergence vectors, update_seed_filter
                // :100-156).  The replay columns all carry the seed's first node and DO merge — but all the chain needs from the
                // merge is whether some cell improved on the node's vector (converged != ninf), and a replay column's diagonal cell
                // always does: the replayed characters are the query's own, so that cell scores sroot + (t + 1) m, more than any
                // earlier column (fewer graph characters, hence at most sroot + (t' + 1) m) left at its query position.  (Needs
                // m > 0 >= gaps, mismatches <= m, sroot >= 0, 0 <= rel_score_cutoff <= 1: checked before the kernel is launched.)
                // The vector itself is never needed: a later seed ending in that node, or the graph leading back to it, bails.
                const bool replay = in_seed && next_offset < k;      // (columns of the seed's first node)
#if MGX_LANE_PROBE & 1
                const bool probe = false;
#else
                const bool probe = !replay || f_offset == seed_off - 1;
#endif
                uint32_t hs = lane_hash(next, hmask);
                uint64_t he = probe ? gld(htab() + hs) : 0;
                // label-aware: the row of the column's node (flush() :81-137 intersects it with the parent's set when the table is
                // flushed — at the next fork, at the latest before backtracking: the set must stay { L })
                uint64_t hrow = 0;
                if (P.labeled && (uint64_t)next - 1 < P.anno_rows) hrow = gld(P.anno_head + ((uint64_t)next - 1));
                in.next_offset = next_offset; in.score = 0; in.in_seed = in_seed;
                in.best_score = best_score; in.min_cell_score = min_cell_score; in.rel_cutoff = cfg.rel_score_cutoff;
                in.partial_sum_offset = 0; in.psum_lin = m; in.psum = nullptr; in.seed_off = seed_off; in.q = nullptr; in.row = nullptr;
                prof.rowp = ccode == 1 ? LP.t4[0] : ccode == 2 ? LP.t4[1] : ccode == 3 ? LP.t4[2] : LP.t4[3];
                LaneColumnOut out;
                LANE_T(3);
                const int rc = lane_column(in, S, F, out, prof);
                LANE_T(2);
                if (rc == LC_FALLBACK) LANE_BAIL(14);
                if (P.labeled && rc != LC_POP && !((hrow & 0xFFFF) == 1 && (uint32_t)(hrow >> 16) == LANE_LABEL())) LANE_BAIL(30);
                const uint32_t table_cap_before = table_cap;
                if ((uint32_t)tsize == table_cap) table_cap = imax<uint32_t>(1u, 2 * table_cap);
                ++cols_done;
                min_cell_score = out.min_cell_score;
                if (rc != LC_POP) {                                                    // (else pop(table.size() - 1), :646-653)
                    {
                        const uint64_t tsb = table_size_bytes() + (uint64_t)136 * (table_cap - table_cap_before)
                                             + (uint64_t)(3 * ref_capacity((uint32_t)out.size0, (uint32_t)out.pushes)) * 4;
                        tsb_lo = (uint32_t)tsb; tsb_hi = (uint32_t)(tsb >> 32);
                    }
                    const int32_t max_val = out.max_val;
                    if ((int32_t)((uint32_t)max_val - (uint32_t)xdrop_cutoff) > xdrop) xdrop_cutoff = max_val - xdrop;
                    if (LANE_TIE_MODE() && max_val >= best_score) LANE_BAIL(27);        // (tie mode: the best score stands)
                    best_score = imax(best_score, max_val);
                    const int32_t my_idx = tsize;
                    if (probe) {
                        for (;;) {
                            if ((uint32_t)(he >> 44) != ptag) break;                   // free (or left by another read / pass)
                            if ((uint32_t)he == next) LANE_BAIL(15);                 // seen before
                            hs = (hs + 1) & hmask;
                            he = gld(htab() + hs);
                        }
                        gst(htab() + hs, (uint64_t)next | ((uint64_t)ptag << 44) | ((uint64_t)(uint32_t)my_idx << 32));
                    }
                    LANE_T(7);                                                         // (timers: the node table's share of "commit")
                    // (replay columns: see above; the column's own maximum stands in for the merged score, ninf neither way)
                    int32_t converged = out.converged;
                    const int32_t size = out.size, org = out.org;
                    // commit: the slot (flags, node, base, geometry, parent) and the S row
                    const int32_t base = max_val == NINF ? 0 : max_val;
                    {
                        // (the S row: where something may read it — a replay column of the backward pass (merged below), a later
                        // seed's last node — and else not at all; the trace's last cell, if it is in such a column, bails)
                        const bool keep_row = (pass && replay) || ((LANE_CU(CD_S8_FILTER) >> lane_hash(next, 31u)) & 1u);
                        uint32_t *sl = lane_slot_word(slots(), my_idx, 0);               // (word w: sl + w * LANE_WAVE)
#if MGX_LANE_PROBE & 2
                        if (my_idx > 100000)
#endif
                        {
#pragma unroll
                            for (int b = 0; b < 8; ++b) gst(sl + b * LANE_WAVE, out.fw[b]);
                            gst(sl + 8 * LANE_WAVE, next);
                            gst(sl + 9 * LANE_WAVE, (uint32_t)base);
                            gst(sl + 10 * LANE_WAVE, (uint32_t)begin | ((uint32_t)size << 16) | (ccode << 24) | (keep_row ? LANE_GEOM_ROW : 0u));
                            gst(sl + 11 * LANE_WAVE, (uint32_t)f_idx | ((uint32_t)next_offset << 16));
                        }
                        if (keep_row) {
                            uint32_t sw[8];
                            // Every real cell of a column lies at or above the cut-off its pass ran with (update_column keeps
                            // S > cutoff - 1, the scalar tail and the insertion run test >= cutoff), the cut-off is the best score
                            // so far minus xdrop, and a column's maximum exceeds its parent's — hence the best score so far — by at
                            // most one match: d >= -(m + xdrop).  Where that is within a byte (the CLI's scores: 2 + 27) no cell
                            // can be too wide, ninf is what clamps to -128, and a cell costs a max and a subtraction; else cell by cell.
                            if (xdrop >= 0 && m + xdrop <= 127) {
                                const int32_t floor_d = base - 128;
#pragma unroll
                                for (int b = 0; b < LFW / 4; ++b) {
                                    uint32_t v = 0;
#pragma unroll
                                    for (int q4 = 0; q4 < 4; ++q4) {
                                        const int32_t sv = S[4 * b + q4];
#if !defined(__HIPCC__)                 /* the host model checks the argument above on every cell it packs */
                                        if (sv != NINF && (int64_t)sv - (int64_t)base < -127) { fprintf(stderr, "lane_read: an S row's cell below its byte\n"); abort(); }
#endif
                                        v |= ((uint32_t)(imax(sv, floor_d) - base) & 0xFFu) << (8 * q4);
                                    }
                                    sw[b] = v;
                                }
                            } else {
                            bool wide = false;
#pragma unroll
                            for (int b = 0; b < LFW / 4; ++b) {
                                uint32_t v = 0;
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) {
                                    const int32_t sv = S[4 * b + q4];
                                    const int32_t d = (int32_t)((uint32_t)sv - (uint32_t)base);      // (sv may be ninf: its d is not used, but must not overflow)
                                    wide |= sv != NINF && d < -127;
                                    v |= (sv == NINF ? 0x80u : ((uint32_t)d & 0xFFu)) << (8 * q4);
                                }
                                sw[b] = v;
                            }
                            if (wide) LANE_BAIL(17);
                            }
                            uint32_t *sr = lane_s8_word(slots(), LP.max_cols, my_idx, 0);
#pragma unroll
                            for (int b = 0; b < 8; ++b) gst(sr + b * LANE_WAVE, sw[b]);
                        }
                    }
                    tsize = my_idx + 1;
                    if (pass && replay) {
                        // The backward pass replays the PATH's characters, which may differ from the query's.  While they have
                        // all matched, the argument above holds.  After a mismatch this one may: a cell with a score at a
                        // window position above every earlier column of the node lies outside the node's vector (the ranges of
                        // these columns are contiguous), reads ninf there and improves on it — converged != ninf.  With
                        // neither, the merge itself decides: the node's vector is built from the S rows of its columns so far
                        // (lane_merge_column, update_seed_filter :100-156 as chain_step restates it) and kept from there on.
                        int32_t top = -1;
#pragma unroll
                        for (int x = 0; x < LFW; ++x) top = S[x] != NINF ? org + x : top;
                        if (!(top >= begin && top < begin + size)) top = -1;
                        const int32_t ap = clipping + seed_pos + 1;                     // the query character under the diagonal cell
                        const bool same = ap >= 1 && ap <= L && qcode(ap - 1) + 1 == ccode;
                        if (probe) { LANE_SET_REPLAY_MATCHING(1); LANE_SET_REPLAY_EXACT(0); }
                        if (!same) LANE_SET_REPLAY_MATCHING(0);
                        const bool exact_on = LANE_REPLAY_EXACT() != 0;
                        if (exact_on || (!probe && !LANE_REPLAY_MATCHING() && !(top > replay_top))) {
                            int32_t *cv = (int32_t *)(arec() + 32);
                            int32_t c0 = my_idx;
                            if (!exact_on) { c0 = 1; LANE_SET_REPLAY_EXACT(1); gst(arec() + 14, 0u); }       // (no vector yet:) the columns so far, then this one
                            for (int32_t c = c0; c <= my_idx; ++c)
                                converged = lane_merge_column(slots(), LP.max_cols, cv, arec() + 13, c, start, cfg.rel_score_cutoff);
                        }
                        replay_top = probe ? top : imax(replay_top, top);
                    }
                    // a column of the general path (the children of a fork) owns an S / F record of the cell arena there
                    if (forked) cell_top += rec_words((uint32_t)((size + 5 + 3) & ~3));
                    // start cells of this column (bt_begin :815-867)
                    if (next_offset >= imax(k, c_seed_len) - 1) {
                        const int32_t max_pos = out.max_pos;
                        {
                            const uint32_t fl = lane_flags_at(out.fw, max_pos - org);
                            if ((fl & CF_REAL) && (fl & CF_SP_REAL)) {
                                const int32_t eb = max_pos == last_pos ? cfg.right_end_bonus : 0;
                                if (base + eb >= min_start_score) {
                                    const int32_t ap = clipping + max_pos;
                                    const bool is_match = (fl & CF_MATCH) && ap >= 1 && ap <= L && qcode(ap - 1) + 1 == ccode;
                                    const int32_t nod = -iabs(max_pos - next_offset + seed_off - 1);
                                    if (is_match || max_pos == last_pos) cand(base + eb, nod, my_idx, max_pos);
                                    else { c_t_score = base + eb; c_t_nod = nod; c_t_pos = max_pos; }
                                }
                            }
                        }
                        if (size + begin == window_size + 1 && max_pos != last_pos) {
                            const uint32_t fl = lane_flags_at(out.fw, last_pos - org);
                            if ((fl & CF_REAL) && (fl & CF_SP_REAL)) {
                                const int32_t sv = lane_win_at(S, last_pos - org);
                                if (sv + cfg.right_end_bonus >= min_start_score)
                                    cand(sv + cfg.right_end_bonus, -iabs(last_pos - next_offset + seed_off - 1), my_idx, last_pos);
                            }
                        }
                    }
                    if (tie_bad) LANE_BAIL(27);
                    if (converged != NINF) {
                        c_alive = 1; c_conv = converged; c_org = org; c_trim = begin; c_size = size; c_max_val = max_val; c_idx = my_idx;
                        c_node = next;
                    }
                }
            }
            LANE_T(3);
            if (!ext_over) {
                if (compute && forked && kid == 0) {
                    // the first child of a fork waits (window in scratch) while the second is computed from the same parent
                    fa_alive = 0; fa_conv = c_conv; fa_max_val = c_max_val;
                    if (c_alive) {
                        int slot = -1;
#pragma unroll
                        for (int t = LANE_MAX_DEFER - 1; t >= 0; --t) slot = d_score(t) == INT32_MIN ? t : slot;
                        if (slot < 0) LANE_BAIL(28);
                        fa_alive = 1 + slot;
                        uint32_t *ds = dsave(slot);
                        win_save(ds);
                        gst(ds + 64, (uint32_t)c_org); gst(ds + 65, (uint32_t)c_trim); gst(ds + 66, (uint32_t)c_size);
                        gst(ds + 67, (uint32_t)next_offset); gst(ds + 68, (uint32_t)c_idx); gst(ds + 69, (uint32_t)c_t_score);
                        gst(ds + 70, (uint32_t)c_t_nod); gst(ds + 71, (uint32_t)c_t_pos); gst(ds + 72, c_node);
                    }
                    kid = 1;
                    reload = true; reload_slot = -1;
                } else {
                    bool none = true;
#pragma unroll
                    for (int t = 0; t < LANE_MAX_DEFER; ++t) none &= d_score(t) == INT32_MIN;
                    if (compute && !forked && c_alive && none) {
                        // the common case: one child, nothing else in the frontier (chain_step's chain_on): it is the next head
                        if (!((c_trim & 3) + c_size + 3 <= LFW)) LANE_BAIL(16);        // (it would stay behind with a record of its own)
                        f_org = c_org; f_trim = c_trim; f_size = c_size; f_offset = next_offset; f_max_val = c_max_val; f_node = c_node;
                        f_idx = c_idx; t_score = c_t_score; t_nod = c_t_nod; t_pos = c_t_pos;
                    } else {
                        // the frontier hands out the next head (:491-504): the best of the children just computed and the columns
                        // that stayed behind.  A child goes on as the head; a column that stayed behind must be dead by now.
                        if (!(compute && forked)) fa_alive = 0;
                        bool have_head = false, frontier_done = false;
                        // a column goes (back) into the frontier: the parked child has its slot already, the child in registers takes
                        // a free one — window, what makes it the head again, and (score, maximum) for the pops
                        auto stay_behind_c = [&](int avoid) -> bool {      // avoid: the slot whose column is about to become the head
                            int slot = -1;
#pragma unroll
                            for (int t = LANE_MAX_DEFER - 1; t >= 0; --t) slot = (d_score(t) == INT32_MIN && t + 1 != fa_alive && t != avoid) ? t : slot;
                            if (slot < 0) return false;
                            uint32_t *ds = dsave(slot);
                            win_save(ds);
                            gst(ds + 64, (uint32_t)c_org); gst(ds + 65, (uint32_t)c_trim); gst(ds + 66, (uint32_t)c_size);
                            gst(ds + 67, (uint32_t)next_offset); gst(ds + 68, (uint32_t)c_idx); gst(ds + 69, (uint32_t)c_t_score);
                            gst(ds + 70, (uint32_t)c_t_nod); gst(ds + 71, (uint32_t)c_t_pos); gst(ds + 72, c_node);
#pragma unroll
                            for (int t = 0; t < LANE_MAX_DEFER; ++t) { if (t == slot) { d_score(t) = c_conv; d_max(t) = c_max_val; } }
                            // (a chain column that stays behind gets an S / F record in the cell arena; a fork's children have theirs)
                            if (!forked) cell_top += rec_words((uint32_t)LFW);
                            return true;
                        };
                        auto stay_behind_a = [&]() {
#pragma unroll
                            for (int t = 0; t < LANE_MAX_DEFER; ++t) { if (t + 1 == fa_alive) { d_score(t) = fa_conv; d_max(t) = fa_max_val; } }
                        };
                        auto head_from_slot = [&](int slot) {
                            const uint32_t *ds = dsave(slot);
                            f_org = (int32_t)gld(ds + 64); f_trim = (int32_t)gld(ds + 65); f_size = (int32_t)gld(ds + 66);
                            f_offset = (int32_t)gld(ds + 67); f_idx = (int32_t)gld(ds + 68); t_score = (int32_t)gld(ds + 69);
                            t_nod = (int32_t)gld(ds + 70); t_pos = (int32_t)gld(ds + 71); f_node = gld(ds + 72);
                            reload = true; reload_slot = slot;                     // its window comes back at the top of the loop
                        };
                        while (!frontier_done) {
                            int32_t top = INT32_MIN;
                            int n_top = 0;
#pragma unroll
                            for (int t = 0; t < LANE_MAX_DEFER; ++t) {
                                const int32_t ds = d_score(t);
                                if (ds != INT32_MIN) { if (ds > top) { top = ds; n_top = 1; } else if (ds == top) ++n_top; }
                            }
                            if (fa_alive) { if (fa_conv > top) { top = fa_conv; n_top = 1; } else if (fa_conv == top) ++n_top; }
                            if (c_alive) { if (c_conv > top) { top = c_conv; n_top = 1; } else if (c_conv == top) ++n_top; }
                            const bool a_top = fa_alive && fa_conv == top, c_top = c_alive && c_conv == top;
                            if (top == INT32_MIN) {
                                frontier_done = true;                                  // the frontier is empty
                            } else if (a_top || c_top) {
                                bool take_c = false;
                                if (n_top > 1) {
                                    // An equal-score batch (:491-500): the reference pops all of it and processes it from the back —
                                    // the general path's business, except where the ORDER inside the batch cannot matter: every
                                    // column of the batch lies beyond the query's end (nothing but gap cells left to gain: the
                                    // children of a fork there tie by construction, and keep tying level by level until the x-drop
                                    // ends both branches), so each is extended on its own, the best score and the cut-off stand, and
                                    // only the column indices come out in another order.  The lane takes the child in its registers
                                    // and guards the rest of the pass (LANE_TIE_MODE: no new best score, no start cell that competes,
                                    // no early cut-off; the size caps are checked once more when the extension ends).
                                    const int32_t end_diag = window_size + (c_seed_off - 1);          // next_offset of the last column on the query
                                    bool post_end = c_top && next_offset > end_diag;
#pragma unroll
                                    for (int t = 0; t < LANE_MAX_DEFER; ++t)
                                        if (d_score(t) == top && !((int32_t)gld(dsave(t) + 67) > end_diag)) post_end = false;
                                    if (!post_end) LANE_BAIL(27);
                                    LANE_SET_TIE_MODE(1);
                                    take_c = true;
                                }
                                if (a_top && !take_c) {
                                    if (c_alive && !stay_behind_c(fa_alive - 1)) LANE_BAIL(28);
                                    f_max_val = fa_max_val;
                                    head_from_slot(fa_alive - 1);
                                } else {
                                    if (fa_alive) stay_behind_a();
                                    f_org = c_org; f_trim = c_trim; f_size = c_size; f_offset = next_offset; f_max_val = c_max_val;
                                    f_node = c_node; f_idx = c_idx; t_score = c_t_score; t_nod = c_t_nod; t_pos = c_t_pos;
                                }
                                if (!((f_trim & 3) + f_size + 3 <= LFW)) LANE_BAIL(16);    // (fast_fits: the general path would take it)
                                have_head = true;
                                frontier_done = true;
                            } else {
                                // a column that stayed behind is popped: cut-offs, then its band — empty: it is dead; else it is the head
                                if (n_top > 1) {
                                    // (an equal-score batch of such columns: harmless if they are all dead, else the general path)
                                    bool any_live = false;
#pragma unroll
                                    for (int t = 0; t < LANE_MAX_DEFER; ++t) any_live |= d_score(t) == top && d_max(t) >= xdrop_cutoff;
                                    if (any_live) LANE_BAIL(27);
                                }
                                int slot = -1;
#pragma unroll
                                for (int t = LANE_MAX_DEFER - 1; t >= 0; --t) slot = d_score(t) == top ? t : slot;
                                int32_t dm = NINF;
#pragma unroll
                                for (int t = 0; t < LANE_MAX_DEFER; ++t) dm = t == slot ? (int32_t)d_max(t) : dm;
                                bool stop_all = false;
                                if (dm < best_score) {
                                    if ((double)tsize / (double)window_size >= cfg.max_nodes_per_seq_char) stop_all = true;
                                    else if ((double)table_size_bytes() / 1000000.0 > cfg.max_ram_per_alignment) stop_all = true;
                                }
#pragma unroll
                                for (int t = 0; t < LANE_MAX_DEFER; ++t) { if (t == slot) d_score(t) = INT32_MIN; }
                                if (stop_all && LANE_TIE_MODE()) LANE_BAIL(27);
                                if (stop_all) {
                                    frontier_done = true;                              // (extend() drops the whole frontier)
                                } else if (dm >= xdrop_cutoff) {
                                    // it goes on: the children just computed stay behind in their turn
                                    if (fa_alive) stay_behind_a();
                                    if (c_alive && !stay_behind_c(slot)) LANE_BAIL(28);
                                    f_max_val = dm;
                                    head_from_slot(slot);
                                    if (!((f_trim & 3) + f_size + 3 <= LFW)) LANE_BAIL(16);
                                    have_head = true;
                                    frontier_done = true;
                                }
                            }
                        }
                        if (!have_head) ext_over = true;
                    }
                    n_kids = 0;
                }
            }
        }
        LANE_T(4);
        // (tie mode: the modelled size caps were never reached in any order of the tied columns — they are monotone in the table's
        // size, and this is its final one)
        if (LANE_TIE_MODE() && ((double)tsize / (double)window_size >= cfg.max_nodes_per_seq_char
                                || (double)table_size_bytes() / 1000000.0 > cfg.max_ram_per_alignment)) LANE_BAIL(27);
        // ---- after the extension: everything below is derived afresh from the read's index and the lane's cold state ----
        LANE_OPAQUE(read);
        const int32_t seed_off = c_seed_off, seed_len = c_seed_len;
        const uint32_t node0 = c_node0;
        const int32_t sroot = (cfg.left_end_bonus && !clipping) ? cfg.left_end_bonus : 0;
        const int32_t root_pushes = (int32_t)gld(arec() + 19), root_size = 1 + root_pushes;
        const int32_t root_ins = imax(sroot + go, NINF + ge);
        auto root_S = [&](int32_t pos) -> int32_t {
            return pos == 0 ? sroot : (pos >= 1 && pos <= root_pushes ? root_ins + (pos - 1) * ge : NINF);
        };
        (void)seed_len;
        if (!pass) {
        // ---- check_seed (:66-88) of the later seeds against the forward extension's convergence table: each must be dead, or
        // there are more extensions to run.  A node's entry is its first (and only) column here: the node table gives the column,
        // the column's S row the score at the seed's last query position.  (Before the backward pass reuses the tables, as
        // aln_both does.)
        const int s = LANE_STRAND();
        const SeedHdr *hp2 = P.seed_hdr + read;
        const DevSeed *seeds = P.seed_stream + gld(&hp2->off) + (s ? (int32_t)gld(&hp2->n_seeds[0]) : 0);
        const uint32_t *rnodes = (s ? P.nodes_rc : P.nodes_fwd) + gld(P.node_begin + read);
        bool node0_merged = false;
        for (int32_t t = 1; t < n; ++t) {
            const DevSeed *sj = seeds + t;
            const int32_t cl = (int32_t)gld(&sj->clipping), len = (int32_t)gld(&sj->length), so = (int32_t)gld(&sj->offset);
            const int32_t nn = (int32_t)gld(&sj->n_nodes);
            const uint32_t ln = so == 0 ? gld(rnodes + cl + nn - 1) : gld(&sj->node);
            const int32_t lpos = len + cl - 1;
            const int32_t lscore = len * m + (!cl ? cfg.left_end_bonus : 0) + (!(L - cl - len) ? cfg.right_end_bonus : 0);
            if (ln == node0) {
                // The seed's first node: its entry is the merged vector of the replay columns (update_seed_filter :100-156 over
                // the columns in index order, as the backward pass builds it: lane_merge_column) — the first columns of the
                // extension, whose S rows were kept because the filter holds this very node.
                int32_t *cv = (int32_t *)(arec() + 32);
                if (!node0_merged) {
                    gst(arec() + 13, 0u); gst(arec() + 14, 0u);
                    for (int32_t c = 1; c < tsize; ++c) {
                        if (gld(lane_slot_word(slots(), c, 8)) != node0) break;
                        if (!(gld(lane_slot_word(slots(), c, 10)) & LANE_GEOM_ROW)) LANE_BAIL(7);
                        (void)lane_merge_column(slots(), LP.max_cols, cv, arec() + 13, c, start, cfg.rel_score_cutoff);
                    }
                    node0_merged = true;
                }
                const int32_t vstart = (int32_t)gld(arec() + 13), vlen = (int32_t)gld(arec() + 14);
                if (vlen == 0 || lpos < vstart || lpos - vstart >= vlen) LANE_BAIL(19);      // outside the entry's range: the seed lives
                if ((int32_t)gld(cv + lpos) < lscore) LANE_BAIL(19);                    // the seed lives
                continue;
            }
            uint32_t hs = lane_hash(ln, hmask);
            int32_t idx = -1;
            for (;;) {
                const uint64_t he = gld(htab() + hs);
                if ((uint32_t)(he >> 44) != ptag) break;
                if ((uint32_t)he == ln) { idx = (int32_t)((he >> 32) & 0xFFFu); break; }
                hs = (hs + 1) & hmask;
            }
            if (idx < 0) LANE_BAIL(19);                                              // not in the table: the seed lives
            if (gld(lane_slot_word(slots(), idx, 8)) != ln) LANE_BAIL(19);           // (an entry another launch left behind)
            const uint32_t geom = gld(lane_slot_word(slots(), idx, 10));
            const int32_t cbegin = (int32_t)(geom & 0xFFFF), csize = (int32_t)((geom >> 16) & 0xFF);
            const int32_t skip = cbegin ? 0 : 1;
            const int32_t qs = start + cbegin - (cbegin ? 1 : 0), qn = csize - skip;
            if (lpos < qs || lpos - qs >= qn) LANE_BAIL(19);                         // outside the entry's range: the seed lives
            if (!(geom & LANE_GEOM_ROW)) LANE_BAIL(19);                              // (cannot be: the filter holds ln)
            const int32_t a = lpos - start + 1, x = a - (cbegin & ~3);
            int32_t v = NINF;
            if (a - cbegin >= 0 && a - cbegin < csize + 5 && x < LFW) {
                const int32_t d = (int32_t)(int8_t)gld(lane_s8_byte(slots(), LP.max_cols, idx, x));
                if (d != -128) v = (int32_t)gld(lane_slot_word(slots(), idx, 9)) + d;
            }
            if (v < lscore) LANE_BAIL(19);                                           // the seed lives
        }
        }
        // ---- backtrack (:869-1034): the best start cell, one trace.  Without a start cell the extension yields its seed
        // (:1030-1031): forward, a case for the group kernel; backward, the reversed forward alignment — nothing new.
        bool got = b_score != INT32_MIN;
        if (!got && !pass) LANE_BAIL(18);
        int32_t x_score = 0, x_offset = 0, x_clip = 0, x_end_clip = 0, x_n_runs = 0, x_j_hi = 0, x_n_nodes = 0, x_n_seq = 0;
        if (got) {
        const int32_t k_minus_1 = k - 1;
        const int32_t min_trace_length = k - seed_off;
        const int32_t cap = (int32_t)lim.max_path;
        const int32_t j_start = b_i;
        int32_t j = b_i, pos = b_pos;
        const int32_t score = b_score, end_pos = b_pos;
        int32_t n_runs = 0, n_trace = 0, n_seq = 0, n_path = 0;
        uint32_t cur_run = 0;
        int32_t align_offset = seed_off;
        int32_t j_stop = j;                                  // columns (j_stop, j_start] are on the path
        bool bad = false;
        // (the run at hand grows in a register; it is stored when the next one begins and when the trace ends)
        auto push_op = [&](uint32_t op, uint32_t num) {
            if (n_runs == 0 || (cur_run & 7) != op) {
                if (n_runs >= LANE_MAX_RUNS) { bad = true; return; }
                if (n_runs) gst(lane_run_word(slots(), LP.max_cols, n_runs - 1), cur_run);
                cur_run = (num << 3) | op;
                ++n_runs;
            } else {
                cur_run += num << 3;
            }
        };
        auto slot_geom = [&](int32_t jj) -> uint32_t { return gld(lane_slot_word(slots(), jj, 10)); };
        auto slot_link = [&](int32_t jj) -> uint32_t { return gld(lane_slot_word(slots(), jj, 11)); };     // parent | offset << 16
        auto slot_flags = [&](int32_t jj, uint32_t geom, int32_t p) -> uint32_t {
            const int32_t begin = (int32_t)(geom & 0xFFFF), size = (int32_t)((geom >> 16) & 0xFF);
            const int32_t jx = p - begin, x = p - (begin & ~3);
            if (!(jx >= 0 && jx < size + 5 && x < LFW)) return 0;
            return gld(lane_slot_flag(slots(), jj, x));
        };
        for (;;) {
            if (!j) break;
            const uint32_t geom = slot_geom(j);
            const uint32_t link = slot_link(j);
            const uint32_t ccode = (geom >> 24) & 7u;
            const int32_t col_offset = (int32_t)(link >> 16);
            align_offset = imin(col_offset, k_minus_1);
            const uint32_t fl = slot_flags(j, geom, pos);
            const uint32_t last_op = n_runs ? (cur_run & 7) : 99u;
            if (!(fl & CF_REAL)) {
                j_stop = j;
                j = 0;
            } else if (pos && (fl & CF_S_IS_E) && (n_runs == 0 || last_op != OP_DELETION)) {
                uint32_t lop = OP_INSERTION;
                while (lop == OP_INSERTION && !bad) {
                    push_op(lop, 1);
                    lop = (slot_flags(j, geom, pos) & CF_E_EXT) ? OP_INSERTION : OP_MATCH;
                    --pos;
                }
            } else if (pos && (fl & CF_MATCH)) {
                ++n_trace;
                const int32_t ap = clipping + pos;
                const uint32_t op = (ap >= 1 && ap <= L) ? (qcode(ap - 1) + 1 == ccode ? OP_MATCH : OP_MISMATCH) : OP_CLIPPED;
                ++n_seq;
                push_op(op, 1);
                if (col_offset >= k_minus_1) ++n_path;
                --pos;
                j = (int32_t)(link & 0xFFFFu);
                j_stop = j;
            } else if ((fl & CF_S_IS_F) && (n_runs == 0 || last_op != OP_INSERTION)) {
                uint32_t lop = OP_DELETION;
                while (lop == OP_DELETION && j && !bad) {
                    const uint32_t g2 = slot_geom(j);
                    const uint32_t l2 = slot_link(j);
                    const int32_t o2 = (int32_t)(l2 >> 16);
                    align_offset = imin(o2, k_minus_1);
                    lop = (slot_flags(j, g2, pos) & CF_F_EXT) ? OP_DELETION : OP_MATCH;
                    ++n_trace;
                    ++n_seq;
                    push_op(OP_DELETION, 1);
                    if (o2 >= k_minus_1) ++n_path;
                    j = (int32_t)(l2 & 0xFFFFu);
                    j_stop = j;
                }
            } else {
                j_stop = j;
                break;
            }
            if (bad || n_seq > cap || n_path > cap) LANE_BAIL(20);
        }
        if (bad) LANE_BAIL(21);
        if (n_runs) gst(lane_run_word(slots(), LP.max_cols, n_runs - 1), cur_run);
        if (!(n_trace >= min_trace_length && n_path)) LANE_BAIL(22);             // (the next start cell would be tried)
        {
            // the cell the trace ended in (the root's cells are known in closed form)
            int32_t cur_cell_score;
            if (j == 0) {
                cur_cell_score = root_S(pos);
                if (!(pos >= 0 && pos < root_size + 5)) cur_cell_score = NINF;
            } else {
                const uint32_t geom = slot_geom(j);
                const int32_t begin = (int32_t)(geom & 0xFFFF), size = (int32_t)((geom >> 16) & 0xFF);
                const int32_t jx = pos - begin, x = pos - (begin & ~3);
                cur_cell_score = NINF;
                if (jx >= 0 && jx < size + 5 && x < LFW) {
                    if (!(geom & LANE_GEOM_ROW)) LANE_BAIL(29);                      // a trace that ends inside a column without its S row
                    const int32_t v = (int32_t)(int8_t)gld(lane_s8_byte(slots(), LP.max_cols, j, x));
                    const int32_t cb = (int32_t)gld(lane_slot_word(slots(), j, 9));
                    if (v != -128) cur_cell_score = cb + v;
                }
            }
            const int32_t bt_best = score - cur_cell_score;                         // best_score = max(INT32_MIN, .)
            if (score - min_cell_score < bt_best) LANE_BAIL(23);                   // no alignment from this extension
            if (!(score >= min_start_score && (!pos || cur_cell_score == 0) && (pos || cur_cell_score == sroot)
                  && (cfg.allow_left_trim || !j))) LANE_BAIL(24);                 // (the next start cell would be tried)
        }
        // construct_alignment (:774-798) + trim_offset (alignment.cpp:177-190)
        x_clip = clipping + pos;
        x_end_clip = L - (clipping + end_pos);
        x_score = score; x_offset = align_offset; x_n_runs = n_runs;
        // (the path: the columns the trace left by a match or a deletion, from j_start down its parent links — n_seq of them, the
        // n_path with offset >= k - 1 carry the alignment's nodes)
        (void)j_stop;
        x_j_hi = j_start; x_n_seq = n_seq; x_n_nodes = n_path;
        if (x_offset && x_n_nodes > 1) {
            const int32_t trim = imin(x_offset, x_n_nodes - 1);
            if (trim > 0) { x_n_nodes -= trim; x_offset -= trim; }
        }
        }
        if (!pass) {
            // ---- aln_both after the forward pass (:683-736) ----
            gst(arec() + 0, (uint32_t)x_score); gst(arec() + 1, (uint32_t)x_offset); gst(arec() + 2, (uint32_t)x_clip);
            gst(arec() + 3, (uint32_t)x_end_clip); gst(arec() + 4, (uint32_t)x_n_runs); gst(arec() + 5, (uint32_t)x_j_hi);
            gst(arec() + 6, (uint32_t)x_n_nodes); gst(arec() + 7, (uint32_t)x_n_seq);
            LANE_SET_HAVE_ALN(1); LANE_SET_MODE(LANE_EMIT_SLOTS);
            if (!have_rc) break;
            if (!(x_score >= cfg.min_path_score)) LANE_SET_HAVE_ALN(0);         // get_min_path_score with an empty aggregator
            if (!(x_clip && !x_offset)) break;                        // nothing to extend backwards
            // The backward pass: the forward alignment, reversed (Alignment::reverse_complement :563-565 on the RCDBG view: nodes
            // and CIGAR reversed, spelling complemented), is the seed.  Its nodes, characters and CIGAR runs move to the lane's
            // scratch — the backward extension reuses the column slots — in path order.
            if (!x_n_nodes) LANE_BAIL(26);
            {
                int32_t jj = x_j_hi, ni = x_n_nodes - 1;
                for (int32_t xx = x_n_seq - 1; xx >= 0; --xx) {
                    const uint32_t node = gld(lane_slot_word(slots(), jj, 8)), geom = gld(lane_slot_word(slots(), jj, 10)), link = gld(lane_slot_word(slots(), jj, 11));
                    gst(pa_code() + xx, (geom >> 24) & 7u);
                    if ((int32_t)(link >> 16) >= k - 1) { if (ni >= 0) gst(pa_node() + ni, node); --ni; }
                    jj = (int32_t)(link & 0xFFFFu);
                }
                for (int32_t xx = 0; xx < x_n_runs; ++xx) gst(runs_fwd() + xx, gld(lane_run_word(slots(), LP.max_cols, xx)));
            }
            gst(arec() + 8, (uint32_t)LANE_HAVE_ALN()); gst(arec() + 9, (uint32_t)x_score); gst(arec() + 10, (uint32_t)x_clip);
            gst(arec() + 11, (uint32_t)x_end_clip);
            c_fwd_n_nodes = x_n_nodes; c_fwd_n_seq = x_n_seq;
            LANE_SET_MODE(LANE_EMIT_ARRAYS);
            // seedref_from_aln of the reversal: clipping = the alignment's end clipping, the whole spelling is the seed
            gst(arec() + 18, (uint32_t)x_end_clip);
            c_seed_len = x_n_seq; c_seed_off = 0; c_node0 = gld(pa_node() + (x_n_nodes - 1));
            if (!load_strand(1 - LANE_STRAND())) LANE_BAIL(5);
            LANE_CU(CD_S8_FILTER) = 0;                  // (the backward pass: the replay columns' rows only)
            return LR_AGAIN;
        }
        // ---- pass 1 is over (:700-722): the backward alignment, reversed again, joins the aggregator ----
        // (a backward extension without a start cell yields its seed — the forward alignment once more, which the aggregator has
        // or, below its score threshold, takes now: the latter is left to the group kernel)
        const int32_t fw_added = (int32_t)gld(arec() + 8);
        if (!got) { if (!fw_added) LANE_BAIL(31); break; }
        // (reverse_complement() refuses an alignment with an offset: dropped, like one without nodes)
        if (!x_offset && x_n_nodes) {
            bool take = true;
            if (fw_added) {
                const int32_t fw_score = (int32_t)gld(arec() + 9), fw_clip = (int32_t)gld(arec() + 10), fw_end_clip = (int32_t)gld(arec() + 11);
                const int32_t gcut = fw_score > 0 ? (int32_t)((double)fw_score * cfg.rel_score_cutoff) : fw_score;
                // add_alignment (:68-138, one alignment): below the cut-off it is dropped; else it replaces the queue's alignment
                // unless it is less (LocalAlignmentLess, alignment.hpp:337-348; an equal alignment changes nothing either way)
                const int32_t b_qlen = L - x_clip - x_end_clip, f_qlen = L - fw_clip - fw_end_clip;
                const int32_t b_clip_rev = x_end_clip;                                  // its clipping once reversed
                const bool less = fw_score != x_score ? fw_score > x_score : (b_qlen != f_qlen ? b_qlen > f_qlen : b_clip_rev > fw_clip);
                if (x_score < gcut || less) take = false;
            }
            if (take) {
                LANE_SET_HAVE_ALN(1); LANE_SET_MODE(LANE_EMIT_SLOTS_REVERSED);
                gst(arec() + 0, (uint32_t)x_score); gst(arec() + 1, 0u); gst(arec() + 2, (uint32_t)x_end_clip);
                gst(arec() + 3, (uint32_t)x_clip); gst(arec() + 4, (uint32_t)x_n_runs); gst(arec() + 5, (uint32_t)x_j_hi);
                gst(arec() + 6, (uint32_t)x_n_nodes); gst(arec() + 7, (uint32_t)x_n_seq);
            }
        }
        } while (0);
    }
    LANE_T(5);
    {
        ReadResult &rr = R.rr;
        rr.status = ST_OK; rr.n_alignments = 0; rr.score = 0; rr.offset = 0; rr.n_nodes = rr.n_cigar = rr.seq_len = 0;
        rr.orientation = 0; rr.stream_off = 0;
        const SeedHdr *hp = P.seed_hdr + read;
        rr.num_matches_fwd = gld(&hp->num_matching[0]); rr.num_matches_rc = gld(&hp->num_matching[1]);
        rr.n_seeds_fwd = (uint32_t)gld(&hp->n_seeds[0]); rr.n_seeds_rc = (uint32_t)gld(&hp->n_seeds[1]);
        if (P.labeled) {                                     // (what the label filter left)
            const uint32_t nsw = gld(arec() + 15), nmw = gld(arec() + 25);
            rr.n_seeds_fwd = nsw & 0xFFFFu; rr.n_seeds_rc = nsw >> 16; rr.num_matches_fwd = nmw & 0xFFFFu; rr.num_matches_rc = nmw >> 16;
        }
        rr.n_extensions = (uint32_t)n_extensions; rr.n_columns = n > 0 ? (uint32_t)cols_done : 0u;
    }
    // the lane's counters (kept in its scratch, summed when the kernel ends): columns, block lines
    gst(arec() + 26, gld(arec() + 26) + (n > 0 ? (uint32_t)cols_done : 0u));
    gst(arec() + 27, gld(arec() + 27) + (LANE_CU(CD_CTR) & 0xFFFFu));
    gst(arec() + 28, gld(arec() + 28) + (LANE_CU(CD_CTR) >> 16));
    {
        const int32_t have_aln = n > 0 ? LANE_HAVE_ALN() : 0;
        const int32_t a_n_nodes = have_aln ? (int32_t)gld(arec() + 6) : 0;
        R.have_aln = (have_aln && a_n_nodes) ? 1 : 0;
        R.mode = LANE_MODE();
        R.strand = LANE_STRAND();
        R.words = 0; R.trim = 0;
        if (R.have_aln) {
            R.score = (int32_t)gld(arec() + 0); R.offset = (int32_t)gld(arec() + 1); R.clip = (int32_t)gld(arec() + 2);
            R.end_clip = (int32_t)gld(arec() + 3); R.n_runs = (int32_t)gld(arec() + 4); R.j_hi = (int32_t)gld(arec() + 5);
            R.n_nodes = a_n_nodes; R.n_seq = (int32_t)gld(arec() + 7);
            R.words = (uint32_t)R.n_nodes + (uint32_t)((R.clip ? 1 : 0) + R.n_runs + (R.end_clip ? 1 : 0)) + ((uint32_t)R.n_seq + 3) / 4;
            if (P.labeled) { R.words += 2; R.label = LANE_LABEL(); }                 // (label count + the one label behind the alignment)
        }
    }
    LANE_T(6);
    return LR_DONE;
}

#undef replay_top
#undef c_fwd_n_nodes
#undef c_fwd_n_seq
#undef c_seed_len
#undef c_seed_off
#undef c_node0
#undef b_score
#undef b_nod
#undef b_i
#undef b_pos
#undef t_score
#undef t_nod
#undef t_pos
#undef fa_alive
#undef fa_conv
#undef fa_max_val
#undef d_score
#undef d_max
#undef kid_node0
#undef kid_codes
#undef kid_node1
#undef kid_rank0
#undef kid_rank1
#undef table_cap
#undef tsb_lo
#undef tsb_hi
#undef table_size_bytes
#undef cell_top
#undef cols_done
#undef f_node
#undef f_idx
#undef f_max_val

// flat_read_end: the aggregator's one alignment -> output stream at word `so` (R.words of them, handed out by the caller: one
// atomic per wavefront on the device), the result record, the seed dump of the test hook
MGX_DEV void lane_emit(const LaneParams &LP, const uint64_t read, const uint8_t *scratch, const LaneChip &chip, LaneResult &R,
                       const uint64_t so) {
    const AlignParams &P = LP.P;
    const uint8_t *slots = scratch;                      // (lane_slot_word)
    ReadResult &rr = R.rr;
    if (R.have_aln) {
        if (so + R.words > P.out_capacity) {
            rr.status = ST_CAPACITY;                     // (the stage is re-run with the size the cursor asks for)
        } else {
            uint32_t *dst = P.out_stream + so;
            const int32_t n_cigar = (R.clip ? 1 : 0) + R.n_runs + (R.end_clip ? 1 : 0);
            uint8_t *dseq = (uint8_t *)(dst + R.n_nodes + n_cigar);
            const int32_t k_minus_1 = (int32_t)P.g.k - 1;
            int32_t nc = 0;
            if (R.clip) gst(dst + R.n_nodes + nc++, ((uint32_t)R.clip << 3) | OP_CLIPPED);
            if (R.mode == LANE_EMIT_SLOTS) {
                for (int32_t x = R.n_runs - 1; x >= 0; --x) gst(dst + R.n_nodes + nc++, gld(lane_run_word(slots, LP.max_cols, x)));
                // the path, last column first, down the parent links: characters of all its columns, nodes of those whose offset
                // reaches k - 1 (minus the leading ones trim_offset dropped)
                int32_t j = R.j_hi, ni = R.n_nodes - 1;
                for (int32_t x = R.n_seq - 1; x >= 0; --x) {
                    const uint32_t node = gld(lane_slot_word(slots, j, 8)), geom = gld(lane_slot_word(slots, j, 10)), link = gld(lane_slot_word(slots, j, 11));
                    gst(dseq + x, decode_code((geom >> 24) & 7u));
                    if ((int32_t)(link >> 16) >= k_minus_1) { if (ni >= 0) gst(dst + ni, node); --ni; }
                    j = (int32_t)(link & 0xFFFFu);
                }
            } else if (R.mode == LANE_EMIT_ARRAYS) {
                // the forward alignment as the backward pass kept it: path order, runs last first
                const uint32_t *pn = (const uint32_t *)(lane_rest(LP, scratch, chip.lane) + (uint64_t)LP.hash_slots * 8) + 4 * LFW;
                const uint32_t *pc = pn + LP.max_cols, *pr = pc + LP.max_cols;
                for (int32_t x = R.n_runs - 1; x >= 0; --x) gst(dst + R.n_nodes + nc++, gld(pr + x));
                for (int32_t x = 0; x < R.n_nodes; ++x) gst(dst + x, gld(pn + x));
                for (int32_t x = 0; x < R.n_seq; ++x) gst(dseq + x, decode_code(gld(pc + x)));
            } else {
                // the backward alignment reversed (Alignment::reverse_complement): its runs in the order the trace recorded them,
                // its columns in the order the parent links give them, characters complemented
                for (int32_t x = 0; x < R.n_runs; ++x) gst(dst + R.n_nodes + nc++, gld(lane_run_word(slots, LP.max_cols, x)));
                int32_t j = R.j_hi, ni = 0;
                for (int32_t x = 0; x < R.n_seq; ++x) {
                    const uint32_t node = gld(lane_slot_word(slots, j, 8)), geom = gld(lane_slot_word(slots, j, 10)), link = gld(lane_slot_word(slots, j, 11));
                    gst(dseq + x, decode_code(5u - ((geom >> 24) & 7u)));
                    if ((int32_t)(link >> 16) >= k_minus_1) { if (ni < R.n_nodes) gst(dst + ni, node); ++ni; }
                    j = (int32_t)(link & 0xFFFFu);
                }
            }
            if (R.end_clip) gst(dst + R.n_nodes + nc++, ((uint32_t)R.end_clip << 3) | OP_CLIPPED);
            for (int32_t x = R.n_seq; x & 3; ++x) gst(dseq + x, (uint8_t)0);          // (pad the last word)
            if (P.labeled) {
                // the labeled stream layout (align_core.hpp, the labeled result writer): the alignment's label count and labels follow it
                uint32_t *lab = dst + R.n_nodes + n_cigar + ((uint32_t)R.n_seq + 3) / 4;
                gst(lab, 1u); gst(lab + 1, R.label);
            }
            rr.score = R.score; rr.offset = (uint32_t)R.offset;
            rr.n_nodes = (uint32_t)R.n_nodes; rr.n_cigar = (uint32_t)n_cigar; rr.seq_len = (uint32_t)R.n_seq;
            rr.orientation = (uint32_t)R.strand; rr.stream_off = so;
            rr.n_alignments = 1;
        }
    }
    {
        ReadResult *dst = P.results + read;
        gst(&dst->status, rr.status); gst(&dst->n_alignments, rr.n_alignments); gst(&dst->score, rr.score); gst(&dst->offset, rr.offset);
        gst(&dst->n_nodes, rr.n_nodes); gst(&dst->n_cigar, rr.n_cigar); gst(&dst->seq_len, rr.seq_len); gst(&dst->orientation, rr.orientation);
        gst(&dst->stream_off, rr.stream_off);
        gst(&dst->num_matches_fwd, rr.num_matches_fwd); gst(&dst->num_matches_rc, rr.num_matches_rc);
        gst(&dst->n_seeds_fwd, rr.n_seeds_fwd); gst(&dst->n_seeds_rc, rr.n_seeds_rc);
        gst(&dst->n_extensions, rr.n_extensions); gst(&dst->n_columns, rr.n_columns);
    }
    if (P.dbg_seeds) {
        const SeedHdr *hp = P.seed_hdr + read;
        const uint64_t h_off = gld(&hp->off);
        const int32_t ns0 = (int32_t)gld(&hp->n_seeds[0]), ns1 = (int32_t)gld(&hp->n_seeds[1]);
        for (int st = 0; st < 2; ++st) {
            // (label-aware: a strand whose label fell below min_exact_match has lost its seeds)
            const int32_t cnt = P.labeled ? (int32_t)(st ? rr.n_seeds_rc : rr.n_seeds_fwd) : (st ? ns1 : ns0);
            const DevSeed *src = P.seed_stream + h_off + (st ? ns0 : 0);
            for (int32_t i = 0; i < cnt; ++i) {
                DevSeed *d = P.dbg_seeds + ((uint64_t)read * 2 + st) * P.lim.max_seeds + i;
                gst(&d->clipping, gld(&src[i].clipping)); gst(&d->length, gld(&src[i].length));
                gst(&d->offset, gld(&src[i].offset)); gst(&d->n_nodes, gld(&src[i].n_nodes)); gst(&d->node, gld(&src[i].node));
            }
        }
    }
}

} // namespace mgx
