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
    uint64_t *qw; int32_t qstride;       // packed strand of the read: word i at qw[i * qstride]
    uint32_t *runs; int32_t rstride;     // CIGAR runs of the trace, last first: run i at runs[i * rstride]
};

// profile scores over the packed strand (see LaneProfBytes)
struct LaneProfPacked {
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

enum { LR_DONE = 0, LR_BAIL = 1 };

struct LaneCounters { uint32_t rank_lines, select_lines, columns, seeds; };

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

// the single child of `v` on the forward graph (DBGSuccinct::call_outgoing_kmers, dbg_succinct.cpp:110-139, minus the
// sentinel-labelled children the extender drops, aligner_extender_methods.cpp:381-384; dev_graph.hpp outgoing() without its
// arrays): returns the number of children (0, 1, or 2 = "more than one"), the child and its label code in *node / *code
MGX_DEV int lane_single_child(const DevGraph &g, uint32_t vv, uint32_t *node, uint32_t *code, LaneCounters &ctr) {
    LineCtr lc = { 0, 0, 0 };
    const uint64_t v = vv;
    ++lc.rank_lines;
    const Block cur = load_block(g, (uint32_t)(v >> 6));
    const uint32_t w = block_W(cur, (int)(v & 63));
    int n = 0;
    if (!(v > 1 && w == 0)) {
        Block tgt;
        const uint64_t lst = fwd_from(g, v, cur, w % SIGMA, tgt, lc);
        uint64_t first = pred_last_from(g, lst - 1, ((lst - 1) >> 6) == (lst >> 6) ? tgt : load_block(g, (uint32_t)((lst - 1) >> 6)), lc) + 1;
        if (first < 2) first = 2;
        Block b = tgt;
        uint32_t bi = (uint32_t)(lst >> 6);
        for (uint64_t i = first; i <= lst; ++i) {
            if ((uint32_t)(i >> 6) != bi) { bi = (uint32_t)(i >> 6); ++lc.rank_lines; b = load_block(g, bi); }
            const uint32_t c = block_W(b, (int)(i & 63)) % SIGMA;
            if (c != 0 && in_graph(g, i)) { if (n == 0) { *node = (uint32_t)i; *code = c; } ++n; }
        }
    }
    ctr.rank_lines += lc.rank_lines; ctr.select_lines += lc.select_lines;
    return n > 1 ? 2 : n;
}

// what lane_read() leaves for lane_emit(): the result record and where the alignment's pieces are
struct LaneResult {
    ReadResult rr;
    int32_t have_aln;
    int32_t score, offset, clip, end_clip, n_runs, j_lo, n_nodes, n_seq, j_first_node, strand;
    uint32_t words;                      // words of the output stream the alignment takes
};

// One read.  `item`: its position in the launch (tags the node table).  scratch: this lane's LaneParams::scratch slice.
// Returns LR_DONE (R filled: lane_emit() writes results[read] and the output stream) or LR_BAIL (nothing to write).
MGX_DEV int lane_read(const LaneParams &LP, const uint64_t read, const uint32_t item, uint8_t *scratch, const LaneChip &chip,
                      LaneCounters &ctr, LaneResult &R) {
    const AlignParams &P = LP.P;
    const DevConfig &cfg = P.cfg;
    const DevLimits &lim = P.lim;
    const int32_t k = (int32_t)P.g.k;
    const int32_t go = cfg.gap_open, ge = cfg.gap_ext;
    const int32_t m = LP.self_score;
    // ---- the read and its seeds (flat_read_begin) ----
    const uint64_t off = gld(P.offsets + read);
    const int32_t L = (int32_t)(gld(P.offsets + read + 1) - off);
    if (L > (int32_t)lim.Lmax || L > LANE_MAX_L) return LR_BAIL;
    const SeedHdr *hp = P.seed_hdr + read;
    const int32_t h_status = gld(&hp->status);
    if (h_status != ST_OK) return LR_BAIL;
    const uint64_t h_off = gld(&hp->off);
    const uint32_t nm0 = gld(&hp->num_matching[0]), nm1 = gld(&hp->num_matching[1]);
    const int32_t ns0 = (int32_t)gld(&hp->n_seeds[0]), ns1 = (int32_t)gld(&hp->n_seeds[1]);
    const bool have_rc = cfg.fwd_and_rc != 0;
    // align_both_directions (:738-755): the strand with more matches; the other one only if it is within rel_score_cutoff
    const int first = nm0 >= nm1 ? 0 : 1;
    const int s = have_rc ? first : 0;
    if (have_rc) {
        const uint32_t m_first = first ? nm1 : nm0, m_second = first ? nm0 : nm1;
        const int32_t n_second = first ? ns0 : ns1;
        if ((double)m_second >= (double)m_first * cfg.rel_score_cutoff && n_second > 0) return LR_BAIL;     // a second strand to align
    }
    const int32_t n = s ? ns1 : ns0;
    ReadResult &rr = R.rr;
    rr.status = ST_OK; rr.n_alignments = 0; rr.score = 0; rr.offset = 0; rr.n_nodes = rr.n_cigar = rr.seq_len = 0;
    rr.orientation = 0; rr.stream_off = 0;
    rr.num_matches_fwd = nm0; rr.num_matches_rc = nm1; rr.n_seeds_fwd = (uint32_t)ns0; rr.n_seeds_rc = (uint32_t)ns1;
    rr.n_extensions = 0; rr.n_columns = 0;
    const DevSeed *seeds = P.seed_stream + h_off + (s ? ns0 : 0);
    uint8_t *slots = scratch;
    uint8_t *s8rows = scratch + (uint64_t)LP.max_cols * LANE_SLOT_BYTES;
    uint64_t *htab = (uint64_t *)(scratch + (uint64_t)LP.max_cols * (LANE_SLOT_BYTES + LANE_S8_BYTES));
    bool have_aln = false;
    // the alignment, as far as the output needs it
    int32_t a_score = 0, a_offset = 0, a_clip = 0, a_end_clip = 0, a_n_runs = 0;
    int32_t a_j_lo = 0, a_j_hi = 0, a_n_nodes = 0, a_n_seq = 0, a_j_first_node = 0;
    if (n > 0) {
        if (n - 1 > LANE_MAX_LATER) return LR_BAIL;
        // the strand: 2-bit packed by k_pack_reads; any character outside ACGT (psum_lin == 0) is not for this kernel
        const uint64_t wb = packed_word_begin(off, read);
        const int32_t nw = (L + 31) >> 5;
        uint32_t any_inv = 0;
        for (int32_t j = 0; j < LANE_QWORDS; ++j) {
            uint64_t v = 0;
            if (j < nw) { v = gld(LP.pk[s] + wb + j); any_inv |= gld(LP.iv[s] + wb + j); }
            chip.qw[j * chip.qstride] = v;
        }
        if (any_inv) return LR_BAIL;
        auto qcode = [&](int32_t qi) -> uint32_t { return (uint32_t)(chip.qw[(qi >> 5) * chip.qstride] >> (2 * (qi & 31))) & 3u; };
        // ---- seed 0 (seedref_from_seed) and the later seeds ----
        const uint64_t nb = gld(P.node_begin + read);
        const uint32_t *rnodes = (s ? P.nodes_rc : P.nodes_fwd) + nb;
        const DevSeed *s0 = seeds;
        const int32_t clipping = (int32_t)gld(&s0->clipping), seed_len = (int32_t)gld(&s0->length);
        const int32_t seed_off = (int32_t)gld(&s0->offset);
        const uint32_t node0 = seed_off == 0 ? gld(rnodes + clipping) : gld(&s0->node);
        if (node0 == 0) return LR_BAIL;
        const int32_t end_clipping0 = L - clipping - seed_len;
        const int32_t seed_score = seed_len * m + (!clipping ? cfg.left_end_bonus : 0) + (!end_clipping0 ? cfg.right_end_bonus : 0);
        uint32_t later_node[LANE_MAX_LATER];
        int32_t later_pos[LANE_MAX_LATER], later_score[LANE_MAX_LATER];
        uint32_t later_live = 0;           // bit t: later seed t has not been found dead yet
#pragma unroll
        for (int t = 0; t < LANE_MAX_LATER; ++t) {
            later_node[t] = 0; later_pos[t] = 0; later_score[t] = 0;
            if (t + 1 < n) {
                const DevSeed *sj = seeds + t + 1;
                const int32_t cl = (int32_t)gld(&sj->clipping), len = (int32_t)gld(&sj->length), so = (int32_t)gld(&sj->offset);
                const int32_t nn = (int32_t)gld(&sj->n_nodes);
                later_node[t] = so == 0 ? gld(rnodes + cl + nn - 1) : gld(&sj->node);
                later_pos[t] = len + cl - 1;
                later_score[t] = len * m + (!cl ? cfg.left_end_bonus : 0) + (!(L - cl - len) ? cfg.right_end_bonus : 0);
                later_live |= 1u << t;
                if (later_node[t] == node0) return LR_BAIL;          // (its check would need the merged vector of the replay columns)
            }
        }
        // ---- extend_begin (:412-470): set_seed, the root column ----
        const int32_t xdrop = cfg.xdrop;
        int32_t xdrop_cutoff = imax(-xdrop, NINF + 1);
        const int32_t start = clipping, window_size = L - start, qlen = L;
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
        auto root_S = [&](int32_t pos) -> int32_t {
            return pos == 0 ? sroot : (pos >= 1 && pos <= root_pushes ? root_ins + (pos - 1) * ge : NINF);
        };
        if ((uint64_t)rec_words((uint32_t)root_size + 8) > lim.cell_words) return LR_BAIL;
        const uint32_t cell_top = rec_words((uint32_t)((root_size + 5 + 3) & ~3));
        uint32_t table_cap = 1;
        int32_t tsize = 1;
        uint64_t table_size_bytes = (uint64_t)136 * table_cap + (uint64_t)(3 * ref_capacity(1, (uint32_t)root_pushes)) * 4;
        int32_t min_cell_score = 0, best_score = 0;
        rr.n_extensions = 1;
        // the root leaves the frontier and enters the chain window (extend_step: fast_fits + fast_load)
        if (root_size + 3 > LFW) return LR_BAIL;
        int32_t S[LFW], F[LFW];
#pragma unroll
        for (int x = 0; x < LFW; ++x) { S[x] = x < root_size ? root_S(x) : NINF; F[x] = NINF; }
        int32_t f_org = 0, f_trim = 0, f_size = root_size, f_offset = seed_off - 1, f_max_val = sroot;
        uint32_t f_node = node0;
        // backtrack start cells, collected while the columns are in registers (bt_begin :815-867)
        const int32_t seed_dist = imax(k, seed_len) - 1;
        const int32_t last_pos = window_size;
        const int32_t seed_offset = seed_off - 1;
        const int32_t min_start_score = have_rc ? imax(0, cfg.min_cell_score) : imax(0, cfg.min_path_score);
        int32_t b_score = INT32_MIN, b_nod = INT32_MIN, b_i = 0, b_pos = 0;         // the best (score, -off_diag, -i, pos)
        int32_t t_score = INT32_MIN, t_nod = 0, t_pos = 0;                          // the last column's start cell if it were a tip
        auto cand = [&](int32_t sc, int32_t nod, int32_t i, int32_t pos) {
            const bool better = sc != b_score ? sc > b_score : (nod != b_nod ? nod > b_nod : (-i != -b_i ? -i > -b_i : pos > b_pos));
            if (b_score == INT32_MIN || better) { b_score = sc; b_nod = nod; b_i = i; b_pos = pos; }
        };
        const uint32_t tag = ((LP.tag_seed + item) * 0x9E3779B1u >> 12) | 1u;              // 20 bits, never 0
        const uint32_t hmask = LP.hash_slots - 1;
        LaneProfPacked prof;
        prof.qw = chip.qw; prof.qstride = chip.qstride; prof.qlen = qlen; prof.rowp = 0; prof.w = 0;
        // ---- the extension: chain steps (extend_step / chain_step) ----
        for (;;) {
            // early cut-offs when off the optimal path (:521-547)
            if (f_max_val < best_score) {
                if ((double)tsize / (double)window_size >= cfg.max_nodes_per_seq_char) break;
                if ((double)table_size_bytes / 1000000.0 > cfg.max_ram_per_alignment) break;
            }
            LaneColumnIn in;
            in.p_org = f_org; in.p_trim = f_trim; in.p_size = f_size;
            in.xdrop_cutoff = xdrop_cutoff; in.start = start; in.window_size = window_size; in.qlen = qlen; in.go = go; in.ge = ge;
            int32_t begin, prev_end;
            lane_band(in, S, begin, prev_end);
            if (prev_end <= begin) break;
            // the child (call_outgoing :330-387)
            const int32_t next_offset = f_offset + 1;
            const int32_t seed_pos = next_offset - seed_off;
            const bool in_seed = seed_pos >= 0 && seed_pos < seed_len;
            uint32_t next, ccode;
            if (in_seed && next_offset < k) {
                next = node0; ccode = qcode(clipping + seed_pos) + 1;             // the seed's first node, its spelling
            } else {
                const int nc = lane_single_child(P.g, f_node, &next, &ccode, ctr);
                if (nc == 0) {                                                     // a tip: its start cell counts after all
                    if (t_score != INT32_MIN) cand(t_score, t_nod, tsize - 1, t_pos);
                    break;
                }
                if (nc != 1) return LR_BAIL;                                       // a fork
            }
            if (next == 0) return LR_BAIL;
            if (tsize >= (int32_t)LP.max_cols - 1 || tsize >= (int32_t)lim.max_columns - 1) return LR_BAIL;
            if ((uint64_t)cell_top + rec_words((uint32_t)(window_size + 1 - begin + 8)) > lim.cell_words) return LR_BAIL;
            // The node table: first visits only (a node seen before would merge convergence vectors, update_seed_filter
            // :100-156).  The replay columns all carry the seed's first node and DO merge — but all the chain needs from the
            // merge is whether some cell improved on the node's vector (converged != ninf), and a replay column's diagonal cell
            // always does: the replayed characters are the query's own, so that cell scores sroot + (t + 1) m, more than any
            // earlier column (fewer graph characters, hence at most sroot + (t' + 1) m) left at its query position.  (Needs
            // m > 0 >= gaps, mismatches <= m, sroot >= 0, 0 <= rel_score_cutoff <= 1: checked before the kernel is launched.)
            // The vector itself is never needed: a later seed ending in that node, or the graph leading back to it, bails.
            const bool replay = in_seed && next_offset < k;
            const bool probe = !replay || f_offset == seed_off - 1;
            uint32_t hs = lane_hash(next, hmask);
            uint64_t he = probe ? gld(htab + hs) : 0;
            in.next_offset = next_offset; in.score = 0; in.in_seed = in_seed;
            in.best_score = best_score; in.min_cell_score = min_cell_score; in.rel_cutoff = cfg.rel_score_cutoff;
            in.partial_sum_offset = 0; in.psum_lin = m; in.psum = nullptr; in.seed_off = seed_off; in.q = nullptr; in.row = nullptr;
            prof.rowp = ccode == 1 ? LP.t4[0] : ccode == 2 ? LP.t4[1] : ccode == 3 ? LP.t4[2] : LP.t4[3];
            LaneColumnOut out;
            const int rc = lane_column(in, S, F, out, prof);
            if (rc == LC_FALLBACK) return LR_BAIL;
            const uint32_t table_cap_before = table_cap;
            if ((uint32_t)tsize == table_cap) table_cap = imax<uint32_t>(1u, 2 * table_cap);
            ++rr.n_columns;
            min_cell_score = out.min_cell_score;
            if (rc == LC_POP) break;                                               // pop(table.size() - 1) (:646-653)
            table_size_bytes += (uint64_t)136 * (table_cap - table_cap_before)
                                + (uint64_t)(3 * ref_capacity((uint32_t)out.size0, (uint32_t)out.pushes)) * 4;
            const int32_t max_val = out.max_val;
            if ((int32_t)((uint32_t)max_val - (uint32_t)xdrop_cutoff) > xdrop) xdrop_cutoff = max_val - xdrop;
            best_score = imax(best_score, max_val);
            const int32_t my_idx = tsize;
            if (probe) {
                for (;;) {
                    if ((uint32_t)(he >> 44) != tag) break;                        // free (or left by another read)
                    if ((uint32_t)he == next) return LR_BAIL;                     // seen before
                    hs = (hs + 1) & hmask;
                    he = gld(htab + hs);
                }
                gst(htab + hs, (uint64_t)next | ((uint64_t)tag << 44) | ((uint64_t)(uint32_t)my_idx << 32));
            }
            // (replay columns: see above; the column's own maximum stands in for the merged score, ninf neither way)
            const int32_t converged = out.converged;
            const int32_t size = out.size, org = out.org;
            // the frontier hands the column straight back (:491-504) — or it would stay behind with a record of its own
            if (converged != NINF && !((begin & 3) + size + 3 <= LFW)) return LR_BAIL;
            // commit: the slot (flags, node, base, geometry) and the S row
            const int32_t base = max_val == NINF ? 0 : max_val;
            {
                uint32_t sw[8];
                bool wide = false;
#pragma unroll
                for (int b = 0; b < LFW / 4; ++b) {
                    uint32_t v = 0;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int32_t sv = S[4 * b + q4];
                        const int32_t d = sv - base;
                        wide |= sv != NINF && d < -127;
                        v |= (sv == NINF ? 0x80u : ((uint32_t)d & 0xFFu)) << (8 * q4);
                    }
                    sw[b] = v;
                }
                if (wide) return LR_BAIL;
                uint32_t *sl = (uint32_t *)(slots + (uint64_t)my_idx * LANE_SLOT_BYTES);
                uint32_t *sr = (uint32_t *)(s8rows + (uint64_t)my_idx * LANE_S8_BYTES);
#pragma unroll
                for (int b = 0; b < 8; ++b) gst(sl + b, out.fw[b]);
                gst(sl + 8, next);
                gst(sl + 9, (uint32_t)base);
                gst(sl + 10, (uint32_t)begin | ((uint32_t)size << 16) | (ccode << 24));
#pragma unroll
                for (int b = 0; b < 8; ++b) gst(sr + b, sw[b]);
            }
            tsize = my_idx + 1;
            // check_seed (:66-88) of the later seeds whose last node this is: its (first and only) column is in registers
#pragma unroll
            for (int t = 0; t < LANE_MAX_LATER; ++t) {
                if (((later_live >> t) & 1u) && later_node[t] == next) {
                    const int32_t skip = begin ? 0 : 1;
                    const int32_t qs = start + begin - (begin ? 1 : 0), len = size - skip;
                    const int32_t pos = later_pos[t];
                    if (!(pos < qs || pos - qs >= len)) {
                        const int32_t a = pos - start + 1;
                        // (cell_S: outside [trim, trim + size + 5) or the slot's cells the column holds nothing)
                        const int32_t v = (a - begin >= 0 && a - begin < size + 5) ? lane_win_at(S, a - org) : NINF;
                        if (!(v < later_score[t])) later_live &= ~(1u << t);
                    }
                }
            }
            // start cells of this column (bt_begin :815-867)
            t_score = INT32_MIN;
            if (next_offset >= seed_dist) {
                const int32_t max_pos = out.max_pos;
                {
                    const uint32_t fl = lane_flags_at(out.fw, max_pos - org);
                    if ((fl & CF_REAL) && (fl & CF_SP_REAL)) {
                        const int32_t eb = max_pos == last_pos ? cfg.right_end_bonus : 0;
                        if (base + eb >= min_start_score) {
                            const int32_t ap = clipping + max_pos;
                            const bool is_match = (fl & CF_MATCH) && ap >= 1 && ap <= L && qcode(ap - 1) + 1 == ccode;
                            const int32_t nod = -iabs(max_pos - next_offset + seed_offset);
                            if (is_match || max_pos == last_pos) cand(base + eb, nod, my_idx, max_pos);
                            else { t_score = base + eb; t_nod = nod; t_pos = max_pos; }
                        }
                    }
                }
                if (size + begin == window_size + 1 && max_pos != last_pos) {
                    const uint32_t fl = lane_flags_at(out.fw, last_pos - org);
                    if ((fl & CF_REAL) && (fl & CF_SP_REAL)) {
                        const int32_t sv = lane_win_at(S, last_pos - org);
                        if (sv + cfg.right_end_bonus >= min_start_score)
                            cand(sv + cfg.right_end_bonus, -iabs(last_pos - next_offset + seed_offset), my_idx, last_pos);
                    }
                }
            }
            if (converged == NINF) break;
            f_org = org; f_trim = begin; f_size = size; f_offset = next_offset; f_max_val = max_val; f_node = next;
        }
        ctr.columns += rr.n_columns;
        // ---- backtrack (:869-1034): the best start cell, one trace ----
        if (b_score == INT32_MIN) return LR_BAIL;                                  // no start cell: the seed itself would be reported
        if (later_live) return LR_BAIL;                                            // a later seed survives: more extensions to run
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
        auto push_op = [&](uint32_t op, uint32_t num) {
            if (n_runs == 0 || (cur_run & 7) != op) {
                if (n_runs >= LANE_MAX_RUNS) { bad = true; return; }
                cur_run = (num << 3) | op;
                chip.runs[n_runs++ * chip.rstride] = cur_run;
            } else {
                cur_run += num << 3;
                chip.runs[(n_runs - 1) * chip.rstride] = cur_run;
            }
        };
        auto slot_geom = [&](int32_t jj) -> uint32_t { return gld((const uint32_t *)(slots + (uint64_t)jj * LANE_SLOT_BYTES) + 10); };
        auto slot_flags = [&](int32_t jj, uint32_t geom, int32_t p) -> uint32_t {
            const int32_t begin = (int32_t)(geom & 0xFFFF), size = (int32_t)((geom >> 16) & 0xFF);
            const int32_t jx = p - begin, x = p - (begin & ~3);
            if (!(jx >= 0 && jx < size + 5 && x < LFW)) return 0;
            return gld(slots + (uint64_t)jj * LANE_SLOT_BYTES + x);
        };
        for (;;) {
            if (!j) break;
            const uint32_t geom = slot_geom(j);
            const uint32_t ccode = geom >> 24;
            const int32_t col_offset = seed_off - 1 + j;
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
                --j;
                j_stop = j;
            } else if ((fl & CF_S_IS_F) && (n_runs == 0 || last_op != OP_INSERTION)) {
                uint32_t lop = OP_DELETION;
                while (lop == OP_DELETION && j && !bad) {
                    const uint32_t g2 = slot_geom(j);
                    const int32_t o2 = seed_off - 1 + j;
                    align_offset = imin(o2, k_minus_1);
                    lop = (slot_flags(j, g2, pos) & CF_F_EXT) ? OP_DELETION : OP_MATCH;
                    ++n_trace;
                    ++n_seq;
                    push_op(OP_DELETION, 1);
                    if (o2 >= k_minus_1) ++n_path;
                    --j;
                    j_stop = j;
                }
            } else {
                j_stop = j;
                break;
            }
            if (bad || n_seq > cap || n_path > cap) return LR_BAIL;
        }
        if (bad) return LR_BAIL;
        if (!(n_trace >= min_trace_length && n_path)) return LR_BAIL;             // (the next start cell would be tried)
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
                    const int32_t v = (int32_t)(int8_t)gld(s8rows + (uint64_t)j * LANE_S8_BYTES + x);
                    const int32_t cb = (int32_t)gld((const uint32_t *)(slots + (uint64_t)j * LANE_SLOT_BYTES) + 9);
                    if (v != -128) cur_cell_score = cb + v;
                }
            }
            const int32_t bt_best = score - cur_cell_score;                         // best_score = max(INT32_MIN, .)
            if (score - min_cell_score < bt_best) return LR_BAIL;                   // no alignment from this extension
            if (!(score >= min_start_score && (!pos || cur_cell_score == 0) && (pos || cur_cell_score == sroot)
                  && (cfg.allow_left_trim || !j))) return LR_BAIL;                 // (the next start cell would be tried)
        }
        // construct_alignment (:774-798) + trim_offset (alignment.cpp:177-190)
        a_clip = clipping + pos;
        a_end_clip = L - (clipping + end_pos);
        a_score = score; a_offset = align_offset; a_n_runs = n_runs;
        a_j_lo = j_stop + 1; a_j_hi = j_start; a_n_seq = n_seq; a_n_nodes = n_path;
        a_j_first_node = imax(a_j_lo, k - seed_off);
        if (a_j_hi - a_j_lo + 1 != n_seq || a_j_hi - imax(a_j_lo, k - seed_off) + 1 != n_path) return LR_BAIL;   // (cannot happen: every column left appends once)
        if (a_offset && a_n_nodes > 1) {
            const int32_t trim = imin(a_offset, a_n_nodes - 1);
            if (trim > 0) { a_j_first_node += trim; a_n_nodes -= trim; a_offset -= trim; }
        }
        have_aln = true;
        // ---- aln_both after the forward pass (:683-736), no backward pass in this kernel ----
        if (have_rc) {
            if (a_clip && !a_offset) return LR_BAIL;                                // extend backwards from the reversed alignment
            if (!(a_score >= cfg.min_path_score)) have_aln = false;                 // get_min_path_score with an empty aggregator
        }
        (void)seed_score;
    }
    R.have_aln = (have_aln && a_n_nodes) ? 1 : 0;
    R.score = a_score; R.offset = a_offset; R.clip = a_clip; R.end_clip = a_end_clip; R.n_runs = a_n_runs;
    R.j_lo = a_j_lo; R.n_nodes = a_n_nodes; R.n_seq = a_n_seq; R.j_first_node = a_j_first_node; R.strand = s;
    R.words = R.have_aln ? (uint32_t)a_n_nodes + (uint32_t)((a_clip ? 1 : 0) + a_n_runs + (a_end_clip ? 1 : 0)) + ((uint32_t)a_n_seq + 3) / 4 : 0u;
    return LR_DONE;
}

// flat_read_end: the aggregator's one alignment -> output stream at word `so` (R.words of them, handed out by the caller: one
// atomic per wavefront on the device), the result record, the seed dump of the test hook
MGX_DEV void lane_emit(const LaneParams &LP, const uint64_t read, const uint8_t *scratch, const LaneChip &chip, LaneResult &R,
                       const uint64_t so) {
    const AlignParams &P = LP.P;
    const uint8_t *slots = scratch;
    ReadResult &rr = R.rr;
    if (R.have_aln) {
        if (so + R.words > P.out_capacity) {
            rr.status = ST_CAPACITY;                     // (the stage is re-run with the size the cursor asks for)
        } else {
            uint32_t *dst = P.out_stream + so;
            const int32_t n_cigar = (R.clip ? 1 : 0) + R.n_runs + (R.end_clip ? 1 : 0);
            // nodes of the columns whose offset reaches k - 1, first to last, minus what trim_offset dropped
            for (int32_t x = 0; x < R.n_nodes; ++x)
                gst(dst + x, gld((const uint32_t *)(slots + (uint64_t)(R.j_first_node + x) * LANE_SLOT_BYTES) + 8));
            int32_t nc = 0;
            if (R.clip) gst(dst + R.n_nodes + nc++, ((uint32_t)R.clip << 3) | OP_CLIPPED);
            for (int32_t x = R.n_runs - 1; x >= 0; --x) gst(dst + R.n_nodes + nc++, chip.runs[x * chip.rstride]);
            if (R.end_clip) gst(dst + R.n_nodes + nc++, ((uint32_t)R.end_clip << 3) | OP_CLIPPED);
            uint32_t *dseq = dst + R.n_nodes + n_cigar;
            for (int32_t x = 0; x < R.n_seq; x += 4) {
                uint32_t v = 0;
                for (int32_t t = 0; t < 4 && x + t < R.n_seq; ++t) {
                    const uint32_t cc = gld((const uint32_t *)(slots + (uint64_t)(R.j_lo + x + t) * LANE_SLOT_BYTES) + 10) >> 24;
                    v |= (uint32_t)decode_code(cc) << (8 * t);
                }
                gst(dseq + (x >> 2), v);
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
            const int32_t cnt = st ? ns1 : ns0;
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
