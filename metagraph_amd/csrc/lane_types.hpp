// lane_types.hpp — what the host needs to know about the lane-per-read kernel (lane_read.hpp, mgx_lane.hip): its launch
// parameters, its scratch layout and the test of whether a batch's configuration is one the kernel takes at all.
#pragma once
#include <algorithm>
#include <string>

#include "../../include/mgx.h"
#include "align_types.hpp"


namespace mgx {

// longest read a lane takes (packed strand in LDS: LANE_QWORDS words per lane).  The host lays the scratch out for 256; the kernel
// is built twice (mgx_lane.hip): for reads of up to 256 characters, and — MGX_LANE_MAX_L=160 — for batches whose longest read has
// at most 160, where the smaller packed strand is what lets a third wavefront per SIMD fit the LDS.
#ifndef MGX_LANE_MAX_L
#define MGX_LANE_MAX_L 256
#endif
constexpr int LANE_MAX_L = MGX_LANE_MAX_L;
constexpr int LANE_MAX_L_HOST = 256;       // what the scratch layout is sized for, whichever build runs
constexpr int LANE_QWORDS = LANE_MAX_L / 32 + 2;
constexpr int LANE_MAX_RUNS = 16;          // CIGAR runs of a trace kept in LDS
constexpr int LANE_MAX_SEEDS = 512;        // seeds of the strand (the later ones are checked against the extension one by one)
constexpr int LANE_SLOT_WORDS = 12;        // per column: 8 flag words (32 cells, a byte each) + node + base + geometry + link
constexpr int LANE_WAVE = 64;              // lanes whose column slots and S rows are interleaved word by word (below)
constexpr int LANE_MAX_DEFER = 3;          // columns that may stay behind in the frontier (the other children of forks)
constexpr int LANE_DSLOT_WORDS = 80;       // a column that stays behind: its window (64 words) + nine words of column state
constexpr uint32_t LANE_GEOM_ROW = 1u << 31;   // in a slot's geometry word: the column's S row was written
constexpr int LANE_S8_FILTER_SEEDS = 16;   // later seeds whose last nodes select the columns that keep their S row (more: every column does)
constexpr int LANE_S8_WORDS = 8;           // per column: S of the window as 8-bit offsets from `base` (read at the trace's end only)

// what one launch of the lane kernel needs on top of AlignParams
struct LaneParams {
    AlignParams P;
    const uint64_t *pk[2];               // 2-bit packed strands (k_pack_reads): word j of read r at packed_word_begin(offsets[r], r) + j
    const uint32_t *iv[2];               // invalid-character flags, same indexing
    // Scratch of a resident wavefront (round 6: wave-interleaved).  The 64 lanes of a wavefront step their columns in lock-step —
    // lane_read() starts every lane at column 1 and each loop iteration commits one — so the column slots and S rows are laid
    // out [column][word][lane]: the store of "word w of column c" by the 64 lanes is 256 contiguous bytes (four full lines)
    // where lane-private slices made it 64 partial-line writes 18 KB apart, and the trace's reads of a column's words find
    // the lanes' words in the same lines.  What is addressed by a lane's own data (the node table) or touched rarely (parked
    // windows, the forward alignment during a backward pass, counters) stays in a private slice per lane behind them:
    //   wave w:  [ slots: max_cols x LANE_SLOT_WORDS x 64 words | S rows: max_cols x LANE_S8_WORDS x 64 words
    //              | CIGAR runs of the trace: LANE_MAX_RUNS x 64 words | 64 x rest_stride bytes ]
    uint8_t *scratch;
    uint64_t wave_stride;                // bytes per wavefront (lane_wave_scratch_bytes)
    uint64_t rest_stride;                // bytes of a lane's private slice (lane_rest_bytes, rounded to 64)
    uint32_t max_cols;                   // columns a lane's scratch holds (lane_max_cols)
    uint32_t hash_slots;                 // power of two >= 2 * max_cols
    uint32_t tag_seed;                   // changes per launch (node-table entries of earlier launches read as empty)
    uint32_t t4[4];                      // score_matrix[c][q] for c, q in ACGT: row c as four bytes (q = A in the low byte)
    int32_t self_score;                  // score(A, A) == ... == score(T, T) > 0 (else the kernel is not launched)
    uint32_t *bail_list;                 // reads for the group kernel, in processing order
    unsigned long long *bail_count;
    unsigned long long *done_count;
    unsigned long long *bail_hist;       // [32] reads passed on, by the LANE_BAIL code of the test that sent them (lane_read.hpp)
};

// columns of an extension along one path: the root, one per query character, the deletions past the query's end that the x-drop
// still allows, the spare slot of the `size >= capacity - 1` test (a read that needs more goes to the group kernel)
inline uint32_t lane_max_cols(uint32_t Lmax, int32_t xdrop) {
    return Lmax + (uint32_t)std::min<int64_t>(Lmax, std::max<int64_t>(xdrop, 0)) + 8;
}

// a lane's private slice: node table | two parked windows (S, F of 32 cells each; one unused) | the forward alignment while the
// backward pass runs (nodes, character codes, CIGAR runs) | the result's scalars and counters | the merged vector of a replayed
// node | the columns that stayed behind in the frontier
inline uint64_t lane_rest_bytes(uint32_t max_cols, uint32_t hash_slots) {
    return (uint64_t)hash_slots * 8 + 2 * 2 * 32 * 4
           + (uint64_t)max_cols * 8 + LANE_MAX_RUNS * 4 + 32 * 4 + (uint64_t)(LANE_MAX_L_HOST + 8) * 4
           + (uint64_t)LANE_MAX_DEFER * LANE_DSLOT_WORDS * 4;
}
inline uint64_t lane_slots_bytes(uint32_t max_cols) { return (uint64_t)max_cols * LANE_SLOT_WORDS * LANE_WAVE * 4; }
inline uint64_t lane_s8_bytes(uint32_t max_cols) { return (uint64_t)max_cols * LANE_S8_WORDS * LANE_WAVE * 4; }
inline uint64_t lane_runs_bytes() { return (uint64_t)LANE_MAX_RUNS * LANE_WAVE * 4; }
inline uint64_t lane_wave_scratch_bytes(uint32_t max_cols, uint64_t rest_stride) {
    return lane_slots_bytes(max_cols) + lane_s8_bytes(max_cols) + lane_runs_bytes() + (uint64_t)LANE_WAVE * rest_stride;
}


// Whether the lane-per-read kernel may run in front of the group kernel for this configuration (every read it cannot finish
// goes to the group kernel anyway; this only rules out configurations in which its shortcuts would not be exact or no read
// could finish), and the scoring constants it runs with.  `why`: the first reason against.
// labeled: label-aware alignment (LabeledAligner); labeled_flags: AlignParams::labeled — bit 1 must be set (no dummy node's row holds
// a label: the lane reads rows without the W test)
inline bool lane_enabled(const mgx_config &c, const DevConfig &d, uint32_t k, uint32_t Lmax, bool no_fast, LaneParams *LP, std::string *why,
                         uint32_t labeled_flags = 0) {
    auto no = [&](const char *w) { if (why) *why = w; return false; };
    if ((labeled_flags & 1u) && !(labeled_flags & 2u)) return no("label-aware: rows of dummy nodes hold labels");
    if (d.num_alt != 1 || d.post_chain) return no("alternative paths");
    if (d.canonical != 0) return no("CANONICAL / PRIMARY graph");
    if (k > 32 || k < 2) return no("k > 32: reads are not 2-bit packed");
    if (Lmax < 1 || Lmax > (uint32_t)LANE_MAX_L_HOST) return no("reads longer than a lane takes");
    if (no_fast || c.xdrop > 30000) return no("chain path off");
    const char acgt[4] = { 'A', 'C', 'G', 'T' };
    const int m = c.score_matrix[(int)'A'][(int)'A'];
    if (m <= 0) return no("match score");
    for (int x = 0; x < 4; ++x) {
        if (c.score_matrix[(int)acgt[x]][(int)acgt[x]] != m) return no("match scores differ");
        uint32_t row = 0;
        for (int y = 0; y < 4; ++y) {
            const int v = c.score_matrix[(int)acgt[x]][(int)acgt[y]];
            if (v > m) return no("a mismatch scores above a match");
            row |= (uint32_t)(uint8_t)(int8_t)v << (8 * y);
        }
        LP->t4[x] = row;
    }
    if (c.gap_opening_penalty > 0 || c.gap_extension_penalty > 0) return no("positive gap scores");
    if (c.left_end_bonus < 0) return no("negative left end bonus");
    if (!(c.rel_score_cutoff >= 0.0 && c.rel_score_cutoff <= 1.0)) return no("rel_score_cutoff outside [0, 1]");
    LP->self_score = m;
    return true;
}

} // namespace mgx
