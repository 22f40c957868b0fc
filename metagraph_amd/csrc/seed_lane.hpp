// seed_lane.hpp — the seeding phase with ONE READ PER LANE (round 6), in front of the wave-per-read seeder of align_core.hpp
// (make_seeder / k_align<PH_SEED>): 64 reads per wavefront instruction instead of one.
//
// Why: the wave-per-read seeder spends its wavefront on one read's sequential half — the replacement rules of
// SuffixSeeder::generate_seeds, the enumeration, the aggregation are wave-uniform code, 10 k scalar + 10 k vector instructions
// per read and one dependent global round trip after the other with 8 192 reads in flight on the chip.  A lane that seeds its
// own read issues the same round trips for 64 reads at once, and the instruction count per read stops mattering.
//
// Contract: "complete or redo", as the extension's lane kernel (lane_read.hpp).  seed_lane_read() either finishes a read — its
// seeds in the lane's buffer, ready for seed_lane_publish() to write header, seed stream and work key exactly as
// align_read<PH_SEED> does — or returns SL_BAIL having written nothing, and the read goes to the wave program in its own
// launch.  What a lane takes (reads of up to SL_MAX_L characters): UniMEM seeding (max_seed_length > k; SuffixSeeder<UniMEMSeeder>, A/aligner_seeder_methods.cpp:153-358
// over :116-135 of the header) or one seed per k-mer (max_seed_length == k: ExactSeeder :67-93, every label-aware batch), with or
// without sub-k seeds, k <= 32, reads of k .. SL_MAX_L characters in ACGT only,
// plain (BASIC / CANONICAL-mode) graphs, k_map's match lengths and ranges present.  It leaves: reads whose DUST scan could
// mask anything (the exact filter runs in the wave program), more seeds or alternative nodes than its buffer or the limits
// hold, anything else unusual.
//
// The sequential half, restated for one pass over the positions (the wave program keeps per-position tables msl[] / pos_cnt[] /
// pos_full[] / pos_start[] and runs lookups, bookkeeping and aggregation as three loops; results are the same):
//  * base min_seed_length of a position: k where a MEM covers it (its k-mers' positions and the one behind them,
//    :170-188), else min_seed_length;
//  * a reporting sub-k position i with match length sl raises the positions behind it to sl, sl - 1, sl - 2, ... while that
//    exceeds what they hold, and the first failure ends the run (append_suffix_seed :195-213).  A later report inside a run
//    has sl >= the run's value there (else it would not report), so among the runs alive at a position the youngest
//    dominates and all of them end where it ends: ONE running value, decremented per position, reset by a report, dead at
//    the first position whose base value it does not exceed;
//  * positions are visited in increasing order and a position's own seeds are decided when it is visited (a MEM start never
//    reports sub-k seeds: max_len <= k - 1 < k), so the seeds can be emitted in position order as the scan goes — the order
//    the aggregation (:316-357) produces — with num_matching recounted on the way;
//  * lookups are side-effect free, so a position's longest-prefix lookup is made only if it can report (the wave program
//    looks up every position the MEM cover leaves open, side by side).
#pragma once
#include "align_core.hpp"

namespace mgx {

constexpr int SL_MAX_KMERS = 256;          // k-mer positions of a strand (four 64-bit masks)
constexpr int SL_MAX_L = 255;              // longest read a lane takes (the DUST map holds positions as bytes, 255 = none)
// packed strands in LDS, 32 codes per word and one zero word behind the last: two builds of the kernel, for batches of reads of
// up to SL_SHORT_L characters (6 words per strand: 16 wavefronts per CU) and of up to SL_MAX_L (9 words: 12 per CU)
constexpr int SL_SHORT_L = 160, SL_QWORDS_SHORT = 6, SL_QWORDS_LONG = 9;
constexpr int SL_SEED_WORDS = 3;           // a DevSeed as three words
constexpr int SL_PEND_WORDS = 6;           // a pending record (below)
// Two passes (mgx.hip): the first takes every read with a small buffer — at most SL_SEEDS_1 buffer entries, SL_PENDING_1 of
// them reporting sub-k positions, whose look-ups the lanes of a wavefront make side by side — and leaves, among others, the
// reads that need more and the reads its quick DUST scan cannot clear; the second takes what the first left with room for
// every seed a read of SL_MAX_L characters can have, the exact interval test, and the strands below min_exact_match (a
// look-up per matched k-mer).  Reads of a kind side by side again: in one launch the few heavy reads would hold up their
// 63 wave-mates for dozens of look-ups each.
constexpr int SL_SEEDS_1 = 32, SL_PENDING_1 = 8;
// (one seed per matched k-mer — max_seed_length == k, every label-aware batch: ~120 seeds per strand, a look-up per tail position)
constexpr int SL_SEEDS_1_MANY = 192, SL_PENDING_1_MANY = 32;
// (batches with reads of more than SL_SHORT_L characters: more positions)
constexpr int SL_SEEDS_1_MANY_LONG = 288, SL_SEEDS_2_LONG = 288, SL_PENDING_2_LONG = 256;
constexpr int SL_SEEDS_2 = 192, SL_PENDING_2 = 144;

// what one launch of the kernel needs on top of AlignParams
struct SeedLaneParams {
    AlignParams P;
    uint32_t *scratch;                     // per resident wavefront seed_lane_wave_scratch_words() words: word w of seed t of lane l at
                                           // ((t * SL_SEED_WORDS + w) * 64 + l); the pending records behind the seeds
    uint32_t max_entries, max_pending;     // buffer entries / pending records a lane's share of the scratch holds
    uint32_t second_pass;                  // 0: every read of the batch; 1: the reads of in_list (what the first pass left)
    uint32_t long_reads;                   // the SL_QWORDS_LONG build (mgx_launch_seed_lane picks the kernel by it)
    // The first pass lists what it leaves from both ends of one array of list_len entries: the reads its quick DUST scan could not
    // clear from the front, the others (strands below min_exact_match: dozens of look-ups each) from the back — so that the
    // second pass, which takes 64 consecutive entries per wavefront, has reads of a kind side by side.
    const uint32_t *in_list;
    const unsigned long long *in_count, *in_count_back;
    uint64_t list_len;
    uint32_t *bail_list;                   // reads for the next pass / the wave-per-read seeder
    unsigned long long *bail_count, *bail_count_back;      // (bail_count_back: null = one list, from the front)
    unsigned long long *done_count;
    unsigned long long *bail_hist;         // [16] reads passed on, by reason
};

MGX_HD uint64_t seed_lane_wave_scratch_words(uint32_t max_entries, uint32_t max_pending) {
    return ((uint64_t)max_entries * SL_SEED_WORDS + (uint64_t)max_pending * SL_PEND_WORDS + (uint64_t)((SL_MAX_L + 1) / 4)) * 64;
}

// does the batch's configuration suit the kernel at all (mgx.hip; the host model asks the same)
inline bool seed_lane_enabled(const DevConfig &d, uint32_t k, uint32_t Lmax, bool have_match_lengths, bool have_packed) {
    if (!have_match_lengths || !have_packed) return false;
    if (d.canonical >= 2) return false;                                  // PRIMARY: the wrapper seeds from both strands
    if (k > 32 || k < 3) return false;
    if (d.max_seed_length < k) return false;                             // (k-mers are not even mapped then)
    if (d.min_seed_length < 1) return false;
    if (Lmax < k) return false;
    return true;
}

// per-lane views of the on-chip arrays (LDS on the device, plain arrays in the host model)
struct SeedLaneChip {
    uint64_t *qw; int32_t qstride;         // packed strand s, word j: qw[(s * qwords + j) * qstride]
    int32_t qwords, max_l;                 // words per strand, longest read (SL_QWORDS_* / SL_SHORT_L or SL_MAX_L)
    uint32_t *sbuf; int32_t sstride;       // the lane's seed buffer: word w of seed t at sbuf[(t * SL_SEED_WORDS + w) * sstride]
    int32_t max_entries, max_pending;      // its capacities (SeedLaneParams)
    int32_t second_pass;                   // 0 first pass; second pass: 1 = a read the quick DUST scan could not clear, 2 = another
    uint32_t *cnt; int32_t cntstride;      // 64 byte counters (the DUST scans' triplet counts): byte t in word cnt[(t >> 2) * cntstride]
};
constexpr int SL_DUST_WORDS = (SL_MAX_L + 1) / 4;    // the lane's DUST map (sl_dust_map): a byte per character, behind the pending records

struct SeedLaneOut {
    int32_t L;
    int32_t n_seeds[2];                    // seeds per strand
    int32_t n_entries[2];                  // buffer entries per strand (a pending entry may stand for several seeds)
    uint32_t num_matching[2];
    LineCtr ctr;
    uint32_t reason;                       // SL_BAIL: which test sent the read on
#if MGX_SL_TIMERS
    uint64_t t0, t[8];
#endif
};

enum { SL_DONE = 0, SL_BAIL = 1 };
// -DMGX_SL_TIMERS=1: cycles per section of seed_lane_read(), summed over the wavefronts' lane 0 (measurement builds only):
// 0 strands, 1 masks, 2 the scan's own work, 3 walks, 4 the DUST scan, 5 ranges + enumeration, 6 publish, 7 waiting for the wave-mates
// -DMGX_SL_PROBE=bits: ablations for timing (WRONG results): 1 no walks, 2 no DUST scan, 4 no terminus loads, 8 masks only,
// 16 no ranges / enumeration
#ifndef MGX_SL_PROBE
#define MGX_SL_PROBE 0
#endif
#ifndef MGX_SL_TIMERS
#define MGX_SL_TIMERS 0
#endif
#if MGX_SL_TIMERS
#define SL_T(i) do { const uint64_t t_ = cycle_clock(); out.t[i] += t_ - out.t0; out.t0 = t_; } while (0)
#else
#define SL_T(i) ((void)0)
#endif
#define SL_LEAVE(code) do { out.reason = (uint32_t)(code); return SL_BAIL; } while (0)

// 64 small counters, bit-sliced over PLANES 64-bit words (counter t = bit t of every plane): the sdust scan's triplet counts
// in registers — an indexable table per lane would have to live in LDS
template <int PLANES>
struct SlicedCounters {
    uint64_t p[PLANES];
    MGX_DEV void clear() { for (int j = 0; j < PLANES; ++j) p[j] = 0; }
    MGX_DEV int32_t get(uint32_t t) const {
        int32_t v = 0;
#pragma unroll
        for (int j = 0; j < PLANES; ++j) v |= (int32_t)((p[j] >> t) & 1ull) << j;
        return v;
    }
    MGX_DEV void inc(uint32_t t) {
        uint64_t c = 1ull << t;
#pragma unroll
        for (int j = 0; j < PLANES; ++j) { const uint64_t x = p[j] & c; p[j] ^= c; c = x; }
    }
    MGX_DEV void dec(uint32_t t) {
        uint64_t c = 1ull << t;
#pragma unroll
        for (int j = 0; j < PLANES; ++j) { const uint64_t x = ~p[j] & c; p[j] ^= c; c = x; }
    }
};

// code (0 .. 3) of position pos of a packed strand
MGX_DEV uint32_t sl_code(const uint64_t *qw, int32_t qstride, int32_t pos) {
    return (uint32_t)(qw[(pos >> 5) * qstride] >> (2 * (pos & 31))) & 3u;
}

// Would sdust's scan of the whole strand (sdust_core above, l_seq = L, every character valid) ever reach find_perfect?  Only
// find_perfect adds intervals, only intervals get masked: `false` proves is_low_complexity(strand) == false, and with it —
// window_low_complexity's shortcut — that no window of the strand is low-complexity.  `true`: the wave program decides.
// The window's triplets need no queue here: every character is valid, so the window's entry idx is the triplet that ends at
// position (i - wcount + 1 + idx) of the strand.
MGX_DEV bool sl_dust_could_mask(const uint64_t *qw, int32_t qstride, int32_t L) {
    constexpr int32_t T = 20, W = 64, WLEN = 3;
    SlicedCounters<6> cw;                  // counts within the window (at most W - WLEN + 1 = 62 triplets)
    SlicedCounters<3> cv;                  // counts within its live suffix (at most 5)
    cw.clear(); cv.clear();
    int32_t rw = 0, Lw = 0, wcount = 0;
    uint32_t t = 0;
    auto tri = [&](int32_t end) -> uint32_t {                       // the triplet that ends at position `end`
        return (sl_code(qw, qstride, end - 2) << 4) | (sl_code(qw, qstride, end - 1) << 2) | sl_code(qw, qstride, end);
    };
    for (int32_t i = 0; i < L; ++i) {
        t = ((t << 2) | sl_code(qw, qstride, i)) & 63u;
        if (i + 1 < WLEN) continue;
        if (wcount >= W - WLEN + 1) {                               // shift_window
            const uint32_t sv = tri(i - wcount);
            --wcount;
            cw.dec(sv); rw -= cw.get(sv);
            if (Lw > wcount) { --Lw; cv.dec(sv); }
        }
        ++wcount; ++Lw;
        rw += cw.get(t); cw.inc(t);
        cv.inc(t);
        if (cv.get(t) * 10 > T << 1) {
            uint32_t sv;
            do { sv = tri(i - Lw + 1); cv.dec(sv); --Lw; } while (sv != t);
        }
        if (rw * 10 > Lw * T) return true;
    }
    return false;
}

// The DUST filter for one lane, exactly (second pass, the reads the quick scan could not clear).  sdust masks something in a
// string if and only if some interval of at most 62 consecutive triplets has more than T / 10 times (triplets - 1) pairs of
// equal triplets — the definition `orc_sdust_bruteforce` restates and tests/test_sdust_definition.py pins the oracle's and the
// kernels' sdust to.  So is_low_complexity(window) == "such an interval lies inside the window", and one pass over the strand
// answers every window: per end position e the start walks back (byte counters of the triplets seen, r grows by the count
// of the start's triplet) until the first — nearest — start whose interval qualifies; map[a] = the least end of a qualifying
// interval that starts at character a or later (255: none).  A window [x, x + len) is low-complexity iff map[x] <= x + len - 1.
// Intervals of the reverse complement are the mirror images (same pairs of equal triplets), so the forward strand's map
// serves both strands.
MGX_DEV uint8_t *sl_dust_byte(uint32_t *map, int32_t stride, int32_t a) { return (uint8_t *)(map + (a >> 2) * stride) + (a & 3); }
MGX_DEV_NOINLINE void sl_dust_map(const uint64_t *qw, int32_t qstride, int32_t L, uint32_t *cnt, int32_t cntstride, uint32_t *map, int32_t mstride) {
    constexpr int32_t T = 20, SPAN = 61;
    auto at = [&](uint32_t t) -> uint8_t * { return (uint8_t *)(cnt + (t >> 2) * cntstride) + (t & 3u); };
    for (int32_t x = 0; x < (L + 3) / 4; ++x) gst(map + x * mstride, 0xFFFFFFFFu);
    for (int32_t e = 2; e < L; ++e) {
#pragma unroll
        for (int x = 0; x < 16; ++x) cnt[x * cntstride] = 0;
        uint32_t t = (sl_code(qw, qstride, e - 2) << 4) | (sl_code(qw, qstride, e - 1) << 2) | sl_code(qw, qstride, e);
        *at(t) = 1;
        int32_t r = 0;
        for (int32_t d = 1; d <= SPAN && e - d >= 2; ++d) {
            t = (t >> 2) | (sl_code(qw, qstride, e - d - 2) << 4);           // the triplet that ends at e - d
            uint8_t *c = at(t);
            const uint32_t v = *c;
            r += (int32_t)v;
            *c = (uint8_t)(v + 1);
            if (r * 10 > T * d) {
                uint8_t *m = sl_dust_byte(map, mstride, e - d - 2);
                if (gld(m) == 255) gst(m, (uint8_t)e);                       // (ends come in increasing order: the first is the least)
                break;
            }
        }
    }
    uint32_t least = 255;
    for (int32_t a = L - 1; a >= 0; --a) {
        uint8_t *m = sl_dust_byte(map, mstride, a);
        least = imin<uint32_t>(least, (uint32_t)gld(m));
        gst(m, (uint8_t)least);
    }
}
// is_low_complexity(strand s, characters [x, x + len)) from the forward strand's map
MGX_DEV bool sl_dust_window(const uint32_t *map, int32_t mstride, int32_t L, int s, int32_t x, int32_t len) {
    if (s) x = L - x - len;
    if (x < 0 || len < 3) return false;
    return (int32_t)gld(sl_dust_byte(const_cast<uint32_t *>(map), mstride, x)) <= x + len - 1;
}

// BOSS::index_range (boss.hpp:720-764) on a packed strand without invalid characters: codes [i, i + len); as index_range_lane
MGX_DEV int32_t sl_index_range(const DevGraph &g, const uint64_t *qw, int32_t qstride, int32_t i, int32_t len, int32_t min_len,
                               uint64_t *first, uint64_t *last, LineCtr &ctr) {
    *first = 0; *last = 0;
    if (len == 0) { *first = 1; *last = 1; return 0; }
    uint64_t rl = 1, ru = 0;
    int32_t it = 1;
    bool have = false;
    if (g.prefix_len && (int32_t)g.prefix_len <= len) {
        const int32_t wi = i >> 5, sh = 2 * (i & 31);
        uint64_t bits = qw[wi * qstride] >> sh;
        if (sh) bits |= qw[(wi + 1) * qstride] << (64 - sh);
        const uint32_t key = (uint32_t)bits & (uint32_t)((1ull << (2 * g.prefix_len)) - 1ull);
        prefix_range(g, key, &rl, &ru, ctr);
        if (rl <= ru) { have = true; it = (int32_t)g.prefix_len; }
        else if (min_len > (int32_t)g.prefix_len) return 0;
    }
    if (!have) {
        initial_range(g, sl_code(qw, qstride, i) + 1, &rl, &ru);
        if (rl > ru) return 0;
        it = 1;
    }
    for (; it < len; ++it)
        if (!tighten_range(g, &rl, &ru, sl_code(qw, qstride, i + it) + 1, ctr)) break;
    *first = succ_last(g, rl, ctr);
    *last = ru;
    return it;
}

MGX_DEV void sl_store_seed(const SeedLaneChip &chip, int32_t t, int32_t clip, int32_t len, int32_t offset, int32_t n_nodes, uint32_t node) {
    uint32_t *d = chip.sbuf + (t * SL_SEED_WORDS) * chip.sstride;
    gst(d, (uint32_t)clip | ((uint32_t)len << 16));
    gst(d + chip.sstride, (uint32_t)offset | ((uint32_t)n_nodes << 16));
    gst(d + 2 * chip.sstride, node);
}
// A reporting sub-k position whose nodes are not known yet (a "pending" seed): a seed with n_nodes == 0 in the buffer, its
// node word = which pending record (SL_PEND_WORDS words each, behind the seeds) holds what the look-up needs
// and, afterwards, what it found: [0] kind (1 = k_map's range at slot [1], 2 = the range first [1] .. last [2] of the walk that
// was made, 3 = the walk is still to make, 4 = the matched k-mer at position [1]: the range is its node's) -> number of nodes;
// [1 .. 4] -> the nodes; [5] strand | tail position << 1 | a MEM starts at the last k-mer << 2 | its buffer entry << 8
MGX_DEV uint32_t *sl_pending(const SeedLaneChip &chip, int32_t j) {
    return chip.sbuf + (chip.max_entries * SL_SEED_WORDS + j * SL_PEND_WORDS) * chip.sstride;
}

// a mask over a strand's k-mer positions (four words held as four variables: an indexable array would live in scratch memory)
struct SlMask {
    uint64_t w0, w1, w2, w3;
    MGX_DEV void set(int32_t i) { const uint64_t b = 1ull << (i & 63); if (i < 64) w0 |= b; else if (i < 128) w1 |= b; else if (i < 192) w2 |= b; else w3 |= b; }
    MGX_DEV bool test(int32_t i) const { return (((i < 64 ? w0 : i < 128 ? w1 : i < 192 ? w2 : w3) >> (i & 63)) & 1ull) != 0; }
    MGX_DEV bool any() const { return (w0 | w1 | w2 | w3) != 0; }
    // bits_next over [0, n): the first position >= from whose bit equals val; n if none
    MGX_DEV int32_t next(int32_t from, bool val, int32_t n) const {
        const uint64_t flip = val ? 0ull : ~0ull;
        for (int32_t base = from & ~63; base < n && base < 256; base += 64) {
            uint64_t x = (base == 0 ? w0 : base == 64 ? w1 : base == 128 ? w2 : w3) ^ flip;
            if (base < from) x &= ~0ull << (from - base);
            if (x) { const int32_t p = base + ctz64(x); return p < n ? p : n; }
        }
        return n;
    }
};

// One strand: make_seeder<false> (+ strand_without_seeds, which only answers the same question sooner) up to the look-ups
// of the reporting sub-k positions, which are listed (seed_lane_read makes them for all strands and lanes side by side).
// Optimistic: a listed position is taken to report — its range exists and holds at least one node that the ":240-244" rule
// does not drop —, which is what decides the positions behind it; the look-up confirms it or the read leaves the kernel.
// Seeds go to the lane's buffer from index t0 on; *n_out = buffer entries; *nm_out as w.num_matching[s] after make_seeder.
MGX_DEV int sl_strand(const AlignParams &P, const uint64_t read_nb, const int s, const int32_t L, const int32_t n, const int32_t t0,
                      const SeedLaneChip &chip, SeedLaneOut &out, int32_t *n_out, uint32_t *nm_out, int32_t *n_pending, bool *filtered_seeds,
                      const uint32_t *dust) {
    const DevConfig &cfg = P.cfg;
    const DevGraph &g = P.g;
    const int32_t k = (int32_t)g.k;
    const uint32_t *nodes = (s ? P.nodes_rc : P.nodes_fwd) + read_nb;
    const uint8_t *mlen = (s ? P.mlen_rc : P.mlen_fwd) + read_nb;
    const uint64_t *qw = chip.qw + (s * chip.qwords) * chip.qstride;
    const int32_t qs = chip.qstride;
    LineCtr &ctr = out.ctr;
    // kmer_masks: matched k-mers, MEM stops (the terminus of a matched k-mer, inclusive, or an unmatched k-mer)
    SlMask mt = { 0, 0, 0, 0 }, sp = { 0, 0, 0, 0 };
    {
        uint32_t v = gld(nodes);
        for (int32_t i = 0; i < n; ++i) {
            const uint32_t nxt = i + 1 < n ? gld(nodes + i + 1) : 0u;
            bool stop = true;
            if (v) {
                bool term = nxt == 0;
#if !(MGX_SL_PROBE & 4)
                if (!term) { term = (gld(g.terminus + (v >> 6)) >> (v & 63)) & 1; ++ctr.bit_lines; }
#endif
                stop = term;
                mt.set(i);
            }
            if (stop) sp.set(i);
            v = nxt;
        }
    }
    SL_T(1);
    uint32_t nm0 = 0;
    {   // num_exact_matching (A/aligner_seeder_methods.cpp:49-65), run by run
        uint32_t last_match_count = 0;
        int32_t i = 0;
        while (i < n) {
            if (mt.test(i)) {
                const int32_t j = mt.next(i + 1, false, n);
                nm0 += (uint32_t)k + (uint32_t)(j - i) - 1 - last_match_count;
                last_match_count = (uint32_t)k;
                i = j;
            } else {
                const int32_t j = mt.next(i + 1, true, n);
                const uint32_t zeros = (uint32_t)(j - i);
                last_match_count = last_match_count > zeros ? last_match_count - zeros : 0;
                i = j;
            }
        }
    }
    *n_out = 0; *nm_out = nm0; *filtered_seeds = false;
    // one seed per matched k-mer (ExactSeeder, :67-93: max_seed_length <= k) instead of one per MEM; each goes through the DUST filter
    const bool many = (uint32_t)k >= cfg.max_seed_length;
#if MGX_SL_PROBE & 8
    return SL_DONE;
#endif
    if ((uint32_t)L < cfg.min_seed_length) return SL_DONE;
    const bool base_ok = !((double)nm0 < cfg.min_exact_match * (double)L);       // base_seeds: else no MEM is reported
    const int32_t max_seeds = (int32_t)P.lim.max_seeds;
    int32_t ns = 0;                                                              // buffer entries of this strand
    if (cfg.min_seed_length >= (uint32_t)k) {
        // base seeds only (UniMEMSeeder, seeder hpp:116-135; ExactSeeder)
        int32_t it = base_ok ? mt.next(0, true, n) : n;
        while (many && it < n) {
            if (dust && sl_dust_window(dust, chip.sstride, L, s, it, k)) { it = mt.next(it + 1, true, n); continue; }
            if (ns >= max_seeds) SL_LEAVE(6);
            if (t0 + ns >= chip.max_entries) SL_LEAVE(7);
            sl_store_seed(chip, t0 + ns, it, k, 0, 1, gld(nodes + it));
            ++ns;
            *filtered_seeds = true;
            it = mt.next(it + 1, true, n);
        }
        while (it < n) {
            int32_t next = sp.next(it, true, n);
            if (next < n && mt.test(next)) ++next;
            const int32_t mem_length = (next - it) + k - 1;
            if ((uint32_t)mem_length >= cfg.min_seed_length) {
                if (ns >= max_seeds) SL_LEAVE(6);
                if (t0 + ns >= chip.max_entries) SL_LEAVE(7);
                sl_store_seed(chip, t0 + ns, it, mem_length, 0, next - it, 0u);
                ++ns;
            }
            it = mt.next(next, true, n);
        }
        *n_out = ns;
        return SL_DONE;
    }
    // Matched k-mers that base_seeds does not report (num_matching below min_exact_match): every one of them becomes a
    // sub-k position with a look-up of its own (k - 1 characters) — dozens of walks per strand: the wave program's
    // (the second pass takes them: the look-up of a matched k-mer's position needs no walk — its k - 1 characters are the
    // label of the BOSS node the k-mer's edge leaves, so the range is that node's edges: kind 4)
    if (!base_ok && mt.any() && !chip.second_pass) SL_LEAVE(8);
    const int32_t msl0 = (int32_t)cfg.min_seed_length;
    const int32_t nslots = L - msl0 + 1;
    const bool tail_known = mt.test(n - 1);                // (no invalid character; plain graph)
    constexpr int32_t NO_MEM = 0x7FFFFFFF;
    auto next_mem = [&](int32_t from) -> int32_t { const int32_t p = mt.next(from, true, n); return p < n ? p : NO_MEM; };
    int32_t mem_it = base_ok && !many ? next_mem(0) : NO_MEM;      // the next MEM's first position
    int32_t cover_until = -1;                                      // positions <= this one hold k (a MEM's cover)
    int32_t run = 0;                                               // the value the running raise holds at position i (0: none)
    bool mem_at_last = false;                                      // a MEM starts at the last k-mer (:240-244)
    int32_t last_end = 0;
    uint32_t num = 0;
    auto count_matches = [&](int32_t begin, int32_t end) {         // :343-350
        if (begin < last_end) num += (uint32_t)(end - begin - (last_end - begin));
        else num += (uint32_t)(end - begin);
        last_end = end;
    };
    for (int32_t i = 0; i < nslots; ++i) {
        if (many ? (base_ok && i < n && mt.test(i) && !(dust && sl_dust_window(dust, chip.sstride, L, s, i, k))) : i == mem_it) {
            int32_t next = i + 1;
            if (!many) {
                next = sp.next(i, true, n);
                if (next < n && mt.test(next)) ++next;
            }
            const int32_t nn = next - i, mem_length = nn + k - 1;
            if (ns >= max_seeds) SL_LEAVE(6);
            if (t0 + ns >= chip.max_entries) SL_LEAVE(7);
            sl_store_seed(chip, t0 + ns, i, mem_length, 0, nn, many ? gld(nodes + i) : 0u);
            if (many) *filtered_seeds = true;
            ++ns;
            count_matches(i, i + mem_length);
            if (i == n - 1) mem_at_last = true;
            cover_until = i + nn;
            if (!many) mem_it = next_mem(next);
            run = 0;                                               // (a run holds at most k - 1 < k: it ends here)
            continue;
        }
        const int32_t base = i <= cover_until ? k : msl0;
        int32_t eff = base;
        if (run > 0) { if (run > base) eff = run; else run = 0; }
        const int32_t max_len = imin(k - 1, L - i);
        if (max_len >= eff) {
            // the position's longest-prefix lookup, as far as needed (lookup_position of make_seeder)
            int32_t ml = 0, src = 2, known_at = i;                 // src: 1 = k_map's range, 2 = walk, 3 = walk only if it reports
            if (i < n && mt.test(i)) {                     // (a matched k-mer no seed reports: see above; or its seed was masked)
                ml = max_len; src = 4;
            } else if (i < n) {                                    // (max_len == k - 1 here)
                const uint32_t c = gld(mlen + i);
                if (c == MLEN_LT_PREFIX) src = msl0 <= (int32_t)g.prefix_len ? 2 : 0;
                else if (c < MLEN_TAIL) { if ((int32_t)c >= msl0) { ml = (int32_t)c; src = 1; } else src = 0; }
            } else if (tail_known && i == n + 1 && max_len == k - 2 && gld(mlen + n - 1) == MLEN_TAIL) {
                ml = max_len; src = 1; known_at = n - 1;           // (max_len == L - i and >= msl0 hold for every tail position)
            } else if (tail_known) {
                ml = max_len; src = 3;
            }
            uint64_t first = 0, last = 0;
            SL_T(2);
#if MGX_SL_PROBE & 1
            if (src == 2 || src == 3) { src = 0; ml = 0; }
#endif
            if (src == 2) {
                const int32_t m = sl_index_range(g, qw, qs, i, max_len, msl0, &first, &last, ctr);
                ml = (m >= msl0 && first && first <= g.n) ? m : 0;
            }
            SL_T(3);
#if MGX_SL_PROBE & 16
            const bool reports = false;
#else
            // (the complexity filter looks at the eff characters the position must match, :226-229: a masked window reports nothing)
            const bool reports = src && ml && ml >= eff && !(dust && sl_dust_window(dust, chip.sstride, L, s, i, eff));
#endif
            if (reports) {
                // listed; taken to report (see above)
                const int32_t j = *n_pending;
                if (j >= chip.max_pending) SL_LEAVE(7);
                if (ns >= max_seeds) SL_LEAVE(6);
                if (t0 + ns >= chip.max_entries) SL_LEAVE(7);
                sl_store_seed(chip, t0 + ns, i, ml, k - ml, 0, (uint32_t)j);
                ++ns;
                uint32_t *pd = sl_pending(chip, j);
                gst(pd, (uint32_t)src);
                gst(pd + chip.sstride, src == 1 ? (uint32_t)known_at : src == 4 ? (uint32_t)i : (uint32_t)first);
                gst(pd + 2 * chip.sstride, (uint32_t)last);
                // ":240-244": a tail position whose one node is the last k-mer's, which a seed of its own reports, adds nothing — what
                // the scan expects of every tail position behind such a seed (with one seed per k-mer: of all of them)
                const bool expect_dup = i >= n && mem_at_last;
                gst(pd + 5 * chip.sstride, (uint32_t)s | (i >= n ? 2u : 0u) | (mem_at_last ? 4u : 0u) | (expect_dup ? 8u : 0u) | ((uint32_t)(t0 + ns - 1) << 8));
                *n_pending = j + 1;
                if (!expect_dup) {
                    count_matches(i, i + ml);
                    run = ml + 1;                                  // append_suffix_seed: the positions behind take ml, ml - 1, ... (one off below)
                }
            }
        }
        if (run > 0) --run;
    }
    SL_T(2);
    *n_out = ns;
    *nm_out = num;
    return SL_DONE;
}

// One read: build_seeders of align_read<PH_SEED>.  SL_DONE: out is complete and the seeds are in the lane's buffer (strand 0's
// first; a pending seed stands for the nodes of its record).  SL_BAIL: nothing was written anywhere but the lane's own buffer.
MGX_DEV int seed_lane_read(const AlignParams &P, const uint64_t read, const SeedLaneChip &chip, SeedLaneOut &out) {
    const DevConfig &cfg = P.cfg;
    const DevGraph &g = P.g;
    const int32_t k = (int32_t)g.k;
    out.ctr.rank_lines = out.ctr.select_lines = out.ctr.bit_lines = 0;
    out.n_seeds[0] = out.n_seeds[1] = 0; out.num_matching[0] = out.num_matching[1] = 0;
    out.n_entries[0] = out.n_entries[1] = 0;
    const uint64_t off = gld(P.offsets + read);
    const int32_t L = (int32_t)(gld(P.offsets + read + 1) - off);
    out.L = L;
    if (L > (int32_t)P.lim.Lmax || L > chip.max_l || L < k) SL_LEAVE(1);
    const uint64_t nb = gld(P.node_begin + read);
    const int32_t n = (int32_t)(gld(P.node_begin + read + 1) - nb);
    if (n != L - k + 1 || n > SL_MAX_KMERS) SL_LEAVE(2);
    const bool have_rc = cfg.fwd_and_rc != 0;
    {
        const uint64_t wb = packed_word_begin(off, read);
        const int32_t nw = (L + 31) >> 5;
        uint32_t inv = 0;
        for (int s = 0; s < (have_rc ? 2 : 1); ++s)
            for (int32_t j = 0; j < chip.qwords; ++j) {
                uint64_t v = 0;
                if (j < nw) { v = gld(P.pkw[s] + wb + j); inv |= gld(P.ivw[s] + wb + j); }
                chip.qw[(s * chip.qwords + j) * chip.qstride] = v;
            }
        if (inv) SL_LEAVE(3);
    }
    SL_T(0);
    // second pass (the reads the first pass's quick DUST scan could not clear are among these): the exact filter, as a map of the
    // read that answers every window the scan asks about
    uint32_t *dust = nullptr;
    if (cfg.seed_complexity_filter && chip.second_pass) {
        dust = chip.sbuf + (chip.max_entries * SL_SEED_WORDS + chip.max_pending * SL_PEND_WORDS) * chip.sstride;
        sl_dust_map(chip.qw, chip.qstride, L, chip.cnt, chip.cntstride, dust, chip.sstride);
    }
    SL_T(4);
    int32_t n_pending = 0;
    int32_t pend_of[2] = { 0, 0 };                                 // pending seeds per strand
    bool filtered[2] = { false, false };                           // the strand has seeds the DUST filter looked at (one per k-mer)
    for (int s = 0; s < (have_rc ? 2 : 1); ++s) {
        int32_t ne = 0;
        uint32_t nm = 0;
        bool fs = false;
        const int32_t p0 = n_pending;
        if (sl_strand(P, nb, s, L, n, out.n_entries[0], chip, out, &ne, &nm, &n_pending, &fs, dust) != SL_DONE) return SL_BAIL;
        if ((double)L * cfg.min_exact_match > (double)nm) { ne = 0; nm = 0; n_pending = p0; }      // (its look-ups: not needed)
        filtered[s] = fs;
        out.n_entries[s] = ne; out.num_matching[s] = nm;
        pend_of[s] = n_pending - p0;
    }
    // the DUST filter of the strands that report sub-k seeds (the scans of a wavefront's lanes run side by side here)
    if (cfg.seed_complexity_filter && !chip.second_pass) {
        for (int s = 0; s < 2; ++s) {
            bool masked = false;
#if !(MGX_SL_PROBE & 2)
            if (pend_of[s] || filtered[s]) masked = sl_dust_could_mask(chip.qw + (s * chip.qwords) * chip.qstride, chip.qstride, L);
#endif
            if (masked) SL_LEAVE(4);
        }
    }
    // (second pass: the scans above asked the exact filter's map)
    SL_T(4);
    // the look-ups of the listed positions, the j-th of every lane side by side: range, nodes (dbg_succinct.cpp:349-392: the
    // parents of every node of the range)
    const int32_t msl0 = (int32_t)cfg.min_seed_length;
    uint32_t alt_total[2] = { 0, 0 };
    bool tail_reported[2] = { false, false };
    int32_t extra[2] = { 0, 0 };                                   // seeds a strand's pending entries expand to, minus one each
    for (int32_t j = 0; j < n_pending; ++j) {
        uint32_t *pd = sl_pending(chip, j);
        const uint32_t kind = gld(pd), a1 = gld(pd + chip.sstride), a2 = gld(pd + 2 * chip.sstride), fl = gld(pd + 5 * chip.sstride);
        const int s = (int)(fl & 1u);
        const int32_t t = (int32_t)(fl >> 8);                          // its buffer entry
        if (tail_reported[s]) {                                        // (behind a tail position that reports nothing can: below)
            gst(pd, 0u);
            extra[s] -= 1;
            continue;
        }
        const uint32_t w0 = gld(chip.sbuf + (t * SL_SEED_WORDS) * chip.sstride);
        const int32_t i = (int32_t)(w0 & 0xFFFFu), ml = (int32_t)(w0 >> 16);
        uint64_t first = 0, last = 0;
        bool hit = true;
        if (kind == 1) {
            const uint2 r = gld((s ? P.rng_rc : P.rng_fwd) + nb + a1);
            ++out.ctr.bit_lines;
            first = succ_last(g, r.x, out.ctr);
            last = r.y;
            hit = first && first <= g.n;
        } else if (kind == 2) {
            first = a1; last = a2;
        } else if (kind == 4) {
            first = last = succ_last(g, (uint64_t)gld((s ? P.nodes_rc : P.nodes_fwd) + nb + a1), out.ctr);
            hit = first && first <= g.n;
        } else {
            const int32_t m = sl_index_range(g, chip.qw + (s * chip.qwords) * chip.qstride, chip.qstride, i, ml, msl0, &first, &last, out.ctr);
            hit = m >= msl0 && first && first <= g.n;
        }
        if (!hit) SL_LEAVE(9);
        int32_t cnt = 0;
        const uint32_t r_begin = first == last ? 0u : rank_last(g, first, out.ctr);
        const uint32_t r_end = first == last ? 0u : rank_last(g, last, out.ctr);
        for (uint32_t r = r_begin; r <= r_end; ++r) {
            const uint64_t e = first == last ? first : select_last(g, r, out.ctr);
            uint64_t inc[5];
            uint32_t fc[5];
            const int ni = incoming<false, false>(g, e, inc, fc, out.ctr);
            for (int x = 0; x < ni; ++x) {
                if (cnt >= 4) SL_LEAVE(13);
                gst(pd + (1 + cnt) * chip.sstride, (uint32_t)inc[x]);
                ++cnt;
            }
        }
        if (cnt == 0) SL_LEAVE(12);
        if ((uint32_t)cnt > cfg.max_num_seeds_per_locus) SL_LEAVE(13);
        // ":240-244": the one node of a tail position is the last k-mer's node, which a seed of its own reports — no seed then,
        // as the scan expected or not
        const bool dup = (fl & 2u) && cnt == 1 && (fl & 4u) && gld(pd + chip.sstride) == gld((s ? P.nodes_rc : P.nodes_fwd) + nb + n - 1);
        if (dup && !(fl & 8u)) SL_LEAVE(14);
        // A tail position the scan expected to add nothing, and it does report (several nodes end with these characters, or
        // another one): its seeds stay, and the positions behind it — all listed, as expected non-reporters — are raised above
        // what they could match (position i + 1 takes L - i, and has L - i - 1 characters), so none of them reports.  Its
        // matches add nothing to num_matching: the seed at the last k-mer already ends at L.
        if (!dup && (fl & 8u)) tail_reported[s] = true;
        if (dup) cnt = 0;
        alt_total[s] += (uint32_t)cnt;
        if (alt_total[s] > P.lim.max_alt) SL_LEAVE(5);
        gst(pd, (uint32_t)cnt);
        extra[s] += cnt - 1;
    }
    SL_T(5);
    for (int s = 0; s < 2; ++s) {
        out.n_seeds[s] = out.n_entries[s] + extra[s];
        if (out.n_seeds[s] > (int32_t)P.lim.max_seeds) SL_LEAVE(6);
    }
    return SL_DONE;
}

// seed x (expanded numbering: a pending entry counts for its nodes) of the lane's buffer, as three words
struct SlCursor { int32_t t, sub; };
MGX_DEV void sl_next_seed(const SeedLaneChip &chip, SlCursor &c, uint32_t *w) {
    const uint32_t *pd;
    int32_t cnt;
    for (;; ++c.t) {                                               // (a pending entry without nodes stands for no seed)
        const uint32_t *e = chip.sbuf + (c.t * SL_SEED_WORDS) * chip.sstride;
        w[0] = gld(e); w[1] = gld(e + chip.sstride); w[2] = gld(e + 2 * chip.sstride);
        if ((w[1] >> 16) != 0) { ++c.t; return; }
        pd = sl_pending(chip, (int32_t)w[2]);
        cnt = (int32_t)gld(pd);
        if (cnt > 0) break;
    }
    w[1] |= 1u << 16;
    w[2] = gld(pd + (1 + c.sub) * chip.sstride);
    if (++c.sub >= cnt) { c.sub = 0; ++c.t; }
}

// What align_read<PH_SEED> publishes for a finished read: header, seeds (at `so` of the seed stream, handed out by the caller:
// one atomic per wavefront on the device), work key, the seed dump of the test hook.
MGX_DEV void seed_lane_publish(const AlignParams &P, const uint64_t read, const SeedLaneChip &chip, const SeedLaneOut &out, const uint64_t so) {
    const uint32_t total = (uint32_t)(out.n_seeds[0] + out.n_seeds[1]);
    int32_t status = ST_OK;
    uint64_t h_off = 0;
    int32_t first_clip[2] = { 0, 0 };
    {
        const bool room = total && so + total <= P.seed_capacity;
        if (total) { h_off = so; if (!room) status = ST_CAPACITY; }
        uint32_t *dst = (uint32_t *)(P.seed_stream + so);
        SlCursor c = { 0, 0 };
        for (int s = 0; s < 2; ++s)
            for (int32_t x = 0; x < out.n_seeds[s]; ++x) {
                uint32_t w[3];
                sl_next_seed(chip, c, w);
                if (x == 0) first_clip[s] = (int32_t)(w[0] & 0xFFFFu);
                if (room) { gst(dst, w[0]); gst(dst + 1, w[1]); gst(dst + 2, w[2]); dst += 3; }
                if (P.dbg_seeds) {
                    uint32_t *d = (uint32_t *)(P.dbg_seeds + ((uint64_t)read * 2 + s) * P.lim.max_seeds + x);
                    gst(d, w[0]); gst(d + 1, w[1]); gst(d + 2, w[2]);
                }
            }
    }
    SeedHdr *h = P.seed_hdr + read;
    gst(&h->off, h_off);
    gst(&h->num_matching[0], out.num_matching[0]); gst(&h->num_matching[1], out.num_matching[1]);
    gst(&h->n_seeds[0], (uint16_t)out.n_seeds[0]); gst(&h->n_seeds[1], (uint16_t)out.n_seeds[1]);
    gst(&h->status, status); gst(&h->pad, 0u);
    // predicted_work
    uint32_t work = 0;
    if (status == ST_OK) {
        const int first = out.num_matching[0] >= out.num_matching[1] ? 0 : 1;
        if (out.n_seeds[first]) {
            const int32_t clip = first_clip[first];
            const int32_t cols = (out.L - clip) + (clip > 0 ? out.L : 0);
            work = 1u + (uint32_t)imin(4094, cols);
        }
    }
    gst(P.work_key + read, work | ((uint32_t)(read >> WORK_SEGMENT_SHIFT) << 12));
}

#undef SL_LEAVE

} // namespace mgx
