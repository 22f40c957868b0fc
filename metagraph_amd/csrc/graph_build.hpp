// graph_build.hpp — per-index building blocks for turning a BOSS view (W, last, F) into the device
// layout of dev_graph.hpp, and the lane-per-chain k-mer mapping routine.  Each function is pure
// per-thread code: the HIP kernels in kernels.hip call them with one thread per index.
#pragma once
#include "align_core.hpp"

namespace mgx {

// pass 1: bit-planes, last bits and per-block counts (counts[6*b + c]: c = 0..4 unflagged labels, 5 = last)
MGX_DEV void build_block_pass1(uint32_t b, const uint8_t *W, const uint8_t *last, uint64_t n, Block *blocks, uint32_t *counts) {
    Block blk;
    blk.cum[0] = blk.cum[1] = blk.cum[2] = blk.cum[3] = 0;
    blk.last_cum = 0; blk.cum0 = 0;
    uint64_t p0 = 0, p1 = 0, p2 = 0, pf = 0, lb = 0;
    uint32_t cnt[6] = { 0, 0, 0, 0, 0, 0 };
    for (int j = 0; j < 64; ++j) {
        uint64_t i = ((uint64_t)b << 6) + (uint32_t)j;
        if (i > n) break;
        uint32_t w = W[i];
        uint32_t c = w % SIGMA;
        bool flagged = w >= SIGMA;
        if (c & 1) p0 |= 1ull << j;
        if (c & 2) p1 |= 1ull << j;
        if (c & 4) p2 |= 1ull << j;
        if (flagged) pf |= 1ull << j;
        if (i >= 1) {
            if (!flagged) ++cnt[c];
            if (last[i]) { lb |= 1ull << j; ++cnt[5]; }
        }
    }
    blk.p0 = p0; blk.p1 = p1; blk.p2 = p2; blk.pf = pf; blk.last_bits = lb;
    blocks[b] = blk;
    for (int c = 0; c < 6; ++c) counts[6 * (uint64_t)b + c] = cnt[c];
}

// pass 2: cum[6*b + c] holds exclusive prefix sums of the counts; fills cumulative fields and hints
MGX_DEV void build_block_pass2(uint32_t b, Block *blocks, const uint32_t *cum, uint32_t *last_hint,
                               uint32_t *const *w_hint) {
    Block blk = blocks[b];
    const uint32_t *cb = cum + 6 * (uint64_t)b;
    blk.cum0 = cb[0];
    for (int c = 1; c < SIGMA; ++c) blk.cum[c - 1] = cb[c];
    blk.last_cum = cb[5];
    blocks[b] = blk;
    // hint[(r - 1) / 64] = block of the r-th occurrence for every r with (r - 1) % 64 == 0
    {
        uint32_t cnt = (uint32_t)popc64(blk.last_bits);
        uint32_t first_r = cb[5] + 1, last_r = cb[5] + cnt;
        for (uint32_t h = (first_r - 1 + 63) / 64; cnt && h * 64 + 1 <= last_r; ++h) last_hint[h] = b;
    }
    for (uint32_t c = 1; c < SIGMA; ++c) {
        uint64_t m = code_mask(blk, c) & ~blk.pf;
        if (b == 0) m &= ~1ull;
        uint32_t cnt = (uint32_t)popc64(m);
        uint32_t first_r = cb[c] + 1, last_r = cb[c] + cnt;
        for (uint32_t h = (first_r - 1 + 63) / 64; cnt && h * 64 + 1 <= last_r; ++h) w_hint[c - 1][h] = b;
    }
}

// the select anchors (dev_graph.hpp sel_anchor): entry j, one thread per entry.  The last segment is shorter than 2^shift
// ranks: its closing entry is extrapolated so that the segment's slope is right.
MGX_HD uint32_t sel_anchor_shift(uint64_t total_last, uint32_t max_entries) {
    uint32_t s = 6;
    while ((total_last >> s) + 2 > max_entries) ++s;
    return s;
}
MGX_DEV void build_sel_anchor(const DevGraph &g, uint32_t j, uint32_t shift, uint32_t n_entries, uint32_t total_last, uint32_t *out) {
    LineCtr ctr = { 0, 0, 0 };
    const uint64_t r = (uint64_t)j << shift;
    if (r == 0) { out[j] = 0; return; }
    if (r <= total_last) { out[j] = (uint32_t)select_last(g, (uint32_t)r, ctr); return; }
    // the closing entry (and padding behind it)
    const uint64_t r0 = (uint64_t)(j - 1) << shift;
    if (j + 1 < n_entries || r0 > total_last) { out[j] = (uint32_t)g.n; return; }
    const uint64_t p0 = r0 ? select_last(g, (uint32_t)r0, ctr) : 0, p1 = select_last(g, total_last, ctr);
    const uint64_t span = total_last - r0;                     // ranks of the last segment
    if (span == 0) { out[j] = (uint32_t)g.n; return; }         // (total_last a multiple of 2^shift: no rank lies in this segment)
    const uint64_t ext = p0 + (((p1 - p0) << shift) + span - 1) / span;
    out[j] = (uint32_t)(ext > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : ext);
}

// parent pointer used to propagate first characters: P[e] = bwd(e) (boss.cpp:623-636)
MGX_DEV uint32_t build_parent(const DevGraph &g, uint64_t e) {
    if (e == 0) return 0;
    LineCtr ctr = { 0, 0, 0 };
    return (uint32_t)bwd(g, e, ctr);
}

// Suffix keys for the prefix table: after round r, D[e] is the node character r positions from the end.
// key[e] accumulates (D - 1) << 2 * (m - 1 - r); bit 31 marks a '$' inside the last m characters.
MGX_DEV uint32_t build_key_step(uint32_t key, uint32_t d, uint32_t m, uint32_t r) {
    if (d == 0) return key | 0x80000000u;
    return key | ((d - 1) << (2 * (m - 1 - r)));
}

// table boundaries: edges with equal keys are contiguous (co-lex order)
MGX_DEV void build_prefix_entry(const uint32_t *key, uint64_t e, uint64_t n, uint2 *tbl) {
    uint32_t k0 = key[e];
    if (k0 & 0x80000000u) return;
    if (e == 1 || key[e - 1] != k0) tbl[k0].x = (uint32_t)e;
    if (e == n || key[e + 1] != k0) tbl[k0].y = (uint32_t)e;
}

MGX_HD uint32_t choose_prefix_len(uint64_t n_edges, uint32_t k, uint32_t cap = 14) {
    // ~log4(n) + 1 characters resolve a range to O(1) nodes, so that a failed suffix lookup costs one table
    // line instead of a chain of tighten_range steps; the caller caps m by the free HBM (15 = 8.6 GB is the largest
    // the 32-bit keys allow; measured on the bench graph: m = 13 / 14 / 15 -> 1.87 / 1.93 / 1.96 M reads/s)
    uint32_t m = 2;
    while (m < cap && (1ull << (2 * (m - 1))) < n_edges) ++m;
    if (m > k - 1) m = k - 1;
    return m;
}

// CanonicalDBG::map_to_nodes_sequentially (canonical_dbg.cpp:55-146) from the base graph's mappings of both strands: a k-mer
// found forward keeps its id, one found only in the reverse complement gets that node's id + n; the reverse strand's path is
// the mirror image (DeBruijnGraph::reverse_complement_seq_path -> CanonicalDBG::reverse_complement, :551-560).  One call
// rewrites position i of the forward path and position n_kmers - 1 - i of the reverse path in place.
MGX_DEV void canon_merge_pair(const DevGraph &g, const char *seq, int32_t L, int32_t i, uint32_t *nf, uint32_t *nr) {
    const int32_t k = (int32_t)g.k, nk = L - k + 1;
    const uint32_t a = nf[i], b = nr[nk - 1 - i];
    uint32_t f = 0, r = 0;
    if (a) {
        bool pal = !(k & 1);
        for (int32_t j = 0; pal && j < k / 2; ++j)
            pal = encode_char((uint8_t)seq[i + j]) + encode_char((uint8_t)seq[i + k - 1 - j]) == 5;
        f = a; r = pal ? a : a + (uint32_t)g.n;
    } else if (b) {
        f = b + (uint32_t)g.n; r = b;
    }
    nf[i] = f; nr[nk - 1 - i] = r;
}

// MEM terminus bit: has_multiple_outgoing(v) || !has_single_incoming(v) (aligner_seeder_methods.hpp:121-125)
MGX_DEV bool build_terminus(const DevGraph &g, uint64_t v) {
    if (v == 0 || v > g.n) return false;
    LineCtr ctr = { 0, 0, 0 };
    return has_multiple_outgoing(g, v, ctr) || !has_single_incoming(g, v, ctr);
}

// ------------------------------------------------------------------------------------------------
// The same state machine over 2-bit packed reads (k <= 32).  A streaming pre-pass (pack_read_word, one thread per
// 32 bases) writes every strand as 64-bit words of 32 codes (A C G T = 0..3) plus one invalid-character flag per base;
// word j of read r lives at index (offsets[r] >> 5) + r + j.  A lane keeps the codes of positions [i, i + 32) in one
// register pair (`cur`), shifts two bits per k-mer and tops up from the next word every 32 k-mers: every character
// index(), tighten_range, fwd and pick_edge look at (positions i .. i + k - 1) is a shift away, where the byte path
// issues a global load, an alphabet switch and a complement per character.
// ------------------------------------------------------------------------------------------------
// (packed_word_begin: align_core.hpp, next to prepare_query, which reads these words too)

MGX_DEV void pack_read_word(const char *seq, int32_t L, int strand, int32_t j, uint64_t *codes, uint32_t *inv) {
    uint64_t c = 0;
    uint32_t v = 0;
    for (int32_t t = 0; t < 32; ++t) {
        const int32_t pos = 32 * j + t;
        if (pos >= L) break;
        const uint32_t code = strand_code(seq, L, strand, pos);
        if (code == 5 || code == 0) v |= 1u << t;
        else c |= (uint64_t)(code - 1) << (2 * t);
    }
    *codes = c;
    *inv = v;
}

struct MapLanePacked {
    const uint64_t *pk;       // packed codes of this chain's strand, word 0 = positions 0..31
    const uint32_t *iv;       // invalid flags, same indexing
    uint32_t *out;
    uint8_t *out_len;
    uint2 *out_rng;
    int32_t min_rng_len;
    int32_t n_words, n_kmers;
    int32_t i;                // next k-mer position
    int32_t t;
    uint64_t cur, nxt;        // codes of positions [i, i + 32); the not yet consumed codes of the word after them
    uint32_t icur, inxt;      // the same for the invalid flags
    uint64_t edge, rl, ru;
    Block blk;
    int state;
};

MGX_DEV uint32_t packed_code(const MapLanePacked &m, int32_t off) { return (uint32_t)((m.cur >> (2 * off)) & 3) + 1; }

MGX_DEV void packed_begin(MapLanePacked &m) {
    m.cur = gld(m.pk); m.icur = gld(m.iv);
    const int32_t j1 = m.n_words > 1 ? 1 : 0;
    m.nxt = gld(m.pk + j1); m.inxt = gld(m.iv + j1);
}

// i -> i + 1
MGX_DEV void packed_advance(MapLanePacked &m) {
    m.cur = (m.cur >> 2) | ((m.nxt & 3) << 62);
    m.icur = (m.icur >> 1) | ((m.inxt & 1) << 31);
    m.nxt >>= 2; m.inxt >>= 1;
    ++m.i;
    if ((m.i & 31) == 0) {                               // position i + 32 starts word (i >> 5) + 1
        const int32_t j = imin((m.i >> 5) + 1, m.n_words - 1);
        m.nxt = gld(m.pk + j); m.inxt = gld(m.iv + j);
    }
}

template <class FetchChain>
MGX_DEV void map_lane_step_packed(const DevGraph &g, MapLanePacked &m, LineCtr &ctr, FetchChain fetch) {
    const int32_t k = (int32_t)g.k;                      // <= 32
    if (m.state == 0) {
        if (!fetch(m)) { m.state = 3; return; }
        m.i = 0; m.edge = 0;
        if (m.n_kmers > 0) { packed_begin(m); m.state = 1; }
        return;
    }
    if (m.state == 1) {
        if (m.i >= m.n_kmers) { m.state = 0; return; }
        const int32_t i = m.i;
        const uint32_t kmask = k >= 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);
        if (m.icur & kmask) { gst_stream(m.out + i, 0); m.edge = 0; packed_advance(m); return; }
        if (m.edge) {
            Block tgt;
            uint64_t lst = fwd_from(g, m.edge, m.blk, packed_code(m, k - 2), tgt, ctr);
            m.blk = tgt;
            m.edge = lst ? pick_edge_from(g, lst, m.blk, packed_code(m, k - 1), ctr) : 0;
            gst_stream(m.out + i, in_graph(g, m.edge) ? (uint32_t)m.edge : 0);
            packed_advance(m);
            return;
        }
        int32_t t0 = 1;
        if (g.prefix_len && (int32_t)g.prefix_len <= k - 1) {
            const uint32_t key = (uint32_t)(m.cur & ((1ull << (2 * g.prefix_len)) - 1ull));
            prefix_range(g, key, &m.rl, &m.ru, ctr);
            t0 = (int32_t)g.prefix_len;
        } else {
            initial_range(g, packed_code(m, 0), &m.rl, &m.ru);
        }
        if (m.rl > m.ru) {
            if (m.out_len && t0 > 1 && k - 1 < MLEN_LT_PREFIX) gst_stream(m.out_len + i, MLEN_LT_PREFIX);
            gst_stream(m.out + i, 0); m.edge = 0; packed_advance(m);
            return;
        }
        m.t = t0;
        m.state = 2;
        return;
    }
    const int32_t i = m.i;
    if (m.t < k - 1) {
        if (!tighten_range(g, &m.rl, &m.ru, packed_code(m, m.t), ctr)) {
            if (m.out_len && k - 1 < MLEN_LT_PREFIX) {
                gst_stream(m.out_len + i, (uint8_t)m.t);
                if (m.out_rng && m.t >= m.min_rng_len) m.out_rng[i] = make_uint2((uint32_t)m.rl, (uint32_t)m.ru);
            }
            gst_stream(m.out + i, 0); m.edge = 0; packed_advance(m); m.state = 1;
            return;
        }
        ++m.t;
        return;
    }
    ++ctr.rank_lines;
    m.blk = load_block(g, (uint32_t)(m.ru >> 6));
    m.edge = pick_edge_from(g, m.ru, m.blk, packed_code(m, k - 1), ctr);
    gst_stream(m.out + i, in_graph(g, m.edge) ? (uint32_t)m.edge : 0);
    if (!m.edge && m.out_len && k - 1 < MLEN_LT_PREFIX) {
        gst_stream(m.out_len + i, (uint8_t)(k - 1));
        if (m.out_rng && k - 1 >= m.min_rng_len) m.out_rng[i] = make_uint2((uint32_t)m.rl, (uint32_t)m.ru);
    }
    packed_advance(m);
    m.state = 1;
}

} // namespace mgx
