// dev_graph.hpp — the BOSS table laid out for HBM gathers, and its query primitives.
//
// One 64-byte block holds everything a BOSS step needs for 64 consecutive edges: cumulative
// ranks of the four unflagged labels, the cumulative rank of `last`, the `last` bits and W as
// four bit-planes.  `fwd` = one rank in the current block (usually already loaded) + one hint
// lookup (small, cache resident) + one block load at the target; `pick_edge`, `pred_last` and
// the out-edge labels are answered from that same target block.
//
// Semantics follow boss::BOSS (M/src/graph/representation/succinct/boss.{hpp,cpp}) and DBGSuccinct
// (dbg_succinct.cpp); every function cites the lines it restates.  Results are bit-identical to
// rank/select on the plain W/last arrays because rank/select are mathematically defined.
#pragma once
#include "wave.hpp"

namespace mgx {

constexpr int SIGMA = 5;                       // "$ACGT" (kmer/alphabets.hpp:64)

struct alignas(64) Block {
    uint32_t cum[4];       // # of unflagged W == c (c = 1..4) in edges of earlier blocks
    uint32_t last_cum;     // # of set `last` bits in earlier blocks
    uint32_t cum0;         // # of unflagged W == 0 in earlier blocks (slot 0 excluded)
    uint64_t last_bits;
    uint64_t p0, p1, p2;   // label code bits (W % 5)
    uint64_t pf;           // flag: W >= 5 ("not the first edge into its target", boss.hpp W_)
};
static_assert(sizeof(Block) == 64, "one block = one 64-byte line");

struct DevGraph {
    const Block *blocks;
    const uint32_t *last_hint;     // block holding every 64-th set bit of last
    const uint32_t *w_hint[4];     // same for unflagged W == 1..4
    const uint32_t *sel_anchor;    // select_last(j << sel_shift), j = 0 .. sel_n - 1 (positions; the last entry closes the last segment):
    uint32_t sel_shift, sel_n;     // a table small enough for LDS from which the block of select_last(r) is PREDICTED by linear
                                   // interpolation (sel_predict); select_last_scan then scans from the predicted block in
                                   // whichever direction r lies, so the prediction is a hint, not a promise — fwd becomes one
                                   // dependent load (+ the occasional neighbour) where last_hint costs a fetch of its own first
    const uint32_t *firstc;        // first character code of every edge's k-mer, 8 nibbles per word
    const uint64_t *terminus;      // MEM-terminus bit per node (aligner_seeder_methods.hpp:121-125), n_blocks words; PRIMARY
                                   // graphs: n_blocks more words with the bits of the wrapper ids v + n (canon_graph.hpp)
    const uint64_t *valid;         // node mask or nullptr (dbg_succinct.cpp:934-936)
    const uint2 *prefix_tbl;       // [4^prefix_len] edge range (rl, ru) of nodes whose suffix spells the key;
                                   // the device form of BOSS's suffix-range index (boss.hpp:645-663, boss.cpp:3177-3219)
    uint32_t prefix_len;           // 0 = no table
    uint64_t n;                    // number of edges
    uint32_t n_blocks;
    uint32_t k;                    // DBG k
    uint32_t F[SIGMA];
    uint32_t NF[SIGMA];
};

struct LineCtr { uint32_t rank_lines, select_lines, bit_lines; };

// U = true: the address is wave-uniform -> scalar load (SMEM), the block lands in SGPRs
template <bool U>
MGX_DEV Block load_block_t(const DevGraph &g, uint32_t b);

MGX_DEV Block load_block(const DevGraph &g, uint32_t b) {
    const uint4 *p = reinterpret_cast<const uint4 *>(g.blocks + b);
    uint4 a0 = gld(p), a1 = gld(p + 1), a2 = gld(p + 2), a3 = gld(p + 3);
    Block r;
    r.cum[0] = a0.x; r.cum[1] = a0.y; r.cum[2] = a0.z; r.cum[3] = a0.w;
    r.last_cum = a1.x; r.cum0 = a1.y;
    r.last_bits = ((uint64_t)a1.w << 32) | a1.z;
    r.p0 = ((uint64_t)a2.y << 32) | a2.x;
    r.p1 = ((uint64_t)a2.w << 32) | a2.z;
    r.p2 = ((uint64_t)a3.y << 32) | a3.x;
    r.pf = ((uint64_t)a3.w << 32) | a3.z;
    return r;
}

MGX_DEV Block load_block_uniform(const DevGraph &g, uint32_t b) {
    u32x16 v = sload_x16(g.blocks + b);
    Block r;
    r.cum[0] = v[0]; r.cum[1] = v[1]; r.cum[2] = v[2]; r.cum[3] = v[3];
    r.last_cum = v[4]; r.cum0 = v[5];
    r.last_bits = ((uint64_t)v[7] << 32) | v[6];
    r.p0 = ((uint64_t)v[9] << 32) | v[8];
    r.p1 = ((uint64_t)v[11] << 32) | v[10];
    r.p2 = ((uint64_t)v[13] << 32) | v[12];
    r.pf = ((uint64_t)v[15] << 32) | v[14];
    return r;
}
MGX_DEV Block uni_block(const Block &b) {
    Block r;
    for (int i = 0; i < 4; ++i) r.cum[i] = uni(b.cum[i]);
    r.last_cum = uni(b.last_cum); r.cum0 = uni(b.cum0);
    r.last_bits = uni(b.last_bits); r.p0 = uni(b.p0); r.p1 = uni(b.p1); r.p2 = uni(b.p2); r.pf = uni(b.pf);
    return r;
}
template <> MGX_DEV Block load_block_t<false>(const DevGraph &g, uint32_t b) { return load_block(g, b); }
template <> MGX_DEV Block load_block_t<true>(const DevGraph &g, uint32_t b) { return load_block_uniform(g, b); }
template <bool U> MGX_DEV uint32_t load_hint(const uint32_t *p) { if constexpr (U) return sload_u32(p); else return gld(p); }

// bits j <= pos
MGX_DEV uint64_t mask_upto(int pos) { return pos >= 63 ? ~0ull : ((1ull << (pos + 1)) - 1); }

// positions in the block whose label code is c (flag ignored)
MGX_DEV uint64_t code_mask(const Block &b, uint32_t c) {
    uint64_t m0 = (c & 1) ? b.p0 : ~b.p0;
    uint64_t m1 = (c & 2) ? b.p1 : ~b.p1;
    uint64_t m2 = (c & 4) ? b.p2 : ~b.p2;
    return m0 & m1 & m2;
}

MGX_DEV uint32_t block_W(const Block &b, int j) {
    uint32_t c = (uint32_t)((b.p0 >> j) & 1) | ((uint32_t)((b.p1 >> j) & 1) << 1) | ((uint32_t)((b.p2 >> j) & 1) << 2);
    return c + (((b.pf >> j) & 1) ? SIGMA : 0);
}

// position (0-based) of the r-th (1-based) set bit of x; requires popc(x) >= r
MGX_DEV int select64(uint64_t x, int r) {
    int pos = 0;
    uint32_t w = (uint32_t)x;
    int c = __builtin_popcount(w);
    if (r > c) { r -= c; pos = 32; w = (uint32_t)(x >> 32); }
    c = __builtin_popcount(w & 0xFFFFu);
    if (r > c) { r -= c; pos += 16; w >>= 16; }
    w &= 0xFFFFu;
    c = __builtin_popcount(w & 0xFFu);
    if (r > c) { r -= c; pos += 8; w >>= 8; }
    w &= 0xFFu;
    c = __builtin_popcount(w & 0xFu);
    if (r > c) { r -= c; pos += 4; w >>= 4; }
    w &= 0xFu;
    c = __builtin_popcount(w & 0x3u);
    if (r > c) { r -= c; pos += 2; w >>= 2; }
    w &= 0x3u;
    if (r > (int)(w & 1u)) pos += 1;
    return pos;
}

MGX_DEV bool in_graph(const DevGraph &g, uint64_t v) {              // dbg_succinct.cpp:934-936
    if (v == 0 || v > g.n) return false;
    return !g.valid || ((gld(g.valid + (v >> 6)) >> (v & 63)) & 1);       // (a global load, not a FLAT one)
}

template <bool U = false>
MGX_DEV uint32_t get_W(const DevGraph &g, uint64_t i, LineCtr &ctr) {
    ++ctr.rank_lines;
    Block b = load_block_t<U>(g, (uint32_t)(i >> 6));
    return block_W(b, (int)(i & 63));
}

// cumulative count of label c before the block; selects instead of a dynamically indexed array so that a Block
// held in registers stays there (a runtime index would force the whole struct into scratch memory)
MGX_DEV uint32_t block_cum(const Block &b, uint32_t c) {
    uint32_t lo = (c & 1) ? b.cum[0] : b.cum0;            // c = 1 : 0
    uint32_t mid = (c & 1) ? b.cum[2] : b.cum[1];         // c = 3 : 2
    uint32_t r = (c & 2) ? mid : lo;
    return c >= 4 ? b.cum[3] : r;
}

// rank of unflagged c in W[1..i] within a loaded block (boss.cpp:437-441); c in 0..4
MGX_DEV uint32_t block_rank_W(const Block &b, int j, uint32_t c, bool first_block) {
    uint64_t m = code_mask(b, c) & ~b.pf & mask_upto(j);
    if (c == 0 && first_block) m &= ~1ull;          // slot 0 is not an edge ("- (c == 0)")
    return block_cum(b, c) + (uint32_t)popc64(m);
}

template <bool U = false>
MGX_DEV uint32_t rank_W(const DevGraph &g, uint64_t i, uint32_t c, LineCtr &ctr) {
    if (i == 0) return 0;
    ++ctr.rank_lines;
    Block b = load_block_t<U>(g, (uint32_t)(i >> 6));
    return block_rank_W(b, (int)(i & 63), c, (i >> 6) == 0);
}

template <bool U = false>
MGX_DEV uint32_t rank_last(const DevGraph &g, uint64_t i, LineCtr &ctr) {       // boss.cpp:577-581
    if (i == 0) return 0;
    ++ctr.rank_lines;
    Block b = load_block_t<U>(g, (uint32_t)(i >> 6));
    return b.last_cum + (uint32_t)popc64(b.last_bits & mask_upto((int)(i & 63)));
}

// select_last (boss.cpp:588-592): also returns the block that holds the answer
template <bool U = false>
MGX_DEV uint64_t select_last_blk(const DevGraph &g, uint32_t r, Block &b, LineCtr &ctr) {
    uint32_t bi = load_hint<U>(g.last_hint + ((r - 1) >> 6));
    for (;;) {
        ++ctr.select_lines;
        b = load_block_t<U>(g, bi);
        uint32_t c = (uint32_t)popc64(b.last_bits);
        if (b.last_cum + c >= r) break;
        ++bi;
    }
    return ((uint64_t)bi << 6) + (uint32_t)select64(b.last_bits, (int)(r - b.last_cum));
}

// the block select_last(r) is expected in, r >= 1: linear interpolation between the anchors around r (tab = g.sel_anchor or a
// copy of it in faster memory)
MGX_DEV uint32_t sel_predict(const uint32_t *tab, uint32_t shift, uint32_t r) {
    const uint32_t j = r >> shift, lo = tab[j], hi = tab[j + 1];
    const uint32_t frac = r & ((1u << shift) - 1u);
    return (lo + (uint32_t)(((uint64_t)(hi - lo) * frac) >> shift)) >> 6;
}

// select_last(r), r >= 1, by a scan from block `bi` in whichever direction r lies; also returns the block of the answer
template <bool U = false>
MGX_DEV uint64_t select_last_scan(const DevGraph &g, uint32_t r, uint32_t bi, Block &b, LineCtr &ctr) {
    for (;;) {
        ++ctr.select_lines;
        b = load_block_t<U>(g, bi);
        if (b.last_cum >= r) { --bi; continue; }                      // (block 0 has last_cum 0 < r)
        if (b.last_cum + (uint32_t)popc64(b.last_bits) >= r) break;
        ++bi;
    }
    return ((uint64_t)bi << 6) + (uint32_t)select64(b.last_bits, (int)(r - b.last_cum));
}

template <bool U = false>
MGX_DEV uint64_t select_last(const DevGraph &g, uint32_t r, LineCtr &ctr) {
    if (r == 0) return 0;
    Block b;
    return select_last_blk<U>(g, r, b, ctr);
}

// position of the r-th unflagged c in W (wavelet_tree::select as used by boss.cpp:635); c in 1..4
template <bool U = false>
MGX_DEV uint64_t select_W(const DevGraph &g, uint32_t c, uint32_t r, LineCtr &ctr) {
    const uint32_t *wh = c == 1 ? g.w_hint[0] : c == 2 ? g.w_hint[1] : c == 3 ? g.w_hint[2] : g.w_hint[3];     // (as nf_of: no runtime index into g)
    uint32_t bi = load_hint<U>(wh + ((r - 1) >> 6));
    for (;;) {
        ++ctr.select_lines;
        Block b = load_block_t<U>(g, bi);
        uint64_t m = code_mask(b, c) & ~b.pf;
        uint32_t cnt = (uint32_t)popc64(m);
        if (block_cum(b, c) + cnt >= r)
            return ((uint64_t)bi << 6) + (uint32_t)select64(m, (int)(r - block_cum(b, c)));
        ++bi;
    }
}

// last set bit of `last` in [1..i], 0 if none (boss.cpp:598-607); blk = loaded block of i
template <bool U = false>
MGX_DEV uint64_t pred_last_from(const DevGraph &g, uint64_t i, const Block &blk, LineCtr &ctr) {
    if (i == 0) return 0;
    uint64_t m = blk.last_bits & mask_upto((int)(i & 63));
    uint32_t bi = (uint32_t)(i >> 6);
    Block b;
    while (!m) {
        if (bi == 0) return 0;
        --bi;
        ++ctr.rank_lines;
        b = load_block_t<U>(g, bi);
        m = b.last_bits;
    }
    return ((uint64_t)bi << 6) + (uint32_t)(63 - clz64(m));
}

MGX_DEV uint64_t pred_last(const DevGraph &g, uint64_t i, LineCtr &ctr) {
    if (i == 0) return 0;
    ++ctr.rank_lines;
    Block b = load_block(g, (uint32_t)(i >> 6));
    return pred_last_from(g, i, b, ctr);
}

// first set bit of `last` in [i..n] (boss.cpp:613-617); n + 1 if none
MGX_DEV uint64_t succ_last(const DevGraph &g, uint64_t i, LineCtr &ctr) {
    uint32_t bi = (uint32_t)(i >> 6);
    ++ctr.rank_lines;
    Block b = load_block(g, bi);
    uint64_t m = b.last_bits & ~(mask_upto((int)(i & 63)) >> 1);      // bits >= i&63
    while (!m) {
        ++bi;
        if (bi >= g.n_blocks) return g.n + 1;
        ++ctr.rank_lines;
        b = load_block(g, bi);
        m = b.last_bits;
    }
    return ((uint64_t)bi << 6) + (uint32_t)ctz64(m);
}

MGX_DEV uint32_t node_last_value(const DevGraph &g, uint64_t i) {       // boss.cpp:679-690
    if (i == 0) return 0;
    for (uint32_t c = 0; c < SIGMA; ++c)
        if (g.F[c] >= i) return c - 1;
    return SIGMA - 1;
}

// NF[c] for a per-lane c.  g lives in the kernel-argument segment: `g.NF[c]` with a runtime index is a LOAD from it (a
// dependent round trip in front of every select hint); with constant indices the five values are scalar registers.
MGX_DEV uint32_t nf_of(const DevGraph &g, uint32_t c) {
    const uint32_t lo = (c & 1) ? g.NF[1] : g.NF[0];
    const uint32_t mid = (c & 1) ? g.NF[3] : g.NF[2];
    const uint32_t r = (c & 2) ? mid : lo;
    return c >= 4 ? g.NF[4] : r;
}
MGX_DEV uint32_t f_of(const DevGraph &g, uint32_t c) {
    const uint32_t lo = (c & 1) ? g.F[1] : g.F[0];
    const uint32_t mid = (c & 1) ? g.F[3] : g.F[2];
    const uint32_t r = (c & 2) ? mid : lo;
    return c >= 4 ? g.F[4] : r;
}

// fwd(i, c) = select_last(NF[c] + rank_W(i, c)) (boss.cpp:642-652); cur = loaded block of i.
// Returns the target's last edge and its block.
template <bool U = false>
MGX_DEV uint64_t fwd_from(const DevGraph &g, uint64_t i, const Block &cur, uint32_t c, Block &tgt, LineCtr &ctr) {
    uint32_t r = nf_of(g, c) + block_rank_W(cur, (int)(i & 63), c, (i >> 6) == 0);
    if (r == 0) { tgt = cur; return 0; }
    return select_last_blk<U>(g, r, tgt, ctr);
}

MGX_DEV uint64_t fwd(const DevGraph &g, uint64_t i, uint32_t c, LineCtr &ctr) {
    ++ctr.rank_lines;
    Block cur = load_block(g, (uint32_t)(i >> 6));
    Block tgt;
    return fwd_from(g, i, cur, c, tgt, ctr);
}

// pick_edge (boss.cpp:710-722): scan the node's edges backwards from its last edge for label c
// (flagged or not).  blk = loaded block of `edge`; on return blk is the block of the result.
MGX_DEV uint64_t pick_edge_from(const DevGraph &g, uint64_t edge, Block &blk, uint32_t c, LineCtr &ctr) {
    for (;;) {
        int j = (int)(edge & 63);
        uint32_t w = block_W(blk, j);
        if (w == c || w == c + SIGMA) return edge;
        --edge;
        if (edge == 0) return 0;
        if ((edge & 63) == 63) { ++ctr.rank_lines; blk = load_block(g, (uint32_t)(edge >> 6)); }
        if ((blk.last_bits >> (edge & 63)) & 1) return 0;
    }
}

// bwd (boss.cpp:623-636)
template <bool U = false>
MGX_DEV uint64_t bwd(const DevGraph &g, uint64_t i, LineCtr &ctr) {
    uint32_t target_node = rank_last<U>(g, i - 1, ctr) + 1;
    if (target_node == 1) return 1;
    uint32_t c = node_last_value(g, i);
    return select_W<U>(g, c, target_node - nf_of(g, c), ctr);
}

// tighten_range (boss.hpp:682-693).  Narrow ranges are the common case, so the two ranks share one block
// load when rl - 1 and ru fall into the same block, and the two selects share the block of the upper one.
MGX_DEV bool tighten_range(const DevGraph &g, uint64_t *rl, uint64_t *ru, uint32_t s, LineCtr &ctr) {
    const uint64_t lo = *rl - 1, hi = *ru;
    ++ctr.rank_lines;
    Block bh = load_block(g, (uint32_t)(hi >> 6));
    uint32_t rk_ru = hi ? block_rank_W(bh, (int)(hi & 63), s, (hi >> 6) == 0) : 0;
    uint32_t rk_rl;
    if (lo == 0) {
        rk_rl = 1;
    } else if ((lo >> 6) == (hi >> 6)) {
        rk_rl = block_rank_W(bh, (int)(lo & 63), s, (lo >> 6) == 0) + 1;
    } else {
        ++ctr.rank_lines;
        Block bl = load_block(g, (uint32_t)(lo >> 6));
        rk_rl = block_rank_W(bl, (int)(lo & 63), s, (lo >> 6) == 0) + 1;
    }
    if (rk_rl > rk_ru) return false;
    const uint32_t nfs = nf_of(g, s);
    const uint32_t r_hi = nfs + rk_ru, r_lo = nfs + rk_rl - 1;
    Block sb;
    const uint64_t pos_hi = select_last_blk(g, r_hi, sb, ctr);
    *ru = pos_hi;
    if (r_lo == 0) *rl = 1;
    else if (r_lo > sb.last_cum) *rl = ((pos_hi >> 6) << 6) + (uint32_t)select64(sb.last_bits, (int)(r_lo - sb.last_cum)) + 1;
    else *rl = select_last(g, r_lo, ctr) + 1;
    return true;
}

// get_initial_range through the suffix-range table (boss.hpp:645-663): codes[0..prefix_len) must all be
// in 1..4.  Key = first char least significant, exactly the reference's co-lex index.
MGX_DEV void prefix_range(const DevGraph &g, uint32_t key, uint64_t *rl, uint64_t *ru, LineCtr &ctr) {
    ++ctr.bit_lines;
    // the table is far larger than any cache and every entry is read once per lookup: stream it past the caches
    const uint64_t r = gld_stream_u64(g.prefix_tbl + key);
    *rl = (uint32_t)r; *ru = (uint32_t)(r >> 32);
}

MGX_DEV void initial_range(const DevGraph &g, uint32_t s, uint64_t *rl, uint64_t *ru) {   // boss.hpp:665-677
    const uint64_t fs = f_of(g, s);
    *rl = fs + 1 < g.n + 1 ? fs + 1 : g.n + 1;
    *ru = s + 1 < SIGMA ? (uint64_t)f_of(g, s + 1) : g.n;
}

// Children of node v as DBGSuccinct::call_outgoing_kmers reports them (dbg_succinct.cpp:110-139),
// minus sentinel-labelled children which the extender discards (aligner_extender_methods.cpp:381-384).
// Writes up to 4 (node, label code) pairs in edge order; returns the count.
// *sentinel (optional) = v has a sentinel-labelled child that is in the graph.
MGX_DEV int outgoing(const DevGraph &g, uint64_t v, uint64_t *nodes, uint32_t *codes, LineCtr &ctr, bool *sentinel = nullptr) {
    ++ctr.rank_lines;
    Block cur = load_block(g, (uint32_t)(v >> 6));
    uint32_t w = block_W(cur, (int)(v & 63));
    if (v > 1 && w == 0) return 0;
    Block tgt;
    uint64_t lst = fwd_from(g, v, cur, w % SIGMA, tgt, ctr);
    uint64_t first = pred_last_from(g, lst - 1, ((lst - 1) >> 6) == (lst >> 6) ? tgt : load_block(g, (uint32_t)((lst - 1) >> 6)), ctr) + 1;
    if (first < 2) first = 2;
    int n = 0;
    Block b = tgt;
    uint32_t bi = (uint32_t)(lst >> 6);
    for (uint64_t i = first; i <= lst; ++i) {
        if ((uint32_t)(i >> 6) != bi) { bi = (uint32_t)(i >> 6); ++ctr.rank_lines; b = load_block(g, bi); }
        uint32_t c = block_W(b, (int)(i & 63)) % SIGMA;
        if (c != 0 && in_graph(g, i)) { if (n < 4) { nodes[n] = i; codes[n] = c; } ++n; }
        if (c == 0 && sentinel && in_graph(g, i)) *sentinel = true;
    }
    return n < 4 ? n : 4;
}

MGX_DEV uint32_t first_char(const DevGraph &g, uint64_t e, LineCtr &ctr) {
    ++ctr.bit_lines;
    return (gld(g.firstc + (e >> 3)) >> ((e & 7) * 4)) & 0xF;
}

// Parents of v with the first character of each parent k-mer, in the order of
// BOSS::call_incoming_to_target (boss.cpp:766-786) as used by NodeFirstCache::call_incoming_kmers
// (graph_extensions/node_first_cache.cpp:38-52).  Up to 5 parents ($ACGT first chars).
// FIRST = false: the caller only wants the parent nodes (the seeder's suffix matches); the first-character table is then
// not read at all (one random line per parent).
template <bool U = false, bool FIRST = true>
MGX_DEV int incoming(const DevGraph &g, uint64_t v, uint64_t *nodes, uint32_t *first_codes, LineCtr &ctr) {
    uint64_t x = bwd<U>(g, v, ctr);
    uint32_t d = node_last_value(g, v);
    int n = 0;
    if (in_graph(g, x)) { nodes[n] = x; first_codes[n] = FIRST ? first_char(g, x, ctr) : 0u; ++n; }
    // edges after x labelled d + SIGMA, up to the next unflagged d
    uint64_t pos = x + 1;
    uint32_t bi = (uint32_t)(pos >> 6);
    while (pos <= g.n) {
        ++ctr.rank_lines;
        Block b = load_block_t<U>(g, bi);
        uint64_t from = ~(mask_upto((int)(pos & 63)) >> 1);          // bits >= pos & 63
        uint64_t cm = code_mask(b, d) & from;
        if (bi == g.n_blocks - 1 && ((g.n + 1) & 63)) cm &= mask_upto((int)(g.n & 63));
        uint64_t stop = cm & ~b.pf;
        uint64_t flg = cm & b.pf;
        if (stop) flg &= mask_upto(ctz64(stop));
        while (flg) {
            int j = ctz64(flg);
            flg &= flg - 1;
            uint64_t e = ((uint64_t)bi << 6) + (uint32_t)j;
            if (in_graph(g, e) && n < 5) { nodes[n] = e; first_codes[n] = FIRST ? first_char(g, e, ctr) : 0u; ++n; }
        }
        if (stop) break;
        ++bi;
        pos = (uint64_t)bi << 6;
    }
    return n;
}

// the same parents, written as 32-bit ids to out[0 .. cap) (memory of the caller's choosing); returns their number, which may
// exceed cap (then only the first cap were written)
MGX_DEV int incoming_nodes32(const DevGraph &g, uint64_t v, uint32_t *out, int cap, LineCtr &ctr) {
    const uint64_t x = bwd(g, v, ctr);
    const uint32_t d = node_last_value(g, v);
    int n = 0;
    if (in_graph(g, x)) { if (n < cap) gst(out + n, (uint32_t)x); ++n; }
    uint64_t pos = x + 1;
    uint32_t bi = (uint32_t)(pos >> 6);
    while (pos <= g.n) {
        ++ctr.rank_lines;
        const Block b = load_block(g, bi);
        const uint64_t from = ~(mask_upto((int)(pos & 63)) >> 1);
        uint64_t cm = code_mask(b, d) & from;
        if (bi == g.n_blocks - 1 && ((g.n + 1) & 63)) cm &= mask_upto((int)(g.n & 63));
        const uint64_t stop = cm & ~b.pf;
        uint64_t flg = cm & b.pf;
        if (stop) flg &= mask_upto(ctz64(stop));
        while (flg) {
            const int j = ctz64(flg);
            flg &= flg - 1;
            const uint64_t e = ((uint64_t)bi << 6) + (uint32_t)j;
            if (in_graph(g, e)) { if (n < cap) gst(out + n, (uint32_t)e); ++n; }
        }
        if (stop) break;
        ++bi;
        pos = (uint64_t)bi << 6;
    }
    return n;
}

// has_multiple_outgoing (dbg_succinct.cpp:609-624)
MGX_DEV bool has_multiple_outgoing(const DevGraph &g, uint64_t v, LineCtr &ctr) {
    if (v == 1) return succ_last(g, 1, ctr) > 2;
    uint32_t d = get_W(g, v, ctr) % SIGMA;
    if (!d) return false;
    uint64_t t = fwd(g, v, d, ctr) - 1;
    ++ctr.rank_lines;
    Block b = load_block(g, (uint32_t)(t >> 6));
    return !((b.last_bits >> (t & 63)) & 1);
}

// succ_W(i, a, a + SIGMA) (boss.cpp:515-570): first position >= i with label code d; returns
// position (n + 1 if none) and whether it is flagged
MGX_DEV uint64_t succ_W_code(const DevGraph &g, uint64_t i, uint32_t d, bool *flagged, LineCtr &ctr) {
    uint32_t bi = (uint32_t)(i >> 6);
    uint64_t from = ~(mask_upto((int)(i & 63)) >> 1);
    while (bi < g.n_blocks) {
        ++ctr.rank_lines;
        Block b = load_block(g, bi);
        uint64_t cm = code_mask(b, d) & from;
        if (bi == g.n_blocks - 1 && ((g.n + 1) & 63)) cm &= mask_upto((int)(g.n & 63));
        if (cm) {
            int j = ctz64(cm);
            *flagged = (b.pf >> j) & 1;
            return ((uint64_t)bi << 6) + (uint32_t)j;
        }
        ++bi;
        from = ~0ull;
    }
    *flagged = false;
    return g.n + 1;
}

// has_single_incoming (dbg_succinct.cpp:658-678) incl. BOSS::is_single_incoming (boss.cpp:802-815)
// and num_incoming_to_target (boss.cpp:821-838)
MGX_DEV bool has_single_incoming(const DevGraph &g, uint64_t v, LineCtr &ctr) {
    if (v == 1) return false;
    uint64_t x = bwd(g, v, ctr);
    uint32_t w = node_last_value(g, v);
    bool first_valid = !g.valid || ((gld(g.valid + (x >> 6)) >> (x & 63)) & 1);
    if (x + 1 == g.n + 1) return first_valid;
    bool flagged;
    if (first_valid) {
        // is_single_incoming(x, w): W[x] == w < SIGMA here (x is the first incoming edge)
        succ_W_code(g, x + 1, w, &flagged, ctr);
        return !flagged;
    }
    // num_incoming_to_target(x, w) == 2
    uint64_t p = succ_W_code(g, x + 1, w, &flagged, ctr);
    if (!flagged) return false;                       // exactly 1
    if (p + 1 > g.n) return true;                     // exactly 2
    succ_W_code(g, p + 1, w, &flagged, ctr);
    return !flagged;
}

} // namespace mgx
