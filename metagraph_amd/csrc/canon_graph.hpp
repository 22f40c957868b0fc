// canon_graph.hpp — the CanonicalDBG wrapper over a PRIMARY-mode BOSS graph (graph/representation/canonical_dbg.cpp) as
// device functions: a PRIMARY graph stores one k-mer of every {k-mer, reverse complement} pair; the wrapper presents both.
//
// Node space (canonical_dbg.cpp:17-53): ids 1 .. n are the base graph's nodes, id v + n is the reverse complement of base
// node v (offset_ = max_index of the base graph = n).  A palindromic k-mer (even k only) has no second id.
//
// Traversal (call_outgoing_kmers :156-240, call_incoming_kmers :242-330, the DBGSuccinct branches of
// adjacent_outgoing_rc_strand / adjacent_incoming_rc_strand :574-684 with NodeFirstCache::get_prefix_rc / get_suffix_rc,
// node_first_cache.cpp:120-174) reduces to one rule.  Let h be the spelling of wrapper node v.  Its children spell
// h[1:] + a and come from two places in the base graph:
//   A: the edges (label a) leaving the BOSS node h[1:]            -> child id = that edge (a base id),
//   B: the parents p (first character c) of the BOSS node RC(h[1:]) -> child id = p + n (p itself if the child is a
//      palindrome), a = complement(c).
// For a base id v, A is read off v directly (DBGSuccinct::call_outgoing_kmers, ascending edge order) and comes first, B needs
// an index_range look-up of RC(h[1:]); for an id v = u + n, B is read off u directly (call_incoming_kmers of u) and comes first,
// A needs the look-up of h[1:] and is reported in DESCENDING edge order (BOSS::call_outgoing walks back from the node's last
// edge, boss.hpp:779-784).  Entries of the second set whose character the first one already reported are skipped.  A
// sentinel anywhere in h[1:] makes the looked-up set empty (index_range rejects the sentinel code).  Parents are the children
// of the reverse complement: call_incoming_kmers(v) == call_outgoing_kmers(v + n) mirrored, which is how the MEM terminus
// bits of both ids of a base node are computed at load time.
//
// canon_children() derives h (k - 1 bwd steps) and the look-up (k - 2 tighten_range steps) for every expansion and holds the
// spelling in three 64-bit registers, hence k <= 64 for PRIMARY graphs; canon_children_tables() further down reads both from
// tables built at load time and is what the kernels use by default.
#pragma once
#include "dev_graph.hpp"

namespace mgx {

// spelling of a node (k <= 64): character j (0 = first) as A C G T = 0..3 in two bits of a 128-bit code, word j / 32, bits
// [2 (j % 32), +2); bit j of `dollar` marks a sentinel
struct Spell { uint64_t code[2], dollar; };

MGX_DEV void spell_put(Spell &s, int32_t j, uint32_t c /* 0..4 */) {
    if (c == 0) s.dollar |= 1ull << j;
    else if (j < 32) s.code[0] |= (uint64_t)(c - 1) << (2 * j);
    else s.code[1] |= (uint64_t)(c - 1) << (2 * (j - 32));
}
MGX_DEV uint32_t spell_get(const Spell &s, int32_t j) {
    if ((s.dollar >> j) & 1) return 0u;
    return (uint32_t)(((j < 32 ? s.code[0] : s.code[1]) >> (2 * (j & 31))) & 3) + 1u;
}

// get_node_sequence of a base node (dbg_succinct.cpp:272-279 -> BOSS::get_node_seq boss.cpp:940-973 + the edge label)
MGX_DEV Spell base_spelling(const DevGraph &g, uint64_t e, LineCtr &ctr) {
    const int32_t k = (int32_t)g.k;
    Spell s = { { 0, 0 }, 0 };
    spell_put(s, k - 1, get_W(g, e, ctr) % SIGMA);
    uint64_t x = e;
    for (int32_t j = k - 2; j >= 0; --j) {
        spell_put(s, j, node_last_value(g, x));
        if (j) x = bwd(g, x, ctr);
    }
    return s;
}

// reverse complement of the first `len` characters
MGX_DEV Spell spell_reverse_complement(const Spell &s, int32_t len) {
    Spell r = { { 0, 0 }, 0 };
    for (int32_t j = 0; j < len; ++j) {
        const uint32_t c = spell_get(s, len - 1 - j);
        spell_put(r, j, c ? 5u - c : 0u);                    // complement('$') == '$'
    }
    return r;
}

// characters 1 .. k - 1 (the node every child starts from)
MGX_DEV Spell spell_tail(const Spell &s, int32_t k) {
    Spell r = { { 0, 0 }, 0 };
    for (int32_t j = 0; j + 1 < k; ++j) spell_put(r, j, spell_get(s, j + 1));
    return r;
}

// is the k-mer its own reverse complement?  (even k, no sentinel)
MGX_DEV bool kmer_is_palindrome(const Spell &s, int32_t k) {
    if ((k & 1) || s.dollar) return false;
    for (int32_t j = 0; j < k / 2; ++j)
        if (spell_get(s, j) + spell_get(s, k - 1 - j) != 5) return false;
    return true;
}

// last edge of the BOSS node spelled by the first k - 1 characters of `t` (no sentinel among them), 0 if there is none:
// BOSS::index_range (boss.hpp:720-764) demanding a full match, as get_prefix_rc / get_suffix_rc do
MGX_DEV uint64_t index_boss_node(const DevGraph &g, const Spell &t, LineCtr &ctr) {
    const int32_t len = (int32_t)g.k - 1;
    uint64_t rl = 1, ru = 0;
    int32_t it = 1;
    if (g.prefix_len && (int32_t)g.prefix_len <= len) {
        const uint32_t key = (uint32_t)(t.code[0] & ((1ull << (2 * g.prefix_len)) - 1));      // prefix_len <= 16
        prefix_range(g, key, &rl, &ru, ctr);
        if (rl > ru) return 0;
        it = (int32_t)g.prefix_len;
    } else {
        initial_range(g, spell_get(t, 0), &rl, &ru);
        if (rl > ru) return 0;
    }
    for (; it < len; ++it)
        if (!tighten_range(g, &rl, &ru, spell_get(t, it), ctr)) return 0;
    return ru;
}

// The representative of node v's k-mer on a CANONICAL-mode graph, as DBGSuccinct::map_to_nodes defines it
// (dbg_succinct.cpp:436-481: "the definition of a canonical k-mer is redefined: use k-mer with smaller index in the BOSS table"):
// the smaller of v and the edge of its reverse complement (BOSS::map_to_edges: the node of the first k - 1 characters, then the
// edge with the last character's label), validated; 0 (npos) for a k-mer with a sentinel, a missing reverse complement or a
// masked edge.  This is the node whose row an annotation of a CANONICAL graph holds (AnnotationBuffer, annotation_buffer.cpp:56-62).
MGX_DEV uint32_t canon_repr_node(const DevGraph &g, uint64_t v) {
    LineCtr ctr = { 0, 0, 0 };
    const int32_t k = (int32_t)g.k;
    if (v < 1 || v > g.n) return 0;
    const Spell h = base_spelling(g, v, ctr);
    if (h.dollar & ((k >= 64 ? 0ull : (1ull << k)) - 1)) return 0;
    const Spell r = spell_reverse_complement(h, k);
    const uint64_t ru = index_boss_node(g, r, ctr);
    if (!ru) return 0;
    const uint32_t c = spell_get(r, k - 1);
    const uint64_t first = pred_last(g, ru - 1, ctr) + 1;
    uint64_t e = 0;
    for (uint64_t i = first; i <= ru && !e; ++i) if (get_W(g, i, ctr) % SIGMA == c) e = i;
    if (!e) return 0;
    const uint64_t m = e < v ? e : v;
    return in_graph(g, m) ? (uint32_t)m : 0u;
}

// CanonicalDBG::call_outgoing_kmers(v) for wrapper node v with spelling h, sentinel-labelled children left out (what the
// extender keeps, aligner_extender_methods.cpp:381-384).  Writes up to 4 (node, code 1..4) pairs in the reference's callback
// order; *sentinel = the base graph reported a sentinel neighbour on the direct side (children[0] / parents[0] of the
// reference, which only its degree counts look at).
MGX_DEV int canon_children(const DevGraph &g, uint32_t v, const Spell &h, uint32_t *nodes, uint8_t *codes, bool *sentinel, LineCtr &ctr) {
    const int32_t k = (int32_t)g.k;
    const uint64_t off = g.n;
    const bool is_rc = v > off;
    uint32_t have = 0;                                       // bit a: a child with code a was reported
    int n = 0;
    *sentinel = false;
    const Spell tail = spell_tail(h, k);                      // h[1:], k - 1 characters
    auto child_id_b = [&](uint64_t p, uint32_t a) -> uint32_t {
        // reverse_complement(p) (:515-549): p + n unless the k-mer (== the child h[1:] + a mirrored) is a palindrome
        Spell child = tail;
        spell_put(child, k - 1, a);
        return kmer_is_palindrome(child, k) ? (uint32_t)p : (uint32_t)(p + off);
    };
    const bool tail_clean = (h.dollar >> 1) == 0;            // no sentinel in h[1:]
    uint64_t nn[5];
    uint32_t cc[5];
    if (!is_rc) {
        const int m = outgoing(g, v, nn, cc, ctr, sentinel);                       // set A, direct
        for (int t = 0; t < m; ++t) { nodes[n] = (uint32_t)nn[t]; codes[n] = (uint8_t)cc[t]; have |= 1u << cc[t]; ++n; }
        if (n == 4 || !tail_clean) return n;
        // set B: parents of the node RC(h[1:])
        const uint64_t e = index_boss_node(g, spell_reverse_complement(tail, k - 1), ctr);
        if (!e) return n;
        const int mi = incoming(g, e, nn, cc, ctr);
        for (int t = 0; t < mi; ++t) {
            if (cc[t] == 0) continue;
            const uint32_t a = 5u - cc[t];
            if (have & (1u << a)) continue;                  // the other strand's copy of a palindrome
            nodes[n] = child_id_b(nn[t], a); codes[n] = (uint8_t)a; have |= 1u << a; ++n;
        }
        return n;
    }
    {
        const uint64_t u = v - off;
        const int mi = incoming(g, u, nn, cc, ctr);                                // set B, direct
        for (int t = 0; t < mi; ++t) {
            if (cc[t] == 0) { *sentinel = true; continue; }
            const uint32_t a = 5u - cc[t];
            nodes[n] = child_id_b(nn[t], a); codes[n] = (uint8_t)a; have |= 1u << a; ++n;
        }
        if (n == 4 || !tail_clean) return n;
        // set A: the edges of the node h[1:], from its last edge backwards
        uint64_t e = index_boss_node(g, tail, ctr);
        if (!e) return n;
        ++ctr.rank_lines;
        Block b = load_block(g, (uint32_t)(e >> 6));
        uint32_t bi = (uint32_t)(e >> 6);
        do {
            if ((uint32_t)(e >> 6) != bi) { bi = (uint32_t)(e >> 6); ++ctr.rank_lines; b = load_block(g, bi); }
            const uint32_t a = block_W(b, (int)(e & 63)) % SIGMA;
            if (a != 0 && in_graph(g, e) && !(have & (1u << a)) && n < 4) {
                nodes[n] = (uint32_t)e; codes[n] = (uint8_t)a; have |= 1u << a; ++n;
            }
            --e;
            if (!e) break;
            if ((uint32_t)(e >> 6) != bi) { bi = (uint32_t)(e >> 6); ++ctr.rank_lines; b = load_block(g, bi); }
        } while (!((b.last_bits >> (e & 63)) & 1));
        return n;
    }
}

// ------------------------------------------------------------------------------------------------
// The same traversal from two precomputed tables instead of spellings and look-ups (DevConfig::canonical == 3, the default).
//
// Everything canon_children derives from the spelling is a property of a BOSS node of the BASE graph: the looked-up node is
// RC(X) for X = the target node of v (base id) or the source node of u (id u + n), and the palindrome test of a B-set child is
// the palindrome test of the parent edge's own k-mer.  So a table with the last edge of RC(X) for every BOSS node X (0: no such
// node, or X holds a sentinel; 4 bytes per node) and one palindrome bit per edge (even k) replace k - 1 bwd steps and k - 2
// tighten_range steps per expansion by one table line.  The reference reaches the same end with LRU caches
// (canonical_dbg.hpp:121-137); with 288 GB of HBM the table is the cheaper trade: it roughly doubles the 3.5 B/edge index —
// still what a CANONICAL-mode graph of the same data costs.  Both live behind DevGraph::terminus (see primary_tables()).
// ------------------------------------------------------------------------------------------------
struct PrimaryTables {
    const uint64_t *pal;          // bit e: the k-mer of edge e is its own reverse complement (all zero for odd k)
    const uint32_t *rc_node;      // [BOSS node number] -> last edge of the reverse complement's BOSS node, or 0
};
// words of DevGraph::terminus for a PRIMARY graph: terminus | terminus of the ids v + n | pal | rc_node (32-bit entries)
MGX_DEV PrimaryTables primary_tables(const DevGraph &g) {
    PrimaryTables t;
    t.pal = g.terminus + 2ull * g.n_blocks;
    t.rc_node = reinterpret_cast<const uint32_t *>(g.terminus + 3ull * g.n_blocks);
    return t;
}

// what the build kernel stores for the BOSS node whose last edge is `lst` (and, for even k, for edge e): see above
MGX_DEV uint32_t build_rc_node(const DevGraph &g, uint64_t lst) {
    LineCtr ctr = { 0, 0, 0 };
    const int32_t k = (int32_t)g.k;
    const Spell h = base_spelling(g, lst, ctr);              // node = the first k - 1 characters
    if (h.dollar & ((1ull << (k - 1)) - 1)) return 0;
    return (uint32_t)index_boss_node(g, spell_reverse_complement(h, k - 1), ctr);
}
MGX_DEV bool build_pal_bit(const DevGraph &g, uint64_t e) {
    if (g.k & 1) return false;
    LineCtr ctr = { 0, 0, 0 };
    const Spell h = base_spelling(g, e, ctr);
    return kmer_is_palindrome(h, (int32_t)g.k);
}

MGX_DEV int canon_children_tables(const DevGraph &g, uint32_t v, uint32_t *nodes, uint8_t *codes, bool *sentinel, LineCtr &ctr) {
    const PrimaryTables T = primary_tables(g);
    const uint64_t off = g.n;
    uint32_t have = 0;
    int n = 0;
    *sentinel = false;
    uint64_t nn[5];
    uint32_t cc[5];
    auto child_id_b = [&](uint64_t p) -> uint32_t {
        ++ctr.bit_lines;
        return ((gld(T.pal + (p >> 6)) >> (p & 63)) & 1) ? (uint32_t)p : (uint32_t)(p + off);
    };
    if (v <= off) {
        // set A, direct: DBGSuccinct::call_outgoing_kmers (as dev_graph.hpp outgoing()), keeping the target node's number
        ++ctr.rank_lines;
        Block cur = load_block(g, (uint32_t)(v >> 6));
        const uint32_t wv = block_W(cur, (int)(v & 63));
        if (v > 1 && wv == 0) return 0;
        Block tgt;
        const uint64_t lst = fwd_from(g, v, cur, wv % SIGMA, tgt, ctr);
        uint64_t first = pred_last_from(g, lst - 1, ((lst - 1) >> 6) == (lst >> 6) ? tgt : load_block(g, (uint32_t)((lst - 1) >> 6)), ctr) + 1;
        if (first < 2) first = 2;
        Block b = tgt;
        uint32_t bi = (uint32_t)(lst >> 6);
        for (uint64_t i = first; i <= lst; ++i) {
            if ((uint32_t)(i >> 6) != bi) { bi = (uint32_t)(i >> 6); ++ctr.rank_lines; b = load_block(g, bi); }
            const uint32_t c = block_W(b, (int)(i & 63)) % SIGMA;
            if (!in_graph(g, i)) continue;
            if (c == 0) { *sentinel = true; continue; }
            if (n < 4) { nodes[n] = (uint32_t)i; codes[n] = (uint8_t)c; have |= 1u << c; ++n; }
        }
        if (n == 4) return n;
        // set B: parents of the reverse complement of the target node
        const uint32_t node_no = tgt.last_cum + (uint32_t)popc64(tgt.last_bits & mask_upto((int)(lst & 63)));
        ++ctr.bit_lines;
        const uint64_t e = gld(T.rc_node + node_no);
        if (!e) return n;
        const int mi = incoming(g, e, nn, cc, ctr);
        for (int t = 0; t < mi; ++t) {
            if (cc[t] == 0) continue;
            const uint32_t a = 5u - cc[t];
            if (have & (1u << a)) continue;
            nodes[n] = child_id_b(nn[t]); codes[n] = (uint8_t)a; have |= 1u << a; ++n;
        }
        return n;
    }
    const uint64_t u = v - off;
    const int mi = incoming(g, u, nn, cc, ctr);                                    // set B, direct
    for (int t = 0; t < mi; ++t) {
        if (cc[t] == 0) { *sentinel = true; continue; }
        const uint32_t a = 5u - cc[t];
        nodes[n] = child_id_b(nn[t]); codes[n] = (uint8_t)a; have |= 1u << a; ++n;
    }
    if (n == 4) return n;
    // set A: the edges of the reverse complement of u's source node, from its last edge backwards
    const uint32_t node_no = rank_last(g, u - 1, ctr) + 1;
    ++ctr.bit_lines;
    uint64_t e = gld(T.rc_node + node_no);
    if (!e) return n;
    ++ctr.rank_lines;
    Block b = load_block(g, (uint32_t)(e >> 6));
    uint32_t bi = (uint32_t)(e >> 6);
    do {
        const uint32_t a = block_W(b, (int)(e & 63)) % SIGMA;
        if (a != 0 && in_graph(g, e) && !(have & (1u << a)) && n < 4) {
            nodes[n] = (uint32_t)e; codes[n] = (uint8_t)a; have |= 1u << a; ++n;
        }
        --e;
        if (!e) break;
        if ((uint32_t)(e >> 6) != bi) { bi = (uint32_t)(e >> 6); ++ctr.rank_lines; b = load_block(g, bi); }
    } while (!((b.last_bits >> (e & 63)) & 1));
    return n;
}

// number of callbacks of call_outgoing_kmers(v), the sentinel one included (:236-239: reported only when the base graph has
// no dummy mask and nothing else was found) — what has_multiple_outgoing / has_single_incoming count
MGX_DEV int canon_out_degree(const DevGraph &g, uint32_t v, const Spell &h, LineCtr &ctr) {
    uint32_t nodes[4];
    uint8_t codes[4];
    bool sentinel;
    const int n = canon_children(g, v, h, nodes, codes, &sentinel, ctr);
    return n + ((!g.valid && sentinel && n == 0) ? 1 : 0);
}

// MEM terminus bits (aligner_seeder_methods.hpp:121-125: has_multiple_outgoing || !has_single_incoming) of both wrapper ids
// of base node u: bit 0 = id u, bit 1 = id u + n
MGX_DEV uint32_t build_terminus_primary(const DevGraph &g, uint64_t u) {
    if (u == 0 || u > g.n) return 0;
    LineCtr ctr = { 0, 0, 0 };
    const Spell h = base_spelling(g, u, ctr);
    const Spell hr = spell_reverse_complement(h, (int32_t)g.k);
    const int out_fwd = canon_out_degree(g, (uint32_t)u, h, ctr);
    const int out_rc = canon_out_degree(g, (uint32_t)(u + g.n), hr, ctr);          // == the in-degree of u
    return (uint32_t)(out_fwd > 1 || out_rc != 1) | ((uint32_t)(out_rc > 1 || out_fwd != 1) << 1);
}

} // namespace mgx
