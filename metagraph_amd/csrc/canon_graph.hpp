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
// This first version recomputes h (k - 1 bwd steps) and the look-up (k - 2 tighten_range steps) for every expansion and holds
// the spelling in two 64-bit registers, hence k <= 32 for PRIMARY graphs.  The reference caches both behind LRU caches
// (canonical_dbg.hpp:121-137); an incremental h along a chain of columns is the obvious next step.
#pragma once
#include "dev_graph.hpp"

namespace mgx {

// spelling of a node: character j (0 = first) at bits [2j, 2j + 2) as A C G T = 0..3; bit j of `dollar` marks a sentinel
struct Spell { uint64_t code, dollar; };

MGX_DEV void spell_put(Spell &s, int32_t j, uint32_t c /* 0..4 */) {
    if (c == 0) s.dollar |= 1ull << j;
    else s.code |= (uint64_t)(c - 1) << (2 * j);
}
MGX_DEV uint32_t spell_get(const Spell &s, int32_t j) { return ((s.dollar >> j) & 1) ? 0u : (uint32_t)((s.code >> (2 * j)) & 3) + 1u; }

// get_node_sequence of a base node (dbg_succinct.cpp:272-279 -> BOSS::get_node_seq boss.cpp:940-973 + the edge label)
MGX_DEV Spell base_spelling(const DevGraph &g, uint64_t e, LineCtr &ctr) {
    const int32_t k = (int32_t)g.k;
    Spell s = { 0, 0 };
    spell_put(s, k - 1, get_W(g, e, ctr) % SIGMA);
    uint64_t x = e;
    for (int32_t j = k - 2; j >= 0; --j) {
        spell_put(s, j, node_last_value(g, x));
        if (j) x = bwd(g, x, ctr);
    }
    return s;
}

MGX_DEV Spell spell_reverse_complement(const Spell &s, int32_t k) {
    Spell r = { 0, 0 };
    for (int32_t j = 0; j < k; ++j) {
        const uint32_t c = spell_get(s, k - 1 - j);
        spell_put(r, j, c ? 5u - c : 0u);                    // complement('$') == '$'
    }
    return r;
}

// is the k-mer with 2-bit codes `code` its own reverse complement?
MGX_DEV bool kmer_is_palindrome(uint64_t code, int32_t k) {
    if (k & 1) return false;
    for (int32_t j = 0; j < k / 2; ++j)
        if (((code >> (2 * j)) & 3) + ((code >> (2 * (k - 1 - j))) & 3) != 3) return false;
    return true;
}

// last edge of the BOSS node spelled by the k - 1 codes of `t` (first character least significant), 0 if there is none:
// BOSS::index_range (boss.hpp:720-764) demanding a full match, as get_prefix_rc / get_suffix_rc do
MGX_DEV uint64_t index_boss_node(const DevGraph &g, uint64_t t, LineCtr &ctr) {
    const int32_t len = (int32_t)g.k - 1;
    uint64_t rl = 1, ru = 0;
    int32_t it = 1;
    if (g.prefix_len && (int32_t)g.prefix_len <= len) {
        const uint32_t key = (uint32_t)(t & ((1ull << (2 * g.prefix_len)) - 1));
        prefix_range(g, key, &rl, &ru, ctr);
        if (rl > ru) return 0;
        it = (int32_t)g.prefix_len;
    } else {
        initial_range(g, (uint32_t)(t & 3) + 1, &rl, &ru);
        if (rl > ru) return 0;
    }
    for (; it < len; ++it)
        if (!tighten_range(g, &rl, &ru, (uint32_t)((t >> (2 * it)) & 3) + 1, ctr)) return 0;
    return ru;
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
    const uint64_t tail = h.code >> 2;                       // h[1:], k - 1 codes
    const uint64_t kmask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    auto child_id_b = [&](uint64_t p, uint32_t a) -> uint32_t {
        // reverse_complement(p) (:515-549): p + n unless the k-mer (== the child h[1:] + a mirrored) is a palindrome
        const uint64_t child = (tail | ((uint64_t)(a - 1) << (2 * (k - 1)))) & kmask;
        return ((h.dollar >> 1) == 0 && kmer_is_palindrome(child, k)) ? (uint32_t)p : (uint32_t)(p + off);
    };
    const bool tail_clean = (h.dollar >> 1) == 0;            // no sentinel in h[1:]
    uint64_t nn[5];
    uint32_t cc[5];
    if (!is_rc) {
        const int m = outgoing(g, v, nn, cc, ctr, sentinel);                       // set A, direct
        for (int t = 0; t < m; ++t) { nodes[n] = (uint32_t)nn[t]; codes[n] = (uint8_t)cc[t]; have |= 1u << cc[t]; ++n; }
        if (n == 4 || !tail_clean) return n;
        // set B: parents of the node RC(h[1:])
        uint64_t t_rc = 0;
        for (int32_t j = 0; j < k - 1; ++j) t_rc |= (3 - ((tail >> (2 * (k - 2 - j))) & 3)) << (2 * j);
        const uint64_t e = index_boss_node(g, t_rc, ctr);
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
        uint64_t e = index_boss_node(g, tail & (kmask >> 2), ctr);
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
