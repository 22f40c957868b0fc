// map_chain.hpp — BOSS::map_to_edges for one (sequence, strand) chain as a per-lane state machine over the bytes of the
// sequence.  Used by the mapping kernel for k > 32 (graph_build.hpp holds the 2-bit packed variant it normally runs) and by
// the extension kernel to re-map the reverse complement of an alignment's path on CANONICAL-mode graphs
// (Alignment::reverse_complement, alignment.cpp:563-565 -> reverse_complement_seq_path, sequence_graph.cpp:563-573).
// Included from align_core.hpp after the alphabet helpers.
#pragma once

namespace mgx {

// ------------------------------------------------------------------------------------------------
// BOSS::map_to_edges for one (read, strand) chain (boss.cpp:996-1045 via dbg_succinct.cpp:285-305).
// strand 1 maps the reverse complement (sequence_graph.cpp:563-573).  out has L - k + 1 slots.
// ------------------------------------------------------------------------------------------------
MGX_DEV uint32_t strand_code(const char *seq, int32_t L, int strand, int32_t pos) {
    if (!strand) return encode_char((uint8_t)seq[pos]);
    uint32_t c = encode_char((uint8_t)seq[L - 1 - pos]);
    return c == 5 ? 5u : 5u - c;                 // complement: A<->T, C<->G (kBOSSComplementMapDNA)
}

// ------------------------------------------------------------------------------------------------
// BOSS::map_to_edges as a per-lane state machine.  A lane owns one (read, strand) chain at a time and fetches the next
// one from `cursor` as soon as its chain ends; every loop iteration performs at most ONE memory-dependent step
// (a walk step fwd+pick_edge, the suffix-range table lookup, or one tighten_range), so lanes whose chain
// fails early (the non-matching strand) do not idle while their neighbours walk 120 k-mers.
// Per chain the primitives run in the reference's order: index() (table + tighten_range) at a start or after a miss,
// then fwd + pick_edge per base (boss.cpp:996-1045).
// ------------------------------------------------------------------------------------------------
// what index() learned at a position where it found no node: 0..k-1 = characters matched before tighten_range
// failed (k-1: only pick_edge failed), MLEN_LT_PREFIX = the suffix-range table had no entry (match shorter than
// the table's prefix), MLEN_UNKNOWN = index() did not run there.  The seeder's sub-k lookup walks exactly the
// same chain (BOSS::index_range, boss.hpp:720-764), so it can skip lookups that cannot reach min_seed_length.
constexpr uint8_t MLEN_UNKNOWN = 255, MLEN_LT_PREFIX = 254;
// at the LAST k-mer position of a strand, when that k-mer is a node (k_map_pipe only): the range slot of the position holds
// index_range of the read-tail position two further on (its k - 2 characters, all matched) — the one sub-k position of the
// tail that reports a seed when the last k-mer is part of a MEM (SuffixSeeder, aligner_seeder_methods.cpp:216-249); the seeding
// kernel would otherwise walk that range by itself, one read per wavefront
constexpr uint8_t MLEN_TAIL = 253;

struct MapLane {
    const char *seq;
    uint32_t *out;
    uint8_t *out_len;         // may be null; pre-set to MLEN_UNKNOWN by the caller, written only where index() failed
    uint2 *out_rng;           // may be null; (rl, ru) of the matched prefix where it has >= min_rng_len characters
    int32_t min_rng_len;
    int32_t L, n_kmers, strand;
    int32_t i;                // next k-mer position
    int32_t scanned, last_invalid;
    int32_t t;                // next character of the index() being built (state TIGHTEN)
    uint64_t edge, rl, ru;
    Block blk;                // block of `edge` while walking
    int state;                // 0 = need a chain, 1 = at position i, 2 = tightening, 3 = finished
};

template <class FetchChain>
MGX_DEV void map_lane_step(const DevGraph &g, MapLane &m, LineCtr &ctr, FetchChain fetch) {
    const int32_t k = (int32_t)g.k;
    if (m.state == 0) {
        if (!fetch(m)) { m.state = 3; return; }
        m.i = 0; m.scanned = 0; m.last_invalid = -1; m.edge = 0;
        m.state = m.n_kmers > 0 ? 1 : 0;
        return;
    }
    if (m.state == 1) {
        if (m.i >= m.n_kmers) { m.state = 0; return; }
        const int32_t i = m.i;
        for (; m.scanned < i + k; ++m.scanned)
            if (strand_code(m.seq, m.L, m.strand, m.scanned) == 5) m.last_invalid = m.scanned;
        if (m.last_invalid >= i) { gst_stream(m.out + i, 0); m.edge = 0; ++m.i; return; }
        if (m.edge) {
            // edge = fwd(edge, seq[i + k - 2]); edge = pick_edge(edge, seq[i + k - 1])
            uint32_t c_prev = strand_code(m.seq, m.L, m.strand, i + k - 2);
            Block tgt;
            uint64_t lst = fwd_from(g, m.edge, m.blk, c_prev, tgt, ctr);
            m.blk = tgt;
            m.edge = lst ? pick_edge_from(g, lst, m.blk, strand_code(m.seq, m.L, m.strand, i + k - 1), ctr) : 0;
            gst_stream(m.out + i, in_graph(g, m.edge) ? (uint32_t)m.edge : 0);
            ++m.i;
            return;
        }
        // map_to_edge: index(k - 1 codes) then pick_edge (boss.hpp:696-718,766-777); first the initial range
        int32_t t0 = 1;
        if (g.prefix_len && (int32_t)g.prefix_len <= k - 1) {
            uint32_t key = 0;
            for (uint32_t j = 0; j < g.prefix_len; ++j) key |= (strand_code(m.seq, m.L, m.strand, i + (int32_t)j) - 1) << (2 * j);
            prefix_range(g, key, &m.rl, &m.ru, ctr);
            t0 = (int32_t)g.prefix_len;
        } else {
            initial_range(g, strand_code(m.seq, m.L, m.strand, i), &m.rl, &m.ru);
        }
        if (m.rl > m.ru) {
            if (m.out_len && t0 > 1 && k - 1 < MLEN_LT_PREFIX) gst_stream(m.out_len + i, MLEN_LT_PREFIX);
            gst_stream(m.out + i, 0); m.edge = 0; ++m.i;
            return;
        }
        m.t = t0;
        m.state = 2;
        return;
    }
    // state 2: one tighten_range, or the final pick_edge
    const int32_t i = m.i;
    if (m.t < k - 1) {
        if (!tighten_range(g, &m.rl, &m.ru, strand_code(m.seq, m.L, m.strand, i + m.t), ctr)) {
            if (m.out_len && k - 1 < MLEN_LT_PREFIX) {
                gst_stream(m.out_len + i, (uint8_t)m.t);
                if (m.out_rng && m.t >= m.min_rng_len) m.out_rng[i] = make_uint2((uint32_t)m.rl, (uint32_t)m.ru);
            }
            gst_stream(m.out + i, 0); m.edge = 0; ++m.i; m.state = 1;
            return;
        }
        ++m.t;
        return;
    }
    ++ctr.rank_lines;
    m.blk = load_block(g, (uint32_t)(m.ru >> 6));
    m.edge = pick_edge_from(g, m.ru, m.blk, strand_code(m.seq, m.L, m.strand, i + k - 1), ctr);
    gst_stream(m.out + i, in_graph(g, m.edge) ? (uint32_t)m.edge : 0);
    if (!m.edge && m.out_len && k - 1 < MLEN_LT_PREFIX) {
        gst_stream(m.out_len + i, (uint8_t)(k - 1));
        if (m.out_rng && k - 1 >= m.min_rng_len) m.out_rng[i] = make_uint2((uint32_t)m.rl, (uint32_t)m.ru);
    }
    ++m.i;
    m.state = 1;
}

} // namespace mgx
