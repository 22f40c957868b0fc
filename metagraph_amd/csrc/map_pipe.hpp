// map_pipe.hpp — BOSS::map_to_edges (boss.cpp:996-1045) over the 2-bit packed reads as a REQUEST / RESPONSE machine.
//
// map_lane_step_packed (graph_build.hpp, rounds 1-4) gives every lane one chain and lets it call fwd / tighten_range /
// prefix_range; the lanes of a wavefront are in different states, the compiler serialises the branches, and every branch
// waits for its own dependent loads: one loop iteration of a wavefront cost the SUM of the round trips of the walk, the
// table lookup, the range tightening, the final pick and the chain fetch (SQ_WAIT_ANY 87 %, 7 % of the cycles issuing).
// Here an iteration has ONE memory round trip for all 64 lanes: every lane states what it needs next (one 64-byte block,
// one table entry, one top-up word of its read), all requests — and the stores the last iteration left —
// are issued back to back at the top of the iteration, and what follows the single wait is arithmetic on registers only
// (`map_pipe_step`).  The primitives are the ones of dev_graph.hpp cut at their loads; the results (node arrays, match
// lengths, ranges) are those of map_lane_step_packed bit for bit (tests/test_map_pipe.py, the whole emulator suite — the
// host model steps this machine).
//
//   walk     rank in the block of the current edge (registers) -> select_last by a scan from the PREDICTED block
//            (sel_predict over DevGraph::sel_anchor, which the kernel keeps in LDS: fwd is one dependent load where rounds
//            1-4 fetched last_hint first; a table of target blocks per graph block was tried first — right more often,
//            but a second line per block: the kernel then sat on the memory system's request rate) -> pick_edge in the
//            block found
//   lookup   suffix-range table -> [block of ru unless it is the one held] -> scan from the predicted block for
//            select_last(r_hi); r_lo is answered from the same block in the common case (narrow ranges) -> next character
//            ... -> pick_edge
//   chains   the NEXT chain of a lane (its id, then offsets / node_begin, then the first words of the strand) is fetched in
//            the background with the requests of the current one; ids come from a per-wavefront pool refilled by one atomic
//            per 256 chains, so no returning atomic sits on the iteration's critical path.
#pragma once
#include "graph_build.hpp"

namespace mgx {

struct MapArgs {
    const uint64_t *offsets, *node_begin;
    const uint64_t *pk_fwd, *pk_rc;
    const uint32_t *iv_fwd, *iv_rc;
    uint32_t *nodes_fwd, *nodes_rc;
    uint8_t *mlen_fwd, *mlen_rc;          // may be null
    uint2 *rng_fwd, *rng_rc;              // may be null
    int32_t min_rng_len;
    uint64_t n_reads;
    int32_t do_rc;
    unsigned long long *cursor;
};

enum : int32_t { MP_IDLE = 0, MP_SETTLE, MP_TOPUP, MP_PREFIX, MP_SEL, MP_BLK, MP_DONE };
// what a block request is for
enum : int32_t { PU_WALK = 0,     // MP_SEL: select_last(r_hi) is the walk's fwd
                 PU_HI,           // MP_SEL: select_last(r_hi) of tighten_range; MP_BLK: the block of ru (next rank / the final pick)
                 PU_LO,           // MP_SEL: select_last(r_lo); MP_BLK: the block of rl - 1
                 PU_PICK };       // MP_BLK: pick_edge stepped into the previous block
enum : int32_t { BG_WANT = 0, BG_DESC, BG_READY, BG_EOF };
constexpr uint32_t MP_NO_BLOCK = 0xFFFFFFFFu;

struct MapPipe {
    uint64_t wbase, nbase;        // word 0 of the chain's strand in pk_* / iv_*; node_begin of its read
    int32_t strand, n_words, n_kmers;
    int32_t i, t;                 // k-mer position; characters matched so far (lookup), or -1 while walking
    int32_t tail;                 // the read-tail lookup behind the last k-mer (MLEN_TAIL): 0 not tried, 1 in progress, 2 over
    uint32_t last_node;           // what the last settled k-mer stored
    uint64_t cur, nxt;            // codes of positions [i, i + 32); the not yet consumed codes of the word after them
    uint32_t icur, inxt;
    uint32_t edge, rl, ru;        // (edge indices fit 32 bits on the device)
    Block blk;                    // a block held in registers: that of `edge` while walking, that of `ru` in a lookup
    uint32_t blk_idx;
    uint32_t r_hi, r_lo, rk_ru;
    uint32_t req;                 // the request of this iteration: block index / table key / word index
    int32_t state, purpose;
    // what the last settled k-mer leaves behind, written at the top of the NEXT iteration with that iteration's requests (gfx9
    // counts loads and stores with one counter: a store issued after the wait would be waited for before the next requests)
    uint32_t st_flags;            // 1: the node quad, 2: match length, 4: range
    uint32_t st_len;
    uint64_t st_idx, st_rng;      // index into mlen_* / rng_* of strand (st_flags >> 8)
    // node ids leave in quads: four consecutive k-mer positions as ONE 16-byte store where the chain covers the quad (its
    // first and last positions may share a quad with the neighbouring reads: those go as single words).  240 scattered 4-byte
    // stores per read were 18 KB of write traffic per read at the fabric — 40 % of this kernel's traffic (round 5 PMC).
    uint32_t ob[4], ob_mask, ob_strand;
    uint64_t ob_q;                // the quad: positions 4 ob_q .. 4 ob_q + 3 of nodes_* (ob_strand)
    // ... and so do the match lengths of the positions index() failed at (mlen_*, same positions): one 4-byte store where all
    // four positions of a quad failed — every quad of a strand that is not in the graph — else single bytes.  pl: the length
    // of the position at hand, known when its walk fails, entered into the quad when the position settles (a quad on its way
    // out holds the positions before it).
    uint32_t lb, lb_mask;
    int32_t pl;
    // the next chain
    int32_t bg, bg_strand, bg_L;
    uint64_t bg_read, bg_w, bg_nb, bg_cur, bg_nxt;
    uint32_t bg_icur, bg_inxt;
};

MGX_DEV void map_pipe_init(MapPipe &m) {
    m.state = MP_IDLE; m.purpose = PU_WALK; m.bg = BG_WANT; m.blk_idx = MP_NO_BLOCK; m.i = 0; m.n_kmers = 0; m.edge = 0; m.t = -1;
    m.tail = 2; m.last_node = 0;
    m.cur = m.nxt = 0; m.icur = m.inxt = 0; m.rl = m.ru = 0; m.r_hi = m.r_lo = m.rk_ru = 0; m.req = 0;
    m.wbase = m.nbase = 0; m.strand = 0; m.n_words = 0;
    m.st_flags = 0; m.st_len = 0; m.st_idx = m.st_rng = 0;
    m.ob[0] = m.ob[1] = m.ob[2] = m.ob[3] = 0; m.ob_mask = 0; m.ob_strand = 0; m.ob_q = 0;
    m.lb = 0; m.lb_mask = 0; m.pl = -1;
    m.blk = Block{};
}

// chain ids for the lanes of one wavefront that ask for one (`want`); ~0 = none asked.  Must be called by all lanes.
#if MGX_WAVE_EMU
struct ChainClaim {
    unsigned long long *cursor;
    MGX_DEV explicit ChainClaim(unsigned long long *c) : cursor(c) {}
    MGX_DEV uint64_t get(bool want) { return want ? (uint64_t)(*cursor)++ : ~0ull; }
};
#else
struct ChainClaim {
    static constexpr uint64_t CHUNK = 256;
    unsigned long long *cursor;
    uint64_t next, end;           // wave-uniform
    MGX_DEV explicit ChainClaim(unsigned long long *c) : cursor(c), next(0), end(0) {}
    MGX_DEV uint64_t get(bool want) {
        const uint64_t mask = __ballot(want);
        if (!mask) return ~0ull;
        const uint64_t n = (uint64_t)__builtin_popcountll(mask);
        const uint64_t rank = (uint64_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        const uint64_t avail = end - next;
        uint64_t id;
        if (avail >= n) {
            id = next + rank;
            next += n;
        } else {
            unsigned long long b = 0;
            if (rank == 0 && want) b = atomicAdd(cursor, (unsigned long long)CHUNK);
            const uint32_t src = (uint32_t)__builtin_ctzll(mask);
            const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int32_t)(b >> 32), (int32_t)src) << 32)
                                  | (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)b, (int32_t)src);
            id = rank < avail ? next + rank : base + (rank - avail);
            next = base + (n - avail);
            end = base + CHUNK;
        }
        return want ? id : ~0ull;
    }
};
#endif

MGX_DEV uint32_t mp_code(const MapPipe &m, int32_t off) { return (uint32_t)((m.cur >> (2 * off)) & 3) + 1; }

// the select predictor over the graph's own table (the HIP kernel reads a copy in LDS instead)
struct SelPredictGlobal {
    const uint32_t *tab; uint32_t shift;
    MGX_DEV uint32_t operator()(uint32_t r) const { return sel_predict(tab, shift, r); }
};

// One iteration of one lane.  Returns false once the lane has nothing left to do (it must still be called while any
// lane of its wavefront has: the chain-id pool is wave-wide).
template <class Claim, class Pred>
MGX_DEV bool map_pipe_step(const DevGraph &g, const MapArgs &a, MapPipe &m, LineCtr &ctr, Claim &claim, const Pred &pred) {
    const int32_t k = (int32_t)g.k;                      // <= 32
    const uint64_t n_chains = a.do_rc ? 2 * a.n_reads : a.n_reads;
    // ------------------------------------------------------------------------------------------ requests
    const int32_t st = m.state;
    const uint64_t cid = claim.get(m.bg == BG_WANT);
    if (m.st_flags) {
        const bool rcs = (m.st_flags >> 8) != 0;
        if (m.st_flags & 1) {
            uint32_t *q = (m.ob_strand ? a.nodes_rc : a.nodes_fwd) + (m.ob_q << 2);
            if (m.ob_mask == 0xFu) gst4((int32_t *)q, (int32_t)m.ob[0], (int32_t)m.ob[1], (int32_t)m.ob[2], (int32_t)m.ob[3]);
            else {
                if (m.ob_mask & 1u) gst_stream(q, m.ob[0]);
                if (m.ob_mask & 2u) gst_stream(q + 1, m.ob[1]);
                if (m.ob_mask & 4u) gst_stream(q + 2, m.ob[2]);
                if (m.ob_mask & 8u) gst_stream(q + 3, m.ob[3]);
            }
            if (m.lb_mask) {
                uint8_t *ql = (m.ob_strand ? a.mlen_rc : a.mlen_fwd) + (m.ob_q << 2);
                if (m.lb_mask == 0xFu) gst_stream((uint32_t *)ql, m.lb);
                else {
                    if (m.lb_mask & 1u) gst_stream(ql, (uint8_t)m.lb);
                    if (m.lb_mask & 2u) gst_stream(ql + 1, (uint8_t)(m.lb >> 8));
                    if (m.lb_mask & 4u) gst_stream(ql + 2, (uint8_t)(m.lb >> 16));
                    if (m.lb_mask & 8u) gst_stream(ql + 3, (uint8_t)(m.lb >> 24));
                }
            }
        }
        if (m.st_flags & 2) gst_stream((rcs ? a.mlen_rc : a.mlen_fwd) + m.st_idx, (uint8_t)m.st_len);
        if (m.st_flags & 4) gst((uint64_t *)((rcs ? a.rng_rc : a.rng_fwd) + m.st_idx), m.st_rng);
    }
    // (no initialisers on the device: merging a loaded value with a constant costs register copies inside the conditional
    // block, and the copies wait for the load — every request would then be waited for before the next one is issued)
#if MGX_WAVE_EMU
    Block A = Block{};
    uint64_t v8 = 0, o0 = 0, o1 = 0, nb = 0, p0 = 0, p1 = 0;
    uint32_t v4 = 0, q0 = 0, q1 = 0;
#else
    Block A;
    uint64_t v8, o0, o1, nb, p0, p1;
    uint32_t v4, q0, q1;
#endif
    if (st == MP_SEL || st == MP_BLK) A = load_block(g, m.req);
    if (st == MP_PREFIX) v8 = gld_stream_u64(g.prefix_tbl + m.req);
    if (st == MP_TOPUP) {
        v8 = gld((m.strand ? a.pk_rc : a.pk_fwd) + m.wbase + m.req);
        v4 = gld((m.strand ? a.iv_rc : a.iv_fwd) + m.wbase + m.req);
    }
    // the next chain, in the background
    if (m.bg == BG_WANT) {
        if (cid >= n_chains) {
            m.bg = BG_EOF;
        } else {
            m.bg_read = a.do_rc ? (cid >> 1) : cid;
            m.bg_strand = a.do_rc ? (int32_t)(cid & 1) : 0;
            o0 = gld(a.offsets + m.bg_read); o1 = gld(a.offsets + m.bg_read + 1);
            nb = gld(a.node_begin + m.bg_read);
        }
    } else if (m.bg == BG_DESC) {
        const int32_t j1 = m.bg_L > 32 ? 1 : 0;
        const uint64_t *pk = (m.bg_strand ? a.pk_rc : a.pk_fwd) + m.bg_w;
        const uint32_t *iv = (m.bg_strand ? a.iv_rc : a.iv_fwd) + m.bg_w;
        p0 = gld(pk); p1 = gld(pk + j1);
        q0 = gld(iv); q1 = gld(iv + j1);
    }
    // ------------------------------------------------------------------------------------------ responses (registers only)
    if (m.bg == BG_DESC) {
        m.bg_cur = p0; m.bg_nxt = p1; m.bg_icur = q0; m.bg_inxt = q1;
        m.bg = BG_READY;
    } else if (m.bg == BG_WANT) {
        m.bg_L = (int32_t)(o1 - o0);
        m.bg_w = packed_word_begin(o0, m.bg_read);
        m.bg_nb = nb;
        m.bg = m.bg_L >= k ? BG_DESC : BG_WANT;          // a read without a k-mer has no chain: ask for the next id
    }
    if (m.st_flags & 1) { m.ob_mask = 0; m.lb_mask = 0; m.lb = 0; }
    m.st_flags = 0;
    if (st == MP_DONE) return false;

    // outputs of the settled k-mer: pending stores (one settled k-mer per iteration, MP_SETTLE otherwise)
    auto out_node = [&](uint32_t v) {
        const uint64_t gi = m.nbase + (uint64_t)m.i;
        const uint32_t slot = (uint32_t)gi & 3u;
        m.ob_q = gi >> 2; m.ob_strand = (uint32_t)m.strand;
        if (slot == 0) m.ob[0] = v; else if (slot == 1) m.ob[1] = v; else if (slot == 2) m.ob[2] = v; else m.ob[3] = v;
        m.ob_mask |= 1u << slot;
        if (slot == 3 || m.i + 1 >= m.n_kmers) m.st_flags |= 1u;           // the quad, or the chain, is complete: out with it
    };
    auto out_len = [&](uint8_t v) { m.pl = (int32_t)v; };                        // (into the quad when the position settles)
    auto out_rng = [&]() { m.st_flags |= 4u | ((uint32_t)m.strand << 8); m.st_idx = m.nbase + (uint64_t)m.i; m.st_rng = ((uint64_t)m.ru << 32) | m.rl; };
    const bool lens = a.mlen_fwd && k - 1 < (int32_t)MLEN_LT_PREFIX;
    const bool rngs = a.rng_fwd != nullptr;

    enum { ACT_NONE, ACT_POS, ACT_ADVANCE, ACT_PICK, ACT_TIGHT, ACT_RANK, ACT_SEL, ACT_NEXT_CHAR };
    int act = ACT_NONE;
    uint32_t rk_rl = 0, rk_ru = 0;
    if (st == MP_IDLE) act = ACT_POS;
    else if (st == MP_SETTLE) act = ACT_ADVANCE;
    else if (st == MP_TOPUP) { m.nxt = v8; m.inxt = v4; act = ACT_POS; }
    else if (st == MP_PREFIX) {
        ++ctr.bit_lines;
        m.rl = (uint32_t)v8; m.ru = (uint32_t)(v8 >> 32);
        if (m.rl > m.ru) {
            if (m.tail == 1) { m.tail = 2; m.t = -1; act = ACT_POS; }           // (the seeder will walk it itself, from scratch)
            else {
                if (lens && g.prefix_len > 1) out_len(MLEN_LT_PREFIX);
                m.edge = 0; m.t = -1; act = ACT_ADVANCE;
            }
        } else { m.t = (int32_t)g.prefix_len; act = ACT_TIGHT; }
    } else if (st == MP_SEL) {
        // select_last(r) by a scan in either direction (select_last_scan, dev_graph.hpp)
        ++ctr.select_lines;
        const uint32_t r = m.purpose == PU_LO ? m.r_lo : m.r_hi;
        if (A.last_cum >= r) --m.req;
        else if (A.last_cum + (uint32_t)popc64(A.last_bits) < r) ++m.req;
        else {
            const uint32_t pos = (m.req << 6) + (uint32_t)select64(A.last_bits, (int)(r - A.last_cum));
            if (m.purpose == PU_LO) { m.rl = pos + 1; act = ACT_NEXT_CHAR; }
            else {
                m.blk = A; m.blk_idx = m.req;
                if (m.purpose == PU_WALK) { m.edge = pos; act = ACT_PICK; }
                else {
                    m.ru = pos;
                    if (m.r_lo == 0) { m.rl = 1; act = ACT_NEXT_CHAR; }
                    else if (m.r_lo > A.last_cum) {
                        m.rl = (m.req << 6) + (uint32_t)select64(A.last_bits, (int)(m.r_lo - A.last_cum)) + 1;
                        act = ACT_NEXT_CHAR;
                    } else { m.purpose = PU_LO; m.req = pred(m.r_lo); }
                }
            }
        }
    } else if (st == MP_BLK) {
        ++ctr.rank_lines;
        if (m.purpose == PU_LO) {
            const uint32_t s = mp_code(m, m.t);
            rk_rl = block_rank_W(A, (int)((m.rl - 1) & 63), s, m.req == 0) + 1;
            rk_ru = m.rk_ru;
            act = ACT_SEL;
        } else {
            m.blk = A; m.blk_idx = m.req;
            if (m.purpose == PU_PICK) {
                if ((A.last_bits >> (m.edge & 63)) & 1) { m.edge = 0; act = ACT_ADVANCE; }
                else act = ACT_PICK;
            } else if (m.t < k - 1) act = ACT_RANK;
            else { m.edge = m.ru; act = ACT_PICK; }
        }
    }

    while (act != ACT_NONE) {
        if (act == ACT_NEXT_CHAR) { ++m.t; act = ACT_TIGHT; }
        if (act == ACT_TIGHT) {
            // tighten_range(rl, ru, q[i + t]) (boss.hpp:682-693), or — all k - 1 characters matched — the edge itself
            const uint32_t hb = m.ru >> 6;
            if (m.tail == 1 && m.t >= k - 2) {
                // the read-tail range is complete: into the slots of the last k-mer position
                m.st_flags |= 2u | 4u | ((uint32_t)m.strand << 8);
                m.st_idx = m.nbase + (uint64_t)(m.n_kmers - 1); m.st_len = MLEN_TAIL;
                m.st_rng = ((uint64_t)m.ru << 32) | m.rl;
                m.tail = 2; m.t = -1; act = ACT_POS;
            } else if (hb != m.blk_idx) { m.req = hb; m.state = MP_BLK; m.purpose = PU_HI; act = ACT_NONE; }
            else if (m.t < k - 1) act = ACT_RANK;
            else { m.edge = m.ru; act = ACT_PICK; }
        }
        if (act == ACT_RANK) {
            const uint32_t s = mp_code(m, m.t);
            const uint32_t lo = m.rl - 1, hi = m.ru;
            rk_ru = block_rank_W(m.blk, (int)(hi & 63), s, m.blk_idx == 0);
           
            if (lo == 0) { rk_rl = 1; act = ACT_SEL; }
            else if ((lo >> 6) == m.blk_idx) { rk_rl = block_rank_W(m.blk, (int)(lo & 63), s, m.blk_idx == 0) + 1; act = ACT_SEL; }
            else { m.rk_ru = rk_ru; m.req = lo >> 6; m.state = MP_BLK; m.purpose = PU_LO; act = ACT_NONE; }
        }
        if (act == ACT_SEL) {
            if (rk_rl > rk_ru && m.tail == 1) { m.tail = 2; m.t = -1; act = ACT_POS; }
            else if (rk_rl > rk_ru) {
                // index() fails at character t: what matched, and its range, go to the seeder (graph_build.hpp MLEN_*)
                if (lens) {
                    out_len((uint8_t)m.t);
                    if (rngs && m.t >= a.min_rng_len) out_rng();
                }
                m.edge = 0; m.t = -1; act = ACT_ADVANCE;
            } else {
                const uint32_t s = mp_code(m, m.t);
                const uint32_t nfs = nf_of(g, s);
                m.r_hi = nfs + rk_ru; m.r_lo = nfs + rk_rl - 1;
                m.req = pred(m.r_hi); m.state = MP_SEL; m.purpose = PU_HI; act = ACT_NONE;
            }
        }
        if (act == ACT_PICK) {
            // pick_edge (boss.cpp:710-722) from m.edge backwards, in the block held
            const uint32_t c = mp_code(m, k - 1);
            for (;;) {
                const uint32_t w = block_W(m.blk, (int)(m.edge & 63));
                if (w == c || w == c + SIGMA) { act = ACT_ADVANCE; break; }
                --m.edge;
                if (m.edge == 0) { act = ACT_ADVANCE; break; }
                if ((m.edge & 63) == 63) { m.req = m.edge >> 6; m.state = MP_BLK; m.purpose = PU_PICK; act = ACT_NONE; break; }
                if ((m.blk.last_bits >> (m.edge & 63)) & 1) { m.edge = 0; act = ACT_ADVANCE; break; }
            }
        }
        if (act == ACT_ADVANCE && (m.st_flags & 1)) { m.state = MP_SETTLE; act = ACT_NONE; }      // the quad of the earlier k-mers is still on its way out
        if (act == ACT_ADVANCE) {
            // the k-mer at position i is settled: m.edge (0 = not found)
            m.last_node = in_graph(g, m.edge) ? m.edge : 0u;
            if (m.t == k - 1 && !m.edge && lens) {
                out_len((uint8_t)(k - 1));
                if (rngs && k - 1 >= a.min_rng_len) out_rng();
            }
            if (m.pl >= 0) {
                const uint32_t slot = (uint32_t)(m.nbase + (uint64_t)m.i) & 3u;
                m.lb |= (uint32_t)m.pl << (8 * slot);
                m.lb_mask |= 1u << slot;
                m.pl = -1;
            }
            out_node(m.last_node);
            m.t = -1;
            m.cur = (m.cur >> 2) | ((m.nxt & 3) << 62);
            m.icur = (m.icur >> 1) | ((m.inxt & 1) << 31);
            m.nxt >>= 2; m.inxt >>= 1;
            ++m.i;
            if ((m.i & 31) == 0 && m.i < m.n_kmers) {      // position i + 32 starts word (i >> 5) + 1
                m.req = (uint32_t)imin((m.i >> 5) + 1, m.n_words - 1);
                m.state = MP_TOPUP; act = ACT_NONE;
            } else act = ACT_POS;
        }
        if (act == ACT_POS) {
            if (m.i >= m.n_kmers && m.tail == 0) {
                // behind the last k-mer, if it is a node: index_range of the tail position i + 1 (its k - 2 characters follow
                // the node's first two), for the seeding kernel (MLEN_TAIL)
                m.tail = 2;
                const uint32_t tmask = (1u << (k - 2)) - 1u;
                if (m.last_node && lens && rngs && g.prefix_len && (int32_t)g.prefix_len <= k - 2 && k - 2 >= a.min_rng_len
                        && !((m.icur >> 1) & tmask)) {
                    m.tail = 1;
                    m.cur >>= 2;
                    m.req = (uint32_t)(m.cur & ((1ull << (2 * g.prefix_len)) - 1ull));
                    m.state = MP_PREFIX; act = ACT_NONE;
                }
            } else if (m.i >= m.n_kmers) {
                // chain finished: take the prefetched one
                if (m.bg == BG_READY) {
                    m.wbase = m.bg_w; m.nbase = m.bg_nb; m.strand = m.bg_strand;
                    m.n_words = (m.bg_L + 31) >> 5; m.n_kmers = m.bg_L - k + 1;
                    m.cur = m.bg_cur; m.nxt = m.bg_nxt; m.icur = m.bg_icur; m.inxt = m.bg_inxt;
                    m.i = 0; m.edge = 0; m.t = -1; m.tail = 0; m.last_node = 0;
                    m.bg = BG_WANT;
                    m.state = MP_IDLE;                     // (and on with its first k-mer: act stays ACT_POS)
                } else {
                    m.n_kmers = 0; m.i = 0;
                    m.state = m.bg == BG_EOF ? MP_DONE : MP_IDLE; act = ACT_NONE;
                }
            } else {
                const uint32_t kmask = k >= 32 ? 0xFFFFFFFFu : ((1u << k) - 1u);
                if (m.icur & kmask) { m.edge = 0; act = ACT_ADVANCE; }          // an invalid character in the k-mer
                else if (m.edge) {
                    // fwd (boss.cpp:642-652): rank in the block of the edge, then select_last from the predicted block
                    const uint32_t c = mp_code(m, k - 2);
                    const uint32_t r = nf_of(g, c) + block_rank_W(m.blk, (int)(m.edge & 63), c, m.blk_idx == 0);
                    if (r == 0) { m.edge = 0; act = ACT_ADVANCE; }
                    else { m.r_hi = r; m.req = pred(r); m.state = MP_SEL; m.purpose = PU_WALK; act = ACT_NONE; }
                } else if (g.prefix_len && (int32_t)g.prefix_len <= k - 1) {
                    m.req = (uint32_t)(m.cur & ((1ull << (2 * g.prefix_len)) - 1ull));
                    m.state = MP_PREFIX; act = ACT_NONE;
                } else if (m.st_flags) {
                    m.state = MP_IDLE; act = ACT_NONE;      // (this lookup may fail without a request: not next to a pending k-mer)
                } else {
                    uint64_t rl64, ru64;
                    initial_range(g, mp_code(m, 0), &rl64, &ru64);
                    m.rl = (uint32_t)rl64; m.ru = (uint32_t)ru64;
                    if (rl64 > ru64) { m.edge = 0; act = ACT_ADVANCE; }
                    else { m.t = 1; act = ACT_TIGHT; }
                }
            }
        }
    }
    return m.state != MP_DONE || m.st_flags != 0;
}

} // namespace mgx
