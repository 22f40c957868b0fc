// tests/emu/emu_driver.cpp — TEST INFRASTRUCTURE ONLY.
// Runs the kernels' wave programs (metagraph_amd/csrc/{dev_graph,graph_build,align_core}.hpp, the
// same sources the HIP build compiles) under the lock-step host model tests/emu/wave.hpp, one
// "wave" at a time, so that CPU-only CI can compare kernel logic with the oracle.  Built into
// tests/emu/_build/libmgxemu.so; never linked into or loaded by the product.
#include "wave.hpp"              // tests/emu/wave.hpp (shadows the gfx950 header)

#include <cstdio>
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "graph_build.hpp"
#include "map_pipe.hpp"
#include "host_common.hpp"
#include "lane_read.hpp"
#include "seed_lane.hpp"

using namespace mgx;

namespace {
struct EmuGraph {
    DevGraph g;
    std::vector<Block> blocks;
    std::vector<uint32_t> last_hint, w_hint[4], sel_anchor, firstc;
    std::vector<uint64_t> terminus, valid;
    std::vector<uint2> prefix_tbl;
    uint32_t mode = 0;
    bool tables = false;      // PRIMARY: reverse-complement tables built (default; MGX_PRIMARY_TABLES=0: off)
};

struct EmuRun {
    std::string error;
    std::vector<ReadResult> results;
    std::vector<uint32_t> stream;
    HostResults host;
    std::vector<uint64_t> node_begin, m_fwd, m_rc;
    std::vector<uint8_t> m_len[2];          // k_map's side outputs (match lengths, ranges): test_map_pipe.py
    std::vector<uint2> m_rng[2];
    std::vector<DevSeed> seeds;
    DevLimits lim;
    KernelStats stats;
    uint64_t retried = 0;         // reads handed to pass 2 of the two-pass extension
    uint64_t out_used = 0;        // words of `stream` in use
    uint64_t lane_done = 0, lane_ran = 0;     // MGX_EMU_LANE: reads the lane-per-read path finished; whether it ran at all
    uint64_t lane_bail[32] = { 0 };            // ... and why it sent the others on (LANE_BAIL codes)
    uint64_t seedlane_done = 0, seedlane_ran = 0;       // MGX_EMU_SEEDLANE: reads the lane-per-read seeder finished; whether it ran
    uint64_t seedlane_bail[16] = { 0 };
    std::vector<uint8_t> lane_reason;          // per read: 0 = finished by the lane path, else the code
};
// the label matrix in the device's row-major form (mgx_annot.hip: head word = count:16 | single label or offset into more[])
struct EmuAnno {
    std::vector<uint64_t> head;
    std::vector<uint32_t> count, more;
    uint64_t n_rows = 0;
    uint32_t n_labels = 0;
};
#ifdef MGX_EMU_TRACE
// traced build (make trace): every access of the wave program calls the hooks of trace_hooks.cpp
extern "C" {
void mgx_trace_region(const void *p, uint64_t bytes, const char *name);
void mgx_trace_reset();
void mgx_trace_on(int on);
void mgx_trace_end_read();
uint64_t mgx_trace_report(char *buf, uint64_t cap);
}
// the arena arrays of carve() (everything placed in the arena: lds = nullptr), by name
static void trace_register_arena(const AlignParams &P, uint8_t *base, uint64_t stride) {
    auto w = std::make_unique<Wave>();
    carve(*w, P, base, nullptr, 0);
    struct Pt { const void *p; const char *name; };
    std::vector<Pt> pts = {
        { w->q[0], "q" }, { w->msl, "seeding_scratch" }, { w->st[0].S.hi, "staging" }, { w->pk[0], "pk" }, { w->psum[0], "psum" },
        { w->seeds[0], "seeds" }, { w->alive[0], "alive" }, { w->alt, "alt" }, { w->cells, "cells" }, { w->cols, "cols" },
        { w->queue, "queue" }, { w->next_nodes, "next_nodes" }, { w->tips, "tips" }, { w->prev_starts, "prev_starts" },
        { w->indices, "bt_indices" }, { w->rev_ops, "rev" }, { w->gen_store, "gen_store" },
        { w->ext[0].conv.tab, "conv_slots" }, { w->ext[0].conv.tab + conv_tab_slots(P.lim.hash_size), "conv_recs" }, { w->ext[0].conv.pool, "conv_pool" },
        { w->ext[1].conv.tab, "conv_slots" }, { w->ext[1].conv.tab + conv_tab_slots(P.lim.hash_size), "conv_recs" }, { w->ext[1].conv.pool, "conv_pool" },
        { w->aln[0].nodes, "aln" },
    };
    std::sort(pts.begin(), pts.end(), [](const Pt &a, const Pt &b) { return a.p < b.p; });
    for (size_t i = 0; i < pts.size(); ++i) {
        const uint8_t *lo = (const uint8_t *)pts[i].p, *hi = i + 1 < pts.size() ? (const uint8_t *)pts[i + 1].p : base + stride;
        mgx_trace_region(lo, (uint64_t)(hi - lo), pts[i].name);
    }
}
#define TRACE_END_READ() mgx_trace_end_read()
#else
#define TRACE_END_READ() ((void)0)
#endif
} // namespace

extern "C" {

void *emu_graph_create(const mgx_boss_view *view) {
    auto *G = new EmuGraph();
    G->mode = view->mode;
    const uint64_t n = view->n_edges;
    const uint32_t n_blocks = (uint32_t)((n + 1 + 63) / 64);
    G->blocks.resize(n_blocks);
    std::vector<uint32_t> counts((size_t)n_blocks * 6);
    for (uint32_t b = 0; b < n_blocks; ++b) build_block_pass1(b, view->W, view->last, n, G->blocks.data(), counts.data());
    uint64_t tot[6] = { 0, 0, 0, 0, 0, 0 };
    for (uint32_t b = 0; b < n_blocks; ++b)
        for (int c = 0; c < 6; ++c) { uint32_t v = counts[(size_t)b * 6 + c]; counts[(size_t)b * 6 + c] = (uint32_t)tot[c]; tot[c] += v; }
    G->last_hint.assign(tot[5] / 64 + 2, 0);
    uint32_t *wh[4];
    for (int c = 0; c < 4; ++c) { G->w_hint[c].assign(tot[c + 1] / 64 + 2, 0); wh[c] = G->w_hint[c].data(); }
    for (uint32_t b = 0; b < n_blocks; ++b) build_block_pass2(b, G->blocks.data(), counts.data(), G->last_hint.data(), wh);
    DevGraph &g = G->g;
    memset(&g, 0, sizeof(g));
    g.blocks = G->blocks.data();
    g.last_hint = G->last_hint.data();
    for (int c = 0; c < 4; ++c) g.w_hint[c] = G->w_hint[c].data();
    g.n = n; g.n_blocks = n_blocks; g.k = view->k;
    LineCtr ctr = { 0, 0, 0 };
    for (int c = 0; c < SIGMA; ++c) { g.F[c] = (uint32_t)view->F[c]; }
    for (int c = 0; c < SIGMA; ++c) g.NF[c] = rank_last(g, g.F[c], ctr);
    {
        // a deliberately COARSE table on the model's small graphs (a handful of segments): the scan must cope with bad predictions
        const char *e = getenv("MGX_SEL_ANCHOR_MAX");
        g.sel_shift = sel_anchor_shift(tot[5], e ? (uint32_t)atoi(e) : 6u);
        g.sel_n = (uint32_t)(tot[5] >> g.sel_shift) + 2;
        G->sel_anchor.assign(g.sel_n, 0);
        for (uint32_t j = 0; j < g.sel_n; ++j) build_sel_anchor(g, j, g.sel_shift, g.sel_n, (uint32_t)tot[5], G->sel_anchor.data());
        g.sel_anchor = G->sel_anchor.data();
    }
    if (view->valid) {
        G->valid.assign(n_blocks, 0);
        for (uint64_t e = 0; e <= n; ++e) if (view->valid[e]) G->valid[e >> 6] |= 1ull << (e & 63);
        g.valid = G->valid.data();
    }
    std::vector<uint32_t> P(n + 1);
    std::vector<uint8_t> D0(n + 1), D1(n + 1);
    for (uint64_t e = 0; e <= n; ++e) { P[e] = build_parent(g, e); D0[e] = (uint8_t)node_last_value(g, e); }
    const uint32_t m = choose_prefix_len(n, g.k);
    std::vector<uint32_t> key(n + 1, 0);
    G->prefix_tbl.assign(1ull << (2 * m), uint2{ 1u, 0u });
    for (uint64_t e = 0; e <= n; ++e) key[e] = build_key_step(0u, D0[e], m, 0);
    for (uint32_t r = 0; r + 2 < g.k; ++r) {
        for (uint64_t e = 0; e <= n; ++e) D1[e] = D0[P[e]];
        D0.swap(D1);
        if (r + 1 < m) for (uint64_t e = 0; e <= n; ++e) key[e] = build_key_step(key[e], D0[e], m, r + 1);
    }
    for (uint64_t e = 1; e <= n; ++e) build_prefix_entry(key.data(), e, n, G->prefix_tbl.data());
    g.prefix_tbl = G->prefix_tbl.data();
    g.prefix_len = m;
    G->firstc.assign((n + 1 + 7) / 8 + 1, 0);
    for (uint64_t e = 0; e <= n; ++e) G->firstc[e >> 3] |= (uint32_t)(D0[e] & 0xF) << (4 * (e & 7));
    g.firstc = G->firstc.data();
    // PRIMARY: terminus | terminus of the ids v + n | palindrome bits | rc_node table (canon_graph.hpp primary_tables())
    const uint64_t n_nodes = rank_last(g, n, ctr);
    G->tables = G->mode == MGX_MODE_PRIMARY && !(getenv("MGX_PRIMARY_TABLES") && atoi(getenv("MGX_PRIMARY_TABLES")) == 0);    // as mgx_graph_create
    G->terminus.assign(G->mode == MGX_MODE_PRIMARY ? (size_t)n_blocks * 3 + (G->tables ? (n_nodes + 2) / 2 + 1 : 0) : (size_t)n_blocks, 0);
    g.terminus = G->terminus.data();
    if (G->mode == MGX_MODE_PRIMARY) {
        // the wrapper's degrees (as k_terminus_primary in mgx.hip)
        for (uint64_t v = 1; v <= n; ++v) {
            if (!in_graph(g, v)) continue;
            const uint32_t t = build_terminus_primary(g, v);
            if (t & 1) G->terminus[v >> 6] |= 1ull << (v & 63);
            if (t & 2) G->terminus[n_blocks + (v >> 6)] |= 1ull << (v & 63);
        }
        if (G->tables) {                                       // as k_primary_tables in mgx.hip
            uint32_t *rcn = reinterpret_cast<uint32_t *>(G->terminus.data() + 3ull * n_blocks);
            for (uint64_t e = 1; e <= n; ++e) {
                if (build_pal_bit(g, e)) G->terminus[2ull * n_blocks + (e >> 6)] |= 1ull << (e & 63);
                if (view->last[e]) rcn[rank_last(g, e, ctr)] = build_rc_node(g, e);
            }
        }
        return G;
    }
    for (uint64_t v = 1; v <= n; ++v)
        if (in_graph(g, v) && build_terminus(g, v)) G->terminus[v >> 6] |= 1ull << (v & 63);
    return G;
}

void emu_graph_free(void *h) { delete static_cast<EmuGraph *>(h); }

// graph primitives for layer tests
uint64_t emu_fwd(void *h, uint64_t i, uint32_t c) { LineCtr ctr = { 0, 0, 0 }; return fwd(static_cast<EmuGraph *>(h)->g, i, c, ctr); }
uint64_t emu_bwd(void *h, uint64_t i) { LineCtr ctr = { 0, 0, 0 }; return bwd(static_cast<EmuGraph *>(h)->g, i, ctr); }
uint32_t emu_first_char(void *h, uint64_t e) { LineCtr ctr = { 0, 0, 0 }; return first_char(static_cast<EmuGraph *>(h)->g, e, ctr); }
int emu_terminus(void *h, uint64_t v) { auto &g = static_cast<EmuGraph *>(h)->g; return (g.terminus[v >> 6] >> (v & 63)) & 1; }
uint32_t emu_outgoing(void *h, uint64_t v, int rc, uint64_t *nodes, char *chars) {
    auto &g = static_cast<EmuGraph *>(h)->g;
    LineCtr ctr = { 0, 0, 0 };
    uint64_t nn[5]; uint32_t cc[5];
    int n = rc ? incoming(g, v, nn, cc, ctr) : outgoing(g, v, nn, cc, ctr);
    for (int t = 0; t < n; ++t) { nodes[t] = nn[t]; chars[t] = rc ? (char)complement_char(decode_code(cc[t])) : (char)decode_code(cc[t]); }
    return (uint32_t)n;
}
// CanonicalDBG::call_outgoing_kmers through canon_graph.hpp (wrapper ids); returns count, *sentinel = the degree counts' flag
uint32_t emu_canon_children(void *h, uint64_t v, uint64_t *nodes, char *chars, int *sentinel) {
    auto &g = static_cast<EmuGraph *>(h)->g;
    LineCtr ctr = { 0, 0, 0 };
    const bool is_rc = v > g.n;
    Spell sp = base_spelling(g, is_rc ? v - g.n : v, ctr);
    if (is_rc) sp = spell_reverse_complement(sp, (int32_t)g.k);
    uint32_t nn[4]; uint8_t cc[4]; bool sent;
    int n = static_cast<EmuGraph *>(h)->tables ? canon_children_tables(g, (uint32_t)v, nn, cc, &sent, ctr)
                                               : canon_children(g, (uint32_t)v, sp, nn, cc, &sent, ctr);
    for (int t = 0; t < n; ++t) { nodes[t] = nn[t]; chars[t] = (char)decode_code(cc[t]); }
    *sentinel = sent;
    return (uint32_t)n;
}
// bit 0: MEM terminus of wrapper id v (v <= n: base id, else v - n's reverse complement)
int emu_terminus_primary(void *h, uint64_t v) {
    auto &g = static_cast<EmuGraph *>(h)->g;
    if (v > g.n) return (g.terminus[g.n_blocks + ((v - g.n) >> 6)] >> ((v - g.n) & 63)) & 1;
    return (g.terminus[v >> 6] >> (v & 63)) & 1;
}
// the lane-parallel conservative filter on a bare string (a Wave with just the fields it reads)
int emu_maybe_low_complexity(const char *s, uint32_t len) {
    auto w = std::make_unique<Wave>();
    std::vector<uint8_t> q(len + 16, 0), tc(len + 16);
    std::vector<uint64_t> eq(len + 16);
    memcpy(q.data(), s, len);
    w->L = (int32_t)len;
    w->q[0] = q.data();
    w->dust_t = tc.data();
    w->dust_eq = eq.data();
    return maybe_low_complexity(*w, 0);
}
int emu_is_low_complexity(const char *s, uint32_t len) {
    SdustScratch sd;
    return is_low_complexity((const uint8_t *)s, (int32_t)len, &sd);
}

// columns[j]: ceil(n_rows / 64) words, bit r = row r has label j (as mgx_annotation_create)
void *emu_annotation_create(uint64_t n_rows, uint32_t n_labels, const uint64_t *const *columns) {
    auto *A = new EmuAnno();
    A->n_rows = n_rows; A->n_labels = n_labels;
    A->head.assign(n_rows + 1, 0); A->count.assign(n_rows + 1, 0);
    for (uint32_t j = 0; j < n_labels; ++j)
        for (uint64_t r = 0; r < n_rows; ++r) if ((columns[j][r >> 6] >> (r & 63)) & 1) ++A->count[r];
    std::vector<uint64_t> off(n_rows + 1, 0);
    uint64_t tot = 0;
    for (uint64_t r = 0; r < n_rows; ++r) { off[r] = tot; if (A->count[r] >= 2) tot += A->count[r]; }
    A->more.assign(tot + 1, 0);
    std::vector<uint32_t> fill(n_rows + 1, 0);
    for (uint32_t j = 0; j < n_labels; ++j)
        for (uint64_t r = 0; r < n_rows; ++r) {
            if (!((columns[j][r >> 6] >> (r & 63)) & 1)) continue;
            if (A->count[r] == 1) A->head[r] = 1ull | ((uint64_t)j << 16);
            else A->more[off[r] + fill[r]++] = j;
        }
    for (uint64_t r = 0; r < n_rows; ++r)
        if (A->count[r] >= 2) A->head[r] = (uint64_t)(A->count[r] > 0xFFFE ? 0xFFFF : A->count[r]) | (off[r] << 16);
    return A;
}
void emu_annotation_free(void *a) { delete static_cast<EmuAnno *>(a); }

void *emu_align_anno(void *gh, void *anno, const mgx_config *config, const mgx_limits *limits, const char *seqs, const uint64_t *offsets,
                     uint64_t n, int map_only);
void *emu_align(void *gh, const mgx_config *config, const mgx_limits *limits, const char *seqs, const uint64_t *offsets,
                uint64_t n, int map_only) {
    return emu_align_anno(gh, nullptr, config, limits, seqs, offsets, n, map_only);
}
// anno != nullptr: label-aware alignment (LabeledAligner) with that label matrix
void *emu_align_anno(void *gh, void *anno, const mgx_config *config, const mgx_limits *limits, const char *seqs, const uint64_t *offsets,
                     uint64_t n, int map_only) {
    auto *G = static_cast<EmuGraph *>(gh);
    auto *AN = static_cast<EmuAnno *>(anno);
    auto *R = new EmuRun();
    mgx_config cfg;
    DevConfig dcfg;
    int rc = prepare_config(*config, G->g.k, &cfg, &dcfg, &R->error, AN != nullptr);
    if (rc) return R;
    if (G->mode == MGX_MODE_CANONICAL) { dcfg.canonical = 1; dcfg.fwd_and_rc = 1; }      // as mgx_aligner_create
    if (G->mode == MGX_MODE_PRIMARY) { dcfg.canonical = G->tables ? 3 : 2; dcfg.fwd_and_rc = 1; }
    const uint32_t k = G->g.k;
    uint32_t Lmax = 0;
    R->node_begin.assign(n + 1, 0);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t L = offsets[i + 1] - offsets[i];
        Lmax = std::max<uint32_t>(Lmax, (uint32_t)L);
        R->node_begin[i + 1] = R->node_begin[i] + (L >= k ? L - k + 1 : 0);
    }
    std::vector<uint32_t> nf(R->node_begin[n] + 1, 0), nr(R->node_begin[n] + 1, 0);
    std::vector<uint8_t> lf(R->node_begin[n] + 1, MLEN_UNKNOWN), lr(R->node_begin[n] + 1, MLEN_UNKNOWN);
    std::vector<uint2> gf(R->node_begin[n] + 1), gr(R->node_begin[n] + 1);
    bool mapped = cfg.max_seed_length >= k;
    memset(&R->stats, 0, sizeof(R->stats));
    std::vector<uint64_t> pkf, pkr;          // 2-bit packed strands + invalid flags (k_pack_reads); the lane-per-read path reads them too
    std::vector<uint32_t> ivf, ivr;
    bool have_packed = false;
    if (mapped) {
        // the product's k_map: persistent lanes stepping the per-lane state machine (here: one lane)
        const bool do_rc = map_only || dcfg.fwd_and_rc;
        const uint64_t n_chains = do_rc ? 2 * n : n;
        uint64_t cursor = 0;
        LineCtr ctr = { 0, 0, 0 };
        if (k <= 32 && !getenv("MGX_MAP_BYTES")) {
            // the product's packed path: k_pack_reads, then k_map_packed (same word layout)
            const uint64_t words = (offsets[n] >> 5) + n + 2;
            pkf.assign(words, 0); pkr.assign(words, 0); ivf.assign(words, 0); ivr.assign(words, 0);
            have_packed = true;
            for (uint64_t r = 0; r < n; ++r) {
                const int32_t L = (int32_t)(offsets[r + 1] - offsets[r]);
                for (int32_t j = 0; 32 * j < L; ++j) {
                    const uint64_t wd = packed_word_begin(offsets[r], r) + (uint64_t)j;
                    pack_read_word(seqs + offsets[r], L, 0, j, &pkf[wd], &ivf[wd]);
                    pack_read_word(seqs + offsets[r], L, 1, j, &pkr[wd], &ivr[wd]);
                }
            }
            if (!getenv("MGX_MAP_LANES")) {
                // the product's k_map_pipe: the request / response machine of map_pipe.hpp (here: one lane)
                MapArgs ma;
                ma.offsets = offsets; ma.node_begin = R->node_begin.data();
                ma.pk_fwd = pkf.data(); ma.pk_rc = pkr.data(); ma.iv_fwd = ivf.data(); ma.iv_rc = ivr.data();
                ma.nodes_fwd = nf.data(); ma.nodes_rc = nr.data(); ma.mlen_fwd = lf.data(); ma.mlen_rc = lr.data();
                ma.rng_fwd = gf.data(); ma.rng_rc = gr.data();
                ma.min_rng_len = (int32_t)std::min<uint64_t>(cfg.min_seed_length, 1u << 20);
                ma.n_reads = n; ma.do_rc = do_rc ? 1 : 0;
                unsigned long long cur2 = 0;
                ma.cursor = &cur2;
                MapPipe mp;
                map_pipe_init(mp);
                ChainClaim claim(ma.cursor);
                const SelPredictGlobal pred = { G->g.sel_anchor, G->g.sel_shift };
                while (map_pipe_step(G->g, ma, mp, ctr, claim, pred)) {}
            } else {
            MapLanePacked m;
            m.state = 0;
            auto fetch = [&](MapLanePacked &ml) -> bool {
                uint64_t c = cursor++;
                if (c >= n_chains) return false;
                uint64_t read = do_rc ? (c >> 1) : c;
                const int strand = do_rc ? (int)(c & 1) : 0;
                const int32_t L = (int32_t)(offsets[read + 1] - offsets[read]);
                const uint64_t wd = packed_word_begin(offsets[read], read);
                ml.pk = (strand ? pkr.data() : pkf.data()) + wd;
                ml.iv = (strand ? ivr.data() : ivf.data()) + wd;
                ml.n_words = (L + 31) >> 5;
                ml.out = (strand ? nr.data() : nf.data()) + R->node_begin[read];
                ml.out_len = (strand ? lr.data() : lf.data()) + R->node_begin[read];
                ml.out_rng = (strand ? gr.data() : gf.data()) + R->node_begin[read];
                ml.min_rng_len = (int32_t)std::min<uint64_t>(cfg.min_seed_length, 1u << 20);
                ml.n_kmers = L - (int32_t)k + 1;
                return true;
            };
            while (m.state != 3) map_lane_step_packed(G->g, m, ctr, fetch);
            }
        } else {
        MapLane m;
        m.state = 0;
        auto fetch = [&](MapLane &ml) -> bool {
            uint64_t c = cursor++;
            if (c >= n_chains) return false;
            uint64_t read = do_rc ? (c >> 1) : c;
            ml.strand = do_rc ? (int)(c & 1) : 0;
            ml.L = (int32_t)(offsets[read + 1] - offsets[read]);
            ml.seq = seqs + offsets[read];
            ml.out = (ml.strand ? nr.data() : nf.data()) + R->node_begin[read];
            ml.out_len = (ml.strand ? lr.data() : lf.data()) + R->node_begin[read];
            ml.out_rng = (ml.strand ? gr.data() : gf.data()) + R->node_begin[read];
            ml.min_rng_len = (int32_t)std::min<uint64_t>(cfg.min_seed_length, 1u << 20);
            ml.n_kmers = ml.L - (int32_t)k + 1;
            return true;
        };
        while (m.state != 3) map_lane_step(G->g, m, ctr, fetch);
        }
        R->stats.rank_lines += ctr.rank_lines; R->stats.select_lines += ctr.select_lines;
        if (G->mode == MGX_MODE_PRIMARY && do_rc) {
            // the product's k_canon_merge
            for (uint64_t r = 0; r < n; ++r) {
                const int32_t L = (int32_t)(offsets[r + 1] - offsets[r]);
                for (int32_t i = 0; i + (int32_t)k <= L; ++i)
                    canon_merge_pair(G->g, seqs + offsets[r], L, i, nf.data() + R->node_begin[r], nr.data() + R->node_begin[r]);
            }
        }
    }
    R->m_fwd.assign(nf.begin(), nf.end());
    R->m_rc.assign(nr.begin(), nr.end());
    R->m_len[0] = lf; R->m_len[1] = lr; R->m_rng[0] = gf; R->m_rng[1] = gr;
    if (map_only) return R;
    // (MGX_EMU_LABEL_SCALE: the label arenas' multiplier, what mgx_align_batch's capacity retry doubles per attempt)
    const char *ls_env = getenv("MGX_EMU_LABEL_SCALE");
    rc = derive_limits(cfg, limits, Lmax, &R->lim, &R->error, AN != nullptr, ls_env ? (uint32_t)atoi(ls_env) : 1u);
    if (rc) return R;
    const uint64_t stride = arena_bytes(R->lim);
    std::vector<uint8_t> arena(stride, 0);
    std::vector<int8_t> sm(128 * 128);
    memcpy(sm.data(), cfg.score_matrix, 128 * 128);
    R->results.resize(n);
    uint64_t out_words = n * ((uint64_t)R->lim.Lmax * 3 + 64) * std::max<uint64_t>(1, AN ? R->lim.lab_pool : (cfg.post_chain_alignments ? (uint64_t)post_chain_capacity(cfg.num_alternative_paths) : cfg.num_alternative_paths)) + 1024;
    R->stream.assign(out_words, 0);
    R->seeds.assign(n * 2 * (uint64_t)R->lim.max_seeds, DevSeed{ 0, 0, 0, 0, 0 });
    unsigned long long cursors[2] = { 0, 0 };
    AlignParams P;
    memset(&P, 0, sizeof(P));
    std::vector<uint32_t> anno_base;
    P.g = G->g; P.cfg = dcfg; P.lim = R->lim; P.score_matrix = sm.data();
    P.seqs = seqs; P.offsets = offsets; P.node_begin = R->node_begin.data();
    P.nodes_fwd = nf.data(); P.nodes_rc = nr.data(); P.n_reads = n;
    P.mlen_fwd = lf.data(); P.mlen_rc = lr.data();
    P.rng_fwd = gf.data(); P.rng_rc = gr.data();
    P.arena = arena.data(); P.arena_stride = stride;
    P.results = R->results.data(); P.out_stream = R->stream.data(); P.out_capacity = out_words;
    P.out_cursor = &cursors[0]; P.read_cursor = &cursors[1]; P.stats = &R->stats; P.dbg_seeds = R->seeds.data();
    P.no_fast = getenv("MGX_NO_FAST") && atoi(getenv("MGX_NO_FAST")) == 1;     // every column through the general path
    if (have_packed && !getenv("MGX_EMU_NO_PKW")) {              // (as mgx.hip: the strands of k_pack_reads for the seeding phase)
        P.pkw[0] = pkf.data(); P.ivw[0] = ivf.data();
        if (dcfg.fwd_and_rc) { P.pkw[1] = pkr.data(); P.ivw[1] = ivr.data(); }
    }
    if (AN) {
        P.labeled = 1; P.no_alias = 1;        // as mgx.hip
        if (G->mode == MGX_MODE_CANONICAL) {   // k_canon_repr
            anno_base.assign(G->g.n + 1, 0);
            for (uint64_t v = 1; v <= G->g.n; ++v) anno_base[v] = canon_repr_node(G->g, v);
            P.anno_base = anno_base.data();
        }
        {
            bool clean = true;                 // k_anno_dummy_rows: no dummy node's row holds a label
            LineCtr lc = { 0, 0, 0 };
            for (uint64_t v = 1; v <= G->g.n && v - 1 < AN->n_rows && clean; ++v)
                if (AN->head[v - 1] && get_W(G->g, v, lc) == 0) clean = false;
            if (clean && !getenv("MGX_EMU_LAB_WCHECK")) P.labeled |= 2u;      // (MGX_EMU_LAB_WCHECK=1: keep the per-access test)
        }
        P.anno_head = AN->head.data(); P.anno_count = AN->count.data(); P.anno_more = AN->more.data(); P.anno_rows = AN->n_rows;
    }
    P.no_compact = getenv("MGX_NO_COMPACT") && atoi(getenv("MGX_NO_COMPACT")) == 1;
    P.no_alias = getenv("MGX_NO_ALIAS") && atoi(getenv("MGX_NO_ALIAS")) == 1;
    P.no_bt_runs = getenv("MGX_NO_BT_RUNS") && atoi(getenv("MGX_NO_BT_RUNS")) == 1;
    auto w = std::make_unique<Wave>();
    SdustScratch sd;
    // model a small LDS so that both placements (LDS / arena) of the fast arrays are exercised
    const char *lds_env = getenv("MGX_EMU_LDS");                 // tests vary the modelled LDS size (0 = everything in the arena)
    std::vector<uint8_t> lds(lds_env ? (size_t)atoi(lds_env) + 16 : 2048);
    std::vector<int8_t> rows(6 * 128);
    load_score_rows(P, rows.data());
    const char *split = getenv("MGX_EMU_SPLIT");
    // the extension phase of one read: through the flat group loop (the product's default with one alignment per seed; a
    // single group here, so every transition of the state machine is exercised, not the interleaving) or the per-read program
    const bool flat = cfg.num_alternative_paths == 1 && !cfg.post_chain_alignments && !AN && !(getenv("MGX_NO_FLAT") && atoi(getenv("MGX_NO_FLAT")) == 1);
    const uint32_t ldsb_all = (uint32_t)(lds_env ? atoi(lds_env) : 2048);
    auto run_extend = [&](uint64_t read, const uint8_t *rec) {
        if (!flat) { align_read<PH_EXTEND>(*w, P, read, 0, &R->stats, &sd, rows.data(), lds.data(), ldsb_all, rec); return; }
#if !(defined(MGX_PARAMS_IN_LDS) && MGX_PARAMS_IN_LDS)
        w->sm_rows = rows.data();
#endif
        ChainRegs CR;
        chain_regs_reset(CR);
        bool given = false;
        uint64_t cur = 0;
        w->fs.act = ACT_FETCH;
        auto fetch = [&](uint64_t &r, const uint8_t *&rc) { if (given) return false; given = true; r = read; rc = rec; return true; };
        do { flat_iteration(*w, P, 0, &R->stats, lds.data(), ldsb_all, CR, cur, fetch); } while (w->fs.act != ACT_EXIT);
    };
    if (split && *split == '1') {
        // the two-kernel pipeline: seeding phase for every read, stable sort by predicted work, extension phase
        std::vector<SeedHdr> hdr(n);
        std::vector<DevSeed> sstream(n * 2 * (uint64_t)R->lim.max_seeds + 16);
        std::vector<uint32_t> key(n), order(n);
        unsigned long long seed_cursor = 0;
        P.seed_hdr = hdr.data(); P.seed_stream = sstream.data(); P.seed_capacity = sstream.size();
        P.seed_cursor = &seed_cursor; P.work_key = key.data();
        // MGX_EMU_SEEDLANE=1: the lane-per-read seeder in front of the wave program's, as mgx.hip launches it (seed_lane_enabled()
        // decides whether the batch qualifies): read i runs as lane i % 64 of one wavefront's interleaved seed buffer; what it
        // leaves is seeded by the wave program from scratch
        std::vector<uint8_t> seeded(n, 0);
        {
            const char *sl_env = getenv("MGX_EMU_SEEDLANE");
            if (sl_env && *sl_env == '1' && P.pkw[0] && (!dcfg.fwd_and_rc || P.pkw[1])
                    && seed_lane_enabled(dcfg, k, R->lim.Lmax, true, true)) {
                const bool long_reads = R->lim.Lmax > (uint32_t)SL_SHORT_L;       // (as mgx.hip picks the kernel build)
                const int32_t qwords = long_reads ? SL_QWORDS_LONG : SL_QWORDS_SHORT, max_l = long_reads ? SL_MAX_L : SL_SHORT_L;
                std::vector<uint64_t> sq(2 * qwords);
                std::vector<uint32_t> scnt(16);
                R->seedlane_ran = 1;
                // (MGX_EMU_SEEDLANE_ONE=1: the first pass only)
                const bool two = !(getenv("MGX_EMU_SEEDLANE_ONE") && atoi(getenv("MGX_EMU_SEEDLANE_ONE")) == 1);
                std::vector<uint64_t> todo(n), left;
                std::vector<uint8_t> why1(n, 0);             // why the first pass left the read (the second pass treats the DUST ones apart)
                for (uint64_t i = 0; i < n; ++i) todo[i] = i;
                for (int ps = 0; ps < (two ? 2 : 1); ++ps) {
                    const bool many = (uint64_t)k >= dcfg.max_seed_length;
                    const int32_t me = ps ? (long_reads ? SL_SEEDS_2_LONG : SL_SEEDS_2) : many ? (long_reads ? SL_SEEDS_1_MANY_LONG : SL_SEEDS_1_MANY) : SL_SEEDS_1;
                    const int32_t mp = ps ? (long_reads ? SL_PENDING_2_LONG : SL_PENDING_2) : many ? SL_PENDING_1_MANY : SL_PENDING_1;
                    std::vector<uint32_t> sbuf(seed_lane_wave_scratch_words((uint32_t)me, (uint32_t)mp), 0);
                    left.clear();
                    for (size_t x = 0; x < todo.size(); ++x) {
                        const uint64_t i = todo[x];
                        SeedLaneChip chip = { sq.data(), 1, qwords, max_l, sbuf.data() + (x % 64), 64, me, mp, !ps ? 0 : why1[i] == 4 ? 1 : 2, scnt.data(), 1 };
                        SeedLaneOut so;
                        so.reason = 0;
                        if (seed_lane_read(P, i, chip, so) == SL_DONE) {
                            const uint64_t at = seed_cursor;
                            seed_cursor += (uint64_t)(so.n_seeds[0] + so.n_seeds[1]);
                            seed_lane_publish(P, i, chip, so, at);
                            R->stats.seeds += (uint64_t)(so.n_seeds[0] + so.n_seeds[1]);
                            seeded[i] = 1;
                            ++R->seedlane_done;
                        } else {
                            left.push_back(i);
                            why1[i] = (uint8_t)so.reason;
                            if (ps || !two) ++R->seedlane_bail[so.reason & 15u];
                        }
                        R->stats.rank_lines += so.ctr.rank_lines; R->stats.select_lines += so.ctr.select_lines; R->stats.bit_lines += so.ctr.bit_lines;
                    }
                    todo = left;
                }
            }
        }
        for (uint64_t i = 0; i < n; ++i)
            if (!seeded[i]) align_read<PH_SEED>(*w, P, i, 0, &R->stats, &sd, rows.data(), lds.data(), (uint32_t)(lds_env ? atoi(lds_env) : 2048));
        for (uint64_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        P.order = order.data();
        const uint32_t ldsb = (uint32_t)(lds_env ? atoi(lds_env) : 2048);
        // MGX_EMU_LANE=1: the lane-per-read kernel in front of the group kernel, as mgx.hip launches it (lane_enabled() decides
        // whether the batch qualifies): every read goes through lane_read() first; the reads it bails on are aligned by the
        // wave program from scratch, in their order
        const char *lane_env = getenv("MGX_EMU_LANE");
        std::vector<uint32_t> lane_rest;
        uint64_t n_ext = n;                  // reads the wave program's extension phase takes
        if (lane_env && *lane_env == '1' && have_packed) {
            LaneParams LP;
            memset(&LP, 0, sizeof(LP));
            std::string why;
            if (lane_enabled(cfg, dcfg, k, R->lim.Lmax, P.no_fast != 0, &LP, &why, P.labeled)) {
                LP.P = P;
                LP.pk[0] = pkf.data(); LP.pk[1] = pkr.data(); LP.iv[0] = ivf.data(); LP.iv[1] = ivr.data();
                LP.max_cols = lane_max_cols(R->lim.Lmax, dcfg.xdrop);
                LP.hash_slots = 4; while (LP.hash_slots < 2 * LP.max_cols) LP.hash_slots *= 2;
                // one wavefront's scratch (column slots and S rows interleaved over 64 lanes, as on the device); read i runs as
                // lane i % 64 of it, so that the interleaved addressing is what the CPU suite exercises
                LP.rest_stride = (lane_rest_bytes(LP.max_cols, LP.hash_slots) + 63) & ~63ull;
                LP.wave_stride = lane_wave_scratch_bytes(LP.max_cols, LP.rest_stride);
                std::vector<uint8_t> scratch_v(LP.wave_stride, 0);
                LP.scratch = scratch_v.data();
                LP.tag_seed = 12345;
                std::vector<uint64_t> qw(LANE_QWORDS);
                std::vector<uint32_t> cold(LANE_COLD_WORDS);
                LaneChip chip = { 0, qw.data(), 1, cold.data(), 1 };
                R->lane_reason.assign(n, 0);
                for (uint64_t i = 0; i < n; ++i) {
                    LaneCounters lc = { 0 };
                    chip.lane = (int32_t)(i % LANE_WAVE);
                    uint8_t *scratch = LP.scratch + 4 * chip.lane;
                    uint32_t *lrec = lane_record(LP, scratch, chip.lane);
                    lrec[27] = 0; lrec[28] = 0;
                    LaneResult LR;
                    memset(&LR, 0, sizeof(LR));
                    int lrc = lane_read(LP, order[i], (uint32_t)i, 0, scratch, chip, lc, LR);
                    if (lrc == LR_AGAIN) lrc = lane_read(LP, order[i], (uint32_t)i, 1, scratch, chip, lc, LR);
                    if (lrc == LR_DONE) {
                        const uint64_t so = cursors[0];
                        cursors[0] += LR.words;
                        lane_emit(LP, order[i], scratch, chip, LR, so);
                        R->stats.columns += LR.rr.n_columns; R->stats.fast_columns += LR.rr.n_columns; R->stats.extensions += LR.rr.n_extensions;
                        R->stats.capacity_errors += LR.rr.status != ST_OK;
                        ++R->lane_done;
                    } else {
                        lane_rest.push_back(order[i]);
                        if (lc.reason < 32) ++R->lane_bail[lc.reason];
                        R->lane_reason[order[i]] = (uint8_t)lc.reason;
                    }
                    R->stats.rank_lines += lrec[27]; R->stats.select_lines += lrec[28];
                }
                order = lane_rest;
                P.order = order.data();
                n_ext = order.size();
                R->lane_ran = 1;
            }
        }
        const char *mp = getenv("MGX_EMU_MULTIPASS");
        std::vector<uint32_t> retry(n + 1);
        unsigned long long retry_count = 0;
        if (AN) {
            // label-aware alignment: one pass, no seed limit (the per-read program)
            for (uint64_t i = 0; i < n_ext; ++i) run_extend(order[i], nullptr);
        } else if (mp && *mp == '1') {
            // the product's multi-pass extension: one seed per read and pass, resume records in between, the retry positions
            // of a pass sorted by their work key for the next one (MGX_EMU_RESUME_CAP: records a pass may write)
            const uint32_t rb = resume_rec_bytes(R->lim, std::max<uint32_t>(1, (uint32_t)cfg.num_alternative_paths));
            const char *cap_env = getenv("MGX_EMU_RESUME_CAP");
            const uint32_t cap = cap_env ? (uint32_t)atoi(cap_env) : (uint32_t)n;
            std::vector<uint8_t> pool_a((size_t)rb * std::max<uint32_t>(cap, 1)), pool_b((size_t)rb * std::max<uint32_t>(cap, 1));
            std::vector<uint32_t> list_a(n + 1), list_b(n + 1), key_a(n + 1), key_b(n + 1), ord(n + 1);
            P.seed_limit = 1; P.resume_rec_bytes = rb; P.resume_cap = cap;
            P.retry_list = list_a.data(); P.retry_key = key_a.data(); P.retry_count = &retry_count; P.resume_out = pool_a.data();
            for (uint64_t i = 0; i < n_ext; ++i)
                run_extend(order[i], nullptr);
            R->retried = retry_count;
            std::vector<uint8_t> *pin = &pool_a, *pout = &pool_b;
            std::vector<uint32_t> *lin = &list_a, *lout = &list_b, *kin = &key_a, *kout = &key_b;
            for (int pass = 1; retry_count; ++pass) {
                const uint64_t c = std::min<uint64_t>(retry_count, cap);
                for (uint64_t i = 0; i < c; ++i) ord[i] = (uint32_t)i;
                std::stable_sort(ord.begin(), ord.begin() + c, [&](uint32_t a, uint32_t b) { return (*kin)[a] < (*kin)[b]; });
                retry_count = 0;
                P.resume_in = pin->data(); P.resume_reads = lin->data();
                P.resume_out = pout->data(); P.retry_list = lout->data(); P.retry_key = kout->data();
                for (uint64_t i = 0; i < c; ++i) {
                    const uint32_t pos = ord[i];
                    run_extend((*lin)[pos], pin->data() + (size_t)pos * rb);
                }
                std::swap(pin, pout); std::swap(lin, lout); std::swap(kin, kout);
            }
        } else {
#ifdef MGX_EMU_TRACE
        if (getenv("MGX_EMU_TRACE_EXTEND")) {
            // traffic model of the product's default extension launch: one pass, no seed limit, only this phase traced
            mgx_trace_reset();
            trace_register_arena(P, arena.data(), stride);
            mgx_trace_region(G->blocks.data(), G->blocks.size() * sizeof(Block), "graph_blocks");
            mgx_trace_region(G->last_hint.data(), G->last_hint.size() * 4, "graph_hints");
            for (int c = 0; c < 4; ++c) mgx_trace_region(G->w_hint[c].data(), G->w_hint[c].size() * 4, "graph_hints");
            mgx_trace_region(G->firstc.data(), G->firstc.size() * 4, "graph_firstc");
            mgx_trace_region(G->terminus.data(), G->terminus.size() * 8, "graph_terminus");
            mgx_trace_region(nf.data(), nf.size() * 4, "in_nodes"); mgx_trace_region(nr.data(), nr.size() * 4, "in_nodes");
            mgx_trace_region(hdr.data(), hdr.size() * sizeof(SeedHdr), "in_seed_hdr");
            mgx_trace_region(sstream.data(), sstream.size() * sizeof(DevSeed), "in_seed_stream");
            mgx_trace_region(seqs, offsets[n], "in_reads");
            mgx_trace_region(R->results.data(), R->results.size() * sizeof(ReadResult), "out_results");
            mgx_trace_region(R->stream.data(), R->stream.size() * 4, "out_stream");
            P.dbg_seeds = nullptr;
            mgx_trace_on(1);
            for (uint64_t i = 0; i < n_ext; ++i) {
                run_extend(order[i], nullptr);
                TRACE_END_READ();
            }
            mgx_trace_on(0);
        } else {
#endif
        // two-pass extension: at most one seed per read first, then the reads that go on, from scratch
        P.seed_limit = 1; P.retry_list = retry.data(); P.retry_count = &retry_count;
        for (uint64_t i = 0; i < n_ext; ++i)
            run_extend(order[i], nullptr);
        P.seed_limit = 0; P.order = retry.data();
        for (uint64_t i = 0; i < retry_count; ++i)
            run_extend(retry[i], nullptr);
        R->retried = retry_count;
#ifdef MGX_EMU_TRACE
        }
#endif
        }
    } else {
        for (uint64_t i = 0; i < n; ++i) align_read(*w, P, i, 0, &R->stats, &sd, rows.data(), lds.data(), (uint32_t)(lds_env ? atoi(lds_env) : 2048));
    }
    R->out_used = std::min<uint64_t>(cursors[0], out_words);
    R->host.decode(R->results.data(), n, R->stream.data(), ~0ull, AN != nullptr);
    return R;
}

// raw device-layout records of the run (what mgx_device_results exposes on the GPU): headers + used stream words
void emu_raw(void *r, const void **headers, uint64_t *n, const uint32_t **stream, uint64_t *words) {
    auto *R = static_cast<EmuRun *>(r);
    *headers = R->results.data(); *n = R->results.size();
    *stream = R->stream.data(); *words = R->out_used;
}
uint64_t emu_retried(void *r) { return static_cast<EmuRun *>(r)->retried; }
const char *emu_error(void *r) { return static_cast<EmuRun *>(r)->error.c_str(); }
void emu_results(void *r, mgx_results *out) { static_cast<EmuRun *>(r)->host.view(out); }
void emu_mapping(void *r, mgx_mapping *out) {
    auto *R = static_cast<EmuRun *>(r);
    out->n_queries = R->node_begin.size() - 1;
    out->node_begin = R->node_begin.data();
    out->nodes_fwd = R->m_fwd.data();
    out->nodes_rc = R->m_rc.data();
}
// k_map's side outputs: per k-mer position and strand the match length byte and, where it was written, the (rl, ru) range
// (entries the kernel does not write hold MLEN_UNKNOWN / the zero range)
uint64_t emu_map_side(void *r, int strand, const uint8_t **mlen, const uint32_t **rng) {
    auto *R = static_cast<EmuRun *>(r);
    *mlen = R->m_len[strand].data();
    *rng = reinterpret_cast<const uint32_t *>(R->m_rng[strand].data());
    return R->m_len[strand].size();
}
// info6 per read + seeds [n][2][max_seeds][4]
uint32_t emu_seed_info(void *r, uint32_t *info6, uint32_t *seeds) {
    auto *R = static_cast<EmuRun *>(r);
    for (size_t i = 0; i < R->results.size(); ++i) {
        const ReadResult &x = R->results[i];
        uint32_t *o = info6 + 6 * i;
        o[0] = x.num_matches_fwd; o[1] = x.num_matches_rc; o[2] = x.n_seeds_fwd; o[3] = x.n_seeds_rc;
        o[4] = x.n_extensions; o[5] = x.n_columns;
    }
    if (seeds)
        for (size_t x = 0; x < R->seeds.size(); ++x) {
            seeds[4 * x] = R->seeds[x].clipping; seeds[4 * x + 1] = R->seeds[x].length; seeds[4 * x + 2] = R->seeds[x].offset;
            seeds[4 * x + 3] = R->seeds[x].offset ? R->seeds[x].node : R->seeds[x].n_nodes;
        }
    return R->lim.max_seeds;
}
void emu_lane_stats(void *r, uint64_t *out2) { out2[0] = static_cast<EmuRun *>(r)->lane_ran; out2[1] = static_cast<EmuRun *>(r)->lane_done; }
void emu_lane_reasons(void *r, uint8_t *out, uint64_t n) { auto &v = static_cast<EmuRun *>(r)->lane_reason; for (uint64_t i = 0; i < n && i < v.size(); ++i) out[i] = v[i]; }
// [0] whether the lane-per-read seeder ran, [1] reads it finished, [2 .. 18) reads it left, by reason
void emu_seedlane_stats(void *r, uint64_t *out18) {
    const EmuRun *R = static_cast<EmuRun *>(r);
    out18[0] = R->seedlane_ran; out18[1] = R->seedlane_done;
    for (int x = 0; x < 16; ++x) out18[2 + x] = R->seedlane_bail[x];
}
void emu_lane_bails(void *r, uint64_t *out32) { for (int x = 0; x < 32; ++x) out32[x] = static_cast<EmuRun *>(r)->lane_bail[x]; }
void emu_stats(void *r, uint64_t *out8) {
    auto &s = static_cast<EmuRun *>(r)->stats;
    out8[0] = s.rank_lines; out8[1] = s.select_lines; out8[2] = s.bit_lines; out8[3] = s.columns;
    out8[4] = s.extensions; out8[5] = s.seeds; out8[6] = s.capacity_errors; out8[7] = s.fast_columns;
}
void emu_free(void *r) { delete static_cast<EmuRun *>(r); }
#ifdef MGX_EMU_TRACE
uint64_t emu_trace_report(char *buf, uint64_t cap) { return mgx_trace_report(buf, cap); }
#endif

} // extern "C"
