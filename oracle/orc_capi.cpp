// ORACLE — TEST INFRASTRUCTURE ONLY.  C API over the CPU restatement so that tests/ (ctypes) and
// bench.py's cpu_baseline leg can drive it.  Result layout = include/mgx.h so that GPU and oracle
// outputs can be compared field by field.  Never linked into libmgx.so.
#include <atomic>
#include <malloc.h>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "orc_align.hpp"

using namespace orc;

namespace orc { extern uint64_t g_oob_reads_total(); extern uint64_t g_unfetched_label_lookups_total(); }

namespace {
struct ResultStore {
    std::vector<uint64_t> aln_begin;
    std::vector<mgx_alignment> alignments;
    std::vector<uint64_t> nodes;
    std::vector<mgx_cigar_op> cigar;
    std::string seqs;
    std::vector<int32_t> status;
    // mapping
    std::vector<uint64_t> node_begin, nodes_fwd, nodes_rc;
    // seeds, flattened: per query and strand
    std::vector<uint64_t> seed_begin[2];           // n_queries + 1
    std::vector<uint32_t> seed_meta[2];            // 4 per seed: clipping, length, offset, n_nodes
    std::vector<uint64_t> seed_node_begin[2];      // per seed + 1
    std::vector<uint64_t> seed_nodes[2];
    std::vector<uint64_t> num_matches[2];
    std::string tsv;
    WorkCounters wc;
    std::string error;
    // label-aware runs: the label set of every alignment (same order as `alignments`)
    std::vector<uint64_t> label_begin, labels;
    // ... and, for annotations with coordinates, the coordinates of every label entry: entry e of `labels` has
    // coords[coord_begin[e] .. coord_begin[e + 1]) (Alignment::label_coordinates); kept alignments for format_coords
    std::vector<uint64_t> coord_begin;
    std::vector<int64_t> coords;
    std::vector<AlignmentResults> kept;
    std::string scratch;
};

void flatten(const std::vector<AlignmentResults> &res, ResultStore *st, int32_t min_path_score) {
    st->aln_begin.assign(1, 0);
    st->node_begin.assign(1, 0);
    for (int s = 0; s < 2; ++s) { st->seed_begin[s].assign(1, 0); st->seed_node_begin[s].assign(1, 0); }
    for (const auto &r : res) {
        for (const auto &a : r.alignments) {
            mgx_alignment m;
            std::memset(&m, 0, sizeof(m));
            m.score = a.score; m.offset = a.offset; m.clipping = a.get_clipping();
            m.end_clipping = a.get_end_clipping(); m.num_matches = a.cigar.get_num_matches();
            m.n_nodes = a.nodes.size(); m.n_cigar = a.cigar.ops.size(); m.seq_len = a.sequence.size();
            m.nodes_begin = st->nodes.size(); m.cigar_begin = st->cigar.size(); m.seq_begin = st->seqs.size();
            m.orientation = a.orientation;
            st->nodes.insert(st->nodes.end(), a.nodes.begin(), a.nodes.end());
            for (auto &op : a.cigar.ops) { mgx_cigar_op o; std::memset(&o, 0, sizeof(o)); o.len = op.second; o.op = op.first; st->cigar.push_back(o); }
            st->seqs += a.sequence;
            st->alignments.push_back(m);
            if (st->label_begin.empty()) st->label_begin.push_back(0);
            st->labels.insert(st->labels.end(), a.label_columns.begin(), a.label_columns.end());
            st->label_begin.push_back(st->labels.size());
            if (st->coord_begin.empty()) st->coord_begin.push_back(0);
            for (size_t e = 0; e < a.label_columns.size(); ++e) {
                if (e < a.label_coordinates.size()) st->coords.insert(st->coords.end(), a.label_coordinates[e].begin(), a.label_coordinates[e].end());
                st->coord_begin.push_back(st->coords.size());
            }
        }
        st->aln_begin.push_back(st->alignments.size());
        st->status.push_back(MGX_OK);
        // mapping: both strands padded to the same per-query count
        size_t n = std::max(r.nodes_fwd.size(), r.nodes_rc.size());
        for (size_t i = 0; i < n; ++i) {
            st->nodes_fwd.push_back(i < r.nodes_fwd.size() ? r.nodes_fwd[i] : 0);
            st->nodes_rc.push_back(i < r.nodes_rc.size() ? r.nodes_rc[i] : 0);
        }
        st->node_begin.push_back(st->nodes_fwd.size());
        const std::vector<Seed> *seeds[2] = { &r.seeds_fwd, &r.seeds_rc };
        for (int s = 0; s < 2; ++s) {
            for (const auto &sd : *seeds[s]) {
                st->seed_meta[s].push_back(sd.clipping);
                st->seed_meta[s].push_back(sd.query_view.size());
                st->seed_meta[s].push_back(sd.offset);
                st->seed_meta[s].push_back(sd.nodes.size());
                st->seed_nodes[s].insert(st->seed_nodes[s].end(), sd.nodes.begin(), sd.nodes.end());
                st->seed_node_begin[s].push_back(st->seed_nodes[s].size());
            }
            st->seed_begin[s].push_back(st->seed_meta[s].size() / 4);
        }
        st->num_matches[0].push_back(r.num_matches_fwd);
        st->num_matches[1].push_back(r.num_matches_rc);
    }
    (void)min_path_score;
}
} // namespace

extern "C" {

void *orc_graph_build(uint32_t k, uint32_t n_seqs, const char **seqs, uint32_t mode, int mask_dummy) {
    try {
        auto *g = new Graph();
        std::vector<std::string> v(seqs, seqs + n_seqs);
        g->boss = build_boss(k, v, (Mode)mode);
        g->mode = (Mode)mode;
        if (mask_dummy) g->mask_dummy_kmers();
        return g;
    } catch (...) { return nullptr; }
}

void *orc_graph_from_boss(const mgx_boss_view *view) {
    auto *g = new Graph();
    g->boss.k_ = view->k - 1;
    g->boss.n = view->n_edges;
    g->boss.W.assign(view->W, view->W + view->n_edges + 1);
    g->boss.last.assign(view->last, view->last + view->n_edges + 1);
    for (int c = 0; c < SIGMA; ++c) g->boss.F[c] = view->F[c];
    if (view->valid) g->valid.assign(view->valid, view->valid + view->n_edges + 1);
    g->mode = (Mode)view->mode;
    g->boss.finalize();
    return g;
}

// precompute the first-character table (what NodeFirstCache amortises in the reference)
void orc_graph_build_first_chars(void *h, uint32_t threads) { static_cast<Graph *>(h)->build_first_chars(threads); }

void orc_graph_free(void *h) { delete static_cast<Graph *>(h); }
uint64_t orc_graph_num_edges(void *h) { return static_cast<Graph *>(h)->boss.n; }
uint32_t orc_graph_k(void *h) { return static_cast<Graph *>(h)->get_k(); }
uint64_t orc_graph_num_nodes(void *h) { return static_cast<Graph *>(h)->num_nodes(); }
int orc_graph_has_mask(void *h) { return !static_cast<Graph *>(h)->valid.empty(); }

// copies W/last (n_edges + 1 bytes each), F (5), valid (n_edges + 1, only if the graph has a mask)
void orc_graph_export(void *h, uint8_t *W, uint8_t *last, uint64_t *F, uint8_t *valid) {
    auto *g = static_cast<Graph *>(h);
    std::memcpy(W, g->boss.W.data(), g->boss.n + 1);
    std::memcpy(last, g->boss.last.data(), g->boss.n + 1);
    for (int c = 0; c < SIGMA; ++c) F[c] = g->boss.F[c];
    if (valid && !g->valid.empty()) std::memcpy(valid, g->valid.data(), g->boss.n + 1);
}

// writes the node's k-mer into out (k bytes)
void orc_graph_node_sequence(void *h, uint64_t node, char *out) {
    std::string s = static_cast<Graph *>(h)->get_node_sequence(node);
    std::memcpy(out, s.data(), s.size());
}

// primitives for parity tests of the device BOSS layer; out arrays sized by the caller
uint64_t orc_boss_fwd(void *h, uint64_t i, uint32_t c) { return static_cast<Graph *>(h)->boss.fwd(i, c); }
uint64_t orc_boss_bwd(void *h, uint64_t i) { return static_cast<Graph *>(h)->boss.bwd(i); }
uint64_t orc_boss_rank_W(void *h, uint64_t i, uint32_t c) { return static_cast<Graph *>(h)->boss.rank_W(i, c); }
uint64_t orc_boss_select_last(void *h, uint64_t r) { return static_cast<Graph *>(h)->boss.select_last(r); }
uint64_t orc_boss_rank_last(void *h, uint64_t i) { return static_cast<Graph *>(h)->boss.rank_last(i); }
int orc_graph_has_multiple_outgoing(void *h, uint64_t v) { return static_cast<Graph *>(h)->has_multiple_outgoing(v); }
int orc_graph_has_single_incoming(void *h, uint64_t v) { return static_cast<Graph *>(h)->has_single_incoming(v); }
// children (rc=0) or RC-graph children (rc=1): up to 8 (node, char) pairs; returns count
uint32_t orc_graph_outgoing(void *h, uint64_t v, int rc, uint64_t *nodes, char *chars) {
    GraphView view{ static_cast<Graph *>(h), rc != 0 };
    uint32_t n = 0;
    view.call_outgoing_kmers(v, [&](node_t nn, char c) { if (n < 8) { nodes[n] = nn; chars[n] = c; } ++n; });
    return n;
}
// ---- CanonicalDBG wrapper over the (PRIMARY) graph h, for the wrapper's own KATs (tests/test_oracle_canonical_wrapper.py) ----
// dir: 0 = call_outgoing_kmers, 1 = call_incoming_kmers; up to 8 (node, char) pairs; returns count
uint32_t orc_canonical_adjacent(void *h, uint64_t v, int dir, uint64_t *nodes, char *chars) {
    CanonicalView view(*static_cast<Graph *>(h));
    uint32_t n = 0;
    auto cb = [&](node_t nn, char c) { if (n < 8) { nodes[n] = nn; chars[n] = c; } ++n; };
    if (dir == 0) view.call_outgoing_kmers(v, cb); else view.call_incoming_kmers(v, cb);
    return n;
}
uint64_t orc_canonical_reverse_complement(void *h, uint64_t v) { return CanonicalView(*static_cast<Graph *>(h)).reverse_complement(v); }
void orc_canonical_node_sequence(void *h, uint64_t v, char *out) {
    std::string s = CanonicalView(*static_cast<Graph *>(h)).get_node_sequence(v);
    std::memcpy(out, s.data(), s.size());
}
// map_to_nodes_sequentially; out has len - k + 1 entries
void orc_canonical_map(void *h, const char *seq, uint32_t len, uint64_t *out) {
    auto nodes = CanonicalView(*static_cast<Graph *>(h)).map_to_nodes_sequentially(std::string_view(seq, len));
    for (size_t i = 0; i < nodes.size(); ++i) out[i] = nodes[i];
}
int orc_canonical_degrees(void *h, uint64_t v) {
    CanonicalView view(*static_cast<Graph *>(h));
    return (view.has_multiple_outgoing(v) ? 1 : 0) | (view.has_single_incoming(v) ? 2 : 0);
}

// suffix matching; returns number of nodes (written up to cap), *match_len = matched length
uint32_t orc_graph_suffix_match(void *h, const char *str, uint32_t len, uint32_t min_len, uint64_t max_matches,
                                uint64_t *nodes, uint32_t cap, uint32_t *match_len) {
    uint32_t n = 0;
    *match_len = 0;
    static_cast<Graph *>(h)->call_nodes_with_suffix_matching_longest_prefix(
        std::string_view(str, len),
        [&](node_t v, uint64_t l) { if (n < cap) nodes[n] = v; ++n; *match_len = l; },
        min_len, max_matches ? max_matches : SIZE_MAX);
    return n;
}

int orc_is_low_complexity(const char *s, uint32_t len) { return is_low_complexity(std::string_view(s, len)); }

// Definition-level check of "sdust masks something" (Morgulis et al. 2006, as implemented by lh3/sdust with T = 20, W = 64),
// written WITHOUT the incremental machinery (windows, running counts, perfect-interval list, L-suffix shortcut) that the
// oracle's and the device's sdust restate: a sequence has a masked region iff some interval of at most W - 2 consecutive
// triplets has  10 * (number of equal triplet pairs) > T * (number of triplets - 1)  — any such interval contains a perfect
// interval (its highest-scoring subinterval), and every perfect interval is such an interval.
// One property of the published implementation is part of the definition used here: a non-ACGT character restarts the
// WORD (no triplet spans it) but not the WINDOW — sdust.c resets only `l` and `t` there, the triplet deque and its
// counts live on — so "consecutive triplets" means consecutive among the triplets that exist, across such characters.
int orc_sdust_bruteforce(const char *s, uint32_t len) {
    const int T = 20, W = 64;
    auto code = [](char c) -> int {
        switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                     case 'T': case 't': case 'U': case 'u': return 3; default: return 4; }
    };
    std::vector<int> trip;                      // the triplets that exist, in order
    for (uint32_t i = 0; i + 3 <= len; ++i) {
        int a = code(s[i]), b = code(s[i + 1]), c = code(s[i + 2]);
        if (!((a | b | c) & 4)) trip.push_back((a << 4) | (b << 2) | c);
    }
    const int n = (int)trip.size();
    for (int i = 0; i < n; ++i) {
        int cnt[64] = { 0 };
        long r = 0;
        for (int j = i; j < n && j - i + 1 <= W - 2; ++j) {
            r += cnt[trip[j]]++;                // pairs (x, j), x < j, with equal triplets
            if (r * 10 > (long)T * (j - i)) return 1;
        }
    }
    return 0;
}
int orc_check_config(const mgx_config *c) { return check_config_scores(*c); }
uint64_t orc_oob_reads() { return g_oob_reads_total(); }
// label look-ups of nodes the reference's AnnotationBuffer was never asked to fetch (CANONICAL-mode graphs: undefined upstream)
uint64_t orc_unfetched_label_lookups() { return g_unfetched_label_lookups_total(); }

// Align a batch with `threads` worker threads (thread pool over sub-batches like cli/align.cpp:415-480).
// validate != 0 additionally runs Alignment::is_valid on every result (alignment.cpp:1316-1345).
void *orc_align_batch(void *h, const mgx_config *config, const char *seqs, const uint64_t *offsets,
                      uint64_t n, uint32_t threads, int validate) {
    auto *g = static_cast<Graph *>(h);
    auto *st = new ResultStore();
    try {
        Aligner aligner(*g, *config);
        std::vector<std::string> queries(n);
        for (uint64_t i = 0; i < n; ++i) queries[i].assign(seqs + offsets[i], seqs + offsets[i + 1]);
        std::vector<AlignmentResults> res(n);
        if (threads > 1) {
            // glibc malloc under hundreds of threads: keep the per-thread arenas from returning memory to the kernel
            // between reads (trim / mmap-threshold churn serialises on the process's mmap lock).  The reference links
            // jemalloc for the same reason (M/CMakeLists.txt:515-518).
            mallopt(M_MMAP_THRESHOLD, 1 << 30);
            mallopt(M_TRIM_THRESHOLD, 1 << 30);
            mallopt(M_TOP_PAD, 64 << 20);
        }
        if (threads <= 1) {
            aligner.align_batch(queries, &res, &st->wc);
        } else {
            std::atomic<uint64_t> next{ 0 };
            // small tasks keep the tail short when threads >> tasks; counters are padded to their own cache lines
            // (adjacent 56-byte counter blocks bumped once per BOSS step would ping-pong between cores)
            const uint64_t chunk = std::max<uint64_t>(8, std::min<uint64_t>(256, n / (8ull * threads) + 1));
            std::vector<std::thread> pool;
            struct alignas(128) PaddedWC { WorkCounters w; };
            std::vector<PaddedWC> wcs(threads);
            for (uint32_t t = 0; t < threads; ++t) {
                pool.emplace_back([&, t]() {
                    for (;;) {
                        uint64_t b = next.fetch_add(chunk);
                        if (b >= n) break;
                        uint64_t e = std::min(n, b + chunk);
                        std::vector<std::string> sub(queries.begin() + b, queries.begin() + e);
                        std::vector<AlignmentResults> r;
                        aligner.align_batch(sub, &r, &wcs[t].w);
                        for (uint64_t i = b; i < e; ++i) res[i] = std::move(r[i - b]);
                    }
                });
            }
            for (auto &th : pool) th.join();
            for (auto &w : wcs) st->wc.add(w.w);
        }
        if (validate) {
            const CanonicalView canon(*g);                                 // PRIMARY graphs are seen through CanonicalDBG
            GraphView view{ g, false, g->mode == PRIMARY ? &canon : nullptr };
            const mgx_config &cfg = aligner.get_config();
            for (uint64_t i = 0; i < n; ++i)
                for (auto &a : res[i].alignments) {
                    std::string why;
                    if (!a.is_valid(view, &cfg, &why)) { st->error = "query " + std::to_string(i) + ": " + why; break; }
                }
        }
        flatten(res, st, config->min_path_score);
        for (uint64_t i = 0; i < n; ++i) st->tsv += format_alignment_tsv(std::to_string(i), res[i], config->min_path_score);
    } catch (const std::exception &e) {
        st->error = e.what();
    }
    return st;
}

// ---- label-aware alignment (LabeledAligner, A/aligner_labeled.{hpp,cpp}; annotation without coordinates) ----
void *orc_annotation_create(void *graph, uint32_t n_labels) {
    auto *g = static_cast<Graph *>(graph);
    auto *a = new Annotation();
    a->resize(g->boss.n, n_labels);                      // rows = graph_to_anno_index(max_index) + 1
    return a;
}
void orc_annotation_free(void *a) { delete static_cast<Annotation *>(a); }
// AnnotatedDBG::annotate_sequence(sequence, { label })
int orc_annotation_annotate(void *a, void *graph, const char *seq, uint32_t len, uint32_t label, char *err, uint32_t err_cap) {
    try { static_cast<Annotation *>(a)->annotate_sequence(*static_cast<Graph *>(graph), std::string_view(seq, len), label); }
    catch (const std::exception &e) { if (err && err_cap) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; } return 1; }
    return 0;
}
void orc_annotation_set(void *a, uint64_t row, uint32_t label) { static_cast<Annotation *>(a)->set(row, label); }
// BinaryMatrix::get_rows for `n` rows: out_begin[n + 1], labels appended to out_labels (capacity cap); returns the label count
uint64_t orc_annotation_get_rows(void *a, const uint64_t *rows, uint64_t n, uint64_t *out_begin, uint64_t *out_labels, uint64_t cap) {
    std::vector<uint64_t> r(rows, rows + n);
    auto res = static_cast<Annotation *>(a)->get_rows(r);
    uint64_t total = 0;
    out_begin[0] = 0;
    for (uint64_t i = 0; i < n; ++i) {
        for (uint64_t l : res[i]) { if (total < cap) out_labels[total] = l; ++total; }
        out_begin[i + 1] = total;
    }
    return total;
}
const uint64_t *orc_annotation_column_words(void *a, uint32_t label, uint64_t *n_words) {
    auto *an = static_cast<Annotation *>(a);
    *n_words = an->columns[label].size();
    return an->columns[label].data();
}
void *orc_align_batch_labeled(void *h, const mgx_config *config, void *annotation, const char *seqs, const uint64_t *offsets,
                              uint64_t n, int validate) {
    auto *g = static_cast<Graph *>(h);
    auto *st = new ResultStore();
    try {
        LabeledAligner aligner(*g, *config, *static_cast<Annotation *>(annotation));
        std::vector<std::string> queries(n);
        for (uint64_t i = 0; i < n; ++i) queries[i].assign(seqs + offsets[i], seqs + offsets[i + 1]);
        std::vector<AlignmentResults> res(n);
        aligner.align_batch(queries, &res);
        if (validate) {
            const CanonicalView canon(*g);
            GraphView view{ g, false, g->mode == PRIMARY ? &canon : nullptr };
            const mgx_config &cfg = aligner.get_config();
            for (uint64_t i = 0; i < n; ++i)
                for (auto &a : res[i].alignments) {
                    std::string why;
                    if (!a.is_valid(view, &cfg, &why)) { st->error = "query " + std::to_string(i) + ": " + why; break; }
                }
        }
        flatten(res, st, config->min_path_score);
        st->kept = std::move(res);                       // (the alignments' string_views point into the results' own query strings)
    } catch (const std::exception &e) {
        st->error = e.what();
    }
    return st;
}
// AnnotatedDBG::annotate_kmer_coords for one (sequence, { label }, first coordinate)
int orc_annotation_annotate_coords(void *a, void *graph, const char *seq, uint32_t len, uint32_t label, uint64_t start) {
    try { static_cast<Annotation *>(a)->annotate_kmer_coords(*static_cast<Graph *>(graph), std::string_view(seq, len), label, start); }
    catch (const std::exception &) { return 1; }
    return 0;
}
// coordinates of the label entries of a labeled run (see ResultStore)
void orc_results_coords(void *r, const uint64_t **coord_begin, const int64_t **coords) {
    auto *st = static_cast<ResultStore *>(r);
    if (st->coord_begin.empty()) st->coord_begin.push_back(0);
    *coord_begin = st->coord_begin.data();
    *coords = st->coords.data();
}
// Alignment::format_coords(CoordToHeader(headers, kmer_counts), k) of alignment `ai` of query `q`: one column (label 0 .. ), its
// sequences' headers as a '\n'-separated string per column ('\t' between columns) and their k-mer counts flattened likewise
const char *orc_results_format_coords(void *r, uint64_t q, uint64_t ai, const char *headers, const uint64_t *kmer_counts,
                                      const uint64_t *n_seqs_per_column, uint32_t n_columns, uint32_t k) {
    auto *st = static_cast<ResultStore *>(r);
    std::vector<std::vector<std::string>> hs(n_columns);
    std::vector<std::vector<uint64_t>> kc(n_columns);
    const char *p = headers;
    size_t flat = 0;
    for (uint32_t c = 0; c < n_columns; ++c) {
        for (uint64_t x = 0; x < n_seqs_per_column[c]; ++x) {
            const char *e = p;
            while (*e && *e != '\n' && *e != '\t') ++e;
            hs[c].emplace_back(p, e);
            p = *e ? e + 1 : e;
            kc[c].push_back(kmer_counts[flat++]);
        }
    }
    st->scratch = st->kept.at(q).alignments.at(ai).format_coords(hs, kc, k);
    return st->scratch.c_str();
}
// MultiIntMatrix::get_row_tuples for one row: labels (ascending) with their coordinates; returns the label count
uint64_t orc_annotation_row_tuples(void *a, uint64_t row, uint64_t *labels, uint64_t *coord_begin, int64_t *coords, uint64_t coord_cap) {
    auto t = static_cast<Annotation *>(a)->get_row_tuples(row);
    uint64_t nc = 0;
    coord_begin[0] = 0;
    for (size_t i = 0; i < t.size(); ++i) {
        labels[i] = t[i].first;
        for (int64_t c : t[i].second) { if (nc < coord_cap) coords[nc] = c; ++nc; }
        coord_begin[i + 1] = nc;
    }
    return t.size();
}
// chain_seeds' sort + DP for n_lists anchor lists (the checker of mgx_chain_seeds), in place
void orc_chain_seeds(const mgx_config *config, mgx_chain_anchor *anchors, const uint64_t *list_begin, const uint32_t *query_size,
                     uint64_t n_lists, uint32_t *backtrace) {
    for (uint64_t l = 0; l < n_lists; ++l)
        chain_anchors(*config, query_size[l], anchors + list_begin[l], list_begin[l + 1] - list_begin[l], backtrace + list_begin[l]);
}
// label sets of the alignments of a run, in the order of mgx_results.alignments
void orc_results_labels(void *r, const uint64_t **begin, const uint64_t **labels) {
    auto *st = static_cast<ResultStore *>(r);
    if (st->label_begin.empty()) st->label_begin.push_back(0);
    *begin = st->label_begin.data();
    *labels = st->labels.data();
}

const char *orc_results_error(void *r) { return static_cast<ResultStore *>(r)->error.c_str(); }
void orc_results_view(void *r, mgx_results *out) {
    auto *st = static_cast<ResultStore *>(r);
    out->n_queries = st->status.size();
    out->aln_begin = st->aln_begin.data();
    out->alignments = st->alignments.data();
    out->nodes = st->nodes.data();
    out->cigar = st->cigar.data();
    out->seqs = st->seqs.data();
    out->status = st->status.data();
    out->labels = nullptr;             // (the oracle hands its label sets out through orc_results_labels)
}
void orc_results_mapping(void *r, mgx_mapping *out) {
    auto *st = static_cast<ResultStore *>(r);
    out->n_queries = st->status.size();
    out->node_begin = st->node_begin.data();
    out->nodes_fwd = st->nodes_fwd.data();
    out->nodes_rc = st->nodes_rc.data();
}
// seeds of one strand: begin[n+1], meta[4*n_seeds], node_begin[n_seeds+1], nodes[], num_matches[n]
void orc_results_seeds(void *r, int strand, const uint64_t **begin, const uint32_t **meta,
                       const uint64_t **node_begin, const uint64_t **nodes, const uint64_t **num_matches) {
    auto *st = static_cast<ResultStore *>(r);
    *begin = st->seed_begin[strand].data();
    *meta = st->seed_meta[strand].data();
    *node_begin = st->seed_node_begin[strand].data();
    *nodes = st->seed_nodes[strand].data();
    *num_matches = st->num_matches[strand].data();
}
const char *orc_results_tsv(void *r) { return static_cast<ResultStore *>(r)->tsv.c_str(); }
// n_map_fwd, n_index_steps, n_terminus, n_expansions, n_columns, n_extensions, n_seeds
void orc_results_counters(void *r, uint64_t *out7) {
    auto &w = static_cast<ResultStore *>(r)->wc;
    out7[0] = w.n_map_fwd; out7[1] = w.n_index_steps; out7[2] = w.n_terminus; out7[3] = w.n_expansions;
    out7[4] = w.n_columns; out7[5] = w.n_extensions; out7[6] = w.n_seeds;
}
void orc_results_free(void *r) { delete static_cast<ResultStore *>(r); }

} // extern "C"
