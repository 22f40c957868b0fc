// host_common.hpp — host-side logic of the C-ABI that does not touch the HIP runtime: config
// validation/clamping (DBGAligner ctor), arena limits, and decoding of the device result stream.
#pragma once
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mgx.h"
#include "align_types.hpp"

namespace mgx {

// alignments per query the device keeps with post_chain_alignments: the kernels built for MGX_MAX_ALTERNATIVE_PATHS have
// 4 x that many alignment buffers per query, 3 x num_alternative_paths of them serve the extensions (align_core.hpp N_ALN)
inline uint32_t post_chain_capacity(uint64_t num_alternative_paths) {
    return 4u * (uint32_t)MGX_MAX_ALTERNATIVE_PATHS - 3u * (uint32_t)std::min<uint64_t>(num_alternative_paths, MGX_MAX_ALTERNATIVE_PATHS);
}

inline bool check_config_scores(const mgx_config &c) {
    // aligner_config.cpp:39-66
    int8_t min_penalty = INT8_MAX;
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) min_penalty = std::min(min_penalty, c.score_matrix[i][j]);
    if (c.gap_opening_penalty * 2 >= min_penalty) return false;
    min_penalty = std::min({ min_penalty, c.gap_opening_penalty, c.gap_extension_penalty });
    return (int64_t)c.min_cell_score >= (int64_t)INT32_MIN - min_penalty;
}

// DBGAligner<>::DBGAligner (dbg_aligner.cpp:33-61) + what this build implements.  Returns MGX_OK or
// an error code with a message in *err.
inline int prepare_config(const mgx_config &in, uint64_t k, mgx_config *out, DevConfig *d, std::string *err, bool labeled = false) {
    *out = in;
    mgx_config &c = *out;
    if (!c.min_seed_length) c.min_seed_length = k;
    if (!c.max_seed_length) c.max_seed_length = k;
    uint64_t lo = std::min(c.min_seed_length, c.max_seed_length), hi = std::max(c.min_seed_length, c.max_seed_length);
    c.min_seed_length = lo;
    c.max_seed_length = hi;
    if (labeled) {
        // LabeledAligner's ctor on top of DBGAligner's (aligner_labeled.cpp:463-465; no coordinates: no chaining)
        c.min_seed_length = std::min<uint64_t>(k, c.min_seed_length);
        c.max_seed_length = std::min<uint64_t>(k, c.max_seed_length);
    }
    if (!check_config_scores(c)) { *err = "Error: sum of min_cell_score and lowest penalty too low."; return MGX_ERR_CONFIG; }
    if (c.chain_alignments) { *err = "seed chaining (chain_alignments) is not implemented"; return MGX_ERR_UNSUPPORTED; }
    if (c.post_chain_alignments && labeled) { *err = "post_chain_alignments with an annotation is not implemented"; return MGX_ERR_UNSUPPORTED; }
    if (!c.global_xdrop) { *err = "per-branch xdrop (labeled+coordinates mode) is not implemented"; return MGX_ERR_UNSUPPORTED; }
    if (c.no_backtrack) { *err = "no_backtrack is not implemented"; return MGX_ERR_UNSUPPORTED; }
    if (c.num_alternative_paths < 1 || c.num_alternative_paths > MGX_MAX_ALTERNATIVE_PATHS) {
        *err = "num_alternative_paths must be in 1.." + std::to_string(MGX_MAX_ALTERNATIVE_PATHS) + " on the device";
        return MGX_ERR_UNSUPPORTED;
    }
    if (c.xdrop <= 0) { *err = "xdrop must be positive"; return MGX_ERR_INVALID; }
    auto sat = [](uint64_t v) { return v >= INF_LEN ? INF_LEN : (uint32_t)v; };
    d->min_seed_length = sat(c.min_seed_length);
    d->max_seed_length = sat(c.max_seed_length);
    d->max_num_seeds_per_locus = sat(c.max_num_seeds_per_locus);
    d->min_cell_score = c.min_cell_score; d->min_path_score = c.min_path_score; d->xdrop = c.xdrop;
    d->min_exact_match = c.min_exact_match; d->max_nodes_per_seq_char = c.max_nodes_per_seq_char;
    d->max_ram_per_alignment = c.max_ram_per_alignment; d->rel_score_cutoff = c.rel_score_cutoff;
    d->gap_open = c.gap_opening_penalty; d->gap_ext = c.gap_extension_penalty;
    d->left_end_bonus = c.left_end_bonus; d->right_end_bonus = c.right_end_bonus;
    d->fwd_and_rc = c.forward_and_reverse_complement; d->allow_left_trim = c.allow_left_trim;
    d->seed_complexity_filter = c.seed_complexity_filter;
    d->num_alt = (uint32_t)c.num_alternative_paths;
    // post_chain_alignments: the aggregator keeps every alignment of a query (aligner_aggregator.hpp:88-96) for chain_host.hpp;
    // on the device "every" is post_chain_capacity() — a query with more gets MGX_ERR_CAPACITY, never a dropped alignment
    d->post_chain = c.post_chain_alignments ? 1u : 0u;
    d->agg_cap = c.post_chain_alignments ? post_chain_capacity(c.num_alternative_paths) : d->num_alt;
    d->canonical = 0;                       // set from the graph's mode by the caller (with fwd_and_rc, dbg_aligner.cpp:225-226)
    return MGX_OK;
}

inline uint32_t next_pow2(uint64_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

// label_scale: multiplies the label-aware aligner's own arenas (the capacity retry of mgx_align_batch doubles it per attempt)
inline int derive_limits(const mgx_config &cfg, const mgx_limits *u, uint32_t Lmax, DevLimits *lim, std::string *err, bool labeled = false,
                         uint32_t label_scale = 1) {
    DevLimits &l = *lim;
    if (u && u->max_query_length && Lmax > u->max_query_length) {
        *err = "a query of length " + std::to_string(Lmax) + " exceeds mgx_limits.max_query_length = " + std::to_string(u->max_query_length);
        return MGX_ERR_CAPACITY;
    }
    l.Lmax = std::max<uint32_t>(8, (Lmax + 7) & ~7u);
    // seed coordinates and per-strand seed counts are 16-bit on the device (DevSeed, SeedHdr): a longer query would wrap
    // them silently, so it is refused here (the reference has no such limit; documented in mgx.h)
    if (l.Lmax > MGX_MAX_QUERY_LENGTH) {
        *err = "a query of length " + std::to_string(Lmax) + " exceeds the device limit of " + std::to_string(MGX_MAX_QUERY_LENGTH) + " bp";
        return MGX_ERR_UNSUPPORTED;
    }
    uint32_t mc;
    if (u && u->max_columns) mc = u->max_columns;
    else if (cfg.max_nodes_per_seq_char < 1e6) mc = (uint32_t)(cfg.max_nodes_per_seq_char * l.Lmax) * 2 + 64;
    else mc = 16 * l.Lmax + 256;
    l.max_columns = std::min<uint32_t>((1u << 24) - 2, std::max<uint32_t>(64, mc));   // 24-bit table index in the queue key
    l.max_seeds = (u && u->max_seeds) ? u->max_seeds : 2 * l.Lmax + 64;
    if (l.max_seeds > 65535) { *err = "mgx_limits.max_seeds must not exceed 65535 (16-bit seed counts on the device)"; return MGX_ERR_UNSUPPORTED; }
    l.max_path = 2 * l.Lmax + 64;
    l.max_alt = std::max<uint32_t>(4096, l.max_seeds);
    uint64_t cw;
    if (u && u->cell_arena_bytes) cw = u->cell_arena_bytes / 4;
    else {
        // S / F records of the cell arena: every column of the general path, and the chain columns that stay behind in the
        // frontier (one 32-cell record).  Round 2 budgeted 3 x 96 words for every possible column; on short-read batches
        // 98 % of the columns are chain columns that own nothing here, and the arena slices are what limits the number of
        // resident groups at 10 M reads — a quarter of that, still 24 cells x 3 per possible column.  A read that runs out gets
        // MGX_ERR_CAPACITY and the adapter's retry doubles the budget.
        uint64_t band = cfg.xdrop < 1000 ? std::min<uint64_t>(l.Lmax + 8, 24) : (l.Lmax + 8);
        cw = (uint64_t)l.max_columns * 3 * band + 3 * (l.Lmax + 16) + 4096;
    }
    l.cell_words = (uint32_t)std::min<uint64_t>(cw, 0xFFFFFF00ull);
    l.hash_size = next_pow2(2ull * ((uint64_t)l.max_columns + l.max_path) + 2);
    // [0, 3 A): extension results, their reversals, backward results (A = num_alternative_paths); then the aggregator's queue
    l.n_aln = 3 * (uint32_t)cfg.num_alternative_paths
              + (cfg.post_chain_alignments ? post_chain_capacity(cfg.num_alternative_paths) : (uint32_t)cfg.num_alternative_paths);
    // Convergence vectors (one per visited node, holding only the query range its columns touched) come from a pool sized by
    // the cell budget: a column of w cells appends w words, a range that outgrows its allocation is re-appended.  Half the
    // cell words per extender never binds on the test and bench workloads; a read that runs out gets MGX_ERR_CAPACITY
    // and the adapter's retry (doubled cell_arena_bytes) doubles the pool with it.
    l.conv_pool_words = (uint32_t)std::min<uint64_t>(0xFFFFFF00ull, (uint64_t)l.cell_words / 2 + 16ull * l.Lmax + 1024);
    l.lab_words = l.lab_ext = l.lab_pool = l.lab_queues = 0;
    if (labeled) {
        // Label-aware alignment: a backtracking reports one alignment per label subset of its seed and the aggregator keeps a
        // queue per label, so the alignment buffers are [0, E) extension results, [E, 2E) their reversals, [2E, 3E) backward
        // results, [3E, 3E + pool) the aggregator's alignments.  A read that outgrows any of them gets MGX_ERR_CAPACITY.
        // The reference bounds none of this; here a read beyond a bound is re-run with all of them doubled (mgx_align_batch).
        const uint32_t ls = std::min<uint32_t>(std::max<uint32_t>(1, label_scale), 8);      // (lab_ext <= 64: a bit per reversal)
        l.lab_ext = 8 * ls;
        l.lab_pool = 32 * ls * (uint32_t)std::max<uint64_t>(1, cfg.num_alternative_paths);
        l.lab_queues = 64 * ls;                    // labels with alignments per read == labels on a read's seeds (filter_seeds)
        l.n_aln = 3 * l.lab_ext + l.lab_pool;
        // label sets: per-extension sets (one per fork and per flushed column that lost labels), per-read sets (seeds,
        // alignments), the seed filter's position bitmaps (one per label seen on the read's seeds)
        l.lab_words = (uint32_t)std::min<uint64_t>(1u << 24, (uint64_t)ls * 16384 + 8ull * l.max_columns + (uint64_t)l.lab_queues * ((l.Lmax + 31) / 32 + 2));
        // every column goes through the general path (pool entries, no aliases) and every backward alignment writes its
        // filter_nodes marks (a vector of the aligned query range per path node): room for a few of them on short reads
        l.conv_pool_words = (uint32_t)std::min<uint64_t>(0xFFFFFF00ull, (uint64_t)l.conv_pool_words
                                                          + std::min<uint64_t>(4ull * (l.Lmax + 8) * (l.Lmax + 8), 1ull << 22));
    }
    return MGX_OK;
}

// host copy of one batch's results in the layout of mgx_results
struct HostResults {
    std::vector<uint64_t> aln_begin, nodes;
    std::vector<mgx_alignment> alns;
    std::vector<mgx_cigar_op> cigar;
    std::vector<char> seqs;
    std::vector<int32_t> status;
    std::vector<uint32_t> labels;      // label columns of all alignments (label-aware alignment only)

    // words one alignment occupies in the stream
    static uint64_t aln_words(uint32_t n_nodes, uint32_t n_cigar, uint32_t seq_len) { return (uint64_t)n_nodes + n_cigar + ((uint64_t)seq_len + 3) / 4; }

    // Stream layout of a read with n_alignments >= 1, from stream_off: alignment 0 = nodes, CIGAR runs (len << 3 | op),
    // path characters (its scalars are in the header); every further alignment = 6 words (score, offset, n_nodes, n_cigar,
    // seq_len, orientation) followed by the same three arrays.  `stream_words`: words available (bounds for untrusted input).
    // Label-aware runs (`labeled`): every alignment's arrays are followed by its label count and its labels.
    bool decode(const ReadResult *rr, uint64_t n, const uint32_t *stream, uint64_t stream_words = ~0ull, bool labeled = false) {
        aln_begin.assign(1, 0);
        alns.clear(); nodes.clear(); cigar.clear(); seqs.clear(); status.clear(); labels.clear();
        for (uint64_t i = 0; i < n; ++i) {
            const ReadResult &r = rr[i];
            status.push_back(r.status);
            if (r.status == ST_OK && r.n_alignments > 0) {
                uint64_t at = r.stream_off;
                for (int32_t a = 0; a < r.n_alignments; ++a) {
                    int32_t score = r.score;
                    uint32_t offset = r.offset, n_nodes = r.n_nodes, n_cigar = r.n_cigar, seq_len = r.seq_len, orientation = r.orientation;
                    if (a) {
                        if (at > stream_words || stream_words - at < 6) return false;
                        const uint32_t *h = stream + at;
                        score = (int32_t)h[0]; offset = h[1]; n_nodes = h[2]; n_cigar = h[3]; seq_len = h[4]; orientation = h[5];
                        at += 6;
                    }
                    const uint64_t words = aln_words(n_nodes, n_cigar, seq_len);
                    if (at > stream_words || words > stream_words - at) return false;
                    const uint32_t *p = stream + at;
                    at += words;
                    mgx_alignment m;
                    memset(&m, 0, sizeof(m));
                    m.score = score; m.offset = offset; m.n_nodes = n_nodes; m.n_cigar = n_cigar; m.seq_len = seq_len;
                    m.orientation = (uint8_t)orientation;
                    m.nodes_begin = nodes.size(); m.cigar_begin = cigar.size(); m.seq_begin = seqs.size();
                    for (uint32_t x = 0; x < n_nodes; ++x) nodes.push_back(p[x]);
                    uint32_t nm = 0;
                    for (uint32_t x = 0; x < n_cigar; ++x) {
                        mgx_cigar_op op;
                        memset(&op, 0, sizeof(op));
                        op.len = p[n_nodes + x] >> 3;
                        op.op = (uint8_t)(p[n_nodes + x] & 7);
                        if (op.op > MGX_OP_NODE_INSERTION) return false;      // 6, 7: not an operator (mgx_format_tsv indexes by it)
                        if (op.op == MGX_OP_MATCH) nm += op.len;
                        cigar.push_back(op);
                    }
                    m.num_matches = nm;
                    if (n_cigar) {
                        const mgx_cigar_op &f = cigar[m.cigar_begin], &b = cigar.back();
                        m.clipping = f.op == MGX_OP_CLIPPED ? f.len : 0;
                        m.end_clipping = b.op == MGX_OP_CLIPPED ? b.len : 0;
                    }
                    const char *sq = reinterpret_cast<const char *>(p + n_nodes + n_cigar);
                    seqs.insert(seqs.end(), sq, sq + seq_len);
                    if (labeled) {
                        if (at >= stream_words) return false;
                        const uint32_t nl = stream[at];
                        if (nl > stream_words - at - 1) return false;
                        m.n_labels = nl; m.labels_begin = labels.size();
                        labels.insert(labels.end(), stream + at + 1, stream + at + 1 + nl);
                        at += 1 + (uint64_t)nl;
                    }
                    alns.push_back(m);
                }
            }
            aln_begin.push_back(alns.size());
        }
        return true;
    }

    // append query q of `src` (its status and alignments) as the next query of *this
    void append_query(const mgx_results &src, uint64_t q) {
        if (aln_begin.empty()) aln_begin.push_back(0);
        status.push_back(src.status ? src.status[q] : 0);
        for (uint64_t ai = src.aln_begin[q]; ai < src.aln_begin[q + 1]; ++ai) {
            mgx_alignment m = src.alignments[ai];
            const uint64_t nb = nodes.size(), cb = cigar.size(), sb = seqs.size(), lb = labels.size();
            nodes.insert(nodes.end(), src.nodes + m.nodes_begin, src.nodes + m.nodes_begin + m.n_nodes);
            cigar.insert(cigar.end(), src.cigar + m.cigar_begin, src.cigar + m.cigar_begin + m.n_cigar);
            seqs.insert(seqs.end(), src.seqs + m.seq_begin, src.seqs + m.seq_begin + m.seq_len);
            if (src.labels && m.n_labels) labels.insert(labels.end(), src.labels + m.labels_begin, src.labels + m.labels_begin + m.n_labels);
            m.nodes_begin = nb; m.cigar_begin = cb; m.seq_begin = sb; m.labels_begin = lb;
            alns.push_back(m);
        }
        aln_begin.push_back(alns.size());
    }

    void view(mgx_results *out) const {
        out->n_queries = status.size();
        out->aln_begin = aln_begin.data();
        out->alignments = alns.data();
        out->nodes = nodes.data();
        out->cigar = cigar.data();
        out->seqs = seqs.data();
        out->status = status.data();
        out->labels = labels.empty() ? nullptr : labels.data();
    }
};

} // namespace mgx
