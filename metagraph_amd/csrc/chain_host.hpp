// chain_host.hpp — post-alignment chaining (DBGAlignerConfig::post_chain_alignments) on the host side of libmgx.
//
// The reference runs chain_alignments<LocalAlignmentLess> (A/aligner_chainer.cpp:555-720) on the alignments a query's
// aggregator hands back (dbg_aligner.cpp:328-332): alignments that leave query characters uncovered are joined into chains —
// the overlap of two neighbours trimmed off the second one, missing graph nodes filled with dummy nodes (0), a gap bridged
// with a '$' — and a second aggregator keeps the best num_alternative_paths of the complete alignments and the chains.  It
// works on the handful of alignments a query has left, so it stays host code here: the device keeps EVERY alignment of a
// query in this mode (aligner_aggregator.hpp:88-96; DevConfig::post_chain), mgx_fetch_results / mgx_align_batch decode them
// and run this file over every query with at least two, and mgx_chain_alignments does the same for results that were decoded
// elsewhere (mgx_results_from_raw after a gather).  Host code, no GPU needed; no label coordinates (the reference refuses
// them here, aligner_chainer.cpp:563-566).
//
// Representation: an alignment is a ChainItem — the window [qb, qe) of its query strand, the run-length CIGAR including
// its clipping runs (a chain has clipping runs INSIDE: the characters between two joined alignments), nodes, spelling, score.
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "host_common.hpp"

namespace mgx {

struct ChainItem {
    int32_t score = 0;
    uint32_t offset = 0;
    uint8_t orientation = 0;
    uint32_t qb = 0, qe = 0;                         // query_view = strand[qb, qe)
    std::vector<uint64_t> nodes;
    std::vector<mgx_cigar_op> cig;
    std::string seq;
    std::vector<uint32_t> labels;

    bool empty() const { return nodes.empty(); }
    uint32_t clip() const { return !cig.empty() && cig.front().op == MGX_OP_CLIPPED ? cig.front().len : 0; }
    uint32_t end_clip() const { return !cig.empty() && cig.back().op == MGX_OP_CLIPPED ? cig.back().len : 0; }
    void clear() { *this = ChainItem(); }
};

inline mgx_cigar_op chain_op(uint8_t op, uint32_t len) { mgx_cigar_op o; memset(&o, 0, sizeof(o)); o.op = op; o.len = len; return o; }
inline void chain_drop_clip(ChainItem &a) { if (a.clip()) a.cig.erase(a.cig.begin()); }             // Alignment::trim_clipping
inline void chain_drop_end_clip(ChainItem &a) { if (a.end_clip()) a.cig.pop_back(); }               // Alignment::trim_end_clipping
// Alignment::extend_query_begin (alignment.hpp:209-214): the clipping run grows until the alignment's query starts at `begin`
inline void chain_clip_back_to(ChainItem &a, uint32_t begin) {
    const uint32_t full_begin = a.qb - a.clip();
    if (full_begin > begin) {
        if (a.clip()) a.cig.front().len += full_begin - begin;
        else a.cig.insert(a.cig.begin(), chain_op(MGX_OP_CLIPPED, full_begin - begin));
    }
}
// Alignment::trim_offset (alignment.cpp:177-190)
inline void chain_trim_offset(ChainItem &a) {
    if (!a.offset || a.nodes.size() <= 1) return;
    const size_t first_dummy = (size_t)(std::find(a.nodes.begin(), a.nodes.end(), (uint64_t)0) - a.nodes.begin()) - 1;
    const size_t trim = std::min(std::min<size_t>(a.offset, a.nodes.size() - 1), first_dummy);
    a.offset -= (uint32_t)trim;
    a.nodes.erase(a.nodes.begin(), a.nodes.begin() + trim);
}

// Alignment::trim_query_prefix (alignment.cpp:192-278): take the first n query characters (and the deletions right behind
// them) off the alignment; returns how far into its current run the CIGAR was cut.  `strand` = the query strand of a.
inline size_t chain_trim_query_prefix(ChainItem &a, size_t n, size_t node_overlap, const mgx_config &cfg, const char *strand) {
    const uint32_t clipping = a.clip();
    const uint32_t query_begin = a.qb - clipping;
    size_t run = clipping ? 1 : 0, in_run = 0, seq_at = 0, node_at = 0;
    auto consume_ref = [&]() {
        ++seq_at;
        if (a.offset < node_overlap) ++a.offset;
        else if (node_at + 1 < a.nodes.size()) ++node_at;
        else a.clear();
    };
    while (n || (run < a.cig.size() && a.cig[run].op == MGX_OP_DELETION)) {
        if (run >= a.cig.size()) { a.clear(); return 0; }
        const mgx_cigar_op cur = a.cig[run];
        const int32_t gap = cur.len - in_run == 1 ? cfg.gap_opening_penalty : cfg.gap_extension_penalty;
        if (cur.op == MGX_OP_MATCH || cur.op == MGX_OP_MISMATCH) {
            a.score -= cfg.score_matrix[(uint8_t)strand[a.qb] & 127][(uint8_t)a.seq[seq_at] & 127];
            ++a.qb; --n;
            consume_ref();
            if (a.empty()) return 0;
        } else if (cur.op == MGX_OP_INSERTION) {
            a.score -= gap;
            ++a.qb; --n;
        } else if (cur.op == MGX_OP_DELETION) {
            a.score -= gap;
            consume_ref();
            if (a.empty()) return 0;
        } else {
            a.clear();                      // a clipping run or dummy nodes inside: "trimming chains not supported" (:248-251)
            return 0;
        }
        if (++in_run == cur.len) { ++run; in_run = 0; }
    }
    if (!clipping && run) a.score -= cfg.left_end_bonus;
    a.nodes.erase(a.nodes.begin(), a.nodes.begin() + node_at);
    a.seq.erase(0, seq_at);
    if (run < a.cig.size()) a.cig[run].len -= (uint32_t)in_run;
    a.cig.erase(a.cig.begin(), a.cig.begin() + run);
    chain_clip_back_to(a, query_begin);
    return in_run;
}

// Alignment::insert_gap_prefix (alignment.cpp:1154-1234): make `a` appendable behind a chain that ends `gap` query characters
// before it (gap >= 0: a '$' and, for a short gap, dummy nodes) or that shares -gap matched characters with it (gap < 0)
inline void chain_insert_gap_prefix(ChainItem &a, int64_t gap, size_t node_overlap, const mgx_config &cfg) {
    size_t extra_nodes = node_overlap + 1;
    if (gap < 0) {
        chain_drop_clip(a);
        extra_nodes += gap - 1;
        if (a.offset) a.nodes.erase(a.nodes.begin(), a.nodes.begin() + (a.offset + gap));
        if (extra_nodes) {
            a.score += cfg.gap_opening_penalty + (int32_t)(extra_nodes - 1) * cfg.gap_extension_penalty;
            a.cig.insert(a.cig.begin(), chain_op(MGX_OP_NODE_INSERTION, (uint32_t)extra_nodes));
        }
    } else {
        chain_drop_clip(a);
        a.seq.insert(a.seq.begin(), '$');
        a.cig.insert(a.cig.begin(), chain_op(MGX_OP_DELETION, 1));
        a.score += cfg.gap_opening_penalty;
        if ((size_t)gap <= node_overlap) {
            chain_trim_offset(a);
            a.score += cfg.gap_opening_penalty + (int32_t)(extra_nodes - 2) * cfg.gap_extension_penalty;
            a.cig.insert(a.cig.begin(), chain_op(MGX_OP_NODE_INSERTION, (uint32_t)(extra_nodes - 1)));
        }
        chain_clip_back_to(a, a.qb - (uint32_t)gap);
    }
    a.nodes.insert(a.nodes.begin(), extra_nodes, (uint64_t)0);
    a.offset = (uint32_t)node_overlap;
}

// Alignment::append without coordinates (alignment.cpp:94-175): labels are intersected; true = the label set shrank
inline bool chain_append(ChainItem &a, ChainItem &&b) {
    bool changed = false;
    if (!a.labels.empty() && b.labels.empty()) a.labels.clear();
    if (!a.labels.empty()) {
        std::vector<uint32_t> both;
        std::set_intersection(a.labels.begin(), a.labels.end(), b.labels.begin(), b.labels.end(), std::back_inserter(both));
        if (both.empty()) { a.clear(); return true; }
        changed = both.size() < a.labels.size();
        a.labels.swap(both);
    }
    a.nodes.insert(a.nodes.end(), b.nodes.begin(), b.nodes.end());
    a.seq += b.seq;
    a.score += b.score;
    if (!b.cig.empty()) {                                   // Cigar::append: the first run of b merges with a's last (:110-116)
        size_t from = 0;
        if (!a.cig.empty() && a.cig.back().op == b.cig.front().op) { a.cig.back().len += b.cig.front().len; from = 1; }
        a.cig.insert(a.cig.end(), b.cig.begin() + from, b.cig.end());
    }
    a.qe = b.qe;
    return changed;
}

// LocalAlignmentLess (alignment.hpp:337-348) and Alignment::operator== (:261-269)
inline bool chain_less(const ChainItem &a, const ChainItem &b) {
    return std::make_tuple(b.score, a.qe - a.qb, a.orientation, a.clip()) > std::make_tuple(a.score, b.qe - b.qb, b.orientation, b.clip());
}
inline bool chain_same(const ChainItem &a, const ChainItem &b) {
    if (a.orientation != b.orientation || a.offset != b.offset || a.score != b.score || a.qb != b.qb || a.qe != b.qe
            || a.seq != b.seq || a.nodes != b.nodes || a.cig.size() != b.cig.size()) return false;
    for (size_t i = 0; i < a.cig.size(); ++i) if (a.cig[i].op != b.cig[i].op || a.cig[i].len != b.cig[i].len) return false;
    return true;
}

// AlignmentAggregator without labels, post_chain_alignments off (aligner_aggregator.hpp:68-202): the best N by LocalAlignmentLess
struct ChainTop {
    size_t cap;
    double rel_score_cutoff;
    std::vector<ChainItem> q;
    int32_t cutoff() const {
        if (q.empty()) return INT32_MIN + 100;
        size_t mx = 0;
        for (size_t t = 1; t < q.size(); ++t) if (chain_less(q[mx], q[t])) mx = t;
        return q[mx].score > 0 ? (int32_t)(q[mx].score * rel_score_cutoff) : q[mx].score;
    }
    void add(ChainItem &&a) {
        if (q.empty()) { q.push_back(std::move(a)); return; }
        if (a.score < cutoff()) return;
        for (const ChainItem &x : q) if (chain_same(a, x)) return;
        if (q.size() < cap) { q.push_back(std::move(a)); return; }
        size_t mn = 0;
        for (size_t t = 1; t < q.size(); ++t) if (chain_less(q[t], q[mn])) mn = t;
        if (chain_less(a, q[mn])) return;
        q[mn] = std::move(a);
    }
    std::vector<ChainItem> take() {                     // get_alignments: sorted, best first (:180-202)
        std::stable_sort(q.begin(), q.end(), chain_less);
        std::reverse(q.begin(), q.end());
        return std::move(q);
    }
};

// construct_alignment_chain (aligner_chainer.cpp:623-720): every way of continuing `chain` with the alignments [from, v.size())
inline void chain_continue(const std::vector<ChainItem> &v, size_t from, ChainItem &&chain, const char *strand, uint32_t L,
                           size_t node_overlap, const mgx_config &cfg, std::vector<int32_t> &best_score, ChainTop &top) {
    if (from == v.size() || chain.qe == L) { top.add(std::move(chain)); return; }
    bool continued = false;
    for (size_t x = from; x < v.size(); ++x) {
        const ChainItem &cand = v[x];
        if (cand.offset) continue;
        if (cand.qb <= chain.qb || cand.qe == chain.qe) continue;
        if (!chain.labels.empty()) {
            std::vector<uint32_t> both;
            std::set_intersection(cand.labels.begin(), cand.labels.end(), chain.labels.begin(), chain.labels.end(), std::back_inserter(both));
            if (both.empty()) continue;
        }
        ChainItem next = cand;
        if (cand.qb >= chain.qe) {
            chain_insert_gap_prefix(next, (int64_t)cand.qb - (int64_t)chain.qe, node_overlap, cfg);
        } else {
            // overlap: trim it off the front of the incoming alignment, then fill in dummy nodes
            const size_t last_run = chain.cig.size() >= 2 ? chain.cig[chain.cig.size() - 2].len : 0;
            const size_t overlap = std::min(last_run, chain_trim_query_prefix(next, chain.qe - cand.qb, node_overlap, cfg, strand));
            if (next.empty() || next.seq.size() <= node_overlap) continue;
            if (next.cig[next.clip() ? 1 : 0].op != MGX_OP_MATCH) continue;
            if (overlap < node_overlap) chain_insert_gap_prefix(next, -(int64_t)overlap, node_overlap, cfg);
            else chain_drop_clip(next);
        }
        const int32_t next_score = chain.score + next.score;
        if (next_score <= best_score[cand.qe]) continue;
        best_score[cand.qe] = next_score;
        ChainItem longer = chain;
        chain_drop_end_clip(longer);
        const bool changed = chain_append(longer, std::move(next));
        if (!longer.empty()) {
            chain_continue(v, x + 1, std::move(longer), strand, L, node_overlap, cfg, best_score, top);
            continued |= changed;
        }
    }
    if (!continued) top.add(std::move(chain));
}

// chain_alignments (aligner_chainer.cpp:555-620) for one query; `fwd` / `rc` = its two strands as the aligner saw them
inline std::vector<ChainItem> chain_query(std::vector<ChainItem> &&alns, const std::string &fwd, const std::string &rc,
                                          const mgx_config &cfg, size_t node_overlap) {
    if (alns.size() < 2 || !cfg.post_chain_alignments) return std::move(alns);
    ChainTop top{ (size_t)std::max<uint64_t>(1, cfg.num_alternative_paths), cfg.rel_score_cutoff, {} };
    std::vector<ChainItem> open;                        // alignments that leave an end of the query uncovered
    for (ChainItem &a : alns) {
        if (!a.clip() && !a.end_clip()) top.add(std::move(a));
        else open.push_back(std::move(a));
    }
    std::sort(open.begin(), open.end(), [](const ChainItem &a, const ChainItem &b) {
        return std::make_tuple(a.orientation, a.qe, a.clip(), b.score, a.seq.size())
             < std::make_tuple(b.orientation, b.qe, b.clip(), a.score, b.seq.size());
    });
    const size_t split = (size_t)(std::find_if(open.begin(), open.end(), [](const ChainItem &a) { return a.orientation != 0; }) - open.begin());
    for (int strand = 0; strand < 2; ++strand) {
        const std::string &q = strand ? rc : fwd;
        std::vector<ChainItem> part(open.begin() + (strand ? split : 0), open.begin() + (strand ? open.size() : split));
        std::vector<int32_t> best_score(q.size() + 1, 0);
        for (size_t x = 0; x < part.size(); ++x) {
            if (part[x].score > best_score[part[x].qe]) {
                best_score[part[x].qe] = part[x].score;
                chain_continue(part, x + 1, ChainItem(part[x]), q.data(), (uint32_t)q.size(), node_overlap, cfg, best_score, top);
            }
        }
    }
    return top.take();
}

// the two strands of a query as AlignmentResults holds them (alignment.cpp:1348-1372)
inline void chain_strands(const char *raw, size_t len, std::string *fwd, std::string *rc) {
    fwd->resize(len); rc->resize(len);
    for (size_t i = 0; i < len; ++i) {
        const int8_t c = (int8_t)raw[i];
        (*fwd)[i] = c >= 0 ? (char)toupper(c) : (char)127;
    }
    static const char up[] = "TVGHEFCDIJMLKNOPQYSAABWXRZ";       // COMPL_TAB (common/seq_tools/reverse_complement.hpp:31-48), upper case
    for (size_t i = 0; i < len; ++i) {
        const unsigned char f = (unsigned char)(*fwd)[len - 1 - i];
        (*rc)[i] = f >= 'A' && f <= 'Z' ? up[f - 'A'] : f == 96 ? (char)64 : (char)f;
    }
}

// Post-chain every query of `in` (whose reads are seqs[offsets[q] .. offsets[q + 1])) into `out`.
// seq_origin: the batch offset of seqs[0] (the caller may hold only the span of the batch whose queries have two or more
// alignments: the reads of the others are never looked at, their lengths come from `offsets`)
inline void chain_results(const mgx_results &in, const char *seqs, const uint64_t *offsets, const mgx_config &cfg, uint32_t k,
                          HostResults *out, uint64_t seq_origin = 0) {
    out->aln_begin.assign(1, 0);
    out->alns.clear(); out->nodes.clear(); out->cigar.clear(); out->seqs.clear(); out->status.clear(); out->labels.clear();
    std::string fwd, rc;
    for (uint64_t q = 0; q < in.n_queries; ++q) {
        out->status.push_back(in.status ? in.status[q] : 0);
        const uint64_t lo = in.aln_begin[q], hi = in.aln_begin[q + 1];
        std::vector<ChainItem> items;
        const uint32_t L = (uint32_t)(offsets[q + 1] - offsets[q]);
        for (uint64_t ai = lo; ai < hi; ++ai) {
            const mgx_alignment &m = in.alignments[ai];
            ChainItem c;
            c.score = m.score; c.offset = m.offset; c.orientation = m.orientation;
            c.qb = m.clipping; c.qe = L - m.end_clipping;
            c.nodes.assign(in.nodes + m.nodes_begin, in.nodes + m.nodes_begin + m.n_nodes);
            c.cig.assign(in.cigar + m.cigar_begin, in.cigar + m.cigar_begin + m.n_cigar);
            c.seq.assign(in.seqs + m.seq_begin, m.seq_len);
            if (in.labels && m.n_labels) c.labels.assign(in.labels + m.labels_begin, in.labels + m.labels_begin + m.n_labels);
            items.push_back(std::move(c));
        }
        if (items.size() >= 2) {
            chain_strands(seqs + (offsets[q] - seq_origin), L, &fwd, &rc);
            items = chain_query(std::move(items), fwd, rc, cfg, k - 1);
        }
        for (const ChainItem &c : items) {
            mgx_alignment m;
            memset(&m, 0, sizeof(m));
            m.score = c.score; m.offset = c.offset; m.orientation = c.orientation;
            m.n_nodes = (uint32_t)c.nodes.size(); m.n_cigar = (uint32_t)c.cig.size(); m.seq_len = (uint32_t)c.seq.size();
            m.nodes_begin = out->nodes.size(); m.cigar_begin = out->cigar.size(); m.seq_begin = out->seqs.size();
            m.clipping = c.clip(); m.end_clipping = c.end_clip();
            for (const mgx_cigar_op &o : c.cig) if (o.op == MGX_OP_MATCH) m.num_matches += o.len;
            m.n_labels = (uint32_t)c.labels.size(); m.labels_begin = out->labels.size();
            out->nodes.insert(out->nodes.end(), c.nodes.begin(), c.nodes.end());
            out->cigar.insert(out->cigar.end(), c.cig.begin(), c.cig.end());
            out->seqs.insert(out->seqs.end(), c.seq.begin(), c.seq.end());
            out->labels.insert(out->labels.end(), c.labels.begin(), c.labels.end());
            out->alns.push_back(m);
        }
        out->aln_begin.push_back(out->alns.size());
    }
}

} // namespace mgx
