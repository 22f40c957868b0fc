// hip_dbg_aligner_adapter.hpp — the reference-side adapter: an mtg::graph::align::IDBGAligner whose align_batch runs on
// an MI355X through libmgx.so (include/mgx.h).
//
// THIS FILE IS MEANT TO BE DROPPED INTO THE REFERENCE TREE (metagraph/src/graph/alignment/) and compiled there: it
// includes the reference's own headers, which are not part of this repository (and the reference cannot be built in
// this image: its third-party submodules are absent), so nothing here compiles it.  It contains no reference code —
// only calls into the reference's public interfaces:
//     IDBGAligner / Query / AlignmentCallback / AlignmentResults      graph/alignment/dbg_aligner.hpp:20-39
//     Alignment(query_view, nodes, sequence, score, cigar, clipping, orientation, offset)   alignment.hpp:179-192
//     Cigar(op, count) / append                                      aligner_cigar.hpp:28-60
//     DBGSuccinct::get_boss(), BOSS::get_W / get_last / get_F         dbg_succinct.hpp, boss.hpp:120-180
//     AnnotatedDBG::Annotator::get_matrix, BinaryMatrix::get_column / num_rows / num_columns
//                                                                    annotation.hpp:59-63, binary_matrix.hpp:29-40
//     Alignment::label_columns                                       alignment.hpp:285
// The switch in cli/align.cpp:452-459 becomes
//     aligner = config->align_device >= 0
//         ? std::unique_ptr<IDBGAligner>(new HipDBGAlignerAdapter(*aln_graph, aligner_config, hip_graph))
//         : std::make_unique<DBGAligner<>>(*aln_graph, aligner_config);
// with one `HipGraphHandle hip_graph(dbg_succinct, device)` created next to the graph load (align.cpp:337-339).
#pragma once

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "graph/alignment/dbg_aligner.hpp"                 // reference
#include "graph/representation/succinct/dbg_succinct.hpp"  // reference
#include "graph/annotated_dbg.hpp"                         // reference (AnnotatedDBG::Annotator)
#include "annotation/int_matrix/base/int_matrix.hpp"       // reference (MultiIntMatrix: coordinates)
#include "mgx.h"                                           // this repository: include/mgx.h

namespace mtg {
namespace graph {
namespace align {

// BOSS table of a loaded DBGSuccinct on the device (once per process; shared by every aligner / thread)
class HipGraphHandle {
  public:
    HipGraphHandle(const DBGSuccinct &dbg, int device = 0) {
        const boss::BOSS &boss = dbg.get_boss();
        const uint64_t n = boss.num_edges();
        // one pass over the wavelet tree / bit vector (boss.hpp: get_W(i), get_last(i)); slot 0 is unused
        std::vector<uint8_t> W(n + 1, 0), last(n + 1, 0), valid;
        for (uint64_t i = 1; i <= n; ++i) { W[i] = boss.get_W(i); last[i] = boss.get_last(i); }
        std::vector<uint64_t> F(boss.alph_size);
        for (size_t c = 0; c < boss.alph_size; ++c) F[c] = boss.get_F(c);
        if (const bit_vector *mask = dbg.get_mask()) {       // dbg_succinct.cpp:934-936; nullptr after reset_mask()
            valid.assign(n + 1, 0);
            for (uint64_t i = 1; i <= n; ++i) valid[i] = (*mask)[i];
        }
        mgx_boss_view v{};
        v.k = dbg.get_k(); v.sigma = boss.alph_size; v.n_edges = n;
        v.W = W.data(); v.last = last.data(); v.F = F.data(); v.valid = valid.empty() ? nullptr : valid.data();
        // PRIMARY: `dbg` is the DBGSuccinct under the CanonicalDBG wrapper of cli/align.cpp:383-387; libmgx applies the wrapper
        // on the device and reports CanonicalDBG's node ids (id + dbg.max_index() = the reverse complement), so the
        // Alignments rebuilt below are valid on the wrapper the caller holds (k <= 64)
        v.mode = dbg.get_mode() == DeBruijnGraph::BASIC ? MGX_MODE_BASIC
               : dbg.get_mode() == DeBruijnGraph::CANONICAL ? MGX_MODE_CANONICAL : MGX_MODE_PRIMARY;
        if (int rc = mgx_graph_create(&v, device, &g_))
            throw std::runtime_error(std::string("mgx_graph_create: ") + mgx_last_error() + " (" + std::to_string(rc) + ")");
    }
    ~HipGraphHandle() { mgx_graph_destroy(g_); }
    HipGraphHandle(const HipGraphHandle &) = delete;
    mgx_graph *handle() const { return g_; }
  private:
    mgx_graph *g_ = nullptr;
};

// DBGAlignerConfig -> mgx_config: same members in the same order (include/mgx.h); spelled out so that a change on either
// side fails to compile instead of silently shifting bytes
inline mgx_config to_mgx_config(const DBGAlignerConfig &c) {
    mgx_config m{};
    m.num_alternative_paths = c.num_alternative_paths;
    m.min_seed_length = c.min_seed_length;
    m.max_seed_length = c.max_seed_length;
    m.max_num_seeds_per_locus = c.max_num_seeds_per_locus;
    m.min_cell_score = c.min_cell_score;
    m.min_path_score = c.min_path_score;
    m.xdrop = c.xdrop;
    m.min_exact_match = c.min_exact_match;
    m.max_nodes_per_seq_char = c.max_nodes_per_seq_char;
    m.max_ram_per_alignment = c.max_ram_per_alignment;
    m.rel_score_cutoff = c.rel_score_cutoff;
    m.gap_opening_penalty = c.gap_opening_penalty;
    m.gap_extension_penalty = c.gap_extension_penalty;
    m.left_end_bonus = c.left_end_bonus;
    m.right_end_bonus = c.right_end_bonus;
    m.forward_and_reverse_complement = c.forward_and_reverse_complement;
    m.chain_alignments = c.chain_alignments;
    m.post_chain_alignments = c.post_chain_alignments;
    m.global_xdrop = c.global_xdrop;
    m.allow_left_trim = c.allow_left_trim;
    m.no_backtrack = c.no_backtrack;
    m.seed_complexity_filter = c.seed_complexity_filter;
    m.alignment_edit_distance = c.alignment_edit_distance;
    m.alignment_match_score = c.alignment_match_score;
    m.alignment_mm_transition_score = c.alignment_mm_transition_score;
    m.alignment_mm_transversion_score = c.alignment_mm_transversion_score;
    static_assert(sizeof(m.score_matrix) == sizeof(c.score_matrix), "score matrix layout");
    for (size_t i = 0; i < 128; ++i)
        for (size_t j = 0; j < 128; ++j) m.score_matrix[i][j] = c.score_matrix[i][j];
    return m;
}

// The columns of a loaded annotation on the device (once per process; shared by every labeled aligner / thread): what
// LabeledAligner's AnnotationBuffer fetches row by row (annotation_buffer.cpp:182) is answered in HBM from a row-major copy.
// Coordinates (annot::matrix::MultiIntMatrix, annotation_buffer.cpp:26-33) have no device form: an annotator that carries them
// is refused here, and the caller keeps LabeledAligner<> — which then chains seeds (aligner_labeled.cpp:457-462).
class HipAnnotationHandle {
  public:
    HipAnnotationHandle(const AnnotatedDBG::Annotator &annotator, int device = 0) : annotator_(annotator) {
        const annot::matrix::BinaryMatrix &matrix = annotator.get_matrix();
        if (dynamic_cast<const annot::matrix::MultiIntMatrix *>(&matrix))
            throw std::runtime_error("HipAnnotationHandle: annotations with coordinates stay on the reference's LabeledAligner");
        // the set rows of every column (BinaryMatrix::get_column, binary_matrix.hpp:40); row = graph_to_anno_index(node) = node - 1
        std::vector<uint64_t> col_begin{ 0 }, rows;
        for (uint64_t j = 0; j < matrix.num_columns(); ++j) {
            const std::vector<annot::matrix::BinaryMatrix::Row> col = matrix.get_column(j);
            rows.insert(rows.end(), col.begin(), col.end());
            col_begin.push_back(rows.size());
        }
        if (int rc = mgx_annotation_create_sparse(matrix.num_rows(), (uint32_t)matrix.num_columns(), col_begin.data(), rows.data(),
                                                  /*on_device=*/0, device, &a_))
            throw std::runtime_error(std::string("mgx_annotation_create_sparse: ") + mgx_last_error() + " (" + std::to_string(rc) + ")");
    }
    ~HipAnnotationHandle() { mgx_annotation_destroy(a_); }
    HipAnnotationHandle(const HipAnnotationHandle &) = delete;
    mgx_annotation *handle() const { return a_; }
    const AnnotatedDBG::Annotator &annotator() const { return annotator_; }
  private:
    const AnnotatedDBG::Annotator &annotator_;
    mgx_annotation *a_ = nullptr;
};

// DBGAligner<>(graph, config) resp. LabeledAligner<>(graph, config, annotator) behind IDBGAligner, on the device
class HipDBGAlignerAdapter : public IDBGAligner {
  public:
    // throws std::runtime_error on a bad score configuration, like DBGAligner<> (dbg_aligner.cpp:55-56), and on configs
    // outside the device path (seed chaining, per-branch x-drop, ...): the caller then falls back to DBGAligner<>
    HipDBGAlignerAdapter(const DeBruijnGraph &graph, const DBGAlignerConfig &config, const HipGraphHandle &hip_graph,
                         const HipAnnotationHandle *hip_annotation = nullptr)
          : graph_(graph), config_(config), hip_graph_(hip_graph), hip_annotation_(hip_annotation) {
        const mgx_config m = to_mgx_config(config);
        if (int rc = hip_annotation ? mgx_labeled_aligner_create(hip_graph.handle(), &m, nullptr, hip_annotation->handle(), &a_)
                                    : mgx_aligner_create(hip_graph.handle(), &m, nullptr, &a_))
            throw std::runtime_error(std::string(mgx_last_error()) + " (" + std::to_string(rc) + ")");
        // one aligner per thread-pool task (cli/align.cpp:440-475): each on a stream of its own, so that the tasks of one device
        // overlap instead of taking turns on the default stream; with `-p N` workers add
        // mgx_aligner_set_pipeline(a_, "device_share=N") so that every handle sizes its arenas for its share of the device
        mgx_aligner_create_stream(a_);
        mgx_config clamped;
        mgx_aligner_get_config(a_, &clamped);               // the ctors' seed-length clamps (dbg_aligner.cpp:37-53, aligner_labeled.cpp:451-456)
        config_.min_seed_length = clamped.min_seed_length;
        config_.max_seed_length = clamped.max_seed_length;
    }
    ~HipDBGAlignerAdapter() override { mgx_aligner_destroy(a_); }

    const DeBruijnGraph &get_graph() const override { return graph_; }
    const DBGAlignerConfig &get_config() const override { return config_; }
    bool has_coordinates() const override { return false; }     // (an annotation with coordinates never gets here: HipAnnotationHandle)

    void align_batch(const std::vector<Query> &seq_batch, const AlignmentCallback &callback) const override {
        std::vector<AlignmentResults> results;
        results.reserve(seq_batch.size());
        for (const auto &q : seq_batch) results.emplace_back(q.second);          // normalises the query (alignment.cpp:1348-1372)
        // ONE call: mgx_align_batch re-aligns the queries whose per-read device arenas overflowed by itself (doubled limits, six
        // doublings at most); a status that is left is an error of this call — IDBGAligner has no way to report it per query
        std::string blob;
        std::vector<uint64_t> offsets(seq_batch.size() + 1, 0);
        for (size_t t = 0; t < seq_batch.size(); ++t) { blob += seq_batch[t].second; offsets[t + 1] = blob.size(); }
        mgx_results res{};
        if (int rc = mgx_align_batch(a_, blob.data(), offsets.data(), seq_batch.size(), 0, &res))
            throw std::runtime_error(std::string("mgx_align_batch: ") + mgx_last_error() + " (" + std::to_string(rc) + ")");
        for (size_t t = 0; t < seq_batch.size(); ++t) {
            if (res.status[t] != MGX_OK)
                throw std::runtime_error("query " + seq_batch[t].first + ": status " + std::to_string(res.status[t])
                                         + " after the capacity retry of mgx_align_batch");
            AlignmentResults &paths = results[t];
            for (uint64_t ai = res.aln_begin[t]; ai < res.aln_begin[t + 1]; ++ai) {
                const mgx_alignment &a = res.alignments[ai];
                const std::string &q = paths.get_query(a.orientation);
                // Alignment's constructor (alignment.hpp:142-152) starts cigar_ with (CLIPPED, clipping) and appends the cigar it
                // is given, and Cigar::append merges equal neighbouring operators: the device CIGAR's own leading soft clip
                // must not be passed a second time (the trailing one is part of the cigar argument, as in the reference)
                Cigar cigar;
                for (uint32_t x = 0; x < a.n_cigar; ++x) {
                    const mgx_cigar_op &op = res.cigar[a.cigar_begin + x];
                    if (x == 0 && op.op == MGX_OP_CLIPPED && a.clipping) continue;
                    cigar.append(static_cast<Cigar::Operator>(op.op), op.len);
                }
                std::vector<DeBruijnGraph::node_index> nodes(res.nodes + a.nodes_begin, res.nodes + a.nodes_begin + a.n_nodes);
                Alignment aln(std::string_view(q).substr(a.clipping, q.size() - a.clipping - a.end_clipping),
                              std::move(nodes), std::string(res.seqs + a.seq_begin, a.seq_len), a.score, std::move(cigar),
                              a.clipping, a.orientation, a.offset);
                if (hip_annotation_) {
                    // Alignment::label_columns (alignment.hpp:285): ascending column indices; the names stay on this side
                    // (format_alignment decodes them through the annotator's encoder, cli/align.cpp:262-281,461-465)
                    aln.label_columns.assign(res.labels + a.labels_begin, res.labels + a.labels_begin + a.n_labels);
                }
                paths.emplace_back(std::move(aln));
            }
        }
        for (size_t i = 0; i < seq_batch.size(); ++i) callback(seq_batch[i].first, std::move(results[i]));
    }

  private:
    const DeBruijnGraph &graph_;
    DBGAlignerConfig config_;
    const HipGraphHandle &hip_graph_;
    const HipAnnotationHandle *hip_annotation_;
    mgx_aligner *a_ = nullptr;
};

// LabeledAligner<>(graph, config, annotator) (aligner_labeled.hpp:125-127): the switch in cli/align.cpp:452-459 becomes
//     aligner = config->align_device >= 0 && hip_annotation
//         ? std::unique_ptr<IDBGAligner>(new HipLabeledAlignerAdapter(*aln_graph, aligner_config, hip_graph, *hip_annotation))
//         : std::make_unique<LabeledAligner<>>(*aln_graph, aligner_config, anno_dbg->get_annotator());
// with `hip_annotation` created next to the annotation load, inside a try block: an annotator with coordinates throws there.
class HipLabeledAlignerAdapter : public HipDBGAlignerAdapter {
  public:
    HipLabeledAlignerAdapter(const DeBruijnGraph &graph, const DBGAlignerConfig &config, const HipGraphHandle &hip_graph,
                             const HipAnnotationHandle &hip_annotation)
          : HipDBGAlignerAdapter(graph, config, hip_graph, &hip_annotation) {}
};

} // namespace align
} // namespace graph
} // namespace mtg
