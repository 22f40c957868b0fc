// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of MetaGraph's DBGAligner path (graph/alignment/*) for BASIC-mode
// DBGSuccinct graphs: SuffixSeeder<UniMEMSeeder> + DefaultColumnExtender + AlignmentAggregator.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/.
// Citations are into /root/reference/metagraph/src/graph/alignment (A/).
//
// Third-party arithmetic restated from published algorithms (libraries absent from the tree):
//   * sdust (github.com/lh3/sdust, symmetric DUST): PARITY WEAKLY PINNED — only the boolean
//     outcome of test_aligner.cpp:1345-1363 and the default-config KATs exercise it.
//   * Priority-Deque / std::sort tie order for num_alternative_paths > 1: parity unpinned.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <string_view>
#include <vector>

#include "../include/mgx.h"
#include "orc_graph.hpp"

namespace orc {

using score_t = int32_t;
constexpr score_t NINF = INT32_MIN + 100;          // A/aligner_config.hpp:31

// A/aligner_cigar.hpp:16-109
struct Cigar {
    typedef std::pair<uint8_t, uint32_t> value_type;
    std::vector<value_type> ops;
    Cigar() {}
    Cigar(uint8_t op, uint32_t num) { if (num) ops.emplace_back(op, num); }
    void append(uint8_t op, uint32_t num = 1);
    void append(Cigar &&other);
    uint32_t get_clipping() const { return ops.size() && ops.front().first == MGX_OP_CLIPPED ? ops.front().second : 0; }
    uint32_t get_end_clipping() const { return ops.size() && ops.back().first == MGX_OP_CLIPPED ? ops.back().second : 0; }
    uint32_t trim_clipping();
    uint32_t trim_end_clipping();
    void extend_clipping(uint32_t n);
    size_t get_num_matches() const;
    std::string to_string() const;
    bool is_valid(std::string_view reference, std::string_view query) const;
    bool operator==(const Cigar &o) const { return ops == o.ops; }
};

typedef uint64_t Label;                      // annotation column = label id (annot::matrix::BinaryMatrix::Column)
typedef std::vector<Label> Columns;          // Alignment::Columns: a sorted label set
typedef std::vector<int64_t> Tuple;          // Alignment::Tuple (alignment.hpp:137): the coordinates of one label, ascending
typedef std::vector<Tuple> CoordinateSet;    // Alignment::CoordinateSet: one Tuple per entry of label_columns

// A/alignment.hpp:32-98
struct Seed {
    std::string_view query_view;
    std::vector<node_t> nodes;
    bool orientation = false;
    size_t offset = 0;
    uint32_t clipping = 0, end_clipping = 0;
    Columns label_columns;                   // :84 (filled by LabeledAligner::filter_seeds)
    CoordinateSet label_coordinates;         // :86-89 coordinates of the seed's first nucleotide, per label
    bool has_label_encoder = false;          // :82 label_encoder != nullptr
    bool empty() const { return nodes.empty(); }
    // :74-80: more nodes for the seed; the query has |next| further characters and the path still spells the view
    void expand(const std::vector<node_t> &next) {
        query_view = std::string_view(query_view.data(), query_view.size() + next.size());
        end_clipping -= next.size();
        nodes.insert(nodes.end(), next.begin(), next.end());
    }
};

// A/alignment.hpp:132-331
struct Alignment {
    std::string_view query_view;
    std::vector<node_t> nodes;
    bool orientation = false;
    size_t offset = 0;
    std::string sequence;
    score_t score = 0;
    Cigar cigar;
    score_t extra_score = 0;
    Columns label_columns;                   // alignment.hpp:285
    CoordinateSet label_coordinates;         // :287-290 per label: the coordinates of the alignment's first nucleotide

    Alignment() {}
    Alignment(std::string_view query, std::vector<node_t> &&nodes_, std::string &&seq, score_t score_,
              Cigar &&cigar_, size_t clipping, bool orientation_, size_t offset_);
    Alignment(const Seed &seed, const mgx_config &config);           // alignment.hpp:154-165

    bool empty() const { return nodes.empty(); }
    size_t size() const { return nodes.size(); }
    uint32_t get_clipping() const { return cigar.get_clipping(); }
    uint32_t get_end_clipping() const { return cigar.get_end_clipping(); }
    void extend_query_begin(const char *begin);                      // alignment.hpp:209-214
    void extend_query_end(const char *end);                          // alignment.hpp:216-222
    size_t trim_offset();                                            // alignment.cpp:177-190
    size_t trim_clipping() { return cigar.trim_clipping(); }         // alignment.hpp:224
    size_t trim_end_clipping() { return cigar.trim_end_clipping(); } // alignment.hpp:225
    bool append(Alignment &&other);                                  // alignment.cpp:94-175
    bool splice(Alignment &&other);                                  // alignment.hpp:196-205
    size_t trim_query_suffix(size_t n, const mgx_config &config, bool trim_excess_deletions = true);                          // :280-362
    size_t trim_reference_prefix(size_t n, size_t node_overlap, const mgx_config &config, bool trim_excess_insertions = true); // :364-455
    size_t trim_reference_suffix(size_t n, const mgx_config &config, bool trim_excess_insertions = true);                     // :457-538
    void splice_with_unknown(Alignment &&other, size_t num_unknown, size_t node_overlap, const mgx_config &config);           // :1048-1152
    std::string format_coords(const std::vector<std::string> &label_names) const;                                             // :20-37
    // :39-92 with annot::CoordToHeader as (headers, k-mer counts) per column
    std::string format_coords(const std::vector<std::vector<std::string>> &headers, const std::vector<std::vector<uint64_t>> &kmer_counts,
                              size_t k) const;
    size_t trim_query_prefix(size_t n, size_t node_overlap, const mgx_config &config, bool trim_excess_deletions = true); // :192-278
    void insert_gap_prefix(ptrdiff_t gap_length, size_t node_overlap, const mgx_config &config);                        // :1154-1234
    void reverse_complement(const GraphView &graph, std::string_view query_rev_comp); // alignment.cpp:540-702
    bool is_valid(const GraphView &graph, const mgx_config *config, std::string *why = nullptr) const; // :1316-1345
    bool operator==(const Alignment &o) const {
        return orientation == o.orientation && offset == o.offset && score == o.score
            && query_view == o.query_view && sequence == o.sequence && cigar == o.cigar && nodes == o.nodes;
    }
};

score_t score_sequences(const mgx_config &c, std::string_view a, std::string_view b);   // aligner_config.hpp:65-70
inline score_t match_score(const mgx_config &c, std::string_view q) { return score_sequences(c, q, q); }
score_t score_cigar(const mgx_config &c, std::string_view ref, std::string_view query, const Cigar &cigar); // aligner_config.cpp:68-126
bool check_config_scores(const mgx_config &c);                                           // aligner_config.cpp:39-66
std::string spell_path(const GraphView &graph, const std::vector<node_t> &path, size_t offset); // alignment.cpp:1239-1314

// lh3 symmetric DUST, as called at A/aligner_seeder_methods.cpp:22-29 (T=20, W=64)
bool is_low_complexity(std::string_view s, int T = 20, int W = 64);

// Per-read statistics of the work the algorithm required (for the roofline model)
struct WorkCounters {
    uint64_t n_map_fwd = 0;        // fwd+pick_edge steps in map_to_edges
    uint64_t n_index_steps = 0;    // tighten_range steps
    uint64_t n_terminus = 0;       // is_mem_terminus evaluations
    uint64_t n_expansions = 0;     // call_outgoing_kmers graph expansions in extension
    uint64_t n_columns = 0;        // DP columns created
    uint64_t n_extensions = 0;
    uint64_t n_seeds = 0;
    void add(const WorkCounters &o) {
        n_map_fwd += o.n_map_fwd; n_index_steps += o.n_index_steps; n_terminus += o.n_terminus;
        n_expansions += o.n_expansions; n_columns += o.n_columns; n_extensions += o.n_extensions;
        n_seeds += o.n_seeds;
    }
};

// One query's results: AlignmentResults (alignment.hpp:366-406)
struct AlignmentResults {
    std::string query, query_rc;
    std::vector<Alignment> alignments;
    // intermediate products exposed for parity tests of the GPU stages
    std::vector<node_t> nodes_fwd, nodes_rc;
    std::vector<Seed> seeds_fwd, seeds_rc;
    size_t num_matches_fwd = 0, num_matches_rc = 0;
};

// DBGAligner<SuffixSeeder<UniMEMSeeder>, DefaultColumnExtender, LocalAlignmentLess> (A/dbg_aligner.hpp:42-99)
class Aligner {
  public:
    Aligner(const Graph &graph, const mgx_config &config);           // dbg_aligner.cpp:33-61 (throws on bad scores)
    const mgx_config &get_config() const { return config_; }
    // dbg_aligner.cpp:251-355; results[i] corresponds to queries[i]
    void align_batch(const std::vector<std::string> &queries, std::vector<AlignmentResults> *results,
                     WorkCounters *counters = nullptr) const;
    AlignmentResults align(std::string_view query) const;            // dbg_aligner.cpp:22-31

  private:
    const Graph &graph_;
    mgx_config config_;
};

// The annotation as the aligner sees it: a binary matrix rows x columns, row = node - 1
// (AnnotatedDBG::graph_to_anno_index, annotated_dbg.hpp:50-52), column = label.  Restated as ColumnMajor
// (annotation/binary_matrix/column_sparse/column_major.cpp:27-44: get_rows tests every column's bit vector at every
// requested row) on plain 64-bit words — one word holds 64 consecutive rows of one label.
struct Annotation {
    uint64_t n_rows = 0;
    std::vector<std::vector<uint64_t>> columns;
    // k-mer coordinates (annot::matrix::MultiIntMatrix behind a ColumnCoordAnnotator): per column, row -> ascending coordinates.
    // has_coordinates: AnnotationBuffer::has_coordinates() (annotation_buffer.hpp) — the annotator carries a MultiIntMatrix.
    bool has_coordinates = false;
    std::vector<std::map<uint64_t, Tuple>> coordinates;
    void resize(uint64_t rows, size_t n_columns) {
        n_rows = rows; columns.assign(n_columns, std::vector<uint64_t>((rows + 63) / 64, 0));
        coordinates.assign(n_columns, {});
    }
    // AnnotatedDBG::annotate_kmer_coords (annotated_dbg.cpp:192-233) for one (sequence, { label }, first coordinate): the i-th
    // k-mer of the sequence has coordinate start + i, recorded for the k-mers found in the graph
    void annotate_kmer_coords(const Graph &graph, std::string_view sequence, size_t column, uint64_t start);
    // MultiIntMatrix::get_row_tuples for one row: (column, coordinates) pairs, columns ascending
    std::vector<std::pair<Label, Tuple>> get_row_tuples(uint64_t row) const;
    void set(uint64_t row, size_t column) { columns[column][row >> 6] |= 1ull << (row & 63); }
    bool get(uint64_t row, size_t column) const { return (columns[column][row >> 6] >> (row & 63)) & 1; }
    std::vector<Columns> get_rows(const std::vector<uint64_t> &rows) const;      // column_major.cpp:27-44 (labels ascending)
    // AnnotatedDBG::annotate_sequence (annotated_dbg.cpp:55-75): every k-mer of `sequence` found in the graph gets `column`
    void annotate_sequence(const Graph &graph, std::string_view sequence, size_t column);
};

// LabeledAligner<SuffixSeeder<ExactSeeder>, LabeledExtender, LocalAlignmentLess> (A/aligner_labeled.{hpp,cpp}).  Annotation
// with coordinates (round 6): the constructor switches seed chaining on and the global x-drop off
// (aligner_labeled.cpp:457-462) — chain_seeds / call_seed_chains_both_strands (aligner_chainer.cpp:64-542), extend_chain /
// align_connect (dbg_aligner.cpp:155-250,388-529), coordinate-consistent call_outgoing and call_alignments
// (aligner_labeled.cpp:245-300,361-448).  Annotation without coordinates: seeds are filtered by label (build_seeders :479-558, filter_seeds :612-721), every column of the DP
// table carries a label set that is intersected along the tree (LabeledExtender::call_outgoing :176-302, flush :81-137),
// backtracking reports one alignment per not-yet-seen seed label set (skip_backtrack_start :304-326, call_alignments
// :328-448), the aggregator keeps a queue per label (aligner_aggregator.hpp:68-138).
class LabeledAligner {
  public:
    LabeledAligner(const Graph &graph, const mgx_config &config, const Annotation &annotation);   // aligner_labeled.cpp:450-466
    const mgx_config &get_config() const { return config_; }
    void align_batch(const std::vector<std::string> &queries, std::vector<AlignmentResults> *results) const;
    AlignmentResults align(std::string_view query) const;

  private:
    const Graph &graph_;
    mgx_config config_;
    const Annotation &annotation_;
};

// chain_seeds (A/aligner_chainer.cpp:383-539: the sort and the banded DP) over one list of anchors, in place: sorted as the
// reference sorts them, chain_score final, backtrace[i] = index of the anchor i's best chain continues with (0xFFFFFFFF: none)
void chain_anchors(const mgx_config &config, uint32_t query_size, mgx_chain_anchor *anchors, size_t n, uint32_t *backtrace);

// cli/align.cpp:254-285 + fmt formatter alignment.hpp:426-433
std::string format_alignment_tsv(const std::string &header, const AlignmentResults &paths, int32_t min_path_score);

} // namespace orc
