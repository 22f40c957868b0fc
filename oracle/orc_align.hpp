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

// A/alignment.hpp:32-98
struct Seed {
    std::string_view query_view;
    std::vector<node_t> nodes;
    bool orientation = false;
    size_t offset = 0;
    uint32_t clipping = 0, end_clipping = 0;
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

// cli/align.cpp:254-285 + fmt formatter alignment.hpp:426-433
std::string format_alignment_tsv(const std::string &header, const AlignmentResults &paths, int32_t min_path_score);

} // namespace orc
