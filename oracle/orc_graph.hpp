// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of the BOSS / DBGSuccinct query path of ratschlab/metagraph.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
// The product (libmgx.so) never includes, links or calls this code.
//
// Parity status: the reference cannot be compiled here (all third-party submodules are
// absent).  rank/select are mathematically specified, so this restatement is
// pinned by the reference's own known-answer tests transcribed under tests/golden/ — incl. the JSON goldens
// genome_MT1.align{,.edit}.json whose node ids fix the BOSS edge numbering of build_boss(), the header of a
// reference-written .dbg file, and the suffix-matching / map_to_edges vectors of test_dbg_succinct.cpp / test_boss.cpp.
//
// Citations are into /root/reference/metagraph/src (M/src).
#pragma once
#include <cstdint>
#include <cstddef>
#include <functional>
#include <string>
#include <string_view>
#include <tuple>
#include <utility>
#include <vector>

namespace orc {

using edge_t = uint64_t;
using node_t = uint64_t;
constexpr int SIGMA = 5;            // "$ACGT", kmer/alphabets.hpp:64
constexpr node_t NPOS = 0;          // graph/representation/base/sequence_graph.hpp:29-30

// kmer/alphabets.hpp:67-76 (kBOSSCharToDNA) through KmerExtractorBOSS::encode
// (kmer/kmer_extractor.cpp:33-36,280-282): bytes < 0 map like byte 0.
uint8_t encode_char(char c);
inline char decode_code(uint8_t c) { return "$ACGT"[c]; }          // kmer_extractor.cpp:284
std::vector<uint8_t> encode_seq(std::string_view s);
// common/seq_tools/reverse_complement.hpp:31-76
char complement_char(char c);
void reverse_complement_inplace(std::string &s);

enum Mode { BASIC = 0, CANONICAL = 1, PRIMARY = 2 };

// Plain-array BOSS table with block rank/select support.
struct Boss {
    size_t k_ = 0;                       // BOSS node length (DBG k - 1)
    uint64_t n = 0;                      // number of edges; arrays have n + 1 entries
    std::vector<uint8_t> W;              // boss.hpp W_
    std::vector<uint8_t> last;           // boss.hpp last_
    uint64_t F[SIGMA] = {0};             // boss.hpp:506-510
    uint64_t NF[SIGMA] = {0};            // boss.cpp:1095-1101

    // support structures (block = 64 edges)
    std::vector<uint32_t> w_cum;         // per block: # of W==c (c < SIGMA) in blocks before it, 5 per block
    std::vector<uint32_t> last_cum;      // per block: # of last bits before it
    std::vector<uint32_t> last_hint;     // position/64 of every 64-th set bit of last
    std::vector<uint32_t> w_hint[SIGMA]; // same for each unflagged W symbol

    void finalize();                     // builds NF and the support structures

    uint8_t get_W(edge_t i) const { return W[i]; }
    bool get_last(edge_t i) const { return last[i]; }
    uint64_t num_edges() const { return n; }                 // boss.cpp:1068-1070

    uint64_t rank_W(edge_t i, uint8_t c) const;              // boss.cpp:437-441
    uint64_t rank_last(edge_t i) const;                      // boss.cpp:577-581
    edge_t select_last(uint64_t r) const;                    // boss.cpp:588-592
    edge_t select_W(uint8_t c, uint64_t r) const;            // wavelet_tree::select as used in boss.cpp:635
    edge_t pred_last(edge_t i) const;                        // boss.cpp:598-607
    edge_t succ_last(edge_t i) const;                        // boss.cpp:613-617
    std::pair<edge_t, uint8_t> succ_W(edge_t i, uint8_t a, uint8_t b) const;   // boss.cpp:515-570
    uint8_t get_node_last_value(edge_t i) const;             // boss.cpp:679-690
    edge_t bwd(edge_t i) const;                              // boss.cpp:623-636
    edge_t fwd(edge_t i, uint8_t c) const;                   // boss.cpp:642-652
    edge_t pick_edge(edge_t edge, uint8_t c) const;          // boss.cpp:710-722
    void call_incoming_to_target(edge_t edge, uint8_t d,
                                 const std::function<void(edge_t)> &cb) const;  // boss.cpp:766-786
    bool is_single_incoming(edge_t i, uint8_t w) const;      // boss.cpp:802-815
    size_t num_incoming_to_target(edge_t x, uint8_t d) const;// boss.cpp:821-838
    bool tighten_range(edge_t *rl, edge_t *ru, uint8_t s) const;               // boss.hpp:682-693
    edge_t index(const uint8_t *begin, const uint8_t *end) const;              // boss.hpp:696-718
    // returns (first, last, matched length)                                   // boss.hpp:720-764
    std::tuple<edge_t, edge_t, size_t> index_range(const uint8_t *begin, const uint8_t *end) const;
    edge_t map_to_edge(const uint8_t *begin, const uint8_t *end) const;        // boss.hpp:766-777
    std::vector<edge_t> map_to_edges(const std::vector<uint8_t> &seq) const;   // boss.cpp:996-1045
    std::vector<uint8_t> get_node_seq(edge_t i) const;       // boss.cpp:940-973
};

// Test-fixture BOSS builder: the edge set of construct_boss_chunk
// (graph/representation/succinct/boss_chunk_construct.cpp:57-171,341-430) laid out by
// initialize_chunk (boss_chunk.cpp:32-125).  k is the DBG k-mer length.
Boss build_boss(size_t k, const std::vector<std::string> &sequences, Mode mode);

// DBGSuccinct restricted to what the aligner consumes (dbg_succinct.cpp).
struct Graph {
    Boss boss;
    Mode mode = BASIC;
    std::vector<uint8_t> valid;          // empty = no mask (cli/align.cpp:337-339 reset_mask)
    std::vector<uint8_t> first_char;     // optional: first character code of every edge's k-mer; stands in for
                                         // NodeFirstCache (graph_extensions/node_first_cache.cpp:9-118), results identical
    void build_first_chars(unsigned threads);

    size_t get_k() const { return boss.k_ + 1; }             // dbg_succinct.cpp:43-45
    uint64_t max_index() const { return boss.n; }            // dbg_succinct.cpp:686-688
    bool in_graph(node_t v) const {                          // dbg_succinct.cpp:934-936
        return v > 0 && v <= boss.n && (valid.empty() || valid[v]);
    }
    node_t validate_edge(edge_t e) const { return in_graph(e) ? e : NPOS; }    // :937-939
    void mask_dummy_kmers();                                 // dbg_succinct.cpp:924-932, boss.cpp:1765-1775

    std::vector<node_t> map_to_nodes_sequentially(std::string_view seq) const; // dbg_succinct.cpp:285-305
    std::vector<node_t> map_to_nodes(std::string_view seq) const;               // dbg_succinct.cpp:428-500 (CANONICAL: the smaller BOSS index of a k-mer and its reverse complement)
    void call_outgoing_kmers(node_t v, const std::function<void(node_t, char)> &cb) const; // :110-139
    void call_incoming_kmers(node_t v, const std::function<void(node_t, char)> &cb) const; // node_first_cache.cpp:38-52
    bool has_multiple_outgoing(node_t v) const;              // dbg_succinct.cpp:609-624
    bool has_single_incoming(node_t v) const;                // dbg_succinct.cpp:658-678
    std::string get_node_sequence(node_t v) const;           // dbg_succinct.cpp:178-190
    void call_nodes_with_suffix_matching_longest_prefix(     // dbg_succinct.cpp:307-393
        std::string_view str, const std::function<void(node_t, uint64_t)> &cb,
        size_t min_match_length, size_t max_num_allowed_matches = SIZE_MAX) const;
    uint64_t num_nodes() const;
};

struct CanonicalView;

// View of a graph as itself, as its reverse complement (graph/representation/rc_dbg.hpp:15-154), or through the
// CanonicalDBG wrapper of a PRIMARY graph (`canon`, never together with rc).
struct GraphView {
    const Graph *g = nullptr;
    bool rc = false;
    const CanonicalView *canon = nullptr;
    size_t get_k() const { return g->get_k(); }
    uint64_t max_index() const;
    // rc_dbg.hpp:88-99: children in the RC graph are parents in G with complemented first char
    void call_outgoing_kmers(node_t v, const std::function<void(node_t, char)> &cb) const;
    void adjacent_incoming_nodes(node_t v, const std::function<void(node_t)> &cb) const;
    std::string get_node_sequence(node_t v) const;           // rc_dbg.hpp:118-122
    std::vector<node_t> map_to_nodes_sequentially(std::string_view seq) const;
    bool has_multiple_outgoing(node_t v) const;
    bool has_single_incoming(node_t v) const;
    // reverse_complement_seq_path (sequence_graph.cpp:563-573)
    void reverse_complement_seq_path(std::string &seq, std::vector<node_t> &path) const;
};

// CanonicalDBG over a PRIMARY DBGSuccinct (graph/representation/canonical_dbg.{hpp,cpp}): every k-mer of the base graph
// also stands for its reverse complement, which gets the id base + offset (offset = base graph's max_index).  Caches of the
// reference (NodeFirstCache LRUs, is_palindrome_cache_) do not change results and are not modelled.
struct CanonicalView {
    const Graph *g = nullptr;
    uint64_t offset = 0;                 // canonical_dbg.cpp:26
    bool k_odd = false;                  // :27
    bool has_sentinel = false;           // :41 (!dbg_succ->get_mask())
    explicit CanonicalView(const Graph &base);

    size_t get_k() const { return g->get_k(); }
    uint64_t max_index() const { return 2 * offset; }        // canonical_dbg.hpp
    node_t get_base_node(node_t v) const { return v > offset ? v - offset : v; }
    node_t reverse_complement(node_t v) const;               // canonical_dbg.cpp:515-549
    void reverse_complement(std::string &seq, std::vector<node_t> &path) const;   // :551-560
    std::vector<node_t> map_to_nodes_sequentially(std::string_view seq) const;     // :55-146
    std::string get_node_sequence(node_t v) const;           // :423-432
    void call_outgoing_kmers(node_t v, const std::string &hint, const std::function<void(node_t, char)> &cb) const;   // :156-240
    void call_incoming_kmers(node_t v, const std::string &hint, const std::function<void(node_t, char)> &cb) const;   // :242-330
    void call_outgoing_kmers(node_t v, const std::function<void(node_t, char)> &cb) const { call_outgoing_kmers(v, get_node_sequence(v), cb); }
    void call_incoming_kmers(node_t v, const std::function<void(node_t, char)> &cb) const { call_incoming_kmers(v, get_node_sequence(v), cb); }
    void adjacent_outgoing_nodes(node_t v, const std::function<void(node_t)> &cb) const;   // :346-359
    void adjacent_incoming_nodes(node_t v, const std::function<void(node_t)> &cb) const;   // :331-344
    bool has_multiple_outgoing(node_t v) const;              // :361-375
    bool has_single_incoming(node_t v) const;                // :377-387

  private:
    edge_t prefix_rc(node_t v, const std::string &spelling) const;     // NodeFirstCache::get_prefix_rc, node_first_cache.cpp:120-146
    edge_t suffix_rc(node_t v, const std::string &spelling) const;     // NodeFirstCache::get_suffix_rc, :148-174
    void adjacent_incoming_rc_strand(node_t v, const std::string &hint, const std::function<void(node_t, char)> &cb) const;  // :574-632
    void adjacent_outgoing_rc_strand(node_t v, const std::string &hint, const std::function<void(node_t, char)> &cb) const;  // :634-684
};

} // namespace orc
