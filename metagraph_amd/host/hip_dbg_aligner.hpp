// hip_dbg_aligner.hpp — C++ host adapter: MetaGraph's aligner operator surface over the C-ABI.
//
// Mirrors, name for name, what `metagraph align` consumes from graph/alignment:
//   IDBGAligner::align_batch / align / get_config      (dbg_aligner.hpp:20-39, dbg_aligner.cpp:22-31)
//   Query, AlignmentCallback                            (dbg_aligner.hpp:22-24)
//   AlignmentResults (normalised query + RC + alignments, alignment.hpp:366-406; ctor alignment.cpp:1348-1372)
//   Alignment getters + Cigar::to_string                (alignment.hpp:132-331, aligner_cigar.cpp:86-96)
//   format_alignment TSV                                (cli/align.cpp:254-285)
// Header-only; links against libmgx.so only (include/mgx.h).  Callbacks are invoked once per query, in
// batch order, on the calling thread, like DBGAligner::align_batch (dbg_aligner.cpp:263,353).
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <algorithm>
#include <stdexcept>
#include <cstdlib>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/mgx.h"

namespace mgx {
namespace host {

typedef mgx_config DBGAlignerConfig;      // field-for-field mirror (aligner_config.hpp:18-94)
typedef uint64_t node_index;

class Cigar {
  public:
    typedef std::pair<uint8_t, uint32_t> value_type;
    const std::vector<value_type> &data() const { return cigar_; }
    std::vector<value_type> &data() { return cigar_; }
    std::string to_string() const {
        static const char op_str[] = "SX=DIG";
        std::string s;
        for (const auto &p : cigar_) s += std::to_string(p.second) + op_str[p.first];
        return s;
    }
    size_t get_num_matches() const {
        size_t n = 0;
        for (const auto &p : cigar_) n += (p.first == MGX_OP_MATCH) * p.second;
        return n;
    }
    uint32_t get_clipping() const { return cigar_.size() && cigar_.front().first == MGX_OP_CLIPPED ? cigar_.front().second : 0; }
    uint32_t get_end_clipping() const { return cigar_.size() && cigar_.back().first == MGX_OP_CLIPPED ? cigar_.back().second : 0; }
  private:
    std::vector<value_type> cigar_;
};

class Alignment {
  public:
    std::string_view get_query_view() const { return query_view_; }
    const std::vector<node_index> &get_nodes() const { return nodes_; }
    std::string_view get_sequence() const { return sequence_; }
    size_t get_offset() const { return offset_; }
    size_t size() const { return nodes_.size(); }
    bool empty() const { return nodes_.empty(); }
    bool get_orientation() const { return orientation_; }
    int32_t get_score() const { return score_; }
    const Cigar &get_cigar() const { return cigar_; }
    uint32_t get_clipping() const { return cigar_.get_clipping(); }
    uint32_t get_end_clipping() const { return cigar_.get_end_clipping(); }
    // Alignment::label_columns (alignment.hpp:285): the labels of a label-aware aligner's alignment, ascending; else empty
    const std::vector<uint64_t> &get_label_columns() const { return label_columns_; }
    // fmt formatter of the reference (alignment.hpp:426-433)
    std::string to_tsv_fields() const {
        return std::string(orientation_ ? "-" : "+") + "\t" + sequence_ + "\t" + std::to_string(score_) + "\t"
            + std::to_string(cigar_.get_num_matches()) + "\t" + cigar_.to_string() + "\t" + std::to_string(offset_);
    }
  private:
    friend class HipDBGAligner;
    std::string_view query_view_;
    std::vector<node_index> nodes_;
    bool orientation_ = false;
    size_t offset_ = 0;
    std::string sequence_;
    int32_t score_ = 0;
    Cigar cigar_;
    std::vector<uint64_t> label_columns_;
};

class AlignmentResults {
  public:
    explicit AlignmentResults(std::string_view query = {}) {
        // alignment.cpp:1348-1372: upper-case, bytes < 0 -> 127, SSO disabled, reverse complement
        query_.reserve(std::max<size_t>(query.size(), 32) + 8);
        for (char ch : query) {
            int8_t c = (int8_t)ch;
            query_.push_back(c >= 0 ? (char)toupper(c) : (char)127);
        }
        query_rc_.reserve(query_.capacity());
        query_rc_.assign(query_.rbegin(), query_.rend());
        for (char &c : query_rc_) c = complement(c);
    }
    AlignmentResults(const AlignmentResults &) = delete;
    AlignmentResults &operator=(const AlignmentResults &) = delete;
    AlignmentResults(AlignmentResults &&) = default;
    AlignmentResults &operator=(AlignmentResults &&) = default;
    const std::string &get_query(bool reverse_complement = false) const { return reverse_complement ? query_rc_ : query_; }
    size_t size() const { return alignments_.size(); }
    bool empty() const { return alignments_.empty(); }
    const Alignment &operator[](size_t i) const { return alignments_[i]; }
    auto begin() const { return alignments_.begin(); }
    auto end() const { return alignments_.end(); }
  private:
    friend class HipDBGAligner;
    static char complement(char c) {      // COMPL_TAB (common/seq_tools/reverse_complement.hpp:31-48)
        static const char up[] = "TVGHEFCDIJMLKNOPQYSAABWXRZ";
        unsigned char u = (unsigned char)c;
        if (u >= 'A' && u <= 'Z') return up[u - 'A'];
        if (u >= 'a' && u <= 'z') return (char)(up[u - 'a'] + 32);
        if (u == 96) return 64;
        return c;
    }
    std::string query_, query_rc_;
    std::vector<Alignment> alignments_;
};

// The graph handle the aligner holds a reference to (DBGSuccinct's role for this path)
class HipBOSSGraph {
  public:
    // W/last: one byte per edge, n_edges + 1 entries; valid may be null (reset_mask(), cli/align.cpp:337-339)
    HipBOSSGraph(uint32_t k, uint64_t n_edges, const uint8_t *W, const uint8_t *last, const uint64_t F[5],
                 const uint8_t *valid = nullptr, int device = 0, uint32_t mode = MGX_MODE_BASIC) {
        mgx_boss_view v{};
        v.k = k; v.sigma = 5; v.n_edges = n_edges; v.W = W; v.last = last; v.F = F; v.valid = valid;
        v.mode = mode; v.on_device = 0;          // DeBruijnGraph::get_mode(); PRIMARY graphs are aligned through CanonicalDBG
        if (int rc = mgx_graph_create(&v, device, &g_)) throw std::runtime_error(std::string("mgx_graph_create: ") + mgx_last_error() + " (" + std::to_string(rc) + ")");
    }
    ~HipBOSSGraph() { mgx_graph_destroy(g_); }
    HipBOSSGraph(const HipBOSSGraph &) = delete;
    size_t get_k() const { return mgx_graph_k(g_); }
    uint64_t max_index() const { return mgx_graph_max_index(g_); }
    mgx_graph *handle() const { return g_; }
  private:
    mgx_graph *g_ = nullptr;
};

// In-process multi-GPU (the natural drop-in for `metagraph align`, which is ONE process with a pool of batch tasks,
// cli/align.cpp:440-475): one replica of the graph per device; worker w of the pool aligns its batches on device w % D.  Reads
// shard by batch, the graph is replicated, nothing is exchanged between devices — results come back per batch on the host.
class HipGraphSet {
  public:
    HipGraphSet(int n_devices, uint32_t k, uint64_t n_edges, const uint8_t *W, const uint8_t *last, const uint64_t F[5],
                const uint8_t *valid = nullptr, uint32_t mode = MGX_MODE_BASIC) {
        if (n_devices < 1) throw std::runtime_error("HipGraphSet: at least one device");
        for (int d = 0; d < n_devices; ++d) replicas_.emplace_back(new HipBOSSGraph(k, n_edges, W, last, F, valid, d, mode));
    }
    size_t size() const { return replicas_.size(); }
    static size_t device_of_worker(size_t worker, size_t n_devices) { return worker % n_devices; }
    const HipBOSSGraph &for_worker(size_t worker) const { return *replicas_[device_of_worker(worker, replicas_.size())]; }
  private:
    std::vector<std::unique_ptr<HipBOSSGraph>> replicas_;
};

class IDBGAligner {
  public:
    typedef std::pair<std::string /* header */, std::string /* seq */> Query;
    typedef std::function<void(const std::string & /* header */, AlignmentResults && /* alignments */)> AlignmentCallback;
    virtual ~IDBGAligner() {}
    virtual const HipBOSSGraph &get_graph() const = 0;
    virtual const DBGAlignerConfig &get_config() const = 0;
    virtual void align_batch(const std::vector<Query> &seq_batch, const AlignmentCallback &callback) const = 0;
    AlignmentResults align(std::string_view query) const {          // dbg_aligner.cpp:22-31
        AlignmentResults result;
        align_batch({ Query{ std::string{}, std::string(query) } },
                    [&](const std::string &, AlignmentResults &&alignment) { std::swap(result, alignment); });
        return result;
    }
    virtual bool has_coordinates() const = 0;
};

// The annotator argument of LabeledAligner<> (aligner_labeled.hpp:125-127): the label matrix on the graph's device, from the
// set rows of every column (what a ColumnCompressed annotation stores; row = AnnotatedDBG::graph_to_anno_index(node) = node - 1)
class HipAnnotation {
  public:
    HipAnnotation(uint64_t n_rows, const std::vector<uint64_t> &col_begin, const std::vector<uint64_t> &rows, int device = 0) {
        if (col_begin.empty()) throw std::runtime_error("HipAnnotation: col_begin needs n_labels + 1 entries");
        if (int rc = mgx_annotation_create_sparse(n_rows, (uint32_t)(col_begin.size() - 1), col_begin.data(), rows.data(), 0, device, &a_))
            throw std::runtime_error(std::string("mgx_annotation_create_sparse: ") + mgx_last_error() + " (" + std::to_string(rc) + ")");
    }
    ~HipAnnotation() { mgx_annotation_destroy(a_); }
    HipAnnotation(const HipAnnotation &) = delete;
    uint32_t num_labels() const { return mgx_annotation_num_labels(a_); }
    mgx_annotation *handle() const { return a_; }
  private:
    mgx_annotation *a_ = nullptr;
};

class HipDBGAligner : public IDBGAligner {
  public:
    // throws std::runtime_error like the reference when check_config_scores() fails (dbg_aligner.cpp:55-56)
    HipDBGAligner(const HipBOSSGraph &graph, const DBGAlignerConfig &config, const mgx_limits *limits = nullptr)
          : graph_(graph) {
        if (int rc = mgx_aligner_create(graph.handle(), &config, limits, &a_))
            throw std::runtime_error(std::string(mgx_last_error()) + " (" + std::to_string(rc) + ")");
        mgx_aligner_get_config(a_, &config_);
        own_stream();
    }
    // LabeledAligner<>(graph, config, annotator): every alignment of the results carries its label_columns
    HipDBGAligner(const HipBOSSGraph &graph, const DBGAlignerConfig &config, const HipAnnotation &annotation,
                  const mgx_limits *limits = nullptr)
          : graph_(graph), annotation_(&annotation) {
        if (int rc = mgx_labeled_aligner_create(graph.handle(), &config, limits, annotation.handle(), &a_))
            throw std::runtime_error(std::string(mgx_last_error()) + " (" + std::to_string(rc) + ")");
        mgx_aligner_get_config(a_, &config_);
        own_stream();
    }
    // `workers` aligners are at work on this device at the same time (cli/align.cpp's thread pool): each sizes its arenas for its share
    void set_device_share(unsigned workers) {
        const std::string opt = "device_share=" + std::to_string(workers ? workers : 1);
        mgx_aligner_set_pipeline(a_, opt.c_str());
    }
    // a result-preserving kernel-selection switch of the library ("key=value", mgx_aligner_set_pipeline): A/B runs
    void set_kernel_option(const std::string &opt) {
        if (int rc = mgx_aligner_set_pipeline(a_, opt.c_str()))
            throw std::runtime_error(std::string(mgx_last_error()) + " (" + std::to_string(rc) + ")");
    }
    ~HipDBGAligner() override { mgx_aligner_destroy(a_); }
    const HipBOSSGraph &get_graph() const override { return graph_; }
    const DBGAlignerConfig &get_config() const override { return config_; }
    bool has_coordinates() const override { return false; }

    void align_batch(const std::vector<Query> &seq_batch, const AlignmentCallback &callback) const override {
        // One call: mgx_align_batch itself re-aligns queries whose per-read device arenas overflowed (MGX_ERR_CAPACITY: the
        // reference has no such limit, its tables grow on the heap) with doubled limits, up to six doublings.  A status that is
        // left after that is an error of this call, as nothing in the reference's interface could carry it.
        std::string blob;
        std::vector<uint64_t> offsets(seq_batch.size() + 1, 0);
        for (size_t t = 0; t < seq_batch.size(); ++t) {
            blob += seq_batch[t].second;
            offsets[t + 1] = blob.size();
        }
        mgx_results res{};
        if (int rc = mgx_align_batch(a_, blob.data(), offsets.data(), seq_batch.size(), 0, &res))
            throw std::runtime_error(std::string("mgx_align_batch: ") + mgx_last_error() + " (" + std::to_string(rc) + ")");
        deliver(res, seq_batch, callback);
    }

    // The two halves for a host that gathers the results of several devices over RCCL (mgx_gather_*, mgx_align --rccl-gather):
    // the kernels of a batch with the results left in HBM — mgx_gather_start(g, handle()) takes them from there — and, on the
    // root, the decoded records of any rank (mgx_results_from_raw) handed to the callback like align_batch does.
    void align_batch_device(const std::vector<Query> &seq_batch) const {
        std::string blob;
        std::vector<uint64_t> offsets(seq_batch.size() + 1, 0);
        for (size_t t = 0; t < seq_batch.size(); ++t) {
            blob += seq_batch[t].second;
            offsets[t + 1] = blob.size();
        }
        if (blob.empty()) blob.push_back('\0');                      // (an empty batch still needs a pointer)
        if (int rc = mgx_align_batch_device(a_, blob.data(), offsets.data(), seq_batch.size(), 0))
            throw std::runtime_error(std::string("mgx_align_batch_device: ") + mgx_last_error() + " (" + std::to_string(rc) + ")");
    }
    mgx_aligner *handle() const { return a_; }
    static void deliver(const mgx_results &res, const std::vector<Query> &seq_batch, const AlignmentCallback &callback) {
        std::vector<AlignmentResults> results;
        results.reserve(seq_batch.size());
        for (const auto &q : seq_batch) results.emplace_back(q.second);
        for (size_t t = 0; t < seq_batch.size(); ++t) {
            if (res.status[t] != MGX_OK)
                throw std::runtime_error("query " + std::to_string(t) + " (" + seq_batch[t].first + "): status "
                                         + std::to_string(res.status[t]) + " (after the capacity retry, where the call has one)");
            AlignmentResults &paths = results[t];
            for (uint64_t ai = res.aln_begin[t]; ai < res.aln_begin[t + 1]; ++ai) {
                const mgx_alignment &m = res.alignments[ai];
                Alignment a;
                a.orientation_ = m.orientation;
                a.offset_ = m.offset;
                a.score_ = m.score;
                a.sequence_.assign(res.seqs + m.seq_begin, m.seq_len);
                a.nodes_.assign(res.nodes + m.nodes_begin, res.nodes + m.nodes_begin + m.n_nodes);
                for (uint32_t x = 0; x < m.n_cigar; ++x)
                    a.cigar_.data().emplace_back(res.cigar[m.cigar_begin + x].op, res.cigar[m.cigar_begin + x].len);
                const std::string &q = paths.get_query(m.orientation);
                a.query_view_ = std::string_view(q).substr(m.clipping, q.size() - m.clipping - m.end_clipping);
                if (res.labels && m.n_labels)
                    a.label_columns_.assign(res.labels + m.labels_begin, res.labels + m.labels_begin + m.n_labels);
                paths.alignments_.push_back(std::move(a));
            }
        }
        for (size_t i = 0; i < seq_batch.size(); ++i) callback(seq_batch[i].first, std::move(results[i]));
    }

  private:
    // the reference builds one aligner per thread-pool task (cli/align.cpp:440-475): every handle works on a stream of its own, so
    // that the tasks of one device overlap instead of taking turns on the default stream (batches arrive from host memory: nothing
    // of the caller's is ordered against it)
    void own_stream() {
        if (getenv("MGX_ADAPTER_DEFAULT_STREAM")) return;          // (A/B switch: every handle on the legacy default stream, as before round 6)
        if (int rc = mgx_aligner_create_stream(a_)) {
            const std::string msg = std::string(mgx_last_error()) + " (" + std::to_string(rc) + ")";
            mgx_aligner_destroy(a_);
            a_ = nullptr;
            throw std::runtime_error(msg);
        }
    }
    const HipBOSSGraph &graph_;
    const HipAnnotation *annotation_ = nullptr;
    DBGAlignerConfig config_;
    mgx_aligner *a_ = nullptr;
};

// format_alignment (cli/align.cpp:254-285), TSV branch; label_names: LabelEncoder::decode for label-aware results (:274-281)
inline std::string format_alignment(const std::string &header, const AlignmentResults &paths, int32_t min_path_score,
                                    const std::vector<std::string> *label_names = nullptr) {
    std::string s = header + "\t" + paths.get_query();
    if (paths.empty()) {
        s += "\t*\t*\t" + std::to_string(min_path_score) + "\t*\t*\t*\n";
    } else {
        for (const auto &p : paths) {
            s += "\t" + p.to_tsv_fields();
            if (p.get_label_columns().size()) {
                s += "\t";
                bool first = true;
                for (uint64_t c : p.get_label_columns()) {
                    if (!first) s += ";";
                    first = false;
                    s += (label_names && c < label_names->size()) ? (*label_names)[c] : std::to_string(c);
                }
            }
        }
        s += "\n";
    }
    return s;
}

} // namespace host
} // namespace mgx
