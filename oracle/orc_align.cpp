// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_align.hpp header).
#include "orc_align.hpp"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <climits>
#include <cmath>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <memory>
#include <numeric>
#include <queue>
#include <stdexcept>
#include <tuple>
#include <unordered_map>
#include <unordered_set>

namespace orc {

typedef ptrdiff_t ssize_t_;

// =============================================================================================
// Cigar (A/aligner_cigar.cpp)
// =============================================================================================
void Cigar::append(uint8_t op, uint32_t num) {
    // aligner_cigar.cpp:98-108
    if (!num) return;
    if (ops.empty() || ops.back().first != op) ops.emplace_back(op, num);
    else ops.back().second += num;
}

void Cigar::append(Cigar &&other) {
    // aligner_cigar.cpp:110-116
    if (other.ops.empty()) return;
    append(other.ops.front().first, other.ops.front().second);
    ops.insert(ops.end(), other.ops.begin() + 1, other.ops.end());
}

uint32_t Cigar::trim_clipping() {
    if (ops.size() && ops.front().first == MGX_OP_CLIPPED) {
        uint32_t r = ops.front().second;
        ops.erase(ops.begin());
        return r;
    }
    return 0;
}

uint32_t Cigar::trim_end_clipping() {
    if (ops.size() && ops.back().first == MGX_OP_CLIPPED) {
        uint32_t r = ops.back().second;
        ops.pop_back();
        return r;
    }
    return 0;
}

void Cigar::extend_clipping(uint32_t n) {
    // aligner_cigar.hpp:75-82
    if (ops.front().first != MGX_OP_CLIPPED) ops.insert(ops.begin(), value_type(MGX_OP_CLIPPED, n));
    else ops.front().second += n;
}

size_t Cigar::get_num_matches() const {
    size_t r = 0;
    for (auto &op : ops) r += (op.first == MGX_OP_MATCH) * op.second;
    return r;
}

std::string Cigar::to_string() const {
    static const char op_str[] = "SX=DIG";   // aligner_cigar.hpp:107
    std::string s;
    for (auto &op : ops) s += std::to_string(op.second) + op_str[op.first];
    return s;
}

bool Cigar::is_valid(std::string_view reference, std::string_view query) const {
    // aligner_cigar.cpp:118-241
    auto ref_it = reference.begin();
    auto alt_it = query.begin();
    for (size_t i = 0; i < ops.size(); ++i) {
        const auto &op = ops[i];
        if (!op.second) return false;
        switch (op.first) {
            case MGX_OP_CLIPPED:
                if ((ref_it != reference.begin() || alt_it != query.begin())
                        && (ref_it != reference.end() || alt_it != query.end())) {
                    if (alt_it > query.end() - op.second) return false;
                    alt_it += op.second;
                }
                break;
            case MGX_OP_MATCH:
            case MGX_OP_MISMATCH:
                if (ref_it > reference.end() - op.second) return false;
                if (alt_it > query.end() - op.second) return false;
                if (std::equal(ref_it, ref_it + op.second, alt_it) == (op.first != MGX_OP_MATCH)) return false;
                ref_it += op.second;
                alt_it += op.second;
                break;
            case MGX_OP_INSERTION:
                if (i && ops[i - 1].first == MGX_OP_DELETION) return false;
                if (alt_it > query.end() - op.second) return false;
                alt_it += op.second;
                break;
            case MGX_OP_DELETION:
                if (i && ops[i - 1].first == MGX_OP_INSERTION) return false;
                if (ref_it > reference.end() - op.second) return false;
                ref_it += op.second;
                break;
            default: break;
        }
    }
    return ref_it == reference.end() && alt_it == query.end();
}

// =============================================================================================
// Config helpers (A/aligner_config.cpp)
// =============================================================================================
score_t score_sequences(const mgx_config &c, std::string_view a, std::string_view b) {
    score_t s = 0;
    for (size_t i = 0; i < a.size(); ++i) s += c.score_matrix[(uint8_t)a[i] & 127][(uint8_t)b[i] & 127];
    return s;
}

bool check_config_scores(const mgx_config &c) {
    // aligner_config.cpp:39-66
    int8_t min_penalty = INT8_MAX;
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) min_penalty = std::min(min_penalty, c.score_matrix[i][j]);
    if (c.gap_opening_penalty * 2 >= min_penalty) return false;
    min_penalty = std::min({ min_penalty, c.gap_opening_penalty, c.gap_extension_penalty });
    return c.min_cell_score >= INT32_MIN - min_penalty;
}

score_t score_cigar(const mgx_config &c, std::string_view reference, std::string_view query, const Cigar &cigar) {
    // aligner_config.cpp:68-126
    if (cigar.ops.empty()) return 0;
    score_t score = (!cigar.get_clipping() ? c.left_end_bonus : 0) + (!cigar.get_end_clipping() ? c.right_end_bonus : 0);
    size_t r = 0, a = 0;
    auto it = cigar.ops.begin();
    if (it->first == MGX_OP_CLIPPED) ++it;
    for (; it != cigar.ops.end(); ++it) {
        const auto &op = *it;
        switch (op.first) {
            case MGX_OP_CLIPPED: if (it + 1 != cigar.ops.end()) a += op.second; break;
            case MGX_OP_MATCH:
                score += match_score(c, reference.substr(r, op.second));
                r += op.second; a += op.second; break;
            case MGX_OP_MISMATCH:
                score += score_sequences(c, reference.substr(r, op.second), query.substr(a, op.second));
                r += op.second; a += op.second; break;
            case MGX_OP_INSERTION:
                score += c.gap_opening_penalty + (op.second - 1) * c.gap_extension_penalty;
                a += op.second; break;
            case MGX_OP_DELETION:
                score += c.gap_opening_penalty + (op.second - 1) * c.gap_extension_penalty;
                r += op.second;
                if (it >= cigar.ops.begin() + 2 && (it - 2)->first == MGX_OP_DELETION
                        && (it - 1)->first == MGX_OP_NODE_INSERTION)
                    score -= c.gap_opening_penalty - c.gap_extension_penalty;
                break;
            case MGX_OP_NODE_INSERTION:
                score += c.gap_opening_penalty + (op.second - 1) * c.gap_extension_penalty; break;
        }
    }
    return score;
}

// =============================================================================================
// Alignment (A/alignment.{hpp,cpp})
// =============================================================================================
Alignment::Alignment(std::string_view query, std::vector<node_t> &&nodes_, std::string &&seq, score_t score_,
                     Cigar &&cigar_, size_t clipping, bool orientation_, size_t offset_)
      : query_view(query), nodes(std::move(nodes_)), orientation(orientation_), offset(offset_),
        sequence(std::move(seq)), score(score_), cigar(MGX_OP_CLIPPED, clipping) {
    cigar.append(std::move(cigar_));       // alignment.hpp:150-152
}

Alignment::Alignment(const Seed &seed, const mgx_config &config)
      : query_view(seed.query_view), nodes(seed.nodes), orientation(seed.orientation), offset(seed.offset),
        sequence(seed.query_view),
        score(match_score(config, seed.query_view) + (!seed.clipping ? config.left_end_bonus : 0)
                + (!seed.end_clipping ? config.right_end_bonus : 0)),
        cigar(MGX_OP_CLIPPED, seed.clipping), label_columns(seed.label_columns) {
    cigar.append(MGX_OP_MATCH, query_view.size());
    cigar.append(MGX_OP_CLIPPED, seed.end_clipping);
}

void Alignment::extend_query_begin(const char *begin) {
    const char *full_query_begin = query_view.data() - get_clipping();
    if (full_query_begin > begin) cigar.extend_clipping(full_query_begin - begin);
}

void Alignment::extend_query_end(const char *end) {
    const char *full_query_end = query_view.data() + query_view.size() + get_end_clipping();
    if (full_query_end < end) cigar.append(MGX_OP_CLIPPED, end - full_query_end);
}

bool Alignment::append(Alignment &&other) {
    // alignment.cpp:94-175; label_coordinates are not restated, label_columns are intersected (:148-160)
    bool ret_val = false;
    if (label_columns.size() && other.label_columns.empty()) label_columns.clear();
    if (label_columns.size()) {
        Columns merged;
        std::set_intersection(label_columns.begin(), label_columns.end(), other.label_columns.begin(), other.label_columns.end(),
                              std::back_inserter(merged));
        if (merged.empty()) { *this = Alignment(); return true; }
        ret_val = merged.size() < label_columns.size();
        std::swap(label_columns, merged);
    }
    nodes.insert(nodes.end(), other.nodes.begin(), other.nodes.end());
    sequence += std::move(other.sequence);
    score += other.score;
    cigar.append(std::move(other.cigar));
    // expand the query window to cover both alignments (:171-173)
    query_view = std::string_view(query_view.data(), (other.query_view.data() + other.query_view.size()) - query_view.data());
    return ret_val;
}

size_t Alignment::trim_query_prefix(size_t n, size_t node_overlap, const mgx_config &config, bool trim_excess_deletions) {
    // alignment.cpp:192-278
    size_t clipping = get_clipping();
    const char *query_begin = query_view.data() - clipping;
    auto it = cigar.ops.begin() + static_cast<bool>(clipping);
    size_t cigar_offset = 0;
    auto s_it = sequence.begin();
    auto node_it = nodes.begin();
    auto consume_ref = [&]() {
        ++s_it;
        if (offset < node_overlap) ++offset;
        else if (node_it + 1 < nodes.end()) ++node_it;
        else *this = Alignment();
    };
    while (n || (trim_excess_deletions && it->first == MGX_OP_DELETION)) {
        if (it == cigar.ops.end()) { *this = Alignment(); return 0; }
        switch (it->first) {
            case MGX_OP_MATCH:
            case MGX_OP_MISMATCH:
                score -= config.score_matrix[(uint8_t)query_view[0] & 127][(uint8_t)*s_it & 127];
                query_view.remove_prefix(1);
                --n;
                consume_ref();
                if (empty()) return 0;
                break;
            case MGX_OP_INSERTION:
                score -= it->second - cigar_offset == 1 ? config.gap_opening_penalty : config.gap_extension_penalty;
                query_view.remove_prefix(1);
                --n;
                break;
            case MGX_OP_DELETION:
                score -= it->second - cigar_offset == 1 ? config.gap_opening_penalty : config.gap_extension_penalty;
                consume_ref();
                if (empty()) return 0;
                break;
            default:
                throw std::runtime_error("trimming chains not supported");       // (assert(false) in the reference, :249-251)
        }
        ++cigar_offset;
        if (cigar_offset == it->second) { ++it; cigar_offset = 0; }
    }
    if (!clipping && it != cigar.ops.begin()) score -= config.left_end_bonus;
    nodes.erase(nodes.begin(), node_it);
    sequence.erase(sequence.begin(), s_it);
    it->second -= cigar_offset;
    cigar.ops.erase(cigar.ops.begin(), it);
    extend_query_begin(query_begin);
    return cigar_offset;
}

void Alignment::insert_gap_prefix(ptrdiff_t gap_length, size_t node_overlap, const mgx_config &config) {
    // alignment.cpp:1154-1234
    size_t extra_nodes = node_overlap + 1;
    if (gap_length < 0) {
        // alignments overlap: extra_nodes = k - 1 - matching_overlap
        trim_clipping();
        extra_nodes += gap_length - 1;
        if (offset) nodes.erase(nodes.begin(), nodes.begin() + offset + gap_length);
        if (extra_nodes) {
            score += config.gap_opening_penalty + (score_t)(extra_nodes - 1) * config.gap_extension_penalty;
            cigar.ops.insert(cigar.ops.begin(), Cigar::value_type(MGX_OP_NODE_INSERTION, (uint32_t)extra_nodes));
        }
    } else {
        // no overlap: extra_nodes = k
        trim_clipping();
        sequence = std::string(1, '$') + sequence;
        cigar.ops.insert(cigar.ops.begin(), Cigar::value_type(MGX_OP_DELETION, 1));
        score += config.gap_opening_penalty;
        if (static_cast<size_t>(gap_length) <= node_overlap) {
            // overlap is small, so add only the required dummy nodes
            trim_offset();
            score += config.gap_opening_penalty + (score_t)(extra_nodes - 2) * config.gap_extension_penalty;
            cigar.ops.insert(cigar.ops.begin(), Cigar::value_type(MGX_OP_NODE_INSERTION, (uint32_t)(extra_nodes - 1)));
        }
        extend_query_begin(query_view.data() - gap_length);
    }
    nodes.insert(nodes.begin(), extra_nodes, 0);
    offset = node_overlap;
}

size_t Alignment::trim_offset() {
    // alignment.cpp:177-190
    if (!offset || nodes.size() <= 1) return 0;
    size_t first_dummy = (std::find(nodes.begin(), nodes.end(), NPOS) - nodes.begin()) - 1;
    size_t trim = std::min(std::min(offset, nodes.size() - 1), first_dummy);
    offset -= trim;
    nodes.erase(nodes.begin(), nodes.begin() + trim);
    return trim;
}

void Alignment::reverse_complement(const GraphView &graph, std::string_view query_rev_comp) {
    // alignment.cpp:540-702
    trim_offset();
    if (graph.rc) {                                    // :547-561 (RCDBG branch)
        if (offset) {
            *this = Alignment();
        } else {
            std::reverse(cigar.ops.begin(), cigar.ops.end());
            std::reverse(nodes.begin(), nodes.end());
            reverse_complement_inplace(sequence);
            orientation = !orientation;
            query_view = query_rev_comp.substr(get_clipping(),
                                               query_rev_comp.size() - get_clipping() - get_end_clipping());
        }
        return;
    }
    // The generic branch (:563-702): the graph holds the reverse complement itself — a CANONICAL-mode DBGSuccinct or a
    // PRIMARY one behind the CanonicalDBG wrapper (graph.canon).
    const Graph &g = *graph.g;
    const CanonicalView *canonical = graph.canon;
    if (g.mode != CANONICAL && !canonical)
        throw std::runtime_error("orc::Alignment::reverse_complement: plain graphs are reversed through the RCDBG view");
    if (!offset) {
        graph.reverse_complement_seq_path(sequence, nodes);
    } else {
        // :566-693: one node, `offset` characters of its k-mer are not part of the alignment
        sequence = graph.get_node_sequence(nodes[0]).substr(0, offset) + sequence;
        if (sequence[0] == '$') {
            // :569-651: a source dummy k-mer: walk forwards (always the last outgoing edge) until the k-mer holds no sentinel
            const Boss &boss = g.boss;
            size_t num_sentinels = sequence.find_last_of('$') + 1;
            if (canonical && nodes[0] != canonical->get_base_node(nodes[0])) { *this = Alignment(); return; }   // :592-597
            size_t num_first_steps = canonical ? std::min(offset, num_sentinels) : offset;
            edge_t edge = nodes[0];
            uint8_t edge_label = boss.get_W(edge) % SIGMA;
            for (size_t i = 0; i < num_first_steps; ++i) {
                edge = boss.fwd(edge, edge_label);
                edge_label = boss.get_W(edge) % SIGMA;
                if (edge_label == 0) { *this = Alignment(); return; }      // reverse complement not found
                nodes[0] = g.validate_edge(edge);
                sequence.push_back(decode_code(edge_label));
            }
            for (size_t i = num_first_steps; i < offset; ++i) {            // :626-645 (CanonicalDBG only)
                node_t next_node = 0;
                char last_char = 0;
                canonical->call_outgoing_kmers(nodes[0], [&](node_t next, char c) {
                    if (c == '$') return;
                    next_node = next;
                    last_char = c;
                });
                if (!next_node) { *this = Alignment(); return; }
                nodes[0] = next_node;
                sequence.push_back(last_char);
            }
            sequence = sequence.substr(offset);
            graph.reverse_complement_seq_path(sequence, nodes);
            sequence.assign(sequence.data() + offset, g.get_k() - offset);
        } else {
            graph.reverse_complement_seq_path(sequence, nodes);
            // :667-689: trim the ending of the reverse complement that corresponds to the added prefix; of several
            // possible predecessors the first one is taken
            for (size_t i = 0; i < offset; ++i) {
                size_t indegree = 0;
                graph.adjacent_incoming_nodes(nodes[0], [&](node_t prev) {
                    ++indegree;
                    if (indegree == 1) nodes[0] = prev;
                });
                if (!indegree) { *this = Alignment(); return; }
                sequence.pop_back();
            }
        }
    }
    std::reverse(cigar.ops.begin(), cigar.ops.end());
    orientation = !orientation;
    query_view = query_rev_comp.substr(get_clipping(), query_rev_comp.size() - get_clipping() - get_end_clipping());
}

std::string spell_path(const GraphView &graph, const std::vector<node_t> &path, size_t offset) {
    // alignment.cpp:1239-1314
    std::string seq;
    if (path.empty()) return seq;
    size_t k = graph.get_k();
    size_t num_dummy = 0, num_unknown = 0;
    if (path.front()) {
        seq += graph.get_node_sequence(path.front()).substr(offset);
    } else {
        num_unknown = k - offset;
        seq += std::string(num_unknown, '$');
        num_dummy = 1;
    }
    auto patch = [&](node_t v) {
        std::string next_seq = graph.get_node_sequence(v);
        auto it = seq.end() - next_seq.size();
        for (char c : next_seq) {
            if (*it == '$' && c != '$') { --num_unknown; *it = c; }
            ++it;
        }
    };
    for (size_t i = 1; i < path.size(); ++i) {
        if (path[i]) {
            if (num_dummy) {
                seq += '$';
                ++num_unknown;
                patch(path[i]);
                num_dummy = 0;
            } else {
                char next = '\0';
                graph.call_outgoing_kmers(path[i - 1], [&](node_t nn, char c) { if (nn == path[i]) next = c; });
                if (!next) throw std::runtime_error("invalid edge");
                seq += next;
                if (num_unknown) patch(path[i]);
            }
        } else {
            seq += '$';
            ++num_dummy;
            ++num_unknown;
        }
    }
    return seq;
}

bool Alignment::is_valid(const GraphView &graph, const mgx_config *config, std::string *why) const {
    // alignment.cpp:1316-1345
    if (empty()) return true;
    try {
        std::string spelling = spell_path(graph, nodes, offset);
        if (spelling != sequence) { if (why) *why = "stored sequence incorrect: " + spelling + " vs " + sequence; return false; }
    } catch (const std::runtime_error &) { if (why) *why = "invalid edge in path"; return false; }
    if (!cigar.is_valid(sequence, query_view)) { if (why) *why = "invalid cigar " + cigar.to_string(); return false; }
    score_t cs = config ? score_cigar(*config, sequence, query_view, cigar) : 0;
    cs += extra_score;
    if (config && score != cs) { if (why) *why = "score mismatch " + std::to_string(score) + " vs cigar " + std::to_string(cs); return false; }
    return true;
}

// =============================================================================================
// sdust — lh3's symmetric DUST (github.com/lh3/sdust, sdust.c), restated.  Third-party code absent
// from the reference tree; called at A/aligner_seeder_methods.cpp:22-29 with T=20, W=64.
// =============================================================================================
namespace {
constexpr int SD_WLEN = 3;
constexpr int SD_WTOT = 1 << (SD_WLEN << 1);
constexpr int SD_WMSK = SD_WTOT - 1;
struct PerfIntv { int start, finish, r, l; };

struct SdustState {
    std::deque<int> w;
    std::vector<PerfIntv> P;   // sorted by descending start, then ascending finish
    std::vector<std::pair<int, int>> res;
};

inline void sd_shift_window(int t, std::deque<int> &w, int T, int W, int *L, int *rw, int *rv, int *cw, int *cv) {
    if ((int)w.size() >= W - SD_WLEN + 1) {
        int s = w.front();
        w.pop_front();
        *rw -= --cw[s];
        if (*L > (int)w.size()) { --*L; *rv -= --cv[s]; }
    }
    w.push_back(t);
    ++*L;
    *rw += cw[t]++;
    *rv += cv[t]++;
    if (cv[t] * 10 > T << 1) {
        int s;
        do {
            s = w[w.size() - *L];
            *rv -= --cv[s];
            --*L;
        } while (s != t);
    }
}

inline void sd_save_masked_regions(SdustState &st, int start) {
    auto &P = st.P;
    if (P.empty() || P.back().start >= start) return;
    const PerfIntv &p = P.back();
    bool saved = false;
    if (!st.res.empty()) {
        int s = st.res.back().first, f = st.res.back().second;
        if (p.start <= f) { saved = true; st.res.back() = { s, f > p.finish ? f : p.finish }; }
    }
    if (!saved) st.res.emplace_back(p.start, p.finish);
    int i;
    for (i = (int)P.size() - 1; i >= 0 && P[i].start < start; --i) {}
    P.resize(i + 1);
}

inline void sd_find_perfect(SdustState &st, int T, int start, int L, int rv, const int *cv) {
    int c[SD_WTOT], r = rv, max_r = 0, max_l = 0;
    std::memcpy(c, cv, sizeof(c));
    auto &w = st.w;
    auto &P = st.P;
    for (int i = (int)w.size() - L - 1; i >= 0; --i) {
        int t = w[i];
        r += c[t]++;
        int new_r = r, new_l = (int)w.size() - i - 1;
        if (new_r * 10 > T * new_l) {
            int j;
            for (j = 0; j < (int)P.size() && P[j].start >= i + start; ++j) {
                const PerfIntv &p = P[j];
                if (max_r == 0 || p.r * max_l > max_r * p.l) { max_r = p.r; max_l = p.l; }
            }
            if (max_r == 0 || new_r * max_l >= max_r * new_l) {
                max_r = new_r; max_l = new_l;
                PerfIntv np{ i + start, (int)w.size() + (SD_WLEN - 1) + start, new_r, new_l };
                P.insert(P.begin() + j, np);
            }
        }
    }
}
} // namespace

bool is_low_complexity(std::string_view s, int T, int W) {
    SdustState st;
    int rv = 0, rw = 0, L = 0, cv[SD_WTOT], cw[SD_WTOT];
    std::memset(cv, 0, sizeof(cv));
    std::memset(cw, 0, sizeof(cw));
    int l_seq = (int)s.size();
    int l = 0;
    unsigned t = 0;
    for (int i = 0; i <= l_seq; ++i) {
        int b = 4;
        if (i < l_seq) {
            switch (s[i]) {     // seq_nt4_table
                case 'A': case 'a': b = 0; break;
                case 'C': case 'c': b = 1; break;
                case 'G': case 'g': b = 2; break;
                case 'T': case 't': case 'U': case 'u': b = 3; break;
                default: b = 4;
            }
        }
        if (b < 4) {
            ++l;
            t = (t << 2 | b) & SD_WMSK;
            if (l >= SD_WLEN) {
                int start = (l - W > 0 ? l - W : 0) + (i + 1 - l);
                sd_save_masked_regions(st, start);
                sd_shift_window((int)t, st.w, T, W, &L, &rw, &rv, cw, cv);
                if (rw * 10 > L * T) sd_find_perfect(st, T, start, L, rv, cv);
            }
        } else {
            int start = (l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l);
            while (!st.P.empty()) sd_save_masked_regions(st, start++);
            l = 0; t = 0;
        }
    }
    return !st.res.empty();
}

// =============================================================================================
// Seeders (A/aligner_seeder_methods.{hpp,cpp})
// =============================================================================================
namespace {

struct SeederState {
    std::vector<Seed> seeds;
    size_t num_matching = 0;
    bool has_seeds_fn = true;      // false for the empty ManualMatchingSeeder replacement
};

size_t num_exact_matching(const std::vector<node_t> &query_nodes, size_t k) {
    // aligner_seeder_methods.cpp:49-65
    size_t num_matching = 0, last_match_count = 0;
    for (auto it = query_nodes.begin(); it != query_nodes.end(); ++it) {
        if (*it) {
            auto jt = std::find(it + 1, query_nodes.end(), NPOS);
            num_matching += k + std::distance(it, jt) - 1 - last_match_count;
            last_match_count = k;
            it = jt - 1;
        } else if (last_match_count) {
            --last_match_count;
        }
    }
    return num_matching;
}

std::vector<Seed> exact_get_seeds(const Graph &graph, std::string_view query, bool orientation,
                                  const std::vector<node_t> &query_nodes, const mgx_config &config,
                                  size_t num_matching) {
    // aligner_seeder_methods.cpp:67-93
    size_t k = graph.get_k();
    if (num_matching < config.min_exact_match * query.size()) return {};
    std::vector<Seed> seeds;
    if (config.max_seed_length < k) return seeds;
    size_t end_clipping = query.size() - k;
    for (size_t i = 0; i < query_nodes.size(); ++i, --end_clipping) {
        if (query_nodes[i] != NPOS) {
            std::string_view window = query.substr(i, k);
            if (!config.seed_complexity_filter || !is_low_complexity(window)) {
                Seed s;
                s.query_view = window; s.nodes = { query_nodes[i] }; s.orientation = orientation;
                s.offset = 0; s.clipping = i; s.end_clipping = end_clipping;
                seeds.push_back(std::move(s));
            }
        }
    }
    return seeds;
}

std::vector<Seed> mem_get_seeds(const GraphView &graph, std::string_view query, bool orientation,
                                const std::vector<node_t> &query_nodes, const mgx_config &config,
                                size_t num_matching, WorkCounters *wc) {
    // aligner_seeder_methods.cpp:360-424 with the UniMEMSeeder terminator (hpp:116-135)
    size_t k = graph.get_k();
    if (k >= config.max_seed_length)
        return exact_get_seeds(*graph.g, query, orientation, query_nodes, config, num_matching);
    if (num_matching < config.min_exact_match * query.size()) return {};

    std::vector<uint8_t> flags(query_nodes.size(), 0);
    for (size_t i = 0; i < flags.size(); ++i) {
        if (query_nodes[i] != NPOS) {
            bool term = i + 1 == query_nodes.size() || query_nodes[i + 1] == NPOS;
            if (!term) {
                if (wc) ++wc->n_terminus;
                term = graph.has_multiple_outgoing(query_nodes[i]) || !graph.has_single_incoming(query_nodes[i]);
            }
            flags[i] = 2 | (term ? 1 : 0);
        }
    }
    std::vector<Seed> seeds;
    auto it = flags.begin();
    while ((it = std::find_if(it, flags.end(), [](uint8_t f) { return f & 2; })) != flags.end()) {
        auto next = std::find_if(it, flags.end(), [](uint8_t f) { return (f & 1) == 1 || (f & 2) == 0; });
        if (next != flags.end() && ((*next) & 2)) ++next;
        size_t i = it - flags.begin();
        size_t mem_length = (next - it) + k - 1;
        if (mem_length >= config.min_seed_length) {
            Seed s;
            s.query_view = query.substr(i, mem_length);
            s.nodes.assign(query_nodes.begin() + i, query_nodes.begin() + i + (next - it));
            s.orientation = orientation; s.offset = 0; s.clipping = i;
            s.end_clipping = query.size() - i - mem_length;
            seeds.push_back(std::move(s));
        }
        it = next;
    }
    return seeds;
}

// SuffixSeeder<UniMEMSeeder>(graph, query, orientation, nodes, config): ctor + generate_seeds()
SeederState make_suffix_seeder(const GraphView &view, std::string_view query, bool orientation,
                               const std::vector<node_t> &query_nodes, const mgx_config &config,
                               WorkCounters *wc) {
    const Graph &graph = *view.g;                    // get_base_dbg_succ (:141-151)
    SeederState st;
    size_t k = graph.get_k();
    st.num_matching = num_exact_matching(query_nodes, k);           // ExactSeeder ctor (:37-47)

    // generate_seeds (:153-358)
    if (query.size() < config.min_seed_length) return st;
    if (config.min_seed_length >= k) {
        st.seeds = mem_get_seeds(view, query, orientation, query_nodes, config, st.num_matching, wc);
        return st;
    }

    size_t nslots = query.size() - config.min_seed_length + 1;
    std::vector<std::vector<Seed>> suffix_seeds(nslots);
    std::vector<size_t> min_seed_length(nslots, config.min_seed_length);

    for (auto &&seed : mem_get_seeds(view, query, orientation, query_nodes, config, st.num_matching, wc)) {
        size_t i = seed.clipping;
        for (size_t j = 0; j < seed.nodes.size(); ++j) min_seed_length[i + j] = k;
        if (i + seed.nodes.size() < min_seed_length.size()) min_seed_length[i + seed.nodes.size()] = k;
        suffix_seeds[i].emplace_back(std::move(seed));
    }

    auto append_suffix_seed = [&](size_t i, node_t alt_node, size_t seed_length) {
        // :195-213
        std::string_view seed_seq = query.substr(i, seed_length);
        if (seed_length > min_seed_length[i]) suffix_seeds[i].clear();
        min_seed_length[i] = seed_length;
        Seed s;
        s.query_view = seed_seq; s.nodes = { alt_node }; s.orientation = orientation;
        s.offset = k - seed_length; s.clipping = i; s.end_clipping = query.size() - i - seed_seq.size();
        suffix_seeds[i].push_back(std::move(s));
        for (++i; i < min_seed_length.size() && seed_length > min_seed_length[i]; ++i) {
            min_seed_length[i] = seed_length--;
            suffix_seeds[i].clear();
        }
    };

    size_t last_full_id = query.size() >= k ? query.size() - k + 1 : min_seed_length.size();
    for (size_t i = 0; i < min_seed_length.size(); ++i) {
        size_t max_seed_length = std::min({ (size_t)config.max_seed_length, k - 1, query.size() - i });
        size_t seed_length = 0;
        std::vector<node_t> alt_nodes;
        if (config.seed_complexity_filter && is_low_complexity(query.substr(i, min_seed_length[i]))) continue;
        if (wc && max_seed_length >= min_seed_length[i]) wc->n_index_steps += max_seed_length;
        graph.call_nodes_with_suffix_matching_longest_prefix(
            query.substr(i, max_seed_length),
            [&](node_t alt_node, uint64_t len) { seed_length = len; alt_nodes.push_back(alt_node); },
            min_seed_length[i]);
        if (i >= last_full_id && alt_nodes.size() == 1
                && min_seed_length[last_full_id - 1] == k
                && suffix_seeds[last_full_id - 1].size() == 1
                && alt_nodes[0] == suffix_seeds[last_full_id - 1][0].nodes[0])
            continue;
        for (node_t alt_node : alt_nodes) append_suffix_seed(i, alt_node, seed_length);
    }

    if (const CanonicalView *canonical = view.canon) {
        // :251-314: sub-k matches in the reverse complement.  Matching is query prefix -> node suffix, so the query index j of
        // a match to the reverse complement follows from the match length.
        std::string query_rc(query);
        reverse_complement_inplace(query_rc);
        const Boss &boss = graph.boss;
        for (size_t i = 0; i + config.min_seed_length <= query_rc.size(); ++i) {
            size_t max_seed_length = std::min({ (size_t)config.max_seed_length, k - 1, query.size() - i });
            size_t j_min = query_rc.size() - i - max_seed_length;
            size_t j_max = query_rc.size() - i - config.min_seed_length;
            while (j_min <= j_max && min_seed_length[j_min] > max_seed_length) { ++j_min; --max_seed_length; }
            if (j_min > j_max) continue;
            auto encoded = encode_seq(std::string_view(query_rc.data() + i, max_seed_length));
            auto [first, last, seed_length] = boss.index_range(encoded.data(), encoded.data() + encoded.size());
            size_t j = query_rc.size() - i - seed_length;
            if (seed_length < config.min_seed_length || seed_length < min_seed_length[j]
                    || (config.seed_complexity_filter && is_low_complexity(query.substr(j, seed_length))))
                continue;
            // suffix_to_prefix (:95-139): matched ***ATG, want ATG***: every node whose PREFIX is the match
            using Range = std::tuple<edge_t, edge_t, size_t>;
            auto call_nodes_in_range = [&](const Range &r) {
                for (edge_t e = std::get<0>(r); e <= std::get<1>(r); ++e) {
                    node_t node = graph.validate_edge(e);
                    if (node) append_suffix_seed(j, canonical->reverse_complement(node), seed_length);
                }
            };
            Range start(boss.pred_last(first - 1) + 1, last, seed_length);
            if (seed_length == boss.k_) { call_nodes_in_range(start); continue; }
            std::vector<Range> stack{ start };
            while (!stack.empty()) {
                Range cur = stack.back();
                stack.pop_back();
                ++std::get<2>(cur);
                for (uint8_t c = 1; c < SIGMA; ++c) {
                    Range next = cur;
                    if (boss.tighten_range(&std::get<0>(next), &std::get<1>(next), c)) {
                        if (std::get<2>(next) == boss.k_) call_nodes_in_range(next);
                        else stack.push_back(next);
                    }
                }
            }
        }
    }

    // aggregate (:316-357)
    st.seeds.clear();
    st.num_matching = 0;
    size_t last_end = 0;
    for (size_t i = 0; i < suffix_seeds.size(); ++i) {
        auto &pos_seeds = suffix_seeds[i];
        if (pos_seeds.empty()) continue;
        bool full = !pos_seeds[0].offset;
        size_t n_pos = pos_seeds.size();
        if (full) {
            st.seeds.emplace_back(std::move(pos_seeds[0]));
        } else if (n_pos <= config.max_num_seeds_per_locus) {
            for (auto &&s : pos_seeds) st.seeds.emplace_back(std::move(s));
        }
        if (full || n_pos <= config.max_num_seeds_per_locus) {
            size_t begin = st.seeds.back().clipping;
            size_t end = begin + st.seeds.back().query_view.size();
            if (begin < last_end) st.num_matching += end - begin - (last_end - begin);
            else st.num_matching += end - begin;
            last_end = end;
        }
    }
    return st;
}

// ISeeder::get_alignments (aligner_seeder_methods.hpp:20-29)
std::vector<Alignment> seeds_to_alignments(const std::vector<Seed> &seeds, const mgx_config &config) {
    std::vector<Alignment> alns;
    alns.reserve(seeds.size());
    for (const Seed &s : seeds) {
        alns.emplace_back(s, config);
        alns.back().trim_offset();
    }
    return alns;
}

// =============================================================================================
// Extender (A/aligner_extender_methods.{hpp,cpp})
// =============================================================================================
constexpr size_t kPadding = 5;                          // extender hpp:107
std::atomic<uint64_t> g_oob_reads{0};
std::atomic<uint64_t> g_unfetched_label_lookups{0};

// std::vector<score_t> with the capacity behaviour the reference relies on (padding reads/writes
// past size()).  Growth mirrors libstdc++ (capacity doubles on overflow).
struct PVec {
    std::vector<score_t> buf;   // buf.size() == capacity
    size_t sz = 0;
    size_t size() const { return sz; }
    size_t capacity() const { return buf.size(); }
    score_t *data() { return buf.data(); }
    const score_t *data() const { return buf.data(); }
    score_t &operator[](size_t i) { return buf[i]; }
    const score_t &operator[](size_t i) const { return buf[i]; }
    score_t &back() { return buf[sz - 1]; }
    void create(size_t size) { buf.assign(size + kPadding, NINF); sz = size; }   // DPTColumn::create (:389-410)
    void push_back(score_t v) {
        if (sz == buf.size()) buf.resize(std::max<size_t>(1, 2 * sz), NINF);
        buf[sz++] = v;
    }
    void reserve(size_t n) { if (n > buf.size()) buf.resize(n, NINF); }
    // reads past capacity() are undefined behaviour in the reference; counted and read as NINF here
    score_t at_ub(size_t i) const { if (i >= buf.size()) { ++g_oob_reads; return NINF; } return buf[i]; }
    void fill_padding() { std::fill(buf.begin() + sz, buf.end(), NINF); }
};

struct Column {                                          // DPTColumn, extender hpp:129-147
    PVec S, E, F;
    node_t node;
    size_t parent_i;
    char c;
    ssize_t_ offset, max_pos, trim;
    size_t xdrop_cutoff_i;
    score_t score;
};
constexpr size_t kSizeofColumn = 136;                    // sizeof(DPTColumn) on LP64 (3 vectors + 8 fields)

void update_column(size_t prev_end, const score_t *S_prev_v, const score_t *F_prev_v,
                   PVec &S_v, PVec &E_v, PVec &F_v, const score_t *profile_scores,
                   score_t xdrop_cutoff, const mgx_config &config, score_t init_score, size_t offset) {
    // aligner_extender_methods.cpp:209-290, restated lane-exactly in blocks of 4
    constexpr size_t width = kPadding - 1;
    const score_t go = config.gap_opening_penalty, ge = config.gap_extension_penalty;
    for (size_t j = 0; j < prev_end; j += width) {
        score_t match[4], del[4];
        for (size_t l = 0; l < 4; ++l) {
            if (j) {
                match[l] = S_prev_v[j - 1 + l] + profile_scores[j + l] + init_score;
            } else {
                // shuffle 0b10010000 of S_prev[0..3] -> (S0,S0,S1,S2); lane 0 then replaced by ninf
                score_t sp = (l == 0) ? S_prev_v[0] : S_prev_v[l - 1];
                match[l] = sp + profile_scores[l] + init_score;
                if (l == 0) match[l] = NINF;
            }
            if (offset > 1) del[l] = std::max(S_prev_v[j + l] + go, F_prev_v[j + l] + ge) + init_score;
            else del[l] = NINF;
            F_v[j + l] = del[l];
            match[l] = std::max(match[l], del[l]);
            E_v[j + 1 + l] = match[l] + go;
        }
        E_v[j + 1] = std::max(E_v[j] + ge, E_v[j + 1]);
        E_v[j + 2] = std::max(E_v[j + 1] + ge, E_v[j + 2]);
        E_v[j + 3] = std::max(E_v[j + 2] + ge, E_v[j + 3]);
        E_v[j + 4] = std::max(E_v[j + 3] + ge, E_v[j + 4]);
        for (size_t l = 0; l < 4; ++l) {
            score_t m = std::max(match[l], E_v[j + l]);
            S_v[j + l] = m > xdrop_cutoff - 1 ? m : NINF;
        }
    }
    if (S_v.size() > std::max<size_t>(1, prev_end)) {
        size_t j = S_v.size() - 1;
        score_t match = std::max(S_prev_v[j - 1] + init_score + profile_scores[j], E_v[j]);
        if (match >= xdrop_cutoff) S_v[j] = match;
    }
}

void extend_ins_end(PVec &S, PVec &E, PVec &F, size_t max_size, score_t xdrop_cutoff, const mgx_config &config) {
    // aligner_extender_methods.cpp:293-328
    if (S.size() < max_size) {
        score_t ins_score = std::max(S.back() + config.gap_opening_penalty, E.back() + config.gap_extension_penalty);
        if (ins_score >= xdrop_cutoff) {
            S.push_back(ins_score);
            E.push_back(ins_score);
            F.push_back(NINF);
            while (E.back() + config.gap_extension_penalty >= xdrop_cutoff && E.size() < max_size) {
                E.push_back(E.back() + config.gap_extension_penalty);
                S.push_back(E.back());
                F.push_back(NINF);
            }
            S.reserve(S.size() + kPadding);
            E.reserve(E.size() + kPadding);
            F.reserve(F.size() + kPadding);
            S.fill_padding();
            E.fill_padding();
            F.fill_padding();
        }
    }
}

// =============================================================================================
// AnnotationBuffer (A/annotation_buffer.{hpp,cpp}), annotation without coordinates
// =============================================================================================
constexpr Label kNannot = std::numeric_limits<Label>::max();      // aligner_labeled.cpp:20 ("dummy index for unfetched annotations")

class AnnotationBuffer {
  public:
    // graph: the graph the aligner runs on — a BASIC DBGSuccinct, or a PRIMARY one seen through the CanonicalDBG wrapper
    // (`canon`); a CANONICAL-mode DBGSuccinct would take the spell_path + map_to_nodes branch (:60-63), not restated
    AnnotationBuffer(const Graph &graph, const CanonicalView *canon, const Annotation &annotation)
          : graph_(graph), canonical_(canon), annotation_(annotation) {
        column_sets_.push_back(Columns{});               // "the first element is the empty label set" (hpp:76-78)
        column_index_[Columns{}] = 0;
    }

    void queue_path(std::vector<node_t> &&path) { queued_paths_.push_back(std::move(path)); }     // hpp:29-31

    // annotation_buffer.cpp:34-193
    void fetch_queued_annotations() {
        std::vector<node_t> queued_nodes;
        std::vector<uint64_t> queued_rows;
        for (const auto &path : queued_paths_) {
            std::vector<node_t> base_path;
            if (graph_.mode == CANONICAL) {
                // "TODO: avoid this call of spell_path" (:58-62): the path spelled and mapped again — on a CANONICAL-mode graph
                // map_to_nodes gives the k-mer's representative, the smaller BOSS index of the k-mer and its reverse complement
                for (node_t node : path) {
                    std::vector<node_t> m = node ? graph_.map_to_nodes(graph_.get_node_sequence(node)) : std::vector<node_t>{};
                    base_path.push_back(m.size() == 1 ? m[0] : NPOS);
                }
            } else if (canonical_) {
                base_path.reserve(path.size());
                for (node_t node : path) base_path.emplace_back(canonical_->get_base_node(node));
            } else {
                base_path = path;                        // BASIC (the buffer's graph is never an RCDBG view)
            }
            for (size_t i = 0; i < path.size(); ++i) {
                if (base_path[i] == NPOS) { node_to_cols_.try_emplace(path[i], 0); continue; }
                if (!graph_.boss.get_W(base_path[i])) {  // "skip dummy nodes"
                    node_to_cols_.try_emplace(base_path[i], 0);
                    if (graph_.mode == CANONICAL && base_path[i] != path[i]) node_to_cols_.emplace(path[i], 0);
                    continue;
                }
                uint64_t row = base_path[i] - 1;         // AnnotatedDBG::graph_to_anno_index
                if (graph_.mode != CANONICAL) {
                    if (node_to_cols_.try_emplace(base_path[i], kNannotIdx).second) {
                        queued_rows.push_back(row);
                        queued_nodes.push_back(base_path[i]);
                    }
                    continue;
                }
                // CANONICAL (:96-135): a node and its representative share one label set; whichever of the two is known
                // first gives it to the other
                auto find_a = node_to_cols_.find(path[i]);
                auto find_b = node_to_cols_.find(base_path[i]);
                if (find_a == node_to_cols_.end() && find_b == node_to_cols_.end()) {
                    node_to_cols_.try_emplace(path[i], kNannotIdx);
                    queued_rows.push_back(row);
                    queued_nodes.push_back(path[i]);
                    if (path[i] != base_path[i]) {
                        node_to_cols_.emplace(base_path[i], kNannotIdx);
                        queued_rows.push_back(row);
                        queued_nodes.push_back(base_path[i]);
                    }
                } else if (find_a == node_to_cols_.end()) {
                    node_to_cols_.try_emplace(path[i], find_b->second);
                    if (find_b->second == kNannotIdx) { queued_rows.push_back(row); queued_nodes.push_back(path[i]); }
                } else if (find_b == node_to_cols_.end()) {
                    node_to_cols_.try_emplace(base_path[i], find_a->second);
                } else {
                    size_t label_i = std::min(find_a->second, find_b->second);
                    if (label_i != kNannotIdx) { find_a->second = label_i; find_b->second = label_i; }
                }
            }
        }
        queued_paths_.clear();
        if (queued_nodes.empty()) return;
        std::vector<Columns> rows = annotation_.get_rows(queued_rows);
        for (size_t x = 0; x < rows.size(); ++x) {
            std::sort(rows[x].begin(), rows[x].end());
            size_t label_i = cache_column_set(std::move(rows[x]));
            node_to_cols_[queued_nodes[x]] = label_i;    // push_node_labels: BASIC and the canonical wrapper both key by the base node
            if (graph_.mode == CANONICAL && !canonical_) {
                // (:150-158) a CANONICAL-mode graph: the queued node, and its representative if that has no entry yet
                const node_t base_node = (node_t)(queued_rows[x] + 1);
                if (base_node != queued_nodes[x]) node_to_cols_.try_emplace(base_node, label_i);
            }
        }
    }

    // get_labels_and_coords().first (:195-217)
    const Columns *get_labels(node_t node) const {
        if (canonical_) node = canonical_->get_base_node(node);       // (CANONICAL-mode graphs: every queued node has an entry of its own)
        auto it = node_to_cols_.find(node);
        if (it == node_to_cols_.end() || it->second == kNannotIdx) {
            // On a CANONICAL-mode graph the reference can ask for a node it never queued: the nodes of a reversed alignment that
            // seeds the backward pass (replayed "in_seed", aligner_labeled.cpp:186-195) are the reverse complements of the forward
            // pass's nodes, and only the one of each pair that IS the representative got an entry (annotation_buffer.cpp:96-135).
            // The reference asserts (aligner_labeled.cpp:143-146, :110) and is undefined in a release build.  Defined here — and on
            // the device — as what the buffer would have answered had the node been queued: the labels of its representative;
            // counted (orc_unfetched_label_lookups), like the out-of-bounds reads of backtrack.
            if (graph_.mode != CANONICAL || canonical_ || !node) return nullptr;
            ++g_unfetched_label_lookups;
            auto *self = const_cast<AnnotationBuffer *>(this);
            std::vector<node_t> m = graph_.map_to_nodes(graph_.get_node_sequence(node));
            const node_t base = m.size() == 1 ? m[0] : NPOS;
            size_t label_i = 0;
            if (base != NPOS && graph_.boss.get_W(base)) {
                std::vector<Columns> rows = annotation_.get_rows({ base - 1 });
                std::sort(rows[0].begin(), rows[0].end());
                label_i = self->cache_column_set(std::move(rows[0]));
            }
            self->node_to_cols_[node] = label_i;
            return &column_sets_[label_i];
        }
        return &column_sets_[it->second];
    }

    // cache_column_set (hpp:55-60): VectorSet::emplace — the index of the equal set already stored, or of a new entry
    size_t cache_column_set(Columns &&cols) {
        auto it = column_index_.find(cols);
        if (it != column_index_.end()) return it->second;
        size_t idx = column_sets_.size();
        column_index_.emplace(cols, idx);
        column_sets_.push_back(std::move(cols));
        return idx;
    }
    const Columns &get_cached_column_set(size_t i) const { return column_sets_.at(i); }
    size_t num_nodes_buffered() const { return node_to_cols_.size(); }

  private:
    static constexpr size_t kNannotIdx = std::numeric_limits<size_t>::max();
    const Graph &graph_;
    const CanonicalView *canonical_;
    const Annotation &annotation_;
    std::deque<Columns> column_sets_;                    // (deque: references handed out stay valid while sets are added)
    std::map<Columns, size_t> column_index_;
    std::unordered_map<node_t, size_t> node_to_cols_;
    std::vector<std::vector<node_t>> queued_paths_;
};

class Extender {
  public:
    Extender(const Graph &graph, const mgx_config &config, std::string_view query, WorkCounters *wc,
             const CanonicalView *canon = nullptr, AnnotationBuffer *annotation_buffer = nullptr)
          : base_(&graph), config_(config), query_(query), wc_(wc), ab_(annotation_buffer) {
        view_.g = &graph;
        view_.rc = false;
        view_.canon = canon;
        // aligner_extender_methods.cpp:22-60
        partial_sums_.assign(query_.size(), 0);
        for (size_t i = 0; i < query_.size(); ++i) partial_sums_[i] = sm(query_[i], query_[i]);
        std::partial_sum(partial_sums_.rbegin(), partial_sums_.rend(), partial_sums_.rbegin());
        partial_sums_.push_back(0);
        for (int i = 0; i < 6; ++i) {
            profile_score_[i].assign(query_.size() + kPadding, 0);
            profile_op_[i].assign(query_.size() + kPadding, MGX_OP_CLIPPED);
            char c = i != 5 ? decode_code(i) : '\0';
            for (size_t j = 0; j < query_.size(); ++j) {
                profile_score_[i][j + 1] = sm(c, query_[j]);
                profile_op_[i][j + 1] = char_to_op(c, query_[j]);
            }
        }
    }

    void set_graph(bool rc) { view_.rc = rc; }
    const GraphView &view() const { return view_; }
    size_t num_extensions() const { return num_extensions_; }
    size_t num_explored_nodes() const { return explored_nodes_previous_ + conv_checker_.size(); }

    // SeedFilteringExtender::get_extensions (extender hpp:38-51)
    std::vector<Alignment> get_extensions(const Alignment &seed, score_t min_path_score, bool force_fixed_seed) {
        seed_ = &seed;                                   // set_seed (:90-98)
        explored_nodes_previous_ += conv_checker_.size();
        conv_checker_.clear();
        if (ab_) {
            // LabeledExtender::set_seed (aligner_labeled.cpp:139-174, no coordinates): the first node of the seed has
            // already been flushed; the seed's labels are what backtracking still has to account for
            last_flushed_table_i_ = 1;
            remaining_labels_i_ = ab_->cache_column_set(Columns(seed.label_columns));
            node_labels_.assign(1, remaining_labels_i_);
        }
        return extend(min_path_score, force_fixed_seed);
    }

    bool check_seed(const Alignment &seed) const {
        // aligner_extender_methods.cpp:66-88
        if (seed.empty()) return false;
        node_t node = seed.nodes.back();
        if (view_.rc) node += view_.max_index();
        auto it = conv_checker_.find(node);
        if (it == conv_checker_.end()) return true;
        size_t pos = seed.query_view.size() + seed.get_clipping() - 1;
        const auto &[start, vec] = it->second;
        return pos < start || pos - start >= vec.size() || vec[pos - start] < seed.score;
    }

    bool filter_nodes(node_t node, size_t query_start, size_t query_end) {
        // aligner_extender_methods.cpp:158-207 (note: no RCDBG offset is applied to node here)
        constexpr score_t mscore = -NINF;
        size_t size = query_end - query_start;
        auto it = conv_checker_.find(node);
        if (it == conv_checker_.end()) {
            conv_checker_.emplace(node, std::make_pair(query_start, std::vector<score_t>(size, mscore)));
            return true;
        }
        auto &[start, vec] = it->second;
        if (query_start + size <= start) {
            vec.insert(vec.begin(), start - query_start, NINF);
            std::fill(vec.begin(), vec.begin() + size, mscore);
            start = query_start;
            return true;
        }
        if (query_start >= start + vec.size()) {
            vec.insert(vec.end(), query_start - start - vec.size(), NINF);
            vec.insert(vec.end(), size, mscore);
            return true;
        }
        if (query_start < start) {
            vec.insert(vec.begin(), start - query_start, NINF);
            start = query_start;
        }
        if (query_start + size > start + vec.size()) vec.resize(query_start + size - start, NINF);
        bool converged = true;
        score_t *v = vec.data() + query_start - start;
        for (size_t j = 0; j < size; ++j) {
            if (mscore > v[j]) { converged = false; v[j] = mscore; }
        }
        return !converged;
    }

  private:
    const Graph *base_;
    GraphView view_;
    const mgx_config &config_;
    std::string_view query_;
    WorkCounters *wc_;
    const Alignment *seed_ = nullptr;
    // LabeledExtender (aligner_labeled.hpp:86-106); ab_ == nullptr: DefaultColumnExtender
    AnnotationBuffer *ab_ = nullptr;
    size_t last_flushed_table_i_ = 0;
    std::vector<size_t> node_labels_;
    size_t remaining_labels_i_ = 0;
    Columns label_intersection_, label_diff_;
    std::vector<score_t> partial_sums_;
    std::vector<score_t> profile_score_[6];
    std::vector<uint8_t> profile_op_[6];
    std::vector<Column> table;
    size_t table_cap_ = 0;
    size_t table_size_bytes_ = 0;
    std::unordered_set<size_t> prev_starts;
    std::vector<std::pair<size_t, score_t>> xdrop_cutoffs_;
    size_t num_extensions_ = 0;
    score_t min_cell_score_ = 0;
    typedef std::pair<size_t, std::vector<score_t>> ScoreVec;
    std::unordered_map<node_t, ScoreVec> conv_checker_;
    size_t explored_nodes_previous_ = 0;

    score_t sm(char a, char b) const { return config_.score_matrix[(uint8_t)a & 127][(uint8_t)b & 127]; }
    static uint8_t char_to_op(char a, char b) {
        // initialize_opt_table, aligner_cigar.cpp:10-51: MATCH iff same letter of "ACGT", case-insensitively
        auto up = [](char x) { return (char)toupper((unsigned char)x); };
        char ua = up(a), ub = up(b);
        bool valid = (ua == 'A' || ua == 'C' || ua == 'G' || ua == 'T');
        return (valid && ua == ub) ? MGX_OP_MATCH : MGX_OP_MISMATCH;
    }

    // DefaultColumnExtender::pop (hpp:190) / LabeledExtender::pop (aligner_labeled.hpp:76-81)
    void pop(size_t i) {
        table.erase(table.begin() + i);
        if (ab_) {
            last_flushed_table_i_ = std::min(i, last_flushed_table_i_);
            node_labels_.erase(node_labels_.begin() + i);
        }
    }

    // LabeledExtender::flush (aligner_labeled.cpp:81-137)
    void flush() {
        ab_->fetch_queued_annotations();
        for ( ; last_flushed_table_i_ < table.size(); ++last_flushed_table_i_) {
            Column &table_elem = table[last_flushed_table_i_];
            size_t parent_i = table_elem.parent_i;
            auto clear = [&]() {
                node_labels_[last_flushed_table_i_] = 0;
                std::fill(table_elem.S.buf.begin(), table_elem.S.buf.begin() + table_elem.S.sz, NINF);
                std::fill(table_elem.E.buf.begin(), table_elem.E.buf.begin() + table_elem.E.sz, NINF);
                std::fill(table_elem.F.buf.begin(), table_elem.F.buf.begin() + table_elem.F.sz, NINF);
            };
            if (!node_labels_[parent_i]) { clear(); continue; }
            if (table_elem.node == NPOS) continue;
            const Columns &parent_labels = ab_->get_cached_column_set(node_labels_[parent_i]);
            const Columns *cur_labels = ab_->get_labels(table_elem.node);
            if (!cur_labels) throw std::logic_error("oracle: flush(): labels of a table node were not fetched");
            Columns intersect_labels;
            std::set_intersection(parent_labels.begin(), parent_labels.end(), cur_labels->begin(), cur_labels->end(),
                                  std::back_inserter(intersect_labels));
            if (intersect_labels.empty()) clear();
            else node_labels_[last_flushed_table_i_] = ab_->cache_column_set(std::move(intersect_labels));
        }
    }

    // LabeledExtender::call_outgoing (aligner_labeled.cpp:176-302, the branch without coordinates)
    void call_outgoing_labeled(node_t node, const std::function<void(node_t, char, score_t)> &callback,
                               size_t table_i, bool force_fixed_seed) {
        size_t next_offset = table[table_i].offset + 1;
        bool in_seed = next_offset - seed_->offset < seed_->sequence.size()
                        && (next_offset < view_.get_k() || force_fixed_seed);
        std::vector<std::tuple<node_t, char, score_t>> outgoing;
        call_outgoing(node, [&](node_t next, char c, score_t score) {
            outgoing.emplace_back(next, c, score);
            if (!in_seed) ab_->queue_path({ next });
        }, table_i, force_fixed_seed);
        if (outgoing.empty()) return;
        if (outgoing.size() == 1) {
            // "Assume that annotations are preserved in unitigs. Violations of this assumption are corrected after the
            // next flush"
            const auto &[next, c, score] = outgoing[0];
            node_labels_.emplace_back(node_labels_[table_i]);
            callback(next, c, score);
            return;
        }
        flush();
        if (!node_labels_[table_i]) return;
        const Columns columns = ab_->get_cached_column_set(node_labels_[table_i]);     // (copy: the set may grow below)
        for (const auto &[next, c, score] : outgoing) {
            const Columns *next_labels = ab_->get_labels(next);
            if (!next_labels) throw std::logic_error("oracle: call_outgoing(): labels of a child were not fetched");
            Columns intersect_labels;
            std::set_intersection(columns.begin(), columns.end(), next_labels->begin(), next_labels->end(),
                                  std::back_inserter(intersect_labels));
            if (intersect_labels.size()) {
                node_labels_.push_back(ab_->cache_column_set(std::move(intersect_labels)));
                callback(next, c, score);
            }
        }
    }

    // terminate_backtrack_start (hpp:178-180 / aligner_labeled.hpp:49-52)
    bool terminate_backtrack_start(const std::vector<Alignment> &extensions) const {
        return ab_ ? !remaining_labels_i_ : extensions.size() >= config_.num_alternative_paths;
    }
    // skip_backtrack_start (hpp:183-185 / aligner_labeled.cpp:304-326)
    bool skip_backtrack_start(size_t i) {
        if (!prev_starts.emplace(i).second) return true;
        if (!ab_) return false;
        const Columns &end_labels = ab_->get_cached_column_set(node_labels_[i]);
        const Columns &left_labels = ab_->get_cached_column_set(remaining_labels_i_);
        label_intersection_.clear();
        label_diff_.clear();
        // utils::set_intersection_difference (common/algorithms.hpp:160-180): a & b, a - b
        auto a = left_labels.begin(), a_end = left_labels.end();
        auto b = end_labels.begin(), b_end = end_labels.end();
        while (a != a_end) {
            if (b == b_end || *a < *b) { label_diff_.push_back(*a); ++a; }
            else if (*a > *b) { ++b; }
            else { label_intersection_.push_back(*a); ++a; ++b; }
        }
        label_diff_.push_back(kNannot);
        return label_intersection_.empty();
    }
    // call_alignments (hpp:200-214 / aligner_labeled.cpp:328-448 without coordinates)
    void call_alignments(Alignment &&alignment, std::vector<Alignment> &extensions) {
        if (ab_) {
            alignment.label_columns = std::move(label_intersection_);
            label_intersection_ = Columns{};
            if (label_diff_.size() && label_diff_.back() == kNannot) {
                label_diff_.pop_back();
                remaining_labels_i_ = ab_->cache_column_set(std::move(label_diff_));
                label_diff_ = Columns{};
            }
        }
        extensions.emplace_back(std::move(alignment));
    }

    void table_emplace(Column &&col) {
        if (table.size() == table_cap_) table_cap_ = std::max<size_t>(1, 2 * table_cap_);
        table.push_back(std::move(col));
    }

    static Column create_column(size_t size, node_t node, size_t parent_i, char c, ssize_t_ offset,
                                ssize_t_ max_pos, ssize_t_ trim, size_t xdrop_cutoff_i, score_t score) {
        Column col;
        col.S.create(size); col.E.create(size); col.F.create(size);
        col.node = node; col.parent_i = parent_i; col.c = c; col.offset = offset;
        col.max_pos = max_pos; col.trim = trim; col.xdrop_cutoff_i = xdrop_cutoff_i; col.score = score;
        return col;
    }

    score_t update_seed_filter(node_t node, size_t query_start, const score_t *s_begin, const score_t *s_end) {
        // aligner_extender_methods.cpp:100-156
        if (node == NPOS) return *std::max_element(s_begin, s_end);
        if (view_.rc) node += view_.max_index();
        size_t size = s_end - s_begin;
        auto it = conv_checker_.find(node);
        if (it == conv_checker_.end()) {
            conv_checker_.emplace(node, ScoreVec(query_start, std::vector<score_t>(s_begin, s_end)));
            return *std::max_element(s_begin, s_end);
        }
        auto &[start, vec] = it->second;
        if (query_start + size <= start) {
            vec.insert(vec.begin(), start - query_start, NINF);
            std::copy(s_begin, s_end, vec.begin());
            start = query_start;
            return *std::max_element(s_begin, s_end);
        }
        if (query_start >= start + vec.size()) {
            vec.insert(vec.end(), query_start - start - vec.size(), NINF);
            vec.insert(vec.end(), s_begin, s_end);
            return *std::max_element(s_begin, s_end);
        }
        if (query_start < start) {
            vec.insert(vec.begin(), start - query_start, NINF);
            start = query_start;
        }
        if (query_start + size > start + vec.size()) vec.resize(query_start + size - start, NINF);
        score_t max_changed_value = NINF;
        score_t *v = vec.data() + query_start - start;
        for (size_t j = 0; j < size; ++j) {
            if (s_begin[j] > v[j] * config_.rel_score_cutoff) {
                v[j] = std::max(v[j], s_begin[j]);
                max_changed_value = std::max(max_changed_value, v[j]);
            }
        }
        return max_changed_value;
    }

    void call_outgoing(node_t node, const std::function<void(node_t, char, score_t)> &callback,
                       size_t table_i, bool force_fixed_seed) {
        // aligner_extender_methods.cpp:330-387 (non-canonical graphs)
        size_t k = view_.get_k();
        size_t next_offset = table[table_i].offset + 1;
        size_t seed_pos = next_offset - seed_->offset;
        bool in_seed = seed_pos < seed_->sequence.size();
        if (in_seed && next_offset < k) {
            callback(seed_->nodes.front(), seed_->sequence[seed_pos], 0);
        } else if (in_seed && force_fixed_seed) {
            size_t node_i = next_offset - k + 1;
            node_t next_node = seed_->nodes[node_i];
            char next_c = seed_->sequence[seed_pos];
            callback(next_node, next_c, next_node ? 0 : (!node ? config_.gap_extension_penalty
                                                               : config_.gap_opening_penalty));
        } else {
            if (wc_) ++wc_->n_expansions;
            view_.call_outgoing_kmers(node, [&](node_t next, char c) {
                if (c != '$') callback(next, c, 0);
            });
        }
    }

    std::vector<Alignment> extend(score_t min_path_score, bool force_fixed_seed) {
        // aligner_extender_methods.cpp:412-772 with target_length = 0, target_node = npos,
        // trim_offset_after_extend = true, trim_query_suffix = 0, added_xdrop = 0
        ++num_extensions_;
        if (wc_) ++wc_->n_extensions;
        min_path_score = std::max(0, min_path_score);
        table.clear();      // std::vector::clear keeps capacity: table_cap_ carries over between extensions
        prev_starts.clear();

        score_t xdrop = std::min(config_.xdrop, INT32_MAX - 0) + 0;
        xdrop_cutoffs_.assign(1, std::make_pair(size_t(0), std::max(-xdrop, NINF + 1)));

        size_t start = seed_->get_clipping();
        std::string_view window(seed_->query_view.data(), query_.data() + query_.size() - seed_->query_view.data());
        score_t partial_sum_offset = partial_sums_.at(start + window.size());
        ssize_t_ seed_offset = static_cast<ssize_t_>(seed_->offset) - 1;
        const size_t k = view_.get_k();

        table_emplace(create_column(1, seed_->nodes.front(), static_cast<size_t>(-1), '\0', seed_offset, 0, 0, 0u, 0));
        {
            Column &r = table[0];
            r.S[0] = config_.left_end_bonus && !seed_->get_clipping() ? config_.left_end_bonus : 0;
            extend_ins_end(r.S, r.E, r.F, window.size() + 1 - r.trim, xdrop_cutoffs_[0].second, config_);
            table_size_bytes_ = kSizeofColumn * table_cap_table()
                + (r.S.capacity() + r.E.capacity() + r.F.capacity()) * sizeof(score_t);
        }

        typedef std::tuple<score_t, ssize_t_, size_t, score_t> TableIt;
        min_cell_score_ = 0;
        score_t best_score = 0;
        std::priority_queue<TableIt> queue;
        queue.emplace(0, 0, 0, 0);
        std::vector<size_t> tips;

        while (queue.size()) {
            std::vector<TableIt> next_nodes{ queue.top() };
            queue.pop();
            while (queue.size() && std::get<0>(queue.top()) == std::get<0>(next_nodes.back())) {
                next_nodes.push_back(queue.top());
                queue.pop();
            }

            while (next_nodes.size()) {
                size_t i = std::get<2>(next_nodes.back());
                next_nodes.pop_back();

                std::vector<std::tuple<node_t, char, score_t>> outgoing;
                size_t next_offset = table[i].offset + 1;
                ssize_t_ begin = 0;
                ssize_t_ prev_end = window.size() + 1;
                size_t prev_xdrop_cutoff_i = table[i].xdrop_cutoff_i;
                score_t prev_xdrop_cutoff = xdrop_cutoffs_[prev_xdrop_cutoff_i].second;
                bool in_seed = next_offset - seed_->offset < seed_->sequence.size();
                {
                    const Column &col = table[i];
                    double node_counter = table.size();    // global_xdrop
                    if (col.S[col.max_pos - col.trim] < best_score) {
                        if (node_counter / window.size() >= config_.max_nodes_per_seq_char) {
                            queue = std::priority_queue<TableIt>();
                            next_nodes.clear();
                            continue;
                        }
                        if (static_cast<double>(table_size_bytes_) / 1'000'000 > config_.max_ram_per_alignment) {
                            queue = std::priority_queue<TableIt>();
                            next_nodes.clear();
                            continue;
                        }
                    }
                    size_t sz = col.S.size();
                    size_t b = 0;
                    while (b < sz && !(col.S[b] >= prev_xdrop_cutoff)) ++b;
                    size_t e = sz;
                    while (e > 0 && !(col.S[e - 1] >= prev_xdrop_cutoff)) --e;
                    begin = b + col.trim;
                    prev_end = e + col.trim;
                    if (prev_end <= begin) continue;

                    const node_t col_node = col.node;      // (a flush inside the labeled call may not move `col`, but keep a copy)
                    auto collect = [&](node_t next, char c, score_t s) {
                        c = toupper((unsigned char)c);
                        outgoing.emplace_back(next, c, s);
                    };
                    if (ab_) call_outgoing_labeled(col_node, collect, i, force_fixed_seed);
                    else call_outgoing(col_node, collect, i, force_fixed_seed);

                    if (outgoing.empty()) { tips.push_back(i); continue; }
                }

                size_t end = std::min(static_cast<size_t>(prev_end), window.size()) + 1;

                for (const auto &[next, c, score] : outgoing) {
                    size_t table_sizediff = table_cap_table();
                    table_emplace(create_column(end - begin, next, i, c, static_cast<ssize_t_>(next_offset),
                                                begin, begin, prev_xdrop_cutoff_i, score));
                    if (wc_) ++wc_->n_columns;
                    const Column &prev = table[i];
                    Column &cur = table.back();
                    score_t &xdrop_cutoff = xdrop_cutoffs_[cur.xdrop_cutoff_i].second;

                    update_column(prev_end - cur.trim,
                                  prev.S.data() + cur.trim - prev.trim,
                                  prev.F.data() + cur.trim - prev.trim,
                                  cur.S, cur.E, cur.F,
                                  profile_score_[encode_char(c)].data() + start + cur.trim,
                                  xdrop_cutoff, config_, score, cur.offset);
                    extend_ins_end(cur.S, cur.E, cur.F, window.size() + 1 - cur.trim, xdrop_cutoff, config_);

                    ssize_t_ cur_offset = begin;
                    ssize_t_ diag_i = cur.offset - seed_offset;
                    bool has_extension = in_seed;
                    const score_t *partial_sums = &partial_sums_[start + cur.trim];
                    // The reference is built with -mfma on AVX2 hosts (CMakeLists.txt:187-188) and GCC/Clang
                    // contract a*b+c inside one expression, so this is an fma (DESIGN.md "floating point").
                    score_t extension_cutoff = static_cast<score_t>(
                        std::fma(static_cast<double>(best_score), config_.rel_score_cutoff,
                                 static_cast<double>(partial_sum_offset)));

                    for (size_t j = 0; j < cur.S.size(); ++j, ++cur_offset) {
                        if (cur.S[j] != NINF) min_cell_score_ = std::min(min_cell_score_, cur.S[j]);
                        if (std::make_pair(cur.S[j], std::abs(cur.max_pos - diag_i))
                                > std::make_pair(cur.S[cur.max_pos - begin], std::abs(cur_offset - diag_i))) {
                            cur.max_pos = j + begin;
                        }
                        if (!has_extension && cur.S[j] + partial_sums[j] >= extension_cutoff) has_extension = true;
                    }

                    score_t max_val = cur.S[cur.max_pos - cur.trim];
                    // target_length == 0: offset - seed_offset >= 1 always, so has_extension is left as computed

                    if (!in_seed && max_val < xdrop_cutoff) { pop(table.size() - 1); continue; }
                    if (!in_seed && !has_extension) { pop(table.size() - 1); continue; }

                    table_sizediff = table_cap_table() - table_sizediff;
                    table_size_bytes_ += kSizeofColumn * table_sizediff
                        + (cur.S.capacity() + cur.E.capacity() + cur.F.capacity()) * sizeof(score_t);

                    // signed overflow in the reference when xdrop is INT32_MAX (unit-test default); it wraps
                    if (static_cast<score_t>(static_cast<uint32_t>(max_val) - static_cast<uint32_t>(xdrop_cutoff)) > xdrop)
                        xdrop_cutoff = max_val - xdrop;
                    best_score = std::max(best_score, max_val);

                    size_t vec_offset = start + begin - static_cast<bool>(begin);
                    score_t *s_begin = cur.S.data() + !begin;
                    score_t *s_end = cur.S.data() + cur.S.size();

                    score_t converged_score = update_seed_filter(next, vec_offset, s_begin, s_end);
                    if (converged_score != NINF) {
                        TableIt next_score{ converged_score, -std::abs(cur.max_pos - diag_i), table.size() - 1, max_val };
                        if (next_nodes.size() && converged_score == std::get<0>(next_nodes[0])) {
                            next_nodes.emplace_back(std::move(next_score));
                        } else {
                            queue.emplace(std::move(next_score));
                        }
                    }
                }
            }
        }

        std::sort(tips.begin(), tips.end());
        auto extensions = backtrack(min_path_score, window, config_.right_end_bonus, tips, k);
        for (auto &ext : extensions) ext.trim_offset();
        return extensions;
    }

    size_t table_cap_table() const { return table_cap_; }

    std::vector<Alignment> backtrack(score_t min_path_score, std::string_view window, score_t right_end_bonus,
                                     const std::vector<size_t> &tips, size_t k) {
        // aligner_extender_methods.cpp:800-1034 with target_node = npos
        if (ab_) flush();                               // LabeledExtender::backtrack (aligner_labeled.hpp:32-43)
        std::vector<Alignment> extensions;
        size_t seed_clipping = seed_->get_clipping();
        ssize_t_ seed_offset = static_cast<ssize_t_>(seed_->offset) - 1;
        ssize_t_ k_minus_1 = k - 1;
        ssize_t_ last_pos = window.size();
        ssize_t_ seed_dist = std::max(k, seed_->sequence.size()) - 1;
        score_t min_start_score = min_path_score;
        size_t min_trace_length = k - seed_->offset;

        std::vector<std::tuple<score_t, ssize_t_, ssize_t_, ssize_t_>> indices;
        auto it = tips.begin();
        for (size_t i = 1; i < table.size(); ++i) {
            while (it != tips.end() && i > *it) ++it;
            auto check_and_add_pos = [&](ssize_t_ start_pos, bool is_tip) {
                const Column &col = table[i];
                const Column &par = table[col.parent_i];
                if (start_pos < par.trim + 1) return;
                size_t pos = start_pos - col.trim;
                size_t pos_p = start_pos - par.trim - 1;
                if (col.S[pos] == NINF || par.S.at_ub(pos_p) == NINF) return;
                score_t end_bonus = start_pos == last_pos ? right_end_bonus : 0;
                uint8_t s = encode_char(col.c);
                if (col.S[pos] + end_bonus >= min_start_score) {
                    bool is_match = col.S[pos] == par.S.at_ub(pos_p) + col.score + profile_score_[s][seed_clipping + start_pos]
                        && profile_op_[s][seed_clipping + start_pos] == MGX_OP_MATCH;
                    if (is_match || start_pos == last_pos || is_tip) {
                        indices.emplace_back(col.S[pos] + end_bonus, -std::abs(start_pos - col.offset + seed_offset),
                                             -static_cast<ssize_t_>(i), start_pos);
                    }
                }
            };
            if (table[i].offset < seed_dist) continue;
            bool is_tip = (it != tips.end() && i == *it);
            check_and_add_pos(table[i].max_pos, is_tip);
            if ((ssize_t_)(table[i].S.size() + table[i].trim) == (ssize_t_)window.size() + 1
                    && table[i].max_pos != last_pos) {
                check_and_add_pos(last_pos, is_tip);
            }
        }

        std::make_heap(indices.begin(), indices.end());
        score_t best_score = INT32_MIN;

        for (auto rit = indices.rbegin(); rit != indices.rend(); ++rit) {
            std::pop_heap(indices.begin(), rit.base());
            const auto [start_score, neg_off_diag, neg_j_start, start_pos] = *rit;
            (void)neg_off_diag;
            if (terminate_backtrack_start(extensions)) break;
            size_t j = -neg_j_start;
            if (skip_backtrack_start(j)) continue;

            std::vector<node_t> path;
            std::vector<size_t> trace;
            Cigar ops;
            std::string seq;
            score_t score = start_score;
            if (score - min_cell_score_ < best_score) break;

            size_t dummy_counter = 0;
            ssize_t_ pos = start_pos;
            ssize_t_ end_pos = pos;
            size_t align_offset = seed_->offset;
            score_t extra_score = 0;

            auto append_node = [&](node_t node, char c, ssize_t_ offset, uint8_t op) {
                seq += c;
                ops.append(op);
                if (offset >= k_minus_1) {
                    path.emplace_back(node);
                    if (!node) {
                        ++dummy_counter;
                    } else if (dummy_counter) {
                        ops.append(MGX_OP_NODE_INSERTION, dummy_counter);
                        extra_score -= config_.gap_opening_penalty + (dummy_counter - 1) * config_.gap_extension_penalty;
                        dummy_counter = 0;
                    }
                }
            };

            while (j) {
                const Column &col = table[j];
                const Column &par = table[col.parent_i];
                const ssize_t_ trim = col.trim, trim_p = par.trim;
                align_offset = std::min(col.offset, k_minus_1);
                if (pos == col.max_pos) prev_starts.emplace(j);
                uint8_t s = encode_char(col.c);

                if (col.S[pos - trim] == NINF) {
                    j = 0;
                } else if (pos && col.S[pos - trim] == col.E[pos - trim]
                        && (ops.ops.empty() || ops.ops.back().first != MGX_OP_DELETION)) {
                    uint8_t last_op = MGX_OP_INSERTION;
                    while (last_op == MGX_OP_INSERTION) {
                        ops.append(last_op);
                        last_op = col.E[pos - trim] == col.E[pos - trim - 1] + config_.gap_extension_penalty
                            ? MGX_OP_INSERTION : MGX_OP_MATCH;
                        --pos;
                    }
                } else if (pos && pos >= trim_p + 1
                        && col.S[pos - trim] == par.S.at_ub(pos - trim_p - 1) + col.score
                            + profile_score_[s][seed_clipping + pos]) {
                    trace.emplace_back(j);
                    extra_score += col.score;
                    append_node(col.node, col.c, col.offset, profile_op_[s][seed_clipping + pos]);
                    --pos;
                    j = col.parent_i;
                } else if (col.S[pos - trim] == col.F[pos - trim]
                        && (ops.ops.empty() || ops.ops.back().first != MGX_OP_INSERTION)) {
                    uint8_t last_op = MGX_OP_DELETION;
                    while (last_op == MGX_OP_DELETION && j) {
                        const Column &c2 = table[j];
                        const Column &p2 = table[c2.parent_i];
                        align_offset = std::min(c2.offset, k_minus_1);
                        last_op = c2.F[pos - c2.trim] == p2.F.at_ub(pos - p2.trim) + c2.score + config_.gap_extension_penalty
                            ? MGX_OP_DELETION : MGX_OP_MATCH;
                        trace.emplace_back(j);
                        extra_score += c2.score;
                        append_node(c2.node, c2.c, c2.offset, MGX_OP_DELETION);
                        j = c2.parent_i;
                    }
                } else {
                    break;
                }
            }

            if (trace.size() >= min_trace_length && path.size() && path.back()) {
                score_t cur_cell_score = table[j].S[pos - table[j].trim];
                best_score = std::max(best_score, score - cur_cell_score);
                if (score - min_cell_score_ < best_score) break;
                if (score >= min_start_score
                        && (!pos || cur_cell_score == 0)
                        && (pos || cur_cell_score == table[0].S[0])
                        && (config_.allow_left_trim || !j)) {
                    call_alignments(construct_alignment(ops, pos, window.substr(pos, end_pos - pos),
                                                        path, seq, score, align_offset, extra_score), extensions);
                }
            }
        }

        if (extensions.empty() && seed_->score >= min_path_score) extensions.emplace_back(*seed_);
        return extensions;
    }

    Alignment construct_alignment(Cigar cigar, size_t clipping, std::string_view window,
                                  std::vector<node_t> final_path, std::string match, score_t score,
                                  size_t offset, score_t extra_score) const {
        // aligner_extender_methods.cpp:774-798
        cigar.append(MGX_OP_CLIPPED, clipping);
        std::reverse(cigar.ops.begin(), cigar.ops.end());
        std::reverse(final_path.begin(), final_path.end());
        std::reverse(match.begin(), match.end());
        Alignment ext(window, std::move(final_path), std::move(match), score, std::move(cigar), 0,
                      seed_->orientation, offset);
        ext.extend_query_begin(query_.data());
        ext.extend_query_end(query_.data() + query_.size());
        ext.extra_score = extra_score;
        return ext;
    }
};

// =============================================================================================
// Aggregator (A/aligner_aggregator.hpp), unlabeled case
// =============================================================================================
struct LocalAlignmentLess {                          // alignment.hpp:337-348
    bool operator()(const Alignment &a, const Alignment &b) const {
        return std::make_tuple(b.score, a.query_view.size(), a.orientation, a.get_clipping())
             > std::make_tuple(a.score, b.query_view.size(), b.orientation, b.get_clipping());
    }
};

class Aggregator {
  public:
    explicit Aggregator(const mgx_config &config) : config_(config) {}

    bool add_alignment(Alignment &&alignment) {
        // aligner_aggregator.hpp:68-138 (no labels)
        auto a = std::make_shared<Alignment>(std::move(alignment));
        if (queue_.empty()) { queue_.push_back(a); return true; }
        if (a->score < get_global_cutoff()) return false;
        for (const auto &aln : queue_) if (*a == *aln) return (bool)config_.post_chain_alignments;       // :88-91
        // "If post-alignment chaining is requested, never skip any alignments" (:92-96)
        if (config_.post_chain_alignments || queue_.size() < config_.num_alternative_paths) { queue_.push_back(a); return true; }
        auto min_it = std::min_element(queue_.begin(), queue_.end(),
            [&](const auto &x, const auto &y) { return cmp_(*x, *y); });
        if (cmp_(*a, **min_it)) return false;
        *min_it = a;
        return true;
    }

    score_t get_global_cutoff() const {
        // aligner_aggregator.hpp:141-149
        if (queue_.empty()) return NINF;
        auto max_it = std::max_element(queue_.begin(), queue_.end(),
            [&](const auto &x, const auto &y) { return cmp_(*x, *y); });
        score_t cur_max = (*max_it)->score;
        return cur_max > 0 ? cur_max * config_.rel_score_cutoff : cur_max;
    }

    std::vector<Alignment> get_alignments() {
        // aligner_aggregator.hpp:180-202
        auto ptrs = queue_;
        queue_.clear();
        std::stable_sort(ptrs.begin(), ptrs.end(), [&](const auto &x, const auto &y) { return cmp_(*x, *y); });
        std::vector<Alignment> out;
        for (auto it = ptrs.rbegin(); it != ptrs.rend(); ++it)
            if ((*it)->size()) out.emplace_back(std::move(**it));
        return out;
    }

  private:
    const mgx_config &config_;
    std::vector<std::shared_ptr<Alignment>> queue_;
    LocalAlignmentLess cmp_;
};

// construct_alignment_chain (A/aligner_chainer.cpp:623-720)
static void construct_alignment_chain(size_t node_overlap, const mgx_config &config, std::string_view query, Alignment &&chain,
                                      std::vector<Alignment>::iterator begin, std::vector<Alignment>::iterator end,
                                      std::vector<score_t> *best_score, const std::function<void(Alignment &&)> &callback) {
    const char *chain_begin = chain.query_view.data();
    const char *chain_end = chain.query_view.data() + chain.query_view.size();
    if (begin == end || chain_end == query.data() + query.size()) { callback(std::move(chain)); return; }
    score_t score = chain.score;
    bool called = false;
    for (auto it = begin; it != end; ++it) {
        if (it->offset) continue;                                    // "TODO: handle this case later" (:647-649)
        const char *next_begin = it->query_view.data();
        const char *next_end = it->query_view.data() + it->query_view.size();
        if (next_begin <= chain_begin || next_end == chain_end) continue;
        if (chain.label_columns.size()) {                            // utils::share_element (:657-663)
            Columns shared;
            std::set_intersection(it->label_columns.begin(), it->label_columns.end(), chain.label_columns.begin(),
                                  chain.label_columns.end(), std::back_inserter(shared));
            if (shared.empty()) continue;
        }
        Alignment aln = *it;
        if (next_begin >= chain_end) {
            // no overlap
            aln.insert_gap_prefix(next_begin - chain_end, node_overlap, config);
        } else {
            // trim, then fill in dummy nodes: first trim front of the incoming alignment (:672-680)
            size_t overlap = std::min(static_cast<size_t>((chain.cigar.ops.end() - 2)->second),
                                      aln.trim_query_prefix(chain_end - it->query_view.data(), node_overlap, config));
            if (aln.empty() || aln.sequence.size() <= node_overlap
                    || (aln.cigar.ops.begin() + static_cast<bool>(aln.get_clipping()))->first != MGX_OP_MATCH)
                continue;
            if (overlap < node_overlap) aln.insert_gap_prefix(-(ptrdiff_t)overlap, node_overlap, config);
            else aln.trim_clipping();
        }
        score_t next_score = score + aln.score;
        if (next_score <= (*best_score)[next_end - query.data()]) continue;
        (*best_score)[next_end - query.data()] = next_score;
        // use append instead of splice because any clipping in aln represents internally clipped characters (:703-707)
        Alignment next_chain = chain;
        next_chain.trim_end_clipping();
        bool changed = next_chain.append(std::move(aln));
        if (next_chain.size()) {
            construct_alignment_chain(node_overlap, config, query, std::move(next_chain), it + 1, end, best_score, callback);
            called |= changed;
        }
    }
    if (!called) callback(std::move(chain));
}

// chain_alignments<LocalAlignmentLess> (A/aligner_chainer.cpp:555-620): post-alignment chaining of a query's alignments
// (config.post_chain_alignments; node_overlap = k - 1, dbg_aligner.cpp:328-332)
std::vector<Alignment> chain_alignments(std::vector<Alignment> &&alignments, std::string_view query, std::string_view rc_query,
                                        const mgx_config &config, size_t node_overlap) {
    if (alignments.size() < 2 || !config.post_chain_alignments) return std::move(alignments);
    mgx_config no_chain_config = config;
    no_chain_config.post_chain_alignments = 0;
    Aggregator aggregator(no_chain_config);
    alignments.erase(std::remove_if(alignments.begin(), alignments.end(), [&](Alignment &a) {
        if (!a.get_clipping() && !a.get_end_clipping()) { aggregator.add_alignment(std::move(a)); return true; }
        return false;
    }), alignments.end());
    std::sort(alignments.begin(), alignments.end(), [](const Alignment &a, const Alignment &b) {
        return std::make_tuple(a.orientation, a.get_clipping() + a.query_view.size(), a.get_clipping(), b.score, a.sequence.size())
             < std::make_tuple(b.orientation, b.get_clipping() + b.query_view.size(), b.get_clipping(), a.score, b.sequence.size());
    });
    auto run = [&](std::string_view this_query, auto begin, auto end) {
        std::vector<score_t> best_score(this_query.size() + 1, 0);
        for (auto it = begin; it != end; ++it) {
            size_t end_pos = it->query_view.data() + it->query_view.size() - this_query.data();
            if (it->score > best_score[end_pos]) {
                best_score[end_pos] = it->score;
                construct_alignment_chain(node_overlap, config, this_query, Alignment(*it), it + 1, end, &best_score,
                                          [&](Alignment &&chain) { aggregator.add_alignment(std::move(chain)); });
            }
        }
    };
    // recursively construct chains
    auto split_it = std::find_if(alignments.begin(), alignments.end(), [](const Alignment &a) { return a.orientation; });
    run(query, alignments.begin(), split_it);
    run(rc_query, split_it, alignments.end());
    return aggregator.get_alignments();
}

// align_core (A/dbg_aligner.cpp:360-384); filter_seed for unlabeled seeds clears the seed (:105-108)
void align_core(std::vector<Alignment> seeds, Extender &extender,
                const std::function<void(Alignment &&)> &callback,
                const std::function<score_t(const Alignment &)> &get_min_path_score,
                bool force_fixed_seed) {
    for (size_t i = 0; i < seeds.size(); ++i) {
        if (seeds[i].empty()) continue;
        score_t min_path_score = get_min_path_score(seeds[i]);
        for (auto &&ext : extender.get_extensions(seeds[i], min_path_score, force_fixed_seed)) callback(std::move(ext));
        for (size_t j = i + 1; j < seeds.size(); ++j)
            if (seeds[j].size() && !extender.check_seed(seeds[j])) seeds[j] = Alignment();
    }
}

// =============================================================================================
// AlignmentAggregator with labels (A/aligner_aggregator.hpp:24-206): one queue per label + the global one
// =============================================================================================
class LabeledAggregator {
    typedef std::shared_ptr<Alignment> Ptr;
    typedef std::vector<Ptr> Queue;                      // PriorityDeque: only its minimum / maximum / content are observable
  public:
    explicit LabeledAggregator(const mgx_config &config) : config_(config) {}

    // :68-138; returns true if the alignment was added
    bool add_alignment(Alignment &&alignment) {
        auto a = std::make_shared<Alignment>(std::move(alignment));
        if (unlabeled_.empty()) {
            unlabeled_.push_back(a);
            for (Label c : a->label_columns) queue_of(c).push_back(a);
            return true;
        }
        if (a->score < get_global_cutoff()) return false;
        auto push_to_queue = [&](Queue &queue) {
            for (const auto &aln : queue) if (*a == *aln) return false;       // (post_chain_alignments == false)
            if (queue.size() < config_.num_alternative_paths) { queue.push_back(a); return true; }
            auto min_it = minimum(queue);
            if (cmp_(*a, **min_it)) return false;
            *min_it = a;                                 // queue.update(queue.begin(), a): the minimum is replaced
            return true;
        };
        if (a->label_columns.empty()) return push_to_queue(unlabeled_);
        if (path_queue_.empty()) {
            // the first labeled alignment: the global queue only serves the global cut-off from now on (:110-117)
            if (unlabeled_.size() > 1) {
                Ptr mx = *maximum(unlabeled_);
                unlabeled_.clear();
                unlabeled_.push_back(std::move(mx));
            }
        }
        bool added = false;
        for (Label c : a->label_columns) added |= push_to_queue(queue_of(c));
        if (!added) return false;
        if (!cmp_(*a, **maximum(unlabeled_))) *minimum(unlabeled_) = a;       // update(begin(), a) on a one-element queue
        return true;
    }

    score_t get_global_cutoff() const {                  // :141-149
        if (unlabeled_.empty()) return NINF;
        score_t cur_max = (*maximum(unlabeled_))->score;
        return cur_max > 0 ? cur_max * config_.rel_score_cutoff : cur_max;
    }
    score_t get_score_cutoff(const Columns &labels) const {      // :152-166
        score_t global_min = get_global_cutoff();
        score_t min_score = std::numeric_limits<score_t>::max();
        for (Label label : labels) {
            min_score = std::min(min_score, get_label_cutoff(label));
            if (min_score < global_min) return global_min;
        }
        return min_score;
    }
    size_t num_aligned_labels() const { return path_queue_.size(); }

    std::vector<Alignment> get_alignments() {            // :180-202
        std::vector<Ptr> ptrs;
        for (const auto &entry : path_queue_) std::copy(entry.second.begin(), entry.second.end(), std::back_inserter(ptrs));
        std::copy(unlabeled_.begin(), unlabeled_.end(), std::back_inserter(ptrs));
        path_queue_.clear();
        unlabeled_.clear();
        // std::sort in the reference; equal alignments are the same object or compare equal in every reported field but the
        // labels, and a moved-from duplicate is dropped below, so the order among equals only shows when two DIFFERENT
        // alignments tie under LocalAlignmentLess (unpinned upstream, like the unlabeled case)
        std::stable_sort(ptrs.begin(), ptrs.end(), [&](const Ptr &x, const Ptr &y) { return cmp_(*x, *y); });
        std::vector<Alignment> out;
        for (auto it = ptrs.rbegin(); it != ptrs.rend(); ++it) {
            if ((*it)->size()) {
                out.emplace_back(std::move(**it));
                **it = Alignment();
            }
        }
        return out;
    }

  private:
    const mgx_config &config_;
    std::vector<std::pair<Label, Queue>> path_queue_;   // VectorMap<Label, PathQueue>: insertion order
    Queue unlabeled_;
    LocalAlignmentLess cmp_;

    Queue &queue_of(Label c) {
        for (auto &e : path_queue_) if (e.first == c) return e.second;
        path_queue_.emplace_back(c, Queue{});
        return path_queue_.back().second;
    }
    const Queue *find_queue(Label c) const {
        for (const auto &e : path_queue_) if (e.first == c) return &e.second;
        return nullptr;
    }
    Queue::iterator minimum(Queue &q) const {
        return std::min_element(q.begin(), q.end(), [&](const Ptr &x, const Ptr &y) { return cmp_(*x, *y); });
    }
    Queue::const_iterator maximum(const Queue &q) const {
        return std::max_element(q.begin(), q.end(), [&](const Ptr &x, const Ptr &y) { return cmp_(*x, *y); });
    }
    Queue::iterator maximum(Queue &q) const {
        return std::max_element(q.begin(), q.end(), [&](const Ptr &x, const Ptr &y) { return cmp_(*x, *y); });
    }
    score_t get_label_cutoff(Label label) const {       // :168-177
        const Queue *q = find_queue(label);
        if (!q || q->size() < config_.num_alternative_paths) return NINF;
        return (*std::min_element(q->begin(), q->end(), [&](const Ptr &x, const Ptr &y) { return cmp_(*x, *y); }))->score;
    }
};

// filter_seed (A/dbg_aligner.cpp:105-149, no coordinates): a seed that check_seed rejected keeps only the labels the
// previous seed did not have
void filter_seed_labeled(const Alignment &prev, Alignment &a) {
    if (prev.label_columns.empty()) { a = Alignment(); return; }
    Columns diff;
    std::set_difference(a.label_columns.begin(), a.label_columns.end(), prev.label_columns.begin(), prev.label_columns.end(),
                        std::back_inserter(diff));
    if (diff.empty()) a = Alignment();
    else std::swap(a.label_columns, diff);
}

// align_core (A/dbg_aligner.cpp:360-384) with labeled seeds
void align_core_labeled(std::vector<Alignment> seeds, Extender &extender,
                        const std::function<void(Alignment &&)> &callback,
                        const std::function<score_t(const Alignment &)> &get_min_path_score,
                        bool force_fixed_seed) {
    for (size_t i = 0; i < seeds.size(); ++i) {
        if (seeds[i].empty()) continue;
        score_t min_path_score = get_min_path_score(seeds[i]);
        for (auto &&ext : extender.get_extensions(seeds[i], min_path_score, force_fixed_seed)) callback(std::move(ext));
        for (size_t j = i + 1; j < seeds.size(); ++j)
            if (seeds[j].size() && !extender.check_seed(seeds[j])) filter_seed_labeled(seeds[i], seeds[j]);
    }
}

// get_num_char_matches_in_seeds (A/alignment.hpp:100-127).  Quirk kept: `aln` refers to the seed that has the offset, so the
// inner loop's condition does not change and runs to the end — nothing after the first sub-k seed is counted.
size_t get_num_char_matches_in_seeds(const std::vector<Seed> &seeds) {
    size_t num_matching = 0, last_q_end = 0;
    for (size_t i = 0; i < seeds.size(); ++i) {
        const Seed &aln = seeds[i];
        if (aln.nodes.empty()) continue;
        size_t q_begin = aln.clipping, q_end = q_begin + aln.query_view.size();
        if (q_end > last_q_end) {
            num_matching += q_end - q_begin;
            if (q_begin < last_q_end) num_matching -= last_q_end - q_begin;
        }
        if (aln.offset) i = seeds.size() - 1;
        last_q_end = q_end;
    }
    return num_matching;
}

} // namespace

// =============================================================================================
// Annotation (ColumnMajor get_rows, AnnotatedDBG::annotate_sequence)
// =============================================================================================
std::vector<Columns> Annotation::get_rows(const std::vector<uint64_t> &rows) const {
    // column_major.cpp:27-44: for every column, for every requested row: test the bit
    std::vector<Columns> out(rows.size());
    for (size_t j = 0; j < columns.size(); ++j)
        for (size_t i = 0; i < rows.size(); ++i)
            if (rows[i] < n_rows && get(rows[i], j)) out[i].push_back(j);
    return out;
}

void Annotation::annotate_sequence(const Graph &graph, std::string_view sequence, size_t column) {
    // annotated_dbg.cpp:55-75: graph_->map_to_nodes(sequence, [&](node_index i) { if (i > 0) indices.push_back(graph_to_anno_index(i)); })
    // BASIC graphs: map_to_nodes == map_to_nodes_sequentially; a PRIMARY graph is annotated through its CanonicalDBG wrapper
    // (CanonicalDBG::map_to_nodes reports base nodes, canonical_dbg.cpp:148-154)
    std::vector<node_t> nodes;
    if (graph.mode == PRIMARY) {
        CanonicalView canon(graph);
        nodes = canon.map_to_nodes_sequentially(sequence);
        for (node_t &v : nodes) v = canon.get_base_node(v);
    } else {
        nodes = graph.map_to_nodes(sequence);            // (CANONICAL mode: the k-mers' representatives)
    }
    for (node_t v : nodes) if (v > 0) set(v - 1, column);
}

// =============================================================================================
// LabeledAligner (A/aligner_labeled.cpp:450-721 + DBGAligner::align_batch / align_both_directions with labeled seeds)
// =============================================================================================
LabeledAligner::LabeledAligner(const Graph &graph, const mgx_config &config, const Annotation &annotation)
      : graph_(graph), config_(config), annotation_(annotation) {
    // DBGAligner ctor (dbg_aligner.cpp:33-61) ...
    size_t k = graph_.get_k();
    if (!config_.min_seed_length) config_.min_seed_length = k;
    if (!config_.max_seed_length) config_.max_seed_length = k;
    uint64_t lo = std::min(config_.min_seed_length, config_.max_seed_length);
    uint64_t hi = std::max(config_.min_seed_length, config_.max_seed_length);
    config_.min_seed_length = lo;
    config_.max_seed_length = hi;
    if (!check_config_scores(config_))
        throw std::runtime_error("Error: sum of min_cell_score and lowest penalty too low.");
    if (config_.chain_alignments || config_.post_chain_alignments || !config_.global_xdrop || config_.no_backtrack)
        throw std::runtime_error("oracle: chaining / per-branch xdrop / no_backtrack are out of scope");
    // ... then LabeledAligner's (aligner_labeled.cpp:463-465; no coordinates: no chaining)
    config_.min_seed_length = std::min<uint64_t>(k, config_.min_seed_length);
    config_.max_seed_length = std::min<uint64_t>(k, config_.max_seed_length);
}

AlignmentResults LabeledAligner::align(std::string_view query) const {
    std::vector<AlignmentResults> res;
    align_batch({ std::string(query) }, &res);
    return std::move(res[0]);
}

namespace {
// LabeledAligner::filter_seeds (aligner_labeled.cpp:612-721, no coordinates); returns the new num_matching
size_t filter_seeds(std::vector<Seed> &seeds, const AnnotationBuffer &ab, const mgx_config &config, size_t k) {
    if (seeds.empty()) return 0;
    size_t query_size = seeds[0].clipping + seeds[0].end_clipping + seeds[0].query_view.size();
    Columns labels;
    {
        std::vector<std::pair<Label, std::vector<bool>>> label_mapper;          // VectorMap: insertion order
        auto indicator_of = [&](Label c) -> std::vector<bool> & {
            for (auto &e : label_mapper) if (e.first == c) return e.second;
            label_mapper.emplace_back(c, std::vector<bool>());
            return label_mapper.back().second;
        };
        for (const Seed &seed : seeds) {
            size_t end = seed.clipping + k - seed.offset;
            const Columns *node_labels = ab.get_labels(seed.nodes[0]);
            if (!node_labels) throw std::logic_error("oracle: filter_seeds(): seed labels not fetched");
            for (Label label : *node_labels) {
                std::vector<bool> &indicator = indicator_of(label);
                if (indicator.empty()) indicator.assign(query_size, false);
                for (size_t i = seed.clipping; i < end; ++i) indicator[i] = true;
            }
        }
        if (label_mapper.empty()) { seeds.clear(); return 0; }
        std::vector<std::pair<Label, uint64_t>> label_counts;
        for (const auto &e : label_mapper)
            label_counts.emplace_back(e.first, (uint64_t)std::count(e.second.begin(), e.second.end(), true));
        // std::sort(..., utils::GreaterSecond()): only the SET of labels at or above the cut-off is used afterwards
        std::stable_sort(label_counts.begin(), label_counts.end(), [](const auto &a, const auto &b) { return a.second > b.second; });
        double cutoff = config.min_exact_match * query_size;
        auto it = std::find_if(label_counts.begin(), label_counts.end(), [cutoff](const auto &a) { return a.second < cutoff; });
        label_counts.erase(it, label_counts.end());
        for (const auto &lc : label_counts) labels.push_back(lc.first);
    }
    if (labels.empty()) { seeds.clear(); return 0; }
    std::sort(labels.begin(), labels.end());
    for (Seed &seed : seeds) {
        if (!seed.has_label_encoder) {
            seed.label_columns.clear();
            const Columns *fetch_labels = ab.get_labels(seed.nodes[0]);
            std::set_intersection(fetch_labels->begin(), fetch_labels->end(), labels.begin(), labels.end(),
                                  std::back_inserter(seed.label_columns));
            if (seed.label_columns.size()) seed.has_label_encoder = true;
        }
    }
    seeds.erase(std::remove_if(seeds.begin(), seeds.end(),
                               [](const Seed &a) { return !a.has_label_encoder || a.label_columns.empty(); }), seeds.end());
    return get_num_char_matches_in_seeds(seeds);
}
} // namespace

void LabeledAligner::align_batch(const std::vector<std::string> &queries, std::vector<AlignmentResults> *results) const {
    results->clear();
    results->resize(queries.size());
    const size_t k = graph_.get_k();
    const CanonicalView canon_store(graph_);
    const CanonicalView *canon = graph_.mode == PRIMARY ? &canon_store : nullptr;
    const GraphView gview{ &graph_, false, canon };
    const bool canonical_mode = graph_.mode == CANONICAL || canon;
    AnnotationBuffer annotation_buffer(graph_, canon, annotation_);              // aligner member, shared by the batch

    // build_seeders (dbg_aligner.cpp:193-248) for the whole batch, then the label filter (aligner_labeled.cpp:479-558)
    struct QuerySeeds { std::vector<Seed> fwd, rc; size_t nm_fwd = 0, nm_rc = 0; bool has_rc = false; };
    std::vector<QuerySeeds> batch(queries.size());
    for (size_t qi = 0; qi < queries.size(); ++qi) {
        const std::string &raw = queries[qi];
        AlignmentResults &res = (*results)[qi];
        res.query.reserve(std::max<size_t>(raw.size(), 32) + 8);
        for (char ch : raw) { int8_t c = (int8_t)ch; res.query.push_back(c >= 0 ? (char)toupper(c) : (char)127); }
        res.query_rc.reserve(res.query.capacity());
        res.query_rc = res.query;
        reverse_complement_inplace(res.query_rc);
        std::string_view this_query = res.query, reverse = res.query_rc;
        std::vector<node_t> nodes;
        if (config_.max_seed_length >= k) nodes = gview.map_to_nodes_sequentially(raw);
        else if (this_query.size() >= k) nodes.resize(this_query.size() - k + 1);
        SeederState seeder = make_suffix_seeder(gview, this_query, false, nodes, config_, nullptr);
        if (this_query.size() * config_.min_exact_match > seeder.num_matching) { seeder.seeds.clear(); seeder.num_matching = 0; }
        QuerySeeds &qs = batch[qi];
        qs.fwd = std::move(seeder.seeds); qs.nm_fwd = seeder.num_matching;
        qs.has_rc = config_.forward_and_reverse_complement || canonical_mode;
        if (qs.has_rc) {
            std::vector<node_t> nodes_rc = nodes;
            if (config_.max_seed_length >= k) { std::string dummy(raw); gview.reverse_complement_seq_path(dummy, nodes_rc); }
            SeederState seeder_rc = make_suffix_seeder(gview, reverse, true, nodes_rc, config_, nullptr);
            if (reverse.size() * config_.min_exact_match > seeder_rc.num_matching) { seeder_rc.seeds.clear(); seeder_rc.num_matching = 0; }
            qs.rc = std::move(seeder_rc.seeds); qs.nm_rc = seeder_rc.num_matching;
        }
        for (const Seed &seed : qs.fwd) annotation_buffer.queue_path(std::vector<node_t>(seed.nodes));
        for (const Seed &seed : qs.rc) annotation_buffer.queue_path(std::vector<node_t>(seed.nodes));
    }
    annotation_buffer.fetch_queued_annotations();
    for (QuerySeeds &qs : batch) {
        if (qs.fwd.size()) qs.nm_fwd = filter_seeds(qs.fwd, annotation_buffer, config_, k);
        if (qs.has_rc && qs.rc.size()) qs.nm_rc = filter_seeds(qs.rc, annotation_buffer, config_, k);
    }

    // DBGAligner::align_batch (dbg_aligner.cpp:263-355)
    for (size_t qi = 0; qi < queries.size(); ++qi) {
        AlignmentResults &res = (*results)[qi];
        QuerySeeds &qs = batch[qi];
        std::string_view this_query = res.query, reverse = res.query_rc;
        res.seeds_fwd = qs.fwd; res.seeds_rc = qs.rc;
        res.num_matches_fwd = qs.nm_fwd; res.num_matches_rc = qs.nm_rc;
        LabeledAggregator aggregator(config_);
        auto add_alignment = [&](Alignment &&a) { aggregator.add_alignment(std::move(a)); };
        auto get_min_path_score = [&](const Alignment &seed) {
            return std::max(config_.min_path_score, seed.label_columns.size() ? aggregator.get_score_cutoff(seed.label_columns)
                                                                              : aggregator.get_global_cutoff());
        };
        Extender extender(graph_, config_, this_query, nullptr, canon, &annotation_buffer);
        if (qs.has_rc) {
            Extender extender_rc(graph_, config_, reverse, nullptr, canon, &annotation_buffer);
            auto fwd_seeds = seeds_to_alignments(qs.fwd, config_);
            auto bwd_seeds = seeds_to_alignments(qs.rc, config_);
            // align_both_directions (dbg_aligner.cpp:531-758), the branch without chaining
            auto aln_both = [&](std::string_view query, std::string_view query_rc, std::vector<Alignment> &&seeds,
                                Extender &fwd_extender, Extender &bwd_extender) {
                const bool use_rcdbg = !canonical_mode && config_.forward_and_reverse_complement;
                auto is_reversible = [&](const Alignment &a) { return canonical_mode && a.orientation && !a.offset; };
                fwd_extender.set_graph(false);
                bwd_extender.set_graph(use_rcdbg);
                const GraphView plain = gview;
                if (seeds.empty()) return;
                for (size_t i = 0; i < seeds.size(); ++i) {
                    if (seeds[i].empty()) continue;
                    score_t min_path_score = config_.min_cell_score;
                    auto extensions = fwd_extender.get_extensions(seeds[i], min_path_score, false);
                    std::vector<Alignment> rc_of_alignments;
                    for (Alignment &path : extensions) {
                        if (path.score >= get_min_path_score(path)) {
                            if (is_reversible(path)) {
                                Alignment out_path = path;
                                out_path.reverse_complement(plain, query_rc);
                                add_alignment(std::move(out_path));
                            } else {
                                add_alignment(Alignment(path));
                            }
                        }
                        if (!path.get_clipping() || path.offset) continue;
                        path.reverse_complement(bwd_extender.view(), query_rc);
                        if (path.empty()) continue;
                        rc_of_alignments.emplace_back(std::move(path));
                    }
                    align_core_labeled(std::move(rc_of_alignments), bwd_extender,
                        [&](Alignment &&path) {
                            if (use_rcdbg || is_reversible(path)) {
                                path.reverse_complement(bwd_extender.view(), query);
                                if (path.empty()) return;
                                for (node_t node : path.nodes)
                                    fwd_extender.filter_nodes(node, path.get_clipping(), query.size() - path.get_end_clipping());
                            }
                            add_alignment(std::move(path));
                        },
                        get_min_path_score, true);
                    for (size_t j = i + 1; j < seeds.size(); ++j)
                        if (seeds[j].size() && !fwd_extender.check_seed(seeds[j])) filter_seed_labeled(seeds[i], seeds[j]);
                }
            };
            size_t fwd_num_matches = qs.nm_fwd, bwd_num_matches = qs.nm_rc;
            if (fwd_num_matches >= bwd_num_matches) {
                aln_both(this_query, reverse, std::move(fwd_seeds), extender, extender_rc);
                if (bwd_num_matches >= fwd_num_matches * config_.rel_score_cutoff)
                    aln_both(reverse, this_query, std::move(bwd_seeds), extender_rc, extender);
            } else {
                aln_both(reverse, this_query, std::move(bwd_seeds), extender_rc, extender);
                if (fwd_num_matches >= bwd_num_matches * config_.rel_score_cutoff)
                    aln_both(this_query, reverse, std::move(fwd_seeds), extender, extender_rc);
            }
        } else {
            align_core_labeled(seeds_to_alignments(qs.fwd, config_), extender, add_alignment, get_min_path_score, false);
        }
        res.alignments = aggregator.get_alignments();
    }
}

uint64_t g_oob_reads_total() { return g_oob_reads.load(); }
uint64_t g_unfetched_label_lookups_total() { return g_unfetched_label_lookups.load(); }

// =============================================================================================
// DBGAligner (A/dbg_aligner.cpp)
// =============================================================================================
Aligner::Aligner(const Graph &graph, const mgx_config &config) : graph_(graph), config_(config) {
    // dbg_aligner.cpp:33-61
    size_t k = graph_.get_k();
    if (!config_.min_seed_length) config_.min_seed_length = k;
    if (!config_.max_seed_length) config_.max_seed_length = k;
    uint64_t lo = std::min(config_.min_seed_length, config_.max_seed_length);
    uint64_t hi = std::max(config_.min_seed_length, config_.max_seed_length);
    config_.min_seed_length = lo;
    config_.max_seed_length = hi;
    if (!check_config_scores(config_))
        throw std::runtime_error("Error: sum of min_cell_score and lowest penalty too low.");
    if (config_.chain_alignments) config_.allow_left_trim = false;
    // PRIMARY graphs are wrapped into CanonicalDBG (dbg_aligner.cpp:52-53; cli/align.cpp:383-399 wrap_graph)
    // (post_chain_alignments: restated — chain_alignments above; seed chaining is not)
    if (config_.chain_alignments || !config_.global_xdrop || config_.no_backtrack)
        throw std::runtime_error("oracle: seed chaining / per-branch xdrop / no_backtrack are out of scope");
}

AlignmentResults Aligner::align(std::string_view query) const {
    std::vector<AlignmentResults> res;
    align_batch({ std::string(query) }, &res);
    return std::move(res[0]);
}

void Aligner::align_batch(const std::vector<std::string> &queries, std::vector<AlignmentResults> *results,
                          WorkCounters *counters) const {
    // dbg_aligner.cpp:251-355
    results->clear();
    results->resize(queries.size());
    const size_t k = graph_.get_k();
    // a PRIMARY graph is seen through the CanonicalDBG wrapper, which reports CANONICAL mode (canonical_dbg.hpp)
    const CanonicalView canon_store(graph_);
    const CanonicalView *canon = graph_.mode == PRIMARY ? &canon_store : nullptr;
    const GraphView gview{ &graph_, false, canon };
    const bool canonical_mode = graph_.mode == CANONICAL || canon;

    for (size_t qi = 0; qi < queries.size(); ++qi) {
        const std::string &raw = queries[qi];
        AlignmentResults &res = (*results)[qi];
        WorkCounters wc;
        // AlignmentResults ctor (alignment.cpp:1348-1372)
        res.query.reserve(std::max<size_t>(raw.size(), 32) + 8);     // disables SSO like the reference
        for (char ch : raw) {
            int8_t c = (int8_t)ch;
            res.query.push_back(c >= 0 ? (char)toupper(c) : (char)127);
        }
        res.query_rc.reserve(res.query.capacity());
        res.query_rc = res.query;
        reverse_complement_inplace(res.query_rc);

        std::string_view this_query = res.query;
        std::string_view reverse = res.query_rc;

        // build_seeders (dbg_aligner.cpp:193-248)
        std::vector<node_t> nodes;
        if (config_.max_seed_length >= k) {
            nodes = gview.map_to_nodes_sequentially(raw);
            wc.n_map_fwd += nodes.size();
        } else if (this_query.size() >= k) {
            nodes.resize(this_query.size() - k + 1);
        }
        res.nodes_fwd = nodes;
        SeederState seeder = make_suffix_seeder(gview, this_query, false, nodes, config_, &wc);
        if (this_query.size() * config_.min_exact_match > seeder.num_matching) { seeder.seeds.clear(); seeder.num_matching = 0; }

        bool have_rc = config_.forward_and_reverse_complement || canonical_mode;              // dbg_aligner.cpp:225-226
        SeederState seeder_rc;
        if (have_rc) {
            std::vector<node_t> nodes_rc = nodes;
            if (config_.max_seed_length >= k) {
                std::string dummy(raw);
                gview.reverse_complement_seq_path(dummy, nodes_rc);        // sequence_graph.cpp:563-573
                wc.n_map_fwd += nodes_rc.size();
            }
            res.nodes_rc = nodes_rc;
            seeder_rc = make_suffix_seeder(gview, reverse, true, nodes_rc, config_, &wc);
            if (reverse.size() * config_.min_exact_match > seeder_rc.num_matching) { seeder_rc.seeds.clear(); seeder_rc.num_matching = 0; }
        }
        res.seeds_fwd = seeder.seeds;
        res.seeds_rc = seeder_rc.seeds;
        res.num_matches_fwd = seeder.num_matching;
        res.num_matches_rc = seeder_rc.num_matching;
        wc.n_seeds += seeder.seeds.size() + seeder_rc.seeds.size();

        Aggregator aggregator(config_);
        auto add_alignment = [&](Alignment &&a) { aggregator.add_alignment(std::move(a)); };
        auto get_min_path_score = [&](const Alignment &) {
            return std::max(config_.min_path_score, aggregator.get_global_cutoff());
        };

        Extender extender(graph_, config_, this_query, &wc, canon);
        if (have_rc) {
            Extender extender_rc(graph_, config_, reverse, &wc, canon);
            // align_both_directions (dbg_aligner.cpp:531-758), no chaining
            auto fwd_seeds = seeds_to_alignments(seeder.seeds, config_);
            auto bwd_seeds = seeds_to_alignments(seeder_rc.seeds, config_);

            auto aln_both = [&](std::string_view query, std::string_view query_rc, std::vector<Alignment> &&seeds,
                                Extender &fwd_extender, Extender &bwd_extender) {
                // :644-655: a CANONICAL-mode graph holds both strands itself — the backward pass runs on the same graph, and
                // an alignment on the reverse strand is reported as the forward-strand alignment it mirrors
                const bool use_rcdbg = !canonical_mode && config_.forward_and_reverse_complement;
                auto is_reversible = [&](const Alignment &a) { return canonical_mode && a.orientation && !a.offset; };
                fwd_extender.set_graph(false);
                bwd_extender.set_graph(use_rcdbg);
                const GraphView plain = gview;
                if (seeds.empty()) return;
                for (size_t i = 0; i < seeds.size(); ++i) {
                    if (seeds[i].empty()) continue;
                    score_t min_path_score = config_.min_cell_score;
                    auto extensions = fwd_extender.get_extensions(seeds[i], min_path_score, false);
                    std::vector<Alignment> rc_of_alignments;
                    for (Alignment &path : extensions) {
                        if (path.score >= get_min_path_score(path)) {
                            if (is_reversible(path)) {
                                Alignment out_path = path;
                                out_path.reverse_complement(plain, query_rc);
                                add_alignment(std::move(out_path));
                            } else {
                                add_alignment(Alignment(path));
                            }
                        }
                        if (!path.get_clipping() || path.offset) continue;
                        path.reverse_complement(bwd_extender.view(), query_rc);
                        if (path.empty()) continue;
                        rc_of_alignments.emplace_back(std::move(path));
                    }
                    align_core(std::move(rc_of_alignments), bwd_extender,
                        [&](Alignment &&path) {
                            if (use_rcdbg || is_reversible(path)) {
                                path.reverse_complement(bwd_extender.view(), query);
                                if (path.empty()) return;
                                for (node_t node : path.nodes)
                                    fwd_extender.filter_nodes(node, path.get_clipping(), query.size() - path.get_end_clipping());
                            }
                            add_alignment(std::move(path));
                        },
                        get_min_path_score, true);
                    for (size_t j = i + 1; j < seeds.size(); ++j)
                        if (seeds[j].size() && !fwd_extender.check_seed(seeds[j])) seeds[j] = Alignment();
                }
            };

            size_t fwd_num_matches = seeder.num_matching;
            size_t bwd_num_matches = seeder_rc.num_matching;
            if (fwd_num_matches >= bwd_num_matches) {
                aln_both(this_query, reverse, std::move(fwd_seeds), extender, extender_rc);
                if (bwd_num_matches >= fwd_num_matches * config_.rel_score_cutoff)
                    aln_both(reverse, this_query, std::move(bwd_seeds), extender_rc, extender);
            } else {
                aln_both(reverse, this_query, std::move(bwd_seeds), extender_rc, extender);
                if (fwd_num_matches >= bwd_num_matches * config_.rel_score_cutoff)
                    aln_both(this_query, reverse, std::move(fwd_seeds), extender, extender_rc);
            }
        } else {
            align_core(seeds_to_alignments(seeder.seeds, config_), extender, add_alignment, get_min_path_score, false);
        }

        // dbg_aligner.cpp:328-332 (a pass-through unless config.post_chain_alignments, aligner_chainer.cpp:556-561)
        res.alignments = chain_alignments(aggregator.get_alignments(), this_query, reverse, config_, k - 1);
        if (counters) counters->add(wc);
    }
}

std::string format_alignment_tsv(const std::string &header, const AlignmentResults &paths, int32_t min_path_score) {
    // cli/align.cpp:262-285, alignment.hpp:426-433
    std::string s = header + "\t" + paths.query;
    if (paths.alignments.empty()) {
        s += "\t*\t*\t" + std::to_string(min_path_score) + "\t*\t*\t*\n";
    } else {
        for (const auto &a : paths.alignments) {
            s += std::string("\t") + (a.orientation ? "-" : "+") + "\t" + a.sequence + "\t" + std::to_string(a.score)
               + "\t" + std::to_string(a.cigar.get_num_matches()) + "\t" + a.cigar.to_string() + "\t"
               + std::to_string(a.offset);
        }
        s += "\n";
    }
    return s;
}

} // namespace orc
