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
#include <set>
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
        cigar(MGX_OP_CLIPPED, seed.clipping), label_columns(seed.label_columns), label_coordinates(seed.label_coordinates) {
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

// utils::match_indexed_values (common/algorithms.hpp:184-206): callback(index, value1, value2) for every index both sorted
// index ranges hold
template <class Callback>
static void match_indexed_values(const Columns &i1, const CoordinateSet &v1, const Columns &i2, const CoordinateSet &v2, const Callback &callback) {
    size_t a = 0, b = 0;
    while (a < i1.size() && b < i2.size()) {
        if (i1[a] < i2[b]) { ++a; continue; }
        if (i1[a] == i2[b]) { callback(i1[a], v1[a], v2[b]); ++a; }
        ++b;
    }
}
// utils::set_intersection with a delta (common/algorithms.hpp:208-225): the elements a of A with a + delta in B
static void set_intersection_delta(const Tuple &A, const Tuple &B, Tuple *out, int64_t delta) {
    size_t a = 0, b = 0;
    while (a < A.size() && b < B.size()) {
        if (A[a] + delta < B[b]) ++a;
        else if (A[a] + delta > B[b]) ++b;
        else { out->push_back(A[a]); ++a; ++b; }
    }
}

bool Alignment::append(Alignment &&other) {
    // alignment.cpp:94-175
    bool ret_val = false;
    if (label_coordinates.size() && other.label_coordinates.empty()) label_coordinates.clear();
    if (label_columns.size() && other.label_columns.empty()) label_columns.clear();
    if (label_coordinates.size()) {
        // (:107-149) coordinates: the alignments fit together only where a coordinate of `other` continues one of this
        Columns merged_columns;
        CoordinateSet merged_coordinates;
        const int64_t len = (int64_t)sequence.size();
        match_indexed_values(label_columns, label_coordinates, other.label_columns, other.label_coordinates,
            [&](Label col, const Tuple &coords, const Tuple &other_coords) {
                Tuple merged;
                set_intersection_delta(coords, other_coords, &merged, len);
                if (merged.size()) { merged_columns.push_back(col); merged_coordinates.push_back(std::move(merged)); }
            });
        if (merged_columns.empty()) { *this = Alignment(); return true; }
        ret_val = merged_columns.size() < label_columns.size();
        if (!ret_val) {
            for (size_t i = 0; i < label_columns.size(); ++i)
                if (merged_coordinates[i].size() < label_coordinates[i].size()) { ret_val = true; break; }
        }
        std::swap(label_columns, merged_columns);
        std::swap(label_coordinates, merged_coordinates);
    } else if (label_columns.size()) {
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
    {
        const int64_t seq_trim = s_it - sequence.begin();              // (:261-266)
        for (Tuple &coords : label_coordinates) for (int64_t &c : coords) c += seq_trim;
    }
    if (!clipping && it != cigar.ops.begin()) score -= config.left_end_bonus;
    nodes.erase(nodes.begin(), node_it);
    sequence.erase(sequence.begin(), s_it);
    it->second -= cigar_offset;
    cigar.ops.erase(cigar.ops.begin(), it);
    extend_query_begin(query_begin);
    return cigar_offset;
}

bool Alignment::splice(Alignment &&other) {
    // alignment.hpp:196-205
    if (empty()) { std::swap(*this, other); return label_columns.size(); }
    trim_end_clipping();
    other.trim_clipping();
    return append(std::move(other));
}

// What the four trimming routines share (alignment.cpp:192-538): CIGAR operators are consumed one at a time from one end, the
// score gives back what each was worth, reference-consuming operators advance through sequence and nodes.
size_t Alignment::trim_query_suffix(size_t n, const mgx_config &config, bool trim_excess_deletions) {
    // alignment.cpp:280-362
    const size_t end_clipping = get_end_clipping();
    const char *query_end = query_view.data() + query_view.size() + end_clipping;
    trim_end_clipping();
    auto it = cigar.ops.rbegin();
    size_t cigar_offset = 0;
    auto s_it = sequence.rbegin();
    auto node_it = nodes.rbegin();
    auto consume_ref = [&]() {
        ++s_it;
        if (node_it + 1 < nodes.rend()) ++node_it;
        else *this = Alignment();
    };
    while (n || (trim_excess_deletions && it != cigar.ops.rend() && it->first == MGX_OP_DELETION)) {
        if (it == cigar.ops.rend()) { *this = Alignment(); return 0; }
        switch (it->first) {
            case MGX_OP_MATCH:
            case MGX_OP_MISMATCH:
                score -= config.score_matrix[(uint8_t)query_view.back() & 127][(uint8_t)*s_it & 127];
                query_view.remove_suffix(1);
                --n;
                consume_ref();
                if (empty()) return 0;
                break;
            case MGX_OP_INSERTION:
                score -= it->second - cigar_offset == 1 ? config.gap_opening_penalty : config.gap_extension_penalty;
                query_view.remove_suffix(1);
                --n;
                break;
            case MGX_OP_DELETION:
                score -= it->second - cigar_offset == 1 ? config.gap_opening_penalty : config.gap_extension_penalty;
                consume_ref();
                if (empty()) return 0;
                break;
            default:
                throw std::runtime_error("trimming chains not supported");
        }
        ++cigar_offset;
        if (cigar_offset == it->second) { ++it; cigar_offset = 0; }
    }
    if (!end_clipping && (cigar_offset || it.base() != cigar.ops.end())) score -= config.right_end_bonus;
    nodes.erase(node_it.base(), nodes.end());
    sequence.erase(s_it.base(), sequence.end());
    it->second -= cigar_offset;
    cigar.ops.erase(it.base(), cigar.ops.end());
    extend_query_end(query_end);
    return cigar_offset;
}

size_t Alignment::trim_reference_prefix(size_t n, size_t node_overlap, const mgx_config &config, bool trim_excess_insertions) {
    // alignment.cpp:364-455
    const size_t clipping = get_clipping();
    const char *query_begin = query_view.data() - clipping;
    auto it = cigar.ops.begin() + static_cast<bool>(clipping);
    size_t cigar_offset = 0;
    auto s_it = sequence.begin();
    auto node_it = nodes.begin();
    int64_t seq_trim = 0;
    auto consume_ref = [&]() {
        if (*s_it != '$') { ++seq_trim; --n; }
        ++s_it;
        if (offset < node_overlap) ++offset;
        else if (node_it + 1 < nodes.end()) ++node_it;
        else *this = Alignment();
    };
    while (n || (trim_excess_insertions && it != cigar.ops.end() && it->first == MGX_OP_INSERTION)) {
        if (it == cigar.ops.end()) { *this = Alignment(); return 0; }
        switch (it->first) {
            case MGX_OP_MATCH:
            case MGX_OP_MISMATCH:
                score -= config.score_matrix[(uint8_t)query_view[0] & 127][(uint8_t)*s_it & 127];
                query_view.remove_prefix(1);
                consume_ref();
                if (empty()) return 0;
                break;
            case MGX_OP_INSERTION:
                score -= it->second - cigar_offset == 1 ? config.gap_opening_penalty : config.gap_extension_penalty;
                query_view.remove_prefix(1);
                break;
            case MGX_OP_DELETION:
                score -= it->second - cigar_offset == 1 ? config.gap_opening_penalty : config.gap_extension_penalty;
                consume_ref();
                if (empty()) return 0;
                break;
            case MGX_OP_NODE_INSERTION:
                score -= it->second - cigar_offset == 1 ? config.gap_opening_penalty : config.gap_extension_penalty;
                break;
            default:
                throw std::runtime_error("trim_reference_prefix: a clipping inside the alignment");
        }
        ++cigar_offset;
        if (cigar_offset == it->second) { ++it; cigar_offset = 0; }
    }
    for (Tuple &coords : label_coordinates) for (int64_t &c : coords) c += seq_trim;
    if (!clipping && (cigar_offset || it != cigar.ops.begin())) score -= config.left_end_bonus;
    nodes.erase(nodes.begin(), node_it);
    sequence.erase(sequence.begin(), s_it);
    it->second -= cigar_offset;
    cigar.ops.erase(cigar.ops.begin(), it);
    extend_query_begin(query_begin);
    return cigar_offset;
}

size_t Alignment::trim_reference_suffix(size_t n, const mgx_config &config, bool trim_excess_insertions) {
    // alignment.cpp:457-538
    const size_t end_clipping = get_end_clipping();
    const char *query_end = query_view.data() + query_view.size() + end_clipping;
    trim_end_clipping();
    auto it = cigar.ops.rbegin();
    size_t cigar_offset = 0;
    auto s_it = sequence.rbegin();
    auto node_it = nodes.rbegin();
    auto consume_ref = [&]() {
        --n;
        ++s_it;
        if (node_it + 1 < nodes.rend()) ++node_it;
        else *this = Alignment();
    };
    while (n || (trim_excess_insertions && it != cigar.ops.rend() && it->first == MGX_OP_INSERTION)) {
        if (it == cigar.ops.rend()) { *this = Alignment(); return 0; }
        switch (it->first) {
            case MGX_OP_MATCH:
            case MGX_OP_MISMATCH:
                score -= config.score_matrix[(uint8_t)query_view.back() & 127][(uint8_t)*s_it & 127];
                query_view.remove_suffix(1);
                consume_ref();
                if (empty()) return 0;
                break;
            case MGX_OP_INSERTION:
                score -= it->second - cigar_offset == 1 ? config.gap_opening_penalty : config.gap_extension_penalty;
                query_view.remove_suffix(1);
                break;
            case MGX_OP_DELETION:
                score -= it->second - cigar_offset == 1 ? config.gap_opening_penalty : config.gap_extension_penalty;
                consume_ref();
                if (empty()) return 0;
                break;
            default:
                throw std::runtime_error("trimming chains not supported");
        }
        ++cigar_offset;
        if (cigar_offset == it->second) { ++it; cigar_offset = 0; }
    }
    if (!end_clipping && (cigar_offset || it.base() != cigar.ops.end())) score -= config.right_end_bonus;
    nodes.erase(node_it.base(), nodes.end());
    sequence.erase(s_it.base(), sequence.end());
    it->second -= cigar_offset;
    cigar.ops.erase(it.base(), cigar.ops.end());
    extend_query_end(query_end);
    return cigar_offset;
}

void Alignment::splice_with_unknown(Alignment &&other, size_t num_unknown, size_t node_overlap, const mgx_config &config) {
    // alignment.cpp:1048-1152: `other` follows this alignment after num_unknown characters the graph does not hold
    if (other.offset) { *this = Alignment(); return; }                  // "Can't splice in sub-k alignment"
    ptrdiff_t overlap = (ptrdiff_t)(get_clipping() + query_view.size()) - (ptrdiff_t)other.get_clipping();
    other.trim_clipping();
    const score_t go = config.gap_opening_penalty, ge = config.gap_extension_penalty;
    if (overlap <= 0) {
        // a gap between the alignments: characters missing from the graph (typically N)
        trim_end_clipping();
        size_t query_gap = (size_t)(-overlap);
        if (query_gap > num_unknown) {
            cigar.append(MGX_OP_INSERTION, (uint32_t)(query_gap - num_unknown));
            score += go + (score_t)(query_gap - num_unknown - 1) * ge;
            query_view = std::string_view(query_view.data(), query_view.size() + query_gap - num_unknown);
            query_gap = num_unknown;
        }
        const char *start = other.query_view.data() - query_gap;
        if (query_gap) {
            other.cigar.ops.insert(other.cigar.ops.begin(), Cigar::value_type(MGX_OP_MISMATCH, (uint32_t)query_gap));
            other.score += score_sequences(config, std::string_view(start, query_gap), std::string(num_unknown, '$'));
        }
        if (num_unknown > query_gap) {
            other.cigar.ops.insert(other.cigar.ops.begin(), Cigar::value_type(MGX_OP_DELETION, (uint32_t)(num_unknown - query_gap)));
            other.score += go + (score_t)(num_unknown - query_gap - 1) * ge;
        }
        other.cigar.ops.insert(other.cigar.ops.begin(),
                               Cigar::value_type(MGX_OP_NODE_INSERTION, (uint32_t)(node_overlap + num_unknown - other.offset)));
        other.query_view = std::string_view(start, other.query_view.size() + query_gap);
        other.score += go + (score_t)(node_overlap + num_unknown - other.offset - 1) * ge;
    } else {
        // the read has fewer copies of a repeat than the graph at a gap of the graph: the deletion is rebuilt
        const std::vector<node_t> nodes_before = nodes;
        const std::string seq_before = sequence;
        trim_query_suffix((size_t)overlap, config, false);
        if (empty()) { *this = Alignment(); return; }                   // "Not enough nodes in this alignment for splicing"
        overlap = (ptrdiff_t)(seq_before.size() - sequence.size());
        trim_end_clipping();
        if (overlap) {
            cigar.ops.emplace_back(MGX_OP_DELETION, (uint32_t)overlap);
            nodes.insert(nodes.end(), nodes_before.end() - overlap, nodes_before.end());
            sequence += seq_before.substr(seq_before.size() - overlap);
        }
        other.cigar.ops.insert(other.cigar.ops.begin(), Cigar::value_type(MGX_OP_DELETION, (uint32_t)num_unknown));
        other.cigar.ops.insert(other.cigar.ops.begin(), Cigar::value_type(MGX_OP_NODE_INSERTION, (uint32_t)(node_overlap + num_unknown)));
        other.score += go * 2 + (score_t)(node_overlap + num_unknown - 1 + overlap + num_unknown - 1) * ge;
    }
    other.sequence = std::string(num_unknown, '$') + other.sequence;
    other.nodes.insert(other.nodes.begin(), node_overlap + num_unknown - other.offset, NPOS);
    other.offset = node_overlap;
    for (Tuple &tuple : other.label_coordinates) for (int64_t &c : tuple) c -= (int64_t)num_unknown;
    append(std::move(other));
}

std::string Alignment::format_coords(const std::vector<std::string> &label_names) const {
    // alignment.cpp:20-37: "<label>:<start>-<end>[:<start>-<end>...][;...]", 1-based inclusive
    std::string out;
    for (size_t i = 0; i < label_coordinates.size(); ++i) {
        if (i) out += ";";
        out += label_names.at(label_columns[i]);
        for (int64_t coord : label_coordinates[i])
            out += ":" + std::to_string(coord + 1) + "-" + std::to_string(coord + (int64_t)sequence.size());
    }
    return out;
}

std::string Alignment::format_coords(const std::vector<std::vector<std::string>> &headers,
                                     const std::vector<std::vector<uint64_t>> &kmer_counts, size_t k) const {
    // alignment.cpp:39-92; CoordToHeader::map_single_coord (annotation/coord_to_header.cpp): the sequence of a column a global
    // k-mer coordinate falls into (prefix sums of the k-mer counts) and the coordinate within it
    if (label_coordinates.empty()) return "";
    typedef std::pair<Label, size_t> Key;                              // (column, sequence)
    std::vector<std::pair<Key, std::vector<std::pair<uint64_t, uint64_t>>>> seq_ranges;     // VectorMap: insertion order
    auto ranges_of = [&](const Key &key) -> std::vector<std::pair<uint64_t, uint64_t>> & {
        for (auto &e : seq_ranges) if (e.first == key) return e.second;
        seq_ranges.emplace_back(key, std::vector<std::pair<uint64_t, uint64_t>>());
        return seq_ranges.back().second;
    };
    const uint64_t L = sequence.size();
    for (size_t i = 0; i < label_columns.size(); ++i) {
        const Label col = label_columns[i];
        const std::vector<uint64_t> &counts = kmer_counts.at(col);
        const size_t n_seqs = counts.size();
        for (int64_t coord : label_coordinates[i]) {
            size_t seq_id = 0;
            uint64_t local = (uint64_t)coord;
            while (seq_id < n_seqs && local >= counts[seq_id]) { local -= counts[seq_id]; ++seq_id; }
            uint64_t remaining = L;
            while (remaining) {
                if (seq_id >= n_seqs) break;                            // (past the last indexed sequence: truncated)
                const uint64_t nt_len = counts[seq_id] + k - 1;
                const uint64_t span = std::min(remaining, nt_len - local);
                ranges_of(Key(col, seq_id)).emplace_back(local, local + span - 1);
                remaining -= span;
                ++seq_id;
                local = 0;
            }
        }
    }
    std::string out;
    for (size_t e = 0; e < seq_ranges.size(); ++e) {
        if (e) out += ";";
        out += headers.at(seq_ranges[e].first.first).at(seq_ranges[e].first.second);
        for (const auto &r : seq_ranges[e].second) out += ":" + std::to_string(r.first + 1) + "-" + std::to_string(r.second + 1);
    }
    return out;
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
            if (has_coordinates()) {
                // (:168-181) labels and coordinates of a row, stored side by side: label_coords_ of a node = one Tuple per label,
                // in the order of its label set (BASIC graphs only: the constructor drops coordinates for the other modes, :26-33)
                CoordinateSet cs;
                for (auto &lt : annotation_.get_row_tuples(queued_rows[x])) cs.push_back(std::move(lt.second));
                node_coords_[queued_nodes[x]] = std::move(cs);
            }
            size_t label_i = cache_column_set(std::move(rows[x]));
            node_to_cols_[queued_nodes[x]] = label_i;    // push_node_labels: BASIC and the canonical wrapper both key by the base node
            if (graph_.mode == CANONICAL && !canonical_) {
                // (:150-158) a CANONICAL-mode graph: the queued node, and its representative if that has no entry yet
                const node_t base_node = (node_t)(queued_rows[x] + 1);
                if (base_node != queued_nodes[x]) node_to_cols_.try_emplace(base_node, label_i);
            }
        }
    }

    bool has_coordinates() const { return annotation_.has_coordinates && graph_.mode == BASIC && !canonical_; }
    // get_labels_and_coords (:195-217): nullptrs for a node that was never fetched; the coordinates of a node without labels
    // (a dummy node) are an empty set
    std::pair<const Columns *, const CoordinateSet *> get_labels_and_coords(node_t node) const {
        std::pair<const Columns *, const CoordinateSet *> ret(get_labels(node), nullptr);
        if (ret.first && has_coordinates()) {
            auto it = node_coords_.find(node);
            ret.second = it == node_coords_.end() ? &empty_coords_ : &it->second;
        }
        return ret;
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
    std::unordered_map<node_t, CoordinateSet> node_coords_;       // label_coords_ (hpp:84), keyed by node instead of by map position
    const CoordinateSet empty_coords_;
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
    const Graph &graph() const { return *base_; }
    const mgx_config &config() const { return config_; }
    size_t num_extensions() const { return num_extensions_; }
    size_t num_explored_nodes() const { return explored_nodes_previous_ + conv_checker_.size(); }

    // SeedFilteringExtender::get_extensions (extender hpp:38-51)
    // target_length / target_node (hpp:34-45): stop once target_length nucleotides have been aligned to and backtrack from
    // target_node only (align_connect); trim_query_suffix: the window ends that many characters before the query does;
    // added_xdrop: on top of the configured x-drop (saturating)
    std::vector<Alignment> get_extensions(const Alignment &seed, score_t min_path_score, bool force_fixed_seed,
                                          size_t target_length = 0, node_t target_node = NPOS, bool trim_offset_after_extend = true,
                                          size_t trim_query_suffix = 0, score_t added_xdrop = 0) {
        seed_ = &seed;                                   // set_seed (:90-98)
        explored_nodes_previous_ += conv_checker_.size();
        conv_checker_.clear();
        if (ab_) {
            // LabeledExtender::set_seed (aligner_labeled.cpp:139-174): the first node of the seed has already been flushed; the
            // seed's labels are what backtracking still has to account for
            last_flushed_table_i_ = 1;
            remaining_labels_i_ = ab_->cache_column_set(Columns(seed.label_columns));
            node_labels_.assign(1, remaining_labels_i_);
            // base_coords_: the coordinates of the seed's first k-mer — of its last one when the extension runs backwards
            base_coords_ = seed.label_coordinates;
            if (base_coords_.size()) {
                if (view_.rc) {
                    for (Tuple &coords : base_coords_) for (int64_t &c : coords) c += (int64_t)seed.nodes.size() - 1 - (int64_t)seed.offset;
                } else if (seed.offset) {
                    for (Tuple &coords : base_coords_) for (int64_t &c : coords) c -= (int64_t)seed.offset;
                }
            }
        }
        return extend(min_path_score, force_fixed_seed, target_length, target_node, trim_offset_after_extend, trim_query_suffix, added_xdrop);
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
    CoordinateSet base_coords_;                          // aligner_labeled.hpp:100
    std::vector<score_t> scores_reached_;                // extender hpp:156 (global_xdrop == false)
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
        if (ab_->get_labels_and_coords(node).second) {
            // (:245-300) label AND coordinate consistency, with the seed as the basis: a child keeps the labels of `columns` in
            // which one of its coordinates lies `dist` behind a coordinate of the seed's first k-mer
            const int64_t dist = (int64_t)next_offset - (int64_t)view_.get_k() + 1;
            for (const auto &[next, c, score] : outgoing) {
                const Columns *base_labels = &seed_->label_columns;
                const CoordinateSet *base_coords = &base_coords_;
                auto [next_labels, next_coords] = ab_->get_labels_and_coords(next);
                if (!next_labels || !next_coords) throw std::logic_error("oracle: call_outgoing(): coordinates of a child were not fetched");
                if (view_.rc) { std::swap(base_labels, next_labels); std::swap(base_coords, next_coords); }   // backwards: the delta is negated
                Columns intersect_labels;
                auto col_it = columns.begin();
                const auto col_end = columns.end();
                bool stop = false;
                match_indexed_values(*base_labels, *base_coords, *next_labels, *next_coords,
                    [&](Label cc, const Tuple &coords, const Tuple &other_coords) {
                        if (stop) return;                             // (the reference leaves through an exception)
                        while (col_it != col_end && cc > *col_it) ++col_it;
                        if (col_it == col_end) { stop = true; return; }
                        if (cc < *col_it) return;
                        // overlap_with_diff (:22-44): some a in coords, b in other_coords with a + dist == b
                        size_t a = 0, b = 0;
                        bool hit = false;
                        while (a < coords.size() && b < other_coords.size()) {
                            if (coords[a] + dist == other_coords[b]) { hit = true; break; }
                            if (coords[a] + dist < other_coords[b]) ++a; else ++b;
                        }
                        if (hit) intersect_labels.push_back(cc);
                    });
                if (intersect_labels.size()) {
                    node_labels_.push_back(ab_->cache_column_set(std::move(intersect_labels)));
                    callback(next, c, score);
                }
            }
            return;
        }
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
    // call_alignments (hpp:200-214 / aligner_labeled.cpp:328-448); clipping: the window position the trace ended at
    void call_alignments(Alignment &&alignment, size_t clipping, std::vector<Alignment> &extensions) {
        auto call_alignment = [&]() {
            if (ab_ && label_diff_.size() && label_diff_.back() == kNannot) {
                label_diff_.pop_back();
                remaining_labels_i_ = ab_->cache_column_set(std::move(label_diff_));
                label_diff_ = Columns{};
            }
            extensions.emplace_back(std::move(alignment));
        };
        if (!ab_) { call_alignment(); return; }
        if (!ab_->has_coordinates()) {
            alignment.label_columns = std::move(label_intersection_);
            label_intersection_ = Columns{};
            call_alignment();
            return;
        }
        // (:361-447) the labels of label_intersection_ whose coordinates at the alignment's first and last node lie as far apart
        // as the alignment is long; the alignment's coordinates are those of its first nucleotide
        auto [base_labels, base_coords] = ab_->get_labels_and_coords(alignment.nodes.front());
        if (!base_labels || !base_coords) throw std::logic_error("oracle: call_alignments(): the first node's coordinates were not fetched");
        if (!clipping) base_labels = &seed_->label_columns;
        int64_t dist = (int64_t)alignment.nodes.size() - 1;
        if (!clipping) {
            base_coords = &seed_->label_coordinates;
            dist -= (int64_t)seed_->offset;
            if (view_.rc) dist = (int64_t)alignment.sequence.size() - (int64_t)seed_->sequence.size();
        }
        auto label_it = label_intersection_.begin();
        const auto label_end_it = label_intersection_.end();
        if (alignment.nodes.size() == 1) {
            auto it = base_labels->begin();
            const auto end = base_labels->end();
            auto c_it = base_coords->begin();
            while (label_it != label_end_it && it != end) {
                if (*label_it < *it) ++label_it;
                else if (*label_it > *it) { ++it; ++c_it; }
                else {
                    alignment.label_columns.emplace_back(*it);
                    alignment.label_coordinates.emplace_back(*c_it);
                    ++it; ++c_it; ++label_it;
                }
            }
        } else {
            auto [cur_labels, cur_coords] = ab_->get_labels_and_coords(alignment.nodes.back());
            if (!cur_labels || !cur_coords) throw std::logic_error("oracle: call_alignments(): the last node's coordinates were not fetched");
            if (view_.rc) { std::swap(cur_labels, base_labels); std::swap(cur_coords, base_coords); }
            bool stop = false;
            match_indexed_values(*base_labels, *base_coords, *cur_labels, *cur_coords,
                [&](Label c, const Tuple &coords, const Tuple &other_coords) {
                    if (stop) return;
                    while (label_it != label_end_it && c > *label_it) ++label_it;
                    if (label_it == label_end_it) { stop = true; return; }
                    if (c < *label_it) return;
                    Tuple overlap;
                    set_intersection_delta(coords, other_coords, &overlap, dist);
                    if (overlap.size()) {
                        alignment.label_columns.emplace_back(c);
                        alignment.label_coordinates.emplace_back(std::move(overlap));
                    }
                });
        }
        if (alignment.label_coordinates.empty()) return;
        if (view_.rc && alignment.offset) {
            for (Tuple &coords : alignment.label_coordinates) for (int64_t &c : coords) c += (int64_t)alignment.offset;
        }
        call_alignment();
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

    std::vector<Alignment> extend(score_t min_path_score, bool force_fixed_seed, size_t target_length, node_t target_node,
                                  bool trim_offset_after_extend, size_t trim_query_suffix, score_t added_xdrop) {
        // aligner_extender_methods.cpp:412-772
        ++num_extensions_;
        if (wc_) ++wc_->n_extensions;
        min_path_score = std::max(0, min_path_score);
        table.clear();      // std::vector::clear keeps capacity: table_cap_ carries over between extensions
        prev_starts.clear();

        score_t xdrop = std::min(config_.xdrop, INT32_MAX - added_xdrop) + added_xdrop;      // saturating (:429-430)
        xdrop_cutoffs_.assign(1, std::make_pair(size_t(0), std::max(-xdrop, NINF + 1)));
        const bool global_xdrop = config_.global_xdrop != 0;
        if (!global_xdrop) scores_reached_.assign(1, 0);

        size_t start = seed_->get_clipping();
        std::string_view window(seed_->query_view.data(),
                                query_.data() + query_.size() - seed_->query_view.data() - trim_query_suffix);
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
                    double node_counter = global_xdrop ? (double)table.size() : (double)((ssize_t_)next_offset - seed_offset);   // (:521-523)
                    if (col.S[col.max_pos - col.trim] < best_score) {
                        if (node_counter / window.size() >= config_.max_nodes_per_seq_char) {
                            if (global_xdrop) {                    // (per-branch x-drop: only this branch stops, :531-536)
                                queue = std::priority_queue<TableIt>();
                                next_nodes.clear();
                            }
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
                    // (:578-586) global_xdrop == false: every child of a fork gets a cut-off of its own, starting from its parent's
                    const bool forked_xdrop = !global_xdrop && outgoing.size() > 1;
                    size_t xdrop_cutoffs_sizediff = 0;
                    if (forked_xdrop) {
                        const size_t cap_before = xdrop_cutoffs_.capacity();
                        xdrop_cutoffs_.emplace_back(table.size(), prev_xdrop_cutoff);
                        xdrop_cutoffs_sizediff = xdrop_cutoffs_.capacity() - cap_before;
                    }
                    size_t table_sizediff = table_cap_table();
                    table_emplace(create_column(end - begin, next, i, c, static_cast<ssize_t_>(next_offset),
                                                begin, begin, forked_xdrop ? xdrop_cutoffs_.size() - 1 : prev_xdrop_cutoff_i, score));
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

                    size_t scores_reached_sizediff = 0;
                    bool scores_reached_cutoff = true;
                    if (!global_xdrop) {                           // (:637-641)
                        const size_t cap_before = scores_reached_.capacity();
                        scores_reached_.resize(cur.S.size() + cur.trim + 1, NINF);
                        scores_reached_sizediff = scores_reached_.capacity() - cap_before;
                    }
                    for (size_t j = 0; j < cur.S.size(); ++j, ++cur_offset) {
                        if (cur.S[j] != NINF) min_cell_score_ = std::min(min_cell_score_, cur.S[j]);
                        if (std::make_pair(cur.S[j], std::abs(cur.max_pos - diag_i))
                                > std::make_pair(cur.S[cur.max_pos - begin], std::abs(cur_offset - diag_i))) {
                            cur.max_pos = j + begin;
                        }
                        if (!global_xdrop) {                       // (:652-655; the flag is overwritten cell by cell, as there)
                            scores_reached_[cur.trim + j] = std::max(scores_reached_[cur.trim + j], cur.S[j]);
                            scores_reached_cutoff = (cur.S[j] >= scores_reached_[cur.trim + j] * config_.rel_score_cutoff);
                        }
                        if (!has_extension && scores_reached_cutoff && cur.S[j] + partial_sums[j] >= extension_cutoff) has_extension = true;
                    }

                    score_t max_val = cur.S[cur.max_pos - cur.trim];
                    // (:676-681) a target length: every column up to it goes on, none beyond it
                    if (static_cast<size_t>(cur.offset - seed_offset) < target_length + 1) has_extension = true;
                    else if (target_length) has_extension = false;

                    if (!in_seed && max_val < xdrop_cutoff) { pop(table.size() - 1); if (forked_xdrop) xdrop_cutoffs_.pop_back(); continue; }
                    if (!in_seed && !has_extension) { pop(table.size() - 1); if (forked_xdrop) xdrop_cutoffs_.pop_back(); continue; }

                    table_sizediff = table_cap_table() - table_sizediff;
                    table_size_bytes_ += kSizeofColumn * table_sizediff
                        + (cur.S.capacity() + cur.E.capacity() + cur.F.capacity()) * sizeof(score_t)
                        + sizeof(score_t) * scores_reached_sizediff + sizeof(std::pair<size_t, score_t>) * xdrop_cutoffs_sizediff;

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
        auto extensions = backtrack(min_path_score, window, trim_query_suffix ? 0 : config_.right_end_bonus, tips, k, target_node);
        if (trim_offset_after_extend) for (auto &ext : extensions) ext.trim_offset();
        return extensions;
    }

    size_t table_cap_table() const { return table_cap_; }

    std::vector<Alignment> backtrack(score_t min_path_score, std::string_view window, score_t right_end_bonus,
                                     const std::vector<size_t> &tips, size_t k, node_t target_node) {
        // aligner_extender_methods.cpp:800-1034
        if (ab_) flush();                               // LabeledExtender::backtrack (aligner_labeled.hpp:32-43)
        std::vector<Alignment> extensions;
        size_t seed_clipping = seed_->get_clipping();
        ssize_t_ seed_offset = static_cast<ssize_t_>(seed_->offset) - 1;
        ssize_t_ k_minus_1 = k - 1;
        ssize_t_ last_pos = window.size();
        ssize_t_ seed_dist = std::max(k, seed_->sequence.size()) - 1;
        score_t min_start_score = target_node ? NINF : min_path_score;
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
                if (target_node) {
                    // (:840-845) a target node: only its cells reached by a match (or tips) start a trace
                    if (is_tip || (col.node == target_node
                            && col.S[pos] == par.S.at_ub(pos_p) + col.score + profile_score_[s][seed_clipping + start_pos])) {
                        indices.emplace_back(col.S[pos] + end_bonus, -std::abs(start_pos - col.offset + seed_offset),
                                             -static_cast<ssize_t_>(i), start_pos);
                    }
                } else if (col.S[pos] + end_bonus >= min_start_score) {
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
            if (!target_node || is_tip) check_and_add_pos(table[i].max_pos, is_tip);
            if ((ssize_t_)(table[i].S.size() + table[i].trim) == (ssize_t_)window.size() + 1
                    && (target_node || table[i].max_pos != last_pos)) {
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
                                                        path, seq, score, align_offset, extra_score), (size_t)pos, extensions);
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

// =============================================================================================
// Seed chaining (A/aligner_chainer.cpp:21-542) and chain extension (A/dbg_aligner.cpp:63-250,388-529): what LabeledAligner
// runs when its annotation carries coordinates (aligner_labeled.cpp:457-462).  Round 6; SURVEY 8 row f3.
// =============================================================================================
typedef std::vector<std::pair<Alignment, int64_t>> Chain;              // aligner_chainer.hpp: (alignment, coordinate distance to the one before)
struct early_term {};

// get_num_char_matches_in_seeds (alignment.hpp:100-127) over the alignments of a chain (same quirk as the Seed form below)
size_t get_num_char_matches_in_chain(const Chain &chain) {
    size_t num_matching = 0, last_q_end = 0;
    for (size_t i = 0; i < chain.size(); ++i) {
        const Alignment &aln = chain[i].first;
        if (aln.empty()) continue;
        size_t q_begin = aln.get_clipping(), q_end = q_begin + aln.query_view.size();
        if (q_end > last_q_end) {
            num_matching += q_end - q_begin;
            if (q_begin < last_q_end) num_matching -= last_q_end - q_begin;
        }
        if (aln.offset) i = chain.size() - 1;
        last_q_end = q_end;
    }
    return num_matching;
}

struct TableElem {                                                      // aligner_chainer.cpp:23-36
    Label label;
    int64_t coordinate;
    int32_t seed_clipping, seed_end;
    score_t chain_score;
    uint32_t current_seed_index;
};
inline bool table_elem_greater(const TableElem &a, const TableElem &b) {          // :38-41
    return std::tie(a.label, a.coordinate, a.seed_clipping, a.seed_end) > std::tie(b.label, b.coordinate, b.seed_clipping, b.seed_end);
}
constexpr uint32_t kNid = std::numeric_limits<uint32_t>::max();

// chain_seeds (:341-542).  The reference's loop over j is AVX2 code, eight anchors at a time with gathers that reach past the
// table's end (masked); lane by lane it computes what is written here: the anchors of one label in decreasing coordinate order,
// each tried as the successor-in-the-table (= predecessor on the query) of the 64 anchors behind it.  The gap cost is FLOAT
// arithmetic — linear + 0.5 * log2 — rounded to the nearest integer (cvtps_epi32: the current rounding mode, nearest-even),
// not truncated as in the scalar version the source keeps under `#if 0`.
struct ChainTables { std::vector<TableElem> dp_table; std::vector<uint32_t> backtrace; size_t num_seeds = 0, num_nodes = 0; };
void chain_dp(const mgx_config &config, int64_t query_size, std::map<Label, size_t> &label_sizes, std::vector<TableElem> *dp_table,
              std::vector<uint32_t> *backtrace);
ChainTables chain_seeds(const mgx_config &config, std::string_view query, std::vector<Seed> &seeds) {
    ChainTables T;
    if (seeds.empty()) return T;
    for (const Seed &a : seeds)
        if (a.label_coordinates.empty()) throw std::runtime_error("Chaining only supported for seeds with coordinates");
    const int64_t query_size = (int64_t)query.size();
    std::reverse(seeds.begin(), seeds.end());
    std::map<Label, size_t> label_sizes;
    for (size_t i = 0; i < seeds.size(); ++i) {
        for (size_t j = 0; j < seeds[i].label_coordinates.size(); ++j) {
            const Label c = seeds[i].label_columns[j];
            const Tuple &coords = seeds[i].label_coordinates[j];
            const size_t take = std::min<size_t>(coords.size(), config.max_num_seeds_per_locus);
            for (size_t t = 0; t < take; ++t) {                         // from the largest coordinate down
                const int64_t coord = coords[coords.size() - 1 - t];
                ++label_sizes[c];
                T.dp_table.push_back(TableElem{ c, coord, (int32_t)seeds[i].clipping,
                                                (int32_t)(seeds[i].clipping + seeds[i].query_view.size()),
                                                (score_t)seeds[i].query_view.size(), (uint32_t)i });
            }
        }
        seeds[i].label_columns = Columns{};
        seeds[i].label_coordinates = CoordinateSet{};
    }
    T.num_seeds = T.dp_table.size();
    chain_dp(config, query_size, label_sizes, &T.dp_table, &T.backtrace);
    return T;
}

// the sort and the DP of chain_seeds (:383-539) over a filled table
void chain_dp(const mgx_config &config, int64_t query_size, std::map<Label, size_t> &label_sizes, std::vector<TableElem> *dp_table,
              std::vector<uint32_t> *backtrace) {
    struct { std::vector<TableElem> &dp_table; std::vector<uint32_t> &backtrace; } T{ *dp_table, *backtrace };
    T.backtrace.assign(T.dp_table.size(), kNid);
    if (T.dp_table.empty()) return;
    // "sort seeds by label, then by decreasing reference coordinate"
    std::sort(T.dp_table.begin(), T.dp_table.end(), table_elem_greater);
    const size_t bandwidth = 65;
    // "scoring function derived from minimap2": float sl = static_cast<float>(min_seed_length) * 0.01 (a double product, narrowed)
    const float sl = (float)((double)(float)config.min_seed_length * 0.01);
    std::vector<TableElem> &dp = T.dp_table;
    size_t cur_label_end = 0, i = 0;
    while (cur_label_end < dp.size()) {
        cur_label_end += label_sizes[dp[i].label];
        for ( ; i < cur_label_end; ++i) {
            const int64_t prev_coord = dp[i].coordinate;
            const int32_t prev_clipping = dp[i].seed_clipping;
            const score_t prev_score = dp[i].chain_score;
            if (!prev_clipping) continue;
            const size_t it_end = std::min(bandwidth, cur_label_end - i) + i;
            const int64_t coord_cutoff = prev_coord - query_size;
            for (size_t j = i + 1; j < it_end; ++j) {
                TableElem &e = dp[j];
                if (coord_cutoff > e.coordinate) break;
                const int32_t dist = prev_clipping - e.seed_clipping;
                const int32_t coord_dist = (int32_t)(prev_coord - e.coordinate);          // (epi64 -> epi32: the low half)
                if (dist > 0 && std::max(dist, coord_dist) < (int32_t)query_size) {
                    const score_t match = std::min(std::min(dist, coord_dist), e.seed_end - e.seed_clipping);
                    score_t cur_score = prev_score + match;
                    const int32_t coord_diff = std::abs(coord_dist - dist);
                    if (coord_diff > 0) {
                        const float linear_penalty = (float)coord_diff * sl;
                        const float log_penalty = std::log2((float)(coord_diff + 1)) * 0.5f;
                        cur_score -= (score_t)std::lrintf(linear_penalty + log_penalty);      // cvtps_epi32
                    }
                    if (cur_score >= e.chain_score) {
                        e.chain_score = cur_score;
                        T.backtrace[j] = (uint32_t)i;
                    }
                }
            }
        }
    }
}

struct ChainHash {                                                       // :51-62
    std::size_t operator()(const Chain &chain) const {
        uint64_t hash = 0;
        for (const auto &[aln, dist] : chain) {
            for (node_t node : aln.nodes) hash ^= node + 0x9e3779b9 + (hash << 6) + (hash >> 2);
            hash ^= (uint64_t)dist + 0x9e3779b9 + (hash << 6) + (hash >> 2);
        }
        return hash;
    }
};

// call_seed_chains_both_strands (:64-339).  Chains of equal score are collected in a std::unordered_multiset and handed out in ITS
// iteration order, as in the reference (the same container of the same standard library, filled in the same order).
std::pair<size_t, size_t> call_seed_chains_both_strands(std::string_view forward, std::string_view reverse, const mgx_config &config,
                                                        std::vector<Seed> &&fwd_seeds, std::vector<Seed> &&bwd_seeds,
                                                        const std::function<void(Chain &&, score_t)> &callback,
                                                        const std::function<bool(Label)> &skip_column) {
    auto useless = [](const Seed &a) { return a.empty() || a.label_columns.empty(); };
    fwd_seeds.erase(std::remove_if(fwd_seeds.begin(), fwd_seeds.end(), useless), fwd_seeds.end());
    bwd_seeds.erase(std::remove_if(bwd_seeds.begin(), bwd_seeds.end(), useless), bwd_seeds.end());
    if (fwd_seeds.empty() && bwd_seeds.empty()) return { 0, 0 };
    std::vector<Seed> both_seeds[2] = { std::move(fwd_seeds), std::move(bwd_seeds) };
    ChainTables tables[2] = { chain_seeds(config, forward, both_seeds[0]), chain_seeds(config, reverse, both_seeds[1]) };
    const size_t num_seeds = tables[0].num_seeds + tables[1].num_seeds, num_nodes = tables[0].num_nodes + tables[1].num_nodes;
    // chains by backtracking, best chain score first
    std::vector<std::tuple<score_t, uint32_t, ptrdiff_t>> starts;
    std::vector<bool> both_used[2] = { std::vector<bool>(tables[0].dp_table.size(), false), std::vector<bool>(tables[1].dp_table.size(), false) };
    for (uint32_t s2 = 0; s2 < 2; ++s2)
        for (size_t x = 0; x < tables[s2].dp_table.size(); ++x) starts.emplace_back(tables[s2].dp_table[x].chain_score, s2, -(ptrdiff_t)x);
    if (starts.empty()) return { num_seeds, num_nodes };
    std::sort(starts.begin(), starts.end(), std::greater<std::tuple<score_t, uint32_t, ptrdiff_t>>());
    score_t last_chain_score = std::numeric_limits<score_t>::min();
    std::unordered_multiset<Chain, ChainHash> chains;
    auto flush_chains = [&]() {
        auto it = chains.begin();
        Chain last_chain = *it;
        for (++it; it != chains.end(); ++it) {
            const Chain &chain = *it;
            if (chain != last_chain) {
                callback(std::move(last_chain), last_chain_score);
                last_chain = *it;
                continue;
            }
            // the same seeds as the chain before: their label / coordinate sets are merged (:137-180)
            for (size_t x = 0; x < chain.size(); ++x) {
                Columns columns;
                Alignment &mine = last_chain[x].first;
                const Alignment &theirs = chain[x].first;
                if (theirs.label_coordinates.size()) {
                    CoordinateSet coord_union;
                    size_t a = 0, b = 0;
                    while (a < mine.label_columns.size() || b < theirs.label_columns.size()) {      // match_indexed_values with both diffs
                        if (a == mine.label_columns.size()) {
                            columns.push_back(theirs.label_columns[b]); coord_union.push_back(theirs.label_coordinates[b]); ++b;
                        } else if (b == theirs.label_columns.size() || mine.label_columns[a] < theirs.label_columns[b]) {
                            columns.push_back(mine.label_columns[a]); coord_union.push_back(mine.label_coordinates[a]); ++a;
                        } else {
                            if (mine.label_columns[a] == theirs.label_columns[b]) {
                                columns.push_back(mine.label_columns[a]);
                                coord_union.emplace_back();
                                std::set_union(mine.label_coordinates[a].begin(), mine.label_coordinates[a].end(),
                                               theirs.label_coordinates[b].begin(), theirs.label_coordinates[b].end(),
                                               std::back_inserter(coord_union.back()));
                                ++a;
                            } else {
                                columns.push_back(theirs.label_columns[b]); coord_union.push_back(theirs.label_coordinates[b]);
                            }
                            ++b;
                        }
                    }
                    std::swap(mine.label_coordinates, coord_union);
                } else {
                    std::set_union(mine.label_columns.begin(), mine.label_columns.end(), theirs.label_columns.begin(),
                                   theirs.label_columns.end(), std::back_inserter(columns));
                }
                std::swap(mine.label_columns, columns);
            }
        }
        callback(std::move(last_chain), last_chain_score);
        chains.clear();
    };
    for (const auto &[chain_score, j, neg_i] : starts) {
        std::vector<bool> &used = both_used[j];
        uint32_t i = (uint32_t)(-neg_i);
        if (used[i]) continue;
        const std::vector<TableElem> &dp_table = tables[j].dp_table;
        const std::vector<Seed> &seeds = both_seeds[j];
        const std::vector<uint32_t> &seed_backtrace = tables[j].backtrace;
        std::vector<std::pair<Seed, int64_t>> chain_seeds_v;
        while (i != kNid) {
            const TableElem &e = dp_table[i];
            if (skip_column(e.label)) break;
            used[i] = true;
            chain_seeds_v.emplace_back(seeds[e.current_seed_index], e.coordinate);
            // (has_labels: the aligner is a LabeledAligner) one label, one coordinate per chain seed
            chain_seeds_v.back().first.label_columns.assign(1, e.label);
            chain_seeds_v.back().first.label_coordinates.assign(1, Tuple(1, e.coordinate));
            i = seed_backtrace[i];
        }
        if (chain_seeds_v.empty()) continue;
        // overlapping seeds whose query and coordinate shifts agree are merged (:214-243)
        for (size_t x = chain_seeds_v.size() - 1; x > 0; --x) {
            Seed &cur_seed = chain_seeds_v[x].first;
            Seed &prev_seed = chain_seeds_v[x - 1].first;
            const size_t prev_end = prev_seed.clipping + prev_seed.query_view.size();
            if (prev_end > cur_seed.clipping) {
                const size_t coord_dist = (size_t)(cur_seed.label_coordinates[0][0] + (int64_t)cur_seed.query_view.size()
                                                   - prev_seed.label_coordinates[0][0] - (int64_t)prev_seed.query_view.size());
                const size_t dist = cur_seed.clipping + cur_seed.query_view.size() - prev_end;
                if (dist == coord_dist && cur_seed.nodes.size() >= dist) {
                    prev_seed.expand(std::vector<node_t>(cur_seed.nodes.end() - dist, cur_seed.nodes.end()));
                    cur_seed = Seed();
                }
            }
        }
        auto drop_empty = [&]() {
            chain_seeds_v.erase(std::remove_if(chain_seeds_v.begin(), chain_seeds_v.end(),
                                               [](const std::pair<Seed, int64_t> &a) { return a.first.empty(); }), chain_seeds_v.end());
        };
        drop_empty();
        // "Drop coord-redundant seeds": two consecutive seeds with the same reference coordinate — the longer one stays (:248-274)
        for (size_t x = chain_seeds_v.size(); x-- > 1;) {
            Seed &prev_seed = chain_seeds_v[x - 1].first;
            Seed &cur_seed = chain_seeds_v[x].first;
            if (prev_seed.empty() || cur_seed.empty()) continue;
            if (chain_seeds_v[x - 1].second == chain_seeds_v[x].second) {
                if (prev_seed.query_view.size() <= cur_seed.query_view.size()) prev_seed = Seed(); else cur_seed = Seed();
            }
        }
        drop_empty();
        for (size_t x = chain_seeds_v.size() - 1; x > 0; --x) chain_seeds_v[x].second -= chain_seeds_v[x - 1].second;
        chain_seeds_v[0].second = 0;
        if (chain_seeds_v[0].first.label_columns.empty()) continue;
        Chain chain;
        chain.reserve(chain_seeds_v.size());
        for (const auto &c : chain_seeds_v) chain.emplace_back(Alignment(c.first, config), c.second);
        if (chains.empty()) { chains.emplace(std::move(chain)); last_chain_score = chain_score; continue; }
        if (chain_score == last_chain_score) { chains.emplace(std::move(chain)); continue; }
        flush_chains();
        chains.emplace(std::move(chain));
        last_chain_score = chain_score;
    }
    flush_chains();
    return { num_seeds, num_nodes };
}

// split_seed (dbg_aligner.cpp:63-100): all but the last k characters of the alignment, and its last k-mer
std::pair<Alignment, Alignment> split_seed(size_t k, const mgx_config &config, const Alignment &alignment) {
    if (alignment.sequence.size() < k * 2 || std::find(alignment.nodes.begin(), alignment.nodes.end(), NPOS) != alignment.nodes.end())
        return std::make_pair(Alignment(), alignment);
    std::pair<Alignment, Alignment> ret(alignment, alignment);
    ret.first.trim_reference_suffix(k, config, false);
    size_t trim_nodes = k;                               // "ensure that there's no DELETION at the splice point"
    if (ret.first.size()) {
        auto it = ret.first.cigar.ops.rbegin();
        if (it->first == MGX_OP_CLIPPED) ++it;
        if (it->first == MGX_OP_DELETION) {
            const size_t n_del = it->second;
            trim_nodes += n_del;
            ret.first.trim_reference_suffix(n_del, config, false);
        }
    }
    ret.second.trim_reference_prefix(alignment.sequence.size() - trim_nodes, k - 1, config, true);
    return ret;
}

// align_connect (dbg_aligner.cpp:151-191): `first` is extended until it reaches the end of `second`; false: the connection
// failed, `first` went to partial_alignments and `second` takes its place
bool align_connect(size_t k, const mgx_config &config, Alignment &first, Alignment &second, int64_t coord_dist, Extender &extender,
                   std::vector<Alignment> &partial_alignments) {
    auto [left, next] = split_seed(k, config, first);
    coord_dist += (int64_t)second.sequence.size() + (int64_t)next.sequence.size() - (int64_t)first.sequence.size();
    auto extensions = extender.get_extensions(next, NINF /* min_path_score */, true /* force_fixed_seed */, (size_t)coord_dist /* target_length */,
                                              second.nodes.back() /* target_node */, false /* trim_offset_after_extend */,
                                              second.get_end_clipping() /* trim_query_suffix */, first.score - next.score /* added_xdrop */);
    if (extensions.size() && extensions[0].get_end_clipping() < first.get_end_clipping()) {
        left.splice(std::move(extensions[0]));
        std::swap(left, first);
        if (first.score >= second.score) return true;
    }
    partial_alignments.emplace_back(first);
    std::swap(first, second);
    return false;
}

// extend_chain (dbg_aligner.cpp:385-529): a full alignment from a chain — the query between consecutive chain seeds is aligned
// to the graph (align_connect); pieces that could not be connected are spliced with runs of unknown characters where their
// coordinates allow (splice_with_unknown); the best piece is extended to both ends of the query
void extend_chain(std::string_view query, std::string_view query_rc, Extender &extender, Chain &&chain, const CanonicalView *canon,
                  AnnotationBuffer *annotation_buffer, const std::function<void(Alignment &&)> &callback) {
    const Graph &graph = extender.graph();
    const mgx_config &config = extender.config();
    const size_t k = graph.get_k();
    LocalAlignmentLess less;
    Alignment cur = std::move(chain[0].first);
    Alignment best = cur;
    std::vector<Alignment> partial_alignments;
    std::vector<int64_t> coord_offsets{ 0 };
    int64_t coord_offset = 0;
    for (size_t i = 1; i < chain.size(); ++i) {
        coord_offset += chain[i].second;
        if (!align_connect(k, config, cur, chain[i].first, coord_offset, extender, partial_alignments)) {
            if (less(best, partial_alignments.back())) best = partial_alignments.back();
            coord_offsets.push_back(coord_offset);
            coord_offset = 0;
        }
        if (less(best, cur)) best = cur;
    }
    if (partial_alignments.size()) {
        partial_alignments.emplace_back(cur);
        Alignment *first = &partial_alignments[0];
        for (size_t i = 1; i < partial_alignments.size(); ++i) {
            Alignment &next = partial_alignments[i];
            if (next.sequence.size() < k) { first = &next; continue; }
            const int64_t num_unknown = coord_offsets[i] - (int64_t)first->sequence.size();
            if (num_unknown > 0 && (int64_t)next.get_clipping() > num_unknown) {
                Alignment merged = *first;
                Alignment next_fixed = next;
                next_fixed.trim_offset();
                merged.splice_with_unknown(std::move(next_fixed), (size_t)num_unknown, k - 1, config);
                if (merged.size()) std::swap(*first, merged);
                else { first = &next; continue; }
            } else {
                first = &next;
                continue;
            }
            if (less(best, *first)) best = *first;
        }
        if (less(best, *first)) best = *first;
    }
    if (best.get_end_clipping()) {                       // "Extending back"
        auto extensions = extender.get_extensions(best, NINF, true);
        if (extensions.size() && extensions[0].get_end_clipping() < best.get_end_clipping() && extensions[0].score > best.score)
            std::swap(best, extensions[0]);
    }
    best.trim_offset();
    if (best.get_clipping()) {                           // "Extending front": on the reverse-complement view of the graph
        const bool use_rcdbg = graph.mode != CANONICAL;
        GraphView rc_view{ &graph, use_rcdbg, canon };
        Alignment rev = best;
        rev.reverse_complement(rc_view, query_rc);
        if (rev.size() && rev.nodes.back()) {
            Extender extender_rc(graph, config, query_rc, nullptr, canon, annotation_buffer);
            extender_rc.set_graph(use_rcdbg);
            auto extensions = extender_rc.get_extensions(rev, NINF, true);
            if (extensions.size() && extensions[0].get_end_clipping() < rev.get_end_clipping()) {
                extensions[0].reverse_complement(rc_view, query);
                if (extensions[0].size()) std::swap(best, extensions[0]);
            }
        }
    }
    callback(std::move(best));
    for (Alignment &aln : partial_alignments) if (!aln.empty()) callback(std::move(aln));
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

void Annotation::annotate_kmer_coords(const Graph &graph, std::string_view sequence, size_t column, uint64_t start) {
    // annotated_dbg.cpp:192-233: map_to_nodes over the sequence, the coordinate counts every k-mer (found or not)
    if (sequence.size() < graph.get_k()) return;
    has_coordinates = true;
    uint64_t coord = start;
    for (node_t v : graph.map_to_nodes(sequence)) {
        if (v > 0) {
            set(v - 1, column);
            Tuple &t = coordinates[column][v - 1];
            t.insert(std::upper_bound(t.begin(), t.end(), (int64_t)coord), (int64_t)coord);
        }
        ++coord;
    }
}

std::vector<std::pair<Label, Tuple>> Annotation::get_row_tuples(uint64_t row) const {
    std::vector<std::pair<Label, Tuple>> out;
    for (size_t j = 0; j < columns.size(); ++j) {
        if (row >= n_rows || !get(row, j)) continue;
        auto it = coordinates[j].find(row);
        out.emplace_back((Label)j, it == coordinates[j].end() ? Tuple{} : it->second);
    }
    return out;
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
    if (config_.chain_alignments || config_.post_chain_alignments || config_.no_backtrack)
        throw std::runtime_error("oracle: chain_alignments set by the caller / post-chaining of labeled alignments / no_backtrack are out of scope");
    // (dbg_aligner.cpp:58-60 ran with the caller's chain_alignments == false: allow_left_trim stays as configured)
    // ... then LabeledAligner's (aligner_labeled.cpp:456-465): "do not use a global xdrop cutoff since we need separate cutoffs
    // for each label"; coordinates switch seed chaining on (AnnotationBuffer::has_coordinates: BASIC graphs only)
    if (annotation_.has_coordinates && graph_.mode == BASIC) {
        config_.global_xdrop = 0;
        config_.chain_alignments = 1;
    }
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
            auto [fetch_labels, fetch_coords] = ab.get_labels_and_coords(seed.nodes[0]);
            if (ab.has_coordinates()) {
                // matched_intersection (:594-610): the kept labels with the node's coordinates; a sub-k seed's coordinates are
                // those of its first nucleotide (:693-699)
                size_t a = 0, b = 0;
                while (a < fetch_labels->size() && b < labels.size()) {
                    if ((*fetch_labels)[a] < labels[b]) ++a;
                    else if ((*fetch_labels)[a] > labels[b]) ++b;
                    else { seed.label_columns.push_back((*fetch_labels)[a]); seed.label_coordinates.push_back((*fetch_coords)[a]); ++a; ++b; }
                }
                if (seed.offset) for (Tuple &tuple : seed.label_coordinates) for (int64_t &coord : tuple) coord += (int64_t)seed.offset;
            } else {
                std::set_intersection(fetch_labels->begin(), fetch_labels->end(), labels.begin(), labels.end(),
                                      std::back_inserter(seed.label_columns));
            }
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
            if (config_.chain_alignments) {
                // align_both_directions, the chaining branch (dbg_aligner.cpp:545-644): chains of seeds with consistent coordinates,
                // best first; each is extended seed to seed (extend_chain); a label is finished once the aggregator turns an
                // alignment of it away
                if (!annotation_buffer.has_coordinates()) throw std::runtime_error("Chaining only supported for seeds with coordinates");
                std::vector<Seed> fwd_chain_seeds = qs.fwd, bwd_chain_seeds = qs.rc;
                if (!(fwd_chain_seeds.empty() && bwd_chain_seeds.empty())) {
                    LabeledAggregator chain_aggregator(config_);
                    std::set<Label> all_columns, finished_columns;
                    for (const Seed &seed : fwd_chain_seeds) all_columns.insert(seed.label_columns.begin(), seed.label_columns.end());
                    for (const Seed &seed : bwd_chain_seeds) all_columns.insert(seed.label_columns.begin(), seed.label_columns.end());
                    try {
                        call_seed_chains_both_strands(this_query, reverse, config_, std::move(fwd_chain_seeds), std::move(bwd_chain_seeds),
                            [&](Chain &&chain, score_t) {
                                if (config_.num_alternative_paths <= 1 && finished_columns.size() == all_columns.size()) throw early_term();
                                const double exact_match_fraction = static_cast<double>(get_num_char_matches_in_chain(chain)) / this_query.size();
                                if (exact_match_fraction < config_.min_exact_match) throw early_term();
                                const bool rev = chain[0].first.orientation;
                                extend_chain(rev ? reverse : this_query, rev ? this_query : reverse, rev ? extender_rc : extender,
                                             std::move(chain), canon, &annotation_buffer,
                                             [&](Alignment &&aln) {
                                                 const Columns cur_columns = aln.label_columns;
                                                 if (!chain_aggregator.add_alignment(std::move(aln)))
                                                     finished_columns.insert(cur_columns.begin(), cur_columns.end());
                                             });
                            },
                            [&](Label column) { return finished_columns.count(column) != 0; });
                    } catch (const early_term &) {}
                    for (Alignment &alignment : chain_aggregator.get_alignments()) {
                        if (alignment.score < get_min_path_score(alignment)) continue;
                        if (graph_.mode == CANONICAL && alignment.orientation) {
                            Alignment rev(alignment);
                            rev.reverse_complement(gview, this_query);
                            if (rev.size()) std::swap(rev, alignment);
                        }
                        add_alignment(std::move(alignment));
                    }
                }
                res.alignments = aggregator.get_alignments();
                continue;
            }
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

// chain_seeds' sort + DP over caller-filled anchors (the checker of mgx_chain_seeds): anchors = TableElem records of one list
void chain_anchors(const mgx_config &config, uint32_t query_size, mgx_chain_anchor *anchors, size_t n, uint32_t *backtrace) {
    std::vector<TableElem> dp(n);
    std::map<Label, size_t> label_sizes;
    for (size_t i = 0; i < n; ++i) {
        dp[i] = TableElem{ anchors[i].label, anchors[i].coordinate, anchors[i].seed_clipping, anchors[i].seed_end, anchors[i].chain_score,
                           anchors[i].seed_index };
        ++label_sizes[anchors[i].label];
    }
    std::vector<uint32_t> back;
    chain_dp(config, (int64_t)query_size, label_sizes, &dp, &back);
    for (size_t i = 0; i < n; ++i) {
        anchors[i].label = dp[i].label; anchors[i].coordinate = dp[i].coordinate; anchors[i].seed_clipping = dp[i].seed_clipping;
        anchors[i].seed_end = dp[i].seed_end; anchors[i].chain_score = dp[i].chain_score; anchors[i].seed_index = dp[i].current_seed_index;
        backtrace[i] = back[i];
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
