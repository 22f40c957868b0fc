// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_graph.hpp header).
#include "orc_graph.hpp"

#include <algorithm>
#include <cassert>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <thread>

namespace orc {

// ---------------------------------------------------------------------------------------------
// alphabet tables
// ---------------------------------------------------------------------------------------------
uint8_t encode_char(char ch) {
    // kmer/alphabets.hpp:67-76; bytes < 0 -> table[0] (kmer_extractor.cpp:33-36)
    int8_t s = static_cast<int8_t>(ch);
    if (s < 0) return 5;
    switch (s) {
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 3;
        case 'T': case 't': case 'U': case 'u': return 4;
        default: return 5;
    }
}

std::vector<uint8_t> encode_seq(std::string_view s) {
    std::vector<uint8_t> r(s.size());
    for (size_t i = 0; i < s.size(); ++i) r[i] = encode_char(s[i]);
    return r;
}

char complement_char(char c) {
    // COMPL_TAB, common/seq_tools/reverse_complement.hpp:31-48 (seqtk table)
    static const unsigned char up[26] = {
        'T','V','G','H','E','F','C','D','I','J','M','L','K','N','O',
        'P','Q','Y','S','A','A','B','W','X','R','Z' };
    unsigned char u = static_cast<unsigned char>(c);
    if (u >= 'A' && u <= 'Z') return up[u - 'A'];
    if (u >= 'a' && u <= 'z') return up[u - 'a'] + ('a' - 'A');
    if (u == 96) return 64;  // table quirk: entry 96 is 64
    return c;
}

void reverse_complement_inplace(std::string &s) {
    // reverse_complement.hpp:50-63
    std::reverse(s.begin(), s.end());
    for (char &c : s) c = complement_char(c);
}

// ---------------------------------------------------------------------------------------------
// Boss: support structures
// ---------------------------------------------------------------------------------------------
void Boss::finalize() {
    assert(W.size() == n + 1 && last.size() == n + 1);
    size_t nb = (n + 1 + 63) / 64;
    w_cum.assign((nb + 1) * SIGMA, 0);
    last_cum.assign(nb + 1, 0);
    last_hint.clear();
    for (int c = 0; c < SIGMA; ++c) w_hint[c].clear();
    uint64_t cnt[SIGMA] = {0}, lc = 0;
    for (size_t b = 0; b < nb; ++b) {
        for (int c = 0; c < SIGMA; ++c) w_cum[b * SIGMA + c] = cnt[c];
        last_cum[b] = lc;
        size_t e = std::min<size_t>((b + 1) * 64, n + 1);
        for (size_t i = b * 64; i < e; ++i) {
            if (i == 0) continue;                        // slot 0 is not an edge
            if (W[i] < SIGMA) {
                if (cnt[W[i]] % 64 == 0) w_hint[W[i]].push_back(b);
                ++cnt[W[i]];
            }
            if (last[i]) {
                if (lc % 64 == 0) last_hint.push_back(b);
                ++lc;
            }
        }
    }
    for (int c = 0; c < SIGMA; ++c) w_cum[nb * SIGMA + c] = cnt[c];
    last_cum[nb] = lc;
    for (int c = 0; c < SIGMA; ++c) NF[c] = rank_last(F[c]);     // boss.cpp:1095-1101
}

uint64_t Boss::rank_W(edge_t i, uint8_t c) const {
    // boss.cpp:437-441: occurrences of c in W[1..i] (slot 0 excluded)
    assert(c < SIGMA && i <= n);
    size_t b = i / 64;
    uint64_t r = w_cum[b * SIGMA + c];
    for (size_t j = std::max<size_t>(b * 64, 1); j <= i; ++j) r += (W[j] == c);
    return r;
}

uint64_t Boss::rank_last(edge_t i) const {
    assert(i <= n);
    size_t b = i / 64;
    uint64_t r = last_cum[b];
    for (size_t j = std::max<size_t>(b * 64, 1); j <= i; ++j) r += last[j];
    return r;
}

edge_t Boss::select_last(uint64_t r) const {
    if (r == 0) return 0;
    size_t b = last_hint[(r - 1) / 64];
    while (last_cum[b + 1] < r) ++b;
    uint64_t c = last_cum[b];
    for (size_t j = std::max<size_t>(b * 64, 1);; ++j) {
        c += last[j];
        if (c == r) return j;
    }
}

edge_t Boss::select_W(uint8_t ch, uint64_t r) const {
    assert(r >= 1);
    size_t b = w_hint[ch][(r - 1) / 64];
    while (w_cum[(b + 1) * SIGMA + ch] < r) ++b;
    uint64_t c = w_cum[b * SIGMA + ch];
    for (size_t j = std::max<size_t>(b * 64, 1);; ++j) {
        c += (W[j] == ch);
        if (c == r) return j;
    }
}

edge_t Boss::pred_last(edge_t i) const {
    // boss.cpp:598-607: last set bit in last[1..i], 0 if none
    while (i && !last[i]) --i;
    return i;
}

edge_t Boss::succ_last(edge_t i) const {
    // boss.cpp:613-617: first set bit in last[i..]
    while (i <= n && !last[i]) ++i;
    return i;
}

std::pair<edge_t, uint8_t> Boss::succ_W(edge_t i, uint8_t a, uint8_t b) const {
    // boss.cpp:515-570: first occurrence of a or b in W[i..n]; (n+1, 0) if none
    for (; i <= n; ++i) {
        if (W[i] == a) return { i, a };
        if (W[i] == b) return { i, b };
    }
    return { n + 1, 0 };
}

uint8_t Boss::get_node_last_value(edge_t i) const {
    // boss.cpp:679-690
    if (i == 0) return 0;
    for (uint8_t c = 0; c < SIGMA; ++c)
        if (F[c] >= i) return c - 1;
    return SIGMA - 1;
}

edge_t Boss::bwd(edge_t i) const {
    // boss.cpp:623-636
    uint64_t target_node = rank_last(i - 1) + 1;
    if (target_node == 1) return 1;
    uint8_t c = get_node_last_value(i);
    return select_W(c, target_node - NF[c]);
}

edge_t Boss::fwd(edge_t i, uint8_t c) const {
    // boss.cpp:642-652
    return select_last(NF[c] + rank_W(i, c));
}

edge_t Boss::pick_edge(edge_t edge, uint8_t c) const {
    // boss.cpp:710-722
    do {
        uint8_t w = W[edge];
        if (w == c || w == c + SIGMA) return edge;
    } while (--edge && !last[edge]);
    return 0;
}

void Boss::call_incoming_to_target(edge_t edge, uint8_t d,
                                   const std::function<void(edge_t)> &cb) const {
    // boss.cpp:766-786
    cb(edge);
    uint8_t d_next;
    while (++edge <= n) {
        std::tie(edge, d_next) = succ_W(edge, d, d + SIGMA);
        if (d_next != d + SIGMA) break;
        cb(edge);
    }
}

bool Boss::is_single_incoming(edge_t i, uint8_t w) const {
    // boss.cpp:802-815
    if (w > SIGMA) return false;
    ++i;
    return i == n + 1 || succ_W(i, w, w + SIGMA).second != w + SIGMA;
}

size_t Boss::num_incoming_to_target(edge_t x, uint8_t d) const {
    // boss.cpp:821-838
    if (x + 1 == n + 1) return 1;
    size_t indeg = 0;
    call_incoming_to_target(x, d, [&](edge_t) { ++indeg; });
    return indeg;
}

bool Boss::tighten_range(edge_t *rl, edge_t *ru, uint8_t s) const {
    // boss.hpp:682-693
    uint64_t rk_rl = rank_W(*rl - 1, s) + 1;
    uint64_t rk_ru = rank_W(*ru, s);
    if (rk_rl > rk_ru) return false;
    *rl = select_last(NF[s] + rk_rl - 1) + 1;
    *ru = select_last(NF[s] + rk_ru);
    return true;
}

static inline void initial_range(const Boss &b, uint8_t s, edge_t *rl, edge_t *ru) {
    // boss.hpp:665-677 (no suffix-range index: it only changes speed)
    *rl = b.F[s] + 1 < b.n + 1 ? b.F[s] + 1 : b.n + 1;
    *ru = s + 1 < SIGMA ? b.F[s + 1] : b.n;
}

edge_t Boss::index(const uint8_t *begin, const uint8_t *end) const {
    // boss.hpp:696-718
    assert(begin + k_ == end);
    if (std::find(begin, end, (uint8_t)SIGMA) != end) return 0;
    edge_t rl, ru;
    initial_range(*this, *begin, &rl, &ru);
    if (rl > ru) return 0;
    for (auto it = begin + 1; it != end; ++it)
        if (!tighten_range(&rl, &ru, *it)) return 0;
    return ru;
}

std::tuple<edge_t, edge_t, size_t> Boss::index_range(const uint8_t *begin, const uint8_t *end) const {
    // boss.hpp:720-764
    if (begin == end) return { 1, 1, 0 };
    if (std::find(begin, end, (uint8_t)SIGMA) != end) return { 0, 0, 0 };
    edge_t rl, ru;
    initial_range(*this, *begin, &rl, &ru);
    if (rl > ru) return { 0, 0, 0 };
    auto it = begin + 1;
    for (; it != end; ++it)
        if (!tighten_range(&rl, &ru, *it)) return { succ_last(rl), ru, size_t(it - begin) };
    return { succ_last(rl), ru, size_t(it - begin) };
}

edge_t Boss::map_to_edge(const uint8_t *begin, const uint8_t *end) const {
    // boss.hpp:766-777
    edge_t edge = index(begin, end - 1);
    return edge && *(end - 1) < SIGMA ? pick_edge(edge, *(end - 1)) : 0;
}

std::vector<edge_t> Boss::map_to_edges(const std::vector<uint8_t> &seq) const {
    // boss.cpp:1003-1045 (terminate/skip hooks are inert on the align path)
    std::vector<edge_t> out;
    if (seq.size() <= k_) return out;
    // utils::drag_and_mark_segments(seq, alph_size, k_ + 1), common/algorithms.hpp:58-74
    std::vector<bool> invalid(seq.size(), false);
    {
        size_t last_occ = std::find(seq.begin(), seq.end(), (uint8_t)SIGMA) - seq.begin();
        for (size_t i = last_occ; i < seq.size(); ++i) {
            if (seq[i] == SIGMA) last_occ = i;
            if (i - last_occ < k_ + 1) invalid[i] = true;
        }
    }
    for (size_t i = 0; i + k_ + 1 <= seq.size(); ++i) {
        if (invalid[i + k_]) { out.push_back(0); continue; }
        edge_t edge = map_to_edge(seq.data() + i, seq.data() + i + k_ + 1);
        out.push_back(edge);
        while (edge && ++i + k_ < seq.size()) {
            if (invalid[i + k_]) { out.push_back(0); break; }
            edge = fwd(edge, seq[i + k_ - 1]);
            edge = pick_edge(edge, seq[i + k_]);
            out.push_back(edge);
        }
    }
    return out;
}

std::vector<uint8_t> Boss::get_node_seq(edge_t x) const {
    // boss.cpp:940-973 without the suffix-range shortcut
    std::vector<uint8_t> ret(k_);
    size_t i = k_;
    ret[--i] = get_node_last_value(x);
    while (i > 0) {
        x = bwd(x);
        ret[--i] = get_node_last_value(x);
    }
    return ret;
}

// ---------------------------------------------------------------------------------------------
// Test-fixture BOSS builder
// ---------------------------------------------------------------------------------------------
namespace {
// an edge = node (k_ codes) + label; BOSS order = co-lex on node, then label (kmer_boss.hpp:15-26,58-63)
struct EdgeKey {
    std::vector<uint8_t> node;   // codes 0..4
    uint8_t label;
};
struct EdgeLess {
    bool operator()(const EdgeKey &a, const EdgeKey &b) const {
        for (size_t i = a.node.size(); i-- > 0;) {
            if (a.node[i] != b.node[i]) return a.node[i] < b.node[i];
        }
        return a.label < b.label;
    }
};
} // namespace

Boss build_boss(size_t k, const std::vector<std::string> &sequences, Mode mode) {
    if (k < 2) throw std::runtime_error("k must be at least 2");
    const size_t kb = k - 1;  // BOSS node length
    std::set<EdgeKey, EdgeLess> real;
    auto add_kmers = [&](const std::string &s) {
        auto enc = encode_seq(s);
        for (size_t i = 0; i + k <= enc.size(); ++i) {
            bool ok = true;
            for (size_t j = 0; j < k; ++j) if (enc[i + j] >= SIGMA || enc[i + j] == 0) { ok = false; break; }
            if (!ok) continue;
            EdgeKey e{ std::vector<uint8_t>(enc.begin() + i, enc.begin() + i + kb), enc[i + kb] };
            real.insert(std::move(e));
        }
    };
    for (const auto &s : sequences) {
        add_kmers(s);
        if (mode == CANONICAL) {              // boss_chunk_construct.cpp:357-359
            std::string rc(s);
            reverse_complement_inplace(rc);
            add_kmers(rc);
        }
    }

    // index real edges by source node and by target node
    std::set<std::vector<uint8_t>> real_sources;
    std::set<std::pair<std::vector<uint8_t>, uint8_t>> node_tail_label;   // (node[1..], label)
    for (const auto &e : real) {
        real_sources.insert(e.node);
        node_tail_label.insert({ std::vector<uint8_t>(e.node.begin() + 1, e.node.end()), e.label });
    }

    std::set<EdgeKey, EdgeLess> all(real.begin(), real.end());

    // dummy sinks (boss_chunk_construct.cpp:57-101): target node without outgoing real edge
    for (const auto &e : real) {
        std::vector<uint8_t> target(e.node.begin() + 1, e.node.end());
        target.push_back(e.label);
        if (!real_sources.count(target)) all.insert(EdgeKey{ target, 0 });
    }

    // dummy sources with prefix length 1 (boss_chunk_construct.cpp:126-171)
    std::set<EdgeKey, EdgeLess> level;
    for (const auto &node : real_sources) {
        // prev_kmer: node' = $ node[0..kb-2], label = node[kb-1]
        std::vector<uint8_t> tail(node.begin(), node.end() - 1);     // a1..a_{kb-1}
        uint8_t label = node.back();
        // redundant iff some real edge  x a1..a_{kb-1} -> a_kb  exists
        if (node_tail_label.count({ tail, label })) continue;
        EdgeKey d;
        d.node.reserve(kb);
        d.node.push_back(0);
        d.node.insert(d.node.end(), tail.begin(), tail.end());
        d.label = label;
        level.insert(std::move(d));
    }
    // prefix lengths 2..kb (boss_chunk_construct.cpp:380-397)
    for (size_t c = 1; c <= kb; ++c) {
        for (const auto &d : level) all.insert(d);
        if (c == kb) break;
        std::set<EdgeKey, EdgeLess> next;
        for (const auto &d : level) {
            EdgeKey p;
            p.node.reserve(kb);
            p.node.push_back(0);
            p.node.insert(p.node.end(), d.node.begin(), d.node.end() - 1);
            p.label = d.node.back();
            next.insert(std::move(p));
        }
        level.swap(next);
    }

    // lay out (boss_chunk.cpp:32-125); the root edge $...$ -> $ comes first (boss_chunk_construct.cpp:404-409)
    std::vector<EdgeKey> edges;
    edges.push_back(EdgeKey{ std::vector<uint8_t>(kb, 0), 0 });
    for (const auto &e : all) {
        if (e.label == 0 && std::all_of(e.node.begin(), e.node.end(), [](uint8_t x) { return x == 0; }))
            continue;
        edges.push_back(e);
    }
    std::sort(edges.begin(), edges.end(), EdgeLess());

    Boss b;
    b.k_ = kb;
    b.W.push_back(0);
    b.last.push_back(0);
    uint64_t curpos = 1;
    uint8_t lastF = 0;
    std::vector<const EdgeKey*> last_kmer(SIGMA, nullptr);
    for (size_t idx = 0; idx < edges.size(); ++idx) {
        const EdgeKey &e = edges[idx];
        uint8_t curW = e.label;
        uint8_t curF = e.node[kb - 1];
        bool same_node_next = idx + 1 < edges.size() && edges[idx + 1].node == e.node;
        if (same_node_next) {
            if (curW == 0 && curF > 0) continue;         // redundant dummy sink (boss_chunk.cpp:83-85)
            b.last.push_back(0);
        } else {
            b.last.push_back(1);
        }
        if (curW) {
            const EdgeKey *lk = last_kmer[curW];
            // same node except the first char -> not the first incoming edge (boss_chunk.cpp:92-101)
            if (lk && std::equal(lk->node.begin() + 1, lk->node.end(), e.node.begin() + 1)) {
                curW += SIGMA;
            } else {
                last_kmer[e.label] = &e;
            }
        }
        b.W.push_back(curW);
        while (curF > lastF && lastF + 1 < SIGMA) b.F[++lastF] = curpos - 1;
        ++curpos;
    }
    while (++lastF < SIGMA) b.F[lastF] = curpos - 1;
    b.n = curpos - 1;
    b.finalize();
    return b;
}

// ---------------------------------------------------------------------------------------------
// Graph (DBGSuccinct)
// ---------------------------------------------------------------------------------------------
void Graph::mask_dummy_kmers() {
    // dbg_succinct.cpp:924-932 -> BOSS::mark_all_dummy_edges (boss.cpp:1743-1775), flipped
    valid.assign(boss.n + 1, 1);
    valid[0] = 0;
    for (edge_t i = 1; i <= boss.n; ++i) {
        if (i > 1 && boss.W[i] == 0) { valid[i] = 0; continue; }
        auto seq = boss.get_node_seq(i);
        if (std::find(seq.begin(), seq.end(), 0) != seq.end()) valid[i] = 0;
    }
    valid[1] = 0;
}

uint64_t Graph::num_nodes() const {
    if (valid.empty()) return boss.n;
    uint64_t c = 0;
    for (auto v : valid) c += v;
    return c;
}

std::vector<node_t> Graph::map_to_nodes_sequentially(std::string_view seq) const {
    // dbg_succinct.cpp:285-305 (no Bloom filter)
    std::vector<node_t> nodes;
    if (seq.size() < get_k()) return nodes;
    auto edges = boss.map_to_edges(encode_seq(seq));
    nodes.reserve(edges.size());
    for (auto e : edges) nodes.push_back(validate_edge(e));
    return nodes;
}

std::vector<node_t> Graph::map_to_nodes(std::string_view seq) const {
    // dbg_succinct.cpp:428-500 (no Bloom filter).  CANONICAL mode: "the definition of a canonical k-mer is redefined: use k-mer
    // with smaller index in the BOSS table"; a k-mer that is missing skips its reverse complement
    if (mode != CANONICAL) return map_to_nodes_sequentially(seq);
    std::vector<node_t> nodes;
    if (seq.size() < get_k()) return nodes;
    std::string rc(seq);
    reverse_complement_inplace(rc);
    auto fwd_edges = boss.map_to_edges(encode_seq(seq));
    auto rc_edges = boss.map_to_edges(encode_seq(rc));
    const size_t n = fwd_edges.size();
    nodes.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        edge_t e = fwd_edges[i];
        if (e) e = std::min(e, rc_edges[n - 1 - i]);
        nodes.push_back(validate_edge(e));
    }
    return nodes;
}

void Graph::call_outgoing_kmers(node_t v, const std::function<void(node_t, char)> &cb) const {
    // dbg_succinct.cpp:110-139
    uint8_t w = 0;
    if (v > 1 && !(w = boss.get_W(v))) return;
    edge_t lst = boss.fwd(v, w % SIGMA);
    edge_t first = boss.pred_last(lst - 1) + 1;
    for (edge_t i = std::max<edge_t>(2, first); i <= lst; ++i) {
        if (in_graph(i)) cb(i, decode_code(boss.get_W(i) % SIGMA));
    }
}

void Graph::build_first_chars(unsigned threads) {
    // D_0[e] = last character of e's source node; D_{r+1}[e] = D_r[bwd(e)]; after k - 2 rounds D[e] is the
    // first character (get_minus_k_value(e, k_ - 1), boss.cpp:696-704)
    const uint64_t n = boss.n;
    threads = std::max(1u, threads);
    std::vector<uint32_t> P(n + 1, 0);
    std::vector<uint8_t> D0(n + 1, 0), D1(n + 1, 0);
    auto par = [&](const std::function<void(uint64_t, uint64_t)> &f) {
        std::vector<std::thread> pool;
        uint64_t chunk = (n + 1 + threads - 1) / threads;
        for (unsigned t = 0; t < threads; ++t) {
            uint64_t b = t * chunk, e = std::min<uint64_t>(n + 1, b + chunk);
            if (b < e) pool.emplace_back(f, b, e);
        }
        for (auto &th : pool) th.join();
    };
    par([&](uint64_t b, uint64_t e) {
        for (uint64_t i = std::max<uint64_t>(b, 1); i < e; ++i) { P[i] = (uint32_t)boss.bwd(i); D0[i] = boss.get_node_last_value(i); }
    });
    for (size_t r = 0; r + 1 < boss.k_; ++r) {
        par([&](uint64_t b, uint64_t e) { for (uint64_t i = b; i < e; ++i) D1[i] = D0[P[i]]; });
        D0.swap(D1);
    }
    first_char = std::move(D0);
}

void Graph::call_incoming_kmers(node_t v, const std::function<void(node_t, char)> &cb) const {
    if (!first_char.empty()) {
        boss.call_incoming_to_target(boss.bwd(v), boss.get_node_last_value(v), [&](edge_t prev) {
            if (in_graph(prev)) cb(prev, decode_code(first_char[prev]));
        });
        return;
    }
    // NodeFirstCache::call_incoming_kmers (graph_extensions/node_first_cache.cpp:27-52):
    // first char = get_minus_k_value(edge, k_-1).first (boss.cpp:696-704)
    boss.call_incoming_to_target(boss.bwd(v), boss.get_node_last_value(v), [&](edge_t prev) {
        if (in_graph(prev)) {
            auto seq = boss.get_node_seq(prev);
            cb(prev, decode_code(seq[0]));
        }
    });
}

bool Graph::has_multiple_outgoing(node_t v) const {
    // dbg_succinct.cpp:609-624
    if (v == 1) return boss.succ_last(1) > 2;
    uint8_t d = boss.get_W(v) % SIGMA;
    if (!d) return false;
    return !boss.get_last(boss.fwd(v, d) - 1);
}

bool Graph::has_single_incoming(node_t v) const {
    // dbg_succinct.cpp:658-678
    if (v == 1) return false;
    edge_t x = boss.bwd(v);
    uint8_t w = boss.get_node_last_value(v);
    size_t first_valid = valid.empty() || valid[x];
    if (x + 1 == boss.n + 1) return first_valid;
    if (first_valid) return boss.is_single_incoming(x, w);
    return boss.num_incoming_to_target(x, w) == 2;
}

std::string Graph::get_node_sequence(node_t v) const {
    // dbg_succinct.cpp:272-279
    auto seq = boss.get_node_seq(v);
    std::string s;
    for (auto c : seq) s += decode_code(c);
    s += decode_code(boss.get_W(v) % SIGMA);
    return s;
}

void Graph::call_nodes_with_suffix_matching_longest_prefix(
        std::string_view str, const std::function<void(node_t, uint64_t)> &cb,
        size_t min_match_length, size_t max_num_allowed_matches) const {
    // dbg_succinct.cpp:307-393
    if (!max_num_allowed_matches || str.size() < min_match_length) return;
    auto encoded = encode_seq(str);
    if (std::find(encoded.begin(), encoded.end(), (uint8_t)SIGMA) != encoded.end()) return;
    const uint8_t *b = encoded.data();
    const uint8_t *e = b + std::min(get_k() - 1, encoded.size());
    auto [first, lst, match_size] = boss.index_range(b, e);

    if (str.size() == get_k() && match_size + 1 == get_k()) {
        edge_t edge = boss.pick_edge(lst, encoded.back());
        if (edge && in_graph(edge)) { cb(edge, get_k()); return; }
    }
    if (match_size < min_match_length) return;

    uint64_t rank_first = boss.rank_last(first);
    uint64_t rank_lst = boss.rank_last(lst);
    if (max_num_allowed_matches < SIZE_MAX) {
        std::vector<node_t> nodes;
        for (uint64_t i = rank_first; i <= rank_lst && nodes.size() <= max_num_allowed_matches; ++i) {
            edge_t ed = boss.select_last(i);
            boss.call_incoming_to_target(boss.bwd(ed), boss.get_node_last_value(ed), [&](edge_t inc) {
                if (in_graph(inc)) nodes.push_back(inc);
            });
        }
        if (nodes.size() > max_num_allowed_matches) return;
        for (auto nd : nodes) cb(nd, match_size);
    } else {
        for (uint64_t i = rank_first; i <= rank_lst; ++i) {
            edge_t ed = boss.select_last(i);
            boss.call_incoming_to_target(boss.bwd(ed), boss.get_node_last_value(ed), [&](edge_t inc) {
                if (in_graph(inc)) cb(inc, match_size);
            });
        }
    }
}

// ---------------------------------------------------------------------------------------------
// GraphView (RCDBG)
// ---------------------------------------------------------------------------------------------
uint64_t GraphView::max_index() const { return canon ? canon->max_index() : g->max_index(); }

void GraphView::call_outgoing_kmers(node_t v, const std::function<void(node_t, char)> &cb) const {
    if (canon) { canon->call_outgoing_kmers(v, cb); return; }
    if (!rc) { g->call_outgoing_kmers(v, cb); return; }
    // rc_dbg.hpp:88-99
    g->call_incoming_kmers(v, [&](node_t prev, char c) { cb(prev, complement_char(c)); });
}

void GraphView::adjacent_incoming_nodes(node_t v, const std::function<void(node_t)> &cb) const {
    if (canon) { canon->adjacent_incoming_nodes(v, cb); return; }
    g->call_incoming_kmers(v, [&](node_t prev, char) { cb(prev); });      // dbg_succinct.cpp:161-176
}

std::string GraphView::get_node_sequence(node_t v) const {
    if (canon) return canon->get_node_sequence(v);
    std::string s = g->get_node_sequence(v);
    if (rc) reverse_complement_inplace(s);
    return s;
}

std::vector<node_t> GraphView::map_to_nodes_sequentially(std::string_view seq) const {
    return canon ? canon->map_to_nodes_sequentially(seq) : g->map_to_nodes_sequentially(seq);
}
bool GraphView::has_multiple_outgoing(node_t v) const { return canon ? canon->has_multiple_outgoing(v) : g->has_multiple_outgoing(v); }
bool GraphView::has_single_incoming(node_t v) const { return canon ? canon->has_single_incoming(v) : g->has_single_incoming(v); }

void GraphView::reverse_complement_seq_path(std::string &seq, std::vector<node_t> &path) const {
    if (canon) { canon->reverse_complement(seq, path); return; }          // sequence_graph.cpp:566-569
    reverse_complement_inplace(seq);
    path = g->map_to_nodes_sequentially(seq);
}

// ---------------------------------------------------------------------------------------------
// CanonicalView (CanonicalDBG over a PRIMARY DBGSuccinct)
// ---------------------------------------------------------------------------------------------
CanonicalView::CanonicalView(const Graph &base)
      : g(&base), offset(base.max_index()), k_odd(base.get_k() % 2), has_sentinel(base.valid.empty()) {}

node_t CanonicalView::reverse_complement(node_t v) const {
    // canonical_dbg.cpp:515-549 (the palindrome cache only memoises the string comparison below)
    if (v > offset) return v - offset;
    if (k_odd) return v + offset;
    std::string seq = g->get_node_sequence(v), rev = seq;
    reverse_complement_inplace(rev);
    return rev == seq ? v : v + offset;
}

void CanonicalView::reverse_complement(std::string &seq, std::vector<node_t> &path) const {
    // :551-560
    reverse_complement_inplace(seq);
    std::vector<node_t> rev(path.size());
    for (size_t i = 0; i < path.size(); ++i) rev[path.size() - 1 - i] = path[i] ? reverse_complement(path[i]) : path[i];
    path.swap(rev);
}

std::vector<node_t> CanonicalView::map_to_nodes_sequentially(std::string_view sequence) const {
    // :55-146 without terminate(); the k-odd BOSS shortcut (:96-126: skip the forward look-up of k-mers found on the
    // other strand) yields what the plain branch does: a node found forward wins, else the one found in the reverse
    // complement (+offset), else npos — for odd k no k-mer is in the base graph on both strands
    std::vector<node_t> out;
    const size_t k = get_k();
    if (sequence.size() < k) return out;
    const size_t total = sequence.size() - k + 1;
    std::vector<node_t> fwd = g->map_to_nodes_sequentially(sequence);
    size_t matched = 0;
    while (matched < fwd.size() && fwd[matched]) ++matched;           // "map until the first mismatch"
    for (size_t i = 0; i < matched; ++i) out.push_back(fwd[i]);
    if (matched == total) return out;
    std::string_view rest = sequence.substr(matched);
    std::string rev_seq(rest);
    reverse_complement_inplace(rev_seq);
    std::vector<node_t> rev_path = g->map_to_nodes_sequentially(rev_seq);
    std::vector<node_t> path = g->map_to_nodes_sequentially(rest);
    for (size_t j = 0; j < path.size(); ++j) {
        node_t r = rev_path[rev_path.size() - 1 - j];
        if (path[j]) out.push_back(path[j]);
        else if (r) out.push_back(r + offset);
        else out.push_back(NPOS);
    }
    return out;
}

std::string CanonicalView::get_node_sequence(node_t v) const {
    node_t base = get_base_node(v);
    std::string seq = g->get_node_sequence(base);
    if (base != v) reverse_complement_inplace(seq);
    return seq;
}

edge_t CanonicalView::prefix_rc(node_t, const std::string &spelling) const {
    // node_first_cache.cpp:120-146
    std::string rev = spelling;
    rev.pop_back();
    if (rev[0] == '$') return 0;
    reverse_complement_inplace(rev);
    auto enc = encode_seq(rev);
    auto [e1, e2, len] = g->boss.index_range(enc.data(), enc.data() + enc.size());
    (void)e1;
    return len == enc.size() ? e2 : 0;
}

edge_t CanonicalView::suffix_rc(node_t, const std::string &spelling) const {
    // node_first_cache.cpp:148-174
    std::string rev = spelling.substr(1);
    if (rev[0] == '$') return 0;
    reverse_complement_inplace(rev);
    auto enc = encode_seq(rev);
    auto [e1, e2, len] = g->boss.index_range(enc.data(), enc.data() + enc.size());
    (void)e2;
    return len == enc.size() ? e1 : 0;
}

void CanonicalView::adjacent_incoming_rc_strand(node_t v, const std::string &hint,
                                                const std::function<void(node_t, char)> &cb) const {
    // :574-632 (DBGSuccinct branch): AGCCAT -> *AGCCA -> TGGCT*; children of the node TGGCT in the base graph
    const Boss &boss = g->boss;
    edge_t rc_edge = prefix_rc(v, hint);
    if (!rc_edge) return;
    edge_t e = rc_edge;                                  // BOSS::call_outgoing (boss.hpp:779-784)
    do {
        node_t prev = e;
        if (g->in_graph(prev)) {
            char c = decode_code(boss.get_W(e) % SIGMA);
            if (!(hint.back() == '$' && c == '$')) cb(prev, c);
        }
    } while (--e && !boss.get_last(e));
}

void CanonicalView::adjacent_outgoing_rc_strand(node_t v, const std::string &hint,
                                                const std::function<void(node_t, char)> &cb) const {
    // :634-684 (DBGSuccinct branch): ATGGCT -> TGGCT* -> *AGCCA; parents of the node AGCCA in the base graph
    const Boss &boss = g->boss;
    edge_t rc_edge = suffix_rc(v, hint);
    if (!rc_edge) return;
    boss.call_incoming_to_target(boss.bwd(rc_edge), boss.get_node_last_value(rc_edge), [&](edge_t prev_edge) {
        node_t prev = prev_edge;
        if (!g->in_graph(prev)) return;
        char c = decode_code(boss.get_node_seq(prev_edge)[0]);         // get_first_char (node_first_cache.cpp:9-23)
        if (hint[0] == '$' && c == '$') return;
        cb(prev, c);
    });
}

void CanonicalView::call_outgoing_kmers(node_t v, const std::string &hint,
                                        const std::function<void(node_t, char)> &cb) const {
    // :156-240
    if (v > offset) {
        std::string rc_hint = hint;
        reverse_complement_inplace(rc_hint);
        call_incoming_kmers(v - offset, rc_hint, [&](node_t next, char c) { cb(reverse_complement(next), complement_char(c)); });
        return;
    }
    node_t children[SIGMA] = { 0, 0, 0, 0, 0 };
    size_t left = SIGMA - (has_sentinel ? 1 : 0);
    g->call_outgoing_kmers(v, [&](node_t next, char c) {
        if (c != '$') { cb(next, c); --left; }
        children[encode_char(c) % SIGMA] = next;            // '$' -> 0
    });
    if (!left) return;
    adjacent_outgoing_rc_strand(v, hint, [&](node_t next, char c) {
        c = complement_char(c);
        uint8_t s = c == '$' ? 0 : encode_char(c);
        if (children[s] != NPOS && c != '$') {
            if (k_odd) throw std::runtime_error("primary graph contains both forward and reverse complement");
            return;                                          // `next` is a palindrome
        }
        next = reverse_complement(next);
        if (c != '$') { cb(next, c); children[s] = next; --left; }
    });
    if (has_sentinel && children[0] && left + 1 == (size_t)SIGMA) cb(children[0], '$');
}

void CanonicalView::call_incoming_kmers(node_t v, const std::string &hint,
                                        const std::function<void(node_t, char)> &cb) const {
    // :242-330
    if (v > offset) {
        std::string rc_hint = hint;
        reverse_complement_inplace(rc_hint);
        call_outgoing_kmers(v - offset, rc_hint, [&](node_t prev, char c) { cb(reverse_complement(prev), complement_char(c)); });
        return;
    }
    node_t parents[SIGMA] = { 0, 0, 0, 0, 0 };
    size_t left = SIGMA - (has_sentinel ? 1 : 0);
    g->call_incoming_kmers(v, [&](node_t prev, char c) {       // NodeFirstCache::call_incoming_kmers
        if (c != '$') { cb(prev, c); --left; }
        parents[c == '$' ? 0 : encode_char(c)] = prev;
    });
    if (!left) return;
    adjacent_incoming_rc_strand(v, hint, [&](node_t prev, char c) {
        c = complement_char(c);
        uint8_t s = c == '$' ? 0 : encode_char(c);
        if (parents[s] != NPOS && c != '$') {
            if (k_odd) throw std::runtime_error("primary graph contains both forward and reverse complement");
            return;
        }
        prev = reverse_complement(prev);
        if (c != '$') { cb(prev, c); parents[s] = prev; --left; }
    });
    if (has_sentinel && parents[0] && left + 1 == (size_t)SIGMA) cb(parents[0], '$');
}

void CanonicalView::adjacent_incoming_nodes(node_t v, const std::function<void(node_t)> &cb) const {
    if (v > offset) adjacent_outgoing_nodes(v - offset, [&](node_t prev) { cb(reverse_complement(prev)); });
    else call_incoming_kmers(v, [&](node_t prev, char) { cb(prev); });
}

void CanonicalView::adjacent_outgoing_nodes(node_t v, const std::function<void(node_t)> &cb) const {
    if (v > offset) adjacent_incoming_nodes(v - offset, [&](node_t next) { cb(reverse_complement(next)); });
    else call_outgoing_kmers(v, [&](node_t next, char) { cb(next); });
}

bool CanonicalView::has_multiple_outgoing(node_t v) const {
    size_t n = 0;
    adjacent_outgoing_nodes(v, [&](node_t) { ++n; });
    return n > 1;
}

bool CanonicalView::has_single_incoming(node_t v) const {
    size_t n = 0;
    adjacent_incoming_nodes(v, [&](node_t) { ++n; });
    return n == 1;
}

} // namespace orc
