// Readers of the reference's own files: `.dbg` (DBGSuccinct::load_without_mask, dbg_succinct.cpp:690-711
// over BOSS::load, boss.cpp:338-394) and `.column.annodbg` (ColumnCompressed::load, annotate_column_compressed.cpp:436-481).
// Host code only; the decoded W / last / F and the columns' set rows go to mgx_graph_create / mgx_annotation_create_sparse.
//
// The reference serialises its vectors with sdsl-lite, an empty submodule under /root/reference: the container layouts below
// are restated from sdsl-lite's published serialisation (`serialize()` of int_vector, rrr_vector, sd_vector, wt_pc + byte_tree,
// rank_support_v / v5, select_support_mcl) and PINNED on the two files the reference wrote itself,
// examples/data/graphs/test_DNA_graph.dbg (SMALL state: wt_huff<rrr_vector<63>> + an rrr `last`; it decodes bit for bit to the
// W / last of our builder on test_DNA_sequences.fa and ends on the byte the parser expects) and test_DNA_graph.column.annodbg
// (legacy label encoder, one inverted sd_vector column; parsed to the last byte) — tests/test_boss_files.py.  Not pinned by a
// reference-written file, only by our own writer of the same layout (tests/sdsl_writer.py) plus the size checks below that make
// a wrong guess an error instead of a wrong graph: the STAT and FAST states (plain bit_vector + rank_support_v / v5 +
// select_support_mcl), sd_vector for `last`, STAT_VECTOR columns and the "LE-v2.0" label encoder.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace mgx { namespace files {

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

inline std::vector<uint8_t> read_whole_file(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw ParseError("cannot open " + path);
    std::vector<uint8_t> buf;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long n = ftell(f);
        if (n > 0) buf.resize((size_t)n);
        rewind(f);
    }
    size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    if (got != buf.size()) throw ParseError("short read of " + path);
    return buf;
}

class Cursor {
  public:
    Cursor(const uint8_t *p, size_t n) : p_(p), n_(n) {}
    size_t at() const { return o_; }
    size_t left() const { return n_ - o_; }
    const uint8_t *take(size_t n, const char *what) {
        if (n > n_ - o_) throw ParseError(std::string("file ends inside ") + what);
        const uint8_t *r = p_ + o_;
        o_ += n;
        return r;
    }
    uint8_t u8(const char *what) { return *take(1, what); }
    uint16_t u16(const char *what) { uint16_t v; memcpy(&v, take(2, what), 2); return v; }
    uint64_t le(const char *what) { uint64_t v; memcpy(&v, take(8, what), 8); return v; }          // sdsl write_member: raw
    uint64_t be(const char *what) {                                                                 // serialize_number, serialization.cpp:30-41
        const uint8_t *b = take(8, what);
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v = (v << 8) | b[i];
        return v;
    }
  private:
    const uint8_t *p_;
    size_t n_, o_ = 0;
};

// sdsl::int_vector<0>::serialize: size in BITS (u64), width (u8), ceil(bits / 64) words; int_vector<w != 0> (bit_vector = 1,
// int_vector<64>): no width byte.
struct Bits {
    uint64_t bits = 0;
    uint8_t width = 1;
    std::vector<uint64_t> w;
    uint64_t size() const { return width ? bits / width : 0; }
    bool bit(uint64_t i) const { return (w[i >> 6] >> (i & 63)) & 1; }
    uint64_t get(uint64_t pos, unsigned len) const {                        // int_vector::get_int(pos, len), len <= 64
        if (!len) return 0;
        const uint64_t lo = w[pos >> 6] >> (pos & 63);
        const unsigned got = 64 - (unsigned)(pos & 63);
        uint64_t v = got < len ? lo | (w[(pos >> 6) + 1] << got) : lo;
        return len == 64 ? v : v & ((1ull << len) - 1);
    }
    uint64_t at(uint64_t i) const { return get(i * width, width); }
};
inline void read_words(Cursor &c, Bits *b, bool keep, const char *what) {
    if (b->bits > (uint64_t)c.left() * 8) throw ParseError(std::string("size field of ") + what + " exceeds the file");
    const uint64_t nw = (b->bits + 63) / 64;
    const uint8_t *src = c.take(nw * 8, what);
    if (keep) { b->w.resize(nw + 1); memcpy(b->w.data(), src, nw * 8); b->w[nw] = 0; }
}
inline Bits read_int_vector(Cursor &c, const char *what, bool keep = true) {
    Bits b;
    b.bits = c.le(what);
    b.width = c.u8(what);
    if (b.width > 64) throw ParseError(std::string("width of ") + what + " above 64");
    read_words(c, &b, keep, what);
    return b;
}
inline Bits read_fixed_vector(Cursor &c, uint8_t width, const char *what, bool keep = true) {      // bit_vector, int_vector<64>
    Bits b;
    b.bits = c.le(what);
    b.width = width;
    read_words(c, &b, keep, what);
    return b;
}
inline uint64_t popcount_bits(const Bits &b) {
    uint64_t n = 0;
    for (uint64_t i = 0; i + 1 < b.w.size(); ++i) {
        uint64_t x = b.w[i];
        if ((i + 1) * 64 > b.bits) x &= (b.bits & 63) ? (1ull << (b.bits & 63)) - 1 : 0;
        n += (uint64_t)__builtin_popcountll(x);
    }
    return n;
}

// rank_support_v<1>: int_vector<64> of 2 words per 512 bits (+ 1 block); rank_support_v5<1>: 2 words per 2048 bits (+ 1).
// Skipped; the expected size is checked so that a layout other than the one assumed is an error.
inline void skip_rank_support(Cursor &c, uint64_t bv_bits, unsigned block_shift, const char *what) {
    Bits b = read_fixed_vector(c, 64, what, false);
    const uint64_t want = ((bv_bits >> block_shift) + 1) * 2 * 64;
    if (b.bits != want && !(bv_bits == 0 && b.bits == 0))
        throw ParseError(std::string(what) + ": " + std::to_string(b.bits) + " bits, expected " + std::to_string(want));
}
// select_support_mcl: arg_cnt; if arg_cnt: superblock (int_vector<>), mini_or_long (bit_vector), then per 4096 arguments one
// int_vector<> (the long superblock's positions or the mini blocks).  Skipped; arg_cnt is checked by the caller.
inline uint64_t skip_select_mcl(Cursor &c, const char *what) {
    const uint64_t cnt = c.le(what);
    if (cnt) {
        read_int_vector(c, what, false);
        read_fixed_vector(c, 1, what, false);
        for (uint64_t sb = (cnt + 4095) >> 12, i = 0; i < sb; ++i) read_int_vector(c, what, false);
    }
    return cnt;
}

struct Binomial63 {                                        // C(n, k) for n <= 63: every entry below 2^63
    uint64_t c[64][64];
    Binomial63() {
        memset(c, 0, sizeof(c));
        for (int n = 0; n < 64; ++n) {
            c[n][0] = 1;
            for (int k = 1; k <= n; ++k) c[n][k] = c[n - 1][k - 1] + (k <= n - 1 ? c[n - 1][k] : 0);
        }
    }
};
inline const Binomial63 &binomial63() { static const Binomial63 b; return b; }

// sdsl::rrr_vector<63>::serialize as the reference's sdsl fork writes it: size (u64), bt (int_vector<>, 6 bits per block: the
// block's popcount), btnr (bit_vector: per block hi(C(63, bt)) + 1 bits, the block's rank in the enumeration of its class),
// btnrp and rank (int_vector<>: samples every 32 blocks, not needed for a sequential decode).  Enumeration
// (rrr_helper::bin_to_nr): scanning the block from bit 0, a set bit at position p with j ones left (it included) adds
// C(62 - p, j).  A block with more ones than zeros is stored as the number of its COMPLEMENT (class 63 - bt); bt itself stays
// the popcount.  Both facts are read off the reference-written test_DNA_graph.dbg (W's tree: bt 34, number of the 29-bit
// complement; last: bt 25, its own number).
inline Bits read_rrr63(Cursor &c, const char *what) {
    Bits out;
    out.bits = c.le(what);
    out.width = 1;
    const Bits bt = read_int_vector(c, what), btnr = read_fixed_vector(c, 1, what);
    read_int_vector(c, what, false);
    read_int_vector(c, what, false);
    if (bt.width != 6 && bt.size()) throw ParseError(std::string(what) + ": block types are not 6 bits wide (not an rrr_vector<63>)");
    if (out.bits > bt.size() * 63) throw ParseError(std::string(what) + ": fewer block types than blocks");
    out.w.assign((out.bits + 63) / 64 + 2, 0);
    const auto &C = binomial63().c;
    uint64_t pos = 0;
    for (uint64_t b = 0; b * 63 < out.bits; ++b) {
        const unsigned k = (unsigned)bt.at(b);
        if (k > 63) throw ParseError(std::string(what) + ": block type above 63");
        uint64_t block = 0;
        if (k == 63) block = ~0ull >> 1;
        else if (k) {
            const unsigned len = 64 - (unsigned)__builtin_clzll(C[63][k]);       // hi(C) + 1
            if (pos + len > btnr.bits) throw ParseError(std::string(what) + ": block numbers end early");
            uint64_t nr = btnr.get(pos, len);
            pos += len;
            const bool inv = 2 * k > 63;
            unsigned j = inv ? 63 - k : k;
            for (unsigned p = 0; p < 63 && j; ++p)
                if (nr >= C[62 - p][j]) { nr -= C[62 - p][j]; --j; block |= 1ull << p; }
            if (j || nr) throw ParseError(std::string(what) + ": block number outside its class");
            if (inv) block = ~block & (~0ull >> 1);
        }
        const uint64_t at = b * 63;
        out.w[at >> 6] |= block << (at & 63);
        if (at & 63) out.w[(at >> 6) + 1] |= block >> (64 - (at & 63));
    }
    const uint64_t nw = (out.bits + 63) / 64;
    if (out.bits & 63) out.w[nw - 1] &= (1ull << (out.bits & 63)) - 1;
    out.w.resize(nw + 1);
    out.w[nw] = 0;
    return out;
}

// sdsl::sd_vector<>::serialize: size (u64), wl (u8), low (int_vector<>, wl bits per one), high (bit_vector: the i-th one's
// upper part h is the 1 at position h + i), select_support_mcl<1> and <0> over high.  Calls f(position) per set bit, ascending.
// `expect`: the length the vector must have (a length field is the one number of these files that nothing else in the file
// bounds: checked before anything of that size is built).
template <class F>
inline uint64_t read_sd_vector(Cursor &c, const char *what, uint64_t expect, F &&f) {
    const uint64_t size = c.le(what);
    if (size != expect) throw ParseError(std::string(what) + ": " + std::to_string(size) + " bits, expected " + std::to_string(expect));
    const uint8_t wl = c.u8(what);
    const Bits low = read_int_vector(c, what), high = read_fixed_vector(c, 1, what);
    const uint64_t ones = skip_select_mcl(c, what), zeros = skip_select_mcl(c, what);
    if (wl > 63 || (low.size() && low.width != wl)) throw ParseError(std::string(what) + ": low part width differs from wl");
    const uint64_t m = popcount_bits(high);
    if (m != ones || zeros != high.bits - m || low.size() < m)
        throw ParseError(std::string(what) + ": select supports disagree with the high part");
    uint64_t i = 0, prev = 0;
    for (uint64_t wi = 0; wi + 1 < high.w.size(); ++wi)
        for (uint64_t x = high.w[wi]; x; x &= x - 1) {
            const uint64_t p = wi * 64 + (uint64_t)__builtin_ctzll(x);
            if (p >= high.bits) break;
            const uint64_t v = ((p - i) << wl) | (wl ? low.at(i) : 0);
            if (v >= size || (i && v <= prev)) throw ParseError(std::string(what) + ": positions not ascending below size");
            f(v);
            prev = v;
            ++i;
        }
    return size;
}

// bit_vector_stat (bit_vector_sdsl<sdsl::bit_vector, rank_support_v5<1>, select_support_mcl<1>, select_support_scan<0>>,
// bit_vector_sdsl.hpp:243-279): the vector, num_set_bits (serialize_number), rank support, select support, nothing for scan.
inline Bits read_bit_vector_stat(Cursor &c, const char *what) {
    Bits v = read_fixed_vector(c, 1, what);
    const uint64_t ones = c.be(what);
    skip_rank_support(c, v.bits, 11, what);
    const uint64_t cnt = skip_select_mcl(c, what);
    if (popcount_bits(v) != ones || cnt != ones) throw ParseError(std::string(what) + ": stored number of set bits differs from the vector's");
    return v;
}

// bit_vector_adaptive::load (bit_vector_adaptive.hpp:105-123): representation code, then that vector.
enum { CODE_RRR = 0, CODE_SD = 1, CODE_STAT = 2, CODE_IL4096 = 3 };
inline Bits read_adaptive(Cursor &c, const char *what, uint64_t expect) {
    const uint64_t code = c.be(what);
    if (code == CODE_RRR) return read_rrr63(c, what);
    if (code == CODE_STAT) return read_bit_vector_stat(c, what);
    if (code == CODE_SD) {
        std::vector<uint64_t> ones;
        const uint64_t size = read_sd_vector(c, what, expect, [&](uint64_t p) { ones.push_back(p); });
        const bool inverted = c.u8(what) != 0;                                      // bit_vector_sd::load, bit_vector_sd.hpp:252-271
        Bits v;
        v.bits = size;
        v.width = 1;
        v.w.assign((size + 63) / 64 + 1, 0);
        for (uint64_t p : ones) v.w[p >> 6] |= 1ull << (p & 63);
        if (inverted) {
            for (uint64_t i = 0; i + 1 < v.w.size(); ++i) v.w[i] = ~v.w[i];
            if (size & 63) v.w[v.w.size() - 2] &= (1ull << (size & 63)) - 1;
        }
        return v;
    }
    if (code == CODE_IL4096) throw Unsupported(std::string(what) + ": bit_vector_il<4096> is not read");
    throw ParseError(std::string(what) + ": unknown bit vector representation " + std::to_string(code));
}

// sdsl::wt_pc<huff_shape, ...>::serialize: size, sigma, the levels' bit vector, its rank / select1 / select0 supports, the tree
// (byte_tree: node count, per node bv_pos u64, bv_pos_rank u64, parent u16, child[2] u16; then c_to_leaf u16[256] and
// path u64[256]).  A leaf has no children and keeps its symbol in bv_pos_rank.  Returns the sequence.
struct WtNode { uint64_t bv_pos, bv_pos_rank; uint16_t parent, child[2]; };
inline std::vector<uint8_t> decode_wt(uint64_t size, uint64_t sigma, const Bits &bv, Cursor &c, const char *what) {
    const uint64_t n_nodes = c.le(what);
    if (n_nodes > 511 || (sigma && n_nodes != 2 * sigma - 1)) throw ParseError(std::string(what) + ": tree size does not fit sigma");
    std::vector<WtNode> nodes(n_nodes);
    for (auto &v : nodes) { v.bv_pos = c.le(what); v.bv_pos_rank = c.le(what); v.parent = c.u16(what); v.child[0] = c.u16(what); v.child[1] = c.u16(what); }
    c.take(256 * 2 + 256 * 8, what);
    if (size && (n_nodes < 3 || size > bv.bits))     // two symbols or more: every element takes a bit of the root's stretch
        throw ParseError(std::string(what) + ": " + std::to_string(size) + " elements over " + std::to_string(bv.bits) + " level bits");
    std::vector<uint8_t> seq(size);
    if (!size) return seq;
    std::vector<uint64_t> cur(n_nodes, 0);
    for (uint64_t i = 0; i < size; ++i) {
        uint16_t v = 0;
        for (unsigned depth = 0; nodes[v].child[0] != 0xFFFF; ++depth) {
            const uint64_t at = nodes[v].bv_pos + cur[v]++;
            if (at >= bv.bits || depth > 255) throw ParseError(std::string(what) + ": walk leaves the level bit vector");
            v = nodes[v].child[bv.bit(at)];
            if (v >= n_nodes) throw ParseError(std::string(what) + ": child index outside the tree");
        }
        if (nodes[v].bv_pos_rank > 255) throw ParseError(std::string(what) + ": symbol above 255");
        seq[i] = (uint8_t)nodes[v].bv_pos_rank;
    }
    return seq;
}

struct BossFile {
    uint32_t k = 0;               // DBG k = BOSS k + 1
    uint32_t mode = 0;            // DeBruijnGraph::Mode: 0 BASIC, 1 CANONICAL, 2 PRIMARY
    uint32_t state = 0;           // BOSS::State as stored (STATE_* below)
    uint32_t sigma = 0;           // alphabet size incl. '$' = F's length (5 for DNA; the device takes nothing else)
    uint64_t n_edges = 0;
    std::vector<uint64_t> F;
    std::vector<uint8_t> W, last; // n_edges + 1 entries, slot 0 unused
};
// BOSS::State, boss.hpp:325: enum State { SMALL = 1, DYN, STAT, FAST }.
enum { STATE_SMALL = 1, STATE_DYN = 2, STATE_STAT = 3, STATE_FAST = 4 };

inline BossFile parse_dbg(const uint8_t *data, size_t n) {
    Cursor c(data, n);
    BossFile g;
    // BOSS::load, boss.cpp:338-394: F (load_number_vector_raw), k, state
    const uint64_t nf = c.be("F");
    if (nf < 2 || nf > 64) throw ParseError("F has " + std::to_string(nf) + " entries");
    g.sigma = (uint32_t)nf;
    for (uint64_t i = 0; i < nf; ++i) g.F.push_back(c.be("F"));
    unsigned logsigma = 1;                           // bits_per_char_W_ = hi(alph_size - 1) + 2, boss.cpp:62
    while ((1u << (logsigma - 1)) < nf) ++logsigma;
    const uint64_t boss_k = c.be("k");
    if (boss_k < 1 || boss_k > 255) throw ParseError("k out of range");
    g.k = (uint32_t)boss_k + 1;
    g.state = (uint32_t)c.be("state");
    std::vector<uint8_t> W;
    Bits last;
    auto wt = [&](bool rrr) {                       // wavelet_tree_sdsl<...>::load, wavelet_tree.cpp:397-419
        const uint64_t size = c.le("W"), sigma = c.le("W");
        Bits bv;
        if (rrr) bv = read_rrr63(c, "W (rrr levels)");      // rank_support_rrr / select_support_rrr serialise nothing
        else {
            bv = read_fixed_vector(c, 1, "W (levels)");
            skip_rank_support(c, bv.bits, 9, "W (rank_support_v)");
            const uint64_t ones = skip_select_mcl(c, "W (select_1)"), zeros = skip_select_mcl(c, "W (select_0)");
            if (ones != popcount_bits(bv) || zeros != bv.bits - ones) throw ParseError("W: select supports disagree with the level bits");
        }
        W = decode_wt(size, sigma, bv, c, "W (tree)");
        if (c.be("W (logsigma)") != logsigma) throw ParseError("W: logsigma does not fit the alphabet");
    };
    switch (g.state) {
        case STATE_STAT: wt(false); last = read_bit_vector_stat(c, "last"); break;
        case STATE_SMALL: wt(true); last = read_adaptive(c, "last", W.size()); break;
        case STATE_FAST: {                          // partite_vector<>::load, wavelet_tree.cpp:519-545: the sequence itself + one bitmap per code
            const Bits iv = read_int_vector(c, "W (int_vector)");
            if (iv.width != logsigma) throw ParseError("W: int_vector width does not fit the alphabet");
            W.resize(iv.size());
            for (uint64_t i = 0; i < W.size(); ++i) W[i] = (uint8_t)iv.at(i);
            for (unsigned j = 0; j < (1u << logsigma); ++j) read_bit_vector_stat(c, "W (bitmap)");
            last = read_bit_vector_stat(c, "last");
            break;
        }
        case STATE_DYN: throw Unsupported("DYN-state graphs are not read: `metagraph transform --state small|stat|fast` first");
        default: throw ParseError("unknown BOSS state " + std::to_string(g.state));
    }
    if (W.empty() || last.bits != W.size()) throw ParseError("W and last differ in length");
    g.n_edges = W.size() - 1;
    for (uint8_t w : W) if (w >= 2 * nf) throw ParseError("W symbol outside the alphabet");
    for (uint64_t i = 0; i < nf; ++i) if (g.F[i] > g.n_edges || (i && g.F[i] < g.F[i - 1])) throw ParseError("F is not ascending below the number of edges");
    g.last.resize(W.size());
    for (uint64_t i = 0; i < W.size(); ++i) g.last[i] = last.bit(i);
    g.W = std::move(W);
    // DBGSuccinct::load_without_mask, dbg_succinct.cpp:701; the suffix-range index that may follow is not used (the device
    // builds its own table)
    const uint64_t mode = c.be("mode");
    if (mode > 2) throw ParseError("unknown graph mode " + std::to_string(mode));
    g.mode = (uint32_t)mode;
    return g;
}
inline BossFile read_dbg(const std::string &path) {
    const std::vector<uint8_t> buf = read_whole_file(path);
    return parse_dbg(buf.data(), buf.size());
}

// The `.edgemask` next to a `.dbg` (DBGSuccinct::load, dbg_succinct.cpp:719-752): the valid-edge bits, n_edges + 1 of them with
// bit 0 clear; a bit_vector_stat for FAST-state graphs, a bit_vector_small otherwise (:724-741).  `metagraph align` drops the
// mask after loading (cli/align.cpp:337-339); callers that keep it (the unit tests' masked graphs) pass it as mgx_boss_view.valid.
inline std::vector<uint8_t> parse_edgemask(const uint8_t *data, size_t n, uint32_t state, uint64_t n_edges) {
    Cursor c(data, n);
    const Bits v = state == STATE_FAST ? read_bit_vector_stat(c, "edge mask") : read_adaptive(c, "edge mask", n_edges + 1);
    if (v.bits != n_edges + 1 || v.bit(0)) throw ParseError("edge mask is not compatible with the graph");      // :749-752
    std::vector<uint8_t> out(n_edges + 1);
    for (uint64_t i = 0; i <= n_edges; ++i) out[i] = v.bit(i);
    return out;
}

struct ColumnFile {
    uint64_t n_rows = 0;
    std::vector<std::string> labels;
    std::vector<uint64_t> col_begin{ 0 }, rows;     // rows[col_begin[j] .. col_begin[j + 1]): the set rows of column j, ascending
};

inline std::string read_short_string(Cursor &c) {   // load_string, serialization.cpp:235-256: UTF-8 coded length, bytes
    uint64_t n = c.u8("label");
    if (n >= 0x80) {                                 // decode_utf8: multi-byte lengths
        int extra = n >= 0xFC ? 5 : n >= 0xF8 ? 4 : n >= 0xF0 ? 3 : n >= 0xE0 ? 2 : n >= 0xC0 ? 1 : -1;
        if (extra < 0) throw ParseError("label length: bad UTF-8 lead byte");
        n &= (1u << (6 - extra)) - 1;
        for (int i = 0; i < extra; ++i) { const uint8_t b = c.u8("label"); if ((b & 0xC0) != 0x80) throw ParseError("label length: bad UTF-8 continuation"); n = (n << 6) | (b & 0x3F); }
    }
    const uint8_t *s = c.take(n, "label");
    return std::string((const char *)s, n);
}

inline ColumnFile parse_columns(const uint8_t *data, size_t n) {
    Cursor c(data, n);
    ColumnFile f;
    f.n_rows = c.be("num_rows");                     // annotate_column_compressed.cpp:452
    if (f.n_rows >= (1ull << 40)) throw ParseError("more rows than the device matrix addresses (2^40)");
    // LabelEncoder<std::string>::load, annotation.cpp:46-86
    if (c.left() >= 7 && !memcmp(data + c.at(), "LE-v2.0", 7)) {
        c.take(7, "label encoder");
        // VectorSet = tsl::ordered_set with IndexType uint64: protocol version, number of elements, bucket count, max load factor
        // (float), the values in insertion order (Serializer: serialize_number length + bytes), then one (index, hash) pair of
        // u64 per bucket
        if (c.le("label encoder") != 1) throw ParseError("label encoder: unknown ordered_set protocol version");
        const uint64_t ne = c.le("label encoder"), nb = c.le("label encoder");
        c.take(4, "label encoder");
        if (ne > c.left() || nb > c.left()) throw ParseError("label encoder: counts exceed the file");
        for (uint64_t i = 0; i < ne; ++i) { const uint64_t len = c.be("label"); const uint8_t *s = c.take(len, "label"); f.labels.emplace_back((const char *)s, len); }
        c.take(nb * 16, "label encoder buckets");
    } else {
        // before v2.0: load_string_vector (the map's keys), load_number_vector (their codes), load_string_vector (decode order)
        const uint64_t nk = c.be("label encoder");
        if (nk > c.left()) throw ParseError("label encoder: counts exceed the file");
        for (uint64_t i = 0; i < nk; ++i) read_short_string(c);
        read_int_vector(c, "label encoder", false);
        const uint64_t nd = c.be("label encoder");
        if (nd > c.left()) throw ParseError("label encoder: counts exceed the file");
        for (uint64_t i = 0; i < nd; ++i) f.labels.push_back(read_short_string(c));
    }
    for (size_t j = 0; j < f.labels.size(); ++j) {   // bit_vector_smart per label, :460-476
        const uint64_t code = c.be("column");
        uint64_t size = 0;
        if (code == CODE_SD) {
            std::vector<uint64_t> ones;
            size = read_sd_vector(c, "column", f.n_rows, [&](uint64_t p) { ones.push_back(p); });
            if (c.u8("column")) {                    // inverted: the stored positions are the rows WITHOUT the label
                // (the expansion has one entry per row: a crafted header must not make a tiny file allocate terabytes — a real
                // column file of n_rows rows is at least n_rows / 8 bytes of some column only if dense, so bound the TOTAL
                // of expanded rows instead: 2^33 entries = 64 GB of row indices is beyond anything one load call should do)
                if (size > (1ull << 33) || f.rows.size() + size > (1ull << 33)) throw Unsupported("column: inverted column of more than 2^33 rows");
                size_t q = 0;
                for (uint64_t r = 0; r < size; ++r) { if (q < ones.size() && ones[q] == r) ++q; else f.rows.push_back(r); }
            } else f.rows.insert(f.rows.end(), ones.begin(), ones.end());
        } else if (code == CODE_STAT || code == CODE_RRR) {
            const Bits v = code == CODE_STAT ? read_bit_vector_stat(c, "column") : read_rrr63(c, "column");
            size = v.bits;
            for (uint64_t wi = 0; wi + 1 < v.w.size(); ++wi)
                for (uint64_t x = v.w[wi]; x; x &= x - 1) {
                    const uint64_t r = wi * 64 + (uint64_t)__builtin_ctzll(x);
                    if (r >= v.bits) throw ParseError("column: a bit is set behind the vector's last position");      // (tail bits of the last word)
                    f.rows.push_back(r);
                }
        } else if (code == CODE_IL4096) throw Unsupported("column: bit_vector_il<4096> is not read");
        else throw ParseError("column: unknown bit vector representation " + std::to_string(code));
        if (size != f.n_rows) throw ParseError("inconsistent column size");          // :466-467
        f.col_begin.push_back(f.rows.size());
    }
    if (c.left()) throw ParseError("bytes left after the last column");
    return f;
}
inline ColumnFile read_columns(const std::string &path) {
    const std::vector<uint8_t> buf = read_whole_file(path);
    return parse_columns(buf.data(), buf.size());
}

}}  // namespace mgx::files
