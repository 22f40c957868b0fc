// mgx.hip — libmgx.so: gfx950 kernels and the C-ABI of include/mgx.h.
// Host side is plain C++ around the HIP runtime; there is no CPU execution path for the aligner.
#include <hip/hip_runtime.h>
#include "pack_swar.hpp"
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/mgx.h"
#define MGX_NO_EXTEND 1      // this unit holds k_map and the seeding kernel; the extension lives in mgx_grp.hip
#include "graph_build.hpp"
#include "map_pipe.hpp"
#include "host_common.hpp"
#include "chain_host.hpp"
#include "lane_types.hpp"
#include "seed_lane.hpp"

using namespace mgx;

// =================================================================================================
// kernels
// =================================================================================================
__global__ void k_build_pass1(const uint8_t *W, const uint8_t *last, uint64_t n, Block *blocks, uint32_t *counts, uint32_t n_blocks) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_blocks) build_block_pass1(b, W, last, n, blocks, counts);
}

struct HintPtrs { uint32_t *last_hint; uint32_t *w_hint[4]; };
__global__ void k_sel_anchor(DevGraph g, uint32_t shift, uint32_t n_entries, uint32_t total_last, uint32_t *out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_entries) build_sel_anchor(g, j, shift, n_entries, total_last, out);
}

__global__ void k_build_pass2(Block *blocks, const uint32_t *cum, HintPtrs hp, uint32_t n_blocks) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_blocks) build_block_pass2(b, blocks, cum, hp.last_hint, hp.w_hint);
}

__global__ void k_parent(DevGraph g, uint32_t *P, uint8_t *D) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e <= g.n) { P[e] = build_parent(g, e); D[e] = (uint8_t)node_last_value(g, e); }
}

__global__ void k_gather(const uint32_t *P, const uint8_t *Din, uint8_t *Dout, uint64_t n) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e <= n) Dout[e] = Din[P[e]];
}

__global__ void k_key_step(const uint8_t *D, uint32_t *key, uint64_t n, uint32_t m, uint32_t r) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e <= n) key[e] = build_key_step(r ? key[e] : 0u, D[e], m, r);
}

__global__ void k_prefix_init(uint2 *tbl, uint64_t entries) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < entries) tbl[i] = make_uint2(1u, 0u);        // empty range (rl > ru)
}

__global__ void k_prefix_fill(const uint32_t *key, uint64_t n, uint2 *tbl) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 1 && e <= n) build_prefix_entry(key, e, n, tbl);
}

__global__ void k_pack_firstc(const uint8_t *D, uint32_t *firstc, uint64_t n) {
    uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi * 8 <= n) {
        uint32_t v = 0;
        for (int j = 0; j < 8; ++j) {
            uint64_t e = wi * 8 + j;
            if (e <= n) v |= (uint32_t)(D[e] & 0xF) << (4 * j);
        }
        firstc[wi] = v;
    }
}

__global__ void k_pack_valid(const uint8_t *valid, uint64_t *bits, uint64_t n) {
    uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi * 64 <= n) {
        uint64_t v = 0;
        for (int j = 0; j < 64; ++j) {
            uint64_t e = wi * 64 + j;
            if (e <= n && valid[e]) v |= 1ull << j;
        }
        bits[wi] = v;
    }
}

__global__ void k_terminus(DevGraph g, uint64_t *bits) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // blockDim multiple of 64
    bool t = v <= g.n && in_graph(g, v) && build_terminus(g, v);
    uint64_t m = __ballot(t);
    if ((threadIdx.x & 63) == 0 && (v >> 6) < ((g.n + 64) >> 6)) bits[v >> 6] = m;
}

// PRIMARY graphs: the wrapper's terminus bits of both ids of every base node (canon_graph.hpp)
__global__ void k_terminus_primary(DevGraph g, uint64_t *bits, uint64_t *bits_rc) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // blockDim multiple of 64
    const uint32_t t = (v <= g.n && in_graph(g, v)) ? build_terminus_primary(g, v) : 0u;
    const uint64_t m0 = __ballot(t & 1u), m1 = __ballot(t & 2u);
    if ((threadIdx.x & 63) == 0 && (v >> 6) < ((g.n + 64) >> 6)) { bits[v >> 6] = m0; bits_rc[v >> 6] = m1; }
}

// PRIMARY graphs (unless MGX_PRIMARY_TABLES=0): palindrome bit of every edge and, per BOSS node, the last edge of its
// reverse complement's node (canon_graph.hpp, canon_children_tables)
__global__ void k_primary_tables(DevGraph g, uint64_t *pal, uint32_t *rc_node) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // blockDim multiple of 64
    const bool live = e >= 1 && e <= g.n;
    const uint64_t m = __ballot(live && build_pal_bit(g, e));
    if ((threadIdx.x & 63) == 0 && (e >> 6) < ((g.n + 64) >> 6)) pal[e >> 6] = m;
    if (live) {
        const Block b = load_block(g, (uint32_t)(e >> 6));
        if ((b.last_bits >> (e & 63)) & 1)
            rc_node[b.last_cum + (uint32_t)popc64(b.last_bits & mask_upto((int)(e & 63)))] = build_rc_node(g, e);
    }
}

// PRIMARY graphs: base-graph mappings of both strands -> the wrapper's paths (canon_merge_pair), one wavefront per read
__global__ void k_canon_merge(DevGraph g, const char *seqs, const uint64_t *offsets, const uint64_t *node_begin,
                              uint32_t *nodes_fwd, uint32_t *nodes_rc, uint64_t n_reads) {
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const int lane = (int)(threadIdx.x & 63);
    for (uint64_t read = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; read < n_reads; read += n_waves) {
        const uint64_t off = offsets[read], nb = node_begin[read];
        const int32_t L = (int32_t)(offsets[read + 1] - off);
        const int32_t nk = (int32_t)(node_begin[read + 1] - nb);
        for (int32_t i = lane; i < nk; i += 64) canon_merge_pair(g, seqs + off, L, i, nodes_fwd + nb, nodes_rc + nb);
    }
}

__global__ void k_kmer_counts(const uint64_t *offsets, uint64_t n_reads, uint32_t k, uint64_t *counts, unsigned long long *lmax) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long L = 0;
    if (i < n_reads) {
        L = offsets[i + 1] - offsets[i];
        counts[i] = L >= k ? L - k + 1 : 0;
    }
    if (i == n_reads) counts[i] = 0;
    // the batch's longest read: one atomic per wavefront (10 M atomics on one address were most of this kernel's 1.6 ms)
    for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(L, d); L = o > L ? o : L; }
    if ((threadIdx.x & 63) == 0 && L) atomicMax(lmax, L);
}

#ifndef MGX_MAP_BLOCKS_PER_CU
#define MGX_MAP_BLOCKS_PER_CU 8          // 256-thread blocks resident per CU (8 = 32 waves: full occupancy if registers allow)
#endif
// persistent lanes, one (read, strand) chain at a time (map_lane_step); 6 waves/SIMD (80 VGPRs, 1 spill) measured best
// for this latency-bound gather kernel (5: +6 %, 8: +3 %)
__global__ void __launch_bounds__(256, 6) k_map(DevGraph g, const char *seqs, const uint64_t *offsets, const uint64_t *node_begin,
                                             uint32_t *nodes_fwd, uint32_t *nodes_rc, uint8_t *mlen_fwd, uint8_t *mlen_rc,
                                             uint2 *rng_fwd, uint2 *rng_rc, int min_rng_len,
                                             uint64_t n_reads, int do_rc, unsigned long long *cursor, KernelStats *stats) {
    LineCtr ctr = { 0, 0, 0 };
    MapLane m;
    m.state = 0;
    const uint64_t n_chains = do_rc ? 2 * n_reads : n_reads;
    auto fetch = [&](MapLane &ml) -> bool {
        uint64_t c = atomicAdd(cursor, 1ull);
        if (c >= n_chains) return false;
        uint64_t read = do_rc ? (c >> 1) : c;
        ml.strand = do_rc ? (int)(c & 1) : 0;
        uint64_t off = offsets[read];
        ml.L = (int32_t)(offsets[read + 1] - off);
        ml.seq = seqs + off;
        ml.out = (ml.strand ? nodes_rc : nodes_fwd) + node_begin[read];
        ml.out_len = (ml.strand ? mlen_rc : mlen_fwd) + node_begin[read];
        uint2 *rg = ml.strand ? rng_rc : rng_fwd;
        ml.out_rng = rg ? rg + node_begin[read] : nullptr;
        ml.min_rng_len = min_rng_len;
        ml.n_kmers = ml.L - (int32_t)g.k + 1;
        return true;
    };
    while (m.state != 3) map_lane_step(g, m, ctr, fetch);
    // per-wave reduction of the line counters
    uint32_t r = ctr.rank_lines, s = ctr.select_lines, b = ctr.bit_lines;
    for (int d = 32; d >= 1; d >>= 1) { r += __shfl_xor(r, d, 64); s += __shfl_xor(s, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0 && (r | s | b)) {
        atomicAdd(&stats->rank_lines, (unsigned long long)r);
        atomicAdd(&stats->select_lines, (unsigned long long)s);
        atomicAdd(&stats->bit_lines, (unsigned long long)b);
        atomicAdd(&stats->map_lines, (unsigned long long)r + s + b);
    }
}

// 2-bit packing pre-pass of the mapping kernel (graph_build.hpp, pack_read_word): one thread per (read, 32-base word)
__global__ void k_pack_reads(const char *seqs, const uint64_t *offsets, uint64_t n_reads, uint32_t words_per_read, int do_rc,
                             uint64_t *pk_fwd, uint64_t *pk_rc, uint32_t *iv_fwd, uint32_t *iv_rc) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t read = t / words_per_read;
    const int32_t j = (int32_t)(t % words_per_read);
    if (read >= n_reads) return;
    const uint64_t off = offsets[read];
    const int32_t L = (int32_t)(offsets[read + 1] - off);
    if (32 * j >= L) return;
    const uint64_t w = packed_word_begin(off, read) + (uint64_t)j;
    uint64_t c; uint32_t v;
    // 32 characters at a time (pack_swar.hpp) where the word's 32 bytes lie inside the batch; the byte loop at its two ends
    const uint64_t total = offsets[n_reads];
    auto load32 = [&](uint64_t at, uint64_t *b) {
        const char *p = seqs + at;
        for (int x = 0; x < 4; ++x) { uint64_t q; __builtin_memcpy(&q, p + 8 * x, 8); b[x] = q; }
    };
    const int32_t p0 = 32 * j, nf = L - p0 < 32 ? L - p0 : 32;
    if (off + (uint64_t)p0 + 32 <= total) {
        uint64_t b[4];
        load32(off + (uint64_t)p0, b);
        mgx_pack::pack32(b, nf, 0, &c, &v);
    } else pack_read_word(seqs + off, L, 0, j, &c, &v);
    pk_fwd[w] = c; iv_fwd[w] = v;
    if (do_rc) {
        // strand position 32 j + t of the reverse complement is read position L - 1 - 32 j - t: the 32 bytes from lo on, backwards
        const int64_t lo = (int64_t)L - 32 * (int64_t)j - 32;
        if ((int64_t)off + lo >= 0) {
            uint64_t b[4];
            load32((uint64_t)((int64_t)off + lo), b);
            mgx_pack::pack32(b, nf, 1, &c, &v);
        } else pack_read_word(seqs + off, L, 1, j, &c, &v);
        pk_rc[w] = c; iv_rc[w] = v;
    }
}

// k_map over the packed reads (k <= 32): same chains, same primitives, same outputs (map_lane_step_packed)
#ifndef MGX_MAP_PACKED_WAVES
#define MGX_MAP_PACKED_WAVES 6
#endif
__global__ void __launch_bounds__(256, MGX_MAP_PACKED_WAVES) k_map_packed(DevGraph g, const uint64_t *offsets, const uint64_t *node_begin,
                                             const uint64_t *pk_fwd, const uint64_t *pk_rc, const uint32_t *iv_fwd, const uint32_t *iv_rc,
                                             uint32_t *nodes_fwd, uint32_t *nodes_rc, uint8_t *mlen_fwd, uint8_t *mlen_rc,
                                             uint2 *rng_fwd, uint2 *rng_rc, int min_rng_len,
                                             uint64_t n_reads, int do_rc, unsigned long long *cursor, KernelStats *stats) {
    LineCtr ctr = { 0, 0, 0 };
    MapLanePacked m;
    m.state = 0;
    const uint64_t n_chains = do_rc ? 2 * n_reads : n_reads;
    auto fetch = [&](MapLanePacked &ml) -> bool {
        uint64_t c = atomicAdd(cursor, 1ull);
        if (c >= n_chains) return false;
        const uint64_t read = do_rc ? (c >> 1) : c;
        const int strand = do_rc ? (int)(c & 1) : 0;
        const uint64_t off = offsets[read];
        const int32_t L = (int32_t)(offsets[read + 1] - off);
        const uint64_t w = packed_word_begin(off, read);
        ml.pk = (strand ? pk_rc : pk_fwd) + w;
        ml.iv = (strand ? iv_rc : iv_fwd) + w;
        ml.n_words = (L + 31) >> 5;
        ml.out = (strand ? nodes_rc : nodes_fwd) + node_begin[read];
        ml.out_len = (strand ? mlen_rc : mlen_fwd) + node_begin[read];
        uint2 *rg = strand ? rng_rc : rng_fwd;
        ml.out_rng = rg ? rg + node_begin[read] : nullptr;
        ml.min_rng_len = min_rng_len;
        ml.n_kmers = L - (int32_t)g.k + 1;
        return true;
    };
    while (m.state != 3) map_lane_step_packed(g, m, ctr, fetch);
    uint32_t r = ctr.rank_lines, s = ctr.select_lines, b = ctr.bit_lines;
    for (int d = 32; d >= 1; d >>= 1) { r += __shfl_xor(r, d, 64); s += __shfl_xor(s, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0 && (r | s | b)) {
        atomicAdd(&stats->rank_lines, (unsigned long long)r);
        atomicAdd(&stats->select_lines, (unsigned long long)s);
        atomicAdd(&stats->bit_lines, (unsigned long long)b);
        atomicAdd(&stats->map_lines, (unsigned long long)r + s + b);
    }
}

// k_map as a request / response machine (map_pipe.hpp): one memory round trip per iteration for all lanes of a wavefront
#ifndef MGX_MAP_PIPE_WAVES
#define MGX_MAP_PIPE_WAVES 3
#endif
#ifndef MGX_SEL_ANCHOR_MAX
#define MGX_SEL_ANCHOR_MAX 8192      // entries of the select-anchor table (32 KB of LDS per workgroup of k_map_pipe)
#endif
__shared__ uint32_t s_sel_anchor[MGX_SEL_ANCHOR_MAX];       // DevGraph::sel_anchor, one copy per workgroup
struct SelPredictLds {
    uint32_t shift;
    __device__ __forceinline__ uint32_t operator()(uint32_t r) const { return sel_predict(s_sel_anchor, shift, r); }
};
__global__ void __launch_bounds__(256, MGX_MAP_PIPE_WAVES) k_map_pipe(DevGraph g, MapArgs a, KernelStats *stats) {
    for (uint32_t j = threadIdx.x; j < g.sel_n; j += blockDim.x) s_sel_anchor[j] = g.sel_anchor[j];
    __syncthreads();
    LineCtr ctr = { 0, 0, 0 };
    MapPipe m;
    map_pipe_init(m);
    ChainClaim claim(a.cursor);
    const SelPredictLds pred = { g.sel_shift };
    for (;;) {
        const bool live = map_pipe_step(g, a, m, ctr, claim, pred);
        if (!__ballot(live)) break;
    }
    uint32_t r = ctr.rank_lines, s = ctr.select_lines, b = ctr.bit_lines;
    for (int d = 32; d >= 1; d >>= 1) { r += __shfl_xor(r, d, 64); s += __shfl_xor(s, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0 && (r | s | b)) {
        atomicAdd(&stats->rank_lines, (unsigned long long)r);
        atomicAdd(&stats->select_lines, (unsigned long long)s);
        atomicAdd(&stats->bit_lines, (unsigned long long)b);
        atomicAdd(&stats->map_lines, (unsigned long long)r + s + b);
    }
}

#include "seed_kernel.hpp"

__global__ void k_iota(uint32_t *v, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

// label-aware alignment: does any dummy node (W == 0: the reference's AnnotationBuffer gives those no labels whatever their row
// says, annotation_buffer.cpp:64-68) have a label in the matrix?  Checked once per aligner; if none does — annotations are built
// from real k-mers — the kernels skip the W look-up in front of every row access.
// label-aware alignment on a CANONICAL-mode graph: node -> the representative of its k-mer (canon_repr_node), whose row holds
// the node's labels
__global__ void k_canon_repr(DevGraph g, uint32_t *out) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > g.n) return;
    out[v] = v ? canon_repr_node(g, v) : 0u;
}
__global__ void k_anno_dummy_rows(DevGraph g, const uint64_t *head, uint64_t n_rows, uint32_t *flag) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (v > g.n || v - 1 >= n_rows || head[v - 1] == 0) return;
    LineCtr lc = { 0, 0, 0 };
    if (get_W(g, v, lc) == 0) atomicOr(flag, 1u);
}

// =================================================================================================
// host side
// =================================================================================================
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? MGX_ERR_OOM : MGX_ERR_NO_DEVICE, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                \
    } while (0)

// Device blocks of destroyed aligners, kept for the next aligner on the same device.  The reference builds one aligner per
// thread-pool task (cli/align.cpp:440-475); with hipMalloc / hipFree per handle every task paid for fresh multi-GB arenas
// (first-touch of a new 20 - 40 GB block: seconds, `profiles/r06_workers_one_device.txt`) and every hipFree synchronised the
// whole device under the other workers.  A block returns here when its aligner is destroyed (its stream has been
// synchronised by then) and is handed out best-fit; the pool is emptied when an allocation fails and by mgx_device_trim().
struct DevPool {
    enum { MAX_DEV = 16 };
    std::mutex mu;
    std::multimap<size_t, void *> blocks[MAX_DEV];
    size_t held[MAX_DEV] = { 0 };
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d >= 0 && d < MAX_DEV ? d : -1; }
    void *take(size_t want, size_t *got) {
        const int d = dev();
        if (d < 0) return nullptr;
        std::lock_guard<std::mutex> lock(mu);
        auto it = blocks[d].lower_bound(want);
        if (it == blocks[d].end() || it->first > want + want / 2 + (1u << 20)) return nullptr;       // (a much larger block stays for who needs it)
        void *p = it->second;
        *got = it->first;
        held[d] -= it->first;
        blocks[d].erase(it);
        return p;
    }
    void give(void *p, size_t bytes) {
        const int d = dev();
        if (d < 0) { (void)hipFree(p); return; }
        std::lock_guard<std::mutex> lock(mu);
        blocks[d].emplace(bytes, p);
        held[d] += bytes;
    }
    size_t held_bytes() {
        const int d = dev();
        if (d < 0) return 0;
        std::lock_guard<std::mutex> lock(mu);
        return held[d];
    }
    void flush() {
        const int d = dev();
        if (d < 0) return;
        std::lock_guard<std::mutex> lock(mu);
        for (auto &e : blocks[d]) (void)hipFree(e.second);
        blocks[d].clear();
        held[d] = 0;
    }
};
static DevPool g_pool;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool pooled = false;              // an aligner's buffer: comes from / returns to g_pool (the graph's own tables do not)
    ~DevBuf() { release(); }
    void release() {
        if (!p) return;
        if (pooled) g_pool.give(p, bytes); else (void)hipFree(p);
        p = nullptr; bytes = 0;
    }
    // headroom: 1/8 on top for buffers that grow with the batch; none for the arena, which is sized from the free memory.
    // (A block from the pool holds whatever its last owner left: like a fresh hipMalloc block, its contents are undefined —
    // every owner initialises what it reads, keyed on `bytes` having changed.)
    int ensure(size_t n, bool exact = false) {
        if (n <= bytes) return MGX_OK;
        release();
        size_t want = exact ? n : n + n / 8 + 256;
        if (pooled) {
            size_t got = 0;
            if (void *q = g_pool.take(want, &got)) { p = q; bytes = got; return MGX_OK; }
        }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess && g_pool.held_bytes()) {      // what the pool holds is free memory: give it back and try again
            (void)hipGetLastError();
            g_pool.flush();
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) { p = nullptr; return fail(MGX_ERR_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
        bytes = want;
        return MGX_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

struct mgx_graph {
    int device = 0;
    DevGraph g;
    DevBuf blocks, last_hint, w_hint[4], sel_anchor, firstc, terminus, valid, prefix_tbl;
    uint64_t bytes = 0;
    uint32_t mode = MGX_MODE_BASIC;
    bool primary_tables = false;      // PRIMARY: reverse-complement tables built (default; MGX_PRIMARY_TABLES=0 turns them off)
    // What label-aware aligners derive from the whole graph, computed once per graph (and annotation) instead of once per
    // aligner — the host adapters build one aligner per batch and worker thread:
    mutable std::mutex label_mu;
    mutable DevBuf canon_repr;        // CANONICAL-mode graphs: node -> the representative of its k-mer (k_canon_repr, 4 B per node)
    mutable bool canon_repr_ready = false;
    mutable std::map<uint64_t, bool> dummy_clean;   // per annotation, by its process-unique id (mgx_annotation_uid: never reused, unlike the handle's address)
};

extern "C" int mgx_launch_align_grp8(const void *params, uint32_t n_groups, uint32_t lds_bytes, int phase, void *stream);    // mgx_grp.hip, MGX_GROUP=8
extern "C" int mgx_grp_waves_per_simd8(void);
extern "C" unsigned mgx_grp_static_lds8(void);
// the seeding kernel with the CanonicalDBG branches (mgx_primary.hip); wps8 selects the 8-waves-per-SIMD instantiation
extern "C" int mgx_launch_seed_primary(const void *params, uint32_t blocks, uint32_t lds_bytes, int wps8, void *stream);
// the same kernel with room for MGX_MAX_ALTERNATIVE_PATHS alignments per query (num_alternative_paths > 1) and with the
// CanonicalDBG branches (PRIMARY graphs)
extern "C" int mgx_launch_align_grp8_alt(const void *params, uint32_t n_groups, uint32_t lds_bytes, int phase, void *stream);
extern "C" unsigned mgx_grp_static_lds8_alt(void);
extern "C" int mgx_grp_waves_per_simd8_alt(void);
extern "C" int mgx_launch_ext64(const void *params, uint32_t blocks, uint32_t lds_bytes, void *stream);      // mgx_ext64.hip
extern "C" int mgx_launch_lane(const void *d_params, uint32_t blocks, void *stream);                         // mgx_lane.hip
extern "C" int mgx_launch_seed_lane(const void *d_params, uint32_t blocks, int long_reads, void *stream);                    // mgx_seedlane.hip
extern "C" int mgx_seed_lane_waves_per_simd(void);
// the 64-lane extension kernel with the label-aware extender compiled in (mgx_lab64.hip: -DMGX_WITH_LABELS=1)
extern "C" int mgx_launch_align_grp8_lab(const void *params, uint32_t n_groups, uint32_t lds_bytes, int phase, void *stream);   // mgx_grp.hip, the labeled build
extern "C" unsigned mgx_grp_static_lds8_lab(void);
extern "C" int mgx_grp_waves_per_simd8_lab(void);
extern "C" int mgx_grp_max_alt8_lab(void);
extern "C" int mgx_launch_lab64(const void *params, uint32_t blocks, uint32_t lds_bytes, void *stream);
extern "C" unsigned mgx_lab64_static_lds(void);
extern "C" int mgx_lab64_waves_per_simd(void);
extern "C" int mgx_lab64_max_alt(void);
extern "C" void mgx_annotation_device_view(const mgx_annotation *a, int *device, uint64_t *n_rows, const uint64_t **head,
                                           const uint32_t **count, const uint32_t **more);                    // mgx_annot.hip
extern "C" uint64_t mgx_annotation_uid(const mgx_annotation *a);                                            // mgx_annot.hip
extern "C" int mgx_annotation_has_coordinates(const mgx_annotation *a);                                     // mgx_annot.hip
extern "C" int mgx_lane_waves_per_simd(void);
// (measurement builds only, tools/build_lane_short_variant.sh: the kernel's -DMGX_LANE_SHORT build — reads of up to 160 characters,
// three wavefronts per SIMD; profiles/r06_ab10_lane_three_waves.txt is why the product does not carry it)
#ifdef MGX_WITH_LANE_SHORT
extern "C" int mgx_launch_lane_short(const void *d_params, uint32_t blocks, void *stream);                   // mgx_lane.hip, -DMGX_LANE_SHORT
extern "C" int mgx_lane_short_waves_per_simd(void);
#else
static int mgx_launch_lane_short(const void *, uint32_t, void *) { return (int)hipErrorNotSupported; }
static int mgx_lane_short_waves_per_simd(void) { return 0; }
#endif
extern "C" unsigned mgx_ext64_static_lds(void);
extern "C" int mgx_ext64_waves_per_simd(void);
extern "C" int mgx_launch_align_grp8_prim(const void *params, uint32_t n_groups, uint32_t lds_bytes, int phase, void *stream);
extern "C" unsigned mgx_grp_static_lds8_prim(void);
extern "C" int mgx_grp_waves_per_simd8_prim(void);

// The pipeline run_align launches: seeding by one wavefront per read (k_align<PH_SEED>), a radix sort of the reads by
// predicted extension work, extension by 8-lane groups (8 reads per wavefront, mgx_grp.hip).  (Round 1 also carried
// fused and 16- / 64- / 1-lane instantiations for A/B measurements; they are gone.)
enum AlignMode { MODE_SPLIT8 = 4, MODE_BAD = -1 };

// launches of every extension kernel since the library was loaded (mgx_kernel_launch_counts; bit order of MGX_KERNEL_*)
static std::atomic<uint64_t> g_kernel_launches[5];

// Measurement probes (occupancy scans, LDS caps, ablations that give WRONG results) exist only in -DMGX_PROBES builds
// (tools/build_variant.sh); the product library ignores their environment variables.
#ifdef MGX_PROBES
static bool probe_env_set(const char *name) { const char *e = getenv(name); return e && *e && strcmp(e, "0") != 0; }
static uint32_t probe_env_u32(const char *name, uint32_t dflt) { const char *e = getenv(name); return e ? (uint32_t)atoi(e) : dflt; }
static uint32_t probe_env_pct(const char *name) { const char *e = getenv(name); return e ? (uint32_t)std::min(100, std::max(1, atoi(e))) : 100u; }
#else
static inline bool probe_env_set(const char *) { return false; }
static inline uint32_t probe_env_u32(const char *, uint32_t dflt) { return dflt; }
static inline uint32_t probe_env_pct(const char *) { return 100u; }
#endif
static AlignMode parse_mode(const char *e) {
    if (!e) return MODE_BAD;
    if (!strcmp(e, "split8")) return MODE_SPLIT8;
    return MODE_BAD;
}
static AlignMode default_mode() { return MODE_SPLIT8; }

struct mgx_aligner {
    const mgx_graph *graph = nullptr;
    const mgx_annotation *anno = nullptr;      // label-aware alignment (mgx_labeled_aligner_create)
    bool anno_dummy_clean = false;             // no row of a dummy node holds a label (k_anno_dummy_rows)
    mgx_config cfg;
    DevConfig dcfg;
    mgx_limits user_lim;
    bool have_user_lim = false;
    DevBuf mlen_fwd, mlen_rc;     // k_map's index() match lengths, one byte per k-mer position
    DevBuf pk_fwd, pk_rc, iv_fwd, iv_rc;      // 2-bit packed reads + invalid flags of the mapping kernel (k <= 32)
    DevBuf rng_fwd, rng_rc;       // and the (rl, ru) of matches >= min_seed_length (8 B per position; optional)
    bool have_rng = false;
    DevBuf score_matrix, seqs, offsets, counts, node_begin, nodes_fwd, nodes_rc, arena, results, stream, cursors, d_stats, d_stats_map, scan_tmp, dbg_seeds;
    DevBuf seed_hdr, seed_stream, work_key, work_key_sorted, order_in, order, sort_tmp, retry_list;    // split pipeline
    DevBuf resume_pool[2], retry_list2, retry_key[2];      // multi-pass extension: resume records, retry lists and keys (ping-pong)
    DevBuf lane_scratch, lane_params, lane_bail, lane_hist;           // lane-per-read kernel: per-lane scratch, its parameter block, the reads it passes on
    bool packed_valid = false;    // pk_* / iv_* hold this batch's strands (k <= 32 and the batch was mapped)
    uint32_t lane_epoch = 0;
    uint64_t lane_done = 0;
    unsigned long long lane_hist_h[32] = { 0 };
    DevBuf seedlane_scratch, seedlane_params, seedlane_bail, seedlane_hist;   // lane-per-read seeder: seed buffers, parameter block, the reads it leaves, why
    uint64_t seedlane_launched = 0;    // the last batch ran it (its counters are read back by collect_stats)
    uint32_t n_passes = 0;
    DevLimits lim;
    uint64_t n_reads = 0, total_kmers = 0;
    uint32_t n_slots = 0;
    bool keep_seeds = false;
    // host copies
    std::vector<ReadResult> h_results;
    std::vector<uint32_t> h_stream;
    HostResults host;
    HostResults host_chained;          // post_chain_alignments: `host` after chain_host.hpp
    HostResults host_retried;          // mgx_align_batch: the results with the re-aligned capacity queries in place
    bool retry_capacity = true;        // (mgx_aligner_set_pipeline "retry_capacity=0": statuses are handed to the caller)
    uint32_t label_scale = 1;          // the label arenas' multiplier (derive_limits; doubled per attempt of the capacity retry)
    const char *last_d_seqs = nullptr;           // the batch mgx_align_batch_device ran last (device pointers)
    const uint64_t *last_d_offsets = nullptr;
    // Every launch, asynchronous copy and hipcub call of this aligner goes to this stream, and its blocking copies synchronise
    // this stream only (copy_sync).  0 = the legacy default stream (what rounds 1-5 used throughout): ordered with everything
    // else the process does on blocking streams.  mgx_aligner_set_stream() gives a handle its own stream, so that the handles
    // of several worker threads on one device (cli/align.cpp:440-475: one aligner per thread-pool task) overlap instead of
    // serialising on the default stream.
    hipStream_t hstream = nullptr;
    int device = 0;                    // the graph's device
    bool own_stream = false;           // mgx_aligner_create_stream: destroyed with the aligner
    uint64_t stage_generation = 0;               // counts stage_batch calls (mgx_map_batch re-stages the same buffers) ...
    uint64_t aligned_generation = 0;             // ... and which of them mgx_align_batch_device aligned: post-chaining and the
                                                 // capacity retry read the batch back only while it is still the staged one
    std::vector<uint64_t> m_node_begin, m_fwd, m_rc;
    mgx_stats hstats;
    hipEvent_t ev[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };     // ([7]: the lane-per-read seeder is through)
    bool split_ran = false;
    uint64_t kernels_ran = 0;     // MGX_KERNEL_* bits of the extension kernels the last batch launched
    uint64_t seed_scale = 1, seed_cap = 0;
    AlignMode mode = default_mode();
    uint64_t arena_stride = 0;
    uint64_t out_words = 0;       // capacity of `stream` for the current batch shape
    uint64_t out_min_words = 0;   // raised when a batch overflowed the heuristic size
    bool no_fast = false;         // test hook: extension through the general path only
    // Result-preserving kernel-selection switches (mgx_aligner_set_pipeline "key=value"; INTEGRATION.md "run-time switches").
    // Every combination gives the same alignments; the parity suite runs the kernels the automatic choice would not pick for
    // its small batches through them.  -1 = automatic.
    struct Options {
        int ext64 = 1;            // 0: never the 64-lane one-read-per-wavefront kernel (small batches run the 8-lane groups)
        int groups_per_wave = -1; // 0 = all 8 groups of a wavefront take reads, 1 .. 8 = that many
        int multi_pass = -1;      // multi-pass extension on / off
        int two_pass = 0;
        int no_compact = 0, no_alias = 0, no_bt_runs = 0, no_flat = 0;
        int primary_alt_build = 0;
        int lane = -1;            // the lane-per-read kernel in front of the extension kernel: -1 auto, 0 off, 1 forced
        int lane_short = 0;       // (measurement builds with MGX_WITH_LANE_SHORT) 1: batches whose longest read has <= 160 characters run the lane kernel's three-wavefront build
        int seed_lane = -1;       // the lane-per-read seeder in front of the seeding kernel (seed_lane.hpp): -1 auto, 0 off, 1 forced
        int seed_wps = 8;         // wavefronts per SIMD of the short-read seeding kernel: 8 (64 VGPRs, spills) or 4 (102 VGPRs, tables in LDS)
        int device_share = 1;     // handles expected to run on this device at the same time (worker threads, -p N): the per-slot
                                  // arenas of this handle are sized for 1 / device_share of the machine instead of all of it
        int map_pipe = 1;         // k_map as the request / response machine (map_pipe.hpp): 1 = for batches of >= 65536 chains, 2 = always,
                                  // 0 = never (one chain step per lane and iteration: rounds 1-4)
    } opt;
};

extern "C" {

int mgx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
const char *mgx_last_error(void) { return g_err.c_str(); }
// the other translation units of libmgx.so (mgx_annot.hip) report through the same thread-local message
extern "C" void mgx_set_last_error(const char *msg) { g_err = msg ? msg : ""; }
uint32_t mgx_abi_version(void) { return MGX_ABI_VERSION; }

void mgx_config_init_default(mgx_config *c) {
    memset(c, 0, sizeof(*c));
    c->num_alternative_paths = 1;
    c->max_num_seeds_per_locus = UINT64_MAX;
    c->min_cell_score = INT32_MIN + 100;
    c->min_path_score = 0;
    c->xdrop = INT32_MAX;
    c->max_nodes_per_seq_char = 1.7976931348623157e308;
    c->max_ram_per_alignment = 1.7976931348623157e308;
    c->gap_opening_penalty = -5;
    c->gap_extension_penalty = -2;
    c->forward_and_reverse_complement = 1;
    c->global_xdrop = 1;
    c->allow_left_trim = 1;
    c->seed_complexity_filter = 1;
}

void mgx_config_set_dna_matrix(mgx_config *c, int8_t match, int8_t mm_transition, int8_t mm_transversion) {
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) c->score_matrix[i][j] = mm_transversion;
    c->score_matrix['A']['G'] = c->score_matrix['G']['A'] = mm_transition;
    c->score_matrix['C']['T'] = c->score_matrix['T']['C'] = mm_transition;
    c->score_matrix['A']['A'] = c->score_matrix['C']['C'] = c->score_matrix['G']['G'] = c->score_matrix['T']['T'] = match;
}

void mgx_config_set_unit_matrix(mgx_config *c, int8_t match) {
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) c->score_matrix[i][j] = (int8_t)-match;
    c->score_matrix['A']['A'] = c->score_matrix['C']['C'] = c->score_matrix['G']['G'] = c->score_matrix['T']['T'] = match;
}

void mgx_config_set_scoring_matrix(mgx_config *c) {
    if (c->alignment_edit_distance) {
        mgx_config_set_unit_matrix(c, 1);
        c->left_end_bonus = 0;
        c->right_end_bonus = 0;
    } else {
        mgx_config_set_dna_matrix(c, c->alignment_match_score, (int8_t)-c->alignment_mm_transition_score,
                                  (int8_t)-c->alignment_mm_transversion_score);
    }
}

void mgx_config_init_cli(mgx_config *c, uint32_t k) {
    mgx_config_init_default(c);
    c->min_seed_length = std::min<uint64_t>(19, k);
    c->max_seed_length = UINT64_MAX;
    c->max_num_seeds_per_locus = 1000;
    c->xdrop = 27;
    c->min_exact_match = 0.7;
    c->max_nodes_per_seq_char = 5.0;
    c->max_ram_per_alignment = 200.0;
    c->rel_score_cutoff = 0.95;
    c->gap_opening_penalty = -6;
    c->gap_extension_penalty = -2;
    c->left_end_bonus = 5;
    c->right_end_bonus = 5;
    c->alignment_edit_distance = 0;            /* cli/config/config.hpp:117-120 */
    c->alignment_match_score = 2;
    c->alignment_mm_transition_score = 3;
    c->alignment_mm_transversion_score = 3;
    mgx_config_set_scoring_matrix(c);
}

void mgx_limits_init_default(mgx_limits *l, uint32_t max_query_length) {
    memset(l, 0, sizeof(*l));
    l->max_query_length = max_query_length;      // the rest: 0 = derive from the config at batch time
}

// ------------------------------------------------------------------------------------------------
// graph
// ------------------------------------------------------------------------------------------------
int mgx_graph_create(const mgx_boss_view *view, int device, mgx_graph **out) {
    if (!view || !out || !view->W || !view->last || !view->F) return fail(MGX_ERR_INVALID, "null BOSS view");
    if (view->sigma != SIGMA) return fail(MGX_ERR_UNSUPPORTED, "only the DNA alphabet $ACGT (sigma = 5) is implemented");
    // CANONICAL: a DBGSuccinct that stores both strands (same index, different aligner flow).  PRIMARY: one k-mer of every
    // pair is stored and the reference aligns through the CanonicalDBG wrapper (dbg_aligner.cpp:52-53) — canon_graph.hpp.
    if (view->mode != MGX_MODE_BASIC && view->mode != MGX_MODE_CANONICAL && view->mode != MGX_MODE_PRIMARY)
        return fail(MGX_ERR_INVALID, "unknown graph mode %u", view->mode);
    if (view->k < 2 || view->k > 255) return fail(MGX_ERR_INVALID, "k out of range");
    if (view->n_edges == 0 || view->n_edges >= 0xFFFFFFF0ull) return fail(MGX_ERR_UNSUPPORTED, "edge count must fit 32 bits");
    if (view->mode == MGX_MODE_PRIMARY) {
        if (view->k > 64) return fail(MGX_ERR_UNSUPPORTED, "PRIMARY-mode graphs need k <= 64 (the load-time kernels hold node spellings in registers)");
        if (view->n_edges * 2 >= 0xFFFFFFF0ull) return fail(MGX_ERR_UNSUPPORTED, "PRIMARY mode: ids of both strands must fit 32 bits");
    }
    if (mgx_device_count() <= device) return fail(MGX_ERR_NO_DEVICE, "HIP device %d not available", device);
    HIP_TRY(hipSetDevice(device));
    auto *G = new mgx_graph();
    std::unique_ptr<mgx_graph> guard(G);
    G->device = device;
    G->mode = view->mode;
    const uint64_t n = view->n_edges;
    const uint32_t n_blocks = (uint32_t)((n + 1 + 63) / 64);
    DevBuf dW, dLast, dValid, counts, cum;
    const uint8_t *W = view->W, *last = view->last;
    if (!view->on_device) {
        if (int rc = dW.ensure(n + 1)) return rc;
        if (int rc = dLast.ensure(n + 1)) return rc;
        HIP_TRY(hipMemcpy(dW.p, view->W, n + 1, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dLast.p, view->last, n + 1, hipMemcpyHostToDevice));
        W = dW.as<uint8_t>();
        last = dLast.as<uint8_t>();
    }
    if (int rc = G->blocks.ensure((size_t)n_blocks * sizeof(Block))) return rc;
    if (int rc = counts.ensure((size_t)n_blocks * 6 * 4)) return rc;
    k_build_pass1<<<(n_blocks + 255) / 256, 256>>>(W, last, n, G->blocks.as<Block>(), counts.as<uint32_t>(), n_blocks);
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> hc((size_t)n_blocks * 6);
    HIP_TRY(hipMemcpy(hc.data(), counts.p, hc.size() * 4, hipMemcpyDeviceToHost));
    uint64_t tot[6] = { 0, 0, 0, 0, 0, 0 };
    for (uint32_t b = 0; b < n_blocks; ++b)
        for (int c = 0; c < 6; ++c) { uint32_t v = hc[(size_t)b * 6 + c]; hc[(size_t)b * 6 + c] = (uint32_t)tot[c]; tot[c] += v; }
    if (int rc = cum.ensure(hc.size() * 4)) return rc;
    HIP_TRY(hipMemcpy(cum.p, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    HintPtrs hp;
    if (int rc = G->last_hint.ensure((tot[5] / 64 + 2) * 4)) return rc;
    hp.last_hint = G->last_hint.as<uint32_t>();
    for (int c = 0; c < 4; ++c) {
        if (int rc = G->w_hint[c].ensure((tot[c + 1] / 64 + 2) * 4)) return rc;
        hp.w_hint[c] = G->w_hint[c].as<uint32_t>();
    }
    k_build_pass2<<<(n_blocks + 255) / 256, 256>>>(G->blocks.as<Block>(), cum.as<uint32_t>(), hp, n_blocks);
    HIP_TRY(hipGetLastError());

    DevGraph &g = G->g;
    memset(&g, 0, sizeof(g));
    g.blocks = G->blocks.as<Block>();
    g.last_hint = hp.last_hint;
    for (int c = 0; c < 4; ++c) g.w_hint[c] = hp.w_hint[c];
    g.n = n;
    g.n_blocks = n_blocks;
    g.k = view->k;
    for (int c = 0; c < SIGMA; ++c) {
        if (view->F[c] > n) return fail(MGX_ERR_INVALID, "F[%d] out of range", c);
        g.F[c] = (uint32_t)view->F[c];
    }
    // NF[c] = rank_last(F[c]) (boss.cpp:1095-1101)
    {
        std::vector<Block> hb(1);
        for (int c = 0; c < SIGMA; ++c) {
            uint64_t i = g.F[c];
            if (i == 0) { g.NF[c] = 0; continue; }
            HIP_TRY(hipMemcpy(hb.data(), g.blocks + (i >> 6), sizeof(Block), hipMemcpyDeviceToHost));
            uint64_t m = hb[0].last_bits & ((i & 63) == 63 ? ~0ull : ((1ull << ((i & 63) + 1)) - 1));
            g.NF[c] = hb[0].last_cum + (uint32_t)__builtin_popcountll(m);
        }
    }
    {
        g.sel_shift = sel_anchor_shift(tot[5], MGX_SEL_ANCHOR_MAX);
        g.sel_n = (uint32_t)(tot[5] >> g.sel_shift) + 2;
        if (int rc = G->sel_anchor.ensure((size_t)g.sel_n * 4)) return rc;
        k_sel_anchor<<<(g.sel_n + 255) / 256, 256>>>(g, g.sel_shift, g.sel_n, (uint32_t)tot[5], G->sel_anchor.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        g.sel_anchor = G->sel_anchor.as<uint32_t>();
    }
    // node mask
    if (view->valid) {
        const uint8_t *valid = view->valid;
        if (!view->on_device) {
            if (int rc = dValid.ensure(n + 1)) return rc;
            HIP_TRY(hipMemcpy(dValid.p, view->valid, n + 1, hipMemcpyHostToDevice));
            valid = dValid.as<uint8_t>();
        }
        if (int rc = G->valid.ensure((size_t)n_blocks * 8)) return rc;
        k_pack_valid<<<(n_blocks + 255) / 256, 256>>>(valid, G->valid.as<uint64_t>(), n);
        HIP_TRY(hipGetLastError());
        g.valid = G->valid.as<uint64_t>();
    }
    // first characters: k - 2 rounds of D[e] <- D[bwd(e)]
    {
        DevBuf P, D0, D1;
        if (int rc = P.ensure((n + 1) * 4)) return rc;
        if (int rc = D0.ensure(n + 1)) return rc;
        if (int rc = D1.ensure(n + 1)) return rc;
        uint32_t nb = (uint32_t)((n + 1 + 255) / 256);
        k_parent<<<nb, 256>>>(g, P.as<uint32_t>(), D0.as<uint8_t>());
        HIP_TRY(hipGetLastError());
        uint8_t *din = D0.as<uint8_t>(), *dout = D1.as<uint8_t>();
        // suffix-range table over the last m node characters, accumulated during the same rounds
        // the longest table whose 8 B x 4^m fit a sixteenth of the free HBM (m = 15: 8.6 GB, 14: 2.1 GB, 13: 0.5 GB)
        uint32_t cap_m = 12;
        {
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            while (cap_m < 15 && (8ull << (2 * (cap_m + 1))) <= free_b / 16) ++cap_m;
        }
        if (const char *e = getenv("MGX_PREFIX_LEN_MAX")) cap_m = (uint32_t)std::min(15, std::max(2, atoi(e)));
        const uint32_t m = choose_prefix_len(n, g.k, cap_m);
        DevBuf key;
        if (int rc = key.ensure((n + 1) * 4)) return rc;
        const uint64_t entries = 1ull << (2 * m);
        if (int rc = G->prefix_tbl.ensure(entries * sizeof(uint2))) return rc;
        k_prefix_init<<<(uint32_t)((entries + 255) / 256), 256>>>(G->prefix_tbl.as<uint2>(), entries);
        k_key_step<<<nb, 256>>>(din, key.as<uint32_t>(), n, m, 0);
        for (uint32_t r = 0; r + 2 < g.k; ++r) {
            k_gather<<<nb, 256>>>(P.as<uint32_t>(), din, dout, n);
            std::swap(din, dout);
            if (r + 1 < m) k_key_step<<<nb, 256>>>(din, key.as<uint32_t>(), n, m, r + 1);
        }
        HIP_TRY(hipGetLastError());
        k_prefix_fill<<<nb, 256>>>(key.as<uint32_t>(), n, G->prefix_tbl.as<uint2>());
        HIP_TRY(hipGetLastError());
        g.prefix_tbl = G->prefix_tbl.as<uint2>();
        g.prefix_len = m;
        if (int rc = G->firstc.ensure(((n + 1 + 7) / 8) * 4 + 4)) return rc;
        k_pack_firstc<<<(uint32_t)(((n + 8) / 8 + 255) / 256), 256>>>(din, G->firstc.as<uint32_t>(), n);
        HIP_TRY(hipGetLastError());
        g.firstc = G->firstc.as<uint32_t>();
        HIP_TRY(hipDeviceSynchronize());
    }
    // MEM terminus bits
    // (PRIMARY graphs: terminus | terminus of the ids v + n | palindrome bits | rc_node table — canon_graph.hpp primary_tables().
    // MGX_PRIMARY_TABLES=0 leaves the last two out (4 bytes per BOSS node less) and the wrapper re-derives spellings and look-ups
    // per expansion: same results, ~7x the random lines)
    G->primary_tables = G->mode == MGX_MODE_PRIMARY && !(getenv("MGX_PRIMARY_TABLES") && atoi(getenv("MGX_PRIMARY_TABLES")) == 0);
    const size_t terminus_bytes = G->mode != MGX_MODE_PRIMARY ? (size_t)n_blocks * 8
                                  : (size_t)n_blocks * 8 * 3 + (G->primary_tables ? (size_t)(tot[5] + 2) * 4 : 0);
    if (int rc = G->terminus.ensure(terminus_bytes, true)) return rc;
    if (G->mode == MGX_MODE_PRIMARY) HIP_TRY(hipMemset(G->terminus.p, 0, terminus_bytes));
    g.terminus = G->terminus.as<uint64_t>();
    if (G->mode == MGX_MODE_PRIMARY) {
        k_terminus_primary<<<n_blocks, 64>>>(g, G->terminus.as<uint64_t>(), G->terminus.as<uint64_t>() + n_blocks);
        if (G->primary_tables)
            k_primary_tables<<<n_blocks, 64>>>(g, G->terminus.as<uint64_t>() + 2ull * n_blocks,
                                               reinterpret_cast<uint32_t *>(G->terminus.as<uint64_t>() + 3ull * n_blocks));
    } else {
        k_terminus<<<n_blocks, 64>>>(g, G->terminus.as<uint64_t>());
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    G->bytes = G->blocks.bytes + G->last_hint.bytes + G->sel_anchor.bytes + G->firstc.bytes + G->terminus.bytes + G->valid.bytes
               + G->prefix_tbl.bytes;
    for (int c = 0; c < 4; ++c) G->bytes += G->w_hint[c].bytes;
    *out = guard.release();
    return MGX_OK;
}

void mgx_graph_destroy(mgx_graph *g) { delete g; }
uint32_t mgx_graph_k(const mgx_graph *g) { return g->g.k; }
uint64_t mgx_graph_max_index(const mgx_graph *g) {
    return g->mode == MGX_MODE_PRIMARY ? 2 * g->g.n : g->g.n;            // CanonicalDBG::max_index (canonical_dbg.hpp:96)
}
uint64_t mgx_graph_device_bytes(const mgx_graph *g) { return g->bytes; }
uint64_t mgx_graph_num_edges(const mgx_graph *g) { return g->g.n; }
uint32_t mgx_graph_mode(const mgx_graph *g) { return g->mode; }

// ------------------------------------------------------------------------------------------------
// aligner
// ------------------------------------------------------------------------------------------------
static int aligner_create(const mgx_graph *g, const mgx_config *config, const mgx_limits *limits, const mgx_annotation *anno,
                          mgx_aligner **out) {
    if (!g || !config || !out) return fail(MGX_ERR_INVALID, "null argument");
    auto *A = new mgx_aligner();
    std::unique_ptr<mgx_aligner> guard(A);
    A->graph = g;
    A->anno = anno;
    A->device = g->device;
    // every device buffer of an aligner goes through the per-device block pool (DevPool): the next aligner takes them over
    for (DevBuf *b : { &A->mlen_fwd, &A->mlen_rc, &A->pk_fwd, &A->pk_rc, &A->iv_fwd, &A->iv_rc, &A->rng_fwd, &A->rng_rc, &A->score_matrix,
                       &A->seqs, &A->offsets, &A->counts, &A->node_begin, &A->nodes_fwd, &A->nodes_rc, &A->arena, &A->results, &A->stream,
                       &A->cursors, &A->d_stats, &A->d_stats_map, &A->scan_tmp, &A->dbg_seeds, &A->seed_hdr, &A->seed_stream, &A->work_key,
                       &A->work_key_sorted, &A->order_in, &A->order, &A->sort_tmp, &A->retry_list, &A->resume_pool[0], &A->resume_pool[1],
                       &A->retry_list2, &A->retry_key[0], &A->retry_key[1], &A->lane_scratch, &A->lane_params, &A->lane_bail, &A->lane_hist, &A->seedlane_scratch, &A->seedlane_params,
                       &A->seedlane_bail, &A->seedlane_hist })
        b->pooled = true;
    {
        std::string err;
        int rc = prepare_config(*config, g->g.k, &A->cfg, &A->dcfg, &err, anno != nullptr);
        if (rc) return fail(rc, "%s", err.c_str());
    }
    if (anno) {
        // LabeledAligner<>(graph, config, annotator) (aligner_labeled.hpp:125-127).  On the device: BASIC-, PRIMARY- and
        // CANONICAL-mode graphs (labels looked up by base node through the CanonicalDBG wrapper, resp. by the k-mer's
        // representative: include/mgx.h), annotation without coordinates, as many alternative paths per label as the
        // labeled kernel build holds.
        if (A->cfg.num_alternative_paths > (uint64_t)std::min(mgx_lab64_max_alt(), mgx_grp_max_alt8_lab()))
            return fail(MGX_ERR_UNSUPPORTED, "label-aware alignment: num_alternative_paths <= %d on the device", std::min(mgx_lab64_max_alt(), mgx_grp_max_alt8_lab()));
        int adev = 0; uint64_t arows = 0; const uint64_t *h; const uint32_t *c, *m;
        mgx_annotation_device_view(anno, &adev, &arows, &h, &c, &m);
        if (adev != g->device) return fail(MGX_ERR_INVALID, "the annotation lives on device %d, the graph on device %d", adev, g->device);
        // An annotation with coordinates makes LabeledAligner chain seeds (aligner_labeled.cpp:457-462: chain_alignments on, global
        // x-drop off).  Of that mode the device has the annotation (mgx_annotation_get_row_tuples) and the chaining DP
        // (mgx_chain_seeds); the extension between chain seeds (dbg_aligner.cpp:155-250,388-529) is not built: refused, so that
        // the caller keeps the reference's aligner — never answered with the plain label-aware mode
        if (mgx_annotation_has_coordinates(anno) && g->mode == MGX_MODE_BASIC)
            return fail(MGX_ERR_UNSUPPORTED, "label-aware alignment with coordinates (seed chaining) is not on the device");
        if (arows < g->g.n) return fail(MGX_ERR_INVALID, "the annotation has %llu rows, the graph %llu nodes (row = node - 1)",
                                        (unsigned long long)arows, (unsigned long long)g->g.n);
    }
    if (g->mode == MGX_MODE_CANONICAL) { A->dcfg.canonical = 1; A->dcfg.fwd_and_rc = 1; }     // dbg_aligner.cpp:225-226
    if (g->mode == MGX_MODE_PRIMARY) { A->dcfg.canonical = g->primary_tables ? 3 : 2; A->dcfg.fwd_and_rc = 1; }   // through the wrapper
    mgx_config &c = A->cfg;
    if (limits) { A->user_lim = *limits; A->have_user_lim = true; }
    HIP_TRY(hipSetDevice(g->device));
    if (anno) {
        int adev = 0; uint64_t arows = 0; const uint64_t *h; const uint32_t *c, *m;
        mgx_annotation_device_view(anno, &adev, &arows, &h, &c, &m);
        std::lock_guard<std::mutex> lock(g->label_mu);
        const uint64_t key = mgx_annotation_uid(anno);
        auto it = g->dummy_clean.find(key);
        if (it == g->dummy_clean.end()) {
            uint32_t *d_flag = nullptr, flag = 1;
            HIP_TRY(hipMalloc(&d_flag, 4));
            HIP_TRY(hipMemset(d_flag, 0, 4));
            if (g->g.n) k_anno_dummy_rows<<<(uint32_t)((g->g.n + 255) / 256), 256>>>(g->g, h, arows, d_flag);
            HIP_TRY(hipMemcpy(&flag, d_flag, 4, hipMemcpyDeviceToHost));
            (void)hipFree(d_flag);
            it = g->dummy_clean.emplace(key, flag == 0).first;
        }
        A->anno_dummy_clean = it->second;
        if (g->mode == MGX_MODE_CANONICAL && !g->canon_repr_ready) {
            if (int rc = g->canon_repr.ensure((g->g.n + 1) * 4)) return rc;
            k_canon_repr<<<(uint32_t)((g->g.n + 256) / 256), 256>>>(g->g, g->canon_repr.as<uint32_t>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            g->canon_repr_ready = true;
        }
    }
    if (int rc = A->score_matrix.ensure(128 * 128)) return rc;
    HIP_TRY(hipMemcpy(A->score_matrix.p, c.score_matrix, 128 * 128, hipMemcpyHostToDevice));
    if (int rc = A->cursors.ensure(128)) return rc;
    if (int rc = A->d_stats.ensure(sizeof(KernelStats))) return rc;
    if (int rc = A->d_stats_map.ensure(sizeof(KernelStats))) return rc;
    for (auto &e : A->ev) HIP_TRY(hipEventCreate(&e));
    memset(&A->hstats, 0, sizeof(A->hstats));
    memset(&A->lim, 0, sizeof(A->lim));
    *out = guard.release();
    return MGX_OK;
}
int mgx_aligner_create(const mgx_graph *g, const mgx_config *config, const mgx_limits *limits, mgx_aligner **out) {
    return aligner_create(g, config, limits, nullptr, out);
}
int mgx_labeled_aligner_create(const mgx_graph *g, const mgx_config *config, const mgx_limits *limits, const mgx_annotation *annotation,
                               mgx_aligner **out) {
    if (!annotation) return fail(MGX_ERR_INVALID, "null argument");
    return aligner_create(g, config, limits, annotation, out);
}

void mgx_aligner_destroy(mgx_aligner *a) {
    if (!a) return;
    (void)hipSetDevice(a->device);                          // (its own copy: the graph may be gone by now — a binding's GC order)
    (void)hipStreamSynchronize(a->hstream);
    for (auto &e : a->ev) if (e) (void)hipEventDestroy(e);
    if (a->own_stream) (void)hipStreamDestroy(a->hstream);
    delete a;
}

int mgx_aligner_get_config(const mgx_aligner *a, mgx_config *out) { *out = a->cfg; return MGX_OK; }

int mgx_aligner_get_limits(const mgx_aligner *a, mgx_limits *out) {
    if (!a || !out) return fail(MGX_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    out->max_query_length = a->lim.Lmax;
    out->max_columns = a->lim.max_columns;
    out->max_seeds = a->lim.max_seeds;
    out->cell_arena_bytes = (uint64_t)a->lim.cell_words * 4;
    return MGX_OK;
}

// a blocking copy that synchronises the aligner's stream only (hipMemcpy would synchronise the whole device's default stream)
static hipError_t copy_sync(mgx_aligner *A, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    if (hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, A->hstream)) return e;
    return hipStreamSynchronize(A->hstream);
}

// stage inputs, compute k-mer slot offsets and Lmax on the device
static int stage_batch(mgx_aligner *A, const char *seqs, const uint64_t *offsets, uint64_t n, int on_device,
                       const char **d_seqs, const uint64_t **d_offsets, uint32_t *Lmax_out) {
    const uint32_t k = A->graph->g.k;
    ++A->stage_generation;
    if (on_device) {
        *d_seqs = seqs;
        *d_offsets = offsets;
    } else {
        uint64_t total = offsets[n];
        if (int rc = A->seqs.ensure(total + 16)) return rc;
        if (int rc = A->offsets.ensure((n + 1) * 8)) return rc;
        HIP_TRY(hipMemcpyAsync(A->seqs.p, seqs, total, hipMemcpyHostToDevice, A->hstream));
        HIP_TRY(hipMemcpyAsync(A->offsets.p, offsets, (n + 1) * 8, hipMemcpyHostToDevice, A->hstream));
        *d_seqs = A->seqs.as<char>();
        *d_offsets = A->offsets.as<uint64_t>();
    }
    if (int rc = A->counts.ensure((n + 2) * 8)) return rc;
    if (int rc = A->node_begin.ensure((n + 2) * 8)) return rc;
    unsigned long long *cur = A->cursors.as<unsigned long long>();
    HIP_TRY(hipMemsetAsync(cur, 0, 128, A->hstream));
    k_kmer_counts<<<(uint32_t)((n + 1 + 255) / 256), 256, 0, A->hstream>>>(*d_offsets, n, k, A->counts.as<uint64_t>(), cur + 2);
    HIP_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, A->counts.as<uint64_t>(), A->node_begin.as<uint64_t>(), (int)(n + 1), A->hstream));
    if (int rc = A->scan_tmp.ensure(tmp_bytes + 16)) return rc;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(A->scan_tmp.p, tmp_bytes, A->counts.as<uint64_t>(), A->node_begin.as<uint64_t>(), (int)(n + 1), A->hstream));
    uint64_t total_kmers = 0;
    unsigned long long lmax = 0;
    HIP_TRY(copy_sync(A, &total_kmers, A->node_begin.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost));
    HIP_TRY(copy_sync(A, &lmax, cur + 2, 8, hipMemcpyDeviceToHost));
    A->total_kmers = total_kmers;
    A->n_reads = n;
    *Lmax_out = (uint32_t)lmax;
    if (int rc = A->nodes_fwd.ensure((total_kmers + 1) * 4)) return rc;
    if (int rc = A->nodes_rc.ensure((total_kmers + 1) * 4)) return rc;
    if (int rc = A->mlen_fwd.ensure(total_kmers + 1)) return rc;
    if (int rc = A->mlen_rc.ensure(total_kmers + 1)) return rc;
    {
        // the range arrays are an optimisation: only when they fit comfortably next to everything else
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        free_b += g_pool.held_bytes();
        const size_t need = 2 * (total_kmers + 1) * sizeof(uint2);
        A->have_rng = need <= A->rng_fwd.bytes + A->rng_rc.bytes || need < free_b / 4;
        if (A->have_rng) {
            if (int rc = A->rng_fwd.ensure((total_kmers + 1) * sizeof(uint2))) return rc;
            if (int rc = A->rng_rc.ensure((total_kmers + 1) * sizeof(uint2))) return rc;
        }
    }
    return MGX_OK;
}

static int run_map(mgx_aligner *A, const char *d_seqs, const uint64_t *d_offsets, uint64_t n, bool do_rc, bool mapped, uint32_t Lmax) {
    // k_map's counters live in their own block: a re-run of the alignment stage (stream overflow) resets only its own
    HIP_TRY(hipMemsetAsync(A->d_stats_map.p, 0, sizeof(KernelStats), A->hstream));
    HIP_TRY(hipMemsetAsync(A->d_stats.p, 0, sizeof(KernelStats), A->hstream));
    A->packed_valid = false;
    if (!mapped) {
        // max_seed_length < k: nodes are not mapped (dbg_aligner.cpp:209-213)
        HIP_TRY(hipMemsetAsync(A->nodes_fwd.p, 0, (A->total_kmers + 1) * 4, A->hstream));
        HIP_TRY(hipMemsetAsync(A->nodes_rc.p, 0, (A->total_kmers + 1) * 4, A->hstream));
        HIP_TRY(hipMemsetAsync(A->mlen_fwd.p, 0xFF, A->total_kmers + 1, A->hstream));
        HIP_TRY(hipMemsetAsync(A->mlen_rc.p, 0xFF, A->total_kmers + 1, A->hstream));
        return MGX_OK;
    }
    unsigned long long *map_cursor = A->cursors.as<unsigned long long>() + 4;
    HIP_TRY(hipMemsetAsync(map_cursor, 0, 8, A->hstream));
    HIP_TRY(hipMemsetAsync(A->mlen_fwd.p, 0xFF, A->total_kmers + 1, A->hstream));       // MLEN_UNKNOWN
    if (do_rc) HIP_TRY(hipMemsetAsync(A->mlen_rc.p, 0xFF, A->total_kmers + 1, A->hstream));
    HIP_TRY(hipEventRecord(A->ev[0], A->hstream));
    {
        // persistent lanes: enough wavefronts to fill the device, never more lanes than chains
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, A->graph->device));
        const uint64_t chains = (do_rc ? 2 : 1) * n;
        uint64_t blocks = std::min<uint64_t>((uint64_t)prop.multiProcessorCount * MGX_MAP_BLOCKS_PER_CU, (chains + 255) / 256);
        if (blocks == 0) blocks = 1;
        const bool no_pack = probe_env_set("MGX_MAP_BYTES");          // A/B probe: the byte-per-character path
        if (A->graph->g.k <= 32 && Lmax > 0 && !no_pack) {
            // words: every read starts at (offset >> 5) + read and takes ceil(L / 32) of them
            const uint64_t words = ((A->total_kmers + n * (uint64_t)A->graph->g.k) >> 5) + n + 2;
            if (int rc = A->pk_fwd.ensure(words * 8)) return rc;
            if (int rc = A->iv_fwd.ensure(words * 4)) return rc;
            if (do_rc) { if (int rc = A->pk_rc.ensure(words * 8)) return rc; if (int rc = A->iv_rc.ensure(words * 4)) return rc; }
            const uint32_t wpr = (Lmax + 31) / 32;
            const uint64_t threads = n * wpr;
            k_pack_reads<<<(uint32_t)((threads + 255) / 256), 256, 0, A->hstream>>>(d_seqs, d_offsets, n, wpr, do_rc ? 1 : 0, A->pk_fwd.as<uint64_t>(),
                                                                    A->pk_rc.as<uint64_t>(), A->iv_fwd.as<uint32_t>(), A->iv_rc.as<uint32_t>());
            HIP_TRY(hipGetLastError());
            A->packed_valid = true;
            // (a batch of few chains — config 0: 1000 long queries — leaves most lanes of either kernel idle; there the one-step-
            // per-lane machine's single iteration per k-mer beats the pipe's 2.4: 30 vs 46 ms, profiles/r05_config0_*.json)
            if (A->opt.map_pipe == 1 ? chains >= 65536 : A->opt.map_pipe > 1) {
                MapArgs ma;
                ma.offsets = d_offsets; ma.node_begin = A->node_begin.as<uint64_t>();
                ma.pk_fwd = A->pk_fwd.as<uint64_t>(); ma.pk_rc = A->pk_rc.as<uint64_t>();
                ma.iv_fwd = A->iv_fwd.as<uint32_t>(); ma.iv_rc = A->iv_rc.as<uint32_t>();
                ma.nodes_fwd = A->nodes_fwd.as<uint32_t>(); ma.nodes_rc = A->nodes_rc.as<uint32_t>();
                ma.mlen_fwd = A->mlen_fwd.as<uint8_t>(); ma.mlen_rc = A->mlen_rc.as<uint8_t>();
                ma.rng_fwd = A->have_rng ? A->rng_fwd.as<uint2>() : nullptr; ma.rng_rc = A->have_rng ? A->rng_rc.as<uint2>() : nullptr;
                ma.min_rng_len = (int32_t)std::min<uint64_t>(A->cfg.min_seed_length, 1u << 20);
                ma.n_reads = n; ma.do_rc = do_rc ? 1 : 0; ma.cursor = map_cursor;
                const uint64_t pblocks = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)prop.multiProcessorCount * MGX_MAP_PIPE_WAVES, (chains + 255) / 256));
                k_map_pipe<<<(uint32_t)pblocks, 256, 0, A->hstream>>>(A->graph->g, ma, A->d_stats_map.as<KernelStats>());
            } else
            k_map_packed<<<(uint32_t)blocks, 256, 0, A->hstream>>>(A->graph->g, d_offsets, A->node_begin.as<uint64_t>(),
                                         A->pk_fwd.as<uint64_t>(), A->pk_rc.as<uint64_t>(), A->iv_fwd.as<uint32_t>(), A->iv_rc.as<uint32_t>(),
                                         A->nodes_fwd.as<uint32_t>(), A->nodes_rc.as<uint32_t>(),
                                         A->mlen_fwd.as<uint8_t>(), A->mlen_rc.as<uint8_t>(),
                                         A->have_rng ? A->rng_fwd.as<uint2>() : nullptr, A->have_rng ? A->rng_rc.as<uint2>() : nullptr,
                                         (int)std::min<uint64_t>(A->cfg.min_seed_length, 1u << 20), n, do_rc ? 1 : 0,
                                         map_cursor, A->d_stats_map.as<KernelStats>());
        } else {
            k_map<<<(uint32_t)blocks, 256, 0, A->hstream>>>(A->graph->g, d_seqs, d_offsets, A->node_begin.as<uint64_t>(),
                                         A->nodes_fwd.as<uint32_t>(), A->nodes_rc.as<uint32_t>(),
                                         A->mlen_fwd.as<uint8_t>(), A->mlen_rc.as<uint8_t>(),
                                         A->have_rng ? A->rng_fwd.as<uint2>() : nullptr, A->have_rng ? A->rng_rc.as<uint2>() : nullptr,
                                         (int)std::min<uint64_t>(A->cfg.min_seed_length, 1u << 20), n, do_rc ? 1 : 0,
                                         map_cursor, A->d_stats_map.as<KernelStats>());
        }
    }
    HIP_TRY(hipGetLastError());
    if (A->graph->mode == MGX_MODE_PRIMARY && do_rc && n) {
        k_canon_merge<<<(uint32_t)std::min<uint64_t>((n + 3) / 4, 65536), 256, 0, A->hstream>>>(A->graph->g, d_seqs, d_offsets, A->node_begin.as<uint64_t>(),
                                                                             A->nodes_fwd.as<uint32_t>(), A->nodes_rc.as<uint32_t>(), n);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(A->ev[1], A->hstream));
    return MGX_OK;
}

// bits of the work-sort key: 12 of predicted work + the segment number (align_core.hpp, WORK_SEGMENT_SHIFT)
static int work_key_bits(uint64_t n) {
    int b = 12;
    for (uint64_t seg = n ? (n - 1) >> WORK_SEGMENT_SHIFT : 0; seg; seg >>= 1) ++b;
    return std::min(b, 32);
}

static int run_align(mgx_aligner *A, const char *d_seqs, const uint64_t *d_offsets, uint64_t n, uint32_t Lmax) {
    {
        std::string err;
        int rc = derive_limits(A->cfg, A->have_user_lim ? &A->user_lim : nullptr, Lmax, &A->lim, &err, A->anno != nullptr, A->label_scale);
        if (rc) return fail(rc, "%s", err.c_str());
    }
    const bool labeled = A->anno != nullptr;
    const DevLimits &l = A->lim;
    const uint64_t stride = arena_bytes(l);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, A->graph->device));
    // The lane-per-read kernel (mgx_lane.hip) in front of the group kernel: does this batch's configuration qualify?  Its
    // scratch is taken before the arena is sized from what is free.
    LaneParams LP;
    memset(&LP, 0, sizeof(LP));
    std::string lane_why;
    // (two builds of the kernel: reads of up to 160 characters at three wavefronts per SIMD, up to 256 at two)
    const bool lane_short = A->opt.lane_short != 0 && l.Lmax <= 160 && mgx_lane_short_waves_per_simd() > 0;
    const uint32_t lane_blocks = (uint32_t)prop.multiProcessorCount * 4u * (uint32_t)(lane_short ? mgx_lane_short_waves_per_simd() : mgx_lane_waves_per_simd());
    // (label-aware batches, round 6: the lane takes the reads whose seeds and columns all carry one and the same single label —
    // lane_read.hpp; it needs the annotation's "no dummy node's row holds a label" flag, as it reads rows without the W test)
    bool lane_ok = A->opt.lane != 0 && A->packed_valid && A->mode == MODE_SPLIT8 && !A->opt.two_pass && A->opt.multi_pass != 1
                   && lane_enabled(A->cfg, A->dcfg, A->graph->g.k, l.Lmax, A->no_fast, &LP, &lane_why,
                                   labeled ? (1u | (A->anno_dummy_clean ? 2u : 0u)) : 0u)
                   && (A->opt.lane == 1 || n >= (uint64_t)lane_blocks * 16);       // (a small batch: the spread / 64-lane kernels are quicker)
    if (lane_ok) {
        LP.max_cols = lane_max_cols(l.Lmax, A->dcfg.xdrop);
        LP.hash_slots = next_pow2(2ull * LP.max_cols);
        if (const char *hm = getenv("MGX_LANE_HASH_MULT")) LP.hash_slots *= (uint32_t)std::max(1, atoi(hm));      // (measurement only)
        LP.rest_stride = (lane_rest_bytes(LP.max_cols, LP.hash_slots) + 63) & ~63ull;
        LP.wave_stride = lane_wave_scratch_bytes(LP.max_cols, LP.rest_stride);
        const size_t before = A->lane_scratch.bytes;
        if (A->lane_scratch.ensure((size_t)lane_blocks * LP.wave_stride, true) != MGX_OK) lane_ok = false;      // (no room: the group kernel alone)
        else if (A->lane_scratch.bytes != before) HIP_TRY(hipMemsetAsync(A->lane_scratch.p, 0, A->lane_scratch.bytes, A->hstream));    // node tables start empty
    }
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const AlignMode mode = A->mode;
    const bool split = mode == MODE_SPLIT8;      // always
    const uint64_t wave_slots = (uint64_t)prop.multiProcessorCount * 4 * MGX_ALIGN_WAVES_PER_SIMD;   // seeding kernel: one wavefront per read
    // (the extension kernel that will run: the same alt / prim selection as launch_groups below)
    const bool sel_primary = A->dcfg.canonical >= 2;
    // (post_chain_alignments keeps up to MGX_MAX_ALTERNATIVE_PATHS alignments per query: the build with room for them)
    const uint64_t agg_cap = std::max<uint64_t>(1, A->cfg.post_chain_alignments ? (uint64_t)post_chain_capacity(A->cfg.num_alternative_paths) : A->cfg.num_alternative_paths);
    const bool sel_alt = agg_cap > 1 || (sel_primary && A->opt.primary_alt_build == 1);
    const uint64_t ext_wps = sel_alt ? mgx_grp_waves_per_simd8_alt() : sel_primary ? mgx_grp_waves_per_simd8_prim() : mgx_grp_waves_per_simd8();
    // (label-aware batches: seeding as ever, extension on the one-read-per-wavefront labeled kernel)
    const uint64_t want_slots = labeled ? std::max<uint64_t>(wave_slots, (uint64_t)prop.multiProcessorCount * 4 * 8 * mgx_grp_waves_per_simd8_lab())
                                        : std::max<uint64_t>(split ? wave_slots : 0, (uint64_t)prop.multiProcessorCount * 4 * 8 * ext_wps);
    // The arena gets what is free after the buffers this stage allocates AFTER it (result records, output stream, seed
    // stream, sort arrays: estimated generously) and a margin; buffers kept from an earlier batch are already outside
    // `free_b`.  (Half of the free memory, as before, left 15 % of the extension kernel's groups without a slice at
    // 10 M reads next to a host framework's cached allocations.)
    const uint64_t later = n * (sizeof(ReadResult) + sizeof(SeedHdr) + 24 + 16)
                           + (n * (((uint64_t)l.Lmax + l.Lmax / 4 + 40) * agg_cap) + 1024) * 4
                           + (n * 24 + A->total_kmers / 8 + 4096) * A->seed_scale * sizeof(DevSeed) + (1ull << 30);
    const uint64_t held = A->results.bytes + A->stream.bytes + A->seed_stream.bytes + A->seed_hdr.bytes;     // re-used as far as they reach
    const uint64_t need_later = later > held ? later - held : 0;
    const uint64_t avail = (uint64_t)free_b + A->arena.bytes + g_pool.held_bytes();     // a growing arena frees its old block first; what the
                                                                                          // block pool holds is handed out or freed on demand
    uint64_t budget = avail > need_later ? (avail - need_later) / 10 * 9 : avail / 2;
    // (device_share: this handle is one of several at work on the device — each gets its share of the slots the machine can keep
    // resident and of the memory, instead of the first handles taking all of it and the later ones what is left)
    const uint64_t share = (uint64_t)std::max(1, A->opt.device_share);
    budget /= share;
    uint64_t slots = std::min<uint64_t>(std::min<uint64_t>((want_slots + share - 1) / share, std::max<uint64_t>(n, 1)), std::max<uint64_t>(1, budget / stride));
    if (slots == 0) slots = 1;
    {
        // The hash tables of the convergence checker are cleared by generation tags that persist in each slice,
        // so a slice only needs zeroing when its layout (stride) changes or the buffer is new.
        const size_t before = A->arena.bytes;
        if (int rc = A->arena.ensure(slots * stride, true)) return rc;
        if (A->arena.bytes != before || A->arena_stride != stride) {
            HIP_TRY(hipMemsetAsync(A->arena.p, 0, A->arena.bytes, A->hstream));
            A->arena_stride = stride;
        }
    }
    A->n_slots = (uint32_t)slots;
    if (probe_env_set("MGX_DEBUG_SLOTS")) fprintf(stderr, "run_align: n %llu stride %llu want_slots %llu slots %llu free %.1f GB\n", (unsigned long long)n, (unsigned long long)stride, (unsigned long long)want_slots, (unsigned long long)slots, free_b / 1e9);
    if (int rc = A->results.ensure(n * sizeof(ReadResult))) return rc;
    uint64_t words_per_read = ((uint64_t)l.Lmax + l.Lmax / 4 + 40) * agg_cap;
    if (labeled) words_per_read = words_per_read * 2 + 16;      // (an alignment per label group + the label lists; heuristic as below)
    // heuristic size (one alignment per read: nodes + CIGAR runs + path characters); a batch that needs more is re-run
    // with what it asked for (mgx_align_batch_device), so the size is never a correctness limit
    uint64_t out_words = std::max<uint64_t>(n * words_per_read + 1024, A->out_min_words);
    if (int rc = A->stream.ensure(out_words * 4)) return rc;
    A->out_words = out_words;
    if (A->keep_seeds) {
        if (int rc = A->dbg_seeds.ensure(n * 2 * (uint64_t)l.max_seeds * sizeof(DevSeed))) return rc;
        HIP_TRY(hipMemsetAsync(A->dbg_seeds.p, 0, n * 2 * (uint64_t)l.max_seeds * sizeof(DevSeed), A->hstream));
    }
    unsigned long long *cur = A->cursors.as<unsigned long long>();
    HIP_TRY(hipMemsetAsync(cur, 0, 32, A->hstream));
    HIP_TRY(hipMemsetAsync(A->d_stats.p, 0, sizeof(KernelStats), A->hstream));     // counters of this run of the stage only
    AlignParams P;
    memset(&P, 0, sizeof(P));
    P.g = A->graph->g;
    P.cfg = A->dcfg;
    P.lim = l;
    P.score_matrix = A->score_matrix.as<int8_t>();
    P.seqs = d_seqs;
    P.offsets = d_offsets;
    P.node_begin = A->node_begin.as<uint64_t>();
    P.nodes_fwd = A->nodes_fwd.as<uint32_t>();
    P.nodes_rc = A->nodes_rc.as<uint32_t>();
    P.mlen_fwd = A->mlen_fwd.as<uint8_t>();
    P.mlen_rc = A->mlen_rc.as<uint8_t>();
    P.rng_fwd = A->have_rng ? A->rng_fwd.as<uint2>() : nullptr;
    P.rng_rc = A->have_rng ? A->rng_rc.as<uint2>() : nullptr;
    P.n_reads = n;
    P.arena = A->arena.as<uint8_t>();
    P.arena_stride = stride;
    P.results = A->results.as<ReadResult>();
    P.out_stream = A->stream.as<uint32_t>();
    P.out_capacity = out_words;
    P.out_cursor = cur;
    P.read_cursor = cur + 1;
    P.stats = A->d_stats.as<KernelStats>();
    P.dbg_seeds = A->keep_seeds ? A->dbg_seeds.as<DevSeed>() : nullptr;
    P.no_fast = A->no_fast;
    if (A->packed_valid) {
        P.pkw[0] = A->pk_fwd.as<uint64_t>(); P.ivw[0] = A->iv_fwd.as<uint32_t>();
        P.pkw[1] = A->dcfg.fwd_and_rc ? A->pk_rc.as<uint64_t>() : nullptr; P.ivw[1] = A->dcfg.fwd_and_rc ? A->iv_rc.as<uint32_t>() : nullptr;
    }
    P.no_compact = A->opt.no_compact != 0;
    P.no_alias = A->opt.no_alias != 0;
    P.no_bt_runs = A->opt.no_bt_runs != 0;
    P.no_flat = A->opt.no_flat != 0 || A->cfg.post_chain_alignments;       // (the flat group loop is the one-alignment-per-query driver)
    if (labeled) {
        int adev = 0;
        mgx_annotation_device_view(A->anno, &adev, &P.anno_rows, &P.anno_head, &P.anno_count, &P.anno_more);
        P.labeled = 1u | (A->anno_dummy_clean ? 2u : 0u);
        P.anno_base = A->graph->mode == MGX_MODE_CANONICAL ? A->graph->canon_repr.as<uint32_t>() : nullptr;
        P.no_alias = 1;           // (a flush clears columns in place: convergence entries must not alias their S windows)
    }
#ifdef MGX_PROBES
    P.ablate = getenv("MGX_ABLATE") ? (uint32_t)atoi(getenv("MGX_ABLATE")) : 0u;      // timing probes: WRONG results (probe builds only)
#endif
    size_t sort_tmp_bytes = 0;
    if (split) {
        // seeds travel from the seeding kernel to the extension kernel through a compact stream
        // typical reads carry a handful of seeds; long or repetitive ones scale with their k-mer count.  The stream is
        // sized by a heuristic times A->seed_scale; mgx_align_batch_device re-runs the stage with a larger scale if the
        // seeding kernel ran out of room (the cursor keeps counting), so the size is never a correctness limit.
        const uint64_t seed_cap = std::min<uint64_t>(n * 2 * (uint64_t)l.max_seeds,
                                                     (n * 24 + A->total_kmers / 8 + 4096) * A->seed_scale);
        A->seed_cap = seed_cap;
        if (int rc = A->seed_hdr.ensure(n * sizeof(SeedHdr))) return rc;
        if (int rc = A->seed_stream.ensure(seed_cap * sizeof(DevSeed))) return rc;
        if (int rc = A->work_key.ensure(n * 4)) return rc;
        if (int rc = A->work_key_sorted.ensure(n * 4)) return rc;
        if (int rc = A->order_in.ensure(n * 4)) return rc;
        if (int rc = A->order.ensure(n * 4)) return rc;
        if (int rc = A->retry_list.ensure(n * 4 + 4)) return rc;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp_bytes, A->work_key.as<uint32_t>(), A->work_key_sorted.as<uint32_t>(),
                                                   A->order_in.as<uint32_t>(), A->order.as<uint32_t>(), (int)n, 0, work_key_bits(n), A->hstream));
        if (int rc = A->sort_tmp.ensure(sort_tmp_bytes)) return rc;
        P.seed_hdr = A->seed_hdr.as<SeedHdr>();
        P.seed_stream = A->seed_stream.as<DevSeed>();
        P.seed_capacity = seed_cap;
        P.seed_cursor = cur + 2;
        P.work_key = A->work_key.as<uint32_t>();
    }
    HIP_TRY(hipEventRecord(A->ev[2], A->hstream));
    // latency-critical scalar arrays go to LDS when they fit next to the other resident waves of the CU
    // per-wavefront share of the CU's 160 KB minus the kernel's static LDS (control block, sdust scratch, score rows);
    // overshooting by a few bytes costs a whole resident wavefront per CU
    // (the seeding kernels: control block + the interval lists of the sdust scratch; no score rows — seed_kernel.hpp)
    const uint32_t static_lds = (uint32_t)((sizeof(Wave) + SDUST_LDS_BYTES_REGTAB + 16 + 127) & ~127ull);
    uint32_t lds_budget = (160u * 1024u) / (4 * MGX_ALIGN_WAVES_PER_SIMD) - static_lds - 64u;
    uint32_t lds_bytes = std::min<uint32_t>(fast_lds_bytes(l.Lmax), lds_budget) & ~15u;
    const uint32_t w_slots = (uint32_t)std::min<uint64_t>(slots, wave_slots);
    auto launch_groups = [&](int phase) -> int {
        // tuning probe: MGX_EXT_GROUPS_PCT=50 launches half the resident groups (occupancy experiments)
        const uint32_t pct = probe_env_pct("MGX_EXT_GROUPS_PCT");
        const uint32_t groups = 8;
        // three builds of the extension kernel (mgx_grp.hip): the product, the product with the CanonicalDBG branches (PRIMARY
        // graphs), and the one with room for alternative paths (either kind of graph; MGX_PRIMARY_ALT_BUILD=1: A/B switch that
        // sends PRIMARY graphs there as rounds 2-3 did)
        const bool prim_to_alt = A->opt.primary_alt_build == 1;
        const bool primary = A->dcfg.canonical >= 2;
        const bool alt = agg_cap > 1 || (primary && prim_to_alt);
        const bool prim = primary && !alt;
        const uint32_t waves_cu = 4u * (uint32_t)(alt ? mgx_grp_waves_per_simd8_alt() : prim ? mgx_grp_waves_per_simd8_prim() : mgx_grp_waves_per_simd8());
        const uint32_t static_lds = alt ? mgx_grp_static_lds8_alt() : prim ? mgx_grp_static_lds8_prim() : mgx_grp_static_lds8();
        uint32_t per_wave = (160u * 1024u) / waves_cu - static_lds - 64u;
        uint32_t per_group = std::min<uint32_t>(fast_lds_bytes(l.Lmax), per_wave / groups) & ~15u;
        per_group = std::min<uint32_t>(per_group, probe_env_u32("MGX_EXT_LDS_CAP", per_group)) & ~15u;   // tuning probe
        {
            // Fewer reads than resident groups (long-read batches, single queries): spread them over the wavefronts.  The 8
            // groups of a wavefront execute in lock-step, and reads of 1 .. 12 kbp side by side wait for each other's general
            // steps two thirds of the time (config 0: 1.35 s with 8 reads per wavefront on 125 of 3072 wavefronts).  The arena
            // slices stay what they are: slot = wavefront x groups_per_wave + group.
            const uint64_t launch_groups_n = std::min<uint64_t>(slots, std::max<uint64_t>(1, slots * pct / 100));
            const uint64_t resident_waves = (uint64_t)prop.multiProcessorCount * waves_cu;
            const int gpw_env = A->opt.groups_per_wave;      // 0 = all 8
            const uint64_t items = P.n_items ? P.n_items : n;                       // (a later pass of the multi-pass extension: its retry positions)
            const uint64_t busy = std::min<uint64_t>(launch_groups_n, std::max<uint64_t>(1, items));
            const uint64_t want_gpw = std::min<uint64_t>(groups, std::max<uint64_t>(1, (busy + resident_waves - 1) / resident_waves));
            P.groups_per_wave = gpw_env >= 0 ? (uint32_t)std::min(8, gpw_env) : (want_gpw < groups ? (uint32_t)want_gpw : 0u);
        }
        // One read per wavefront: the 64-lane instantiation (mgx_ext64.hip) — the read has the wavefront to itself, so it may as
        // well use all of its lanes.  MGX_EXT64=0: A/B switch.
        if (labeled) {
            // label-aware extension: the labeled builds — 8 reads per wavefront (mgx_grp.hip, per-read program), or one read
            // per wavefront on the 64-lane kernel (mgx_lab64.hip) for batches with fewer reads than resident wavefronts
            const uint32_t wcu64 = 4u * (uint32_t)mgx_lab64_waves_per_simd();
            const uint64_t resident64 = (uint64_t)prop.multiProcessorCount * wcu64;
            const uint64_t items = P.n_items ? P.n_items : n;
            if ((items <= resident64 && A->opt.ext64 != 0) || A->opt.ext64 == 2) {
                const uint32_t lds64 = std::min<uint32_t>(fast_lds_bytes(l.Lmax), (160u * 1024u) / wcu64 - mgx_lab64_static_lds() - 128u) & ~15u;
                P.groups_per_wave = 1;
                A->kernels_ran |= MGX_KERNEL_LAB64;
                return mgx_launch_lab64(&P, (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(slots, resident64)), lds64, A->hstream);
            }
            const uint32_t wcu = 4u * (uint32_t)mgx_grp_waves_per_simd8_lab();
            const uint32_t per_wave_l = (160u * 1024u) / wcu - mgx_grp_static_lds8_lab() - 64u;
            const uint32_t per_group_l = std::min<uint32_t>(fast_lds_bytes(l.Lmax), per_wave_l / 8) & ~15u;
            const int gpw_opt = A->opt.groups_per_wave;
            P.groups_per_wave = gpw_opt > 0 ? (uint32_t)std::min(8, gpw_opt) : 0u;
            A->kernels_ran |= MGX_KERNEL_GRP8_LAB;
            return mgx_launch_align_grp8_lab(&P, (uint32_t)slots, per_group_l, phase, A->hstream);
        }
        const bool ext64 = A->opt.ext64 != 0;
        if (ext64 && phase == PH_EXTEND && P.groups_per_wave == 1) {
            const uint32_t wcu = 4u * (uint32_t)mgx_ext64_waves_per_simd();
            const uint32_t lds64 = std::min<uint32_t>(fast_lds_bytes(l.Lmax), (160u * 1024u) / wcu - mgx_ext64_static_lds() - 128u) & ~15u;
            A->kernels_ran |= MGX_KERNEL_EXT64;
            ++g_kernel_launches[3];
            return mgx_launch_ext64(&P, (uint32_t)std::min<uint64_t>(slots, std::max<uint64_t>(1, slots * pct / 100)), lds64, A->hstream);
        }
        A->kernels_ran |= alt ? MGX_KERNEL_GRP8_ALT : prim ? MGX_KERNEL_GRP8_PRIM : MGX_KERNEL_GRP8;
        ++g_kernel_launches[alt ? 2 : prim ? 1 : 0];
        return (alt ? mgx_launch_align_grp8_alt : prim ? mgx_launch_align_grp8_prim : mgx_launch_align_grp8)(&P, (uint32_t)std::min<uint64_t>(slots, std::max<uint64_t>(1, slots * pct / 100)), per_group, phase, A->hstream);      // never more groups than arena slices (a partial wavefront is fine: the kernel returns for slot >= n_groups)
    };
    A->split_ran = split;
    A->kernels_ran = 0;
    A->seedlane_launched = 0;
    if (split) {
        // The lane-per-read seeder (seed_lane.hpp, mgx_seedlane.hip) first: every lane seeds its own read and either publishes
        // header, seeds and work key as the seeding kernel would, or lists the read for that kernel, which then seeds the
        // list from scratch (no host round trip in between: the list's length stays on the device).
        const bool seed_lane = A->opt.seed_lane != 0 && !probe_env_set("MGX_SEED_GROUPS") && A->packed_valid && P.pkw[0] && P.ivw[0]
                               && (!A->dcfg.fwd_and_rc || (P.pkw[1] && P.ivw[1])) && P.mlen_fwd && P.rng_fwd
                               && (!A->dcfg.fwd_and_rc || (P.mlen_rc && P.rng_rc))
                               && seed_lane_enabled(A->dcfg, (uint32_t)A->graph->g.k, l.Lmax, true, true)
                               && (A->opt.seed_lane == 1 || n >= 4096);
        if (seed_lane && n) {
            const bool long_reads = l.Lmax > (uint32_t)SL_SHORT_L;            // (the kernel's build with nine packed words per strand)
            const uint32_t resident = (uint32_t)prop.multiProcessorCount * 4 * (uint32_t)mgx_seed_lane_waves_per_simd();
            const uint32_t blocks1 = (uint32_t)std::min<uint64_t>(resident, (n + 63) / 64);
            // (the second pass: what the first left — a sixth of a typical batch — with the big buffers)
            const uint32_t blocks2 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(resident / (long_reads ? 2 : 1), (n / 4 + 63) / 64));
            const bool many = (uint64_t)A->graph->g.k >= A->dcfg.max_seed_length;
            const uint32_t me1 = many ? (long_reads ? SL_SEEDS_1_MANY_LONG : SL_SEEDS_1_MANY) : SL_SEEDS_1, mp1 = many ? SL_PENDING_1_MANY : SL_PENDING_1;
            const uint32_t me2 = long_reads ? SL_SEEDS_2_LONG : SL_SEEDS_2, mp2 = long_reads ? SL_PENDING_2_LONG : SL_PENDING_2;
            const uint64_t words1 = seed_lane_wave_scratch_words(me1, mp1), words2 = seed_lane_wave_scratch_words(me2, mp2);
            if (int rc = A->seedlane_scratch.ensure((size_t)std::max<uint64_t>(blocks1 * words1, blocks2 * words2) * 4)) return rc;
            if (int rc = A->seedlane_params.ensure(2 * sizeof(SeedLaneParams))) return rc;
            if (int rc = A->seedlane_bail.ensure(2 * (n * 4 + 4))) return rc;
            if (int rc = A->seedlane_hist.ensure(32 * 8)) return rc;          // ([16 .. 24): section timers of -DMGX_SL_TIMERS builds)
            HIP_TRY(hipMemsetAsync(A->seedlane_hist.p, 0, 32 * 8, A->hstream));
            HIP_TRY(hipMemsetAsync(cur + 8, 0, 32, A->hstream));
            const bool two = A->opt.seed_lane != 2;                          // (seed_lane=2: A/B switch, the first pass only)
            SeedLaneParams SP[2];
            memset(SP, 0, sizeof(SP));
            uint32_t *list1 = A->seedlane_bail.as<uint32_t>(), *list2 = list1 + n + 1;
            for (int ps = 0; ps < 2; ++ps) {
                SP[ps].P = P;
                SP[ps].P.n_items = n;
                SP[ps].scratch = A->seedlane_scratch.as<uint32_t>();
                SP[ps].max_entries = ps ? me2 : me1; SP[ps].max_pending = ps ? mp2 : mp1;
                SP[ps].long_reads = long_reads ? 1u : 0u;
                SP[ps].second_pass = (uint32_t)ps;
                SP[ps].in_list = ps ? list1 : nullptr; SP[ps].in_count = ps ? cur + 8 : nullptr; SP[ps].in_count_back = ps ? cur + 11 : nullptr;
                SP[ps].list_len = n;
                SP[ps].bail_list = ps ? list2 : list1;
                SP[ps].bail_count = ps ? cur + 10 : cur + 8;
                SP[ps].bail_count_back = (!ps && two) ? cur + 11 : nullptr;
                SP[ps].done_count = cur + 9;
                // (why a read left: counted where it leaves for the wave program)
                SP[ps].bail_hist = (ps || !two) ? A->seedlane_hist.as<unsigned long long>() : nullptr;
            }
            HIP_TRY(copy_sync(A, A->seedlane_params.p, SP, sizeof(SP), hipMemcpyHostToDevice));
            if (int rc = mgx_launch_seed_lane(A->seedlane_params.p, blocks1, long_reads, A->hstream)) return fail(MGX_ERR_NO_DEVICE, "lane-per-read seeder: %d", rc);
            HIP_TRY(hipMemsetAsync(cur + 1, 0, 8, A->hstream));                      // rewind the read cursor
            if (two) {
                if (int rc = mgx_launch_seed_lane(A->seedlane_params.as<SeedLaneParams>() + 1, blocks2, long_reads, A->hstream)) return fail(MGX_ERR_NO_DEVICE, "lane-per-read seeder, second pass: %d", rc);
                HIP_TRY(hipMemsetAsync(cur + 1, 0, 8, A->hstream));
            }
            P.seed_list = two ? list2 : list1;
            P.n_items_ptr = two ? cur + 10 : cur + 8;
            A->seedlane_launched = 1;
        }
        HIP_TRY(hipEventRecord(A->ev[7], A->hstream));
        if (probe_env_set("MGX_SEED_GROUPS")) {                 // A/B probe (needs a -DMGX_GRP_SEED_PROBE build of mgx_grp.hip)
            if (int rc = launch_groups(PH_SEED)) return fail(MGX_ERR_NO_DEVICE, "group seeding kernel: %d", rc);
        } else if (A->opt.seed_wps == 8 && l.Lmax <= 192 && slots >= (uint64_t)prop.multiProcessorCount * 4 * 8) {
            // (measured on 150-bp reads: the kernel is 10 % faster with 4944 B of LDS per wavefront than with 5056 B, although
            // both leave room for 32 wavefronts per CU; hence the wider margin)
            const uint32_t budget8 = (160u * 1024u) / (4 * MGX_SEED_WPS) - static_lds - 192u;
            uint32_t lds8 = std::min<uint32_t>(fast_lds_bytes(l.Lmax), budget8) & ~15u;
            lds8 = std::min<uint32_t>(lds8, probe_env_u32("MGX_SEED_LDS_CAP", lds8)) & ~15u;     // tuning probe
            if (probe_env_set("MGX_SEED_LDS_PRINT")) fprintf(stderr, "k_seed: static_lds %u budget8 %u lds8 %u (fast_lds_bytes %u)\n", static_lds, budget8, lds8, fast_lds_bytes(l.Lmax));
            if (A->dcfg.canonical >= 2) {
                if (int rc = mgx_launch_seed_primary(&P, (uint32_t)prop.multiProcessorCount * 4 * MGX_SEED_WPS, lds8, 1, A->hstream))
                    return fail(MGX_ERR_NO_DEVICE, "seeding kernel (PRIMARY): %d", rc);
            } else {
                // tuning probe: MGX_SEED_WAVES_PCT=50 launches half the resident wavefronts (is the kernel bound by what each
                // wavefront waits for, or by what all of them move?)
                const uint32_t spct = probe_env_pct("MGX_SEED_WAVES_PCT");
                k_align<PH_SEED, MGX_SEED_WPS><<<std::max(1u, (uint32_t)prop.multiProcessorCount * 4 * MGX_SEED_WPS * spct / 100), 64, lds8, A->hstream>>>(P, lds8);
            }
        } else if (A->dcfg.canonical >= 2) {
            if (int rc = mgx_launch_seed_primary(&P, w_slots, lds_bytes, 0, A->hstream)) return fail(MGX_ERR_NO_DEVICE, "seeding kernel (PRIMARY): %d", rc);
        } else {
            k_align<PH_SEED><<<w_slots, 64, lds_bytes, A->hstream>>>(P, lds_bytes);
        }
        HIP_TRY(hipGetLastError());
        P.seed_list = nullptr; P.n_items_ptr = nullptr;
        HIP_TRY(hipEventRecord(A->ev[4], A->hstream));
        k_iota<<<(uint32_t)((n + 255) / 256), 256, 0, A->hstream>>>(A->order_in.as<uint32_t>(), n);
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(A->sort_tmp.p, sort_tmp_bytes, A->work_key.as<uint32_t>(), A->work_key_sorted.as<uint32_t>(),
                                                   A->order_in.as<uint32_t>(), A->order.as<uint32_t>(), (int)n, 0, work_key_bits(n), A->hstream));
        HIP_TRY(hipMemsetAsync(cur + 1, 0, 8, A->hstream));                      // rewind the read cursor
        P.order = A->order.as<uint32_t>();
        HIP_TRY(hipEventRecord(A->ev[5], A->hstream));
        // Passes.  MGX_MULTI_PASS=1 (default when a batch carries many seeds per read, see below): a pass extends at most
        // one seed per read; reads with live seeds left write a resume record and are re-sorted by the work of their next
        // seed for the next pass (AlignParams::resume_*).  Reads of a wavefront then differ by one extension at most,
        // instead of waiting for the mate with the most seeds.  MGX_TWO_PASS=1: the older variant (pass 2 from scratch).
        const int two_pass_env = A->opt.two_pass;
        const int multi_env = A->opt.multi_pass;
        bool multi = multi_env == 1 && !labeled && !A->cfg.post_chain_alignments;      // (resume records hold num_alternative_paths alignments)
        if (multi_env < 0 && !labeled && !A->cfg.post_chain_alignments) {
            // automatic: worth it when reads run many extensions, i.e. carry many seeds (sub-k seeds of a pan-genome: ~100 per
            // read; a plain read: a handful).  The seeding kernel has finished counting them by now.
            unsigned long long seeds_total = 0;
            HIP_TRY(copy_sync(A, &seeds_total, cur + 2, 8, hipMemcpyDeviceToHost));
            multi = n > 0 && seeds_total / n >= 24;
        }
        A->n_passes = 1;
        A->lane_done = 0;
        memset(A->lane_hist_h, 0, sizeof(A->lane_hist_h));
        HIP_TRY(hipEventRecord(A->ev[6], A->hstream));
        bool nothing_left = false;
        if (lane_ok && !multi && n) {
            // every read through the lane kernel first, in the sorted order; what it lists goes on to the group kernel below
            if (int rc = A->lane_bail.ensure(n * 4 + 4)) return rc;
            if (int rc = A->lane_params.ensure(sizeof(LaneParams))) return rc;
            if (int rc = A->lane_hist.ensure(64 * 8)) return rc;          // ([32 .. 63]: section timers of -DMGX_LANE_TIMERS builds)
            HIP_TRY(hipMemsetAsync(A->lane_hist.p, 0, 64 * 8, A->hstream));
            HIP_TRY(hipMemsetAsync(cur + 5, 0, 16, A->hstream));
            LP.P = P;
            LP.P.n_items = n;
            LP.pk[0] = A->pk_fwd.as<uint64_t>(); LP.pk[1] = A->pk_rc.as<uint64_t>();
            LP.iv[0] = A->iv_fwd.as<uint32_t>(); LP.iv[1] = A->iv_rc.as<uint32_t>();
            LP.scratch = A->lane_scratch.as<uint8_t>();
            LP.tag_seed = ++A->lane_epoch * 0x632BE5ABu;
            LP.bail_list = A->lane_bail.as<uint32_t>();
            LP.bail_count = cur + 5;
            LP.done_count = cur + 6;
            LP.bail_hist = A->lane_hist.as<unsigned long long>();
            HIP_TRY(copy_sync(A, A->lane_params.p, &LP, sizeof(LP), hipMemcpyHostToDevice));
            uint32_t blocks = (uint32_t)std::min<uint64_t>(lane_blocks, (n + 63) / 64);
            // (measurement only: MGX_LANE_BLOCKS_PCT < 100 launches that share of the resident wavefronts — is the kernel bound by
            // the latency of its own dependent accesses, or by what the memory system serves per second?)
            if (const char *pct = getenv("MGX_LANE_BLOCKS_PCT")) blocks = std::max<uint32_t>(1u, (uint32_t)((uint64_t)blocks * (uint64_t)atoi(pct) / 100));
            if (int rc = lane_short ? mgx_launch_lane_short(A->lane_params.p, blocks, A->hstream) : mgx_launch_lane(A->lane_params.p, blocks, A->hstream)) return fail(MGX_ERR_NO_DEVICE, "lane kernel: %d", rc);
            A->kernels_ran |= MGX_KERNEL_LANE;
            ++g_kernel_launches[4];
            HIP_TRY(hipEventRecord(A->ev[6], A->hstream));
            unsigned long long counts[2] = { 0, 0 };
            HIP_TRY(copy_sync(A, counts, cur + 5, 16, hipMemcpyDeviceToHost));     // (synchronises with the kernel)
            A->lane_done = counts[1];
            HIP_TRY(copy_sync(A, A->lane_hist_h, A->lane_hist.p, 32 * 8, hipMemcpyDeviceToHost));
            if (getenv("MGX_LANE_TIMERS")) {
                unsigned long long t[16];
                HIP_TRY(copy_sync(A, t, A->lane_hist.as<unsigned long long>() + 32, sizeof(t), hipMemcpyDeviceToHost));
                fprintf(stderr, "k_lane timers (cycles of lane 0, summed over %u wavefronts): setup %llu children %llu column %llu node-table %llu commit %llu frontier %llu trace %llu result %llu | emit %llu kernel %llu\n",
                        blocks, t[0], t[1], t[2], t[7], t[3], t[4], t[5], t[6], t[8], t[9]);
            }
            HIP_TRY(hipMemsetAsync(cur + 1, 0, 8, A->hstream));                          // rewind the read cursor
            P.order = A->lane_bail.as<uint32_t>();
            P.n_items = counts[0];
            nothing_left = counts[0] == 0;
        }
        if (nothing_left) {
            // (every read finished in the lane kernel)
        } else if (multi) {
            const uint32_t rb = resume_rec_bytes(l, (uint32_t)std::max<uint64_t>(1, A->cfg.num_alternative_paths));
            size_t fb = 0, tb = 0;
            HIP_TRY(hipMemGetInfo(&fb, &tb));
            fb += g_pool.held_bytes();
            const uint64_t have = A->resume_pool[0].bytes + A->resume_pool[1].bytes;
            uint64_t cap = std::min<uint64_t>(n, ((uint64_t)fb / 2 + have) / (2ull * rb));      // two pools
            if (cap > 0xFFFFFFF0ull) cap = 0xFFFFFFF0ull;
            if (cap == 0) multi = false;
            if (multi) {
                for (int b = 0; b < 2; ++b) {
                    if (int rc = A->resume_pool[b].ensure(cap * rb, true)) return rc;
                    if (int rc = A->retry_key[b].ensure(n * 4 + 4)) return rc;
                }
                if (int rc = A->retry_list2.ensure(n * 4 + 4)) return rc;
                DevBuf *lists[2] = { &A->retry_list, &A->retry_list2 };
                P.seed_limit = 1;
                P.resume_rec_bytes = rb; P.resume_cap = (uint32_t)cap;
                P.retry_count = cur + 3;
                P.resume_in = nullptr; P.resume_reads = nullptr;
                int out = 0;
                uint64_t items = n;
                for (uint32_t pass = 0;; ++pass) {
                    P.resume_out = A->resume_pool[out].as<uint8_t>();
                    P.retry_list = lists[out]->as<uint32_t>();
                    P.retry_key = A->retry_key[out].as<uint32_t>();
                    // later passes take more seeds per read, so that the number of launches stays small
                    P.seed_limit = pass < 8 ? 1 : pass < 16 ? 4 : pass < 24 ? 16 : 0;
                    HIP_TRY((hipError_t)launch_groups(PH_EXTEND));
                    unsigned long long c = 0;
                    HIP_TRY(copy_sync(A, &c, cur + 3, 8, hipMemcpyDeviceToHost));     // (synchronises with the pass)
                    c = std::min<unsigned long long>(c, cap);
                    A->n_passes = pass + 1;
                    if (c == 0 || P.seed_limit == 0) break;
                    // next pass: the retry positions of this one, sorted by their work key
                    k_iota<<<(uint32_t)((c + 255) / 256), 256, 0, A->hstream>>>(A->order_in.as<uint32_t>(), c);
                    size_t tmp_bytes = sort_tmp_bytes;
                    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(A->sort_tmp.p, tmp_bytes, A->retry_key[out].as<uint32_t>(), A->work_key_sorted.as<uint32_t>(),
                                                               A->order_in.as<uint32_t>(), A->order.as<uint32_t>(), (int)c, 0, 12, A->hstream));
                    HIP_TRY(hipMemsetAsync(cur + 1, 0, 8, A->hstream));              // rewind the read cursor
                    HIP_TRY(hipMemsetAsync(cur + 3, 0, 8, A->hstream));              // and the retry counter
                    P.order = A->order.as<uint32_t>();
                    P.n_items = c;
                    P.resume_in = A->resume_pool[out].as<uint8_t>();
                    P.resume_reads = lists[out]->as<uint32_t>();
                    items = c;
                    out ^= 1;
                }
                (void)items;
            }
        }
        if (!multi && !nothing_left) {
            const bool two_pass = two_pass_env == 1 && !labeled;
            for (int pass = 0; pass < (two_pass ? 2 : 1); ++pass) {
                if (two_pass && pass == 0) {
                    P.seed_limit = 1;
                    P.retry_list = A->retry_list.as<uint32_t>();
                    P.retry_count = cur + 3;
                } else if (two_pass) {
                    HIP_TRY(hipMemsetAsync(cur + 1, 0, 8, A->hstream));              // rewind the read cursor
                    P.seed_limit = 0;
                    P.order = A->retry_list.as<uint32_t>();
                    P.n_items_ptr = cur + 3;
                }
                HIP_TRY((hipError_t)launch_groups(PH_EXTEND));
            }
        }
    }
    HIP_TRY(hipEventRecord(A->ev[3], A->hstream));
    return MGX_OK;
}

static int collect_stats(mgx_aligner *A, bool mapped, bool aligned) {
    HIP_TRY(hipStreamSynchronize(A->hstream));
    KernelStats ks, km;
    HIP_TRY(copy_sync(A, &ks, A->d_stats.p, sizeof(ks), hipMemcpyDeviceToHost));
    HIP_TRY(copy_sync(A, &km, A->d_stats_map.p, sizeof(km), hipMemcpyDeviceToHost));
    ks.rank_lines += km.rank_lines; ks.select_lines += km.select_lines; ks.bit_lines += km.bit_lines; ks.map_lines += km.map_lines;
    mgx_stats &s = A->hstats;
    memset(&s, 0, sizeof(s));
    s.n_reads = A->n_reads;
    s.n_rank_lines = ks.rank_lines; s.n_select_lines = ks.select_lines; s.n_bit_lines = ks.bit_lines;
    s.n_columns = ks.columns; s.n_extensions = ks.extensions; s.n_seeds = ks.seeds;
    s.n_map_lines = ks.map_lines; s.n_capacity_errors = ks.capacity_errors; s.n_seed_lines = ks.seed_lines;
    s.n_fast_columns = ks.fast_columns;
    s.extend_kernels = A->kernels_ran;
    s.n_lane_reads = A->lane_done;
    s.n_lane_lines = ks.lane_lines; s.n_lane_columns = ks.lane_columns;
    for (int x = 0; x < 32; ++x) s.lane_bail_reads[x] = A->lane_hist_h[x];
    if (aligned && A->seedlane_launched && getenv("MGX_SL_TIMERS")) {
        unsigned long long t[8];
        HIP_TRY(copy_sync(A, t, A->seedlane_hist.as<unsigned long long>() + 16, sizeof(t), hipMemcpyDeviceToHost));
        fprintf(stderr, "k_seed_lane timers (cycles of lane 0, summed over the wavefronts): strands %llu masks %llu scan %llu walks %llu dust %llu ranges+enumerate %llu publish %llu wave-mates %llu\n",
                t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
    }
    if (aligned && A->seedlane_launched) {
        unsigned long long done = 0, why[16];
        HIP_TRY(copy_sync(A, &done, A->cursors.as<unsigned long long>() + 9, 8, hipMemcpyDeviceToHost));
        HIP_TRY(copy_sync(A, why, A->seedlane_hist.p, sizeof(why), hipMemcpyDeviceToHost));
        s.n_seed_lane_reads = done;
        for (int x = 0; x < 16; ++x) s.seed_lane_left_reads[x] = why[x];
    }
    for (int x = 0; x < 8; ++x) { s.phase_cycles[x] = ks.cyc[x]; s.extend_cycles[x] = ks.xcyc[x]; }
    float ms = 0;
    if (mapped) { HIP_TRY(hipEventElapsedTime(&ms, A->ev[0], A->ev[1])); s.seed_kernel_ms = ms; }
    if (aligned) { HIP_TRY(hipEventElapsedTime(&ms, A->ev[2], A->ev[3])); s.align_kernel_ms = ms; }
    if (aligned && A->split_ran) {
        HIP_TRY(hipEventElapsedTime(&ms, A->ev[2], A->ev[4])); s.seeding_ms = ms;
        HIP_TRY(hipEventElapsedTime(&ms, A->ev[4], A->ev[5])); s.sort_ms = ms;
        HIP_TRY(hipEventElapsedTime(&ms, A->ev[5], A->ev[3])); s.extend_ms = ms;
        HIP_TRY(hipEventElapsedTime(&ms, A->ev[5], A->ev[6])); s.lane_ms = ms;
        if (A->seedlane_launched) { HIP_TRY(hipEventElapsedTime(&ms, A->ev[2], A->ev[7])); s.seed_lane_ms = ms; }
    }
    return MGX_OK;
}

int mgx_map_batch(mgx_aligner *A, const char *seqs, const uint64_t *offsets, uint64_t n, int on_device, mgx_mapping *out) {
    if (!A || !seqs || !offsets || !out) return fail(MGX_ERR_INVALID, "null argument");
    if (mgx_device_count() <= A->graph->device) return fail(MGX_ERR_NO_DEVICE, "no HIP device");
    HIP_TRY(hipSetDevice(A->graph->device));
    const char *d_seqs; const uint64_t *d_offsets; uint32_t Lmax;
    if (int rc = stage_batch(A, seqs, offsets, n, on_device, &d_seqs, &d_offsets, &Lmax)) return rc;
    if (int rc = run_map(A, d_seqs, d_offsets, n, true, true, Lmax)) return rc;
    if (int rc = collect_stats(A, true, false)) return rc;
    A->m_node_begin.resize(n + 1);
    HIP_TRY(copy_sync(A, A->m_node_begin.data(), A->node_begin.p, (n + 1) * 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> f(A->total_kmers), r(A->total_kmers);
    if (A->total_kmers) {
        HIP_TRY(copy_sync(A, f.data(), A->nodes_fwd.p, A->total_kmers * 4, hipMemcpyDeviceToHost));
        HIP_TRY(copy_sync(A, r.data(), A->nodes_rc.p, A->total_kmers * 4, hipMemcpyDeviceToHost));
    }
    A->m_fwd.assign(f.begin(), f.end());
    A->m_rc.assign(r.begin(), r.end());
    out->n_queries = n;
    out->node_begin = A->m_node_begin.data();
    out->nodes_fwd = A->m_fwd.data();
    out->nodes_rc = A->m_rc.data();
    return MGX_OK;
}

// the device blocks kept from destroyed aligners (DevPool) go back to the driver
int mgx_device_trim(int device) {
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    HIP_TRY(hipSetDevice(device));
    g_pool.flush();
    HIP_TRY(hipSetDevice(cur));
    return MGX_OK;
}

int mgx_aligner_set_stream(mgx_aligner *A, void *hip_stream) {
    if (!A) return fail(MGX_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(A->graph->device));
    HIP_TRY(hipStreamSynchronize(A->hstream));              // what the handle has in flight finishes where it was issued
    if (A->own_stream) { (void)hipStreamDestroy(A->hstream); A->own_stream = false; }
    A->hstream = (hipStream_t)hip_stream;
    return MGX_OK;
}
int mgx_aligner_create_stream(mgx_aligner *A) {
    if (!A) return fail(MGX_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(A->graph->device));
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (int rc = mgx_aligner_set_stream(A, st)) { (void)hipStreamDestroy(st); return rc; }
    A->own_stream = true;
    return MGX_OK;
}
void *mgx_aligner_get_stream(const mgx_aligner *A) { return A ? (void *)A->hstream : nullptr; }

int mgx_aligner_set_pipeline(mgx_aligner *A, const char *name) {
    if (!A) return fail(MGX_ERR_INVALID, "null argument");
    AlignMode m = parse_mode(name);
    // "+general" / "+chain" suffix-free test switches: the extension's register-resident chain path off / on
    if (name && !strcmp(name, "general")) { A->no_fast = true; return MGX_OK; }
    if (name && !strcmp(name, "chain")) { A->no_fast = false; return MGX_OK; }
    if (name && strchr(name, '=')) {
        const std::string key(name, strchr(name, '=') - name);
        const int v = atoi(strchr(name, '=') + 1);
        mgx_aligner::Options &o = A->opt;
        if (key == "ext64") o.ext64 = v;
        else if (key == "groups_per_wave") o.groups_per_wave = std::min(8, v);
        else if (key == "multi_pass") o.multi_pass = v;
        else if (key == "two_pass") o.two_pass = v;
        else if (key == "no_compact") o.no_compact = v;
        else if (key == "no_alias") o.no_alias = v;
        else if (key == "no_bt_runs") o.no_bt_runs = v;
        else if (key == "no_flat") o.no_flat = v;
        else if (key == "primary_alt_build") o.primary_alt_build = v;
        else if (key == "lane") o.lane = v;
        else if (key == "lane_short") o.lane_short = v;
        else if (key == "device_share") o.device_share = std::max(1, std::min(v, 64));
        else if (key == "map_pipe") o.map_pipe = v;
        else if (key == "seed_wps") o.seed_wps = v;
        else if (key == "seed_lane") o.seed_lane = v;
        else if (key == "retry_capacity") A->retry_capacity = v != 0;
        else return fail(MGX_ERR_INVALID, "unknown option '%s'", name);
        return MGX_OK;
    }
    if (m == MODE_BAD) return fail(MGX_ERR_INVALID, "unknown pipeline '%s'", name ? name : "(null)");
    A->mode = m;
    return MGX_OK;
}

// MGX_HOST_TIMERS=1: host-side wall time per stage of a batch on stderr (where does a one-read batch spend its 2 ms?)
struct HostStageTimer {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    static bool on() { static const bool v = probe_env_set("MGX_HOST_TIMERS"); return v; }
    explicit HostStageTimer(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~HostStageTimer() {
        if (on()) fprintf(stderr, "mgx host stage %-14s %8.1f us\n", name,
                          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
};

// kernels only; results stay in HBM
int mgx_align_batch_device(mgx_aligner *A, const char *seqs, const uint64_t *offsets, uint64_t n, int on_device) {
    if (!A || !seqs || !offsets) return fail(MGX_ERR_INVALID, "null argument");
    if (mgx_device_count() <= A->graph->device) return fail(MGX_ERR_NO_DEVICE, "no HIP device");
    HIP_TRY(hipSetDevice(A->graph->device));
    const char *d_seqs; const uint64_t *d_offsets; uint32_t Lmax;
    if (n == 0) { A->n_reads = 0; return MGX_OK; }
    { HostStageTimer t("stage_batch"); if (int rc = stage_batch(A, seqs, offsets, n, on_device, &d_seqs, &d_offsets, &Lmax)) return rc; }
    A->last_d_seqs = d_seqs; A->last_d_offsets = d_offsets; A->aligned_generation = A->stage_generation;
    bool mapped = A->cfg.max_seed_length >= A->graph->g.k;
    { HostStageTimer t("run_map"); if (int rc = run_map(A, d_seqs, d_offsets, n, A->dcfg.fwd_and_rc != 0, mapped, Lmax)) return rc; }
    for (;;) {
        { HostStageTimer t("run_align"); if (int rc = run_align(A, d_seqs, d_offsets, n, Lmax)) return rc; }
        { HostStageTimer t("collect_stats"); if (int rc = collect_stats(A, mapped, true)) return rc; }
        // both streams keep counting past their capacity: reads that found no room got a capacity status and the
        // stage is redone with what it asked for
        unsigned long long out_wanted = 0, seeds_wanted = 0;
        HIP_TRY(copy_sync(A, &out_wanted, A->cursors.as<unsigned long long>(), 8, hipMemcpyDeviceToHost));
        bool again = false;
        if (out_wanted > A->out_words) { A->out_min_words = out_wanted + out_wanted / 8 + 1024; again = true; }
        if (A->split_ran) {
            HIP_TRY(copy_sync(A, &seeds_wanted, A->cursors.as<unsigned long long>() + 2, 8, hipMemcpyDeviceToHost));
            if (seeds_wanted > A->seed_cap) {
                A->seed_scale = seeds_wanted / std::max<uint64_t>(1, A->seed_cap / A->seed_scale) + 2;
                again = true;
            }
        }
        if (!again) break;
    }
    A->h_results.clear();          // host copies of an earlier batch must not be mistaken for this one's
    return MGX_OK;
}

static int retry_capacity_queries(mgx_aligner *A, const char *d_seqs, const uint64_t *d_offsets, uint64_t n, mgx_results *out);

int mgx_fetch_results(mgx_aligner *A, mgx_results *out) {
    HostStageTimer t_fetch("fetch_results");
    const uint64_t n = A->n_reads;
    HIP_TRY(hipSetDevice(A->graph->device));
    A->h_results.resize(n);
    unsigned long long used = 0;
    if (n) {
        HIP_TRY(copy_sync(A, A->h_results.data(), A->results.p, n * sizeof(ReadResult), hipMemcpyDeviceToHost));
        HIP_TRY(copy_sync(A, &used, A->cursors.p, 8, hipMemcpyDeviceToHost));
        used = std::min<unsigned long long>(used, A->out_words);
    }
    A->h_stream.resize(used);
    if (used) HIP_TRY(copy_sync(A, A->h_stream.data(), A->stream.p, used * 4, hipMemcpyDeviceToHost));
    A->host.decode(A->h_results.data(), n, A->h_stream.data(), ~0ull, A->anno != nullptr);
    A->host.view(out);
    // the aligned batch is read back below (post-chaining, the capacity retry): only while the staging buffers still hold it —
    // mgx_map_batch on this aligner re-stages them; a caller's device buffers are the caller's to keep until the fetch
    const bool batch_intact = A->aligned_generation == A->stage_generation && A->last_d_seqs && A->last_d_offsets;
    if (A->cfg.post_chain_alignments && n) {
        // chain_alignments (dbg_aligner.cpp:328-332) on the host: needs the reads of the queries with two or more alignments once
        // more — one transfer of the span of the batch that holds them (none at all when no query has two)
        uint64_t q_lo = n, q_hi = 0;
        for (uint64_t q = 0; q < n; ++q) if (out->aln_begin[q + 1] - out->aln_begin[q] >= 2) { q_lo = std::min(q_lo, q); q_hi = q + 1; }
        if (!batch_intact)
            return fail(MGX_ERR_INVALID, "post_chain_alignments: another batch was staged on this aligner between mgx_align_batch_device and mgx_fetch_results");
        std::vector<uint64_t> offs(n + 1, 0);
        HIP_TRY(copy_sync(A, offs.data(), A->last_d_offsets, (n + 1) * 8, hipMemcpyDeviceToHost));
        std::vector<char> reads(1);
        uint64_t b0 = 0;
        if (q_lo < q_hi) {
            b0 = offs[q_lo];
            const uint64_t b1 = offs[q_hi];
            reads.resize(b1 - b0 + 1);
            if (b1 > b0) HIP_TRY(copy_sync(A, reads.data(), A->last_d_seqs + b0, b1 - b0, hipMemcpyDeviceToHost));
        }
        mgx_results plain = *out;
        chain_results(plain, reads.data(), offs.data(), A->cfg, A->graph->g.k, &A->host_chained, b0);
        A->host_chained.view(out);
    }
    // reads the batch's limits were too small for: once more, with larger ones (below)
    if (!batch_intact) return MGX_OK;              // (statuses stay: the reads are no longer where they were)
    return retry_capacity_queries(A, A->last_d_seqs, A->last_d_offsets, n, out);
}

int mgx_device_results(mgx_aligner *A, const void **headers, uint64_t *header_bytes, uint64_t *n_queries,
                       const void **stream, uint64_t *stream_words) {
    if (!A) return fail(MGX_ERR_INVALID, "null argument");
    unsigned long long used = 0;
    if (A->n_reads) HIP_TRY(copy_sync(A, &used, A->cursors.p, 8, hipMemcpyDeviceToHost));
    used = std::min<unsigned long long>(used, A->out_words);
    *headers = A->results.p; *header_bytes = sizeof(ReadResult); *n_queries = A->n_reads;
    *stream = A->stream.p; *stream_words = used;
    return MGX_OK;
}

uint64_t mgx_device_stream_capacity(const mgx_aligner *A) { return A ? A->out_words : 0; }

struct mgx_raw_store { HostResults host; };

int mgx_results_from_raw(const void *headers, uint64_t n, const uint32_t *stream, uint64_t stream_words,
                         mgx_raw_store **store, mgx_results *out) {
    return mgx_results_from_raw_labeled(headers, n, stream, stream_words, 0, store, out);
}
int mgx_results_from_raw_labeled(const void *headers, uint64_t n, const uint32_t *stream, uint64_t stream_words, int labeled,
                                 mgx_raw_store **store, mgx_results *out) {
    if ((!headers && n) || !store || !out || (!stream && stream_words)) return fail(MGX_ERR_INVALID, "null argument");
    const ReadResult *rr = static_cast<const ReadResult *>(headers);
    auto *S = new mgx_raw_store();
    if (!S->host.decode(rr, n, stream, stream_words, labeled != 0)) {
        delete S;
        return fail(MGX_ERR_INVALID, "a record points outside the stream (%llu words)", (unsigned long long)stream_words);
    }
    S->host.view(out);
    *store = S;
    return MGX_OK;
}

void mgx_raw_store_free(mgx_raw_store *store) { delete store; }

// chain_alignments (aligner_chainer.cpp:555-720) over decoded results — what mgx_fetch_results does itself when the aligner's
// config has post_chain_alignments set; for results decoded elsewhere (mgx_results_from_raw after a gather).  Host code.
int mgx_chain_alignments(const mgx_config *config, uint32_t k, const mgx_results *in, const char *seqs, const uint64_t *offsets,
                         mgx_raw_store **store, mgx_results *out) {
    if (!config || !in || !store || !out || (in->n_queries && (!seqs || !offsets))) return fail(MGX_ERR_INVALID, "null argument");
    if (k < 2) return fail(MGX_ERR_INVALID, "k out of range");
    auto *S = new mgx_raw_store();
    chain_results(*in, seqs, offsets, *config, k, &S->host);
    S->host.view(out);
    *store = S;
    return MGX_OK;
}

// Queries whose per-read arenas overflowed (status MGX_ERR_CAPACITY: the reference has no such limit, its tables grow on the
// heap) are aligned again by a temporary aligner with doubled limits, up to six doublings, and take their place in the
// results — what the C++ adapter (host/hip_dbg_aligner.hpp) did for its callers since round 2, now for every caller of
// mgx_fetch_results / mgx_align_batch (the Python binding and the device-resident batches among them).  What doubling cannot cure stays a capacity status: a query with more
// alignments than the post_chain_alignments queue holds.  (Label-aware alignment: the retry also doubles the label arenas —
// alignments per backtracking, aggregator pool, label queues, label sets; derive_limits' label_scale.)
static int retry_capacity_queries(mgx_aligner *A, const char *d_seqs, const uint64_t *d_offsets, uint64_t n, mgx_results *out) {
    std::vector<uint64_t> todo;
    // (a query whose post_chain_alignments queue overflowed is flagged by the kernel — RR_CAUSE_QUEUE — and stays a status: no
    // limit cures it, and six rounds of ever larger arenas would be paid for nothing)
    for (uint64_t q = 0; q < n; ++q)
        if (out->status[q] == MGX_ERR_CAPACITY && !(q < A->h_results.size() && A->h_results[q].orientation == RR_CAUSE_QUEUE)) todo.push_back(q);
    if (todo.empty() || !A->retry_capacity || !d_seqs || !d_offsets) return MGX_OK;
    // the reads in question, from the batch as the device holds it (the caller's buffers, or the upload of a host batch): ONE
    // transfer of the span of the batch between the first and the last of them
    std::vector<uint64_t> h_off(n + 1);
    HIP_TRY(copy_sync(A, h_off.data(), d_offsets, (n + 1) * 8, hipMemcpyDeviceToHost));
    const uint64_t b0 = h_off[todo.front()], b1 = h_off[todo.back() + 1];
    std::vector<char> h_seq(b1 - b0 + 1);
    if (b1 > b0) HIP_TRY(copy_sync(A, h_seq.data(), d_seqs + b0, b1 - b0, hipMemcpyDeviceToHost));
    const char *seqs = h_seq.data();
    auto seq_of = [&](uint64_t q) { return seqs + (h_off[q] - b0); };
    const uint64_t *offsets = h_off.data();
    // results so far, query by query (replaced below)
    std::vector<HostResults> fixed(todo.size());
    std::vector<uint8_t> have(todo.size(), 0);
    mgx_limits lim;
    mgx_aligner_get_limits(A, &lim);
    mgx_aligner *tmp = nullptr;
    struct Guard { mgx_aligner *&p; ~Guard() { if (p) mgx_aligner_destroy(p); } } guard{ tmp };
    std::vector<uint64_t> pending(todo.size());
    for (size_t t = 0; t < todo.size(); ++t) pending[t] = t;
    for (int attempt = 0; attempt < 6 && !pending.empty(); ++attempt) {
        lim.max_query_length = 0;                       // the (smaller) batch sets its own
        lim.max_columns = lim.max_columns * 2;
        lim.max_seeds = std::min<uint32_t>(65535u, lim.max_seeds * 2);
        lim.cell_arena_bytes = lim.cell_arena_bytes * 2;
        if (tmp) { mgx_aligner_destroy(tmp); tmp = nullptr; }
        // (an attempt that fails — out of memory at the inflated limits, say — ends the retry: what the batch and the earlier
        // attempts produced stays valid and is what the caller gets, statuses included)
        if (aligner_create(A->graph, &A->cfg, &lim, A->anno, &tmp) != MGX_OK) break;
        tmp->opt = A->opt; tmp->mode = A->mode; tmp->no_fast = A->no_fast;
        tmp->hstream = A->hstream;                        // (borrowed: the retry runs where the batch ran)
        tmp->retry_capacity = false;
        tmp->label_scale = 2u << attempt;
        std::string blob;
        std::vector<uint64_t> offs(pending.size() + 1, 0);
        for (size_t t = 0; t < pending.size(); ++t) {
            const uint64_t q = todo[pending[t]];
            blob.append(seq_of(q), offsets[q + 1] - offsets[q]);
            offs[t + 1] = blob.size();
        }
        mgx_results sub{};
        if (mgx_align_batch(tmp, blob.data(), offs.data(), pending.size(), 0, &sub) != MGX_OK) break;
        std::vector<uint64_t> again;
        for (size_t t = 0; t < pending.size(); ++t) {
            if (sub.status[t] == MGX_ERR_CAPACITY) {
                // (its queue overflowed this time: final)
                if (!(t < tmp->h_results.size() && tmp->h_results[t].orientation == RR_CAUSE_QUEUE)) again.push_back(pending[t]);
                continue;
            }
            fixed[pending[t]].append_query(sub, t);
            have[pending[t]] = 1;
        }
        pending.swap(again);
        mgx_aligner_get_limits(tmp, &lim);
    }
    bool any = false;
    for (uint8_t h : have) any |= h != 0;
    if (!any) return MGX_OK;
    HostResults merged;
    size_t t = 0;
    for (uint64_t q = 0; q < n; ++q) {
        if (t < todo.size() && todo[t] == q) {
            if (have[t]) { mgx_results v; fixed[t].view(&v); merged.append_query(v, 0); }
            else merged.append_query(*out, q);
            ++t;
        } else merged.append_query(*out, q);
    }
    A->host_retried = std::move(merged);
    A->host_retried.view(out);
    A->hstats.n_capacity_retried = (uint64_t)std::count(have.begin(), have.end(), (uint8_t)1);
    return MGX_OK;
}

int mgx_align_batch(mgx_aligner *A, const char *seqs, const uint64_t *offsets, uint64_t n, int on_device, mgx_results *out) {
    if (!out) return fail(MGX_ERR_INVALID, "null argument");
    if (int rc = mgx_align_batch_device(A, seqs, offsets, n, on_device)) return rc;
    return mgx_fetch_results(A, out);          // (with the capacity retry)
}

// test hook: keep the per-read seed lists of the next batches (device -> mgx_fetch_seeds)
void mgx_aligner_keep_seeds(mgx_aligner *A, int keep) { A->keep_seeds = keep != 0; }

// per read: num_matches fwd/rc, n_seeds fwd/rc, n_extensions, n_columns (6 x u32), and optionally the seeds
int mgx_fetch_seed_info(mgx_aligner *A, uint32_t *info6, uint32_t *seeds /* [n][2][max_seeds][4] or NULL */, uint32_t *max_seeds_out) {
    const uint64_t n = A->n_reads;
    if (A->h_results.size() != n || n == 0) {
        A->h_results.resize(n);
        if (n) HIP_TRY(copy_sync(A, A->h_results.data(), A->results.p, n * sizeof(ReadResult), hipMemcpyDeviceToHost));
    }
    for (uint64_t i = 0; i < n; ++i) {
        const ReadResult &r = A->h_results[i];
        uint32_t *o = info6 + 6 * i;
        o[0] = r.num_matches_fwd; o[1] = r.num_matches_rc; o[2] = r.n_seeds_fwd; o[3] = r.n_seeds_rc;
        o[4] = r.n_extensions; o[5] = r.n_columns;
    }
    if (max_seeds_out) *max_seeds_out = A->lim.max_seeds;
    if (seeds && A->keep_seeds && n) {
        std::vector<DevSeed> h(n * 2 * (uint64_t)A->lim.max_seeds);
        HIP_TRY(copy_sync(A, h.data(), A->dbg_seeds.p, h.size() * sizeof(DevSeed), hipMemcpyDeviceToHost));
        for (size_t x = 0; x < h.size(); ++x) {
            seeds[4 * x] = h[x].clipping; seeds[4 * x + 1] = h[x].length; seeds[4 * x + 2] = h[x].offset;
            seeds[4 * x + 3] = h[x].offset ? h[x].node : h[x].n_nodes;
        }
    }
    return MGX_OK;
}

int mgx_aligner_stats(const mgx_aligner *A, mgx_stats *out) { *out = A->hstats; return MGX_OK; }
void mgx_kernel_launch_counts(uint64_t *out5) { for (int x = 0; x < 5; ++x) out5[x] = g_kernel_launches[x].load(); }

size_t mgx_format_tsv(const mgx_results *res, uint64_t qi, const char *header, const char *query, size_t query_len,
                      int32_t min_path_score, char *buf, size_t buf_len) {
    return mgx_format_tsv_labeled(res, qi, header, query, query_len, min_path_score, nullptr, 0, buf, buf_len);
}
size_t mgx_format_tsv_labeled(const mgx_results *res, uint64_t qi, const char *header, const char *query, size_t query_len,
                              int32_t min_path_score, const char *const *label_names, uint32_t n_label_names,
                              char *buf, size_t buf_len) {
    // cli/align.cpp:262-285 + alignment.hpp:426-433; the query is printed normalised (AlignmentResults ctor)
    std::string s(header);
    s += '\t';
    for (size_t i = 0; i < query_len; ++i) {
        int8_t c = (int8_t)query[i];
        s += c >= 0 ? (char)toupper(c) : (char)127;
    }
    if (res->aln_begin[qi] == res->aln_begin[qi + 1]) {
        s += "\t*\t*\t" + std::to_string(min_path_score) + "\t*\t*\t*\n";
    } else {
        static const char ops[] = "SX=DIG";
        for (uint64_t ai = res->aln_begin[qi]; ai < res->aln_begin[qi + 1]; ++ai) {
            const mgx_alignment &a = res->alignments[ai];
            s += a.orientation ? "\t-\t" : "\t+\t";
            s.append(res->seqs + a.seq_begin, a.seq_len);
            s += '\t' + std::to_string(a.score) + '\t' + std::to_string(a.num_matches) + '\t';
            for (uint32_t x = 0; x < a.n_cigar; ++x) {
                const mgx_cigar_op &op = res->cigar[a.cigar_begin + x];
                s += std::to_string(op.len) + ops[op.op];
            }
            s += '\t' + std::to_string(a.offset);
            if (res->labels && a.n_labels) {
                // cli/align.cpp:274-281: the label names (LabelEncoder::decode), joined by ';'
                s += '\t';
                for (uint32_t x = 0; x < a.n_labels; ++x) {
                    const uint32_t lbl = res->labels[a.labels_begin + x];
                    if (x) s += ';';
                    if (label_names && lbl < n_label_names) s += label_names[lbl];
                    else s += std::to_string(lbl);
                }
            }
        }
        s += '\n';
    }
    if (buf && buf_len) {
        size_t nc = std::min(buf_len - 1, s.size());
        memcpy(buf, s.data(), nc);
        buf[nc] = 0;
    }
    return s.size();
}

// ---- metagraph align --json (cli/align.cpp:287-305): Alignment::to_json (alignment.cpp:883-963) + path_json (:704-881),
// written the way Json::writeString does with indentation "" (jsoncpp: object keys in lexicographic order, no white
// space, doubles as %.17g with ".0" appended to integral values, strings with the standard escapes).
namespace {
void json_string(std::string &o, const char *p, size_t n) {
    o += '"';
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)p[i];
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            default:
                if (c < 0x20 || c >= 0x7F) { char b[8]; snprintf(b, sizeof(b), "\\u%04X", (unsigned)c); o += b; }
                else o += (char)c;
        }
    }
    o += '"';
}
void json_double(std::string &o, double v) {
    char b[40];
    snprintf(b, sizeof(b), "%.17g", v);
    o += b;
    if (!strpbrk(b, ".eEn")) o += ".0";
}
// one edit object: keys from_length, sequence, to_length (only those set)
void json_edit(std::string &o, bool &first, long long from_len, const char *seq, size_t seq_len, long long to_len) {
    if (!first) o += ',';
    first = false;
    o += '{';
    bool f2 = true;
    if (from_len >= 0) { o += "\"from_length\":" + std::to_string(from_len); f2 = false; }
    if (seq) { if (!f2) o += ','; o += "\"sequence\":"; json_string(o, seq, seq_len); f2 = false; }
    if (to_len >= 0) { if (!f2) o += ','; o += "\"to_length\":" + std::to_string(to_len); }
    o += '}';
}
}  // namespace

size_t mgx_format_json(const mgx_results *res, uint64_t qi, const char *header, const char *query, size_t query_len,
                       uint32_t k, char *buf, size_t buf_len) {
    // the query as AlignmentResults keeps it (alignment.cpp:1348-1372): upper case, bytes < 0 -> 127; and its reverse
    // complement (COMPL_TAB, reverse_complement.hpp:31-62)
    std::string fwd(query_len, 0), rc(query_len, 0);
    for (size_t i = 0; i < query_len; ++i) {
        const int8_t c = (int8_t)query[i];
        fwd[i] = c >= 0 ? (char)toupper(c) : (char)127;
    }
    for (size_t i = 0; i < query_len; ++i) {
        const unsigned char c = (unsigned char)fwd[query_len - 1 - i];
        static const char up[] = "TVGHEFCDIJMLKNOPQYSAABWXRZ";
        rc[i] = (c >= 'A' && c <= 'Z') ? up[c - 'A'] : (c >= 'a' && c <= 'z') ? (char)(up[c - 'a'] + 32) : c == 96 ? (char)64 : (char)c;
    }
    std::string s;
    const size_t hl = strlen(header);
    if (res->aln_begin[qi] == res->aln_begin[qi + 1]) {
        // Alignment().to_json: an empty alignment carries its name and an empty sequence
        s += "{\"name\":"; json_string(s, header, hl); s += ",\"sequence\":\"\"}\n";
    }
    for (uint64_t ai = res->aln_begin[qi]; ai < res->aln_begin[qi + 1]; ++ai) {
        const mgx_alignment &a = res->alignments[ai];
        const bool secondary = ai != res->aln_begin[qi];
        const std::string &full = a.orientation ? rc : fwd;
        const char *qv = full.data() + a.clipping;                     // query_view_
        const size_t qv_len = query_len - a.clipping - a.end_clipping;
        const mgx_cigar_op *cg = res->cigar + a.cigar_begin;
        // mgx_cigar_op.op: 0 clipped, 1 mismatch, 2 match, 3 deletion, 4 insertion, 5 node insertion (aligner_cigar.hpp:19-26)
        static const char opc[] = "SX=DIG";
        s += "{\"annotation\":{\"cigar\":\"";
        for (uint32_t x = 0; x < a.n_cigar; ++x) s += std::to_string(cg[x].len) + opc[cg[x].op];
        s += '"';
        if (a.seq_len) { s += ",\"ref_sequence\":"; json_string(s, res->seqs + a.seq_begin, a.seq_len); }
        s += "},\"identity\":";
        json_double(s, qv_len ? (double)a.num_matches / (double)qv_len : 0.0);
        if (secondary) s += ",\"is_secondary\":true";
        s += ",\"name\":"; json_string(s, header, hl);
        if (a.n_nodes) {
            // path_json: mappings of the first node (may cover up to node_size characters), then one per further node
            const uint64_t *nodes = res->nodes + a.nodes_begin;
            uint32_t ci = 0;
            if (a.n_cigar && cg[0].op == 0) ++ci;
            uint64_t c_off = 0;
            const char *qs = qv;
            s += ",\"path\":{";
            if (nodes[0] == nodes[a.n_nodes - 1]) s += "\"is_circular\":true,";
            s += "\"length\":" + std::to_string(a.n_nodes) + ",\"mapping\":[";
            long long rank = 1;
            size_t cur = a.offset;
            {
                s += "{\"edit\":[";
                bool first = true;
                while (cur < k && ci < a.n_cigar) {
                    size_t next_pos = std::min<size_t>(k, cur + (cg[ci].len - c_off));
                    const size_t next_size = next_pos - cur;
                    const uint32_t op = cg[ci].op;
                    if (op == 0) { ++ci; c_off = 0; continue; }               // trailing clip
                    if (op == 1) { json_edit(s, first, (long long)next_size, qs, next_size, (long long)next_size); qs += next_size; }
                    else if (op == 4) { json_edit(s, first, -1, qs, next_size, (long long)next_size); qs += next_size; next_pos = cur; }
                    else if (op == 3) json_edit(s, first, (long long)next_size, nullptr, 0, -1);
                    else if (op == 2) { json_edit(s, first, (long long)next_size, nullptr, 0, (long long)next_size); qs += next_size; }
                    c_off += next_size;
                    cur = next_pos;
                    if (c_off == cg[ci].len) { ++ci; c_off = 0; }
                }
                s += "],\"position\":{\"node_id\":" + std::to_string(nodes[0]);
                if (a.offset) s += ",\"offset\":" + std::to_string(a.offset);
                s += "},\"rank\":" + std::to_string(rank++) + "}";
            }
            for (uint32_t ni = 1; ni < a.n_nodes && ci < a.n_cigar; ++ni) {
                s += ",{\"edit\":[";
                bool first = true;
                if (cg[ci].op == 4 || cg[ci].op == 0) {
                    const size_t len = cg[ci].len - c_off;
                    json_edit(s, first, -1, qs, len, (long long)len);
                    qs += len;
                    ++ci; c_off = 0;
                }
                if (ci < a.n_cigar) {
                    const uint32_t op = cg[ci].op;
                    if (op == 1) { json_edit(s, first, 1, qs, 1, 1); ++qs; }
                    else if (op == 3) json_edit(s, first, 1, nullptr, 0, -1);
                    else if (op == 2) { json_edit(s, first, 1, nullptr, 0, 1); ++qs; }
                    if (++c_off == cg[ci].len) { c_off = 0; ++ci; }
                }
                s += "],\"position\":{\"node_id\":" + std::to_string(nodes[ni]) + ",\"offset\":" + std::to_string(k - 1);
                s += "},\"rank\":" + std::to_string(rank++) + "}";
            }
            s += "],\"name\":\"\"}";
        }
        if (a.clipping) s += ",\"query_position\":" + std::to_string(a.clipping);
        s += std::string(",\"read_mapped\":") + (qv_len ? "true" : "false");
        if (a.orientation) s += ",\"read_on_reverse_strand\":true";
        s += ",\"score\":" + std::to_string(a.score);
        s += ",\"sequence\":"; json_string(s, full.data(), full.size());
        if (a.clipping) s += ",\"soft_clipped\":true";
        s += "}\n";
    }
    if (buf && buf_len) {
        size_t nc = std::min(buf_len - 1, s.size());
        memcpy(buf, s.data(), nc);
        buf[nc] = 0;
    }
    return s.size();
}

} // extern "C"
