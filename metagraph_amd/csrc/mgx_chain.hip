// mgx_chain.hip — chain_seeds (A/aligner_chainer.cpp:341-542) on the device: the anchor DP of seed chaining, the one vectorised
// loop of the chaining half of SURVEY 8 row f3 (round 6).
//
// The reference sorts a (query, strand)'s anchors — one per (seed, label, coordinate) — by label, then by DECREASING reference
// coordinate, and lets every anchor i offer itself as the chain predecessor-on-the-reference of the 64 anchors behind it in
// that order (bandwidth 65), eight at a time with AVX2 gathers.  Here one wavefront takes one anchor list and its 64 lanes
// take the 64 candidates j = i + 1 .. i + 64 of the current i at once: the reference's inner loop, eight lanes wide there, is
// exactly one wavefront wide here.  i advances sequentially (the score of anchor i must be final when it is offered); the
// scores and back pointers of the list live in LDS for the pass.
//
// Arithmetic (bit-exactness): integer except the gap cost, which the reference computes in FLOAT — coord_diff * sl +
// log2f(coord_diff + 1) * 0.5, rounded to the nearest integer by cvtps_epi32 — through the host's libm.  The device does not
// call its own log2f (another implementation may round a last bit differently): the cost is a function of coord_diff alone,
// 0 <= coord_diff < query size, so the HOST tabulates it with the reference's expression and the kernel looks it up.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mgx.h"

extern "C" int mgx_device_count(void);
extern "C" void mgx_set_last_error(const char *msg);

namespace {

int cfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    mgx_set_last_error(buf);
    return code;
}
#define HIP_TRY_CH(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { cleanup(); return cfail(MGX_ERR_NO_DEVICE, "%s: %s", #e, hipGetErrorString(r_)); } } while (0)

constexpr uint32_t CHAIN_MAX_LIST = 6144;       // anchors of one list the kernel keeps in LDS (8 B each)
constexpr uint32_t kNid = 0xFFFFFFFFu;

// sort keys of one LSD pass, gathered through the permutation so far
__global__ void k_chain_keys32(const mgx_chain_anchor *a, const uint32_t *perm, uint64_t n, int field, int32_t *keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mgx_chain_anchor &e = a[perm[i]];
    keys[i] = field == 0 ? e.seed_end : e.seed_clipping;
}
__global__ void k_chain_keys64(const mgx_chain_anchor *a, const uint32_t *perm, uint64_t n, int field, uint64_t *keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mgx_chain_anchor &e = a[perm[i]];
    // (coordinates are signed: the sign bit flipped gives the unsigned order)
    keys[i] = field == 0 ? ((uint64_t)e.coordinate ^ 0x8000000000000000ull) : e.label;
}
__global__ void k_chain_iota(uint32_t *perm, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = (uint32_t)i;
}
__global__ void k_chain_permute(const mgx_chain_anchor *a, const uint32_t *perm, uint64_t n, mgx_chain_anchor *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[perm[i]];
}

// one wavefront per anchor list (sorted): the banded DP; chain_score of the anchors is updated in place, back[] = index in the
// sorted list (relative to the list's begin) or kNid
__global__ void __launch_bounds__(64) k_chain_dp(mgx_chain_anchor *anchors, const uint64_t *list_begin, const uint32_t *query_size,
                                                 uint64_t n_lists, const int32_t *penalty, uint32_t penalty_len, uint32_t *back) {
    __shared__ int32_t s_score[CHAIN_MAX_LIST];
    __shared__ uint32_t s_back[CHAIN_MAX_LIST];
    const uint64_t li = blockIdx.x;
    if (li >= n_lists) return;
    const uint64_t b0 = list_begin[li];
    const uint32_t n = (uint32_t)(list_begin[li + 1] - b0);
    mgx_chain_anchor *A = anchors + b0;
    const int32_t qs = (int32_t)query_size[li];
    const int lane = (int)threadIdx.x;
    for (uint32_t x = lane; x < n; x += 64) { s_score[x] = A[x].chain_score; s_back[x] = kNid; }
    __syncthreads();
    // the label groups: anchors of equal label are contiguous; a group's end is found by its first lane-visible change
    uint32_t i = 0;
    while (i < n) {
        // cur_label_end: the end of the label group that starts at i (uniform scan, 64 anchors at a time)
        const uint64_t label = A[i].label;
        uint32_t end = i + 1;
        for (;;) {
            const uint32_t x = end + (uint32_t)lane;
            const bool same = x < n && A[x].label == label;
            const uint64_t m = __ballot(same);
            if (m == ~0ull) { end += 64; continue; }
            end += (uint32_t)__builtin_ctzll(~m);
            break;
        }
        for ( ; i < end; ++i) {
            const int64_t prev_coord = A[i].coordinate;
            const int32_t prev_clipping = A[i].seed_clipping;
            if (!prev_clipping) continue;                                   // (uniform)
            const int32_t prev_score = s_score[i];
            const uint32_t it_end = min(65u, end - i) + i;
            const uint32_t j = i + 1 + (uint32_t)lane;
            if (j < it_end) {
                const mgx_chain_anchor e = A[j];
                const int64_t coord_cutoff = prev_coord - (int64_t)qs;
                // (coordinates fall along the list: past the first anchor below the cut-off every later one is below it too)
                if (!(coord_cutoff > e.coordinate)) {
                    const int32_t dist = prev_clipping - e.seed_clipping;
                    const int32_t coord_dist = (int32_t)(prev_coord - e.coordinate);
                    if (dist > 0 && max(dist, coord_dist) < qs) {
                        const int32_t match = min(min(dist, coord_dist), e.seed_end - e.seed_clipping);
                        int32_t cur = prev_score + match;
                        const int32_t coord_diff = abs(coord_dist - dist);
                        if (coord_diff > 0) cur -= penalty[min((uint32_t)coord_diff, penalty_len - 1)];
                        if (cur >= s_score[j]) { s_score[j] = cur; s_back[j] = i; }
                    }
                }
            }
            __syncthreads();                                                 // (one wavefront: orders the LDS writes before the next i's reads)
        }
    }
    for (uint32_t x = lane; x < n; x += 64) { A[x].chain_score = s_score[x]; back[b0 + x] = s_back[x]; }
}

} // namespace

extern "C" {

/* chain_seeds for n_lists anchor lists (include/mgx.h). */
int mgx_chain_seeds(const mgx_config *config, int device, const mgx_chain_anchor *anchors, const uint64_t *list_begin,
                    const uint32_t *query_size, uint64_t n_lists, mgx_chain_anchor *sorted_out, uint32_t *backtrace_out) {
    if (!config || !list_begin || !query_size || (n_lists && (!sorted_out || !backtrace_out))) return cfail(MGX_ERR_INVALID, "mgx_chain_seeds: null argument");
    const uint64_t n = n_lists ? list_begin[n_lists] : 0;
    if (n && !anchors) return cfail(MGX_ERR_INVALID, "mgx_chain_seeds: null argument");
    if (n >= 0x7FFFFFF0ull || n_lists >= 0x7FFFFFF0ull)       // (the segmented sort counts items and segments in int)
        return cfail(MGX_ERR_UNSUPPORTED, "mgx_chain_seeds: %llu anchors in %llu lists in one call", (unsigned long long)n, (unsigned long long)n_lists);
    uint32_t max_q = 1;
    for (uint64_t l = 0; l < n_lists; ++l) {
        if (list_begin[l] > list_begin[l + 1]) return cfail(MGX_ERR_INVALID, "mgx_chain_seeds: list_begin must be ascending");
        if (list_begin[l + 1] - list_begin[l] > CHAIN_MAX_LIST)
            return cfail(MGX_ERR_UNSUPPORTED, "mgx_chain_seeds: a list of %llu anchors (at most %u per list on the device)",
                         (unsigned long long)(list_begin[l + 1] - list_begin[l]), CHAIN_MAX_LIST);
        max_q = std::max(max_q, query_size[l]);
    }
    if (mgx_device_count() <= device) return cfail(MGX_ERR_NO_DEVICE, "no HIP device");
    if (!n) return MGX_OK;
    // the gap cost by coordinate difference, with the reference's float expression (aligner_chainer.cpp:404,466-481) on the host
    const float sl = (float)((double)(float)config->min_seed_length * 0.01);
    std::vector<int32_t> penalty(max_q + 1, 0);
    for (uint32_t d = 1; d <= max_q; ++d)
        penalty[d] = (int32_t)std::lrintf((float)d * sl + std::log2((float)(d + 1)) * 0.5f);
    mgx_chain_anchor *d_a = nullptr, *d_s = nullptr;
    uint64_t *d_lb = nullptr, *d_k64 = nullptr, *d_k64o = nullptr;
    uint32_t *d_qs = nullptr, *d_perm = nullptr, *d_perm2 = nullptr, *d_back = nullptr;
    int32_t *d_k32 = nullptr, *d_k32o = nullptr, *d_pen = nullptr;
    void *d_tmp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_a); (void)hipFree(d_s); (void)hipFree(d_lb); (void)hipFree(d_k64); (void)hipFree(d_k64o); (void)hipFree(d_qs);
        (void)hipFree(d_perm); (void)hipFree(d_perm2); (void)hipFree(d_back); (void)hipFree(d_k32); (void)hipFree(d_k32o); (void)hipFree(d_pen); (void)hipFree(d_tmp);
    };
    HIP_TRY_CH(hipSetDevice(device));
    HIP_TRY_CH(hipMalloc(&d_a, n * sizeof(mgx_chain_anchor)));
    HIP_TRY_CH(hipMalloc(&d_s, n * sizeof(mgx_chain_anchor)));
    HIP_TRY_CH(hipMalloc(&d_lb, (n_lists + 1) * 8));
    HIP_TRY_CH(hipMalloc(&d_qs, n_lists * 4));
    HIP_TRY_CH(hipMalloc(&d_perm, n * 4));
    HIP_TRY_CH(hipMalloc(&d_perm2, n * 4));
    HIP_TRY_CH(hipMalloc(&d_back, n * 4));
    HIP_TRY_CH(hipMalloc(&d_k32, n * 4));
    HIP_TRY_CH(hipMalloc(&d_k32o, n * 4));
    HIP_TRY_CH(hipMalloc(&d_k64, n * 8));
    HIP_TRY_CH(hipMalloc(&d_k64o, n * 8));
    HIP_TRY_CH(hipMalloc(&d_pen, penalty.size() * 4));
    HIP_TRY_CH(hipMemcpy(d_a, anchors, n * sizeof(mgx_chain_anchor), hipMemcpyHostToDevice));
    HIP_TRY_CH(hipMemcpy(d_lb, list_begin, (n_lists + 1) * 8, hipMemcpyHostToDevice));
    HIP_TRY_CH(hipMemcpy(d_qs, query_size, n_lists * 4, hipMemcpyHostToDevice));
    HIP_TRY_CH(hipMemcpy(d_pen, penalty.data(), penalty.size() * 4, hipMemcpyHostToDevice));
    // std::sort(dp_table, greater<TableElem>): (label, coordinate, seed_clipping, seed_end) descending within every list — four
    // stable segmented radix passes from the least significant field up (anchors equal in all four: input order, where the
    // reference's introsort leaves it open)
    const uint32_t tb = 256, blocks = (uint32_t)((n + tb - 1) / tb);
    k_chain_iota<<<blocks, tb>>>(d_perm, n);
    size_t tmp_bytes = 0, need = 0;
    HIP_TRY_CH(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(nullptr, need, d_k32, d_k32o, d_perm, d_perm2, (int)n, (int)n_lists, d_lb, d_lb + 1));
    tmp_bytes = need;
    HIP_TRY_CH(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(nullptr, need, d_k64, d_k64o, d_perm, d_perm2, (int)n, (int)n_lists, d_lb, d_lb + 1));
    tmp_bytes = std::max(tmp_bytes, need);
    HIP_TRY_CH(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)));
    for (int field = 0; field < 2; ++field) {                               // seed_end, then seed_clipping
        k_chain_keys32<<<blocks, tb>>>(d_a, d_perm, n, field, d_k32);
        size_t tb2 = tmp_bytes;
        HIP_TRY_CH(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(d_tmp, tb2, d_k32, d_k32o, d_perm, d_perm2, (int)n, (int)n_lists, d_lb, d_lb + 1));
        std::swap(d_perm, d_perm2);
    }
    for (int field = 0; field < 2; ++field) {                               // coordinate, then label
        k_chain_keys64<<<blocks, tb>>>(d_a, d_perm, n, field, d_k64);
        size_t tb2 = tmp_bytes;
        HIP_TRY_CH(hipcub::DeviceSegmentedRadixSort::SortPairsDescending(d_tmp, tb2, d_k64, d_k64o, d_perm, d_perm2, (int)n, (int)n_lists, d_lb, d_lb + 1));
        std::swap(d_perm, d_perm2);
    }
    k_chain_permute<<<blocks, tb>>>(d_a, d_perm, n, d_s);
    k_chain_dp<<<(uint32_t)n_lists, 64>>>(d_s, d_lb, d_qs, n_lists, d_pen, (uint32_t)penalty.size(), d_back);
    HIP_TRY_CH(hipGetLastError());
    HIP_TRY_CH(hipMemcpy(sorted_out, d_s, n * sizeof(mgx_chain_anchor), hipMemcpyDeviceToHost));
    HIP_TRY_CH(hipMemcpy(backtrace_out, d_back, n * 4, hipMemcpyDeviceToHost));
    cleanup();
    return MGX_OK;
}

} // extern "C"
