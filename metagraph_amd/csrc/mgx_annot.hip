// mgx_annot.hip — the label matrix of label-aware alignment on the device (BASELINE config 3).
//
// What the reference does on this path: AnnotationBuffer::fetch_queued_annotations (annotation_buffer.cpp:34-193) turns the
// queued nodes of a whole batch into rows and calls BinaryMatrix::get_rows ONCE (:182); with a ColumnCompressed annotation
// that is ColumnMajor::get_rows (annotation/binary_matrix/column_sparse/column_major.cpp:27-44): for every one of the ~1000
// columns, a bit test at every requested row — 1000 random accesses per row, the hot spot of config 3.
//
// On the device the matrix is turned ROW-major once, at mgx_annotation_create: a row query then costs one 8-byte access
// in the common case.  Layout (HBM):
//     head[row]   u64   count:16 | first label:24 | offset:24+ ... see pack(): rows with <= 1 label are answered from this
//                       word alone (in config 3 a label is a contiguous genome segment: almost every node has exactly one);
//     more[]      u32   for rows with several labels: their labels in ascending order at head's offset.
// Built from the caller's column bit vectors (one per label, bit r = row r, 64 rows per word — the host form of
// ColumnMajor) by three passes of streaming kernels: count per row, exclusive scan (hipcub), fill in label order, which
// makes every row's list ascending (the order get_rows' callers sort into, annotation_buffer.cpp:185).
//
// get_rows(rows[n]) -> CSR (begin[n + 1], labels[]): count kernel, scan, gather kernel — dependent random accesses only to
// head[] (8 B per row) and, for multi-label rows, one run of more[].
#include <hip/hip_runtime.h>
#include <atomic>
#include <hipcub/hipcub.hpp>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mgx.h"

extern "C" void mgx_set_last_error(const char *msg);     // mgx.hip

namespace {
int afail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    mgx_set_last_error(buf);
    return code;
}
#define HIP_TRY_A(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) return afail(MGX_ERR_NO_DEVICE, "%s: %s", #e, hipGetErrorString(r_)); } while (0)

// head word: bits 0-15 label count (saturating at 0xFFFF = "see more[] length word"), bits 16-63 payload: the single label
// (count == 1) or the offset of the row's list in more[] (count >= 2)
__host__ __device__ inline uint64_t head_pack(uint32_t count, uint64_t payload) { return (uint64_t)(count > 0xFFFE ? 0xFFFF : count) | (payload << 16); }
__host__ __device__ inline uint32_t head_count16(uint64_t h) { return (uint32_t)(h & 0xFFFF); }
__host__ __device__ inline uint64_t head_payload(uint64_t h) { return h >> 16; }

// one thread per 64-row word of ONE column: add its set bits to the rows' counts (columns are processed one launch after
// the other, so two threads never touch the same row)
__global__ void k_annot_count(const uint64_t *col_bits, uint64_t n_words, uint64_t n_rows, uint32_t *count) {
    const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    uint64_t bits = col_bits[wi];
    while (bits) {
        const int b = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        const uint64_t row = wi * 64 + (uint64_t)b;
        if (row < n_rows) ++count[row];
    }
}
// second sweep over the columns, in label order: rows with one label keep it in the head word, the others append to their
// list (fill[] = labels written so far)
__global__ void k_annot_fill(const uint64_t *col_bits, uint64_t n_words, uint64_t n_rows, uint32_t label, const uint32_t *count,
                             const uint64_t *offset, uint32_t *fill, uint64_t *head, uint32_t *more) {
    const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    uint64_t bits = col_bits[wi];
    while (bits) {
        const int b = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        const uint64_t row = wi * 64 + (uint64_t)b;
        if (row >= n_rows) continue;
        const uint32_t c = count[row];
        if (c == 1) head[row] = head_pack(1, label);
        else { more[offset[row] + fill[row]] = label; ++fill[row]; }
    }
}
// rows with several labels: count | offset; rows without: 0
__global__ void k_annot_heads(uint64_t n_rows, const uint32_t *count, const uint64_t *offset, uint64_t *head) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t c = count[r];
    if (c == 0) head[r] = 0;
    else if (c >= 2) head[r] = head_pack(c, offset[r]);
}
// list lengths of multi-label rows only (single labels live in the head word)
__global__ void k_annot_multi(uint64_t n_rows, const uint32_t *count, uint64_t *len) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows) len[r] = count[r] >= 2 ? count[r] : 0;
}

// get_rows, pass 1: labels per requested row (rows >= n_rows have none, like a row past the matrix in the reference's debug
// builds would assert; here it is an empty row)
__global__ void k_rows_count(const uint64_t *head, const uint32_t *count, uint64_t n_rows, const uint64_t *rows, uint64_t n, uint64_t *out_cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = rows[i];
    uint64_t c = 0;
    if (r < n_rows) {
        const uint64_t h = head[r];
        c = head_count16(h);
        if (c == 0xFFFF) c = count[r];
    }
    out_cnt[i] = c;
}
// pass 2: the labels, at the scanned offsets
__global__ void k_rows_gather(const uint64_t *head, const uint32_t *more, uint64_t n_rows, const uint64_t *rows, uint64_t n,
                              const uint64_t *out_begin, uint32_t *out_labels, uint64_t cap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = rows[i];
    if (r >= n_rows) return;
    const uint64_t b = out_begin[i], e = out_begin[i + 1];
    if (e == b || e > cap) return;
    const uint64_t h = head[r];
    if (e - b == 1) { out_labels[b] = (uint32_t)head_payload(h); return; }
    const uint32_t *src = more + head_payload(h);
    for (uint64_t x = 0; x < e - b; ++x) out_labels[b + x] = src[x];
}
// ---- construction from the columns' set rows (mgx_annotation_create_sparse) ----
// pair p (the p-th set bit of the matrix in column order) -> key = row << 24 | label; its label by bisection of col_begin
__global__ void k_pairs_keys(const uint64_t *col_begin, uint32_t n_labels, const uint64_t *rows, uint64_t n_pairs, uint64_t n_rows,
                             uint64_t *keys, uint32_t *bad) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    // (a row outside the matrix would sort — the radix sort covers the bits of n_rows only — into a valid row's run)
    if (rows[p] >= n_rows) { atomicOr(bad, 1u); keys[p] = 0; return; }
    uint32_t lo = 0, hi = n_labels;                      // the last label with col_begin[label] <= p
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (col_begin[mid] <= p) lo = mid; else hi = mid; }
    keys[p] = (rows[p] << 24) | lo;
}
// sorted keys: the first pair of every row counts the row's run; rows with one label are complete after this
__global__ void k_pairs_count(const uint64_t *keys, uint64_t n_pairs, uint64_t n_rows, uint32_t *count, uint64_t *head) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const uint64_t row = keys[p] >> 24;
    if (row >= n_rows || (p && (keys[p - 1] >> 24) == row)) return;
    uint64_t e = p + 1;
    while (e < n_pairs && (keys[e] >> 24) == row) ++e;
    count[row] = (uint32_t)(e - p);
    if (e - p == 1) head[row] = head_pack(1, keys[p] & 0xFFFFFF);
}
__global__ void k_pairs_fill(const uint64_t *keys, uint64_t n_pairs, uint64_t n_rows, const uint32_t *count, const uint64_t *offset, uint32_t *more) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const uint64_t row = keys[p] >> 24;
    if (row >= n_rows || (p && (keys[p - 1] >> 24) == row)) return;
    const uint32_t c = count[row];
    if (c < 2) return;
    for (uint32_t x = 0; x < c; ++x) more[offset[row] + x] = (uint32_t)(keys[p + x] & 0xFFFFFF);      // ascending: the keys are sorted
}
} // namespace

// get_row_tuples: per requested row and label, the number of its coordinates, then the coordinates themselves
__global__ void k_tuple_count(const uint64_t *row_entry, const uint64_t *coord_off, uint64_t n_rows, const uint64_t *rows, uint64_t n,
                              const uint64_t *out_begin, uint64_t *out_cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = rows[i];
    const uint64_t nl = out_begin[i + 1] - out_begin[i];
    for (uint64_t t = 0; t < nl; ++t) {
        uint64_t c = 0;
        if (r < n_rows) { const uint64_t e = row_entry[r] + t; c = coord_off[e + 1] - coord_off[e]; }
        out_cnt[out_begin[i] + t] = c;
    }
}
__global__ void k_tuple_gather(const uint64_t *row_entry, const uint64_t *coord_off, const int64_t *coords, uint64_t n_rows,
                               const uint64_t *rows, uint64_t n, const uint64_t *out_begin, const uint64_t *out_coord_begin, int64_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = rows[i];
    if (r >= n_rows) return;
    const uint64_t nl = out_begin[i + 1] - out_begin[i];
    for (uint64_t t = 0; t < nl; ++t) {
        const uint64_t e = row_entry[r] + t, o = out_coord_begin[out_begin[i] + t];
        for (uint64_t x = coord_off[e]; x < coord_off[e + 1]; ++x) out[o + (x - coord_off[e])] = coords[x];
    }
}

// every annotation object gets an id no other one of this process ever had: what is cached about an annotation elsewhere (the
// per-graph "no dummy node's row holds a label" flag, mgx.hip) is keyed by it — pointers come back from malloc / hipMalloc after
// a destroy + create and would hand a new annotation the old one's entry
static std::atomic<uint64_t> g_annotation_uid{0};

struct mgx_annotation {
    const uint64_t uid = ++g_annotation_uid;
    int device = 0;
    uint64_t n_rows = 0;
    uint32_t n_labels = 0;
    uint64_t n_more = 0;          // entries of more[]
    uint64_t *head = nullptr;
    uint32_t *count = nullptr;    // exact label count per row (rows with >= 0xFFFF labels need it; 4 B per row)
    uint32_t *more = nullptr;
    uint64_t bytes = 0;
    // k-mer coordinates (annot::matrix::MultiIntMatrix behind a ColumnCoordAnnotator; mgx_annotation_set_coordinates), row-major
    // like the labels: the labels of row r are entries row_entry[r] .. row_entry[r + 1) in the order of its label list, and
    // entry e carries the ascending coordinates coords[coord_off[e] .. coord_off[e + 1])
    uint64_t *row_entry = nullptr, *coord_off = nullptr;
    int64_t *coords = nullptr;
    uint64_t n_entries = 0, n_coords = 0;
    // scratch of get_rows, grown on demand; get_rows calls on one handle are serialised (workers of one process share the
    // annotation: mgx_align --devices, cli/align.cpp's thread pool)
    std::mutex rows_mutex;
    uint64_t *d_rows = nullptr, *d_begin = nullptr;
    uint32_t *d_labels = nullptr;
    void *d_tmp = nullptr;
    uint64_t cap_rows = 0, cap_labels = 0, cap_tmp = 0;
};

extern "C" {

/* LabeledAligner<>(graph, config, annotator) takes the annotator (aligner_labeled.hpp:125-127); its binary matrix arrives
 * here as column bit vectors: columns[j] points to ceil(n_rows / 64) words, bit r of the vector = row r has label j
 * (row = AnnotatedDBG::graph_to_anno_index(node) = node - 1, annotated_dbg.hpp:50-52). */
int mgx_annotation_create(uint64_t n_rows, uint32_t n_labels, const uint64_t *const *columns, int device, mgx_annotation **out) {
    if (!out || (!columns && n_labels) || n_labels >= (1u << 24)) return afail(MGX_ERR_INVALID, "mgx_annotation_create: bad arguments");
    for (uint32_t j = 0; j < n_labels && n_rows; ++j) if (!columns[j]) return afail(MGX_ERR_INVALID, "mgx_annotation_create: column %u is null", j);
    if (mgx_device_count() <= device) return afail(MGX_ERR_NO_DEVICE, "HIP device %d not available", device);
    HIP_TRY_A(hipSetDevice(device));
    auto *A = new mgx_annotation();
    A->device = device; A->n_rows = n_rows; A->n_labels = n_labels;
    const uint64_t n_words = (n_rows + 63) / 64;
    uint64_t *d_col = nullptr, *d_off = nullptr, *d_len = nullptr;
    uint32_t *d_fill = nullptr;
    void *d_tmp = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_col); (void)hipFree(d_off); (void)hipFree(d_len); (void)hipFree(d_fill); (void)hipFree(d_tmp); };
    auto bail = [&](int rc) { cleanup(); (void)hipFree(A->head); (void)hipFree(A->count); (void)hipFree(A->more); delete A; return rc; };
#define HIP_TRY_B(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) return bail(afail(MGX_ERR_NO_DEVICE, "%s: %s", #e, hipGetErrorString(r_))); } while (0)
    HIP_TRY_B(hipMalloc(&A->head, std::max<uint64_t>(1, n_rows) * 8));
    HIP_TRY_B(hipMalloc(&A->count, std::max<uint64_t>(1, n_rows) * 4));
    HIP_TRY_B(hipMemset(A->count, 0, std::max<uint64_t>(1, n_rows) * 4));
    HIP_TRY_B(hipMalloc(&d_col, std::max<uint64_t>(1, n_words) * 8));
    const uint32_t tb = 256;
    const uint32_t wblocks = (uint32_t)((n_words + tb - 1) / tb), rblocks = (uint32_t)((n_rows + tb - 1) / tb);
    for (uint32_t j = 0; j < n_labels && n_words; ++j) {
        HIP_TRY_B(hipMemcpy(d_col, columns[j], n_words * 8, hipMemcpyHostToDevice));
        k_annot_count<<<wblocks, tb>>>(d_col, n_words, n_rows, A->count);
    }
    HIP_TRY_B(hipGetLastError());
    // offsets of the multi-label rows' lists
    HIP_TRY_B(hipMalloc(&d_len, (n_rows + 1) * 8));
    HIP_TRY_B(hipMalloc(&d_off, (n_rows + 1) * 8));
    HIP_TRY_B(hipMemset(d_len, 0, (n_rows + 1) * 8));
    if (n_rows) k_annot_multi<<<rblocks, tb>>>(n_rows, A->count, d_len);
    size_t tmp_bytes = 0;
    HIP_TRY_B(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_len, d_off, n_rows + 1));
    HIP_TRY_B(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)));
    HIP_TRY_B(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_len, d_off, n_rows + 1));
    HIP_TRY_B(hipMemcpy(&A->n_more, d_off + n_rows, 8, hipMemcpyDeviceToHost));
    HIP_TRY_B(hipMalloc(&A->more, std::max<uint64_t>(1, A->n_more) * 4));
    HIP_TRY_B(hipMalloc(&d_fill, std::max<uint64_t>(1, n_rows) * 4));
    HIP_TRY_B(hipMemset(d_fill, 0, std::max<uint64_t>(1, n_rows) * 4));
    if (n_rows) k_annot_heads<<<rblocks, tb>>>(n_rows, A->count, d_off, A->head);
    for (uint32_t j = 0; j < n_labels && n_words; ++j) {
        HIP_TRY_B(hipMemcpy(d_col, columns[j], n_words * 8, hipMemcpyHostToDevice));
        k_annot_fill<<<wblocks, tb>>>(d_col, n_words, n_rows, j, A->count, d_off, d_fill, A->head, A->more);
    }
    HIP_TRY_B(hipGetLastError());
    HIP_TRY_B(hipDeviceSynchronize());
#undef HIP_TRY_B
    cleanup();
    A->bytes = n_rows * 12 + A->n_more * 4;
    *out = A;
    return MGX_OK;
}

/* The same matrix from the columns' set rows — what a ColumnCompressed annotation stores (one sd_vector of row indices per
 * label; annotation/representation/column_compressed): rows[col_begin[j] .. col_begin[j + 1]) = the rows that carry label j
 * (any order, no duplicates).  on_device != 0: `rows` is a device pointer (col_begin is always a host array of n_labels + 1
 * entries).  Built by one sort of (row, label) keys on the device. */
int mgx_annotation_create_sparse(uint64_t n_rows, uint32_t n_labels, const uint64_t *col_begin, const uint64_t *rows, int on_device,
                                 int device, mgx_annotation **out) {
    if (!out || !col_begin || n_labels >= (1u << 24) || n_rows >= (1ull << 40)) return afail(MGX_ERR_INVALID, "mgx_annotation_create_sparse: bad arguments");
    const uint64_t n_pairs = col_begin[n_labels];
    if (!rows && n_pairs) return afail(MGX_ERR_INVALID, "mgx_annotation_create_sparse: bad arguments");
    for (uint32_t j = 0; j < n_labels; ++j)
        if (col_begin[j] > col_begin[j + 1]) return afail(MGX_ERR_INVALID, "mgx_annotation_create_sparse: col_begin must be ascending (label %u)", j);
    if (col_begin[0] != 0) return afail(MGX_ERR_INVALID, "mgx_annotation_create_sparse: col_begin[0] must be 0");
    if (mgx_device_count() <= device) return afail(MGX_ERR_NO_DEVICE, "HIP device %d not available", device);
    HIP_TRY_A(hipSetDevice(device));
    auto *A = new mgx_annotation();
    A->device = device; A->n_rows = n_rows; A->n_labels = n_labels;
    uint64_t *d_cb = nullptr, *d_rows = nullptr, *d_keys = nullptr, *d_sorted = nullptr, *d_off = nullptr, *d_len = nullptr;
    void *d_tmp = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_cb); (void)hipFree(d_rows); (void)hipFree(d_keys); (void)hipFree(d_sorted); (void)hipFree(d_off); (void)hipFree(d_len); (void)hipFree(d_tmp); };
    auto bail = [&](int rc) { cleanup(); (void)hipFree(A->head); (void)hipFree(A->count); (void)hipFree(A->more); delete A; return rc; };
#define HIP_TRY_B(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) return bail(afail(MGX_ERR_NO_DEVICE, "%s: %s", #e, hipGetErrorString(r_))); } while (0)
    HIP_TRY_B(hipMalloc(&A->head, std::max<uint64_t>(1, n_rows) * 8));
    HIP_TRY_B(hipMalloc(&A->count, std::max<uint64_t>(1, n_rows) * 4));
    HIP_TRY_B(hipMemset(A->count, 0, std::max<uint64_t>(1, n_rows) * 4));
    HIP_TRY_B(hipMemset(A->head, 0, std::max<uint64_t>(1, n_rows) * 8));
    const uint32_t tb = 256;
    const uint32_t pblocks = (uint32_t)((n_pairs + tb - 1) / tb), rblocks = (uint32_t)((n_rows + tb - 1) / tb);
    HIP_TRY_B(hipMalloc(&d_len, (n_rows + 1) * 8));
    HIP_TRY_B(hipMalloc(&d_off, (n_rows + 1) * 8));
    HIP_TRY_B(hipMemset(d_len, 0, (n_rows + 1) * 8));
    if (n_pairs) {
        HIP_TRY_B(hipMalloc(&d_cb, ((uint64_t)n_labels + 1) * 8));
        HIP_TRY_B(hipMemcpy(d_cb, col_begin, ((uint64_t)n_labels + 1) * 8, hipMemcpyHostToDevice));
        const uint64_t *src = rows;
        if (!on_device) {
            HIP_TRY_B(hipMalloc(&d_rows, n_pairs * 8));
            HIP_TRY_B(hipMemcpy(d_rows, rows, n_pairs * 8, hipMemcpyHostToDevice));
            src = d_rows;
        }
        HIP_TRY_B(hipMalloc(&d_keys, n_pairs * 8));
        HIP_TRY_B(hipMalloc(&d_sorted, n_pairs * 8));
        uint32_t *d_bad = nullptr, bad = 0;
        HIP_TRY_B(hipMalloc(&d_bad, 4));
        HIP_TRY_B(hipMemset(d_bad, 0, 4));
        k_pairs_keys<<<pblocks, tb>>>(d_cb, n_labels, src, n_pairs, n_rows, d_keys, d_bad);
        const hipError_t rb = hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
        (void)hipFree(d_bad);
        if (rb != hipSuccess) return bail(afail(MGX_ERR_NO_DEVICE, "hipMemcpy: %s", hipGetErrorString(rb)));
        if (bad) return bail(afail(MGX_ERR_INVALID, "mgx_annotation_create_sparse: a row index is outside the matrix (%llu rows)", (unsigned long long)n_rows));
        int end_bit = 24;
        while (end_bit < 64 && (n_rows >> (end_bit - 24))) ++end_bit;
        size_t sort_bytes = 0;
        HIP_TRY_B(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, d_keys, d_sorted, n_pairs, 0, end_bit));
        HIP_TRY_B(hipMalloc(&d_tmp, std::max<size_t>(sort_bytes, 16)));
        HIP_TRY_B(hipcub::DeviceRadixSort::SortKeys(d_tmp, sort_bytes, d_keys, d_sorted, n_pairs, 0, end_bit));
        (void)hipFree(d_tmp); d_tmp = nullptr;
        k_pairs_count<<<pblocks, tb>>>(d_sorted, n_pairs, n_rows, A->count, A->head);
    }
    if (n_rows) k_annot_multi<<<rblocks, tb>>>(n_rows, A->count, d_len);
    size_t tmp_bytes = 0;
    HIP_TRY_B(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_len, d_off, n_rows + 1));
    HIP_TRY_B(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)));
    HIP_TRY_B(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_len, d_off, n_rows + 1));
    HIP_TRY_B(hipMemcpy(&A->n_more, d_off + n_rows, 8, hipMemcpyDeviceToHost));
    HIP_TRY_B(hipMalloc(&A->more, std::max<uint64_t>(1, A->n_more) * 4));
    if (n_rows) k_annot_heads<<<rblocks, tb>>>(n_rows, A->count, d_off, A->head);
    if (n_pairs) k_pairs_fill<<<pblocks, tb>>>(d_sorted, n_pairs, n_rows, A->count, d_off, A->more);
    HIP_TRY_B(hipGetLastError());
    HIP_TRY_B(hipDeviceSynchronize());
#undef HIP_TRY_B
    cleanup();
    A->bytes = n_rows * 12 + A->n_more * 4;
    *out = A;
    return MGX_OK;
}

void mgx_annotation_destroy(mgx_annotation *a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    (void)hipFree(a->head); (void)hipFree(a->count); (void)hipFree(a->more);
    (void)hipFree(a->d_rows); (void)hipFree(a->d_begin); (void)hipFree(a->d_labels); (void)hipFree(a->d_tmp);
    (void)hipFree(a->row_entry); (void)hipFree(a->coord_off); (void)hipFree(a->coords);
    delete a;
}
uint64_t mgx_annotation_device_bytes(const mgx_annotation *a) { return a ? a->bytes : 0; }
uint64_t mgx_annotation_num_rows(const mgx_annotation *a) { return a ? a->n_rows : 0; }
uint32_t mgx_annotation_num_labels(const mgx_annotation *a) { return a ? a->n_labels : 0; }
// the matrix as the label-aware extension kernel reads it (mgx.hip fills AlignParams::anno_* from this; library-internal)
extern "C" void mgx_annotation_device_view(const mgx_annotation *a, int *device, uint64_t *n_rows, const uint64_t **head,
                                           const uint32_t **count, const uint32_t **more) {
    *device = a->device; *n_rows = a->n_rows; *head = a->head; *count = a->count; *more = a->more;
}
extern "C" uint64_t mgx_annotation_uid(const mgx_annotation *a) { return a ? a->uid : 0; }

/* BinaryMatrix::get_rows(rows) (annotation/binary_matrix/base/binary_matrix.hpp; ColumnMajor: column_major.cpp:27-44) for a
 * whole batch of rows — the ONE call of AnnotationBuffer::fetch_queued_annotations (annotation_buffer.cpp:182) — answered
 * as CSR: labels of rows[i] = out_labels[out_begin[i] .. out_begin[i + 1]), ascending.  `cap` = entries out_labels can hold;
 * if the batch has more, nothing is written to out_labels, *n_labels_out tells how many and the call returns MGX_ERR_CAPACITY.
 * rows_on_device / out_on_device: the pointers are device pointers (rows already in HBM, result left in HBM). */
int mgx_annotation_get_rows(mgx_annotation *a, const uint64_t *rows, uint64_t n, int rows_on_device,
                            uint64_t *out_begin, uint32_t *out_labels, uint64_t cap, int out_on_device, uint64_t *n_labels_out) {
    if (!a || (!rows && n) || !out_begin || (!out_labels && cap)) return afail(MGX_ERR_INVALID, "mgx_annotation_get_rows: bad arguments");
    if (mgx_device_count() <= a->device) return afail(MGX_ERR_NO_DEVICE, "no HIP device");
    std::lock_guard<std::mutex> lock(a->rows_mutex);
    HIP_TRY_A(hipSetDevice(a->device));
    const uint32_t tb = 256;
    const uint32_t blocks = (uint32_t)((n + tb - 1) / tb);
    if (n + 1 > a->cap_rows) {
        (void)hipFree(a->d_rows); (void)hipFree(a->d_begin); a->d_rows = a->d_begin = nullptr; a->cap_rows = 0;
        const uint64_t want = n + 1 + n / 8;
        HIP_TRY_A(hipMalloc(&a->d_rows, want * 8));
        HIP_TRY_A(hipMalloc(&a->d_begin, (want + 1) * 8));
        a->cap_rows = want;
    }
    const uint64_t *d_rows = rows;
    if (!rows_on_device && n) { HIP_TRY_A(hipMemcpy(a->d_rows, rows, n * 8, hipMemcpyHostToDevice)); d_rows = a->d_rows; }
    uint64_t *d_begin = out_on_device ? out_begin : a->d_begin;
    // counts -> exclusive scan in place (n + 1 entries, the last one 0 before the scan)
    HIP_TRY_A(hipMemset(d_begin + n, 0, 8));
    if (n) k_rows_count<<<blocks, tb>>>(a->head, a->count, a->n_rows, d_rows, n, d_begin);
    size_t tmp_bytes = 0;
    HIP_TRY_A(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_begin, d_begin, n + 1));
    if (tmp_bytes > a->cap_tmp) { (void)hipFree(a->d_tmp); a->d_tmp = nullptr; HIP_TRY_A(hipMalloc(&a->d_tmp, tmp_bytes + 256)); a->cap_tmp = tmp_bytes + 256; }
    HIP_TRY_A(hipcub::DeviceScan::ExclusiveSum(a->d_tmp, tmp_bytes, d_begin, d_begin, n + 1));
    uint64_t total = 0;
    HIP_TRY_A(hipMemcpy(&total, d_begin + n, 8, hipMemcpyDeviceToHost));
    if (n_labels_out) *n_labels_out = total;
    if (!out_on_device) HIP_TRY_A(hipMemcpy(out_begin, d_begin, (n + 1) * 8, hipMemcpyDeviceToHost));
    if (total > cap) return afail(MGX_ERR_CAPACITY, "mgx_annotation_get_rows: %llu labels, room for %llu", (unsigned long long)total, (unsigned long long)cap);
    uint32_t *d_labels = out_labels;
    if (!out_on_device) {
        if (total > a->cap_labels) {
            (void)hipFree(a->d_labels); a->d_labels = nullptr; a->cap_labels = 0;
            HIP_TRY_A(hipMalloc(&a->d_labels, (total + total / 8 + 16) * 4));
            a->cap_labels = total + total / 8 + 16;
        }
        d_labels = a->d_labels;
    }
    if (n && total) k_rows_gather<<<blocks, tb>>>(a->head, a->more, a->n_rows, d_rows, n, d_begin, d_labels, cap);
    HIP_TRY_A(hipGetLastError());
    if (!out_on_device && total) HIP_TRY_A(hipMemcpy(out_labels, d_labels, total * 4, hipMemcpyDeviceToHost));
    else HIP_TRY_A(hipDeviceSynchronize());
    return MGX_OK;
}


/* k-mer coordinates of an annotation built from the columns' set rows (mgx_annotation_create_sparse with the same n_labels /
 * col_begin / rows): the ascending coordinates of pair x — label j = the column whose range holds x, row rows[x] — are
 * coords[coord_begin[x] .. coord_begin[x + 1]).  What annot::ColumnCoordAnnotator holds next to the binary matrix
 * (annot::matrix::MultiIntMatrix::get_row_tuples); host arrays.  The row-major form the device keeps is built on the host: like
 * everything about annotation CONSTRUCTION, outside the aligner's path. */
int mgx_annotation_set_coordinates(mgx_annotation *a, uint32_t n_labels, const uint64_t *col_begin, const uint64_t *rows,
                                   const uint64_t *coord_begin, const int64_t *coords) {
    if (!a || !col_begin || n_labels != a->n_labels) return afail(MGX_ERR_INVALID, "mgx_annotation_set_coordinates: bad arguments");
    const uint64_t n_pairs = col_begin[n_labels];
    if (n_pairs && (!rows || !coord_begin)) return afail(MGX_ERR_INVALID, "mgx_annotation_set_coordinates: bad arguments");
    if (mgx_device_count() <= a->device) return afail(MGX_ERR_NO_DEVICE, "no HIP device");
    HIP_TRY_A(hipSetDevice(a->device));
    // (row, label) -> pair index, rows ascending and labels ascending within a row: the order of the device's label lists
    std::vector<std::pair<std::pair<uint64_t, uint32_t>, uint64_t>> order;
    order.reserve(n_pairs);
    for (uint32_t j = 0; j < n_labels; ++j)
        for (uint64_t x = col_begin[j]; x < col_begin[j + 1]; ++x) {
            if (rows[x] >= a->n_rows) return afail(MGX_ERR_INVALID, "mgx_annotation_set_coordinates: a row index is outside the matrix");
            order.push_back({ { rows[x], j }, x });
        }
    std::sort(order.begin(), order.end());
    std::vector<uint64_t> row_entry(a->n_rows + 1, 0), coord_off(n_pairs + 1, 0);
    std::vector<int64_t> flat;
    flat.reserve(n_pairs ? coord_begin[n_pairs] : 0);
    for (uint64_t e = 0; e < n_pairs; ++e) {
        const uint64_t x = order[e].second;
        ++row_entry[order[e].first.first + 1];
        for (uint64_t c = coord_begin[x]; c < coord_begin[x + 1]; ++c) {
            if (c > coord_begin[x] && coords[c] < coords[c - 1]) return afail(MGX_ERR_INVALID, "mgx_annotation_set_coordinates: coordinates must be ascending");
            flat.push_back(coords[c]);
        }
        coord_off[e + 1] = flat.size();
    }
    for (uint64_t r = 0; r < a->n_rows; ++r) row_entry[r + 1] += row_entry[r];
    (void)hipFree(a->row_entry); (void)hipFree(a->coord_off); (void)hipFree(a->coords);
    a->row_entry = a->coord_off = nullptr; a->coords = nullptr;
    HIP_TRY_A(hipMalloc(&a->row_entry, row_entry.size() * 8));
    HIP_TRY_A(hipMalloc(&a->coord_off, coord_off.size() * 8));
    HIP_TRY_A(hipMalloc(&a->coords, std::max<size_t>(1, flat.size()) * 8));
    HIP_TRY_A(hipMemcpy(a->row_entry, row_entry.data(), row_entry.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY_A(hipMemcpy(a->coord_off, coord_off.data(), coord_off.size() * 8, hipMemcpyHostToDevice));
    if (flat.size()) HIP_TRY_A(hipMemcpy(a->coords, flat.data(), flat.size() * 8, hipMemcpyHostToDevice));
    a->n_entries = n_pairs; a->n_coords = flat.size();
    a->bytes += row_entry.size() * 8 + coord_off.size() * 8 + flat.size() * 8;
    return MGX_OK;
}
int mgx_annotation_has_coordinates(const mgx_annotation *a) { return a && a->row_entry != nullptr; }

/* MultiIntMatrix::get_row_tuples(rows) for a whole batch — what AnnotationBuffer::fetch_queued_annotations calls instead of
 * get_rows when the annotation carries coordinates (annotation_buffer.cpp:166-181): the labels as mgx_annotation_get_rows
 * returns them, and for label entry e (index into out_labels) the coordinates out_coords[out_coord_begin[e] ..
 * out_coord_begin[e + 1]).  Host arrays; capacities as in get_rows (MGX_ERR_CAPACITY with the counts reported). */
int mgx_annotation_get_row_tuples(mgx_annotation *a, const uint64_t *rows, uint64_t n, uint64_t *out_begin, uint32_t *out_labels,
                                  uint64_t label_cap, uint64_t *out_coord_begin, int64_t *out_coords, uint64_t coord_cap,
                                  uint64_t *n_labels_out, uint64_t *n_coords_out) {
    if (!a || !a->row_entry) return afail(MGX_ERR_INVALID, "mgx_annotation_get_row_tuples: the annotation has no coordinates");
    if (!out_coord_begin || (!out_coords && coord_cap)) return afail(MGX_ERR_INVALID, "mgx_annotation_get_row_tuples: bad arguments");
    uint64_t total = 0;
    if (int rc = mgx_annotation_get_rows(a, rows, n, 0, out_begin, out_labels, label_cap, 0, &total)) { if (n_labels_out) *n_labels_out = total; return rc; }
    if (n_labels_out) *n_labels_out = total;
    std::lock_guard<std::mutex> lock(a->rows_mutex);
    HIP_TRY_A(hipSetDevice(a->device));
    uint64_t *d_rows = nullptr, *d_begin = nullptr, *d_cb = nullptr;
    int64_t *d_out = nullptr;
    void *d_tmp = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_rows); (void)hipFree(d_begin); (void)hipFree(d_cb); (void)hipFree(d_out); (void)hipFree(d_tmp); };
#define HIP_TRY_C(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { cleanup(); return afail(MGX_ERR_NO_DEVICE, "%s: %s", #e, hipGetErrorString(r_)); } } while (0)
    HIP_TRY_C(hipMalloc(&d_rows, std::max<uint64_t>(1, n) * 8));
    HIP_TRY_C(hipMalloc(&d_begin, (n + 1) * 8));
    HIP_TRY_C(hipMalloc(&d_cb, (total + 1) * 8));
    if (n) HIP_TRY_C(hipMemcpy(d_rows, rows, n * 8, hipMemcpyHostToDevice));
    HIP_TRY_C(hipMemcpy(d_begin, out_begin, (n + 1) * 8, hipMemcpyHostToDevice));
    HIP_TRY_C(hipMemset(d_cb, 0, (total + 1) * 8));
    const uint32_t tb = 256, blocks = (uint32_t)((n + tb - 1) / tb);
    if (n) k_tuple_count<<<blocks, tb>>>(a->row_entry, a->coord_off, a->n_rows, d_rows, n, d_begin, d_cb);
    size_t tmp_bytes = 0;
    HIP_TRY_C(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cb, d_cb, total + 1));
    HIP_TRY_C(hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)));
    HIP_TRY_C(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_cb, d_cb, total + 1));
    uint64_t n_coords = 0;
    HIP_TRY_C(hipMemcpy(&n_coords, d_cb + total, 8, hipMemcpyDeviceToHost));
    if (n_coords_out) *n_coords_out = n_coords;
    HIP_TRY_C(hipMemcpy(out_coord_begin, d_cb, (total + 1) * 8, hipMemcpyDeviceToHost));
    if (n_coords > coord_cap) { cleanup(); return afail(MGX_ERR_CAPACITY, "mgx_annotation_get_row_tuples: %llu coordinates, room for %llu", (unsigned long long)n_coords, (unsigned long long)coord_cap); }
    HIP_TRY_C(hipMalloc(&d_out, std::max<uint64_t>(1, n_coords) * 8));
    if (n && n_coords) k_tuple_gather<<<blocks, tb>>>(a->row_entry, a->coord_off, a->coords, a->n_rows, d_rows, n, d_begin, d_cb, d_out);
    HIP_TRY_C(hipGetLastError());
    if (n_coords) HIP_TRY_C(hipMemcpy(out_coords, d_out, n_coords * 8, hipMemcpyDeviceToHost));
#undef HIP_TRY_C
    cleanup();
    return MGX_OK;
}

} // extern "C"
