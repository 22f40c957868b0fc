// mgx_gather.hip — the gather of complete alignments to one rank over RCCL, behind the C-ABI (round 6).
//
// north_star: "query batches shard one-read-set-per-GPU ... RCCL-over-xGMI used only to gather alignment results".  The reference's
// gather is its output loop (cli/align.cpp:469-473: every task prints its queries' lines under a mutex); with one aligner per
// device the results of a batch are two device buffers per rank — n_queries fixed-size records and a stream of 32-bit words
// (mgx_device_results) — and the gather moves both to the root, which decodes them (mgx_results_from_raw) and prints.
// metagraph_amd/gather.py is the same two-phase exchange over torch.distributed for the Python bench; this is the one a C++
// `metagraph align` host calls.  There is no collective anywhere else on the path (reads are sharded, the graph is replicated).
//
//   phase 1: all-gather of (n_queries, used stream words) — 16 bytes per rank — and a host read of the table (the transfers are
//            sized by it);
//   phase 2: one group of point-to-point transfers: every rank but the root sends its records and the used part of its stream,
//            the root posts the matching receives (xGMI is point-to-point: a send / receive pair per rank IS the collective's
//            natural shape there) and copies its own part device-to-device.
//   finish:  the root copies what it received to pinned host memory and hands out per-rank pointers.
//
// librccl.so is opened at the first call (dlopen): libmgx.so does not link it and loads on hosts without RCCL; a missing library
// is MGX_ERR_UNSUPPORTED at mgx_gather_create*, never a silent fallback.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mgx.h"

extern "C" void mgx_set_last_error(const char *msg);     // mgx.hip

namespace {

int gfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    mgx_set_last_error(buf);
    return code;
}

struct Rccl {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

Rccl &rccl_state() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char *n : names) { R.so = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (R.so) break; }
        if (!R.so) { const char *e = dlerror(); R.why = std::string("librccl.so could not be opened: ") + (e ? e : "not found"); return; }
        auto sym = [&](const char *s) -> void * { void *p = dlsym(R.so, s); if (!p && R.why.empty()) R.why = std::string("librccl.so has no symbol ") + s; return p; };
        R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
        R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
        R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
        R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
        R.Send = (decltype(R.Send))sym("ncclSend");
        R.Recv = (decltype(R.Recv))sym("ncclRecv");
        R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
        R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
        if (!R.why.empty()) { dlclose(R.so); R.so = nullptr; }
    });
    return R;
}
Rccl *rccl() { Rccl &R = rccl_state(); return R.so ? &R : nullptr; }
const char *rccl_why() { return rccl_state().why.c_str(); }

struct DevMem {
    void *p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        const hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};
struct HostMem {
    void *p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; bytes = 0;
        const hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};

}  // namespace

struct mgx_gather {
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int rank = 0, world = 1, root = 0, device = 0;
    hipStream_t stream = nullptr;            // the gather's own stream (non-blocking; ordered behind the aligner's by an event)
    hipEvent_t ready = nullptr;
    DevMem d_counts;                         // [world + 1][2] u64: slot `world` = this rank's (n_queries, used words)
    DevMem d_headers, d_stream;              // root: what the other ranks sent, rank after rank
    HostMem h_headers, h_stream;
    std::vector<uint64_t> counts;            // [world][2] after phase 1
    std::vector<uint64_t> h_off, s_off;      // root: byte offset of rank r's records / stream words in the receive buffers
    uint64_t header_bytes = 0;
    bool pending = false;
};

#define HIP_TRY_G(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) return gfail(MGX_ERR_NO_DEVICE, "%s: %s", #e, hipGetErrorString(r_)); } while (0)
#define NCCL_TRY_G(e) do { ncclResult_t r_ = (e); if (r_ != ncclSuccess) return gfail(MGX_ERR_NO_DEVICE, "%s: %s", #e, R->GetErrorString(r_)); } while (0)

static int gather_init(mgx_gather *g) {
    HIP_TRY_G(hipSetDevice(g->device));
    HIP_TRY_G(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    HIP_TRY_G(hipEventCreateWithFlags(&g->ready, hipEventDisableTiming));
    HIP_TRY_G(g->d_counts.ensure((size_t)(g->world + 1) * 16));
    g->counts.assign((size_t)g->world * 2, 0);
    return MGX_OK;
}

extern "C" int mgx_gather_unique_id(void *id_out, uint64_t id_bytes) {
    Rccl *R = rccl();
    if (!R) return gfail(MGX_ERR_UNSUPPORTED, "%s", rccl_why());
    if (!id_out || id_bytes < sizeof(ncclUniqueId)) return gfail(MGX_ERR_INVALID, "mgx_gather_unique_id: %llu bytes needed", (unsigned long long)sizeof(ncclUniqueId));
    ncclUniqueId id;
    NCCL_TRY_G(R->GetUniqueId(&id));
    memset(id_out, 0, (size_t)id_bytes);
    memcpy(id_out, &id, sizeof(id));
    return MGX_OK;
}

extern "C" int mgx_gather_create_rank(const void *unique_id, uint64_t id_bytes, int rank, int world, int root, int device, mgx_gather **out) {
    Rccl *R = rccl();
    if (!R) return gfail(MGX_ERR_UNSUPPORTED, "%s", rccl_why());
    if (!unique_id || id_bytes < sizeof(ncclUniqueId) || !out || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world)
        return gfail(MGX_ERR_INVALID, "mgx_gather_create_rank: bad argument");
    HIP_TRY_G(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    auto *g = new mgx_gather();
    g->rank = rank; g->world = world; g->root = root; g->device = device; g->own_comm = true;
    ncclResult_t r = R->CommInitRank(&g->comm, world, id, rank);
    if (r != ncclSuccess) { delete g; return gfail(MGX_ERR_NO_DEVICE, "ncclCommInitRank: %s", R->GetErrorString(r)); }
    if (int rc = gather_init(g)) { mgx_gather_destroy(g); return rc; }
    *out = g;
    return MGX_OK;
}

extern "C" int mgx_gather_create_comm(void *nccl_comm, int rank, int world, int root, int device, mgx_gather **out) {
    Rccl *R = rccl();
    if (!R) return gfail(MGX_ERR_UNSUPPORTED, "%s", rccl_why());
    if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return gfail(MGX_ERR_INVALID, "mgx_gather_create_comm: bad argument");
    auto *g = new mgx_gather();
    g->comm = (ncclComm_t)nccl_comm; g->rank = rank; g->world = world; g->root = root; g->device = device; g->own_comm = false;
    if (int rc = gather_init(g)) { mgx_gather_destroy(g); return rc; }
    *out = g;
    return MGX_OK;
}

extern "C" int mgx_gather_create_local(const int *devices, int n_devices, int root, mgx_gather **out) {
    Rccl *R = rccl();
    if (!R) return gfail(MGX_ERR_UNSUPPORTED, "%s", rccl_why());
    if (!devices || n_devices < 1 || !out || root < 0 || root >= n_devices) return gfail(MGX_ERR_INVALID, "mgx_gather_create_local: bad argument");
    std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
    NCCL_TRY_G(R->CommInitAll(comms.data(), n_devices, devices));
    for (int r = 0; r < n_devices; ++r) out[r] = nullptr;
    for (int r = 0; r < n_devices; ++r) {
        auto *g = new mgx_gather();
        g->comm = comms[(size_t)r]; g->rank = r; g->world = n_devices; g->root = root; g->device = devices[r]; g->own_comm = true;
        out[r] = g;
        if (int rc = gather_init(g)) {
            for (int q = 0; q < n_devices; ++q) { if (out[q]) mgx_gather_destroy(out[q]); else if (comms[(size_t)q]) (void)R->CommDestroy(comms[(size_t)q]); out[q] = nullptr; }
            return rc;
        }
    }
    return MGX_OK;
}

extern "C" void mgx_gather_destroy(mgx_gather *g) {
    if (!g) return;
    Rccl *R = rccl();
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->own_comm && g->comm && R) (void)R->CommDestroy(g->comm);
    g->d_counts.release(); g->d_headers.release(); g->d_stream.release();
    g->h_headers.release(); g->h_stream.release();
    if (g->ready) (void)hipEventDestroy(g->ready);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

// Collective: every rank calls it with its own aligner after mgx_align_batch_device.  Returns when this rank's transfers are
// enqueued; the aligner's device results must stay untouched until mgx_gather_finish has returned on this rank.
extern "C" int mgx_gather_start(mgx_gather *g, mgx_aligner *a) {
    Rccl *R = rccl();
    if (!R) return gfail(MGX_ERR_UNSUPPORTED, "%s", rccl_why());
    if (!g || !a) return gfail(MGX_ERR_INVALID, "mgx_gather_start: null argument");
    if (g->pending) return gfail(MGX_ERR_INVALID, "mgx_gather_start: the previous gather has not been finished");
    HIP_TRY_G(hipSetDevice(g->device));
    const void *hdr = nullptr, *str = nullptr;
    uint64_t hb = 0, nq = 0, used = 0;
    if (int rc = mgx_device_results(a, &hdr, &hb, &nq, &str, &used)) return rc;          // (waits for the batch's kernels)
    g->header_bytes = hb;
    // the gather's stream runs behind whatever the aligner's stream still holds
    hipStream_t as = (hipStream_t)mgx_aligner_get_stream(a);
    HIP_TRY_G(hipEventRecord(g->ready, as));
    HIP_TRY_G(hipStreamWaitEvent(g->stream, g->ready, 0));
    // ---- phase 1: the table of (n_queries, used words) ----
    uint64_t mine[2] = { nq, used };
    uint64_t *dc = static_cast<uint64_t *>(g->d_counts.p);
    HIP_TRY_G(hipMemcpyAsync(dc + 2 * (size_t)g->world, mine, 16, hipMemcpyHostToDevice, g->stream));
    NCCL_TRY_G(R->AllGather(dc + 2 * (size_t)g->world, dc, 2, ncclUint64, g->comm, g->stream));
    HIP_TRY_G(hipMemcpyAsync(g->counts.data(), dc, (size_t)g->world * 16, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY_G(hipStreamSynchronize(g->stream));
    // ---- phase 2: records and streams to the root ----
    if (g->rank == g->root) {
        g->h_off.assign((size_t)g->world + 1, 0); g->s_off.assign((size_t)g->world + 1, 0);
        for (int r = 0; r < g->world; ++r) {
            g->h_off[(size_t)r + 1] = g->h_off[(size_t)r] + g->counts[2 * (size_t)r] * hb;
            g->s_off[(size_t)r + 1] = g->s_off[(size_t)r] + g->counts[2 * (size_t)r + 1] * 4;
        }
        HIP_TRY_G(g->d_headers.ensure(std::max<size_t>(g->h_off[(size_t)g->world], 16)));
        HIP_TRY_G(g->d_stream.ensure(std::max<size_t>(g->s_off[(size_t)g->world], 16)));
        NCCL_TRY_G(R->GroupStart());
        for (int r = 0; r < g->world; ++r) {
            if (r == g->root) continue;
            const size_t hbytes = (size_t)(g->h_off[(size_t)r + 1] - g->h_off[(size_t)r]), sbytes = (size_t)(g->s_off[(size_t)r + 1] - g->s_off[(size_t)r]);
            if (hbytes) NCCL_TRY_G(R->Recv((char *)g->d_headers.p + g->h_off[(size_t)r], hbytes, ncclUint8, r, g->comm, g->stream));
            if (sbytes) NCCL_TRY_G(R->Recv((char *)g->d_stream.p + g->s_off[(size_t)r], sbytes, ncclUint8, r, g->comm, g->stream));
        }
        NCCL_TRY_G(R->GroupEnd());
        if (nq) HIP_TRY_G(hipMemcpyAsync((char *)g->d_headers.p + g->h_off[(size_t)g->root], hdr, (size_t)(nq * hb), hipMemcpyDeviceToDevice, g->stream));
        if (used) HIP_TRY_G(hipMemcpyAsync((char *)g->d_stream.p + g->s_off[(size_t)g->root], str, (size_t)(used * 4), hipMemcpyDeviceToDevice, g->stream));
    } else {
        NCCL_TRY_G(R->GroupStart());
        if (nq) NCCL_TRY_G(R->Send(hdr, (size_t)(nq * hb), ncclUint8, g->root, g->comm, g->stream));
        if (used) NCCL_TRY_G(R->Send(str, (size_t)(used * 4), ncclUint8, g->root, g->comm, g->stream));
        NCCL_TRY_G(R->GroupEnd());
    }
    g->pending = true;
    return MGX_OK;
}

// Waits for this rank's part of the gather.  On the root: n_queries[r], headers[r], streams[r], stream_words[r] for every rank r
// (arrays of `world` entries supplied by the caller; host pointers, valid until the next mgx_gather_start / mgx_gather_destroy),
// ready for mgx_results_from_raw[_labeled].  On the other ranks the arrays may be NULL and are left alone.
extern "C" int mgx_gather_finish(mgx_gather *g, uint64_t *n_queries, const void **headers, const uint32_t **streams, uint64_t *stream_words) {
    if (!g) return gfail(MGX_ERR_INVALID, "mgx_gather_finish: null argument");
    if (!g->pending) return gfail(MGX_ERR_INVALID, "mgx_gather_finish: no gather in flight");
    HIP_TRY_G(hipSetDevice(g->device));
    if (g->rank == g->root) {
        if (!n_queries || !headers || !streams || !stream_words) return gfail(MGX_ERR_INVALID, "mgx_gather_finish: the root needs its output arrays");
        const size_t hb = (size_t)g->h_off[(size_t)g->world], sb = (size_t)g->s_off[(size_t)g->world];
        HIP_TRY_G(g->h_headers.ensure(std::max<size_t>(hb, 16)));
        HIP_TRY_G(g->h_stream.ensure(std::max<size_t>(sb, 16)));
        if (hb) HIP_TRY_G(hipMemcpyAsync(g->h_headers.p, g->d_headers.p, hb, hipMemcpyDeviceToHost, g->stream));
        if (sb) HIP_TRY_G(hipMemcpyAsync(g->h_stream.p, g->d_stream.p, sb, hipMemcpyDeviceToHost, g->stream));
        HIP_TRY_G(hipStreamSynchronize(g->stream));
        for (int r = 0; r < g->world; ++r) {
            n_queries[r] = g->counts[2 * (size_t)r];
            stream_words[r] = g->counts[2 * (size_t)r + 1];
            headers[r] = (const char *)g->h_headers.p + g->h_off[(size_t)r];
            streams[r] = (const uint32_t *)((const char *)g->h_stream.p + g->s_off[(size_t)r]);
        }
    } else {
        HIP_TRY_G(hipStreamSynchronize(g->stream));
    }
    g->pending = false;
    return MGX_OK;
}

extern "C" int mgx_gather_world(const mgx_gather *g) { return g ? g->world : 0; }
extern "C" int mgx_gather_rank(const mgx_gather *g) { return g ? g->rank : -1; }
extern "C" uint64_t mgx_gather_unique_id_bytes(void) { return sizeof(ncclUniqueId); }
