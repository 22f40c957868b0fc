// wave.hpp — the only place kernels touch gfx950 wave intrinsics.
//
// Kernel bodies are written as "wave programs": uniform (wave-scalar) control flow, with
// lane-parallel regions spelled FOR_LANES(l) { ... } over per-lane values LV<T>, and cross-lane
// traffic only through the wave_* functions below.  On gfx950 a wave is 64 lanes; LV<T> is one
// register per lane and FOR_LANES runs its body once on every lane.
//
// (tests/emu/wave.hpp is a lock-step host model of exactly this interface used by the CPU-only
// unit tests to check kernel logic against the oracle.  It is never part of libmgx.so.)
#ifndef MGX_WAVE_HPP_
#define MGX_WAVE_HPP_
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mem_access.hpp"

#define MGX_DEV __device__ __forceinline__
#ifndef MGX_NOINLINE
#define MGX_NOINLINE 1
#endif
#if MGX_NOINLINE
#define MGX_DEV_NOINLINE __device__ __noinline__
#else
#define MGX_DEV_NOINLINE __device__ __forceinline__
#endif
#define MGX_HD __host__ __device__ __forceinline__
// the control block of a wave program lives in LDS: tell the compiler, so that accesses through a `Wave &` that
// crossed a noinline call boundary become ds_read/ds_write instead of FLAT instructions (which also wait on vmcnt)
#if defined(__HIP_DEVICE_COMPILE__)
#define MGX_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void *)(p)))
#else
#define MGX_ASSUME_LDS(p) ((void)0)
#endif
#define MGX_WAVE_EMU 0

namespace mgx {

constexpr int WAVE = 64;

MGX_DEV int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// one value per lane
template <class T>
struct LV {
    T v;
    MGX_DEV T &operator[](int) { return v; }
    MGX_DEV const T &operator[](int) const { return v; }
};

#define FOR_LANES(l) for (int l = ::mgx::lane_id(), l##_once = 1; l##_once; l##_once = 0)

MGX_DEV uint64_t wave_ballot(const LV<bool> &p) { return __ballot(p.v); }

MGX_DEV int32_t shfl_i32(int32_t v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }

template <class T>
MGX_DEV T wave_bcast(const LV<T> &x, int src) {
    if constexpr (sizeof(T) == 8) {
        uint64_t u = (uint64_t)x.v;
        uint32_t lo = (uint32_t)shfl_i32((int32_t)(uint32_t)u, src);
        uint32_t hi = (uint32_t)shfl_i32((int32_t)(uint32_t)(u >> 32), src);
        return (T)(((uint64_t)hi << 32) | lo);
    } else {
        return (T)shfl_i32((int32_t)x.v, src);
    }
}

// value of lane (l - 1); lane 0 receives `fill`
MGX_DEV LV<int32_t> wave_shift_up1(const LV<int32_t> &x, int32_t fill) {
    LV<int32_t> r;
    r.v = __builtin_amdgcn_update_dpp(fill, x.v, 0x138, 0xF, 0xF, false);     // wave_shr:1, lane 0 keeps `fill`
    return r;
}

// value of lane (l + n) (n >= 0, wave-uniform); lanes past the end receive `fill`
MGX_DEV LV<int32_t> wave_shift_down(const LV<int32_t> &x, int32_t n, int32_t fill) {
    LV<int32_t> r;
    const int src = lane_id() + n;
    const int32_t t = __builtin_amdgcn_ds_bpermute((src & (WAVE - 1)) << 2, x.v);
    r.v = src < WAVE ? t : fill;
    return r;
}

// inclusive prefix max over lanes: 4 row-shift DPP steps inside each row of 16, then row_bcast:15 and
// row_bcast:31 carry the row totals across rows (the classic GCN/CDNA wave64 scan; no LDS traffic)
#define MGX_DPP_MAX(v, ctrl, rmask, bmask)                                                          \
    { int32_t t_ = __builtin_amdgcn_update_dpp(INT32_MIN, v, ctrl, rmask, bmask, false); v = t_ > v ? t_ : v; }
MGX_DEV LV<int32_t> wave_prefix_max(const LV<int32_t> &x) {
    int32_t v = x.v;
    MGX_DPP_MAX(v, 0x111, 0xF, 0xF)      // row_shr:1
    MGX_DPP_MAX(v, 0x112, 0xF, 0xF)      // row_shr:2
    MGX_DPP_MAX(v, 0x114, 0xF, 0xF)      // row_shr:4
    MGX_DPP_MAX(v, 0x118, 0xF, 0xF)      // row_shr:8
    MGX_DPP_MAX(v, 0x142, 0xA, 0xF)      // row_bcast:15 -> rows 1 and 3
    MGX_DPP_MAX(v, 0x143, 0xC, 0xF)      // row_bcast:31 -> rows 2 and 3
    LV<int32_t> r;
    r.v = v;
    return r;
}

MGX_DEV int32_t wave_max(const LV<int32_t> &x) {
    LV<int32_t> p = wave_prefix_max(x);
    return __builtin_amdgcn_readlane(p.v, 63);
}

MGX_DEV int32_t wave_min(const LV<int32_t> &x) {
    LV<int32_t> n;
    n.v = ~x.v;                              // min(x) == ~max(~x) in two's complement
    LV<int32_t> p = wave_prefix_max(n);
    return ~__builtin_amdgcn_readlane(p.v, 63);
}

MGX_DEV uint64_t wave_max_u64(const LV<uint64_t> &x) {
    uint64_t v = x.v;
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int32_t)(uint32_t)v, d, WAVE);
        uint32_t hi = (uint32_t)__shfl_xor((int32_t)(uint32_t)(v >> 32), d, WAVE);
        uint64_t t = ((uint64_t)hi << 32) | lo;
        v = t > v ? t : v;
    }
    return v;
}

MGX_DEV int32_t wave_sum(const LV<int32_t> &x) {
    int32_t v = x.v;
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

// exclusive prefix sum over lanes
MGX_DEV LV<int32_t> wave_prefix_sum_excl(const LV<int32_t> &x) {
    int32_t v = x.v;
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        int32_t t = __shfl_up(v, d, WAVE);
        if (l >= d) v += t;
    }
    LV<int32_t> r;
    r.v = v - x.v;
    return r;
}

// Make this wave's earlier memory writes visible to its later reads made by other lanes.
MGX_DEV void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// wave-uniform value -> scalar register (all lanes must hold the same value)
MGX_DEV int32_t uni(int32_t x) { return __builtin_amdgcn_readfirstlane(x); }
MGX_DEV uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)x); }
MGX_DEV uint64_t uni(uint64_t x) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)x);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
MGX_DEV int64_t uni(int64_t x) { return (int64_t)uni((uint64_t)x); }
MGX_DEV bool uni(bool x) { return __builtin_amdgcn_readfirstlane((int32_t)x) != 0; }

// Wave-uniform loads through the scalar data cache: one SMEM instruction, result in SGPRs, so that all
// arithmetic on it is issued by the scalar unit instead of 64 redundant vector lanes.  Only for
// read-only data (the graph) at wave-uniform addresses.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
MGX_DEV uint64_t sgpr_addr(const void *p) {
    uint64_t a = (uint64_t)p;
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)a);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)(a >> 32));
    return ((uint64_t)hi << 32) | lo;
}
MGX_DEV u32x16 sload_x16(const void *p) {
    u32x16 v;
    uint64_t a = sgpr_addr(p);
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(a) : "memory");
    return v;
}
MGX_DEV uint32_t sload_u32(const uint32_t *p) {
    uint32_t v;
    uint64_t a = sgpr_addr(p);
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(a) : "memory");
    return v;
}

// A 64-entry table of small integers held one entry per lane in a VGPR (wave-uniform index): reads are one
// v_readlane, updates one predicated add; no memory traffic.  Used by sdust for its triplet counters and window.
#define MGX_HAS_REGTAB 1
struct RegTab64 {
    int32_t v;
    MGX_DEV int32_t get(int i) const { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(i)); }
    MGX_DEV void add(int i, int32_t d) { if (lane_id() == i) v += d; }
    MGX_DEV void set(int i, int32_t x) { if (lane_id() == i) v = x; }
    MGX_DEV void fill(int32_t x) { v = x; }
};

// Loads / stores that are known to target global memory (graph, arena): global_* instead of FLAT instructions, so
// that they do not bump lgkmcnt and LDS traffic never waits for them (see mem_access.hpp for why they are spelled
// through native vectors).  Objects must be naturally aligned for their size class (16 / 8 / 4 / 2 / 1 bytes).
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> MGX_DEV T gld(const T *p) { T t; mgx_mem::load_bytes<(int)sizeof(T)>(p, &t); return t; }
template <class T, class V> MGX_DEV void gst(T *p, V v) { const T t = (T)v; mgx_mem::store_bytes<(int)sizeof(T)>(p, &t); }
#else
template <class T> MGX_DEV T gld(const T *p) { return *p; }
template <class T, class V> MGX_DEV void gst(T *p, V v) { *p = (T)v; }
#endif
// loads through pointers that are KNOWN to point into LDS (generic -> local is a truncation on gfx9): ds_read instead of
// FLAT, which would wait on the vector-memory counter as well
#if defined(__HIP_DEVICE_COMPILE__)
MGX_DEV uint8_t lds_u8(const void *p) { return *(const __attribute__((address_space(3))) uint8_t *)(uint32_t)(uint64_t)p; }
MGX_DEV int8_t lds_i8(const void *p) { return *(const __attribute__((address_space(3))) int8_t *)(uint32_t)(uint64_t)p; }
#else
MGX_DEV uint8_t lds_u8(const void *p) { return *(const uint8_t *)p; }
MGX_DEV int8_t lds_i8(const void *p) { return *(const int8_t *)p; }
#endif
// four consecutive int32 as one 16-byte store (p must be 16-byte aligned)
MGX_DEV void gst4(int32_t *p, int32_t a, int32_t b, int32_t c, int32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
    mgx_mem::u32x4 v = { (uint32_t)a, (uint32_t)b, (uint32_t)c, (uint32_t)d };
    *MGX_GPTR(mgx_mem::u32x4, p) = v;
#else
    p[0] = a; p[1] = b; p[2] = c; p[3] = d;
#endif
}
// one-shot 8-byte load that should not displace reusable lines (graph blocks, hints) from L2 / Infinity Cache
// write-once / read-once scalars of the batch streams (node ids, match lengths, ranges): nontemporal
template <class T, class V> MGX_DEV void gst_stream(T *p, V v) {
#if defined(MGX_PROBE_NO_STREAM_STORES)
    (void)p; (void)v;                      // timing probe only (results are WRONG): what do k_map's output stores cost?
#elif defined(MGX_PROBE_PLAIN_STREAM_STORES) && defined(__HIP_DEVICE_COMPILE__)
    *(__attribute__((address_space(1))) T *)p = (T)v;
#elif defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store((T)v, (__attribute__((address_space(1))) T *)p);
#else
    *p = (T)v;
#endif
}
template <class T> MGX_DEV T gld_stream(const T *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) T *)p);
#else
    return *p;
#endif
}
MGX_DEV uint64_t gld_stream_u64(const void *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) unsigned long long *)p);
#else
    return *(const unsigned long long *)p;
#endif
}

MGX_DEV uint64_t cycle_clock() { return __builtin_readcyclecounter(); }
MGX_DEV int popc64(uint64_t x) { return __popcll(x); }
MGX_DEV int ctz64(uint64_t x) { return __ffsll((long long)x) - 1; }           // x != 0
MGX_DEV int clz64(uint64_t x) { return __clzll((long long)x); }                // x != 0
MGX_DEV double fma_f64(double a, double b, double c) { return __fma_rn(a, b, c); }

} // namespace mgx
#endif  // MGX_WAVE_HPP_
