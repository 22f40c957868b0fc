// wave_group.hpp — the wave interface of wave.hpp for SUB-WAVE GROUPS: a hardware wavefront is split into
// 64 / MGX_GROUP independent groups of MGX_GROUP lanes (16 = one DPP row, or 8), and every group runs its
// own wave program (its own read).  All cross-lane operations are scoped to the group: DPP row shifts never
// leave a 16-lane row, ballots are cut to the group's bits, broadcasts go through ds_bpermute.  Control flow
// that is "uniform" inside a wave program is uniform per GROUP here and may diverge between the groups of a
// wavefront; the hardware's EXEC masking serialises such paths, everything else is issued once for all
// groups — which is the point: the scalar bookkeeping of the aligner is shared by 4 (8) reads per issue slot.
#ifndef MGX_WAVE_HPP_
#define MGX_WAVE_HPP_
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mem_access.hpp"

#ifndef MGX_GROUP
#define MGX_GROUP 16
#endif
static_assert(MGX_GROUP == 16 || MGX_GROUP == 8 || MGX_GROUP == 4, "group = 4, 8 or 16 lanes");

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
#define MGX_GROUP_MODE 1

namespace mgx {

constexpr int WAVE = MGX_GROUP;
constexpr int GROUPS_PER_WAVEFRONT = 64 / MGX_GROUP;

MGX_DEV int hw_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
MGX_DEV int lane_id() { return hw_lane() & (WAVE - 1); }
MGX_DEV int group_id() { return hw_lane() / WAVE; }
MGX_DEV int group_base() { return hw_lane() & ~(WAVE - 1); }

template <class T>
struct LV {
    T v;
    MGX_DEV T &operator[](int) { return v; }
    MGX_DEV const T &operator[](int) const { return v; }
};

#define FOR_LANES(l) for (int l = ::mgx::lane_id(), l##_once = 1; l##_once; l##_once = 0)

MGX_DEV uint64_t wave_ballot(const LV<bool> &p) {
    uint64_t m = __ballot(p.v);
    return (m >> group_base()) & ((1ull << WAVE) - 1);
}

MGX_DEV int32_t grp_shfl(int32_t v, int src) { return __builtin_amdgcn_ds_bpermute((group_base() + src) << 2, v); }

template <class T>
MGX_DEV T wave_bcast(const LV<T> &x, int src);
// value of lane (l - 1) of the group; lane 0 receives `fill`
MGX_DEV LV<int32_t> wave_shift_up1(const LV<int32_t> &x, int32_t fill) {
    LV<int32_t> r;
    int32_t t = __builtin_amdgcn_update_dpp(fill, x.v, 0x111, 0xF, 0xF, false);      // row_shr:1
    r.v = lane_id() == 0 ? fill : t;
    return r;
}

// value of lane (l + n) of the group (n >= 0, group-uniform); lanes past the group's end receive `fill`
MGX_DEV LV<int32_t> wave_shift_down(const LV<int32_t> &x, int32_t n, int32_t fill) {
    LV<int32_t> r;
    const int src = lane_id() + n;
    const int32_t t = __builtin_amdgcn_ds_bpermute((group_base() + (src & (WAVE - 1))) << 2, x.v);
    r.v = src < WAVE ? t : fill;
    return r;
}

// inclusive prefix max inside the group: log2(group) row-shift DPP steps
MGX_DEV LV<int32_t> wave_prefix_max(const LV<int32_t> &x) {
    int32_t v = x.v;
    const int l = lane_id();
#define MGX_GRP_MAX(ctrl, d)                                                                       \
    if (WAVE > d) { int32_t t_ = __builtin_amdgcn_update_dpp(INT32_MIN, v, ctrl, 0xF, 0xF, false);  \
                    t_ = (WAVE == 16 || l >= d) ? t_ : INT32_MIN; v = t_ > v ? t_ : v; }
    MGX_GRP_MAX(0x111, 1)
    MGX_GRP_MAX(0x112, 2)
    MGX_GRP_MAX(0x114, 4)
    MGX_GRP_MAX(0x118, 8)
#undef MGX_GRP_MAX
    LV<int32_t> r;
    r.v = v;
    return r;
}

// All-lanes reductions inside the group WITHOUT the LDS crossbar: a butterfly over DPP quad permutes ([1,0,3,2], [2,3,0,1]),
// the half-row mirror (lane i <-> 7 - i: the other quad of an 8-lane group) and, for 16 lanes, the row mirror.  ds_bpermute
// costs an LDS round trip (~100 cycles) per use and the extension kernel is bound by exactly such dependent round trips; a
// DPP step is a VALU instruction.
#define MGX_GRP_BFLY(OP)                                                                                        \
    { int32_t t_ = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false); v = OP(v, t_); }                 \
    { int32_t t_ = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false); v = OP(v, t_); }                 \
    if (WAVE >= 8) { int32_t t_ = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false); v = OP(v, t_); } \
    if (WAVE >= 16) { int32_t t_ = __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false); v = OP(v, t_); }
#define MGX_OP_MAX(a, b) ((a) > (b) ? (a) : (b))
#define MGX_OP_OR(a, b) ((a) | (b))
#define MGX_OP_ADD(a, b) ((a) + (b))
MGX_DEV int32_t grp_all_max(int32_t v) { MGX_GRP_BFLY(MGX_OP_MAX) return v; }
MGX_DEV int32_t grp_all_or(int32_t v) { MGX_GRP_BFLY(MGX_OP_OR) return v; }
MGX_DEV int32_t grp_all_add(int32_t v) { MGX_GRP_BFLY(MGX_OP_ADD) return v; }
#undef MGX_GRP_BFLY

// value of lane `src` (group-uniform) in every lane of the group: the source lane's bits OR-ed across the group
template <class T>
MGX_DEV T wave_bcast(const LV<T> &x, int src) {
    const bool me = lane_id() == src;
    if constexpr (sizeof(T) == 8) {
        const uint64_t u = (uint64_t)x.v;
        const uint32_t lo = (uint32_t)grp_all_or(me ? (int32_t)(uint32_t)u : 0);
        const uint32_t hi = (uint32_t)grp_all_or(me ? (int32_t)(uint32_t)(u >> 32) : 0);
        return (T)(((uint64_t)hi << 32) | lo);
    } else {
        return (T)grp_all_or(me ? (int32_t)x.v : 0);
    }
}

MGX_DEV int32_t wave_max(const LV<int32_t> &x) { return grp_all_max(x.v); }

MGX_DEV int32_t wave_min(const LV<int32_t> &x) { return ~grp_all_max(~x.v); }

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

MGX_DEV int32_t wave_sum(const LV<int32_t> &x) { return grp_all_add(x.v); }

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

MGX_DEV void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// group-uniform values stay in vector registers (they differ between the groups of a wavefront)
template <class T> MGX_DEV T uni(T x) { return x; }

template <class T> MGX_DEV T gld(const T *p);
struct u32x16 { uint32_t v[16]; MGX_DEV uint32_t operator[](int i) const { return v[i]; } };
// "uniform" loads of graph data: per-group uniform here, so plain vector loads — but explicitly GLOBAL ones (the graph
// never lives in LDS; a generic pointer would make them FLAT instructions that also wait on the LDS counter)
#if defined(__HIP_DEVICE_COMPILE__)
#define MGX_AS_GLOBAL_EARLY(T, p) ((const __attribute__((address_space(1))) T *)(p))
#else
#define MGX_AS_GLOBAL_EARLY(T, p) ((const T *)(p))
#endif
MGX_DEV u32x16 sload_x16(const void *p) {
    u32x16 r;
#if defined(__HIP_DEVICE_COMPILE__)
    mgx_mem::load_bytes<64>(p, r.v);
#else
    __builtin_memcpy(r.v, p, 64);
#endif
    return r;
}
MGX_DEV uint32_t sload_u32(const uint32_t *p) { return gld(p); }

#define MGX_HAS_REGTAB 0          // fewer than 64 lanes per read: sdust keeps its tables in memory

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
#if defined(__HIP_DEVICE_COMPILE__)
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
MGX_DEV int ctz64(uint64_t x) { return __ffsll((long long)x) - 1; }
MGX_DEV int clz64(uint64_t x) { return __clzll((long long)x); }
MGX_DEV double fma_f64(double a, double b, double c) { return __fma_rn(a, b, c); }

} // namespace mgx
#endif  // MGX_WAVE_HPP_
