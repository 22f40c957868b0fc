// wave_lane.hpp — the wave interface of wave.hpp collapsed to ONE lane per wave program.
// Compiling the kernels' wave programs against this header turns "one wavefront per read" into
// "one thread per read" (64 reads per hardware wavefront): FOR_LANES runs once, LV<T> is a scalar and all
// cross-lane operations are identities.  Used by mgx_lane.hip for the thread-per-read aligner kernel.
#ifndef MGX_WAVE_HPP_
#define MGX_WAVE_HPP_
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

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
#define MGX_ASSUME_LDS(p) ((void)0)
#define MGX_WAVE_EMU 0
#define MGX_LANE_MODE 1

namespace mgx {

constexpr int WAVE = 1;

MGX_DEV int lane_id() { return 0; }

template <class T>
struct LV {
    T v;
    MGX_DEV T &operator[](int) { return v; }
    MGX_DEV const T &operator[](int) const { return v; }
};

#define FOR_LANES(l) for (int l = 0, l##_once = 1; l##_once; l##_once = 0)

MGX_DEV uint64_t wave_ballot(const LV<bool> &p) { return p.v ? 1ull : 0ull; }
template <class T> MGX_DEV T wave_bcast(const LV<T> &x, int) { return x.v; }
MGX_DEV LV<int32_t> wave_shift_up1(const LV<int32_t> &, int32_t fill) { LV<int32_t> r; r.v = fill; return r; }
MGX_DEV LV<int32_t> wave_prefix_max(const LV<int32_t> &x) { return x; }
MGX_DEV int32_t wave_max(const LV<int32_t> &x) { return x.v; }
MGX_DEV int32_t wave_min(const LV<int32_t> &x) { return x.v; }
MGX_DEV uint64_t wave_max_u64(const LV<uint64_t> &x) { return x.v; }
MGX_DEV int32_t wave_sum(const LV<int32_t> &x) { return x.v; }
MGX_DEV LV<int32_t> wave_prefix_sum_excl(const LV<int32_t> &) { LV<int32_t> r; r.v = 0; return r; }
MGX_DEV void wave_sync() {}

template <class T> MGX_DEV T uni(T x) { return x; }

struct u32x16 { uint32_t v[16]; MGX_DEV uint32_t operator[](int i) const { return v[i]; } };
MGX_DEV u32x16 sload_x16(const void *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    u32x16 r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = c.x; r.v[9] = c.y; r.v[10] = c.z; r.v[11] = c.w; r.v[12] = d.x; r.v[13] = d.y; r.v[14] = d.z; r.v[15] = d.w;
    return r;
}
MGX_DEV uint32_t sload_u32(const uint32_t *p) { return *p; }

#define MGX_HAS_REGTAB 0          // fewer than 64 lanes per read: sdust keeps its tables in memory

// Loads / stores that are known to target global memory (graph, arena): global_* instead of FLAT instructions, so
// that they do not bump lgkmcnt and LDS traffic never waits for them.
#if defined(__HIP_DEVICE_COMPILE__)
#define MGX_AS_GLOBAL(T, p) ((__attribute__((address_space(1))) T *)(p))
#else
#define MGX_AS_GLOBAL(T, p) (p)
#endif
template <class T> MGX_DEV T gld(const T *p) { return *MGX_AS_GLOBAL(const T, p); }
template <class T, class V> MGX_DEV void gst(T *p, V v) { *MGX_AS_GLOBAL(T, p) = (T)v; }
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
