// tests/emu/wave.hpp — TEST INFRASTRUCTURE ONLY.
// Lock-step host model of metagraph_amd/csrc/wave.hpp: a "wave" is 64 lanes evaluated one after the
// other inside every FOR_LANES region, LV<T> is a 64-entry array, and the wave_* functions are
// plain loops.  It lets the CPU-only unit tests run the kernels' wave programs (the very same
// source files as the HIP build) against the oracle.  It is put first on the include path by
// tests/emu/Makefile so that it shadows the gfx950 header; nothing here is ever compiled into
// libmgx.so and the product has no CPU execution path.
// Shares the include guard of metagraph_amd/csrc/wave.hpp: the test driver includes this file first,
// which turns the gfx950 header into a no-op for that translation unit only.
#ifndef MGX_WAVE_HPP_
#define MGX_WAVE_HPP_
#include <stdint.h>
#include <cmath>
#include <cstring>

#define MGX_DEV inline
#define MGX_DEV_NOINLINE
#define MGX_HD inline
// the control block of a wave program lives in LDS: tell the compiler, so that accesses through a `Wave &` that
// crossed a noinline call boundary become ds_read/ds_write instead of FLAT instructions (which also wait on vmcnt)
#define MGX_ASSUME_LDS(p) ((void)0)
#define MGX_WAVE_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r; r.x = x; r.y = y; return r; }

namespace mgx {

#ifndef MGX_EMU_WAVE
#define MGX_EMU_WAVE 64
#endif
constexpr int WAVE = MGX_EMU_WAVE;      // 64 = a wavefront; 16 / 8 / 1 model the sub-wave-group and thread-per-read kernels

inline int lane_id() { return 0; }

template <class T>
struct LV {
    T v[WAVE];
    T &operator[](int l) { return v[l]; }
    const T &operator[](int l) const { return v[l]; }
};

#define FOR_LANES(l) for (int l = 0; l < ::mgx::WAVE; ++l)

inline uint64_t wave_ballot(const LV<bool> &p) {
    uint64_t m = 0;
    for (int l = 0; l < WAVE; ++l) m |= (uint64_t)(p[l] ? 1 : 0) << l;
    return m;
}

template <class T>
inline T wave_bcast(const LV<T> &x, int src) { return x[src]; }

inline LV<int32_t> wave_shift_up1(const LV<int32_t> &x, int32_t fill) {
    LV<int32_t> r;
    r[0] = fill;
    for (int l = 1; l < WAVE; ++l) r[l] = x[l - 1];
    return r;
}

inline LV<int32_t> wave_shift_down(const LV<int32_t> &x, int32_t n, int32_t fill) {
    LV<int32_t> r;
    for (int l = 0; l < WAVE; ++l) r[l] = l + n < WAVE ? x[l + n] : fill;
    return r;
}

inline LV<int32_t> wave_prefix_max(const LV<int32_t> &x) {
    LV<int32_t> r;
    int32_t m = x[0];
    for (int l = 0; l < WAVE; ++l) { m = x[l] > m ? x[l] : m; r[l] = m; }
    return r;
}

inline int32_t wave_max(const LV<int32_t> &x) {
    int32_t m = x[0];
    for (int l = 1; l < WAVE; ++l) m = x[l] > m ? x[l] : m;
    return m;
}

inline int32_t wave_min(const LV<int32_t> &x) {
    int32_t m = x[0];
    for (int l = 1; l < WAVE; ++l) m = x[l] < m ? x[l] : m;
    return m;
}

inline uint64_t wave_max_u64(const LV<uint64_t> &x) {
    uint64_t m = x[0];
    for (int l = 1; l < WAVE; ++l) m = x[l] > m ? x[l] : m;
    return m;
}

inline int32_t wave_sum(const LV<int32_t> &x) {
    int32_t s = 0;
    for (int l = 0; l < WAVE; ++l) s += x[l];
    return s;
}

inline LV<int32_t> wave_prefix_sum_excl(const LV<int32_t> &x) {
    LV<int32_t> r;
    int32_t s = 0;
    for (int l = 0; l < WAVE; ++l) { r[l] = s; s += x[l]; }
    return r;
}

inline void wave_sync() {}

typedef struct { uint32_t v[16]; uint32_t operator[](int i) const { return v[i]; } } u32x16;
inline u32x16 sload_x16(const void *p) { u32x16 r; memcpy(r.v, p, 64); return r; }
inline uint32_t sload_u32(const uint32_t *p) { return *p; }
// host model of wave.hpp's register-resident 64-entry table (only meaningful for a 64-lane wave)
#if MGX_EMU_WAVE == 64
#define MGX_HAS_REGTAB 1
struct RegTab64 {
    int32_t v[64];
    int32_t get(int i) const { return v[i & 63]; }
    void add(int i, int32_t d) { v[i & 63] += d; }
    void set(int i, int32_t x) { v[i & 63] = x; }
    void fill(int32_t x) { for (int i = 0; i < 64; ++i) v[i] = x; }
};
#else
#define MGX_HAS_REGTAB 0
#endif

// host model of wave.hpp's global-memory accessors
template <class T> inline T gld(const T *p) { return *p; }
template <class T, class V> inline void gst(T *p, V v) { *p = (T)v; }
inline uint8_t lds_u8(const void *p) { return *(const uint8_t *)p; }
inline int8_t lds_i8(const void *p) { return *(const int8_t *)p; }
inline void gst4(int32_t *p, int32_t a, int32_t b, int32_t c, int32_t d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
template <class T, class V> inline void gst_stream(T *p, V v) { *p = (T)v; }
template <class T> inline T gld_stream(const T *p) { return *p; }
inline uint64_t gld_stream_u64(const void *p) { uint64_t v; memcpy(&v, p, 8); return v; }

inline uint64_t cycle_clock() { return 0; }
inline int popc64(uint64_t x) { return __builtin_popcountll(x); }
inline int ctz64(uint64_t x) { return __builtin_ctzll(x); }
inline int clz64(uint64_t x) { return __builtin_clzll(x); }
inline double fma_f64(double a, double b, double c) { return std::fma(a, b, c); }

template <class T>
inline T uni(T x) { return x; }

} // namespace mgx
#endif  // MGX_WAVE_HPP_
