// mem_access.hpp — global-memory accessors of the gfx950 wave headers (wave.hpp, wave_group.hpp).
//
// gld / gst read and write objects that are KNOWN to live in global memory (graph, arena, batch streams) with
// global_load / global_store instructions.  A plain dereference of a generic pointer is a FLAT instruction: it bumps
// the LDS counter as well as the vector-memory counter, so every LDS access after it waits for HBM and every wait on
// it becomes `s_waitcnt vmcnt(0) lgkmcnt(0)`.  The objects are moved as native 16- / 8- / 4-byte vectors: casting a
// pointer to an address-space-qualified HIP_vector_type (uint4, int4 ...) or struct does NOT do it — their assignment
// operators take a generic `this`, and the access silently becomes FLAT again (seen in the ISA of round 1's kernels).
#pragma once
#include <stdint.h>

namespace mgx_mem {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define MGX_GPTR(T, p) ((__attribute__((address_space(1))) T *)(uintptr_t)(p))

template <int BYTES>
__device__ __forceinline__ void load_bytes(const void *src, void *dst) {
    static_assert(BYTES > 0, "empty object");
    const char *s = (const char *)src;
    char *d = (char *)dst;
    if constexpr (BYTES >= 16) {
        u32x4 v = *MGX_GPTR(const u32x4, s);
        __builtin_memcpy(d, &v, 16);
        if constexpr (BYTES > 16) load_bytes<BYTES - 16>(s + 16, d + 16);
    } else if constexpr (BYTES >= 8) {
        u32x2 v = *MGX_GPTR(const u32x2, s);
        __builtin_memcpy(d, &v, 8);
        if constexpr (BYTES > 8) load_bytes<BYTES - 8>(s + 8, d + 8);
    } else if constexpr (BYTES >= 4) {
        uint32_t v = *MGX_GPTR(const uint32_t, s);
        __builtin_memcpy(d, &v, 4);
        if constexpr (BYTES > 4) load_bytes<BYTES - 4>(s + 4, d + 4);
    } else if constexpr (BYTES >= 2) {
        uint16_t v = *MGX_GPTR(const uint16_t, s);
        __builtin_memcpy(d, &v, 2);
        if constexpr (BYTES > 2) load_bytes<BYTES - 2>(s + 2, d + 2);
    } else {
        uint8_t v = *MGX_GPTR(const uint8_t, s);
        __builtin_memcpy(d, &v, 1);
    }
}

template <int BYTES>
__device__ __forceinline__ void store_bytes(void *dst, const void *src) {
    char *d = (char *)dst;
    const char *s = (const char *)src;
    if constexpr (BYTES >= 16) {
        u32x4 v;
        __builtin_memcpy(&v, s, 16);
        *MGX_GPTR(u32x4, d) = v;
        if constexpr (BYTES > 16) store_bytes<BYTES - 16>(d + 16, s + 16);
    } else if constexpr (BYTES >= 8) {
        u32x2 v;
        __builtin_memcpy(&v, s, 8);
        *MGX_GPTR(u32x2, d) = v;
        if constexpr (BYTES > 8) store_bytes<BYTES - 8>(d + 8, s + 8);
    } else if constexpr (BYTES >= 4) {
        uint32_t v;
        __builtin_memcpy(&v, s, 4);
        *MGX_GPTR(uint32_t, d) = v;
        if constexpr (BYTES > 4) store_bytes<BYTES - 4>(d + 4, s + 4);
    } else if constexpr (BYTES >= 2) {
        uint16_t v;
        __builtin_memcpy(&v, s, 2);
        *MGX_GPTR(uint16_t, d) = v;
        if constexpr (BYTES > 2) store_bytes<BYTES - 2>(d + 2, s + 2);
    } else {
        uint8_t v;
        __builtin_memcpy(&v, s, 1);
        *MGX_GPTR(uint8_t, d) = v;
    }
}
} // namespace mgx_mem
