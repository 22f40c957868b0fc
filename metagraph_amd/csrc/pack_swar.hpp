// pack_swar.hpp — 32 bases -> 32 2-bit codes + 32 invalid flags without a per-base loop (k_pack_reads, mgx.hip).
// The same results as graph_build.hpp's pack_read_word() — KmerExtractorBOSS::encode (A C G T/U in either case = 0..3, anything else
// invalid) and, for strand 1, the reverse complement — from the word's 32 bytes held as four 64-bit values: eight bases at a time.
// tests/test_pack_swar.py compiles this header for the host and compares it with the byte loop on random and adversarial bytes.
#pragma once
#include <cstdint>

#ifndef MGX_PS_FN
#if defined(__HIPCC__)
#define MGX_PS_FN __host__ __device__ __forceinline__
#else
#define MGX_PS_FN inline
#endif
#endif

namespace mgx_pack {

// 0x80 in every byte of v that is zero (exact: no carries between bytes)
MGX_PS_FN uint64_t zero_bytes(uint64_t v) {
    const uint64_t m = 0x7F7F7F7F7F7F7F7Full;
    return ~(((v & m) + m) | v | m);
}
MGX_PS_FN uint64_t eq_bytes(uint64_t v, uint8_t c) { return zero_bytes(v ^ (0x0101010101010101ull * c)); }

// eight characters -> their codes (A C G T = 0..3) in bits 0..15 (character i: bits 2i, 2i + 1) and bit i of *bad set where character i
// is none of ACGTUacgtu
MGX_PS_FN uint32_t pack8(uint64_t x, uint32_t *bad) {
    const uint64_t u = x & 0xDFDFDFDFDFDFDFDFull;                       // upper case (bit 7 stays: such a byte matches nothing below)
    const uint64_t ok = eq_bytes(u, 0x41) | eq_bytes(u, 0x43) | eq_bytes(u, 0x47) | eq_bytes(u, 0x54) | eq_bytes(u, 0x55);
    *bad = (uint32_t)(((~ok & 0x8080808080808080ull) * 0x0002040810204081ull) >> 56);
    // A 0x41, C 0x43, G 0x47, T 0x54, U 0x55 (and + 0x20): bits 1 ^ 2 give the low code bit, bits 2 ^ 3 the high one
    uint64_t y = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ull;
    y = (y | (y >> 6)) & 0x000F000F000F000Full;
    y = (y | (y >> 12)) & 0x000000FF000000FFull;
    y = (y | (y >> 24)) & 0xFFFFull;
    return (uint32_t)y;
}

// every bit of v twice: bit i -> bits 2i, 2i + 1
MGX_PS_FN uint64_t double_bits(uint32_t v) {
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x | (x << 1);
}

// b: 32 characters (character i in byte i & 7 of b[i >> 3]).  strand 0: the first n of them are positions 0 .. n - 1 of the word.
// strand 1: the LAST n of them, backwards and complemented, are positions 0 .. n - 1 (character 31 is position 0).
MGX_PS_FN void pack32(const uint64_t b[4], int n, int strand, uint64_t *codes, uint32_t *inv) {
    uint64_t c = 0;
    uint32_t v = 0;
    for (int q = 0; q < 4; ++q) {
        uint32_t bad;
        c |= (uint64_t)pack8(b[q], &bad) << (16 * q);
        v |= bad << (8 * q);
    }
    if (strand) {
        // reverse the order of the 2-bit groups (and of the flags), complement: 3 - code
        c = __builtin_bswap64(c);
        c = ((c >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((c & 0x0F0F0F0F0F0F0F0Full) << 4);
        c = ((c >> 2) & 0x3333333333333333ull) | ((c & 0x3333333333333333ull) << 2);
        c = ~c;
        uint32_t r = v;
        r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
        r = ((r >> 2) & 0x33333333u) | ((r & 0x33333333u) << 2);
        r = ((r >> 4) & 0x0F0F0F0Fu) | ((r & 0x0F0F0F0Fu) << 4);
        v = __builtin_bswap32(r);
    }
    if (n < 32) { v &= (1u << n) - 1u; c &= (1ull << (2 * n)) - 1ull; }
    c &= ~double_bits(v);                            // an invalid position holds no code
    *codes = c;
    *inv = v;
}

} // namespace mgx_pack
