// sx_lmskey.hpp -- prefix keys of suffixes: the number a suffix's first C symbols spell, and with it the symbol window of
// the suffix (sx_window.hpp) in the key's unsorted bits.  Shared by the key kernels of sx_lmssort.hip and by the first pass
// of the hybrid sort (sx_radix.hip), which computes the keys of a four-letter text's LMS suffixes itself.
#pragma once
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_window.hpp"

namespace sx {

// The first C symbols of suffix p as one number in base `base` (= largest symbol + 1),
// most significant first: numeric order == lexicographic order, the sentinel (0) is the
// smallest digit, and no bits are wasted when the alphabet is not a power of two
// (DNA + sentinel: base 5, 17 symbols in 40 bits).  Three aligned 16-byte loads when
// C <= 32 (statically indexed: no scratch), byte loads otherwise.
// A 64-bit multiply per symbol would cost more than everything else in the kernel, so the
// symbols are taken G at a time with base^G <= 2^24: inside a group the Horner steps are
// 24-bit multiply-adds, and the 64-bit accumulator is touched once per group (DNA: G = 10,
// two groups for 17 symbols).  G is one of four compile-time sizes so that the group ends
// are static; the last group is the short one.
// Bases up to 6 (DNA + sentinel = 5) go four symbols at a time: the weights base^3, base^2, base, 1 of a
// word's symbols fit a byte each, so one v_dot4_u32_u8 is the Horner step of a whole word; three words
// (base^12 < 2^32) are joined with 24-bit multiply-adds before the 64-bit accumulator is touched.  17
// symbols: 5 dot products, 3 short multiply-adds and one long one instead of 17 extract-multiply-add steps
// (the key kernel is bound by vector instructions, not by memory).
struct pkey_cfg {
    uint32_t base, C;
    uint32_t G;    // 10, 6, 4 or 3: the largest of these with base^G <= 2^24
    uint32_t powG; // base^G
    uint32_t powR; // base^(C mod G)
    uint32_t dot;   // base <= 6: the dot-product form below
    uint32_t coef4; // base^3 | base^2 << 8 | base << 16 | 1 << 24
    uint32_t B4, B12, Br; // base^4, base^12, base^(C mod 4)
    uint32_t powT;  // weight of the last, short group: base^(4 * ((C / 4) mod 3) + C mod 4)
};
static inline pkey_cfg pkey_make(uint32_t base, uint32_t C)
{
    pkey_cfg k{base, C, base <= 5 ? 10u : (base <= 16 ? 6u : (base <= 64 ? 4u : 3u)), 1, 1, 0, 0, 1, 1, 1, 1};
    for (uint32_t i = 0; i < k.G; ++i) k.powG *= base;
    for (uint32_t i = 0; i < C % k.G; ++i) k.powR *= base;
    if (base >= 2 && base <= 6 && C <= 32) {
        k.dot = 1;
        k.coef4 = (base * base * base) | (base * base) << 8 | base << 16 | 1u << 24;
        for (uint32_t i = 0; i < 4; ++i) k.B4 *= base;
        for (uint32_t i = 0; i < 12; ++i) k.B12 *= base;
        for (uint32_t i = 0; i < C % 4; ++i) k.Br *= base;
        for (uint32_t i = 0; i < 4 * ((C / 4) % 3) + C % 4; ++i) k.powT *= base;
    }
    return k;
}
// kw[k] holds symbols 4k .. 4k+3 of the prefix, the first one in the low byte
template <int NW> __device__ __forceinline__ uint64_t prefix_key_dot(const uint32_t (&kw)[NW], const pkey_cfg &kc)
{
    const uint32_t nw = kc.C >> 2, r = kc.C & 3u;
    uint64_t acc = 0;
    uint32_t g = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        if ((uint32_t)k < nw) { // uniform
            g = __builtin_amdgcn_udot4(kw[k], kc.coef4, __umul24(g, kc.B4), false);
            if (k % 3 == 2) { // static
                acc = acc * kc.B12 + g;
                g = 0;
            }
        } else if ((uint32_t)k == nw && r) { // uniform
            g = __builtin_amdgcn_udot4(kw[k], kc.coef4 >> (8u * (4u - r)), __umul24(g, kc.Br), false);
        }
    }
    if (kc.powT > 1) acc = acc * kc.powT + g; // uniform
    return acc;
}
template <int G>
__device__ __forceinline__ uint64_t prefix_key_grouped(const uint64_t (&q)[4], const pkey_cfg &kc)
{
    uint64_t acc = 0;
    uint32_t g = 0;
#pragma unroll
    for (uint32_t s = 0; s < 32; ++s) {
        if (s < kc.C) { // uniform
            g = __umul24(g, kc.base) + (uint32_t)((q[s >> 3] >> (8u * (s & 7u))) & 0xFFull);
            if ((s + 1) % G == 0) { // static
                acc = acc * kc.powG + g;
                g = 0;
            }
        }
    }
    if (kc.C % G) acc = acc * kc.powR + g; // uniform
    return acc;
}
__device__ __forceinline__ uint64_t prefix_key_of(const uint64_t (&q)[4], const pkey_cfg &kc)
{
    if (kc.dot) { // uniform
        const uint32_t kw[8] = {(uint32_t)q[0], (uint32_t)(q[0] >> 32), (uint32_t)q[1], (uint32_t)(q[1] >> 32),
                                (uint32_t)q[2], (uint32_t)(q[2] >> 32), (uint32_t)q[3], (uint32_t)(q[3] >> 32)};
        return prefix_key_dot<8>(kw, kc);
    }
    switch (kc.G) { // uniform
    case 10: return prefix_key_grouped<10>(q, kc);
    case 6: return prefix_key_grouped<6>(q, kc);
    case 4: return prefix_key_grouped<4>(q, kc);
    default: return prefix_key_grouped<3>(q, kc);
    }
}
__device__ __forceinline__ uint64_t prefix_key(const uint8_t *__restrict__ T, uint64_t p, const pkey_cfg &kc)
{
    if (kc.C <= 32) {
        uint64_t q[4];
        load_bytes32(T, p, q);
        return prefix_key_of(q, kc);
    }
    uint64_t acc = 0;
    for (uint32_t s = 0; s < kc.C; ++s) acc = acc * kc.base + (uint64_t)T[p + s];
    return acc;
}

// Key and window of a suffix for DNA-like texts with everything static: C key symbols of a base <= 6, a window
// of CW two-bit codes.  The bytes text[p - CW .. p + C) come from the staged tile as nine aligned words and
// one byte-align step each; window and key are dot products (sx_window.hpp, prefix_key_dot).  The key kernel
// is bound by instruction issue (260 instructions per suffix with run-time C, base and window shape: 2.0 ms
// at 1 GiB); this form needs about 90.
template <int C, int CW, int B>
__device__ __forceinline__ uint64_t key_and_window_dna(const uint8_t *img, uint32_t off, const pkey_cfg &kc,
                                                       uint32_t kbits)
{
    static_assert(B == 2 || B == 3, "two-bit codes (up to 4 symbols) or three-bit codes (5 ... 8 symbols)");
    static_assert(C >= 1 && CW >= 1 && CW * B <= 28 && C + CW <= 32, "window and key inside one 32-byte span");
    const uint32_t *w32 = reinterpret_cast<const uint32_t *>(img + (off & ~3u));
    const uint32_t sh = off & 3u;
    uint32_t raw[9], W[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) raw[k] = w32[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) W[k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], sh); // bytes 4k .. 4k+3 of the span
    W[8] = 0;
    // window: span bytes 0 .. CW-1, the farthest symbol first (wnd_from_bytes).  Two-bit codes: the weights 64, 16, 4,
    // 1 of a word's symbols fit a byte each, one dot product per word; three-bit codes: 512 does not, two symbols
    // (weights 8, 1) per dot product.
    uint32_t a = 0;
#pragma unroll
    for (int k = 0; k < (CW + 3) / 4; ++k) {
        constexpr uint32_t wcoef = (1u << 24) | (1u << (16 + 2)) | (1u << (8 + 4)) | (1u << 6);
        const int nsym = CW - 4 * k < 4 ? CW - 4 * k : 4; // (static: the loop is unrolled)
        if (B == 2) {
            a = __builtin_amdgcn_udot4(W[k], wcoef >> ((8 * (4 - nsym)) & 31), a << (nsym * 2), false);
        } else {
            const int nh = nsym < 2 ? nsym : 2, nl = nsym - nh;
            a = __builtin_amdgcn_udot4(W[k], nh == 2 ? 0x00000108u : 0x00000001u, a << (3 * nh), false);
            if (nl) a = __builtin_amdgcn_udot4(W[k], nl == 2 ? 0x01080000u : 0x00010000u, a << (3 * nl), false);
        }
    }
    constexpr uint32_t bias = ((1u << (B * CW)) - 1u) / ((1u << B) - 1u); // a one in each code field
    const uint32_t wnd = ((a - bias) << kCntBits) | (uint32_t)CW;
    // key: span bytes CW .. CW + C
    constexpr int NW = (C + 3) / 4, R = C % 4, d0 = CW / 4, sb = 8 * (CW % 4);
    static_assert(d0 + NW <= 8, "the key's words end inside the span");
    uint64_t acc = 0;
    uint32_t g = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const uint32_t kw = sb ? (W[d0 + k] >> sb) | (W[d0 + k + 1] << ((32 - sb) & 31)) : W[d0 + k];
        if (k < C / 4) {
            g = __builtin_amdgcn_udot4(kw, kc.coef4, k % 3 ? __umul24(g, kc.B4) : 0u, false);
            if (k % 3 == 2) {
                acc = k == 2 ? (uint64_t)g : acc * kc.B12 + g;
                g = 0;
            }
        } else {
            g = __builtin_amdgcn_udot4(kw, kc.coef4 >> ((8 * (4 - R)) & 31), __umul24(g, kc.Br), false);
        }
    }
    if ((C / 4) % 3 || R) acc = acc * kc.powT + g;
    return acc | (uint64_t)wnd << kbits;
}

// ---- dense keys for texts of four symbols (A C G T: the symbols 1 .. 4, the sentinel 0) ------------------------------
// The base-5 key leaves a fifth of the key space per symbol unused (no symbol is 0 before the end of the text), and
// inside a sub-bucket of the hybrid sort -- a range of 2^16 key values -- the keys that do occur sit in clumps: the
// local sort's bins fill unevenly, and a wave's ranking inside the bins takes as many steps as its fullest bin
// (8 - 10 on random DNA; 3.36 ms at 1 GiB, 2.83 with the steps capped at 5).  Two bits a symbol use every key value:
//     key = (sum over the C symbols of (symbol - 1) * 4^(C - 1 - i)) << lenbits | (symbols before the text's end)
// A suffix that runs into the sentinel counts it and the padding behind it as the smallest symbol, and the length
// field puts it in front of every suffix that has real symbols there: the order of the base-5 keys, and two suffixes
// have equal keys exactly when they did (same C symbols, none at the end of the text).  The sum comes from the same
// dot products with base 4 (symbols as they are: sum(symbol * 4^k), a constant too large), minus the constant, plus
// what the z = C - len zeros at the end took too much.
__device__ __forceinline__ uint64_t dense4_finish(uint64_t raw, uint32_t C, uint32_t z, uint32_t lenbits)
{
    const uint64_t ones = 0x5555555555555555ull; // 4^k summed: 0b...010101
    const uint64_t k0 = ones & ((1ull << (2u * C)) - 1ull), corr = z ? ones & ((1ull << (2u * z)) - 1ull) : 0ull;
    return ((raw - k0 + corr) << lenbits) | (uint64_t)(C - z);
}
__device__ __forceinline__ uint32_t dense4_zeros(uint64_t p, uint32_t C, uint64_t n) { return p + C > n ? (uint32_t)(p + C - n) : 0u; }

// The (key, window) word of LMS suffix p of a DNA-like text in the static form (C = CS key symbols, a window of WS codes of
// BS bits): img holds text[origin, ...) around p (16 bytes in front of the tile, 48 behind), T the whole text for the few
// suffixes at its very start.  dense_n: n + 1 for the dense four-letter keys (dense4_finish), else 0.
template <int CS, int WS, int BS>
__device__ __forceinline__ uint64_t lms_key_static(const uint8_t *img, uint64_t origin, uint32_t p, const uint8_t *__restrict__ T,
                                                   const pkey_cfg &kc, uint32_t kbits, const wnd_cfg &wcfg, uint64_t dense_n,
                                                   uint32_t lenbits)
{
    uint64_t key;
    if (p >= (uint32_t)WS) { // (everywhere but at the very start of the text)
        key = key_and_window_dna<CS, WS, BS>(img, (uint32_t)((uint64_t)(p - (uint32_t)WS) - origin), kc, kbits);
        if (dense_n) { // (uniform) the key bits again, dense; the window above them stays
            const uint64_t kmask = (1ull << kbits) - 1ull;
            key = (key & ~kmask) | dense4_finish(key & kmask /* the sum is below 4^C * 4/3 < 2^(2C+1) <= 2^kbits */, kc.C, dense4_zeros(p, kc.C, dense_n - 1), lenbits);
        }
    } else if (dense_n) {
        uint64_t raw = 0;
        for (uint32_t s2 = 0; s2 < kc.C; ++s2) raw = raw * 4u + (uint64_t)T[(uint64_t)p + s2];
        key = dense4_finish(raw, kc.C, dense4_zeros(p, kc.C, dense_n - 1), lenbits);
        if (wcfg.CW) key |= (uint64_t)wnd_fill<uint32_t>(T, p, wcfg) << kbits;
    } else {
        // the first few positions of the text, in the static forms: symbol by symbol from memory.  (With the general
        // form compiled in here, the compiler evaluated its 64 uniform tests once per workgroup and parked them
        // in a register's lanes -- 130 instructions up front for a path that a handful of suffixes of the whole text take.)
        key = 0;
        for (uint32_t s = 0; s < kc.C; ++s) key = key * kc.base + (uint64_t)T[(uint64_t)p + s];
        if (wcfg.CW) key |= (uint64_t)wnd_fill<uint32_t>(T, p, wcfg) << kbits;
    }
    return key;
}

// What the first pass of the hybrid sort needs to compute these keys itself (sx_radix.hip: radix_scatter_lms_kernel): the
// LMS suffixes in text order are the set bits of the classification's bit array, tile_off[t] the number of them in front of
// classification tile t.  shape: lms_key_shape(C, window symbols, code bits) of the static form; 0 = none.
struct sx_lmskey {
    const uint8_t *T;
    const uint16_t *lmsbits;
    const uint32_t *tile_off;
    uint32_t cls_tiles;  // classification tiles (kClsTile positions each)
    uint32_t m;          // LMS suffixes in all
    pkey_cfg kc;
    wnd_cfg wcfg;
    uint32_t kbits, lenbits;
    uint64_t dense_n;
    uint32_t shape;
};
constexpr uint32_t lms_key_shape(uint32_t C, uint32_t W, uint32_t B) { return (C * 16u + W) * 4u + B; }

} // namespace sx
