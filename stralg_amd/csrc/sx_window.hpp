// sx_window.hpp -- symbol windows carried by suffix-array entries during induction.
#pragma once
#include "sx_common.hpp"
#include "sx_device.hpp"

namespace sx {

// Every suffix-array entry p travels with a window word holding the symbols to
// its left, text[p-1], text[p-2], ... (codes = symbol - 1, B bits each, the
// nearest one in the lowest field) and, in the low 4 bits, how many are valid.
// Inducing p-1 from p pops one symbol; the text is touched again only when a
// window runs dry.
constexpr int kCntBits = 4;
struct wnd_cfg {
    uint32_t B;    // bits per symbol code
    uint32_t CW;   // symbols per window (<= 15)
    uint32_t mask; // (1 << B) - 1
};

template <class WT> __device__ __forceinline__ uint32_t wnd_count(WT w) { return (uint32_t)(w & (WT)15); }
template <class WT> __device__ __forceinline__ uint32_t wnd_first(WT w, const wnd_cfg &c)
{
    return (uint32_t)((w >> kCntBits) & (WT)c.mask) + 1u;
}
// the symbol in front of the entry, 0 for the one entry with nothing in front of it (position 0): its BWT symbol
template <class WT> __device__ __forceinline__ uint8_t wnd_symbol(WT w, const wnd_cfg &c)
{
    return (uint8_t)(wnd_count<WT>(w) ? wnd_first<WT>(w, c) : 0u);
}
template <class WT> __device__ __forceinline__ WT wnd_pop(WT w, const wnd_cfg &c)
{
    const WT cnt = w & (WT)15;
    return (((w >> kCntBits) >> c.B) << kCntBits) | (cnt - 1);
}
// the window of a position from the (up to 15) bytes in front of it: lo = bytes 0..7, hi = bytes 8..15 of
// text[p - cnt .. p)
template <class WT>
__device__ __forceinline__ WT wnd_from_bytes(uint64_t lo, uint64_t hi, uint32_t cnt, const wnd_cfg &c)
{
    WT acc = 0;
    if (cnt == c.CW && c.B <= 2) { // (uniform second test) four symbols per dot-product instruction
        // code fields of B <= 2 bits: the weights 2^(3B), 2^(2B), 2^B, 1 of a word's four symbols fit a byte each,
        // so a word of text is turned into 4 B window bits by one v_dot4_u32_u8; the -1 of every code is taken
        // off at the end as one constant.  (DNA: 14 symbols = 4 dot products instead of 14 extract-shift-or steps;
        // the key kernel of the LMS sort is bound by vector instructions.)
        const uint32_t w[4] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
        const uint32_t nw = c.CW >> 2, r = c.CW & 3u;
        const uint32_t coef4 = (1u << 24) | (1u << (16 + c.B)) | (1u << (8 + 2 * c.B)) | (1u << (3 * c.B));
        uint32_t a = 0, bias = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            if (j < nw) a = __builtin_amdgcn_udot4(w[j], coef4, a << (4 * c.B), false);
            else if (j == nw && r) a = __builtin_amdgcn_udot4(w[j], coef4 >> (8 * (4 - r)), a << (r * c.B), false);
        }
        for (uint32_t i = 0; i < c.CW; ++i) bias |= 1u << (c.B * i); // uniform
        return (WT)(((a - bias) << kCntBits) | cnt);
    }
    if (cnt == c.CW) { // everywhere but at the very start of the text: the loop bound is uniform
#pragma unroll
        for (uint32_t i = 0; i < 15; ++i) {
            if (i < c.CW) {
                const uint64_t byte = ((i < 8 ? lo : hi) >> (8u * (i & 7u))) & 0xFFull;
                acc = (acc << c.B) | (WT)(byte - 1u); // ends with text[p-1] in the lowest field
            }
        }
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 15; ++i) {
            if (i < cnt) {
                const uint64_t byte = ((i < 8 ? lo : hi) >> (8u * (i & 7u))) & 0xFFull;
                acc = (acc << c.B) | (WT)(byte - 1u);
            }
        }
    }
    return (acc << kCntBits) | (WT)cnt;
}

// window of position p, read from the text (p >= 1)
template <class WT>
__device__ __forceinline__ WT wnd_fill(const uint8_t *__restrict__ T, uint32_t p, const wnd_cfg &c)
{
    const uint32_t cnt = p < c.CW ? p : c.CW;
    uint64_t lo, hi;
    load_bytes16(T, (uint64_t)(p - cnt), lo, hi);
    return wnd_from_bytes<WT>(lo, hi, cnt, c);
}

// the same from a staged image of the text in LDS whose byte 0 is text[origin]
template <class WT>
__device__ __forceinline__ WT wnd_fill_lds(const uint8_t *img, uint64_t origin, uint32_t p, const wnd_cfg &c)
{
    const uint32_t cnt = p < c.CW ? p : c.CW;
    uint64_t lo, hi;
    lds_bytes16(img, (uint32_t)((uint64_t)(p - cnt) - origin), lo, hi);
    return wnd_from_bytes<WT>(lo, hi, cnt, c);
}

} // namespace sx

// window layout for a text whose largest symbol is maxc; returns true when windows are 64-bit
static inline bool sx_window_cfg(uint32_t maxc, sx::wnd_cfg &cfg)
{
    cfg.B = (uint32_t)sx_bitlen(maxc > 0 ? maxc - 1 : 0);
    if (cfg.B < 1) cfg.B = 1;
    cfg.mask = (1u << cfg.B) - 1u;
    const bool wide = cfg.B > 4;
    cfg.CW = ((wide ? 64u : 32u) - (uint32_t)sx::kCntBits) / cfg.B;
    if (cfg.CW > 15) cfg.CW = 15;
    return wide;
}
