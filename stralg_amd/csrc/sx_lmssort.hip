// sx_lmssort.hip -- LMS-suffix sort by fixed-length text prefixes (the fast path).
//
// Role of stralg/sa_is.c:340-387 (place_LMS + first induce round + reduce_SA +
// recursion): produce the LMS suffixes in sorted order.  On inputs whose LMS
// suffixes are told apart by their first few dozen symbols -- random DNA and
// byte strings are: the expected longest repeat is ~2 log_sigma n symbols --
// that order is simply a radix sort of (prefix key, position) pairs:
//
//   key(p)  = the first C symbols of suffix p as a base-(max symbol + 1) number; past
//             the end the text reads 0, the unique smallest symbol, so no key
//             containing the sentinel can tie with another.  C is the smallest
//             length at which random text leaves ~3 % of the suffixes tied
//             (17 symbols = 40 bits for 1 GiB of DNA, 5 symbols = 40 bits for bytes):
//             fewer key bits = fewer radix passes.
//   sort    one LSD radix sort (sx_radix.hip) of all m LMS suffixes
//   ties    groups of equal keys are refined by the next C symbols, a few
//             rounds, on the tied elements only.
//
// If too many ties remain (repetitive text) the caller falls back to the
// general path (pieces + names + prefix doubling, sx_reduce.hip), which is
// O(n log n) whatever the input.  Either way the order is the unique one.
#include <math.h>

#include <chrono>
#include <cstdlib>
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_scan.hpp"
#include "sx_internal.hpp"
#include "sx_window.hpp"
#include "sx_lmskey.hpp"

namespace sx {


// One workgroup per classification tile (4096 text positions): the tile's LMS positions are
// listed in LDS from the LMS bit array, then every thread turns listed positions into
// (key, position) pairs at the tile's offset in the global LMS order.  This is the compaction of
// the LMS positions (role of sa_is.c:203-218 place_LMS's scan) and the key generation in one pass.
// CS > 0: kc.C == CS, kc.dot, and the windows are WS codes of BS bits (the host checks).
template <int CS, int WS, int BS>
__global__ __launch_bounds__(kBlock) void lms_tile_keys_kernel(const uint8_t *__restrict__ T,
                                                               const uint16_t *__restrict__ lmsbits,
                                                               const uint32_t *__restrict__ tile_off, pkey_cfg kc,
                                                               uint32_t kbits, wnd_cfg wcfg,
                                                               uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                               uint8_t *__restrict__ dig0, uint32_t dig_shift,
                                                               uint32_t dig_mask, uint64_t dense_n /* dense keys: n + 1, else 0 */,
                                                               uint32_t lenbits)
{
    __shared__ uint32_t lds[kWavesPerBlock];
    __shared__ uint32_t spos[kClsTile / 2 + 1]; // LMS positions are at least two apart
    // The text around the tile, staged once with coalesced loads: text[tile0 - 16 .. tile0 + 4096 + 48).  A suffix
    // then takes its key symbols and its window from LDS; from memory every suffix cost five dependent 16-byte
    // loads and the kernel ran at the latency of those.
    __shared__ __attribute__((aligned(16))) uint8_t img[kClsTile + 64];
    const int t = (int)threadIdx.x;
    const uint64_t tile0 = (uint64_t)blockIdx.x * kClsTile;
    const uint64_t origin = tile0 - 16; // (wraps for tile 0: that chunk is zero-filled, nothing reads it)
    for (uint32_t q = (uint32_t)t; q < (kClsTile + 64) / 16; q += kBlock) {
        uint4 v = {0, 0, 0, 0};
        if (tile0 + 16ull * q >= 16) v = *reinterpret_cast<const uint4 *>(T + tile0 + 16ull * q - 16);
        *reinterpret_cast<uint4 *>(img + 16 * q) = v;
    }
    const uint64_t p0 = tile0 + (uint64_t)t * kClsPerThread;
    uint32_t mask = lmsbits[(uint64_t)blockIdx.x * kBlock + t];
    uint32_t total;
    uint32_t at = block_exclusive_scan<OpAdd>((uint32_t)__popc(mask), lds, total);
    while (mask) {
        const int i = __ffs(mask) - 1;
        mask &= mask - 1u;
        spos[at++] = (uint32_t)(p0 + i);
    }
    __syncthreads();
    const uint32_t dst0 = tile_off[blockIdx.x];
    for (uint32_t i = (uint32_t)t; i < total; i += kBlock) {
        const uint32_t p = spos[i];
        uint64_t key;
        if (CS > 0) {
            key = lms_key_static<(CS > 0 ? CS : 1), (CS > 0 ? WS : 1), (CS > 0 ? BS : 2)>(img, origin, p, T, kc, kbits, wcfg, dense_n, lenbits);
        } else {
            if (kc.C <= 32) { // uniform
                uint64_t q[4];
                lds_bytes32(img, (uint32_t)((uint64_t)p - origin), q);
                key = prefix_key_of(q, kc);
            } else {
                key = prefix_key(T, p, kc);
            }
            if (dense_n) key = dense4_finish(key, kc.C, dense4_zeros(p, kc.C, dense_n - 1), lenbits); // (uniform)
            // The key bits above kbits are not sorted on, they just ride along: put the suffix's
            // symbol window (text[p-1], text[p-2], ... for the induction, sx_window.hpp) there while
            // this part of the text is at hand, instead of gathering it again after the sort.
            if (wcfg.CW) key |= (uint64_t)wnd_fill_lds<uint32_t>(img, origin, p, wcfg) << kbits;
        }
        // written once, next read by another kernel: streaming stores (the sort's first pass gained 4 %)
        __builtin_nontemporal_store(key, keys + dst0 + i);
        __builtin_nontemporal_store(p, vals + dst0 + i);
        // the first radix pass's digit (its histogram reads this byte, not the key), without payload bits
        __builtin_nontemporal_store((uint8_t)((uint32_t)(key >> dig_shift) & dig_mask), dig0 + dst0 + i);
    }
}

// Keys of ALL suffixes (the direct sort of wide alphabets, see sx_sort_by_prefix): position p gets the key of
// text[p .. p+C); the zero padding behind the sentinel reads as more sentinel digits, so suffixes that reach
// the end of the text are ordered correctly and the key of position n is the smallest.
__global__ __launch_bounds__(kBlock) void all_keys_kernel(const uint8_t *__restrict__ T, uint64_t N, pkey_cfg kc,
                                                          uint32_t kbits, wnd_cfg wcfg, uint64_t *__restrict__ keys,
                                                          uint8_t *__restrict__ dig0, uint32_t dig_shift, uint32_t dig_mask)
{
    const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= N) return;
    uint64_t key = prefix_key(T, p, kc);
    // the symbol before the suffix rides in the unsorted key bits (a one-symbol window): after the sort it is the BWT
    if (wcfg.CW) key |= (uint64_t)wnd_fill<uint32_t>(T, (uint32_t)p, wcfg) << kbits;
    keys[p] = key; // (the position is the index: the sort's first pass fills the values in)
    dig0[p] = (uint8_t)((uint32_t)(key >> dig_shift) & dig_mask);
}

// The same for keys of at most 12 symbols (every alphabet that qualifies for the direct sort): a thread takes 16
// consecutive positions from three aligned 16-byte loads, so the symbols of all 16 keys are static byte picks
// from registers, and writes its keys and positions with 16-byte stores.
template <int G>
__device__ __forceinline__ uint64_t key_from_words(const uint32_t (&w)[12], int first, const pkey_cfg &kc)
{
    uint64_t acc = 0;
    uint32_t g = 0;
#pragma unroll
    for (int s = 0; s < 12; ++s) {
        if ((uint32_t)s < kc.C) { // uniform
            const int b = first + s;
            g = __umul24(g, kc.base) + ((w[b >> 2] >> (8 * (b & 3))) & 0xFFu);
            if ((s + 1) % G == 0) {
                acc = acc * kc.powG + g;
                g = 0;
            }
        }
    }
    if (kc.C % G) acc = acc * kc.powR + g; // uniform
    return acc;
}

template <int G>
__global__ __launch_bounds__(kBlock) void all_keys16_kernel(const uint8_t *__restrict__ T, uint64_t N, pkey_cfg kc,
                                                            uint32_t kbits, wnd_cfg wcfg, uint64_t *__restrict__ keys,
                                                            uint8_t *__restrict__ dig0, uint32_t dig_shift, uint32_t dig_mask)
{
    const uint64_t p0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 16u;
    if (p0 >= N) return;
    uint32_t w[12]; // text[p0 .. p0 + 48): T is the padded copy, aligned and readable past N
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const uint4 v = *reinterpret_cast<const uint4 *>(T + p0 + 16 * q);
        w[4 * q] = v.x, w[4 * q + 1] = v.y, w[4 * q + 2] = v.z, w[4 * q + 3] = v.w;
    }
    uint32_t before = p0 ? (uint32_t)T[p0 - 1] : 0u; // the symbol in front of the thread's first suffix
    uint64_t key[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        key[i] = key_from_words<G>(w, i, kc);
        // the symbol before the suffix as a one-symbol window in the unsorted key bits (it becomes the BWT)
        if (wcfg.CW && before) key[i] |= (uint64_t)(((before - 1u) << kCntBits) | 1u) << kbits;
        before = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
    }
    if (p0 + 16 <= N && (((uintptr_t)keys | (uintptr_t)dig0) & 15u) == 0) { // (positions = indices: the first pass fills them in)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            uint4 v;
            v.x = (uint32_t)key[2 * q], v.y = (uint32_t)(key[2 * q] >> 32);
            v.z = (uint32_t)key[2 * q + 1], v.w = (uint32_t)(key[2 * q + 1] >> 32);
            *reinterpret_cast<uint4 *>(keys + p0 + 2 * q) = v;
        }
        uint32_t d[4] = {0, 0, 0, 0}; // the first radix pass's digits of the 16 keys
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i >> 2] |= ((uint32_t)(key[i] >> dig_shift) & dig_mask) << (8 * (i & 3)); // (without payload bits)
        uint4 v;
        v.x = d[0], v.y = d[1], v.z = d[2], v.w = d[3];
        *reinterpret_cast<uint4 *>(dig0 + p0) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (p0 + i < N) {
                keys[p0 + i] = key[i];
                dig0[p0 + i] = (uint8_t)((uint32_t)(key[i] >> dig_shift) & dig_mask);
            }
    }
}

// Members of groups of equal keys -> (index in the sorted order, text position, group head flag), compacted in
// order; the window of every sorted slot is lifted out of the key's payload bits on the way.  Two launches and a
// small scan, no dependence between tiles: `tied_mark` reads the sorted keys once (a thread takes 8 consecutive
// keys, 16-byte loads, and their two neighbours), writes the windows, the tile's count and the tile's members,
// compacted inside the tile, to a staging area (the sort's spare ping-pong buffers, free by now); `tied_gather`
// moves every tile's few members to their place in the global list.  (A single chained launch with decoupled
// look-back read no more bytes but took 1.8 ms against 1.2: with thousands of tiles in flight a look-back walks
// over most of them, and a hop to another XCD's status word costs microseconds.)
constexpr int kTiedItems = 8, kTiedSub = 4;
constexpr int kTiedSubTile = kBlock * kTiedItems, kTiedTile = kTiedSubTile * kTiedSub;
__global__ __launch_bounds__(kBlock) void tied_mark_kernel(const uint64_t *__restrict__ ks, const uint32_t *__restrict__ vs,
                                                           uint64_t m, uint64_t kmask, uint32_t kbits,
                                                           uint32_t *__restrict__ seedw, uint32_t *__restrict__ tile_count,
                                                           uint2 *__restrict__ stage /* (sorted index, position) */,
                                                           uint8_t *__restrict__ stage_head)
{
    __shared__ uint64_t lds[kWavesPerBlock];
    const int t = (int)threadIdx.x;
    const uint32_t tile = blockIdx.x;
    uint32_t fmask = 0, hmask = 0; // tied / group head, bit 8 s + i = key i of this thread in sub-tile s
    uint64_t counts = 0;           // tied keys of this thread per sub-tile, 16-bit fields
#pragma unroll
    for (int sub = 0; sub < kTiedSub; ++sub) {
        const uint64_t j0 = (uint64_t)tile * kTiedTile + (uint64_t)sub * kTiedSubTile + (uint64_t)t * kTiedItems;
        uint64_t raw[kTiedItems];
        if (j0 + kTiedItems <= m && ((uintptr_t)ks & 15u) == 0) {
#pragma unroll
            for (int q = 0; q < kTiedItems / 2; ++q) {
                const uint4 v = *reinterpret_cast<const uint4 *>(ks + j0 + 2 * q);
                raw[2 * q] = pack64(v.x, v.y);
                raw[2 * q + 1] = pack64(v.z, v.w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < kTiedItems; ++i) raw[i] = j0 + i < m ? ks[j0 + i] : 0ull;
        }
        const bool has_prev = j0 > 0 && j0 <= m, has_next = j0 + kTiedItems < m;
        const uint64_t kprev = has_prev ? ks[j0 - 1] & kmask : 0ull, knext = has_next ? ks[j0 + kTiedItems] & kmask : 0ull;
        uint32_t f = 0, h = 0;
#pragma unroll
        for (int i = 0; i < kTiedItems; ++i) {
            const uint64_t k = raw[i] & kmask;
            const bool valid = j0 + i < m;
            const bool eq_prev = valid && (i > 0 ? (raw[i > 0 ? i - 1 : 0] & kmask) == k : (has_prev && kprev == k));
            const bool eq_next =
                valid && (i + 1 < kTiedItems ? (j0 + i + 1 < m && (raw[i + 1 < kTiedItems ? i + 1 : i] & kmask) == k)
                                             : (has_next && knext == k));
            if (eq_prev || eq_next) f |= 1u << i;
            if (!eq_prev) h |= 1u << i;
        }
        fmask |= f << (8 * sub);
        hmask |= h << (8 * sub);
        counts |= (uint64_t)__popc(f) << (16 * sub);
        if (seedw) {
            if (j0 + kTiedItems <= m && ((uintptr_t)seedw & 15u) == 0) {
#pragma unroll
                for (int q = 0; q < kTiedItems / 4; ++q) {
                    uint4 v;
                    v.x = (uint32_t)(raw[4 * q] >> kbits), v.y = (uint32_t)(raw[4 * q + 1] >> kbits);
                    v.z = (uint32_t)(raw[4 * q + 2] >> kbits), v.w = (uint32_t)(raw[4 * q + 3] >> kbits);
                    *reinterpret_cast<uint4 *>(seedw + j0 + 4 * q) = v;
                }
            } else {
#pragma unroll
                for (int i = 0; i < kTiedItems; ++i)
                    if (j0 + i < m) seedw[j0 + i] = (uint32_t)(raw[i] >> kbits);
            }
        }
    }
    // tied keys before this thread inside each sub-tile (one packed scan), and per sub-tile in all
    uint64_t tot;
    const uint64_t ex = block_exclusive_sum64(counts, lds, tot);
    if (t == 0) {
        uint32_t tile_total = 0;
#pragma unroll
        for (int sub = 0; sub < kTiedSub; ++sub) tile_total += (uint32_t)(tot >> (16 * sub)) & 0xFFFFu;
        tile_count[tile] = tile_total;
    }
    if (fmask) {
        uint32_t sub_base = 0;
#pragma unroll
        for (int sub = 0; sub < kTiedSub; ++sub) {
            const uint64_t j0 = (uint64_t)tile * kTiedTile + (uint64_t)sub * kTiedSubTile + (uint64_t)t * kTiedItems;
            uint32_t slot = sub_base + ((uint32_t)(ex >> (16 * sub)) & 0xFFFFu);
#pragma unroll
            for (int i = 0; i < kTiedItems; ++i) {
                if ((fmask >> (8 * sub + i)) & 1u) {
                    const uint64_t at = (uint64_t)tile * kTiedTile + slot;
                    uint2 e;
                    e.x = (uint32_t)(j0 + i), e.y = vs[j0 + i];
                    stage[at] = e;
                    stage_head[at] = (uint8_t)((hmask >> (8 * sub + i)) & 1u);
                    ++slot;
                }
            }
            sub_base += (uint32_t)(tot >> (16 * sub)) & 0xFFFFu;
        }
    }
}

__global__ __launch_bounds__(kBlock) void tied_gather_kernel(const uint32_t *__restrict__ tile_count,
                                                             const uint32_t *__restrict__ tile_off,
                                                             const uint2 *__restrict__ stage,
                                                             const uint8_t *__restrict__ stage_head,
                                                             uint32_t *__restrict__ apos, uint32_t *__restrict__ ap,
                                                             uint8_t *__restrict__ ahead, uint32_t cap)
{
    const uint32_t tile = blockIdx.x, cnt = tile_count[tile], off = tile_off[tile];
    for (uint32_t i = threadIdx.x; i < cnt; i += kBlock) {
        const uint32_t slot = off + i;
        if (slot >= cap) break;
        const uint2 e = stage[(uint64_t)tile * kTiedTile + i];
        apos[slot] = e.x;
        ap[slot] = e.y;
        ahead[slot] = stage_head[(uint64_t)tile * kTiedTile + i];
    }
}

// group id of an active element = index (in the active list) of its group's first member
struct InActHead {
    const uint8_t *ahead;
    __device__ __forceinline__ uint32_t operator()(uint64_t t) const { return ahead[t] ? (uint32_t)t + 1u : 0u; }
};
// ... and, at its head's slot, the group's size (written by the group's last member)
struct OutGroupIds {
    const uint8_t *ahead;
    uint64_t A;
    uint32_t *agid, *gsize;
    __device__ __forceinline__ void operator()(uint64_t t, uint32_t excl, uint32_t v) const
    {
        const uint32_t g = (excl > v ? excl : v) - 1u;
        agid[t] = g;
        if (t + 1 == A || ahead[t + 1]) gsize[g] = (uint32_t)(t + 1) - g;
    }
};
// the sub-list's groups numbered 0, 1, ... (few long groups: the sort by group then takes one pass, not one per byte of a list index)
struct InSubHead {
    const uint32_t *sub_t, *sub_gid;
    __device__ __forceinline__ uint32_t operator()(uint64_t j) const { return sub_t[j] == sub_gid[j] ? 1u : 0u; }
};
struct OutGroupNumber {
    uint32_t *number;
    __device__ __forceinline__ void operator()(uint64_t j, uint32_t excl, uint32_t v) const { number[j] = excl + v - 1u; }
};
// the members of groups with more than `limit` members, compacted (list order: groups stay whole and in order), with the
// next C symbols of each as its key
struct InLargeGroup {
    const uint32_t *agid, *gsize;
    uint32_t limit;
    __device__ __forceinline__ uint32_t operator()(uint64_t t) const { return gsize[agid[t]] > limit ? 1u : 0u; }
};
struct OutLargeMember {
    const uint8_t *T;
    const uint32_t *ap, *agid;
    uint64_t n, skip;
    pkey_cfg kc;
    uint32_t *sub_t, *sub_gid;
    uint64_t *key_keep, *key_sort;
    uint32_t *order;
    __device__ __forceinline__ void operator()(uint64_t t, uint32_t excl, uint32_t v) const
    {
        if (!v) return;
        sub_t[excl] = (uint32_t)t;
        sub_gid[excl] = agid[t];
        const uint64_t q = (uint64_t)ap[t] + skip;
        // q > n cannot happen inside a tie (a key holding the sentinel is unique); stay in bounds anyway
        const uint64_t k = q <= n ? prefix_key(T, q, kc) : 0ull;
        key_keep[excl] = k;
        key_sort[excl] = k;
        order[excl] = excl;
    }
};

__global__ __launch_bounds__(kBlock) void gid_keys_kernel(const uint32_t *__restrict__ order,
                                                          const uint32_t *__restrict__ agid, uint64_t A,
                                                          uint64_t *__restrict__ keys)
{
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t < A) keys[t] = agid[order[t]];
}

// after ordering the sub-list by (group, next key): write the refined order back, find the new boundaries.  Slot j of the
// sub-list is slot sub_t[j] of the active list; a group's members occupy the same slots before and after.
__global__ __launch_bounds__(kBlock) void refine_write_kernel(const uint32_t *__restrict__ order,
                                                              const uint32_t *__restrict__ sub_t,
                                                              const uint32_t *__restrict__ ap,
                                                              const uint32_t *__restrict__ apos,
                                                              const uint32_t *__restrict__ sub_gid,
                                                              const uint64_t *__restrict__ key_keep, uint64_t A3,
                                                              uint32_t *__restrict__ vals_sorted,
                                                              uint32_t *__restrict__ ap_new,
                                                              uint8_t *__restrict__ head_new,
                                                              uint32_t *__restrict__ seedw,
                                                              const uint8_t *__restrict__ T, wnd_cfg wcfg)
{
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= A3) return;
    const uint32_t o = order[j];
    const uint32_t p = ap[sub_t[o]];
    const uint32_t t = sub_t[j];
    vals_sorted[apos[t]] = p; // slot t of the active list keeps its place in the sorted order
    if (seedw) seedw[apos[t]] = p ? wnd_fill<uint32_t>(T, p, wcfg) : 0u; // its window moves with it
    ap_new[t] = p;
    bool head = true;
    if (j > 0) {
        const uint32_t o1 = order[j - 1];
        head = sub_gid[o1] != sub_gid[o] || key_keep[o1] != key_keep[o];
    }
    head_new[t] = head ? 1 : 0;
}

// The same refinement for groups of at most kSmallGroup members (on random text all of them: mostly two suffixes
// sharing a prefix), by the thread of the group's head: next keys, a sorting network, write-back.  Larger groups are only
// counted; if there are any, the round is redone by the sorting path above (nothing this kernel reads is
// overwritten by it).
constexpr int kSmallGroup = 8;
__global__ __launch_bounds__(kBlock) void refine_small_groups_kernel(
    const uint8_t *__restrict__ T, uint64_t n, const uint32_t *__restrict__ ap, const uint32_t *__restrict__ apos,
    const uint8_t *__restrict__ head, uint64_t A, uint64_t skip, pkey_cfg kc, uint32_t *__restrict__ vals_sorted,
    uint32_t *__restrict__ ap_new, uint8_t *__restrict__ head_new, uint32_t *__restrict__ seedw, wnd_cfg wcfg,
    uint32_t *__restrict__ large)
{
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= A || !head[t]) return;
    int size = 1;
    while (size <= kSmallGroup && t + size < A && !head[t + size]) ++size;
    if (size > kSmallGroup) {
        atomicAdd(large, 1u);
        return;
    }
    uint32_t p[kSmallGroup];
    uint64_t k[kSmallGroup];
#pragma unroll
    for (int i = 0; i < kSmallGroup; ++i) {
        p[i] = 0;
        k[i] = ~0ull; // padding sorts last
        if (i < size) {
            p[i] = ap[t + i];
            const uint64_t q = (uint64_t)p[i] + skip;
            k[i] = q <= n ? prefix_key(T, q, kc) : 0ull;
        }
    }
#define SX_CSWAP(a, b)                                                                                                 \
    if (k[b] < k[a]) {                                                                                                 \
        const uint64_t tk = k[a];                                                                                      \
        k[a] = k[b], k[b] = tk;                                                                                        \
        const uint32_t tp = p[a];                                                                                      \
        p[a] = p[b], p[b] = tp;                                                                                        \
    }
    // Batcher's odd-even merge sort for 8 (19 exchanges); groups of at most 4 only need the first three layers
    SX_CSWAP(0, 1) SX_CSWAP(2, 3) SX_CSWAP(4, 5) SX_CSWAP(6, 7)
    SX_CSWAP(0, 2) SX_CSWAP(1, 3) SX_CSWAP(4, 6) SX_CSWAP(5, 7)
    SX_CSWAP(1, 2) SX_CSWAP(5, 6)
    if (size > 4) {
        SX_CSWAP(0, 4) SX_CSWAP(1, 5) SX_CSWAP(2, 6) SX_CSWAP(3, 7)
        SX_CSWAP(2, 4) SX_CSWAP(3, 5)
        SX_CSWAP(1, 2) SX_CSWAP(3, 4) SX_CSWAP(5, 6)
    }
#undef SX_CSWAP
#pragma unroll
    for (int i = 0; i < kSmallGroup; ++i) {
        if (i < size) {
            const uint32_t slot = apos[t + i];
            vals_sorted[slot] = p[i];
            if (seedw) seedw[slot] = p[i] ? wnd_fill<uint32_t>(T, p[i], wcfg) : 0u;
            ap_new[t + i] = p[i];
            head_new[t + i] = (i == 0 || k[i] != k[i > 0 ? i - 1 : 0]) ? 1 : 0;
        }
    }
}

// The same with a lane a member.  Above, only the lanes of group heads work (every other one on random text) and walk
// their members one after the other: 0.87 ms for 5.6 M tied suffixes, a twentieth of the random-read rate.  Here a wave
// takes a window of 64 list slots, windows start every kRsStride = 56 slots: a group of at most eight members whose head
// lies in the window's first 56 slots ends inside the window, so every such group is owned by exactly one wave.  Every
// owned member reads its own next key and window (all reads of the wave in flight together), finds its place among its
// mates through lane shifts (at most seven either way), and the member that belongs at slot j of the group writes there.
// Larger groups are only reported, as above.
constexpr int kRsStride = kWave - kSmallGroup;
__global__ __launch_bounds__(kBlock) void refine_small_groups_wave_kernel(
    const uint8_t *__restrict__ T, uint64_t n, const uint32_t *__restrict__ ap, const uint32_t *__restrict__ apos,
    const uint8_t *__restrict__ head, uint64_t A, uint64_t skip, pkey_cfg kc, uint32_t *__restrict__ vals_sorted,
    uint32_t *__restrict__ ap_new, uint8_t *__restrict__ head_new, uint32_t *__restrict__ seedw, wnd_cfg wcfg,
    uint32_t *__restrict__ large)
{
    const uint64_t base = ((uint64_t)blockIdx.x * kWavesPerBlock + (uint64_t)wave_id()) * kRsStride;
    if (base >= A) return; // (the whole wave)
    const int l = lane_id();
    const uint64_t t = base + (uint64_t)l;
    const bool valid = t < A;
    // a slot opens a group when its head flag is set; the slot after the last one closes the list
    const uint64_t heads = __ballot(((valid && head[t]) || t == A) ? 1 : 0);
    const uint64_t upto = heads & lanemask_le();
    const int first = upto ? 63 - __clzll((unsigned long long)upto) : -1; // lane of the group's head (-1: before the window)
    const uint64_t later = heads & ~lanemask_le();
    const int my_end = later ? __ffsll((unsigned long long)later) - 1 : kWave + kSmallGroup; // lane after the group's last member (unknown: too far)
    const bool mine = valid && first >= 0 && first < kRsStride; // the group's head lies in this wave's stride
    const bool small = mine && my_end - first <= kSmallGroup;
    if (__any((mine && !small && l == first) ? 1 : 0) && l == 0) atomicAdd(large, 1u);
    const uint32_t p = small ? ap[t] : 0u;
    const uint32_t slot_here = small ? apos[t] : 0u;
    uint64_t key = 0;
    if (small) {
        const uint64_t q = (uint64_t)p + skip;
        key = q <= n ? prefix_key(T, q, kc) : 0ull;
    }
    const uint32_t wnd = (small && seedw && p) ? wnd_fill<uint32_t>(T, p, wcfg) : 0u;
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    const int fl = small ? first : -1 - l; // (lanes outside owned groups: a group of their own, never a mate)
    uint32_t less = 0, same_left = 0;
#pragma unroll
    for (int o = 1; o < kSmallGroup; ++o) {
        const int fu = __shfl_up(fl, o, kWave), fd = __shfl_down(fl, o, kWave);
        const uint64_t ku = pack64(__shfl_up(klo, o, kWave), __shfl_up(khi, o, kWave));
        const uint64_t kd = pack64(__shfl_down(klo, o, kWave), __shfl_down(khi, o, kWave));
        const bool mate_u = l >= o && fu == fl, mate_d = l + o < kWave && fd == fl;
        less += (mate_u && ku < key) ? 1u : 0u;
        same_left += (mate_u && ku == key) ? 1u : 0u;
        less += (mate_d && kd < key) ? 1u : 0u;
    }
    const int at = small ? first + (int)(less + same_left) : l; // the lane whose slot this member moves to
    const uint32_t slot_at = __shfl(slot_here, at, kWave);
    if (small) {
        vals_sorted[slot_at] = p;
        if (seedw) seedw[slot_at] = wnd;
        ap_new[base + (uint64_t)at] = p;
        head_new[base + (uint64_t)at] = same_left == 0 ? 1 : 0; // first of its run of equal keys (the group's first member included)
    }
}

// Groups of kSmallGroup + 1 .. kMidGroup members (a family of diverged repeats leaves tens of millions of suffixes in
// groups of some hundred to some thousand: through the radix passes every one of them crossed HBM 12 times a round).
// A workgroup takes the groups whose head lies in its span of the active list -- such a group ends inside the
// workgroup's reach -- computes the members' next keys, orders them by (group, key) in LDS and writes the round's
// result for them.  The order comes from a bitonic network over the members (padded to a power of two): a 63-bit key
// and the group's number would take nine stable 8-bit passes of ballot ranking, ~1200 issue cycles per pass and 64
// members; the network's 55 .. 78 compare-exchange steps cost a fifth of that for the some hundred members a
// workgroup typically owns.  (Members with equal keys stay tied whatever their order.)
constexpr int kRmThreads = 512, kRmWaves = kRmThreads / kWave, kRmItems = 8;
constexpr int kRmCap = kRmThreads * kRmItems; // 4096 list slots in reach (48 KiB of LDS, three workgroups a CU)
#ifndef SX_RM_SPAN
#define SX_RM_SPAN 2048
#endif
constexpr int kRmSpan = SX_RM_SPAN;           // a workgroup owns the groups whose head lies in its span
constexpr int kMidGroup = kRmCap - kRmSpan;   // 2048: a group this long that starts in the span ends within the reach
constexpr int kRmGroups = 256;                // owned groups have more than kSmallGroup members: fewer than 2048 / 9
static_assert(kRmSpan / (kSmallGroup + 1) < kRmGroups && (kRmCap & (kRmCap - 1)) == 0, "group numbers; the network's size");
// (group, key, slot) of one member before that of another
__device__ __forceinline__ bool rm_before(uint64_t ka, uint32_t va, uint64_t kb, uint32_t vb)
{
    const uint32_t ga = va >> 16, gb = vb >> 16;
    return ga != gb ? ga < gb : (ka != kb ? ka < kb : va < vb);
}
__global__ __launch_bounds__(kRmThreads, 4) void refine_mid_groups_kernel(
    const uint8_t *__restrict__ T, uint64_t n, const uint32_t *__restrict__ ap, const uint32_t *__restrict__ apos,
    const uint32_t *__restrict__ agid, const uint32_t *__restrict__ gsize, uint64_t A, uint64_t skip, pkey_cfg kc,
    uint32_t *__restrict__ vals_sorted, uint32_t *__restrict__ ap_new, uint8_t *__restrict__ head_new,
    uint32_t *__restrict__ seedw, wnd_cfg wcfg)
{
    __shared__ uint64_t K[kRmCap]; // next keys of the owned members, dense, in list order
    __shared__ uint32_t V[kRmCap]; // local list slot the member came from | its group's number << 16
    __shared__ uint32_t gfirst[kRmGroups], gstart[kRmGroups]; // a group's first dense index / its head's local list slot
    __shared__ uint32_t s_act[kRmWaves], s_head[kRmWaves], s_diff;
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t g0 = (uint64_t)blockIdx.x * kRmSpan;
    const uint32_t avail = A - g0 < (uint64_t)kRmCap ? (uint32_t)(A - g0) : (uint32_t)kRmCap;
    const uint32_t i0 = (uint32_t)w * (kWave * kRmItems) + (uint32_t)lane; // slots i0 + 64 k: a wave's slots are consecutive
    // ---- which slots in reach are members of owned groups; their dense index and group number -----------------------
    uint32_t act_mask = 0, head_mask = 0, run_act = 0, run_head = 0;
    uint32_t dg[kRmItems]; // dense index within the wave | heads up to here within the wave << 16
    {
        // (every load of a step up front: a branch around the second one would make the gathers wait for each other)
        uint32_t a[kRmItems], sz[kRmItems];
#pragma unroll
        for (int k = 0; k < kRmItems; ++k) {
            const uint32_t i = i0 + (uint32_t)k * kWave;
            a[k] = i < avail ? agid[g0 + i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < kRmItems; ++k) {
            const bool own = a[k] >= g0 && a[k] - g0 < (uint64_t)kRmSpan; // (a head's index is at most its member's: below A)
            sz[k] = gsize[own ? a[k] : g0];
            if (!own) a[k] = 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < kRmItems; ++k) {
            const uint32_t i = i0 + (uint32_t)k * kWave;
            const bool act = a[k] != 0xFFFFFFFFu && sz[k] > (uint32_t)kSmallGroup && sz[k] <= (uint32_t)kMidGroup;
            const bool hd = act && a[k] == g0 + i;
            const uint64_t ba = __ballot(act ? 1 : 0), bh = __ballot(hd ? 1 : 0);
            dg[k] = (run_act + (uint32_t)__popcll(ba & lanemask_lt())) | ((run_head + (uint32_t)__popcll(bh & lanemask_le())) << 16);
            if (act) act_mask |= 1u << k;
            if (hd) head_mask |= 1u << k;
            run_act += (uint32_t)__popcll(ba);
            run_head += (uint32_t)__popcll(bh);
        }
    }
    if (lane == 0) s_act[w] = run_act, s_head[w] = run_head;
    if (t == 0) s_diff = 0;
    __syncthreads();
    uint32_t base_act = 0, base_head = 0, n_act = 0;
#pragma unroll
    for (int ww = 0; ww < kRmWaves; ++ww) {
        if (ww < w) base_act += s_act[ww], base_head += s_head[ww];
        n_act += s_act[ww];
    }
    if (n_act == 0) return; // (uniform)
#pragma unroll
    for (int k = 0; k < kRmItems; ++k) {
        if ((act_mask >> k) & 1u) {
            const uint32_t i = i0 + (uint32_t)k * kWave;
            const uint32_t d = base_act + (dg[k] & 0xFFFFu), g = base_head + (dg[k] >> 16) - 1u;
            V[d] = i | (g << 16);
            if ((head_mask >> k) & 1u) gfirst[g] = d, gstart[g] = i;
        }
    }
    // the network's size: the power of two that holds the members (two members a thread at least)
    uint32_t P = 2 * kRmThreads;
    while (P < n_act) P <<= 1;
    for (uint32_t d = n_act + (uint32_t)t; d < P; d += kRmThreads) K[d] = ~0ull, V[d] = 0xFFFFFFFFu; // padding sorts last
    __syncthreads();
    // ---- next keys of the members, dense: as many steps as hold them (most workgroups own some hundred) -------------
    const uint32_t per = (n_act + (uint32_t)kRmThreads - 1u) / (uint32_t)kRmThreads;
    for (uint32_t k0 = 0; k0 < per; k0 += 4) { // uniform
        uint32_t pp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t d = (uint32_t)t + (k0 + (uint32_t)j) * kRmThreads;
            pp[j] = d < n_act ? ap[g0 + (V[d] & 0xFFFFu)] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t d = (uint32_t)t + (k0 + (uint32_t)j) * kRmThreads;
            const uint64_t q = (uint64_t)pp[j] + skip;
            // q > n cannot happen inside a tie (a key holding the sentinel is unique); stay in bounds anyway
            if (d < n_act) K[d] = q <= n ? prefix_key(T, q, kc) : 0ull;
        }
    }
    __syncthreads();
    {
        bool differs = false; // from its group's head (no member does: every group stays as it is)
        for (uint32_t d = (uint32_t)t; d < n_act; d += kRmThreads) differs = differs || K[d] != K[gfirst[V[d] >> 16]];
        if (__any(differs ? 1 : 0) && lane == 0) s_diff = 1;
    }
    __syncthreads();
    // ---- bitonic network over (group, key, slot) -----------------------------------------------------------------
    if (s_diff) { // (uniform)
        for (uint32_t k = 2; k <= P; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t c = (uint32_t)t; c < P / 2; c += kRmThreads) {
                    const uint32_t i = ((c & ~(j - 1u)) << 1) | (c & (j - 1u)), l = i | j;
                    const uint64_t ki = K[i], kl = K[l];
                    const uint32_t vi = V[i], vl = V[l];
                    const bool up = (i & k) == 0u; // this block of k ascends
                    if (rm_before(kl, vl, ki, vi) == up) K[i] = kl, V[i] = vl, K[l] = ki, V[l] = vi;
                }
                __syncthreads();
            }
        }
    }
    // ---- out: dense slot d is member d - gfirst[g] of group g -----------------------------------------------------
    for (uint32_t d = (uint32_t)t; d < n_act; d += kRmThreads) {
        const uint64_t key = K[d];
        const uint32_t v = V[d], g = v >> 16, r = d - gfirst[g];
        const uint64_t tt = g0 + gstart[g] + r;
        const uint32_t p = ap[g0 + (v & 0xFFFFu)], slot = apos[tt];
        vals_sorted[slot] = p;
        if (seedw) seedw[slot] = p ? wnd_fill<uint32_t>(T, p, wcfg) : 0u;
        ap_new[tt] = p;
        head_new[tt] = (r == 0 || K[d - 1] != key) ? 1 : 0;
    }
}

// Groups that survive two rounds of key refinement are true repeats, and the next 17 symbols rarely end them
// (a 50 kb duplication would need 3000 rounds).  Small groups are therefore finished in one step by comparing
// the suffixes themselves from the symbols already known equal on, 16 bytes at a time; the suffixes of a text
// are distinct (the sentinel), so this settles every member.  Cost is the sum of the common prefixes.  The
// head's thread gives a comparison kSoloCompare symbols; a group with a longer one is taken over by the whole
// wave, every lane comparing 16 bytes of a 1024-byte stretch (a thread alone walks a 50 000-symbol duplication
// in 3000 dependent steps, 0.4 ms during which the rest of the chip has long finished).  A comparison that runs
// beyond kCompareCap symbols gives up and the build takes the general path.
constexpr uint64_t kCompareCap = 1ull << 22;
constexpr uint64_t kSoloCompare = 512;
__device__ __forceinline__ bool first_difference_less(uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1)
{
    // the first differing byte decides (little endian: lowest byte first)
    const uint64_t x = a0 != b0 ? a0 ^ b0 : a1 ^ b1, ua = a0 != b0 ? a0 : a1, ub = a0 != b0 ? b0 : b1;
    const int sh = (__ffsll((unsigned long long)x) - 1) & ~7;
    return ((ua >> sh) & 0xFFull) < ((ub >> sh) & 0xFFull);
}
// by one thread; sets `too_long` (and returns anything) when the suffixes agree on kSoloCompare symbols from `off`
__device__ __forceinline__ bool suffix_less_from(const uint8_t *__restrict__ T, uint32_t a, uint32_t b, uint64_t off, bool &too_long)
{
    for (uint64_t l = off; l - off < kSoloCompare; l += 16) {
        uint64_t a0, a1, b0, b1;
        load_bytes16(T, (uint64_t)a + l, a0, a1);
        load_bytes16(T, (uint64_t)b + l, b0, b1);
        if (a0 != b0 || a1 != b1) return first_difference_less(a0, a1, b0, b1);
    }
    too_long = true;
    return false;
}
// by the whole wave (a, b, off uniform; every lane must call).  The text is readable up to 16 bytes past the
// sentinel at n; the first difference lies at or before the earlier of the two sentinels.
__device__ __forceinline__ bool wave_suffix_less_from(const uint8_t *__restrict__ T, uint64_t n, uint32_t a, uint32_t b,
                                                      uint64_t off, uint32_t *__restrict__ hard)
{
    const uint64_t mine = (uint64_t)lane_id() * 16;
    for (uint64_t l = off;; l += (uint64_t)kWave * 16) {
        const uint64_t pa = (uint64_t)a + l + mine, pb = (uint64_t)b + l + mine;
        const bool in = pa <= n && pb <= n;
        uint64_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
        if (in) {
            load_bytes16(T, pa, a0, a1);
            load_bytes16(T, pb, b0, b1);
        }
        const uint64_t bal = __ballot((in && (a0 != b0 || a1 != b1)) ? 1 : 0);
        if (bal) { // uniform
            const int first = __ffsll((unsigned long long)bal) - 1;
            return first_difference_less(__shfl(a0, first, kWave), __shfl(a1, first, kWave), __shfl(b0, first, kWave),
                                         __shfl(b1, first, kWave));
        }
        if (l - off > kCompareCap || !__any(in ? 1 : 0)) { // uniform
            if (lane_id() == 0) atomicAdd(hard, 1u);
            return false;
        }
    }
}

#define SX_SORT8(LESS)                                                                                                 \
    SX_CSWAP(0, 1, LESS) SX_CSWAP(2, 3, LESS) SX_CSWAP(4, 5, LESS) SX_CSWAP(6, 7, LESS)                                \
    SX_CSWAP(0, 2, LESS) SX_CSWAP(1, 3, LESS) SX_CSWAP(4, 6, LESS) SX_CSWAP(5, 7, LESS)                                \
    SX_CSWAP(1, 2, LESS) SX_CSWAP(5, 6, LESS)                                                                          \
    if (size > 4) {                                                                                                    \
        SX_CSWAP(0, 4, LESS) SX_CSWAP(1, 5, LESS) SX_CSWAP(2, 6, LESS) SX_CSWAP(3, 7, LESS)                            \
        SX_CSWAP(2, 4, LESS) SX_CSWAP(3, 5, LESS)                                                                      \
        SX_CSWAP(1, 2, LESS) SX_CSWAP(3, 4, LESS) SX_CSWAP(5, 6, LESS)                                                 \
    }
#define SX_CSWAP(a, b, LESS)                                                                                           \
    if (p[b] != kPad && (p[a] == kPad || LESS(p[b], p[a]))) {                                                          \
        const uint32_t tp = p[a];                                                                                      \
        p[a] = p[b], p[b] = tp;                                                                                        \
    }
__global__ __launch_bounds__(kBlock) void refine_by_comparison_kernel(
    const uint8_t *__restrict__ T, uint64_t n, const uint32_t *__restrict__ ap, const uint32_t *__restrict__ apos,
    const uint8_t *__restrict__ head, uint64_t A, uint64_t skip, uint32_t *__restrict__ vals_sorted,
    uint32_t *__restrict__ ap_new, uint8_t *__restrict__ head_new, uint32_t *__restrict__ seedw, wnd_cfg wcfg,
    uint32_t *__restrict__ counters /* [0] groups beyond kSmallGroup (left as they are), [1] comparisons given up */)
{
    constexpr uint32_t kPad = 0xFFFFFFFFu;
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    int gsz = 0; // size of the group this thread is the head of
    if (t < A && head[t]) {
        gsz = 1;
        while (gsz <= kSmallGroup && t + gsz < A && !head[t + gsz]) ++gsz;
    }
    bool too_long = false;
    if (gsz > kSmallGroup) { // ap_new / head_new already hold this group unchanged (copied before the launch)
        atomicAdd(&counters[0], 1u);
    } else if (gsz > 0) {
        const int size = gsz;
        uint32_t p[kSmallGroup];
#pragma unroll
        for (int i = 0; i < kSmallGroup; ++i) p[i] = i < size ? ap[t + i] : kPad;
#define SX_LESS_SOLO(x, y) suffix_less_from(T, x, y, skip, too_long)
        SX_SORT8(SX_LESS_SOLO)
#undef SX_LESS_SOLO
        if (!too_long) {
#pragma unroll
            for (int i = 0; i < kSmallGroup; ++i) {
                if (i < size) {
                    const uint32_t slot = apos[t + i];
                    vals_sorted[slot] = p[i];
                    if (seedw) seedw[slot] = p[i] ? wnd_fill<uint32_t>(T, p[i], wcfg) : 0u;
                    ap_new[t + i] = p[i];
                    head_new[t + i] = 1; // every member is told apart
                }
            }
        }
    }
    // groups with a long comparison, one after the other, by the whole wave
    uint64_t todo = __ballot(too_long ? 1 : 0);
    while (todo) { // uniform
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const uint64_t tg = __shfl(t, src, kWave);
        const int size = __shfl(gsz, src, kWave);
        uint32_t p[kSmallGroup];
#pragma unroll
        for (int i = 0; i < kSmallGroup; ++i) p[i] = i < size ? ap[tg + i] : kPad; // (uniform addresses)
#define SX_LESS_WAVE(x, y) wave_suffix_less_from(T, n, x, y, skip, &counters[1])
        SX_SORT8(SX_LESS_WAVE)
#undef SX_LESS_WAVE
        const int i = lane_id();
        if (i < size) {
            uint32_t mine = p[0];
#pragma unroll
            for (int k = 1; k < kSmallGroup; ++k) mine = i == k ? p[k] : mine;
            const uint32_t slot = apos[tg + i];
            vals_sorted[slot] = mine;
            if (seedw) seedw[slot] = mine ? wnd_fill<uint32_t>(T, mine, wcfg) : 0u;
            ap_new[tg + i] = mine;
            head_new[tg + i] = 1;
        }
    }
}
#undef SX_CSWAP
#undef SX_SORT8

struct InStillTied {
    const uint8_t *head;
    uint64_t A;
    __device__ __forceinline__ uint32_t operator()(uint64_t t) const
    {
        const bool single = head[t] && (t + 1 == A || head[t + 1]);
        return single ? 0u : 1u;
    }
};
struct OutStillTied {
    const uint32_t *apos, *ap;
    const uint8_t *head;
    uint32_t *apos2, *ap2;
    uint8_t *head2;
    __device__ __forceinline__ void operator()(uint64_t t, uint32_t excl, uint32_t v) const
    {
        if (!v) return;
        apos2[excl] = apos[t];
        ap2[excl] = ap[t];
        head2[excl] = head[t];
    }
};


// ---- a look at a sample before a prefix sort of a wide-alphabet text ------------------------------------------------
// Natural-language-like texts (words, phrases, mark-up) tie most suffixes on any 63-bit prefix: the prefix-key sort
// fails its tie bound twice (40 bits, then the longest key) and the general path takes over -- after 5 + 8 radix passes
// over every LMS suffix, or 5 over every suffix where the direct sort was tried first (a 1 GiB word text: 60 of 244 ms).
// A suffix that is tied inside a sample is tied in the whole, so the tied share of a 1-in-16 sample under the *longest*
// key is a lower bound of what the second attempt would meet: above the bound, both attempts are skipped.  Only texts
// of more than 8 symbols are looked at (uniform DNA pays 1.5 ms for a look that never tells it anything).
constexpr uint32_t kSampleBins = 512; // (tied and valid counts: one read-back of 1024 words)
__global__ __launch_bounds__(kBlock) void sample_keys_kernel(const uint8_t *__restrict__ T, const uint16_t *__restrict__ lmsbits /* or null */,
                                                            uint64_t N, uint32_t step, uint32_t base, uint32_t C,
                                                            uint64_t threads, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    // one 16-position word in `step` a thread, one suffix of it: its first LMS position (all suffixes: its first
    // position); a word without one leaves the largest key, which the count of ties skips
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= threads) return;
    const uint64_t w = i * step;
    const uint32_t mask = w * 16u >= N ? 0u : (lmsbits ? lmsbits[w] : 1u);
    uint64_t key = ~0ull;
    const uint64_t p = w * 16u + (mask ? (uint32_t)__ffs(mask) - 1u : 0u);
    if (mask && p < N) {
        key = 0;
        for (uint32_t k = 0; k < C; ++k) {
            const uint64_t q = p + k;
            key = key * base + (q < N - 1 ? (uint64_t)T[q] : 0ull); // (the sentinel and what lies behind it: digit 0)
        }
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kBlock) void sample_ties_kernel(const uint64_t *__restrict__ ks, uint32_t S, uint32_t *__restrict__ bins)
{
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    const bool valid = j < S && ks[j] != ~0ull;
    const bool t = valid && ((j > 0 && ks[j - 1] == ks[j]) || (j + 1 < S && ks[j + 1] == ks[j]));
    const uint64_t bt = __ballot(t ? 1 : 0), bv = __ballot(valid ? 1 : 0);
    const uint32_t bin = (blockIdx.x * (uint32_t)kWavesPerBlock + (uint32_t)wave_id()) & (kSampleBins - 1u);
    if (lane_id() == 0) {
        if (bt) atomicAdd(&bins[bin], (uint32_t)__popcll(bt));
        if (bv) atomicAdd(&bins[kSampleBins + bin], (uint32_t)__popcll(bv));
    }
}
} // namespace sx

using namespace sx;

// tied share of a 1-in-16 sample of the LMS suffixes (all suffixes: of the positions) under the longest prefix key a 64-bit
// word holds; scratch from `am` (by value: handed back).  share < 0: no look was taken.
int sx_prefix_ties_sampled(sx_ctx *ctx, const sx_text_info &ti, sx_arena am, bool all_suffixes, double *share)
{
    *share = -1.0;
    const uint64_t m = all_suffixes ? ti.N : ti.m;
    const uint64_t look_from = ctx->sample_min >= 0 ? (uint64_t)ctx->sample_min : (1ull << 20);
    if (ti.maxc < 8 || m < look_from || ctx->prefix_symbols > 0) return 0; // (a forced prefix length: a test of that sort)
    // Symbols that are all equally frequent are not what words, phrases or mark-up look like: the chance that two
    // positions hold the same symbol, times the symbols in use, is 1 for uniform symbols (1 GiB of random bytes: 1.000,
    // which then keeps the millisecond the look costs), 1.7 for English text, 1.3 for proteins
    if (ctx->sample_min < 0) {
        double sum_p2 = 0.0;
        uint32_t used = 0;
        const double n_sym = (double)ti.N - 1.0;
        for (int c = 1; c < 256 && n_sym > 0; ++c) {
            if (!ti.h_all[c]) continue;
            const double pc = (double)ti.h_all[c] / n_sym;
            sum_p2 += pc * pc;
            ++used;
        }
        if (sum_p2 * (double)used < 1.1) return 0;
    }
    const uint32_t base = ti.maxc + 1;
    uint32_t Cmax = 0;
    for (double cap63 = 9.2e18, v = 1.0; v * base <= cap63 && Cmax < 63; v *= base) ++Cmax;
    if (Cmax < 1) Cmax = 1;
    uint64_t top = 1;
    for (uint32_t i = 0; i < Cmax; ++i) top *= base;
    const int kbits = sx_bitlen(top - 1);
    // one suffix a sampled word: LMS suffixes are thinned to about one in 19 (a word in four, its first LMS position),
    // positions 256-fold (short texts, which only tests send here: every word)
    const uint32_t step = m < (1ull << 22) ? 1 : all_suffixes ? 16 : 4;
    const uint64_t words = (ti.N + 15) / 16, threads = (words + step - 1) / step;
    const uint32_t cap = (uint32_t)threads;
    uint64_t *ka = am.take<uint64_t>(cap), *kb = am.take<uint64_t>(cap);
    uint32_t *va = am.take<uint32_t>(cap), *vb = am.take<uint32_t>(cap), *bins = am.take<uint32_t>(2 * kSampleBins);
    if (!ka || !kb || !va || !vb || !bins) return 0; // (no room: no look)
    SX_CHECK(hipMemsetAsync(bins, 0, 2 * kSampleBins * sizeof(uint32_t), ctx->stream));
    sx_launch(ctx, SX_KC_KEYS, threads * 40, sample_keys_kernel, dim3(sx_div_up(threads, kBlock)), dim3(kBlock), ti.T,
              (const uint16_t *)(all_suffixes ? nullptr : ti.lmsbits), ti.N, step, base, Cmax, threads, ka, va);
    int in_b = 0;
    SX_TRY(sx_sort_pairs(ctx, ka, va, kb, vb, cap, 0, kbits < 64 ? kbits + 1 : 64, &in_b)); // (+1: the empty slots' key sorts last)
    sx_launch(ctx, SX_KC_NAMES, (uint64_t)cap * 8, sample_ties_kernel, dim3(sx_div_up(cap, kBlock)), dim3(kBlock),
              (const uint64_t *)(in_b ? kb : ka), cap, bins);
    uint32_t h_bins[2 * kSampleBins];
    SX_TRY(sx_readback(ctx, bins, 2 * kSampleBins, h_bins));
    uint64_t tied = 0, S = 0;
    for (uint32_t i = 0; i < kSampleBins; ++i) tied += h_bins[i], S += h_bins[kSampleBins + i];
    if (S < 1024) return 0;
    *share = (double)tied / (double)S;
    return 0;
}

constexpr uint32_t kLongCap = 16384; // sub-buckets too long for a workgroup that the hybrid sort lists (sx_long_subbuckets)
size_t sx_lms_prefix_bytes(uint64_t m)
{
    const size_t a = 256;
    const uint64_t cap = m / 4 + 1024;
    size_t b = 0;
    b += 2 * (m * 8 + a);   // keys
    b += 3 * (m * 4 + a);   // values, seed windows
    b += 3 * (cap * 8 + a); // refinement keys: kept copy + sort ping-pong
    b += 11 * (cap * 4 + a); // apos x2, ap x2, ap_new, agid, order x2, group sizes, the sub-list of long groups x2
    b += 3 * (cap + a);     // heads
    b += 2 * ((m / 8192 + 2) * 4 + a); // tie counts and offsets per tile of the sorted keys
    b += 3 * (size_t)(m / 4096 + 2) * 4 + a; // the same, and the owned range's start, per workgroup of the local sort
    b += 3 * (size_t)kLongCap * 4 + a;       // starts, lengths, offsets of the sub-buckets too long for a workgroup
    return b + 4096;
}

// Returns 0 and *resolved = 1 with *out = device array of the m LMS suffix positions in
// suffix order; *resolved = 0 when the caller must use the general path.
int sx_sort_lms_by_prefix(sx_ctx *ctx, const sx_text_info &ti, sx_arena &am, const uint32_t **out,
                          const void **seed_windows, int *resolved, bool all_suffixes)
{
    *resolved = 0;
    *seed_windows = nullptr;
    // all_suffixes: the same machinery over every position of the text instead of the LMS positions; what comes
    // out is the suffix array itself (sx_build.hip decides when that is the cheaper way)
    const uint64_t m = all_suffixes ? ti.N : ti.m;
    const uint32_t base = ti.maxc + 1;
    // longest prefix whose number fits 63 bits
    uint32_t Cmax = 0;
    for (double cap63 = 9.2e18, v = 1.0; v * base <= cap63 && Cmax < 63; v *= base) ++Cmax;
    if (Cmax < 1) Cmax = 1;
    // shortest prefix that tells ~97 % of m suffixes of a uniformly random text apart
    uint32_t C = 1;
    {
        // (the direct sort of all suffixes accepts twice the ties to stay within 40 key bits at 32, 128, ... symbols)
        const double eff = ti.maxc > 2 ? (double)ti.maxc : 2.0, target = (all_suffixes ? 16.0 : 32.0) * (double)m;
        double v = eff;
        while (v < target && C < Cmax) {
            v *= eff;
            ++C;
        }
        // Skewed symbol frequencies (a genome with 5 % N, a GC-poor one): two suffixes agree on a symbol with
        // probability sum p^2, not 1 / maxc.  A prefix that is too short costs a whole second sort with the longest
        // key; one symbol too many costs at most a pass.  So the prefix is also made long enough for <= ~6 % ties
        // under the text's own collision rate (uniform text: never the larger of the two; symbols that only occur
        // in long runs, like N, hardly enter the rate but do enter maxc: that case needs the margin).
        double sum_p2 = 0.0;
        const double n_sym = (double)ti.N - 1.0;
        for (int c = 1; c < 256 && n_sym > 0; ++c) {
            const double pc = (double)ti.h_all[c] / n_sym;
            sum_p2 += pc * pc;
        }
        if (sum_p2 > 0.0 && sum_p2 < 1.0) {
            const double eff_c = 1.0 / sum_p2, target_c = 16.0 * (double)m;
            uint32_t Cc = 1;
            for (double u = eff_c; u < target_c && Cc < Cmax; u *= eff_c) ++Cc;
            if (Cc > C) C = Cc;
        }
    }
    // Round 4: four-letter texts take one symbol more where the dense key still leaves at most 19 bits to the local sort
    // (2 C + length field <= 41).  The prefix above leaves ~1.8 % of uniformly random suffixes tied, one symbol more a
    // quarter of that, and the tied suffixes' refinement -- random reads of the text -- cost more than the symbol: same box,
    // 1 GiB 22.13 -> 21.13 ms (refinement 1.00 -> 0.28 ms, every other class unchanged), 256 MiB 6.36 -> 6.19; a symbol
    // less: 23.21.  Taken back below if the text turns out not to take the hybrid sort (plain passes would pay a pass for it).
    bool one_more = false;
    if (!all_suffixes && base == 5 && ctx->prefix_symbols <= 0 && ctx->sort_mode != 1 && C + 1 <= Cmax &&
        2 * (C + 1) + (uint32_t)sx_bitlen(C + 1) <= 41) {
        ++C;
        one_more = true;
    }
    // (the direct sort of all suffixes likewise, while the key stays within 40 bits -- five 8-bit digits either way, at most
    //  16 bits for the local sort: 1 GiB of 20 symbols 58.5 -> 56.3 ms, ties refined in 0.19 instead of 2.7 ms; bytes stay at
    //  five symbols, six would be 48 bits)
    if (all_suffixes && ctx->prefix_symbols <= 0 && C + 1 <= Cmax) {
        double top = 1.0;
        for (uint32_t i = 0; i <= C; ++i) top *= (double)base;
        if (top <= 1099511627776.0) ++C; // base^(C+1) <= 2^40
    }
    if (ctx->prefix_symbols > 0) C = (uint32_t)ctx->prefix_symbols < Cmax ? (uint32_t)ctx->prefix_symbols : Cmax; // (tests)
    const uint32_t cap = (uint32_t)(m / 4 + 1024);
    uint64_t *ka = am.take<uint64_t>(m), *kb = am.take<uint64_t>(m);
    uint32_t *va = am.take<uint32_t>(m), *vb = am.take<uint32_t>(m);
    uint64_t *key_keep = am.take<uint64_t>(cap), *rk_a = am.take<uint64_t>(cap), *rk_b = am.take<uint64_t>(cap);
    uint32_t *apos = am.take<uint32_t>(cap), *apos2 = am.take<uint32_t>(cap);
    uint32_t *ap = am.take<uint32_t>(cap), *ap2 = am.take<uint32_t>(cap), *ap_new = am.take<uint32_t>(cap);
    uint32_t *agid = am.take<uint32_t>(cap), *ord_a = am.take<uint32_t>(cap), *ord_b = am.take<uint32_t>(cap);
    uint32_t *gsize = am.take<uint32_t>(cap), *sub_t = am.take<uint32_t>(cap), *sub_gid = am.take<uint32_t>(cap);
    uint8_t *head = am.take<uint8_t>(cap), *head2 = am.take<uint8_t>(cap), *head_new = am.take<uint8_t>(cap);
    uint32_t *seedw = am.take<uint32_t>(m);
    uint32_t *d_scalar = am.take<uint32_t>(16);
    const uint32_t tied_tiles = sx_div_up(m, kTiedTile);
    uint32_t *tile_cnt = am.take<uint32_t>(tied_tiles), *tile_pos = am.take<uint32_t>(tied_tiles);
    const uint32_t ls_tiles = sx_local_sort_tiles(m);
    uint32_t *tile_lsrt = am.take<uint32_t>(3 * (size_t)ls_tiles); // start, tied members, offset of every local-sort workgroup
    // sub-buckets too long for a workgroup (repeat families, AT-rich prefixes): their starts, lengths, offsets (sx_long_subbuckets)
    uint32_t *long_list = ctx->long_subbuckets_off ? nullptr : am.take<uint32_t>(3 * (size_t)kLongCap);
    if (!tile_cnt || !tile_pos || !seedw || !ka || !kb || !va || !vb || !key_keep || !rk_a || !rk_b || !apos || !apos2 || !ap || !ap2 || !ap_new ||
        !agid || !ord_a || !ord_b || !gsize || !sub_t || !sub_gid || !head || !head2 || !head_new || !d_scalar)
        return sx_fail_msg(ctx, SX_E_INTERNAL, "arena: LMS prefix sort");
    const dim3 block(kBlock);

    // offset of every classification tile in the global order of the LMS positions
    uint32_t *tile_lms = ti.tile_u32, *tile_off = all_suffixes ? nullptr : ti.tile_u32 + 4 * (size_t)ti.ntiles;
    if (!all_suffixes) SX_TRY((device_scan<OpAdd>(ctx, ti.ntiles, InU32{tile_lms}, OutExclusive{tile_off}, nullptr)));

    const uint64_t *ks = nullptr; // sorted keys (not kept by the hybrid sort)
    uint32_t *vs = nullptr;
    uint32_t A = 0;
    int kbits = 64;
    bool embed = false;
    uint32_t embed_chars = 0; // symbols of the windows the keys carry (embed)
    uint64_t kmask = ~0ull;
#ifndef SX_DENSE4
#define SX_DENSE4 1
#endif
    for (int attempt = 0;; ++attempt) {
        // four symbols and the sentinel: dense keys of two bits a symbol and a length field (dense4_finish); the plain-passes
        // mode keeps the base-5 keys (tests run both)
        const uint32_t lenbits = (uint32_t)sx_bitlen(C);
        int kbits_wnd = 64, kbits_base = 64;
        // (dense keys only with the hybrid sort: plain passes run faster on the base-5 keys' uneven digits -- the genome-like
        //  text 41.9 against 40.5 ms --, so the decisions below are taken for the dense keys first and, if they end without
        //  the hybrid sort, once more for the base-5 keys)
        bool dense4 = SX_DENSE4 && !all_suffixes && base == 5 && ctx->sort_mode != 1 && 2 * C + lenbits <= 60;
        wnd_cfg wcfg;
        int sort_db = 8, top_bits = 0; // top_bits: of the hybrid sort; 0: plain passes over all key bits
        bool ls_skewed = false;        // the hybrid sort was chosen by the mean sub-bucket of skewed symbol counts (a genome's: repeat
                                       // families crowd the LDS step's bins, whose workgroups the kernel of rounds 3 and 4 takes at once)
        uint8_t *dig0 = nullptr;
        for (;;) {
        {
            uint64_t top = 1; // base^C - 1 is the largest key
            for (uint32_t i = 0; i < C; ++i) top *= base;
            kbits = dense4 ? (int)(2 * C + lenbits) : sx_bitlen(top - 1);
            if (kbits < 1) kbits = 1;
            kbits_base = sx_bitlen(top - 1) > 0 ? sx_bitlen(top - 1) : 1;
            kbits_wnd = kbits > kbits_base ? kbits : kbits_base; // (dense keys: the windows of the base-5 keys, for which the static kernels are built)
        }
        // Symbol windows in the unsorted key bits, when at least four symbols fit.  Texts of more than 16 symbols, whose
        // induction windows are 64-bit words, take part since round 4: the sort hands the windows on as 32-bit words (the
        // same layout with fewer symbols: code fields of B bits above a 4-bit count), so up to 28 / B symbols, and from
        // two on they are worth it -- an LMS seed's first symbol is its entry's bucket, its second the symbol byte of
        // the entry it induces, and the text is read again where that entry is scanned.  (Without them every seed's
        // window was gathered from the text in suffix order before the L pass: 3.6 * 10^8 random 16-byte reads, 8.3 of
        // the 94 ms of 1 GiB of bytes through the induced-sort passes.)
        const bool wide = sx_window_cfg(ti.maxc, wcfg);
        uint32_t wchars = 0;
        if (64 - kbits_wnd > kCntBits) wchars = (uint32_t)(64 - kbits_wnd - kCntBits) / wcfg.B;
        if (wchars > wcfg.CW) wchars = wcfg.CW;
        if (wide && wchars > (32u - (uint32_t)kCntBits) / wcfg.B) wchars = (32u - (uint32_t)kCntBits) / wcfg.B;
        embed = wide ? wchars >= 2 : wchars >= 4;
        embed_chars = embed ? wchars : 0;
        if (all_suffixes) {
            // no induction follows a direct sort; one symbol of window is the BWT symbol of the suffix, for free
            wchars = 64 - kbits >= (int)(kCntBits + wcfg.B) ? 1u : 0u;
            embed = wchars == 1;
        }
        wcfg.CW = embed ? wchars : 0;
        kmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
        // the key kernels leave the first pass's digits there (one byte each: 8-bit digits only)
        sort_db = sx_sort_digit_bits(ctx);
        dig0 = (uint8_t *)sx_sort_digit_buffer(ctx, m, sort_db);
        if (!dig0) return sx_fail_msg(ctx, SX_E_NOMEM, "sort workspace");
        // Hybrid sort (sx_localsort.hip): only the top 24 key bits go through HBM passes, the sub-buckets they leave are
        // ordered in LDS.  It needs sub-buckets that fit a workgroup: a prefix of 24 / log2(base) symbols must not be too
        // frequent.  Judged here from the text's most frequent symbol (a run of it is the most frequent prefix of a text
        // without repeats); repeats show when the kernel finds a sub-bucket that does not fit, and LSD passes finish the job.
        top_bits = 0;
        if (ctx->sort_mode != 1 && sort_db == 8 && tile_lsrt != nullptr) {
            // Sub-buckets the top bits leave: the low L = kbits - top bits span base^(L / log2 base) key values, so the
            // top bits tell apart prefixes of C - L / log2(base) symbols (A C G T in base 5, 17 symbols, L = 16: 10.1) -- and
            // a text has about (effective alphabet)^(that many) of those, the effective alphabet being 1 / sum p^2 (4,
            // not 5: a fifth of the key space per symbol is never used).  Also: the copies of the most frequent one.
            // Three passes (24 bits).  Four passes (32 bits) leave short sub-buckets in texts of 2 Gi symbols and more and
            // with skewed frequencies too (2 GiB of uniform DNA: 40.3 against 43.2 ms for six plain passes), but are only
            // taken when asked for (SX_FLAG_SORT_MODE 3): texts that long are genomes, their repeat families overflow a
            // workgroup whatever the symbol counts promise, and a failed attempt costs more than a good one saves
            // (the genome-like 1 GiB text: 58 instead of 44 ms).  (Four-letter texts with dense keys: see cand0 below -- their
            // sub-buckets' lengths are known, and 2 GiB ... 8 GiB of uniform DNA take the hybrid sort with 22 / 24 top bits.)
            double pmax = 0.0, sum_p2 = 0.0;
            const double n_sym = (double)ti.N - 1.0;
            for (int c = 1; c < 256 && n_sym > 0; ++c) {
                const double pc = (double)ti.h_all[c] / n_sym;
                if (pc > pmax) pmax = pc;
                sum_p2 += pc * pc;
            }
            const double eff = sum_p2 > 0.0 ? 1.0 / sum_p2 : 1.0;
#ifndef SX_DENSE4_TOP
#define SX_DENSE4_TOP 22 // (1 GiB of DNA, whole step: base-5 keys 23.30 ms; dense keys with 24 top bits 22.79, 22: 22.71, 21: 22.73, 20: 22.79)
#endif
            // Dense keys: a sub-bucket is a prefix of top_bits / 2 symbols, so its mean length (m / eff^symbols) and that of
            // the most frequent symbol's run (m * pmax^symbols) are known better than for base-b keys; LMS suffixes favour
            // some prefixes (the fullest sub-bucket of uniform text holds 3.6 times the mean, measured), so 4.5 times the
            // larger of the two, with half as much again for safety, has to fit the 1024 pairs between a span's end and the
            // reach.  Skewed symbol counts (a genome: 30 % A) give the run of the most frequent symbol many times the mean:
            // such texts have repeat families too, and are left to plain passes as before.  22 top bits where they do
            // (1 GiB ... 2 GiB of uniform DNA), else 24 (up to 8 GiB).
            int cand0 = dense4 ? SX_DENSE4_TOP : 24;
            bool dense_fits = false;
            if (dense4 && ctx->sort_mode == 0) {
                for (int tb = SX_DENSE4_TOP; tb <= 24 && !dense_fits; tb += 2) {
                    const double sy = (double)tb / 2.0, mean_d = (double)m / pow(eff, sy), top_d = (double)m * pow(pmax, sy);
                    if (top_d <= 1.5 * mean_d && 4.5 * top_d * 1.5 <= 1024.0 && sx_local_sort_applies(m, kbits, tb)) cand0 = tb, dense_fits = true;
                }
                // Round 4: skewed symbol counts and repeat families no longer rule the hybrid sort out -- the sub-buckets that
                // do not fit a workgroup are listed and ordered by HBM passes of their own (sx_long_subbuckets) --, so what
                // counts is the mean sub-bucket (the genome-like 1 GiB text: 115 pairs at 22 bits, 6 % of the pairs in long
                // sub-buckets): three HBM passes and the LDS step instead of five passes, a key kernel and the pass that marks
                // the ties.  (A text whose long sub-buckets hold a quarter of the pairs falls back as before.)
                for (int tb = SX_DENSE4_TOP; tb <= 24 && !dense_fits && long_list != nullptr; tb += 2) {
                    const double mean_d = (double)m / pow(eff, (double)tb / 2.0);
                    if (mean_d <= 256.0 && sx_local_sort_applies(m, kbits, tb)) cand0 = tb, dense_fits = true, ls_skewed = true;
                }
            }
            for (int ci = 0; ci < 2 && top_bits == 0; ++ci) {
                const int cand = ci == 0 ? cand0 : 32;
                if (!sx_local_sort_applies(m, kbits, cand)) continue;
                if (ctx->sort_mode >= 2) { // forced (tests): 2 three passes, 3 four
                    if ((ctx->sort_mode == 2) == (ci == 0)) top_bits = cand;
                    continue;
                }
                const double syms = (double)C - (double)(kbits_base - (ci == 0 ? 24 : 32)) / log2((double)base);
                const double mean_bucket = syms > 0.0 ? (double)m / pow(eff, syms) : (double)m;
                const double top_bucket = syms > 0.0 ? (double)m * pow(pmax, syms) : (double)m;
                if (ci == 0 && m >= (1u << 22) && (dense4 ? dense_fits : (mean_bucket <= 300.0 && top_bucket <= 400.0))) top_bits = cand; // (a workgroup owns the sub-buckets that start in its span and end within its reach: one that starts
                                                                                                       //  at the span's last pair may hold kLsCap - kLsSpan = 1024 pairs, one that starts earlier more)
            }
        }
        if (!dense4 || top_bits != 0) break;
        dense4 = false;
        if (one_more && attempt == 0) { // (no hybrid sort after all: the shorter key, as before)
            --C;
            one_more = false;
        }
        }
        const bool hybrid = top_bits != 0;
        const uint32_t dig_shift = hybrid ? (uint32_t)(kbits - top_bits) : 0u;
        const uint32_t dig_mask = kbits - (int)dig_shift >= 8 ? 0xFFu : (1u << (kbits - (int)dig_shift)) - 1u;
        // The direct sort's keys are a function of the text alone, so where the hybrid sort applies its first HBM pass computes
        // them itself (sx_textkey, sx_radix.hip): no key kernel, and 8 bytes a suffix that are neither written nor read back
        // (1 GiB of bytes: 2.4 ms of key kernel and 7.5 GB of the first scatter's reads).  Switch for the A/B and the
        // tests: SX_FLAG_TEXT_KEYS_OFF keeps the key kernel.
        const bool text_keyed = all_suffixes && C <= 12 && hybrid && sort_db == 8 && !ctx->text_keys_off;
        sx_textkey tkey = {ti.T, base, C, 1u, 1u, (uint32_t)kbits, wcfg.CW ? 1u : 0u};
        for (uint32_t i = 0; i < 3; ++i) tkey.pow3 *= base;
        for (uint32_t i = 0; i < C % 3; ++i) tkey.powR *= base;
        // The LMS sort of a four-letter text likewise (sx_lmskey, round 4): the hybrid sort's first pass lists the LMS
        // suffixes of its piece of the text and computes their dense keys itself -- no key kernel, and 13 bytes an LMS suffix
        // (key, position, first digit) that are neither written nor read back.  Same switch.
        const pkey_cfg lms_kc = pkey_make(dense4 ? 4u : base, C);
        const uint32_t lms_shape = (!all_suffixes && lms_kc.dot && wcfg.B == 2 && wcfg.CW) ? lms_key_shape(C, wcfg.CW, 2) : 0u;
        const bool lms_keyed = !all_suffixes && hybrid && dense4 && sort_db == 8 && !ctx->text_keys_off && lms_shape != 0 &&
                               sx_sort_lms_keys_applies(m, ti.ntiles, lms_shape);
        const sx_lmskey lkey = {ti.T, (const uint16_t *)ti.lmsbits, (const uint32_t *)tile_off, ti.ntiles, (uint32_t)m, lms_kc, wcfg,
                                (uint32_t)kbits, lenbits, (uint64_t)ti.n + 1, lms_shape};
        if (text_keyed || lms_keyed) {
        } else if (all_suffixes && C <= 12) {
            const pkey_cfg kc = pkey_make(base, C);
            const dim3 grid16(sx_div_up(m, kBlock * 16));
#define SX_KEYS16(G) sx_launch(ctx, SX_KC_KEYS, m * 10, all_keys16_kernel<G>, grid16, block, ti.T, m, kc, (uint32_t)kbits, wcfg, ka, dig0, dig_shift, dig_mask)
            switch (kc.G) {
            case 10: SX_KEYS16(10); break;
            case 6: SX_KEYS16(6); break;
            case 4: SX_KEYS16(4); break;
            default: SX_KEYS16(3); break;
            }
#undef SX_KEYS16
        } else if (all_suffixes)
            sx_launch(ctx, SX_KC_KEYS, m * 13, all_keys_kernel, dim3(sx_div_up(m, kBlock)), block, ti.T, m, pkey_make(base, C),
                      (uint32_t)kbits, wcfg, ka, dig0, dig_shift, dig_mask);
        else
        {
            const pkey_cfg kc = pkey_make(dense4 ? 4u : base, C);
            // DNA-like texts: everything static for the usual prefix lengths (base 5: the window takes what
            // 64 - kbits - 4 bits hold)
            const bool dna = kc.dot && (wcfg.B == 2 || wcfg.B == 3) && kbits >= 8;
#define SX_TILE_KEYS(CS, WS, BS)                                                                                       \
    sx_launch(ctx, SX_KC_KEYS, m * 12 + ti.N + ti.N / 8, lms_tile_keys_kernel<CS, WS, BS>, dim3(ti.ntiles), block,     \
              ti.T, (const uint16_t *)ti.lmsbits, (const uint32_t *)tile_off, kc, (uint32_t)kbits, wcfg, ka, va, dig0, dig_shift, dig_mask, \
              (uint64_t)(dense4 ? ti.n + 1 : 0), lenbits)
            const uint32_t shape = dna ? (C * 16 + wcfg.CW) * 4 + wcfg.B : 0u;
            switch (shape) {
            // base 5 (A C G T): 64 Mi ... 4 Gi symbols
            case (14 * 16 + 13) * 4 + 2: SX_TILE_KEYS(14, 13, 2); break;
            case (15 * 16 + 12) * 4 + 2: SX_TILE_KEYS(15, 12, 2); break;
            case (16 * 16 + 11) * 4 + 2: SX_TILE_KEYS(16, 11, 2); break;
            case (17 * 16 + 10) * 4 + 2: SX_TILE_KEYS(17, 10, 2); break;
            case (18 * 16 + 9) * 4 + 2: SX_TILE_KEYS(18, 9, 2); break;
            // base 6 (A C G N T)
            case (13 * 16 + 8) * 4 + 3: SX_TILE_KEYS(13, 8, 3); break;
            case (14 * 16 + 7) * 4 + 3: SX_TILE_KEYS(14, 7, 3); break;
            case (15 * 16 + 7) * 4 + 3: SX_TILE_KEYS(15, 7, 3); break;
            case (16 * 16 + 6) * 4 + 3: SX_TILE_KEYS(16, 6, 3); break;
            default: SX_TILE_KEYS(0, 0, 0); break;
            }
#undef SX_TILE_KEYS
        }
        int in_b = 0;
        ctx->stats.sort_local = 0;
        bool listed = false; // the members of groups of equal keys are in (apos, ap, head), A of them
        if (hybrid) {
            // three stable passes on the top 24 bits, then the sub-buckets in LDS: positions, windows and ties in one go
            SX_TRY(sx_sort_pairs(ctx, ka, va, kb, vb, m, kbits - top_bits, kbits, &in_b, all_suffixes, true, 8, text_keyed ? &tkey : nullptr,
                                 lms_keyed ? &lkey : nullptr));
            const uint64_t *kin = in_b ? kb : ka;
            const uint32_t *vin = in_b ? vb : va;
            uint32_t *vo = in_b ? va : vb;
            // (dense keys: a sub-bucket is a prefix of top_bits / 2 symbols; LMS suffixes favour some prefixes -- two thirds of
            //  them begin with the smallest symbol -- so the fullest sub-bucket of uniform text holds 3.6 times the mean: 4.5 times
            //  the most frequent symbol's share to that power, for the local sort's choice of its span)
            uint32_t longest = 0;
            if ((dense4 || all_suffixes) && ctx->sort_mode == 0) {
                double pm = 0.0;
                for (int c = 1; c < 256; ++c) pm = (double)ti.h_all[c] > pm ? (double)ti.h_all[c] : pm;
                pm /= (double)ti.N - 1.0;
                // (all suffixes of a wide alphabet, the direct sort: no favoured prefixes, a sub-bucket is a prefix of
                //  top_bits / log2(base) symbols -- three bytes at 256 symbols --: the most frequent symbol's run and what
                //  chance adds to the fullest of 2^24 sub-buckets)
                const double run = (double)m * pow(pm, dense4 ? (double)top_bits / 2.0 : (double)top_bits / log2((double)base));
                const double est = dense4 ? 4.5 * run : 1.2 * run + 8.0 * sqrt(run) + 16.0;
                longest = est < 1.0 ? 1u : (est > 1e9 ? 1000000000u : (uint32_t)est);
            }
            uint32_t res[3] = {0, 0, 0};
            SX_TRY(sx_local_sort(ctx, kin, vin, m, kbits, top_bits, vo, embed ? seedw : nullptr, tile_lsrt, tile_lsrt + ls_tiles,
                                 tile_lsrt + 2 * (size_t)ls_tiles, (uint2 *)(in_b ? ka : kb), dig0, apos, ap, head, cap, d_scalar, longest,
                                 long_list, kLongCap, res, ls_skewed));
            bool fits = !(res[1] & 1u);
            ctx->stats.long_subbuckets = 0;
            if (fits && res[2] != 0) {
                // sub-buckets that did not fit a workgroup were left out and listed: their members are ordered by HBM passes of
                // their own (the refinement's arrays are free until the ties are refined: a quarter of the pairs at most)
                uint32_t tied_all = res[0], pairs = 0;
                int done = 0;
                SX_TRY(sx_long_subbuckets(ctx, kin, vin, m, kbits, top_bits, res[2], long_list, kLongCap, rk_a, rk_b, ord_a, ord_b, cap - 1024, vo,
                                          embed ? seedw : nullptr, apos, ap, head, cap, res[0], d_scalar + 4, &tied_all, &pairs, &done));
                ctx->stats.long_subbuckets = res[2];
                fits = done != 0;
                res[0] = done ? tied_all : res[0] + pairs; // (not done: those pairs are as good as tied -- the decision below)
            }
            if (fits) {
                A = res[0];
                vs = vo;
                ks = nullptr; // (the sorted keys are not written by this path; nothing below reads them)
                listed = true;
                ctx->stats.sort_local = 1u | (res[1] & 2u) | (top_bits == 32 ? 4u : 0u) | ((text_keyed || lms_keyed) ? 8u : 0u); // (bit 1: some workgroup ordered its pairs by stable passes)
            } else {
                // a sub-bucket too long for a workgroup (a repeated prefix): plain LSD passes over all key bits from here
                // -- unless the workgroups that did finish have listed more tied suffixes already than the refinement
                // takes, and more than a longer key would be tried for: the general path's turn at once (a collection of
                // near-identical sequences: five passes over all pairs and the pass that marks the ties, 11 of 210 ms)
                if (res[0] > cap - 1024 && (attempt != 0 || C >= Cmax || (uint64_t)res[0] * 2 > m)) {
                    ctx->stats.n_names = m - res[0];
                    ctx->stats.key_slots = C;
                    ctx->stats.key_bits = (uint32_t)kbits;
                    return 0;
                }
                uint64_t *k0 = in_b ? kb : ka, *k1 = in_b ? ka : kb;
                uint32_t *v0 = in_b ? vb : va, *v1 = in_b ? va : vb;
                int f = 0;
                SX_TRY(sx_sort_pairs(ctx, k0, v0, k1, v1, m, 0, kbits, &f, false, false, 8));
                ks = f ? k1 : k0;
                vs = f ? v1 : v0;
                in_b = (ks == kb) ? 1 : 0;
            }
        } else {
            SX_TRY(sx_sort_pairs(ctx, ka, va, kb, vb, m, 0, kbits, &in_b, all_suffixes, sort_db == 8, sort_db)); // (all suffixes: value = index)
            ks = in_b ? kb : ka;
            vs = in_b ? vb : va;
        }
        // members of groups with equal keys
        if (!listed) {
            const uint32_t tiles = sx_div_up(m, kTiedTile);
            // staging in the sort's spare buffers: 8 bytes per slot in the other key array, heads in the other value array
            uint2 *stage = (uint2 *)(in_b ? ka : kb);
            uint8_t *stage_head = (uint8_t *)(in_b ? va : vb);
            sx_launch(ctx, SX_KC_NAMES, m * 13, tied_mark_kernel, dim3(tiles), block, ks, (const uint32_t *)vs, m, kmask,
                      (uint32_t)kbits, embed ? seedw : nullptr, tile_cnt, stage, stage_head);
            SX_TRY((device_scan<OpAdd>(ctx, tiles, InU32{tile_cnt}, OutExclusive{tile_pos}, d_scalar, SX_KC_NAMES, 0)));
            sx_launch(ctx, SX_KC_NAMES, 0, tied_gather_kernel, dim3(tiles), block, (const uint32_t *)tile_cnt,
                      (const uint32_t *)tile_pos, (const uint2 *)stage, (const uint8_t *)stage_head, apos, ap, head, cap);
            SX_TRY(sx_readback(ctx, d_scalar, 1, &A));
        }
        ctx->stats.n_names = m - A; // suffixes told apart by the first sort
        ctx->stats.key_slots = C;
        ctx->stats.key_bits = (uint32_t)kbits;
        if (A <= cap - 1024) break;
        // skewed symbol frequencies leave a good quarter of the suffixes tied: once more with the longest key.  With
        // more than half of them tied the text repeats itself (words of a vocabulary, periods): a longer key would
        // tie them again, so the general path takes over without the second sort (8 passes over all pairs).
        if (attempt == 0 && C < Cmax && (uint64_t)A * 2 <= m) {
            C = Cmax;
            continue;
        }
        return 0; // repetitive text: general path
    }

    wnd_cfg full_wcfg; // refined slots get their windows straight from the text
    const bool wide_windows = sx_window_cfg(ti.maxc, full_wcfg);
    if (all_suffixes) full_wcfg.CW = 1; // (the seed windows are 32-bit words: one wide symbol fits, seven do not)
    else if (wide_windows && embed) full_wcfg.CW = embed_chars; // (32-bit words here too: as many symbols as the keys carried)
    // ties are refined with the longest key that fits (Cmax symbols a round)
    uint64_t top_r = 1;
    for (uint32_t i = 0; i < Cmax; ++i) top_r *= base;
    const int kbits_r = sx_bitlen(top_r - 1) > 0 ? sx_bitlen(top_r - 1) : 1;
    // keep what is still tied after a refinement step: (apos, ap_new, head_new) -> (apos, ap, head), A
    auto keep_tied = [&]() -> int {
        SX_TRY((device_compact(ctx, A, InStillTied{head_new, A}, OutStillTied{apos, ap_new, head_new, apos2, ap2, head2},
                               d_scalar, SX_KC_DOUBLING, (uint64_t)A * 20)));
        uint32_t A2 = 0;
        SX_TRY(sx_readback(ctx, d_scalar, 1, &A2));
        uint32_t *tp = apos; apos = apos2; apos2 = tp;
        tp = ap; ap = ap2; ap2 = tp;
        uint8_t *th = head; head = head2; head2 = th;
        A = A2;
        return 0;
    };
    // One refinement step by the next Cmax symbols for the groups of more than kSmallGroup members (the smaller ones
    // are settled by their head's thread): group sizes, then groups of up to kMidGroup members inside a workgroup's
    // LDS, the members of longer ones compacted and ordered by two radix sorts of (group, next key).
    // SX_FLAG_SORT_MODE 1 (plain passes only): every group of more than kSmallGroup members takes the radix sorts.
    const bool lds_tier = ctx->sort_mode != 1;
    const bool trace_rounds = getenv("STRALG_AMD_TRACE_REFINE") != nullptr; // (diagnostic: members and time of every refinement round)
    auto refine_larger_groups = [&](uint64_t skip) -> int {
        const pkey_cfg kc_r = pkey_make(base, Cmax);
        SX_TRY((device_scan<OpMax>(ctx, A, InActHead{head}, OutGroupIds{head, (uint64_t)A, agid, gsize}, nullptr,
                                   SX_KC_DOUBLING, (uint64_t)A * 9)));
        if (lds_tier) {
            ctx->stats.refine_tiers |= 1u;
            sx_launch(ctx, SX_KC_DOUBLING, (uint64_t)A * 60, refine_mid_groups_kernel, dim3(sx_div_up(A, kRmSpan)),
                      dim3(kRmThreads), ti.T, ti.n, (const uint32_t *)ap, (const uint32_t *)apos, (const uint32_t *)agid,
                      (const uint32_t *)gsize, (uint64_t)A, skip, kc_r, vs, ap_new, head_new,
                      embed ? seedw : nullptr, full_wcfg);
        }
        SX_TRY((device_compact(ctx, A, InLargeGroup{agid, gsize, (uint32_t)(lds_tier ? kMidGroup : kSmallGroup)},
                               OutLargeMember{ti.T, ap, agid, ti.n, skip, kc_r, sub_t, sub_gid, key_keep, rk_a, ord_a}, d_scalar,
                               SX_KC_DOUBLING, (uint64_t)A * 12)));
        uint32_t A3 = 0;
        SX_TRY(sx_readback(ctx, d_scalar, 1, &A3));
        if (A3 == 0) return 0;
        ctx->stats.refine_tiers |= 2u;
        uint32_t *sub_num = gsize; // (the sizes are not needed any more)
        SX_TRY((device_scan<OpAdd>(ctx, A3, InSubHead{sub_t, sub_gid}, OutGroupNumber{sub_num}, d_scalar + 3, SX_KC_DOUBLING,
                                   (uint64_t)A3 * 12)));
        // (the groups' numbers are sort digits: how many there are is bounded by their least length -- no read-back of the count,
        //  20 us of idle device a round; the bound is a bit too wide at most, and digits are taken eight bits at a time)
        const uint32_t least = (uint32_t)(lds_tier ? kMidGroup : kSmallGroup) + 1u;
        const uint32_t n_long_max = A3 / least > 1u ? A3 / least : 1u;
        const uint32_t gbits = n_long_max > 1 ? (uint32_t)sx_bitlen(n_long_max - 1) : 1u;
        if (trace_rounds) {
            uint32_t n_long = 0;
            SX_TRY(sx_readback(ctx, d_scalar + 3, 1, &n_long));
            fprintf(stderr, "stralg_amd refine:   %u members of %u groups too long for the LDS tier\n", A3, n_long);
        }
        // order by (group, next key), LSD: stable sort by next key, then stable sort by group
        int f = 0;
        SX_TRY(sx_sort_pairs(ctx, rk_a, ord_a, rk_b, ord_b, A3, 0, kbits_r, &f));
        uint32_t *ord1 = f ? ord_b : ord_a, *ord1_other = f ? ord_a : ord_b;
        uint64_t *k1 = f ? rk_a : rk_b, *k1_other = f ? rk_b : rk_a; // k1: free to overwrite
        sx_launch(ctx, SX_KC_DOUBLING, (uint64_t)A3 * 16, gid_keys_kernel, dim3(sx_div_up(A3, kBlock)), block,
                  (const uint32_t *)ord1, (const uint32_t *)sub_num, (uint64_t)A3, k1);
        SX_TRY(sx_sort_pairs(ctx, k1, ord1, k1_other, ord1_other, A3, 0, (int)gbits, &f));
        const uint32_t *order = f ? ord1_other : ord1;
        sx_launch(ctx, SX_KC_DOUBLING, (uint64_t)A3 * 44, refine_write_kernel, dim3(sx_div_up(A3, kBlock)), block, order,
                  (const uint32_t *)sub_t, (const uint32_t *)ap, (const uint32_t *)apos, (const uint32_t *)sub_num,
                  (const uint64_t *)key_keep, (uint64_t)A3, vs, ap_new, head_new, embed ? seedw : nullptr, ti.T, full_wcfg);
        return 0;
    };
    const bool trace = trace_rounds;
    for (int round = 1; A > 0; ++round) {
        const auto trace_t0 = std::chrono::steady_clock::now();
        const uint32_t trace_A = A;
        struct trace_end {
            bool on; int round; uint32_t a0; const uint32_t &a1; std::chrono::steady_clock::time_point t0;
            ~trace_end() { if (on) fprintf(stderr, "stralg_amd refine: round %d  %u -> %u tied members  %.3f ms\n", round, a0, a1,
                                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
        } trace_guard{trace, round, trace_A, A, trace_t0};
        // Few survivors are repeats proper: they get more rounds (each costs little) and, from the third round on,
        // small groups are finished by comparing the suffixes themselves.  Many survivors after four rounds, or any
        // after 32: repetitive text, general path.
        const bool few = (uint64_t)A * 16 <= m;
        if (round > (few ? 32 : 4)) return 0;
        ctx->stats.doubling_rounds++;
        const uint64_t skip = (uint64_t)C + (uint64_t)Cmax * (round - 1);
        uint32_t counters[2] = {0, 0};
        SX_CHECK(hipMemsetAsync(d_scalar + 1, 0, 2 * sizeof(uint32_t), ctx->stream));
        if (few && round >= 3) {
            SX_CHECK(hipMemcpyAsync(ap_new, ap, (size_t)A * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
            SX_CHECK(hipMemcpyAsync(head_new, head, (size_t)A, hipMemcpyDeviceToDevice, ctx->stream));
            sx_launch(ctx, SX_KC_DOUBLING, (uint64_t)A * 60, refine_by_comparison_kernel, dim3(sx_div_up(A, kBlock)), block, ti.T,
                      ti.n, (const uint32_t *)ap, (const uint32_t *)apos, (const uint8_t *)head, (uint64_t)A, skip, vs, ap_new, head_new,
                      embed ? seedw : nullptr, full_wcfg, d_scalar + 1);
            SX_TRY(sx_readback(ctx, d_scalar + 1, 2, counters));
            if (counters[1]) return 0; // a repeat beyond the comparison cap: general path
            SX_TRY(keep_tied());
            if (A == 0) break;
            SX_TRY(refine_larger_groups(skip)); // what is left sits in groups of more than eight: every member is rewritten
            SX_TRY(keep_tied());
            continue;
        }
        // groups of up to eight members: settled by their head's thread
#ifndef SX_RS_WAVES
#define SX_RS_WAVES 1
#endif
        if (SX_RS_WAVES && ctx->sort_mode != 1)
            sx_launch(ctx, SX_KC_DOUBLING, (uint64_t)A * 60, refine_small_groups_wave_kernel,
                      dim3(sx_div_up(sx_div_up(A, kRsStride), kWavesPerBlock)), block, ti.T, ti.n, (const uint32_t *)ap,
                      (const uint32_t *)apos, (const uint8_t *)head, (uint64_t)A, skip, pkey_make(base, Cmax), vs, ap_new, head_new,
                      embed ? seedw : nullptr, full_wcfg, d_scalar + 1);
        else
        sx_launch(ctx, SX_KC_DOUBLING, (uint64_t)A * 60, refine_small_groups_kernel, dim3(sx_div_up(A, kBlock)), block, ti.T,
                  ti.n, (const uint32_t *)ap, (const uint32_t *)apos, (const uint8_t *)head, (uint64_t)A, skip,
                  pkey_make(base, Cmax), vs, ap_new, head_new, embed ? seedw : nullptr, full_wcfg, d_scalar + 1);
        SX_TRY(sx_readback(ctx, d_scalar + 1, 1, counters));
        if (counters[0]) SX_TRY(refine_larger_groups(skip)); // larger groups exist: they are refined in LDS or by sorting
        SX_TRY(keep_tied());
    }
    *out = vs;
    *seed_windows = embed ? seedw : nullptr;
    *resolved = 1;
    return 0;
}
