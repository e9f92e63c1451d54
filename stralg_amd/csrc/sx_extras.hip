// sx_extras.hip -- consumers of a suffix array / BWT tables that is already on the device
// ("next" rows 3 and 4):
//
//   inverse[sa[i]] = i                                   stralg/suffix_array.c:53-60  compute_inverse
//   lcp[j] = longest common prefix of suffixes sa[j-1], sa[j]; lcp[0] = 0
//                                                        stralg/suffix_array.c:62-85  compute_lcp (Kasai)
//   exact FM-index search: (L, R) <- (C[a] + O(a, L), C[a] + O(a, R)) over the pattern, right to left
//                                                        stralg/bwt.c:164-199  init_bwt_exact_match_iter
//
// LCP keeps Kasai's invariant (the value drops by at most one from text position i to i+1) inside
// chunks of 64 consecutive text positions handled by one thread, suffixes compared 16 bytes at a
// time.  The value a chunk starts from comes from the same invariant over longer distances: the
// values at the chunk starts ("samples") are computed level by level, stride halving -- a new
// sample at distance s to the right of a known one starts its comparison at (that value - s) --
// so that no comparison ever repeats symbols a sample to its left has already matched.  Work is
// O(n) symbol comparisons per level in the worst case (periodic text) and next to nothing on
// ordinary text; with every chunk starting from 0 instead, a text with one long repeat paid the
// repeat's length once per chunk (quadratic on periodic strings).
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_internal.hpp"

namespace sx {

__global__ __launch_bounds__(kBlock) void inverse_kernel(const uint32_t *__restrict__ sa, uint64_t N,
                                                         uint32_t *__restrict__ inv, uint32_t *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const uint32_t p = sa[i];
    if ((uint64_t)p < N) inv[p] = (uint32_t)i;
    else atomicAdd(bad, 1u); // not a suffix array over N positions
}

// ---- the inverse of a long suffix array in two passes -------------------------------------------------------------
// inv[sa[i]] = i is one 4-byte store per entry at a random place: every store pulls a line in and pushes it out
// again (2^28 entries: 9.9 ms, 34 GB of traffic for 2 GiB of arrays).  Two passes keep the stores inside windows
// the L2 holds: the entries are first dealt into partitions by the top bits of their target -- a partition is a
// window of 2^19 targets and, the array being a permutation, receives exactly as many entries, so the partitions'
// places are known in advance and a workgroup only has to reserve its share of each with one atomic add (the
// order inside a partition does not matter) -- as (target, index) pairs, and a second pass walks the pairs in
// partition order, every XCD the partitions of its own eighth of the array.  1 + 2 GiB out, 2 + 1 GiB in.
constexpr int kInvThreads = 512, kInvItems = 16, kInvTile = kInvThreads * kInvItems;
#ifndef SX_INV_WINDOW_BITS
#define SX_INV_WINDOW_BITS 19 // (the CPU test harness builds with 8: several partitions on small arrays)
#endif
constexpr uint32_t kInvWindowBits = SX_INV_WINDOW_BITS, kInvMaxParts = 8192; // N <= 2^32: at most 8192 partitions of 2^19
__global__ __launch_bounds__(kInvThreads) void inverse_partition_kernel(const uint32_t *__restrict__ sa, const uint32_t *__restrict__ values /* or null: the index */,
                                                                         uint32_t values_back /* 1: entry i carries values[i - 1] (entry 0: 0xFFFFFFFF) */,
                                                                         uint64_t N, uint32_t nparts, uint32_t wbits,
                                                                         uint32_t *__restrict__ cursor /* entries dealt into each partition so far */,
                                                                         uint2 *__restrict__ pairs, uint32_t *__restrict__ bad)
{
    __shared__ uint32_t cnt[kInvMaxParts]; // the tile's entries per partition, then where its share of the partition begins
    const uint32_t t = threadIdx.x;
    for (uint64_t tile0 = (uint64_t)blockIdx.x * kInvTile; tile0 < N; tile0 += (uint64_t)gridDim.x * kInvTile) { // uniform
        for (uint32_t q = t; q < nparts; q += kInvThreads) cnt[q] = 0;
        __syncthreads();
        uint32_t p[kInvItems], r[kInvItems];
#pragma unroll
        for (int k = 0; k < kInvItems; ++k) {
            const uint64_t i = tile0 + (uint64_t)k * kInvThreads + t;
            p[k] = i < N ? sa[i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < kInvItems; ++k) {
            const uint64_t i = tile0 + (uint64_t)k * kInvThreads + t;
            r[k] = 0;
            if (i < N) {
                if ((uint64_t)p[k] < N) r[k] = atomicAdd(&cnt[p[k] >> wbits], 1u);
                else atomicAdd(bad, 1u); // not a suffix array over N positions
            }
        }
        __syncthreads();
        for (uint32_t q = t; q < nparts; q += kInvThreads) {
            const uint32_t c = cnt[q];
            if (c) {
                const uint32_t at = atomicAdd(&cursor[q], c);
                const uint64_t first = (uint64_t)q << wbits, room = (N - first < (1ull << wbits) ? N - first : (1ull << wbits));
                if ((uint64_t)at + c > room) { // more entries than targets: not a permutation
                    atomicAdd(bad, 1u);
                    cnt[q] = 0xFFFFFFFFu;
                } else {
                    cnt[q] = (uint32_t)first + at;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kInvItems; ++k) {
            const uint64_t i = tile0 + (uint64_t)k * kInvThreads + t;
            if (i < N && (uint64_t)p[k] < N) {
                const uint32_t base = cnt[p[k] >> wbits];
                if (base != 0xFFFFFFFFu) {
                    uint2 e;
                    e.x = p[k], e.y = values ? (values_back ? (i ? values[i - 1] : 0xFFFFFFFFu) : values[i]) : (uint32_t)i;
                    pairs[(uint64_t)base + r[k]] = e;
                }
            }
        }
        __syncthreads(); // `cnt` is zeroed for the next tile
    }
}

// pairs in partition order -> inv; workgroup b of XCD x = b % 8 takes the x-th eighth of the pairs, in order
// A target array that is not a permutation leaves slots of `pairs` unwritten (the partition pass drops what overflows a
// window and counts it in `bad`): such a slot holds whatever the slab held before, so nothing is stored once `bad` is
// set, and no store leaves [0, N) whatever a slot holds.
__global__ __launch_bounds__(kBlock) void inverse_apply_kernel(const uint2 *__restrict__ pairs, uint64_t N, uint32_t *__restrict__ inv, int by_xcd,
                                                               const uint32_t *__restrict__ bad)
{
    constexpr int kPer = 8;
    if (bad && *bad) return; // (uniform: the partition pass has ended)
    const uint64_t chunks = (N + (uint64_t)kBlock * kPer - 1) / ((uint64_t)kBlock * kPer);
    const uint64_t per_xcd = (chunks + 7) / 8;
    const uint64_t chunk = by_xcd ? (uint64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : (uint64_t)blockIdx.x;
    if ((by_xcd && (blockIdx.x >> 3) >= per_xcd) || chunk >= chunks) return;
    const uint64_t j0 = chunk * kBlock * kPer;
    uint2 e[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const uint64_t j = j0 + (uint64_t)k * kBlock + threadIdx.x;
        if (j < N) { // (streaming loads: the pairs are read once and must not push the window's half-written lines out of the L2)
            const uint64_t raw = __builtin_nontemporal_load(reinterpret_cast<const uint64_t *>(pairs) + j);
            e[k].x = (uint32_t)raw, e[k].y = (uint32_t)(raw >> 32);
        }
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const uint64_t j = j0 + (uint64_t)k * kBlock + threadIdx.x;
        if (j < N && (uint64_t)e[k].x < N) inv[e[k].x] = e[k].y;
    }
}

// ---- round 4: three passes, the last one inside LDS ---------------------------------------------------------------
// The second pass above still scatters 4-byte stores over a 2 MiB window: rocprofv3 counted 9.9 GB written for 1 GiB of
// inverse entries -- the L2 hands the window's lines to memory partially filled.  A window whose entries fit a workgroup's
// LDS has no such problem: its pairs are scattered into LDS and the window leaves as one contiguous block.  Such windows
// (2^15 targets, 128 KiB) are too many to deal the array into at once -- a tile of 8192 entries would write one 8-byte
// pair per window --, so the pairs of every coarse window (2^19 targets, pass 1 as above) are dealt into its 16 fine
// windows first (a pass that stays inside 4 MiB: 512-pair runs), and the fine windows are turned into inverse entries by
// a workgroup each.  10 GiB moved instead of 14, every store a full line.
#ifndef SX_INV_FINE_BITS
#define SX_INV_FINE_BITS 15 // (the CPU test harness builds with 6)
#endif
constexpr uint32_t kInvFineBits = SX_INV_FINE_BITS, kInvFineMax = 4096; // fine windows per coarse window: at most this many
__global__ __launch_bounds__(kInvThreads) void inverse_refine_kernel(const uint2 *__restrict__ pairs, uint64_t N, uint32_t wbits,
                                                                      uint32_t *__restrict__ fine_cursor, uint2 *__restrict__ pairs2,
                                                                      uint32_t *__restrict__ bad)
{
    __shared__ uint32_t cnt[kInvFineMax];
    if (*bad) return; // (the first pass found that the array is no permutation: its pairs have holes)
    const uint32_t t = threadIdx.x;
    for (uint64_t tile0 = (uint64_t)blockIdx.x * kInvTile; tile0 < N; tile0 += (uint64_t)gridDim.x * kInvTile) { // uniform
        // (a permutation fills every coarse window exactly, so the tile's pairs belong to the coarse windows its positions
        //  lie in: one of them at the production sizes, several in the CPU test harness's small windows)
        const uint64_t tile1 = tile0 + kInvTile < N ? tile0 + kInvTile : N;
        const uint32_t fine0 = (uint32_t)((tile0 >> wbits) << (wbits - kInvFineBits));
        const uint32_t F = (uint32_t)((((tile1 - 1) >> wbits) + 1) << (wbits - kInvFineBits)) - fine0;
        for (uint32_t q = t; q < F; q += kInvThreads) cnt[q] = 0;
        __syncthreads();
        uint2 e[kInvItems];
        uint32_t r[kInvItems];
#pragma unroll
        for (int k = 0; k < kInvItems; ++k) {
            const uint64_t i = tile0 + (uint64_t)k * kInvThreads + t;
            e[k].x = 0xFFFFFFFFu, e[k].y = 0;
            if (i < N) {
                const uint64_t raw = __builtin_nontemporal_load(reinterpret_cast<const uint64_t *>(pairs) + i);
                e[k].x = (uint32_t)raw, e[k].y = (uint32_t)(raw >> 32);
            }
        }
#pragma unroll
        for (int k = 0; k < kInvItems; ++k) {
            const uint64_t i = tile0 + (uint64_t)k * kInvThreads + t;
            r[k] = 0;
            if (i < N) {
                const uint32_t f = (e[k].x >> kInvFineBits) - fine0;
                if ((uint64_t)e[k].x < N && f < F) r[k] = atomicAdd(&cnt[f], 1u);
                else atomicAdd(bad, 1u);
            }
        }
        __syncthreads();
        for (uint32_t q = t; q < F; q += kInvThreads) {
            const uint32_t c = cnt[q];
            if (c) {
                const uint32_t at = atomicAdd(&fine_cursor[fine0 + q], c);
                const uint64_t first = (uint64_t)(fine0 + q) << kInvFineBits;
                const uint64_t room = first >= N ? 0 : (N - first < (1ull << kInvFineBits) ? N - first : (1ull << kInvFineBits));
                if ((uint64_t)at + c > room) {
                    atomicAdd(bad, 1u);
                    cnt[q] = 0xFFFFFFFFu;
                } else {
                    cnt[q] = (uint32_t)first + at;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kInvItems; ++k) {
            const uint64_t i = tile0 + (uint64_t)k * kInvThreads + t;
            if (i < N && (uint64_t)e[k].x < N) {
                const uint32_t f = (e[k].x >> kInvFineBits) - fine0;
                if (f < F) {
                    const uint32_t base = cnt[f];
                    if (base != 0xFFFFFFFFu) pairs2[(uint64_t)base + r[k]] = e[k];
                }
            }
        }
        __syncthreads(); // `cnt` is zeroed for the next tile
    }
}

// workgroup = fine window: its pairs into LDS by target, the window out as one block
constexpr int kInvWinThreads = 1024;
__global__ __launch_bounds__(kInvWinThreads) void inverse_window_kernel(const uint2 *__restrict__ pairs2, uint64_t N,
                                                                        uint32_t *__restrict__ inv, const uint32_t *__restrict__ bad)
{
    __shared__ uint32_t w[1u << kInvFineBits];
    if (*bad) return; // (not a permutation: some window's pairs have holes; the caller reports it)
    constexpr uint32_t kWin = 1u << kInvFineBits;
    for (uint64_t f = blockIdx.x; (f << kInvFineBits) < N; f += gridDim.x) { // uniform
        const uint64_t first = f << kInvFineBits;
        const uint32_t n_f = N - first < kWin ? (uint32_t)(N - first) : kWin;
        // (eight loads of a thread in flight together: with a load and its LDS store in one loop body a window's 32 768 pairs
        //  were 32 trips to memory one after the other, and a workgroup of this size is alone on its CU)
        for (uint32_t i0 = 0; i0 < n_f; i0 += kInvWinThreads * 8u) { // uniform
            uint64_t raw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t i = i0 + (uint32_t)j * kInvWinThreads + threadIdx.x;
                raw[j] = i < n_f ? __builtin_nontemporal_load(reinterpret_cast<const uint64_t *>(pairs2) + first + i) : 0ull;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t i = i0 + (uint32_t)j * kInvWinThreads + threadIdx.x;
                if (i < n_f) w[(uint32_t)raw[j] & (kWin - 1u)] = (uint32_t)(raw[j] >> 32);
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x * 4u; i < n_f; i += kInvWinThreads * 4u) {
            if (i + 4u <= n_f && ((uintptr_t)(inv + first) & 15u) == 0) {
                uint4 v;
                v.x = w[i], v.y = w[i + 1], v.z = w[i + 2], v.w = w[i + 3];
                *reinterpret_cast<uint4 *>(inv + first + i) = v;
            } else {
                for (uint32_t e = i; e < n_f && e < i + 4u; ++e) inv[first + e] = w[e];
            }
        }
        __syncthreads(); // `w` is refilled by the next window
    }
}

constexpr int kLcpChunk = 64;

// length of the common prefix of text[a..] and text[b..], starting the comparison at offset l
__device__ __forceinline__ uint32_t extend_match(const uint8_t *__restrict__ T, uint32_t a, uint32_t b, uint32_t l)
{
    for (;;) {
        uint64_t a0, a1, b0, b1;
        load_bytes16(T, (uint64_t)a + l, a0, a1);
        load_bytes16(T, (uint64_t)b + l, b0, b1);
        const uint64_t x0 = a0 ^ b0, x1 = a1 ^ b1;
        if (x0) return l + (uint32_t)((__ffsll((unsigned long long)x0) - 1) >> 3);
        if (x1) return l + 8u + (uint32_t)((__ffsll((unsigned long long)x1) - 1) >> 3);
        l += 16u;
    }
}

// PLCP at the chunk starts c = first + k * step (k = 0, 1, ...), c < chunks: text position i = c * kLcpChunk is compared
// with the suffix in front of it in the suffix array, from offset max(0, plcp[c - back] - back * kLcpChunk) on
// (Kasai's invariant over back * kLcpChunk positions); back == 0: from 0 (the first sample of all).
__global__ __launch_bounds__(kBlock) void lcp_samples_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ sa,
                                                             const uint32_t *__restrict__ inv, uint64_t chunks,
                                                             uint64_t first, uint64_t step, uint64_t back,
                                                             uint32_t *__restrict__ plcp)
{
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t c = first + k * step;
    if (c >= chunks) return;
    const uint64_t i = c * kLcpChunk;
    const uint32_t j = inv[i];
    uint32_t l = 0;
    if (j != 0) {
        if (back) {
            const uint64_t known = plcp[c - back], dist = back * kLcpChunk;
            l = known > dist ? (uint32_t)(known - dist) : 0u;
        }
        l = extend_match(T, (uint32_t)i, sa[j - 1], l);
    }
    plcp[c] = l;
}

__global__ __launch_bounds__(kBlock) void lcp_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ sa,
                                                     const uint32_t *__restrict__ inv, uint64_t N,
                                                     const uint32_t *__restrict__ plcp, uint32_t *__restrict__ lcp)
{
    const uint64_t chunk = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t i0 = chunk * kLcpChunk;
    if (i0 >= N) return;
    const uint64_t i1 = i0 + kLcpChunk < N ? i0 + kLcpChunk : N;
    uint32_t l = plcp[chunk]; // what the first position of the chunk has in common with its predecessor: known
    // Kasai's loop carries only l: which suffix stands in front of position i in the suffix array (j = inv[i], k = sa[j - 1])
    // does not depend on it.  Eight positions' ranks and predecessors are fetched together -- two of a position's three
    // random accesses leave the dependent chain (round 4; one position at a time every step waited for inv, then sa, then
    // the text: 23 ms for 2^28 positions, three memory latencies a position).
    constexpr int kAhead = 8;
    for (uint64_t b = i0; b < i1; b += kAhead) {
        uint32_t jj[kAhead], kk[kAhead];
#pragma unroll
        for (int e = 0; e < kAhead; ++e) jj[e] = b + e < i1 ? inv[b + e] : 0u;
#pragma unroll
        for (int e = 0; e < kAhead; ++e) kk[e] = jj[e] ? sa[jj[e] - 1u] : 0u;
#pragma unroll
        for (int e = 0; e < kAhead; ++e) {
            const uint64_t i = b + e;
            if (i >= i1) break;
            if (jj[e] == 0) { // the sentinel suffix has no predecessor: lcp[0] = 0 (suffix_array.c:74-75)
                lcp[0] = 0;
                l = 0;
                continue;
            }
            // text[n] = 0 differs from every symbol, so the comparison stops before either suffix ends
            l = extend_match(T, (uint32_t)i, kk[e], l);
            lcp[jj[e]] = l;
            l = l > 0 ? l - 1 : 0;
        }
    }
}

// ---- round 5: LCP through Phi -------------------------------------------------------------------------------------
// Kasai's loop above pays three random accesses a position: sa[inv[i] - 1], the text behind it, lcp[inv[i]] (24 ms at 2^28,
// the chip's random-sector rate).  Two of them are permutations and can be streamed: phi[sa[j]] = sa[j - 1] (the suffix in
// front of every text position's own, Karkkainen / Manzini / Puglisi's Phi array) is one permutation scatter -- the
// inverse's three passes with sa[j - 1] as the value --, the loop then reads phi[i] and writes plcp[i] in text order (one
// random access left: the text behind phi[i]), and lcp[inv[i]] = plcp[i] is a second scatter.  Same chunks, same samples,
// same invariant (plcp[i + 1] >= plcp[i] - 1).
constexpr uint32_t kNoPhi = 0xFFFFFFFFu;
__global__ __launch_bounds__(kBlock) void plcp_samples_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ phi, uint64_t chunks,
                                                              uint64_t first, uint64_t step, uint64_t back, uint32_t *__restrict__ plcp)
{
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t c = first + k * step;
    if (c >= chunks) return;
    const uint64_t i = c * kLcpChunk;
    const uint32_t pred = phi[i];
    uint32_t l = 0;
    if (pred != kNoPhi) {
        if (back) {
            const uint64_t known = plcp[c - back], dist = back * kLcpChunk;
            l = known > dist ? (uint32_t)(known - dist) : 0u;
        }
        l = extend_match(T, (uint32_t)i, pred, l);
    }
    plcp[c] = l;
}

// First a look at the first kPlcpHead symbols behind every position and its predecessor, all positions at once (nothing
// depends on the position before: every load of a wave is in flight together, the random-sector rate instead of a chain of
// dependent misses per thread -- 13 ms of the Phi form's 20 at 2^28 were that chain); min(common prefix, kPlcpHead) goes to
// plcp[i].  On ordinary text that IS the value nearly everywhere; the chunked loop below then walks only the positions that
// matched to the end of the head, carrying Kasai's invariant as before.
constexpr uint32_t kPlcpHead = 32;
__global__ __launch_bounds__(kBlock) void plcp_head_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ phi, uint64_t N,
                                                           uint32_t *__restrict__ plcp, uint8_t *__restrict__ chunk_long)
{
    static_assert(kLcpChunk == kWave, "a wave looks at one chunk of the loop below: its ballot is the chunk's flag");
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t pred = i < N ? phi[i] : kNoPhi;
    uint32_t l = 0;
    if (pred != kNoPhi) {
        uint64_t a[4], b[4];
        load_bytes32(T, i, a);
        load_bytes32(T, (uint64_t)pred, b);
        l = kPlcpHead;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            const uint64_t x = a[k] ^ b[k];
            if (x) l = 8u * (uint32_t)k + (uint32_t)((__ffsll((unsigned long long)x) - 1) >> 3);
        }
    }
    if (i < N) plcp[i] = l;
    // whether the chunk holds a position the loop below has to walk on from (on ordinary text next to none does: its threads
    // then leave at once instead of reading 64 positions' phi and plcp at a stride of 256 bytes a lane -- 2.5 ms at 2^28)
    const uint64_t any_long = __ballot((i < N && l >= kPlcpHead) ? 1 : 0);
    if (lane_id() == 0 && i < N) chunk_long[i / kLcpChunk] = any_long ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void plcp_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ phi, uint64_t N,
                                                      const uint32_t *__restrict__ plcp_samples, uint32_t *__restrict__ plcp,
                                                      const uint8_t *__restrict__ chunk_long)
{
    const uint64_t chunk = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t i0 = chunk * kLcpChunk;
    if (i0 >= N || !chunk_long[chunk]) return; // (no position of the chunk matched its head to the end: the head kernel's values stand)
    const uint64_t i1 = i0 + kLcpChunk < N ? i0 + kLcpChunk : N;
    uint32_t carry = plcp_samples[chunk]; // a lower bound of the value at the next position (the chunk's first: the value itself)
    constexpr int kAhead = 8;
    for (uint64_t b = i0; b < i1; b += kAhead) {
        uint32_t kk[kAhead], head[kAhead];
#pragma unroll
        for (int e = 0; e < kAhead; ++e) kk[e] = b + e < i1 ? phi[b + e] : kNoPhi, head[e] = b + e < i1 ? plcp[b + e] : 0u;
#pragma unroll
        for (int e = 0; e < kAhead; ++e) {
            const uint64_t i = b + e;
            if (i >= i1) break;
            uint32_t l = head[e]; // (the sentinel suffix has no predecessor: the head kernel left 0, suffix_array.c:74-75)
            if (l >= kPlcpHead) { // the head matched to its end: on from there, or from what the position before promises
                l = extend_match(T, (uint32_t)i, kk[e], carry > kPlcpHead ? carry : kPlcpHead);
                plcp[i] = l;
            }
            carry = l > 0 ? l - 1 : 0;
        }
    }
}

// One thread per pattern.  Every step is a dependent look-up in a table of gigabytes: the first eight steps touch
// at most 4^8 rows (1.3 MB for DNA: L2-resident whatever the order of the patterns), from step 14 on (256 Mi
// positions) every step is a random 64-byte sector.  10^7 patterns x 30 symbols: 6.9 ms = 87 G look-ups/s = 5.5 TB/s
// of sectors, the chip's random-access rate; taking the patterns in the order of their last symbols (a radix sort of
// the pattern numbers) did not change the kernel's time and cost the sort.
__global__ __launch_bounds__(kBlock) void bwt_exact_search_kernel(const uint32_t *__restrict__ c_table,
                                                                  const uint32_t *__restrict__ o_table, uint64_t N,
                                                                  uint32_t sigma, const uint8_t *__restrict__ patterns,
                                                                  const uint32_t *__restrict__ offsets, uint32_t count,
                                                                  uint32_t *__restrict__ out_l,
                                                                  uint32_t *__restrict__ out_r)
{
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= count) return;
    const uint32_t begin = offsets[q], m = offsets[q + 1] - begin;
    uint32_t L = 0, R = (uint32_t)N;
    if ((uint64_t)m > N) { // bwt.c:178-180: a pattern longer than the text cannot match
        R = 0;
        L = 1;
    }
    // the pattern 16 symbols at a time, from its end (one load per 16 steps)
    for (uint32_t done = 0; done < m && L < R;) {
        const uint32_t take = m - done < 16u ? m - done : 16u;
        const uint64_t at = (uint64_t)begin + m - done - take; // symbols at .. at + take - 1, consumed from the last one
        uint64_t w0 = 0, w1 = 0;
        if (take == 16u) {
            load_bytes16(patterns, at, w0, w1);
        } else {
            for (uint32_t e = 0; e < take; ++e) {
                const uint64_t b = patterns[at + e];
                if (e < 8u) w0 |= b << (8u * e);
                else w1 |= b << (8u * (e - 8u));
            }
        }
        for (uint32_t e = take; e-- > 0 && L < R;) {
            const uint32_t a = (uint32_t)((e < 8u ? w0 >> (8u * e) : w1 >> (8u * (e - 8u))) & 0xFFull);
            if (a == 0 || a >= sigma) { // the reference asserts 0 < a < alphabet_size: no match here
                L = 1;
                R = 0;
                break;
            }
            L = c_table[a] + o_table[(uint64_t)L * sigma + a];
            R = c_table[a] + o_table[(uint64_t)R * sigma + a];
        }
        done += take;
    }
    out_l[q] = L;
    out_r[q] = R;
}

} // namespace sx

using namespace sx;

#ifndef SX_INVERSE_TWO_PASS_FROM
#define SX_INVERSE_TWO_PASS_FROM (1ull << 23) // (shorter arrays' targets fit the caches as they are)
#endif
bool sx_scatter_permutation_applies(uint64_t N) { return N >= SX_INVERSE_TWO_PASS_FROM; }
size_t sx_scatter_permutation_cursor_words() { return kInvMaxParts; }
// out[targets[i]] = values[i] (values == null: i) for a permutation `targets` of [0, N), in the two passes above.
// pairs: N x 8 bytes, cursor: sx_scatter_permutation_cursor_words() words, bad: one word the caller has zeroed (counts
// what is not a permutation)
int sx_scatter_permutation(sx_ctx *ctx, const uint32_t *targets, const uint32_t *values, uint64_t N, uint32_t *out, void *pairs_scratch,
                           uint32_t *cursor, uint32_t *bad, int kclass)
{
    // (2^28 entries, same box: windows of 2^16 ... 2^21 targets 6.1, 6.2, 5.2, 5.0, 5.3, 6.2 ms; the second pass in plain
    //  workgroup order instead of XCD by XCD 5.4; the single-pass kernel 9.9)
    uint32_t wbits = kInvWindowBits;
    const int by_xcd = 1;
    while (((N + (1ull << wbits) - 1) >> wbits) > kInvMaxParts) ++wbits;
    const uint32_t nparts = (uint32_t)((N + (1ull << wbits) - 1) >> wbits);
    uint2 *pairs = (uint2 *)pairs_scratch;
    SX_CHECK(hipMemsetAsync(cursor, 0, (size_t)nparts * 4, ctx->stream));
    uint32_t grid = sx_div_up(N, kInvTile);
    if (grid > 4096) grid = 4096;
    sx_launch(ctx, kclass, N * (values ? 16 : 12), inverse_partition_kernel, dim3(grid), dim3(kInvThreads), targets, values, 0u, N, nparts, wbits,
              cursor, pairs, bad);
    const uint64_t chunks = (N + (uint64_t)kBlock * 8 - 1) / ((uint64_t)kBlock * 8);
    sx_launch(ctx, kclass, N * 12, inverse_apply_kernel, dim3((uint32_t)(((chunks + 7) / 8) * 8)), dim3(kBlock), (const uint2 *)pairs, N, out,
              by_xcd, (const uint32_t *)bad);
    return 0;
}

// whether an array of N entries takes the three-pass form (the fine windows a tile's pairs can fall into must fit the refine
// kernel's counters), and the bits of its coarse windows
static bool permute_three_passes(uint64_t N, uint32_t &wbits)
{
    wbits = kInvFineBits + 6u;
    while (((N + (1ull << wbits) - 1) >> wbits) > 128u) ++wbits;
    return sx_scatter_permutation_applies(N) && wbits >= kInvFineBits &&
           ((((uint64_t)kInvTile >> wbits) + 2) << (wbits - kInvFineBits)) <= kInvFineMax;
}

// out[targets[i]] = i (values null), values[i], or values[i - 1] (values_back; entry 0: 0xFFFFFFFF) for a permutation `targets`
// of [0, N); an array that is no permutation is reported (SX_E_ARG) without a store leaving `out`
static int permute_dev(sx_ctx *ctx, const uint32_t *d_sa, const uint32_t *values, uint32_t values_back, uint64_t N, uint32_t *d_inv)
{
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_BWT, 4096));
    uint32_t *bad = (uint32_t *)ctx->slab[SX_SLAB_BWT].p;
    SX_CHECK(hipMemsetAsync(bad, 0, sizeof(uint32_t), ctx->stream));
    if (sx_scatter_permutation_applies(N)) {
        // three passes: coarse windows, their fine windows, the fine windows through LDS (above).  The coarse windows are
        // chosen so that BOTH dealing passes write long runs: at most 128 of them (a tile of 8192 entries leaves 64-pair,
        // 512-byte runs; with the two-pass form's 512 windows of 2^19 targets the first pass wrote 128-byte runs at
        // arbitrary offsets and took 3.1 of the inverse's 4.5 ms for a third of its traffic), each of at least 64 fine ones
        uint32_t wbits = 0;
        const bool three = permute_three_passes(N, wbits);
        const uint32_t nparts = (uint32_t)((N + (1ull << wbits) - 1) >> wbits);
        const uint64_t nfine = (N + (1ull << kInvFineBits) - 1) >> kInvFineBits;
        const size_t pairs_b = (N * sizeof(uint2) + 255) & ~(size_t)255, cur_b = ((size_t)kInvMaxParts * 4 + 255) & ~(size_t)255,
                     fine_b = ((size_t)nfine * 4 + 255) & ~(size_t)255;
        SX_TRY(sx_slab_ensure(ctx, SX_SLAB_SORT, 2 * pairs_b + cur_b + fine_b + 512));
        char *base = (char *)ctx->slab[SX_SLAB_SORT].p;
        uint32_t *cursor = (uint32_t *)base, *fine_cursor = (uint32_t *)(base + cur_b);
        uint2 *pairs = (uint2 *)(base + cur_b + fine_b), *pairs2 = (uint2 *)(base + cur_b + fine_b + pairs_b);
        // (the fine windows a tile's pairs can fall into must fit the refine kernel's counters)
        if (three) {
            SX_CHECK(hipMemsetAsync(cursor, 0, (size_t)nparts * 4, ctx->stream));
            SX_CHECK(hipMemsetAsync(fine_cursor, 0, (size_t)nfine * 4, ctx->stream));
            uint32_t grid = sx_div_up(N, kInvTile);
            if (grid > 4096) grid = 4096;
            sx_launch(ctx, SX_KC_LCP, N * (values ? 16 : 12), inverse_partition_kernel, dim3(grid), dim3(kInvThreads), d_sa, values, values_back, N,
                      nparts, wbits, cursor, pairs, bad);
            sx_launch(ctx, SX_KC_LCP, N * 16, inverse_refine_kernel, dim3(grid), dim3(kInvThreads), (const uint2 *)pairs, N, wbits, fine_cursor,
                      pairs2, bad);
            uint32_t wgrid = (uint32_t)(nfine < 65536 ? nfine : 65536);
            sx_launch(ctx, SX_KC_LCP, N * 12, inverse_window_kernel, dim3(wgrid), dim3(kInvWinThreads), (const uint2 *)pairs2, N, d_inv,
                      (const uint32_t *)bad);
        } else {
            if (values_back) return sx_fail_msg(ctx, SX_E_INTERNAL, "permute: the two-pass form takes no shifted values");
            SX_TRY(sx_scatter_permutation(ctx, d_sa, values, N, d_inv, pairs, cursor, bad, SX_KC_LCP));
        }
    } else {
        if (values) return sx_fail_msg(ctx, SX_E_INTERNAL, "permute: short arrays take the plain kernel");
        sx_launch(ctx, SX_KC_LCP, N * 8, inverse_kernel, dim3(sx_div_up(N, kBlock)), dim3(kBlock), d_sa, N, d_inv, bad);
    }
    uint32_t h_bad = 0;
    SX_TRY(sx_readback(ctx, bad, 1, &h_bad));
    if (h_bad) return sx_fail_msg(ctx, SX_E_ARG, "sa is not a permutation of [0, N)");
    return 0;
}

static int inverse_dev(sx_ctx *ctx, const uint32_t *d_sa, uint64_t N, uint32_t *d_inv) { return permute_dev(ctx, d_sa, nullptr, 0u, N, d_inv); }

static int lcp_dev(sx_ctx *ctx, const uint8_t *d_text, const uint32_t *d_sa, uint64_t N, uint32_t *d_inv_opt,
                   uint32_t *d_lcp)
{
    const uint64_t n = N - 1;
    const size_t text_b = (n + 128 + 255) & ~(size_t)255, arr_b = ((size_t)N * 4 + 255) & ~(size_t)255;
    uint32_t wbits_unused = 0;
    // Phi form (round 5): arrays that take the three-pass scatter, up to 2^31 entries (its scratch: two more arrays of N words)
    const bool by_phi = permute_three_passes(N, wbits_unused) && N <= (1ull << 31);
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_N, text_b + (d_inv_opt ? 0 : arr_b) + (by_phi ? 2 * arr_b : 0) + 256));
    uint8_t *T = (uint8_t *)ctx->slab[SX_SLAB_N].p; // padded copy: text[n] = 0 and zeros behind it
    uint32_t *inv = d_inv_opt ? d_inv_opt : (uint32_t *)((char *)ctx->slab[SX_SLAB_N].p + text_b);
    uint32_t *phi = (uint32_t *)((char *)ctx->slab[SX_SLAB_N].p + text_b + (d_inv_opt ? 0 : arr_b)), *plcp_full = (uint32_t *)((char *)phi + arr_b);
    if (n) SX_CHECK(hipMemcpyAsync(T, d_text, n, hipMemcpyDeviceToDevice, ctx->stream));
    SX_CHECK(hipMemsetAsync(T + n, 0, text_b - n, ctx->stream));
    SX_TRY(inverse_dev(ctx, d_sa, N, inv));
    if (by_phi) SX_TRY(permute_dev(ctx, d_sa, d_sa, 1u, N, phi)); // phi[sa[j]] = sa[j - 1]
    const uint64_t chunks = (N + kLcpChunk - 1) / kLcpChunk;
    // samples: chunk 0 from scratch, then the chunks at odd multiples of S, S halving, each from the sample S to its left
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_BWT, chunks * 5 + 4096 + 256));
    uint32_t *plcp = (uint32_t *)((char *)ctx->slab[SX_SLAB_BWT].p + 4096); // (the first page holds inverse_dev's counter)
    uint8_t *chunk_long = (uint8_t *)(plcp + chunks);                        // a flag a chunk: some position matched its 32-symbol head
    if (by_phi)
        sx_launch(ctx, SX_KC_LCP, 64, plcp_samples_kernel, dim3(1), dim3(kBlock), (const uint8_t *)T, (const uint32_t *)phi, (uint64_t)1, (uint64_t)0,
                  (uint64_t)1, (uint64_t)0, plcp);
    else
        sx_launch(ctx, SX_KC_LCP, 64, lcp_samples_kernel, dim3(1), dim3(kBlock), (const uint8_t *)T, d_sa, (const uint32_t *)inv, (uint64_t)1,
                  (uint64_t)0, (uint64_t)1, (uint64_t)0, plcp);
    uint64_t S = 1;
    while (S * 2 < chunks) S *= 2;
    for (; S >= 1; S /= 2) {
        const uint64_t count = chunks > S ? (chunks - S + 2 * S - 1) / (2 * S) : 0; // chunks S, 3S, 5S, ... below `chunks`
        if (!count) continue;
        if (by_phi)
            sx_launch(ctx, SX_KC_LCP, count * 80, plcp_samples_kernel, dim3(sx_div_up(count, kBlock)), dim3(kBlock), (const uint8_t *)T,
                      (const uint32_t *)phi, chunks, S, 2 * S, S, plcp);
        else
            sx_launch(ctx, SX_KC_LCP, count * 80, lcp_samples_kernel, dim3(sx_div_up(count, kBlock)), dim3(kBlock), (const uint8_t *)T,
                      d_sa, (const uint32_t *)inv, chunks, S, 2 * S, S, plcp);
    }
    if (by_phi) {
        sx_launch(ctx, SX_KC_LCP, N * (8 + 64), plcp_head_kernel, dim3(sx_div_up(N, kBlock)), dim3(kBlock), (const uint8_t *)T, (const uint32_t *)phi, N,
                  plcp_full, chunk_long);
        sx_launch(ctx, SX_KC_LCP, N * 12, plcp_kernel, dim3(sx_div_up(chunks, kBlock)), dim3(kBlock), (const uint8_t *)T, (const uint32_t *)phi, N,
                  (const uint32_t *)plcp, plcp_full, (const uint8_t *)chunk_long);
        SX_TRY(permute_dev(ctx, inv, plcp_full, 0u, N, d_lcp)); // lcp[inv[i]] = plcp[i]
    } else {
        sx_launch(ctx, SX_KC_LCP, N * 14, lcp_kernel, dim3(sx_div_up(chunks, kBlock)), dim3(kBlock), (const uint8_t *)T, d_sa,
                  (const uint32_t *)inv, N, (const uint32_t *)plcp, d_lcp);
    }
    return 0;
}

extern "C" {

int sx_sa_inverse_dev(sx_ctx *ctx, const uint32_t *d_sa, uint64_t N, uint32_t *d_inv_out)
{
    if (!ctx || !d_sa || !d_inv_out || N == 0 || N > 0xFFFFFFFFull) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    SX_TRY(inverse_dev(ctx, d_sa, N, d_inv_out));
    return sx_sync(ctx);
}

int sx_sa_lcp_dev(sx_ctx *ctx, const uint8_t *d_text, const uint32_t *d_sa, uint64_t N, uint32_t *d_inv_out,
                  uint32_t *d_lcp_out)
{
    if (!ctx || !d_sa || !d_lcp_out || N == 0 || N > 0xFFFFFFFFull || (N > 1 && !d_text)) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    SX_TRY(lcp_dev(ctx, d_text, d_sa, N, d_inv_out, d_lcp_out));
    return sx_sync(ctx);
}

int sx_sa_inverse_lcp(sx_ctx *ctx, const uint8_t *text, const uint32_t *sa, uint64_t N, uint32_t *inv_out,
                      uint32_t *lcp_out)
{
    if (!ctx || !sa || N == 0 || N > 0xFFFFFFFFull || (N > 1 && !text) || (!inv_out && !lcp_out)) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    const uint64_t n = N - 1;
    const size_t text_b = (n + 255) & ~(size_t)255, arr_b = (N * 4 + 255) & ~(size_t)255;
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_IO, text_b + 3 * arr_b + 1024));
    char *base = (char *)ctx->slab[SX_SLAB_IO].p;
    uint8_t *d_text = (uint8_t *)base;
    uint32_t *d_sa = (uint32_t *)(base + text_b + 256), *d_inv = (uint32_t *)(base + text_b + 256 + arr_b),
             *d_lcp = (uint32_t *)(base + text_b + 256 + 2 * arr_b);
    if (n) SX_CHECK(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, ctx->stream));
    SX_CHECK(hipMemcpyAsync(d_sa, sa, N * 4, hipMemcpyHostToDevice, ctx->stream));
    if (lcp_out) SX_TRY(lcp_dev(ctx, d_text, d_sa, N, d_inv, d_lcp));
    else SX_TRY(inverse_dev(ctx, d_sa, N, d_inv));
    if (inv_out) SX_CHECK(hipMemcpyAsync(inv_out, d_inv, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (lcp_out) SX_CHECK(hipMemcpyAsync(lcp_out, d_lcp, N * 4, hipMemcpyDeviceToHost, ctx->stream));
    return sx_sync(ctx);
}

int sx_bwt_exact_search_dev(sx_ctx *ctx, const uint32_t *d_c_table, const uint32_t *d_o_table, uint64_t N,
                            uint32_t sigma, const uint8_t *d_patterns, const uint32_t *d_offsets, uint32_t count,
                            uint32_t *d_l_out, uint32_t *d_r_out)
{
    if (!ctx || !d_c_table || !d_o_table || !d_offsets || !d_l_out || !d_r_out || N == 0 || N > 0xFFFFFFFFull ||
        sigma < 1 || sigma > 256)
        return SX_E_ARG;
    if (count == 0) return 0;
    SX_CHECK(hipSetDevice(ctx->device));
    sx_launch(ctx, SX_KC_SEARCH, 0, bwt_exact_search_kernel, dim3(sx_div_up(count, kBlock)), dim3(kBlock), d_c_table,
              d_o_table, N, sigma, d_patterns, d_offsets, count, d_l_out, d_r_out);
    return sx_sync(ctx);
}

} // extern "C"
