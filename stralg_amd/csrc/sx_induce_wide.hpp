// sx_induce_wide.hpp -- the induced-sort passes over more than 8 buckets: a round as a radix pass (count, offsets, scatter), and every bucket's other-region round up front (bigram counts)
// (included by sx_induce.hip, which holds the passes' host side; one translation unit)
#pragma once
#include "sx_induce_common.hpp"

namespace sx {

// ---- large rounds of wide alphabets (more than 8 buckets) ------------------------------------
// A round is a stable split by one symbol of up to 8 bits: a radix pass (sx_radix.hip) whose "digit bases" are the
// bucket cursors and whose pairs are (window, position) instead of (key, value).  So it is built like one: tiles
// of 8192 entries, a tile-major count table ([tile][256]: every kernel touches whole 1 KiB rows), the ranking with
// four vector instructions per symbol bit, and the tile's output staged in LDS in bucket order so that every
// bucket's run leaves as a contiguous block.  (The kernels above -- 2048-entry tiles, a bucket-major table read with
// a 64-byte sector per count, one look-back thread per bucket and tile -- took 92 of 142 ms of a 1 GiB text of 255
// symbols, whose buckets of 2 M entries they visit one after the other: 50 MB moved in 100 us and more.)
#ifndef SX_WIDE_ITEMS
#define SX_WIDE_ITEMS 16 // entries a thread and tile: 8192-entry tiles, taken in two steps by the scatter (8: 4096-entry tiles --
                         // 1 GiB of bytes 100 against 102 ms, but 12 symbols 53.4 against 49.9: long rounds want the larger tile)
#endif
constexpr int kWideThreads = 512, kWideWaves = kWideThreads / kWave, kWideItems = SX_WIDE_ITEMS;
constexpr int kWideTile = kWideThreads * kWideItems;
constexpr uint32_t kWideChunk = 256;                  // tiles per chunk of the column sums (long rounds)

// the entries [a, b) of the source arrays counted by destination bucket into the LDS row h (zeroed by the caller; a
// barrier on either side is the caller's): symbol bytes where the source has them, windows otherwise
template <class WT>
__device__ __forceinline__ void wide_count_range(const WT *__restrict__ srcW, const uint8_t *__restrict__ srcB, uint32_t a, uint32_t b,
                                                 int mode, uint32_t c, const wnd_cfg &cfg, uint32_t *h, bool aligned)
{
    if (srcB) {
        for (uint64_t q = (uint64_t)(a >> 4) + threadIdx.x; q * 16u < b; q += kWideThreads) {
            const uint64_t e0 = q * 16u;
            uint32_t S[4] = {0, 0, 0, 0};
            if (aligned && e0 >= a && e0 + 16u <= b) {
                load_quad(reinterpret_cast<const uint32_t *>(srcB + e0), S);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (e0 + e >= a && e0 + e < b) S[e >> 2] |= (uint32_t)srcB[e0 + e] << (8 * (e & 3));
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t ch = (S[e >> 2] >> (8 * (e & 3))) & 0xFFu;
                if (ch != 0 && induce_accept(ch, c, mode)) atomicAdd(&h[ch], 1u);
            }
        }
    } else {
        for (uint64_t q = (uint64_t)(a >> 2) + threadIdx.x; q * 4u < b; q += kWideThreads) {
            const uint64_t e0 = q * 4u;
            WT W[4] = {0, 0, 0, 0};
            if (aligned && e0 >= a && e0 + 4u <= b) {
                load_quad(srcW + e0, W);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e0 + e >= a && e0 + e < b) W[e] = srcW[e0 + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (wnd_count<WT>(W[e]) != 0) { // the entry for position 0 is the only one stored with an empty window
                    const uint32_t ch = wnd_first<WT>(W[e], cfg);
                    if (induce_accept(ch, c, mode)) atomicAdd(&h[ch], 1u);
                }
            }
        }
    }
}

template <class WT>
__global__ __launch_bounds__(kWideThreads) void induce_wide_count_kernel(const WT *__restrict__ srcW,
                                                                      const uint8_t *__restrict__ srcB,
                                                                      const uint32_t *__restrict__ range_in, int rev,
                                                                      int mode, uint32_t c, wnd_cfg cfg,
                                                                      uint32_t *__restrict__ hist /* [tile][256] */,
                                                                      uint32_t min_len)
{
    __shared__ uint32_t h[256];
    const uint32_t lo = range_in[0], len = range_in[1] - lo;
    if (len <= min_len) return;
    const uint32_t ntiles = (len + kWideTile - 1) / kWideTile;
    const bool aligned = srcB ? ((uintptr_t)srcB & 15u) == 0 : ((uintptr_t)srcW & 15u) == 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // uniform per workgroup
        if (threadIdx.x < 256) h[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t tile0 = tile * (uint32_t)kWideTile;
        const uint32_t cnt = len - tile0 < (uint32_t)kWideTile ? len - tile0 : (uint32_t)kWideTile;
        const uint32_t a = rev ? lo + len - tile0 - cnt : lo + tile0, b = a + cnt; // the tile's entries: [a, b), any order
        wide_count_range<WT>(srcW, srcB, a, b, mode, c, cfg, h, aligned);
        __syncthreads();
        if (threadIdx.x < 256) hist[(uint64_t)tile * 256 + threadIdx.x] = h[threadIdx.x];
        __syncthreads();
    }
}

// counts -> entries of earlier tiles, per bucket; the cursors move on; the range appended to bucket c.
// Eight workgroups, 32 buckets (128 bytes of every 1 KiB row) each: thread (g, d) owns bucket d over the g-th of 32
// groups of tiles -- 16 rows in flight --, sums it, the groups' sums meet in LDS, and the second walk writes the prefixes.
// Round 4: one workgroup of 1024 threads (four groups of tiles) took 16 us for the 450 tiles of a byte text's round,
// a fifth of that bucket's whole chain of launches; a thread's walk is now an eighth as long (rounds of up to 4096
// tiles; longer ones take the chunked form below).
#ifndef SX_WIDE_OFF_GROUPS
#define SX_WIDE_OFF_GROUPS 32 // (the CPU test harness: 2)
#endif
constexpr int kWideOffGroups = SX_WIDE_OFF_GROUPS, kWideOffCols = 32, kWideOffThreads = kWideOffCols * kWideOffGroups;
constexpr uint32_t kWideOffMaxTiles = 4096;
__global__ __launch_bounds__(kWideOffThreads) void induce_wide_offsets_kernel(uint32_t *__restrict__ hist,
                                                                     const uint32_t *__restrict__ range_in,
                                                                     uint32_t *__restrict__ range_out,
                                                                     const uint32_t *__restrict__ cursor_cur,
                                                                     uint32_t *__restrict__ cursor_nxt, int dir, uint32_t c,
                                                                     uint32_t min_len, int only_form)
{
    __shared__ uint32_t gsum[kWideOffGroups][kWideOffCols];
    const uint32_t len = range_in[1] - range_in[0];
    const uint32_t dd = threadIdx.x % kWideOffCols, g = threadIdx.x / kWideOffCols;
    const uint32_t d = blockIdx.x * kWideOffCols + dd; // gridDim.x = 256 / kWideOffCols
    if (len <= min_len) {
        if (only_form && g == 0) { // (no chained launch follows: an empty range is carried on here)
            cursor_nxt[d] = cursor_cur[d];
            if (d == c && range_out) range_out[0] = range_out[1] = range_in[1];
        }
        return;
    }
    const uint32_t ntiles = (len + kWideTile - 1) / kWideTile;
    const uint32_t per = (ntiles + kWideOffGroups - 1) / kWideOffGroups;
    const uint32_t t0 = g * per < ntiles ? g * per : ntiles, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    constexpr int kBatch = 16;
    uint32_t sum = 0;
    for (uint32_t tb = t0; tb < t1; tb += kBatch) {
        uint32_t x[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) x[i] = tb + i < t1 ? hist[(uint64_t)(tb + i) * 256 + d] : 0u;
#pragma unroll
        for (int i = 0; i < kBatch; ++i) sum += x[i];
    }
    gsum[g][dd] = sum;
    __syncthreads();
    uint32_t run = 0, all = 0;
    for (int gg = 0; gg < kWideOffGroups; ++gg) {
        const uint32_t x = gsum[gg][dd];
        if ((uint32_t)gg < g) run += x;
        all += x;
    }
    for (uint32_t tb = t0; tb < t1; tb += kBatch) {
        uint32_t x[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) x[i] = tb + i < t1 ? hist[(uint64_t)(tb + i) * 256 + d] : 0u;
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            if (tb + i < t1) hist[(uint64_t)(tb + i) * 256 + d] = run;
            run += x[i];
        }
    }
    if (g == 0) {
        const uint32_t cur = cursor_cur[d];
        cursor_nxt[d] = dir > 0 ? cur + all : cur - all;
        if (d == c && range_out) {
            range_out[0] = dir > 0 ? cur : cur - all;
            range_out[1] = dir > 0 ? cur + all : cur;
        }
    }
}

// The same for long rounds in three launches: column sums of chunks of 256 tiles, their prefix (one workgroup; also
// the cursors and the range), and the prefix inside every chunk.
__global__ __launch_bounds__(kBlock) void induce_wide_colsum_kernel(const uint32_t *__restrict__ hist,
                                                                    const uint32_t *__restrict__ range_in,
                                                                    uint32_t *__restrict__ sums, uint32_t min_len)
{
    const uint32_t len = range_in[1] - range_in[0];
    if (len <= min_len) return;
    const uint32_t ntiles = (len + kWideTile - 1) / kWideTile;
    const uint32_t t0 = blockIdx.x * kWideChunk;
    if (t0 >= ntiles) return;
    const uint32_t t1 = t0 + kWideChunk < ntiles ? t0 + kWideChunk : ntiles;
    constexpr int kBatch = 16;
    uint32_t sacc = 0;
    for (uint32_t tb = t0; tb < t1; tb += kBatch) {
        uint32_t x[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) x[i] = tb + i < t1 ? hist[(uint64_t)(tb + i) * 256 + threadIdx.x] : 0u;
#pragma unroll
        for (int i = 0; i < kBatch; ++i) sacc += x[i];
    }
    sums[(uint64_t)blockIdx.x * 256 + threadIdx.x] = sacc;
}
__global__ __launch_bounds__(kBlock) void induce_wide_bases_kernel(uint32_t *__restrict__ sums,
                                                                   const uint32_t *__restrict__ range_in,
                                                                   uint32_t *__restrict__ range_out,
                                                                   const uint32_t *__restrict__ cursor_cur,
                                                                   uint32_t *__restrict__ cursor_nxt, int dir, uint32_t c,
                                                                   uint32_t min_len, int only_form)
{
    const uint32_t len = range_in[1] - range_in[0];
    if (len <= min_len) {
        if (only_form) {
            cursor_nxt[threadIdx.x] = cursor_cur[threadIdx.x];
            if (threadIdx.x == c && range_out) range_out[0] = range_out[1] = range_in[1];
        }
        return;
    }
    const uint32_t ntiles = (len + kWideTile - 1) / kWideTile, nchunks = (ntiles + kWideChunk - 1) / kWideChunk;
    const uint32_t d = threadIdx.x;
    uint32_t run = 0;
    for (uint32_t cb = 0; cb < nchunks; ++cb) {
        const uint32_t x = sums[(uint64_t)cb * 256 + d];
        sums[(uint64_t)cb * 256 + d] = run;
        run += x;
    }
    const uint32_t cur = cursor_cur[d];
    cursor_nxt[d] = dir > 0 ? cur + run : cur - run;
    if (d == c && range_out) {
        range_out[0] = dir > 0 ? cur : cur - run;
        range_out[1] = dir > 0 ? cur + run : cur;
    }
}
__global__ __launch_bounds__(kBlock) void induce_wide_apply_kernel(uint32_t *__restrict__ hist,
                                                                   const uint32_t *__restrict__ range_in,
                                                                   const uint32_t *__restrict__ sums, uint32_t min_len)
{
    const uint32_t len = range_in[1] - range_in[0];
    if (len <= min_len) return;
    const uint32_t ntiles = (len + kWideTile - 1) / kWideTile;
    const uint32_t t0 = blockIdx.x * kWideChunk;
    if (t0 >= ntiles) return;
    const uint32_t t1 = t0 + kWideChunk < ntiles ? t0 + kWideChunk : ntiles;
    constexpr int kBatch = 16;
    uint32_t run = sums[(uint64_t)blockIdx.x * 256 + threadIdx.x];
    for (uint32_t tb = t0; tb < t1; tb += kBatch) {
        uint32_t x[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) x[i] = tb + i < t1 ? hist[(uint64_t)(tb + i) * 256 + threadIdx.x] : 0u;
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            if (tb + i < t1) hist[(uint64_t)(tb + i) * 256 + threadIdx.x] = run;
            run += x[i];
        }
    }
}

// A tile of 8192 entries is taken in kWideTile / (512 * ITEMS) steps of ITEMS entries a thread: 16 for 32-bit windows; 8 for
// 64-bit windows (alphabets of 17 symbols and more), whose 16 entries a thread did not fit 128 registers -- 76 of them
// were spilled, and a byte text's round of a single tile took 45 us.  A later step's entries go behind the earlier ones'.
// Windows that ran dry, refilled from the text with every load in flight before the first is used: need[k] says which
// of a thread's entries (positions val[k] >= 1) want one.  (wnd_fill under a branch per entry made a thread wait for
// each of its random reads in turn: 8 trips to memory of ~1 us each, half the time of a wide alphabet's round --
// measured with clock64 around the phases, tools/wide_probe.py.)
template <class WT, int ITEMS>
__device__ __forceinline__ void refill_windows(const uint8_t *__restrict__ T, const uint32_t (&val)[ITEMS], const bool (&need)[ITEMS],
                                               const wnd_cfg &cfg, WT (&wnd)[ITEMS])
{
    uint64_t lo[ITEMS], hi[ITEMS];
    uint32_t cnt[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        lo[k] = hi[k] = 0;
        cnt[k] = val[k] < cfg.CW ? val[k] : cfg.CW;
        if (need[k]) load_bytes16(T, (uint64_t)(val[k] - cnt[k]), lo[k], hi[k]);
    }
    // (decoded by a rolled loop, one entry after the other: eight inlined copies of wnd_from_bytes' unrolled forms, which
    //  the compiler interleaves, spilled a thousand registers)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        SX_SCHED_FENCE();
        if (need[k]) {
            WT acc = 0;
            uint64_t w = lo[k];
#pragma unroll 1
            for (uint32_t i = 0; i < cnt[k]; ++i) { // text[p - cnt] first: it ends with text[p - 1] in the lowest field
                acc = (acc << cfg.B) | (WT)((w & 0xFFull) - 1ull);
                w = i == 7u ? hi[k] : w >> 8;
            }
            wnd[k] = (acc << kCntBits) | (WT)cnt[k];
        }
    }
    SX_SCHED_FENCE();
}

#ifndef SX_TAIL_AHEAD
#define SX_TAIL_AHEAD 3u
#endif
constexpr uint32_t kTailAhead = SX_TAIL_AHEAD; // symbols a window keeps for the tail kernel's rounds (nearly every bucket ends within three); 0: off
// LDS of one scatter workgroup (the kernels below declare it and hand it to wide_scatter_tile)
template <int ITEMS> struct wide_scatter_lds {
    static constexpr int kSub = kWideThreads * ITEMS;
    uint64_t swnd[kSub]; // the step's output in bucket order: windows first, then reused for the positions;
                         // the per-wave counters live here while the entries are still in registers
    uint8_t sdig[kSub];  // bucket of every staged slot
    uint32_t goff[256];  // destination of the bucket's first staged slot, minus (plus) that slot
    uint32_t scan_lds[kWideWaves];
};

// One tile (`tile`-th of the range [lo, lo + len) in scan order) of a round: stable split of its entries by the first
// symbol of their windows.  pre: entries of earlier tiles for bucket t (threads t < 256); base_d: bucket t's cursor at the
// start of the round.  Ends with a barrier (the LDS may be reused at once).
template <class WT, int ITEMS>
__device__ __forceinline__ void wide_scatter_tile(wide_scatter_lds<ITEMS> &L, const uint32_t *__restrict__ srcP, const WT *__restrict__ srcW,
                                                  uint32_t lo, uint32_t len, uint32_t tile, int rev, int mode, uint32_t c, const wnd_cfg &cfg,
                                                  const uint8_t *__restrict__ T, uint32_t pre, uint32_t base_d, int dir,
                                                  uint32_t *__restrict__ SA, WT *__restrict__ WN, uint8_t *__restrict__ BW,
                                                  uint32_t refill_at = 0 /* windows left with at most this many symbols are read again */)
{
    constexpr int kSub = kWideThreads * ITEMS, kSteps = kWideTile / kSub;
    static_assert(kWideTile % kSub == 0 && kSub * 8 >= kWideWaves * 256 * 4, "steps tile the tile; the counters fit the staging image");
    uint64_t *swnd = L.swnd;
    uint8_t *sdig = L.sdig;
    uint32_t *goff = L.goff, *scan_lds = L.scan_lds;
    uint32_t *wcount = reinterpret_cast<uint32_t *>(swnd);
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
#ifdef SX_WIDE_PROBE
    long long pr[10];
    int pi = 0;
#define SX_PROBE() do { if (pi < 10) pr[pi++] = clock64(); } while (0)
#else
#define SX_PROBE() do { } while (0)
#endif
    for (int step = 0; step < kSteps; ++step) {
        const uint32_t step0 = tile * (uint32_t)kWideTile + (uint32_t)step * kSub;
        if (step0 >= len) break; // uniform
        SX_PROBE();
        for (int i = t; i < kWideWaves * 256; i += kWideThreads) wcount[i] = 0;
        __syncthreads();
        const uint32_t wave0 = step0 + (uint32_t)w * (kWave * ITEMS);
        uint32_t val[ITEMS], lpos[ITEMS]; // position - 1; [12:0] rank, then staged slot, [31:16] bucket, bit 15: taken
        WT wnd[ITEMS];
        // (every load of the step is issued before the first is looked at: with the look inside the loop a thread
        //  waited for each of its 2 * ITEMS loads in turn -- 20 000 of a step's 35 000 cycles)
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const uint32_t i = wave0 + (uint32_t)k * kWave + lane;
            const uint32_t idx = lo + (i < len ? (rev ? len - 1u - i : i) : 0u);
            val[k] = srcP[idx];
            wnd[k] = srcW[idx];
        }
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const uint32_t i = wave0 + (uint32_t)k * kWave + lane;
            const uint32_t p = i < len ? val[k] : 0u;
            const WT ww = wnd[k];
            val[k] = 0;
            wnd[k] = 0;
            bool ok = false;
            uint32_t dig = 0;
            if (p != 0) {
                dig = wnd_first<WT>(ww, cfg);
                ok = induce_accept(dig, c, mode);
                val[k] = p - 1u;
                wnd[k] = wnd_pop<WT>(ww, cfg);
            }
            lpos[k] = ok ? (dig & 0xFFu) << 16 | 0x8000u : 0u;
        }
        SX_PROBE();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const bool ok = (lpos[k] & 0x8000u) != 0;
            lpos[k] |= wave_rank_inorder<8, false>(lpos[k] >> 16, ok, wcount + w * 256);
        }
        __syncthreads();
        SX_PROBE();
        uint32_t tot = 0;
        {
            if (t < 256) {
#pragma unroll
                for (int ww = 0; ww < kWideWaves; ++ww) {
                    const uint32_t x = wcount[ww * 256 + t];
                    wcount[ww * 256 + t] = tot;
                    tot += x;
                }
            }
            const uint32_t inc = wave_inclusive_scan<OpAdd>(tot);
            if (lane == kWave - 1) scan_lds[w] = inc;
            __syncthreads();
            uint32_t base = 0;
            for (int ww = 0; ww < w; ++ww) base += scan_lds[ww];
            const uint32_t ex = base + inc - tot; // the bucket's first staged slot
            if (t < 256) {
#pragma unroll
                for (int ww = 0; ww < kWideWaves; ++ww) wcount[ww * 256 + t] += ex;
                // staged slot i of bucket t lands at goff + i (L pass) / goff - i (S pass)
                goff[t] = dir > 0 ? base_d + pre - ex : base_d - 1u - pre + ex;
            }
        }
        pre += tot;
        __syncthreads();
        uint32_t produced = 0;
        for (int ww = 0; ww < kWideWaves; ++ww) produced += scan_lds[ww];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k)
            if (lpos[k] & 0x8000u) lpos[k] = (lpos[k] & 0xFFFF0000u) | 0x8000u | ((lpos[k] & 0x1FFFu) + wcount[w * 256 + (lpos[k] >> 16)]);
        __syncthreads(); // the counters are part of the staging image
        SX_PROBE();
        // Windows that ran dry go back to the text: the round's only random access.  All of a thread's refills are
        // issued before the first one is used (under a branch per entry each would wait for its own trip to memory:
        // a seventh of the entries of a byte alphabet).
        {
            bool need[ITEMS];
#pragma unroll
            // (an entry that goes into bucket c itself is scanned next by the bucket's tail kernel, one workgroup: its window is
            //  filled up here, where the random reads are spread over the chip, while it still holds kTailAhead symbols)
            for (int k = 0; k < ITEMS; ++k)
                need[k] = (lpos[k] & 0x8000u) && val[k] != 0 &&
                          wnd_count<WT>(wnd[k]) <= (((lpos[k] >> 16) & 0xFFu) == c && refill_at < kTailAhead ? kTailAhead : refill_at);
            refill_windows<WT, ITEMS>(T, val, need, cfg, wnd);
        }
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            if (lpos[k] & 0x8000u) {
                const uint32_t slot = lpos[k] & 0x1FFFu;
                swnd[slot] = (uint64_t)wnd[k];
                sdig[slot] = (uint8_t)(lpos[k] >> 16);
            }
        }
        __syncthreads();
        SX_PROBE();
        uint32_t dstv[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const uint32_t i = (uint32_t)t + (uint32_t)k * kWideThreads;
            dstv[k] = 0;
            if (i < produced) {
                const WT nw = (WT)swnd[i];
                const uint32_t g = goff[sdig[i]];
                dstv[k] = dir > 0 ? g + i : g - i;
                WN[dstv[k]] = nw;
                BW[dstv[k]] = wnd_symbol<WT>(nw, cfg);
            }
        }
        __syncthreads();
        uint32_t *sval = reinterpret_cast<uint32_t *>(swnd);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k)
            if (lpos[k] & 0x8000u) sval[lpos[k] & 0x1FFFu] = val[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const uint32_t i = (uint32_t)t + (uint32_t)k * kWideThreads;
            if (i < produced) SA[dstv[k]] = sval[i];
        }
        __syncthreads(); // LDS is reused by the next step
        SX_PROBE();
#ifdef SX_WIDE_PROBE
        if (t == 0 && tile == 0 && blockIdx.x == 0 && (c == 60 || c == 200) && len > 100000 && step == 0)
            printf("probe c=%u len=%u mode=%d: load %lld rank %lld scan %lld refill+stage %lld store %lld cycles\n", c, len, mode,
                   pr[1] - pr[0], pr[2] - pr[1], pr[3] - pr[2], pr[4] - pr[3], pr[5] - pr[4]);
#endif
    }
}

template <class WT, int ITEMS>
__global__ __launch_bounds__(kWideThreads, 4) void induce_wide_scatter_kernel(
    const uint32_t *__restrict__ srcP, const WT *__restrict__ srcW, const uint32_t *__restrict__ range_in, int rev, int mode,
    uint32_t c, wnd_cfg cfg, const uint8_t *__restrict__ T, const uint32_t *__restrict__ offs /* [tile][256] */,
    const uint32_t *__restrict__ cursor_cur, int dir, uint32_t *__restrict__ SA, WT *__restrict__ WN,
    uint8_t *__restrict__ BW, uint32_t min_len)
{
    __shared__ wide_scatter_lds<ITEMS> lds;
    const int t = (int)threadIdx.x;
    const uint32_t lo = range_in[0], len = range_in[1] - lo;
    if (len <= min_len) return;
    const uint32_t ntiles = (len + kWideTile - 1) / kWideTile;
    const uint32_t base_d = t < 256 ? cursor_cur[t] : 0u;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // uniform per workgroup
        const uint32_t pre = t < 256 ? offs[(uint64_t)tile * 256 + t] : 0u; // entries of earlier tiles (and steps) for bucket t
        wide_scatter_tile<WT, ITEMS>(lds, srcP, srcW, lo, len, tile, rev, mode, c, cfg, T, pre, base_d, dir, SA, WN, BW);
    }
}

// ---- more than 8 buckets: every bucket's "other region" round at once, up front ------------------------------------
// Bucket c's pass is two scans: the entries the pass itself puts into c (L from L, S from S: rounds whose input is made
// as the pass goes) and a region that is complete before the pass begins -- c's LMS seeds in the L pass, c's L-type
// entries in the S pass.  Rounds 1 - 3 scanned that second region bucket by bucket: a count, an offsets and a scatter
// launch each, 2 x 255 times for a byte text, every one bound by its own latency (1 GiB of bytes: 30 of the 94 ms
// of the two passes), because its outputs land behind whatever the bucket's own rounds have appended so far.  But where
// they land is a property of the text.  Bucket d's L region is, in suffix-array order, for c = 0 .. d - 1 the entries
// p (text[p] = d, text[p + 1] = c) whose successor p + 1 is an L-type entry of c, then those whose successor is one
// of c's LMS suffixes, and last the entries with text[p + 1] = d; the first two groups together are the occurrences
// of the bigram (d, c) in the text, BG[d][c].  So with the bigram counts (one pass over the text, bigram_kernel) and
// the number of c's seeds that go to d (the counting launch's column totals) the place of every group is known
// before the pass starts: all buckets' seeds are split and written by ONE count / offsets / scatter (full bandwidth
// instead of 255 latencies), a bucket's pass is its own rounds alone, and bucket_begin_kernel sets the cursors to
// the group starts (and checks that the pass left them where the bigram counts say).  The S pass mirrors it: bucket
// d's S region from its end downwards is, for c = 255 .. d + 1, the entries whose successor is an S-type entry of
// c, then those whose successor is an L-type entry of c -- all L-type entries are final after the L pass.
#ifndef SX_HOIST_GRID_X
#define SX_HOIST_GRID_X 64u // workgroups a bucket in the up-front launches (they loop over the bucket's tiles); the CPU test harness: 2
#endif
#ifndef SX_BIGRAM_GRID
#define SX_BIGRAM_GRID 256u // (the CPU test harness: 2)
#endif
constexpr uint32_t kHoistGridX = SX_HOIST_GRID_X, kBigramGrid = SX_BIGRAM_GRID;
constexpr int kHoistOffGroups = SX_HOIST_GRID_X >= 4u ? 4 : 2; // groups of tiles a bucket's offsets workgroup walks (256 threads each)
constexpr uint32_t kBigramWords = 32768; // LDS counters of bigram_kernel: rows of nk counters, as many rows a pass as fit
constexpr int kBigramThreads = 1024;
__global__ __launch_bounds__(kBigramThreads) void bigram_kernel(const uint8_t *__restrict__ T, uint64_t n, uint32_t nk,
                                                                uint32_t *__restrict__ BG /* [256][256], zeroed */)
{
    __shared__ uint32_t cnt[kBigramWords];
    const uint32_t R = kBigramWords / nk; // rows of the matrix a pass holds (nk <= 256: at least 128)
    const uint64_t pieces = (n + 15) / 16, per = (pieces + gridDim.x - 1) / gridDim.x;
    const uint64_t q0 = (uint64_t)blockIdx.x * per, q1 = q0 + per < pieces ? q0 + per : pieces;
    for (uint32_t r0 = 0; r0 < nk; r0 += R) { // uniform
        const uint32_t rows = nk - r0 < R ? nk - r0 : R;
        for (uint32_t i = threadIdx.x; i < rows * nk; i += kBigramThreads) cnt[i] = 0;
        __syncthreads();
        for (uint64_t q = q0 + threadIdx.x; q < q1; q += kBigramThreads) {
            const uint64_t p0 = q * 16u;
            uint64_t w0, w1;
            load_bytes16(T, p0, w0, w1); // (the build's copy of the text is padded beyond text[n] = 0)
            uint32_t d = (uint32_t)(w0 & 0xFFu);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t nxt = e < 7 ? (uint32_t)(w0 >> (8 * (e + 1))) & 0xFFu
                                           : (e < 15 ? (uint32_t)(w1 >> (8 * (e - 7))) & 0xFFu : (uint32_t)T[p0 + 16u]);
                // the diagonal is never asked for (a symbol's run stays inside its bucket's own rounds), and it is
                // where the lanes of a wave would queue on one counter
                if (p0 + (uint32_t)e < n && d != nxt && d - r0 < rows) atomicAdd(&cnt[(d - r0) * nk + nxt], 1u);
                d = nxt;
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < rows * nk; i += kBigramThreads) {
            const uint32_t v = cnt[i];
            if (v) atomicAdd(&BG[(uint64_t)(r0 + i / nk) * 256u + i % nk], v);
        }
        __syncthreads();
    }
}

// EL[c][d] = begin[d] + sum of BG[d][c'] over c' <= c, c' < d: where bucket d's groups (d, 0 .. c) end;
// ES[c][d] = begin[d + 1] - sum of BG[d][c'] over c' >= c, c' > d: where its groups (d, 255 .. c) end, counted from the bucket's end
__global__ __launch_bounds__(256) void hoist_tables_kernel(const uint32_t *__restrict__ BG, const uint32_t *__restrict__ begin /* 257 */,
                                                         uint32_t nk, uint32_t *__restrict__ EL, uint32_t *__restrict__ ES)
{
    const uint32_t d = threadIdx.x;
    if (d >= nk) return;
    uint32_t acc = begin[d];
    for (uint32_t c = 0; c < nk; ++c) {
        if (c < d) acc += BG[(uint64_t)d * 256u + c];
        EL[(uint64_t)c * 256u + d] = acc;
    }
    acc = begin[d + 1];
    for (uint32_t c = nk; c-- > 0;) {
        if (c > d) acc -= BG[(uint64_t)d * 256u + c];
        ES[(uint64_t)c * 256u + d] = acc;
    }
}

// the region of every bucket c = blockIdx.y -- entries [lo[c], lo[c] + len[c]) of the source arrays -- counted tile by
// tile (hist rows row0[c] ...); desc: lo[256], len[256], row0[256]
template <class WT>
__global__ __launch_bounds__(kWideThreads) void hoist_count_kernel(const WT *__restrict__ srcW, const uint8_t *__restrict__ srcB,
                                                                const uint32_t *__restrict__ desc, int rev, int mode, wnd_cfg cfg,
                                                                uint32_t *__restrict__ hist /* [row][256] */)
{
    __shared__ uint32_t h[256];
    const uint32_t c = blockIdx.y, lo = desc[c], len = desc[256 + c], row0 = desc[512 + c];
    if (len == 0) return;
    const uint32_t ntiles = (len + kWideTile - 1) / kWideTile;
    const bool aligned = srcB ? ((uintptr_t)srcB & 15u) == 0 : ((uintptr_t)srcW & 15u) == 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // uniform per workgroup
        if (threadIdx.x < 256) h[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t tile0 = tile * (uint32_t)kWideTile;
        const uint32_t cnt = len - tile0 < (uint32_t)kWideTile ? len - tile0 : (uint32_t)kWideTile;
        const uint32_t a = rev ? lo + len - tile0 - cnt : lo + tile0, b = a + cnt;
        wide_count_range<WT>(srcW, srcB, a, b, mode, c, cfg, h, aligned);
        __syncthreads();
        if (threadIdx.x < 256) hist[(uint64_t)(row0 + tile) * 256 + threadIdx.x] = h[threadIdx.x];
        __syncthreads();
    }
}

// workgroup c: the tile counts of bucket c's region -> entries of earlier tiles, per destination bucket (in place); the
// column totals tot[c][d] = entries of c's region that go to bucket d; dbase[c][d] = where the first of them lands
// (L pass: the group (d, c) ends at EL[c][d] and these are its last tot entries; S pass: the group ends, downwards, at
// ES[c][d] and these are the last ones before that end -- as the cursor the scatter counts down from)
__global__ __launch_bounds__(kBlock * kHoistOffGroups) void hoist_offsets_kernel(uint32_t *__restrict__ hist, const uint32_t *__restrict__ desc,
                                                                              const uint32_t *__restrict__ E, int dir,
                                                                              uint32_t *__restrict__ tot, uint32_t *__restrict__ dbase)
{
    __shared__ uint32_t gsum[kHoistOffGroups][256];
    const uint32_t c = blockIdx.x, len = desc[256 + c], row0 = desc[512 + c];
    const uint32_t d = threadIdx.x & 255u, g = threadIdx.x >> 8;
    const uint32_t ntiles = (len + kWideTile - 1) / kWideTile;
    uint32_t *rows = hist + (uint64_t)row0 * 256;
    const uint32_t per = (ntiles + kHoistOffGroups - 1) / kHoistOffGroups;
    const uint32_t t0 = g * per < ntiles ? g * per : ntiles, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    constexpr int kBatch = 16;
    uint32_t sum = 0;
    for (uint32_t tb = t0; tb < t1; tb += kBatch) {
        uint32_t x[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) x[i] = tb + i < t1 ? rows[(uint64_t)(tb + i) * 256 + d] : 0u;
#pragma unroll
        for (int i = 0; i < kBatch; ++i) sum += x[i];
    }
    gsum[g][d] = sum;
    __syncthreads();
    uint32_t run = 0, all = 0;
#pragma unroll
    for (int gg = 0; gg < kHoistOffGroups; ++gg) {
        const uint32_t x = gsum[gg][d];
        if ((uint32_t)gg < g) run += x;
        all += x;
    }
    for (uint32_t tb = t0; tb < t1; tb += kBatch) {
        uint32_t x[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) x[i] = tb + i < t1 ? rows[(uint64_t)(tb + i) * 256 + d] : 0u;
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            if (tb + i < t1) rows[(uint64_t)(tb + i) * 256 + d] = run;
            run += x[i];
        }
    }
    if (g == 0) {
        tot[(uint64_t)c * 256 + d] = all;
        const uint32_t e = E[(uint64_t)c * 256 + d];
        dbase[(uint64_t)c * 256 + d] = dir > 0 ? e - all : e + all;
    }
}

template <class WT, int ITEMS>
__global__ __launch_bounds__(kWideThreads, 4) void hoist_scatter_kernel(
    const uint32_t *__restrict__ srcP, const WT *__restrict__ srcW, const uint32_t *__restrict__ desc, int rev, int mode, wnd_cfg cfg,
    const uint8_t *__restrict__ T, const uint32_t *__restrict__ offs /* [row][256] */, const uint32_t *__restrict__ dbase, int dir,
    uint32_t *__restrict__ SA, WT *__restrict__ WN, uint8_t *__restrict__ BW, uint32_t refill_at)
{
    __shared__ wide_scatter_lds<ITEMS> lds;
    const int t = (int)threadIdx.x;
    const uint32_t c = blockIdx.y, lo = desc[c], len = desc[256 + c], row0 = desc[512 + c];
    if (len == 0) return;
    const uint32_t ntiles = (len + kWideTile - 1) / kWideTile;
    const uint32_t base_d = t < 256 ? dbase[(uint64_t)c * 256 + t] : 0u;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // uniform per workgroup
        const uint32_t pre = t < 256 ? offs[(uint64_t)(row0 + tile) * 256 + t] : 0u;
        // The sort's seed windows hold two or three symbols: the entry a seed induces would be left with one, and the
        // round that scans it -- one of the bucket's own, a chain of launches each bound by its latency -- would go back
        // to the text for every such entry.  Here, at full occupancy, the read costs bandwidth only: windows that would be
        // left with a single symbol are read again at once (refill_at = 1 in the L pass).
        wide_scatter_tile<WT, ITEMS>(lds, srcP, srcW, lo, len, tile, rev, mode, c, cfg, T, pre, base_d, dir, SA, WN, BW, refill_at);
    }
}

// Start of bucket c's own rounds in a pass whose other-region rounds were done up front: every bucket the rounds can
// write to gets its cursor set to the start of its group (d, c) -- where the cursor must already be, give or take the
// up-front entries of the buckets since the last one that had rounds of its own (c_from .. c - 1 in the L pass,
// c + 1 .. c_from in the S pass): anything else means the pass and the bigram counts disagree (err) --, the first range
// is what lies in front of bucket c's own group.  A pass stopped by an unfinished bucket (poison) is left as it is.
__global__ __launch_bounds__(256) void bucket_begin_kernel(uint32_t *__restrict__ range, uint32_t *__restrict__ cursor,
                                                         const uint32_t *__restrict__ begin, const uint32_t *__restrict__ E,
                                                         const uint32_t *__restrict__ tot, uint32_t nk, uint32_t c, uint32_t c_from,
                                                         int dir, uint32_t *__restrict__ tickets, uint32_t ntickets,
                                                         const uint32_t *__restrict__ poison, uint32_t *__restrict__ err)
{
    const uint32_t d = threadIdx.x;
    if (d == 0)
        for (uint32_t i = 0; i < ntickets; ++i) tickets[i] = 0;
    if (poison && poison[0]) {
        if (d == 0) range[0] = range[1] = 0;
        return;
    }
    if (d >= nk) return;
    if (dir > 0 && d >= c) {
        const uint32_t want = c == 0 ? begin[d] : E[(uint64_t)(c - 1) * 256 + d];
        uint32_t have = cursor[d];
        for (uint32_t k = c_from; k < c; ++k) have += tot[(uint64_t)k * 256 + d];
        if (have != want) atomicOr(err, 1u);
        cursor[d] = want;
        if (d == c) range[0] = begin[c], range[1] = want;
    } else if (dir < 0 && d <= c) {
        const uint32_t want = c + 1 >= nk ? begin[d + 1] : E[(uint64_t)(c + 1) * 256 + d];
        uint32_t have = cursor[d];
        for (uint32_t k = c_from; k > c; --k) have -= tot[(uint64_t)k * 256 + d];
        if (have != want) atomicOr(err, 2u);
        cursor[d] = want;
        if (d == c) range[0] = want, range[1] = begin[c + 1];
    }
}

} // namespace sx
