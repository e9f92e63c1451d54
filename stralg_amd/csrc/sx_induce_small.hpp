// sx_induce_small.hpp -- the induced-sort passes over at most 8 buckets (DNA): large rounds (count, offsets, scatter), the one-workgroup tail kernel, eight rounds at a time
// (included by sx_induce.hip, which holds the passes' host side; one translation unit)
#pragma once
#include "sx_induce_common.hpp"

namespace sx {

// ---- large rounds: count, offsets, scatter ----------------------------------------------
// A round whose range is longer than chain_max entries is split in three launches (the
// entries are read twice); shorter rounds take the single chained launch below, whose
// look-back walk costs a few microseconds per tile and would dominate a long round.
// Both forms are queued for every round; each checks the range and returns at once when
// the round belongs to the other.
// (threshold: sx_ctx::chain_max_entries, default 256 tiles; SX_FLAG_CHAIN_MAX_ENTRIES)

// The byte form for at most 8 buckets as a kernel of its own: as one branch of the template below it shared that kernel's
// 118 registers (the window form keeps 36 words of windows in flight) and ran four waves a SIMD, each alive for one
// 2 KiB tile: the launches reached 1 TB/s of their byte per entry, bound by nothing but the waves' own latencies.
__global__ __launch_bounds__(kBlock) void induce_count_bytes_kernel(const uint8_t *__restrict__ srcB,
                                                                    const uint32_t *__restrict__ range_in, int rev, int mode,
                                                                    uint32_t c, uint32_t *__restrict__ hist, uint32_t stride,
                                                                    uint32_t nkeys, uint32_t chain_max, uint64_t src_len)
{
    const uint32_t lo = range_in[0], len = range_in[1] - lo;
    if (len <= chain_max) return;
    const uint32_t ntiles = (len + kIndTile - 1) / kIndTile;
    const bool aligned = ((uintptr_t)srcB & 15u) == 0;
    // (Measured and dropped: a lane on 32 consecutive bytes of the tile as two unaligned 16-byte loads -- no straddling
    //  pieces, a third less vector work, but 1.30 against 1.00 ms a step: the launch is bound by its line requests, and a
    //  wave's load then spans 32 lines half used instead of 16 whole ones; grids of 512 ... 16 384 workgroups: no difference.)
    {
        // a wave per tile, all of the tile's pieces in flight at once, no LDS and no barrier
        constexpr int kPieces = kIndTile / 16 / kWave + 1; // the tile's range may start inside a piece
        const int lane = lane_id();
        for (uint32_t tile = blockIdx.x * kWavesPerBlock + wave_id(); tile < ntiles; tile += gridDim.x * kWavesPerBlock) {
            const uint32_t tile0 = tile * (uint32_t)kIndTile;
            const uint32_t cnt = len - tile0 < (uint32_t)kIndTile ? len - tile0 : (uint32_t)kIndTile;
            const uint32_t a = rev ? lo + len - tile0 - cnt : lo + tile0, b = a + cnt; // the tile's entries: [a, b)
            // Every lane loads whole aligned pieces, also the one or two that straddle the ends of [a, b) (they lie
            // inside the array): the bytes outside the range are masked after the bit planes are gathered.  (Reading
            // those pieces byte by byte under a branch made every wave wait for a chain of dependent loads: the launch
            // ran at 1 TB/s of its 1 byte per entry.)
            uint32_t S[kPieces][4], inside[kPieces];
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                const uint64_t e0 = ((uint64_t)(a >> 4) + (uint64_t)lane + (uint64_t)k * kWave) * 16u;
                S[k][0] = S[k][1] = S[k][2] = S[k][3] = 0; // (symbol 0 counts nowhere)
                const uint32_t from = e0 < a ? (uint32_t)(a - e0) : 0u, to = e0 >= b ? 0u : (b - e0 < 16u ? (uint32_t)(b - e0) : 16u);
                inside[k] = from < to ? ((1u << to) - 1u) & ~((1u << from) - 1u) : 0u; // the piece's entries in [a, b)
                if (aligned && e0 + 16u <= src_len) {
                    if (inside[k]) load_quad(reinterpret_cast<const uint32_t *>(srcB + e0), S[k]);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if ((inside[k] >> e) & 1u) S[k][e >> 2] |= (uint32_t)srcB[e0 + e] << (8 * (e & 3));
                }
            }
            uint32_t n_of[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // this lane's entries per symbol (at most 16 * kPieces)
#pragma unroll
            for (int k = 0; k < kPieces; ++k) {
                const uint32_t one = 0x01010101u;
                const uint32_t b0 = gather16(S[k][0] & one, S[k][1] & one, S[k][2] & one, S[k][3] & one, 0);
                const uint32_t b1 = gather16(S[k][0] & (one << 1), S[k][1] & (one << 1), S[k][2] & (one << 1), S[k][3] & (one << 1), 1);
                const uint32_t b2 = gather16(S[k][0] & (one << 2), S[k][1] & (one << 2), S[k][2] & (one << 2), S[k][3] & (one << 2), 2);
// (only the symbols the text holds: DNA counts four of the seven, and the launch is bound by vector instructions)
#define SX_IND_COUNT(A) if ((A) < nkeys) n_of[A] += (uint32_t)__popc(__builtin_amdgcn_bitop3_b32(b0, b1, b2, 1u << ((((A) & 1) << 2) | ((A) & 2) | (((A) >> 2) & 1))) & inside[k]);
                SX_IND_COUNT(1) SX_IND_COUNT(2) SX_IND_COUNT(3) SX_IND_COUNT(4) SX_IND_COUNT(5) SX_IND_COUNT(6) SX_IND_COUNT(7)
#undef SX_IND_COUNT
            }
            uint64_t even = (uint64_t)n_of[2] << 16 | (uint64_t)n_of[4] << 32 | (uint64_t)n_of[6] << 48; // 16-bit fields
            uint64_t odd = (uint64_t)n_of[1] | (uint64_t)n_of[3] << 16 | (uint64_t)n_of[5] << 32 | (uint64_t)n_of[7] << 48;
            even = wave_total_packed(even);
            odd = wave_total_packed(odd);
            if ((uint32_t)lane < nkeys && lane < 8) {
                const uint32_t v = (uint32_t)(((lane & 1) ? odd : even) >> (16 * (lane >> 1))) & 0xFFFFu;
                hist[(uint64_t)lane * stride + tile] = lane != 0 && induce_accept((uint32_t)lane, c, mode) ? v : 0u;
            }
        }
    }
}

// Counting reads one byte per entry, not the entry: every writer of (SA, WN) leaves the entry's symbol
// text[SA[i] - 1] in a byte array next to them (0 for the entry of position 0, which induces nothing) --
// the array that is the BWT in the end.  The LMS seeds have no such bytes (srcB == NULL); there the windows are
// read: a stored window is empty exactly when its entry is position 0, every other window is refilled from the
// text the moment it runs dry.  The counts of a tile do not depend on the order of its entries, so the tile's
// index range is read as aligned 16-byte pieces; the one or two pieces that straddle the range ends are read
// entry by entry.  BITS = 3 (at most 8 buckets): a wave per tile, symbol masks and popcounts per lane
// (sx_device.hpp: gather16) reduced over the wave, instead of 64 lanes queueing on a handful of LDS words.
template <class WT, int BITS>
__global__ __launch_bounds__(kBlock) void induce_count_kernel(const WT *__restrict__ srcW,
                                                              const uint8_t *__restrict__ srcB,
                                                              const uint32_t *__restrict__ range_in, int rev,
                                                              int mode, uint32_t c, wnd_cfg cfg,
                                                              uint32_t *__restrict__ hist, uint32_t stride,
                                                              uint32_t nkeys, uint32_t chain_max,
                                                              uint64_t src_len /* entries of the source arrays */)
{
    static_assert(BITS == 3, "the window form of at most 8 buckets (more buckets: induce_wide_count_kernel)");
    const uint32_t lo = range_in[0], len = range_in[1] - lo;
    if (len <= chain_max) return;
    const uint32_t ntiles = (len + kIndTile - 1) / kIndTile;
    const bool aligned = ((uintptr_t)srcW & 15u) == 0;
    (void)srcB;
    {
        // a wave per tile, all of the tile's quads in flight at once, no LDS and no barrier
        constexpr int kQuads = kIndTile / 4 / kWave + 1; // the tile's range may start inside a quad
        const int lane = lane_id();
        for (uint32_t tile = blockIdx.x * kWavesPerBlock + wave_id(); tile < ntiles; tile += gridDim.x * kWavesPerBlock) {
            const uint32_t tile0 = tile * (uint32_t)kIndTile;
            const uint32_t cnt = len - tile0 < (uint32_t)kIndTile ? len - tile0 : (uint32_t)kIndTile;
            const uint32_t a = rev ? lo + len - tile0 - cnt : lo + tile0, b = a + cnt; // the tile's entries: [a, b)
            // (one test for the wave: every quad it may load lies inside the array -- all tiles but the array's last)
            const bool whole = aligned && ((uint64_t)(a >> 2) + (uint64_t)kQuads * kWave) * 4u <= src_len;
            uint64_t packed = 0; // one 8-bit counter per bucket (a lane sees at most 4 * kQuads entries)
            // The tile in two halves of kHalf quads a lane: all nine at once kept 36 windows and as many addresses alive,
            // 118 registers, four waves a SIMD -- and the launch is bound by its waves' latencies, not by their work.
            constexpr int kHalf = (kQuads + 1) / 2;
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                WT W[kHalf][4];
                uint32_t inside[kHalf]; // (whole aligned quads are loaded, the entries outside [a, b) masked: see above)
#pragma unroll
                for (int kk = 0; kk < kHalf; ++kk) {
                    const int k = h * kHalf + kk;
                    const uint64_t e0 = ((uint64_t)(a >> 2) + (uint64_t)lane + (uint64_t)k * kWave) * 4u;
                    const uint32_t from = e0 < a ? (uint32_t)(a - e0) : 0u, to = e0 >= b ? 0u : (b - e0 < 4u ? (uint32_t)(b - e0) : 4u);
                    inside[kk] = (k < kQuads && from < to) ? ((1u << to) - 1u) & ~((1u << from) - 1u) : 0u;
                    W[kk][0] = W[kk][1] = W[kk][2] = W[kk][3] = 0;
                    if (whole) {
                        if (k < kQuads) load_quad(srcW + e0, W[kk]);
                    } else { // (the array's last tile)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if ((inside[kk] >> e) & 1u) W[kk][e] = srcW[e0 + e];
                    }
                }
#pragma unroll
                for (int kk = 0; kk < kHalf; ++kk) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t ch = wnd_first<WT>(W[kk][e], cfg) & 7u;
                        const bool ok = ((inside[kk] >> e) & 1u) && wnd_count<WT>(W[kk][e]) != 0 && induce_accept(ch, c, mode);
                        packed += (uint64_t)(ok ? 1u : 0u) << (8u * ch);
                    }
                }
            }
            uint64_t even = packed & 0x00FF00FF00FF00FFull, odd = (packed >> 8) & 0x00FF00FF00FF00FFull; // 16-bit fields
            even = wave_total_packed(even);
            odd = wave_total_packed(odd);
            if ((uint32_t)lane < nkeys && lane < 8)
                hist[(uint64_t)lane * stride + tile] = (uint32_t)(((lane & 1) ? odd : even) >> (16 * (lane >> 1))) & 0xFFFFu;
        }
        return;
    }
}

// one workgroup (1024 threads) per destination bucket: exclusive prefix over the tiles, cursor update
__global__ __launch_bounds__(kRowThreads) void induce_offsets_kernel(uint32_t *__restrict__ hist, uint32_t stride,
                                                                const uint32_t *__restrict__ range_in,
                                                                uint32_t *__restrict__ range_out,
                                                                const uint32_t *__restrict__ cursor_cur,
                                                                uint32_t *__restrict__ cursor_nxt, int dir, uint32_t c,
                                                                uint32_t chain_max,
                                                                int only_form /* no chained launch follows (chain_max = 0): an empty range is carried on here */)
{
    __shared__ uint32_t lds[kRowPieces * kRowWaves];
    const uint32_t len = range_in[1] - range_in[0];
    if (len <= chain_max) {
        if (only_form && threadIdx.x == 0) {
            cursor_nxt[blockIdx.x] = cursor_cur[blockIdx.x];
            if (blockIdx.x == c && range_out) range_out[0] = range_out[1] = range_in[1];
        }
        return;
    }
    const uint32_t ntiles = (len + kIndTile - 1) / kIndTile;
    const uint32_t key = blockIdx.x;
    const uint32_t total = wide_scan_row_inplace(hist + (uint64_t)key * stride, ntiles, lds);
    if (threadIdx.x == 0) {
        const uint32_t cur = cursor_cur[key];
        cursor_nxt[key] = dir > 0 ? cur + total : cur - total;
        if (key == c && range_out) {
            range_out[0] = dir > 0 ? cur : cur - total;
            range_out[1] = dir > 0 ? cur + total : cur;
        }
    }
}

// The scatter for at most 8 buckets (DNA, and every alphabet of up to 7 symbols): the ranking of
// the general kernel above costs ~100 vector instructions per entry (a match over the wave per
// item), which is what bounds it, not memory.  Here every thread owns 8 consecutive entries of
// the scan order and counts its own buckets in 8-bit fields of one register pair; the fields,
// widened to 16 bits, are prefix-summed over the workgroup two 64-bit words at a time, and an
// entry's slot in the tile's output is (entries of its bucket in earlier threads) + (its rank
// inside the thread).  The output is staged in LDS in bucket order so that each bucket's run
// leaves as one contiguous block.  MODE fixes the scan direction and the accept test at compile time.
template <class WT, int MODE>
__global__ __launch_bounds__(kBlock) void induce_scatter_small_kernel(
    const uint32_t *__restrict__ srcP, const WT *__restrict__ srcW, const uint32_t *__restrict__ range_in, uint32_t c,
    wnd_cfg cfg, const uint8_t *__restrict__ T, const uint32_t *__restrict__ offs, uint32_t stride,
    const uint32_t *__restrict__ cursor_cur, uint32_t *__restrict__ SA, WT *__restrict__ WN, uint8_t *__restrict__ BW,
    uint32_t nkeys, uint32_t chain_max)
{
    constexpr bool kRev = MODE == MODE_S_FROM_S || MODE == MODE_S_FROM_L; // the S pass scans right to left
    constexpr uint64_t kField16 = 0x00FF00FF00FF00FFull;
    __shared__ uint64_t wsum[2][kWavesPerBlock];
    __shared__ uint64_t sbase[2];  // first slot of every bucket in the staged output, 16-bit fields (even, odd buckets)
    __shared__ uint32_t gadj[8];   // destination of staged slot i of bucket d: gadj[d] + i (L pass), gadj[d] - i (S pass)
    __shared__ uint32_t sP[kIndTile];
    __shared__ WT sW[kIndTile];
    __shared__ uint8_t sD[kIndTile];
    __shared__ uint16_t refill[kIndTile]; // staged slots whose window ran dry
    __shared__ uint32_t nrefill;
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t lo = range_in[0], len = range_in[1] - lo;
    if (len <= chain_max) return;
    const uint32_t ntiles = (len + kIndTile - 1) / kIndTile;
    const uint32_t base_d = t < (int)nkeys ? cursor_cur[t] : 0u;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // uniform per workgroup
        // this tile's first destination per bucket: asked for now, needed after the scan
        const uint32_t pre = t < (int)nkeys ? offs[(uint64_t)t * stride + tile] : 0u;
        if (t == 0) nrefill = 0;
        const uint32_t i0 = tile * (uint32_t)kIndTile + (uint32_t)t * kIndItems; // the thread's first entry, scan order
        uint32_t P[kIndItems];
        WT W[kIndItems];
        if (i0 + kIndItems <= len) { // the 8 entries are contiguous in memory: two 16-byte loads per array (4-byte aligned)
            const uint32_t first = kRev ? lo + len - i0 - kIndItems : lo + i0;
            uint32_t Pm[kIndItems];
            WT Wm[kIndItems];
            __builtin_memcpy(Pm, srcP + first, sizeof(Pm));
            __builtin_memcpy(Wm, srcW + first, sizeof(Wm));
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) {
                P[k] = Pm[kRev ? kIndItems - 1 - k : k];
                W[k] = Wm[kRev ? kIndItems - 1 - k : k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) {
                const uint32_t i = i0 + (uint32_t)k;
                const uint32_t idx = i < len ? (kRev ? lo + len - 1u - i : lo + i) : lo;
                P[k] = i < len ? srcP[idx] : 0u;
                W[k] = i < len ? srcW[idx] : (WT)0;
            }
        }
        uint32_t rnk[kIndItems], dig[kIndItems];
        bool ok[kIndItems];
        uint64_t cnt = 0; // 8-bit count per bucket of this thread's accepted entries
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            const uint32_t ch = wnd_first<WT>(W[k], cfg) & 7u;
            ok[k] = P[k] != 0 && induce_accept(ch, c, MODE);
            dig[k] = ch;
            rnk[k] = (uint32_t)(cnt >> (8u * ch)) & 0xFFu;
            cnt += (uint64_t)(ok[k] ? 1u : 0u) << (8u * ch);
        }
        // exclusive prefix over the threads, both words at once
        const uint64_t own0 = cnt & kField16, own1 = (cnt >> 8) & kField16;
        const uint64_t inc0 = wave_inclusive_sum_packed(own0), inc1 = wave_inclusive_sum_packed(own1);
        if (lane == kWave - 1) wsum[0][w] = inc0, wsum[1][w] = inc1;
        __syncthreads();
        uint64_t ex0 = inc0 - own0, ex1 = inc1 - own1, tot0 = 0, tot1 = 0;
#pragma unroll
        for (int i = 0; i < kWavesPerBlock; ++i) {
            const uint64_t x0 = wsum[0][i], x1 = wsum[1][i];
            if (i < w) ex0 += x0, ex1 += x1;
            tot0 += x0, tot1 += x1;
        }
        if (t < 8) { // bucket t: its first staged slot and where that slot lands in SA
            uint32_t first_slot = 0;
            for (int d = 0; d < t; ++d) first_slot += (uint32_t)(((d & 1) ? tot1 : tot0) >> (16 * (d >> 1))) & 0xFFFFu;
            const uint32_t g = kRev ? base_d - 1u - pre : base_d + pre;
            gadj[t] = kRev ? g + first_slot : g - first_slot;
        }
        if (t == 0) { // the same first slots as two words of 16-bit fields (even buckets, odd buckets)
            uint64_t even = 0, odd = 0;
            uint32_t run = 0;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                if (d & 1) odd |= (uint64_t)run << (16 * (d >> 1));
                else even |= (uint64_t)run << (16 * (d >> 1));
                run += (uint32_t)(((d & 1) ? tot1 : tot0) >> (16 * (d >> 1))) & 0xFFFFu;
            }
            sbase[0] = even, sbase[1] = odd;
        }
        __syncthreads();
        ex0 += sbase[0];
        ex1 += sbase[1];
        const uint32_t produced = (uint32_t)((tot0 & 0xFFFFu) + ((tot0 >> 16) & 0xFFFFu) + ((tot0 >> 32) & 0xFFFFu) + (tot0 >> 48) +
                                             (tot1 & 0xFFFFu) + ((tot1 >> 16) & 0xFFFFu) + ((tot1 >> 32) & 0xFFFFu) + (tot1 >> 48));
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            if (ok[k]) {
                const uint32_t d = dig[k];
                const uint32_t slot = ((uint32_t)(((d & 1u) ? ex1 : ex0) >> (16u * (d >> 1))) & 0xFFFFu) + rnk[k];
                const uint32_t j = P[k] - 1u;
                const WT nw = wnd_pop<WT>(W[k], cfg);
                sP[slot] = j;
                sW[slot] = nw;
                sD[slot] = (uint8_t)d;
                if (j != 0 && wnd_count<WT>(nw) == 0) refill[atomicAdd(&nrefill, 1u)] = (uint16_t)slot;
            }
        }
        __syncthreads();
        // windows that ran dry go back to the text: the round's only random access, taken by as many
        // threads at once as there are such entries
        const uint32_t nre = nrefill;
        if (nre) { // uniform
            for (uint32_t r = (uint32_t)t; r < nre; r += kBlock) {
                const uint32_t slot = refill[r];
                sW[slot] = wnd_fill<WT>(T, sP[slot], cfg);
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            const uint32_t i = (uint32_t)t + (uint32_t)k * kBlock;
            if (i < produced) {
                const uint32_t g = gadj[sD[i]];
                const uint32_t dst = kRev ? g - i : g + i;
                const WT nw = sW[i];
                SA[dst] = sP[i];
                WN[dst] = nw;
                BW[dst] = wnd_symbol<WT>(nw, cfg);
            }
        }
        __syncthreads(); // LDS is reused by the next tile
    }
}

// ---- the same for at most 8 buckets, without a ballot ---------------------------------------------------------
// The ranking of the kernel above costs ~30 vector instructions per entry and round (a match over the wave), and a
// workgroup of 16 waves issues them one wave at a time: 4 us a round.  With at most 8 buckets a thread can count on
// its own: it holds 8 *consecutive* entries of the scan order, counts their buckets in the 8-bit fields of one
// register pair (the rank inside the thread is the field's value at that moment), and the fields, widened to 16 bits,
// are prefix-summed over the workgroup as two 64-bit words (as induce_scatter_small does).  And because an entry's
// window already says where its descendants of the next rounds go -- the j-th symbol to its left is the bucket of the
// j-th one, and they exist as long as the symbols before were c -- up to kTailBatch rounds are taken in one step
// (when the range shrinks slowly: poly-A tracts and microsatellites of differing lengths, where the all-in-a-run jump
// never applies): one set of counters per round, one prefix over threads and rounds, one scatter.
constexpr int kTailBatch = 8;
constexpr uint32_t kRunCap = 255; // symbols of a run the closed form looks at (a byte a length)
#ifndef SX_TAIL_CLOSED_FROM
#define SX_TAIL_CLOSED_FROM 256u // (the CPU test harness: 4, so that short texts' runs take this form too)
#endif
constexpr uint32_t kClosedFrom = SX_TAIL_CLOSED_FROM; // entries of a range from which the closed form is taken
constexpr uint32_t kClosedEntries = 5120;             // ... and up to which: what fits the workgroup's LDS
template <class WT>
__global__ __launch_bounds__(kTailBlock) void induce_tail_small_kernel(uint32_t *SA, WT *WN, uint8_t *BW,
                                                                       const uint32_t *__restrict__ range_in,
                                                                       uint32_t *__restrict__ range_out, int rev, int mode,
                                                                       uint32_t c, wnd_cfg cfg, const uint8_t *__restrict__ T,
                                                                       const uint32_t *__restrict__ cursor_cur,
                                                                       uint32_t *__restrict__ cursor_nxt, int dir,
                                                                       uint32_t max_iters, uint32_t *poison, uint32_t *host_poison)
{
    constexpr uint64_t kField16 = 0x00FF00FF00FF00FFull;
    __shared__ uint64_t wsum[kTailBatch][2][kTailWaves];  // per round and half (even / odd buckets): the waves' totals, then their prefix
    __shared__ uint32_t s_tot[kTailBatch][8], s_base[kTailBatch][8];
    __shared__ uint32_t gbase[8];
    __shared__ uint32_t s_range[2];
    __shared__ uint32_t s_flag;
    // the closed form of a range of runs (below): positions, run lengths, the symbol that ends each run; per run length
    // the entries (then: where the round's entries begin), per bucket and run length the ends (then: where they begin)
    // (the window of the position in front of each run too, so that no round goes back to the text: as many entries as that
    //  leaves room for in 64 KiB of LDS; 32-bit windows only, which is what at most 8 buckets have)
    constexpr uint32_t kCfEntries = sizeof(WT) == 4 ? kClosedEntries : 64u;
    __shared__ uint32_t cf_p[kCfEntries];
    __shared__ WT cf_w[kCfEntries];
    __shared__ uint8_t cf_r[kCfEntries];
    __shared__ uint32_t cf_hr[kRunCap + 1], cf_he[8][kRunCap + 1], cf_scan[kTailWaves], cf_rmax;
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    if (t < 8) gbase[t] = cursor_cur[t];
    if (t == 0) {
        s_range[0] = range_in[0];
        s_range[1] = range_in[1];
    }
    __syncthreads();
    const uint32_t B = cfg.B, cmask = cfg.mask;
    bool last_batch = false; // the last step took a batch of rounds (or the closed form)
    uint32_t val[kIndItems]; // entries t * per .. t * per + per - 1 of the range as it was loaded, in scan order
    WT wnd[kIndItems];
    uint32_t live = 0;       // bit k: entry k belongs to the current range
    uint32_t per = kIndItems; // entries a thread took when the range was loaded: as few as spread it over all the threads
    bool held = false;
    uint32_t prev_len = 0, last_in = ~0u; // the range of the round before (for the jump); of the last step taken (for the batch)
    for (uint32_t it = 0; it < max_iters;) {
        const uint32_t lo = s_range[0], len = s_range[1] - lo;
        if (len == 0 || len > kTailEntries) break; // uniform
        // ---- run jump (as in induce_tail_kernel) --------------------------------------------------------------------
        if (len == prev_len) {
            const uint32_t G = len <= (uint32_t)kTailBlock ? (uint32_t)kTailBlock / len : 1u; // threads per entry
            const uint64_t cpat = 0x0101010101010101ull * (uint64_t)c;
            // thread (i, q) looks at the q-th 16 symbols to the left of entry i; the nearest piece of any entry that is
            // not all c bounds the rounds that can be written at once (a run of 3000 symbols: 2992 rounds in one step,
            // where all G pieces had to be c before -- 16 384 symbols for a single run -- and shorter runs went round by round)
            uint32_t first_other = G;
            for (uint32_t e = (uint32_t)t; e < len * G; e += kTailBlock) {
                const uint32_t i = e / G, q = e % G;
                const uint32_t p = SA[lo + (rev ? len - 1u - i : i)];
                bool all_c = false;
                if (p >= 16u * (q + 1u)) {
                    uint64_t o0, o1;
                    load_bytes16(T, (uint64_t)(p - 16u * (q + 1u)), o0, o1);
                    all_c = o0 == cpat && o1 == cpat;
                }
                if (!all_c && q < first_other) first_other = q;
            }
            if (t == 0) s_flag = G;
            __syncthreads();
            if (first_other < G) atomicMin(&s_flag, first_other);
            __syncthreads();
            const uint32_t L = 16u * s_flag;
            __syncthreads(); // (s_flag is set again by the next step)
            if (L) { // uniform
                const uint32_t cur = gbase[c], total = L * len;
                for (uint32_t o = (uint32_t)t; o < total; o += kTailBlock) {
                    const uint32_t j = o / len + 1u, i = o % len;
                    const uint32_t v = SA[lo + (rev ? len - 1u - i : i)] - j;
                    const uint32_t dst = dir > 0 ? cur + o : cur - 1u - o;
                    const WT nw = v ? wnd_fill<WT>(T, v, cfg) : (WT)0;
                    SA[dst] = v;
                    WN[dst] = nw;
                    BW[dst] = wnd_symbol<WT>(nw, cfg);
                }
                __syncthreads();
                if ((uint32_t)t == c) {
                    gbase[c] = dir > 0 ? cur + total : cur - total;
                    s_range[0] = dir > 0 ? cur + total - len : cur - total;
                    s_range[1] = dir > 0 ? cur + total : cur - total + len;
                }
                held = false; // the range is now what the jump wrote last
                last_batch = false;
                ++it;
                __syncthreads();
                continue;
            }
        }
        prev_len = len;
        // ---- closed form of a range that is runs of differing lengths ------------------------------------------------
        // Thousands of poly-A tracts and microsatellites: the range loses a few entries a round for two hundred rounds,
        // the all-in-a-run jump never applies, and the batched step below costs this one CU 20 instructions for every
        // (entry, round) whether the entry is alive or not (a genome-like 1 GiB text: 0.35 - 0.75 ms a bucket and pass,
        // in proportion to the bucket's runs; 3.3 of the build's 40 ms).  But the rounds of such a range are a function
        // of the runs' lengths alone: entry i with r_i symbols c to its left is in rounds 1 .. r_i (position p_i - j in
        // round j, behind the round's entries i' < i with r_i' >= j), and the symbol d_i that ends its run sends
        // p_i - r_i - 1 to bucket d_i in round r_i + 1 (behind the ends of earlier rounds and of entries i' < i of the
        // same round), if the pass's type test accepts it.  So: the run lengths (up to kRunCap) by one look at the text,
        // their histogram -> where every round begins in bucket c and in the other buckets, and the rounds are written
        // one to a wave, independently.  Taken when the last batch of rounds kept half of its entries.
        if ((mode == MODE_L_FROM_L || mode == MODE_S_FROM_S) && c >= 1u && last_batch && (uint64_t)len * 2 >= last_in &&
            len >= kClosedFrom && len <= kCfEntries) { // uniform
            if (t <= (int)kRunCap) cf_hr[t] = 0;
            for (uint32_t i = (uint32_t)t; i < (kRunCap + 1) * 8; i += kTailBlock) (&cf_he[0][0])[i] = 0;
            __syncthreads();
            const uint64_t cpat = 0x0101010101010101ull * (uint64_t)c;
            const uint32_t per2 = (len + (uint32_t)kTailBlock - 1u) / (uint32_t)kTailBlock;
#pragma unroll 1
            for (uint32_t k = 0; k < per2; ++k) {
                const uint32_t i = (uint32_t)t * per2 + k;
                if (i < len) {
                    const uint32_t p = SA[lo + (rev ? len - 1u - i : i)];
                    uint32_t r = 0;
                    bool open = p != 0;
                    for (uint32_t g = 0; g < (kRunCap + 1) / 64 && open; ++g) { // 64 symbols a look, their loads in flight together
                        uint64_t o[4][2];
#pragma unroll
                        for (uint32_t u = 0; u < 4; ++u) {
                            const uint32_t done = 64u * g + 16u * u;
                            o[u][0] = o[u][1] = ~cpat;
                            if (p >= done + 16u) load_bytes16(T, (uint64_t)(p - done - 16u), o[u][0], o[u][1]);
                        }
#pragma unroll
                        for (uint32_t u = 0; u < 4; ++u) {
                            const uint32_t done = 64u * g + 16u * u;
                            if (!open) {
                            } else if (p >= done + 16u) {
                                const uint64_t x1 = o[u][1] ^ cpat, x0 = o[u][0] ^ cpat;
                                if (x1) r += (uint32_t)__clzll((unsigned long long)x1) >> 3, open = false;
                                else if (x0) r += 8u + ((uint32_t)__clzll((unsigned long long)x0) >> 3), open = false;
                                else r += 16u;
                            } else { // fewer than 16 symbols between the text's start and here
                                uint32_t a = p - done;
                                while (a > 0 && T[a - 1u] == (uint8_t)c) --a, ++r;
                                open = false;
                            }
                        }
                    }
                    if (r > kRunCap) r = kRunCap;
                    // the window of the position in front of the run (of the last symbol looked at, for a longer run): with
                    // it every round's window is known -- the run's symbols that are left, then these
                    const WT wend = p > r ? wnd_fill<WT>(T, p - r, cfg) : (WT)0;
                    cf_p[i] = p;
                    cf_r[i] = (uint8_t)r;
                    cf_w[i] = wend;
                    atomicAdd(&cf_hr[r], 1u);
                    if (r < kRunCap && p > r) { // the run ends inside the look, and not at the text's start
                        const uint32_t d = wnd_first<WT>(wend, cfg);
                        if (induce_accept(d, c, mode)) atomicAdd(&cf_he[d & 7u][r], 1u);
                    }
                }
            }
            __syncthreads();
            // wave 0: entries per run length -> the longest run; entries alive in round u + 1 (S[u]); cf_hr[u] <- sum of S below u
            // = where round u + 1 begins in bucket c.  waves 1 .. 8: ends per run length -> cf_he[d][u] <- ends of shorter runs
            // = where the ends of round u + 1 begin in bucket d.
            if (w <= 8) {
                uint32_t *row = w == 0 ? cf_hr : cf_he[w - 1];
                uint32_t v[4], sum = 0, top = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = row[4 * lane + q];
                    if (v[q]) top = (uint32_t)(4 * lane + q);
                    sum += v[q];
                }
                uint32_t inc = wave_inclusive_scan<OpAdd>(sum);
                if (w == 0) {
                    const uint32_t rmax = wave_reduce_max(top);
                    if (lane == 0) cf_rmax = rmax;
                    // alive after u + 1 rounds' worth of run: len - (entries with r <= u)
                    uint32_t run = inc - sum, s4[4], ssum = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        run += v[q];
                        s4[q] = len - run;
                        ssum += s4[q];
                    }
                    const uint32_t inc2 = wave_inclusive_scan<OpAdd>(ssum);
                    uint32_t run2 = inc2 - ssum;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        row[4 * lane + q] = run2;
                        run2 += s4[q];
                    }
                } else {
                    uint32_t run = inc - sum;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        row[4 * lane + q] = run;
                        run += v[q];
                    }
                }
            }
            __syncthreads();
            const uint32_t rmax = cf_rmax;
            const uint32_t J = rmax < kRunCap ? rmax + 1u : kRunCap; // rounds written here (the last one may hold ends only)
            const uint32_t curC = gbase[c];
            uint32_t curD[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) curD[d] = gbase[d];
            WT crun = 0; // CW codes of symbol c: the window of an entry at least CW symbols inside its run
            for (uint32_t i = 0; i < cfg.CW; ++i) crun = (crun << B) | (WT)(c - 1u);
            const WT field_mask = cfg.CW * B >= 8 * sizeof(WT) ? ~(WT)0 : (((WT)1 << (cfg.CW * B)) - 1u);
            // Sixteen rounds (one a wave) at a time; a round looks at every entry that is left.  Entries whose runs ended before
            // the sixteen are taken out first (in order: a round's places are ranks among the entries alive), once an
            // eighth of them would go: with run lengths spread over 20 ... 200 the rounds see half of the entries on average.
            uint32_t cur_len = len;
            for (uint32_t e0 = 0; e0 < J; e0 += (uint32_t)kTailWaves) { // uniform
            if (e0 > 0) {
                const uint32_t alive = cf_hr[e0] - cf_hr[e0 - 1u]; // entries with r >= e0: what the rounds from e0 + 1 on need (ends: r == j - 1)
                if ((uint64_t)alive * 8 <= (uint64_t)cur_len * 7) { // uniform
                    constexpr uint32_t kPerC = (kCfEntries + (uint32_t)kTailBlock - 1u) / (uint32_t)kTailBlock;
                    const uint32_t perc = (cur_len + (uint32_t)kTailBlock - 1u) / (uint32_t)kTailBlock;
                    uint32_t kp[kPerC], kr[kPerC], keep = 0, cnt = 0;
                    WT kw[kPerC];
#pragma unroll
                    for (uint32_t k = 0; k < kPerC; ++k) {
                        const uint32_t i = (uint32_t)t * perc + k;
                        kp[k] = 0, kr[k] = 0, kw[k] = 0;
                        if (k < perc && i < cur_len) {
                            kp[k] = cf_p[i], kr[k] = cf_r[i], kw[k] = cf_w[i];
                            if (kr[k] >= e0) keep |= 1u << k, ++cnt;
                        }
                    }
                    const uint32_t inc = wave_inclusive_scan<OpAdd>(cnt);
                    if (lane == kWave - 1) cf_scan[w] = inc;
                    __syncthreads(); // every entry is in its thread's registers
                    uint32_t at = inc - cnt;
                    for (int ww = 0; ww < w; ++ww) at += cf_scan[ww];
#pragma unroll
                    for (uint32_t k = 0; k < kPerC; ++k) {
                        if ((keep >> k) & 1u) {
                            cf_p[at] = kp[k], cf_r[at] = (uint8_t)kr[k], cf_w[at] = kw[k];
                            ++at;
                        }
                    }
                    cur_len = alive;
                    __syncthreads();
                }
            }
            const uint32_t nchunks = (cur_len + (uint32_t)kWave - 1u) / (uint32_t)kWave;
            const uint32_t j = e0 + 1u + (uint32_t)w; // uniform per wave
            if (j <= J) {
                uint32_t cc = cf_hr[j - 1u], cd[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) cd[d] = cf_he[d][j - 1u];
                for (uint32_t ch = 0; ch < nchunks; ++ch) {
                    const uint32_t i = ch * (uint32_t)kWave + (uint32_t)lane;
                    const bool valid = i < cur_len;
                    const uint32_t r = valid ? cf_r[i] : 0u, p = valid ? cf_p[i] : 0u;
                    const WT wend = valid ? cf_w[i] : (WT)0;
                    const uint32_t dd = wnd_first<WT>(wend, cfg); // (the symbol that ends the run, if there is one)
                    const bool cont = valid && r >= j;
                    const bool endf = valid && r + 1u == j && r < kRunCap && p > r && induce_accept(dd, c, mode);
                    const uint64_t bc = __ballot(cont ? 1 : 0);
                    if (bc) {
                        if (cont) {
                            const uint32_t o = cc + (uint32_t)__popcll(bc & lanemask_lt());
                            const uint32_t dst = dir > 0 ? curC + o : curC - 1u - o, pos = p - j;
                            // r - j symbols c, then what lies in front of the run: CW symbols of it at most
                            const uint32_t k = r - j, have = k + wnd_count<WT>(wend);
                            WT nw = 0;
                            if (pos != 0) {
                                const WT codes = k >= cfg.CW ? crun : ((crun & (((WT)1 << (k * B)) - 1u)) | ((wend >> kCntBits) << (k * B))) & field_mask;
                                nw = (codes << kCntBits) | (WT)(have < cfg.CW ? have : cfg.CW);
                            }
                            SA[dst] = pos;
                            WN[dst] = nw;
                            BW[dst] = wnd_symbol<WT>(nw, cfg);
                        }
                        cc += (uint32_t)__popcll(bc);
                    }
                    const uint64_t be = __ballot(endf ? 1 : 0);
                    if (be) {
#pragma unroll
                        for (int d = 1; d < 8; ++d) {
                            const bool mine = endf && dd == (uint32_t)d;
                            const uint64_t bd = __ballot(mine ? 1 : 0);
                            if (bd) {
                                if (mine) {
                                    const uint32_t o = cd[d] + (uint32_t)__popcll(bd & lanemask_lt());
                                    const uint32_t dst = dir > 0 ? curD[d] + o : curD[d] - 1u - o, pos = p - j;
                                    const WT nw = pos == 0 ? (WT)0 : wnd_pop<WT>(wend, cfg); // (pos = p - r - 1 > 0: the window held two symbols at least)
                                    SA[dst] = pos;
                                    WN[dst] = nw;
                                    BW[dst] = wnd_symbol<WT>(nw, cfg);
                                }
                                cd[d] += (uint32_t)__popcll(bd);
                            }
                        }
                    }
                }
            }
            __syncthreads(); // (the next sixteen may begin by moving the entries)
            } // (sixteen rounds)
            if (t < 8) {
                // everything written: the cursors behind it; what is still inside its run after kRunCap rounds is the next range
                const uint32_t tot = (uint32_t)t == c ? cf_hr[kRunCap] : cf_he[t][kRunCap]; // (sums below the last index: a run of
                                                                                           //  kRunCap is never an end, and round kRunCap's entries lie below cf_hr[kRunCap])
                const uint32_t before = gbase[t];
                gbase[t] = dir > 0 ? before + tot : before - tot;
                if ((uint32_t)t == c) {
                    const uint32_t from = rmax < kRunCap ? tot : cf_hr[kRunCap - 1u];
                    s_range[0] = dir > 0 ? before + from : before - tot;
                    s_range[1] = dir > 0 ? before + tot : before - from;
                }
            }
            held = false;
            last_batch = true;
            last_in = len;
            it += J;
            __syncthreads();
            continue;
        }
        if (!held) { // the range's entries from memory (the first round of a launch, or after a jump)
            live = 0;
            // (a range of 1700 entries as 8 to a thread would keep four waves busy, one to a SIMD, every wait of theirs
            // in the open: two to a thread spread it over all sixteen)
            per = (len + (uint32_t)kTailBlock - 1u) / (uint32_t)kTailBlock;
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) {
                const uint32_t i = (uint32_t)t * per + (uint32_t)k;
                val[k] = 0;
                wnd[k] = 0;
                if ((uint32_t)k < per && i < len) {
                    const uint32_t idx = lo + (rev ? len - 1u - i : i);
                    val[k] = SA[idx];
                    wnd[k] = WN[idx];
                    live |= 1u << k;
                }
            }
            held = true;
        }
        // rounds of this step: eight when the last step kept at least an eighth of its entries (runs), else one
        const uint32_t nr = (uint64_t)len * 8 >= last_in ? (uint32_t)kTailBatch : 1u; // uniform
        last_in = len;
        last_batch = nr > 1;
        if (nr > 1) { // windows that do not reach nr + 1 symbols deep are refilled first, all of a thread's refills in flight together
            WT fresh[kIndItems];
            uint32_t dry = 0;
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) {
                fresh[k] = 0;
                if ((uint32_t)k < per && ((live >> k) & 1u) && val[k] != 0) { // (k < per: uniform)
                    const uint32_t need = val[k] < (uint32_t)(kTailBatch + 1) ? val[k] : (uint32_t)(kTailBatch + 1);
                    if (wnd_count<WT>(wnd[k]) < need) {
                        dry |= 1u << k;
                        fresh[k] = wnd_fill<WT>(T, val[k], cfg);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kIndItems; ++k)
                if ((dry >> k) & 1u) wnd[k] = fresh[k];
        }
        // A range of a few hundred entries keeps one or two waves busy; the others only take part in the barriers
        // (every instruction a wave of this 16-wave workgroup issues costs the CU a slot).
        const bool wave_live = __any(live != 0u ? 1 : 0); // uniform per wave
        // ---- count: per round, the thread's entries per bucket (8-bit fields), and each entry's rank inside the thread ----
        uint64_t cnt[kTailBatch];
        uint32_t emask[kIndItems]; // bit j: the round-j descendant exists and is accepted; bits 8 + 3 j ..: its rank in the thread
        uint32_t alive_after = 0;  // bit k: entry k's descendant of the last round stayed in bucket c
        uint64_t ex0[kTailBatch], ex1[kTailBatch];
#pragma unroll
        for (int j = 0; j < kTailBatch; ++j) cnt[j] = 0, ex0[j] = 0, ex1[j] = 0;
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) emask[k] = 0;
        if (wave_live) {
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) {
                if ((uint32_t)k >= per) break; // uniform
                bool alive = (live >> k) & 1u;
                const WT codes = wnd[k] >> kCntBits;
#pragma unroll
                for (int j = 0; j < kTailBatch; ++j) {
                    if ((uint32_t)j < nr) { // uniform
                        const uint32_t sym = ((uint32_t)(codes >> (j * B)) & cmask) + 1u;
                        const bool ok = alive && val[k] > (uint32_t)j && induce_accept(sym, c, mode);
                        if (ok) {
                            const uint32_t sh = 8u * (sym & 7u);
                            emask[k] |= (1u << j) | (((uint32_t)(cnt[j] >> sh) & 7u) << (8 + 3 * j));
                            cnt[j] += 1ull << sh;
                        }
                        alive = ok && sym == c;
                    }
                }
                if (alive) alive_after |= 1u << k;
            }
            // ---- entries of earlier threads, per round and bucket: two 64-bit words of 16-bit fields, scanned over the workgroup ----
#pragma unroll
            for (int j = 0; j < kTailBatch; ++j) {
                if ((uint32_t)j < nr) { // uniform
                    const uint64_t own0 = cnt[j] & kField16, own1 = (cnt[j] >> 8) & kField16;
                    const uint64_t inc0 = wave_inclusive_sum_packed(own0), inc1 = wave_inclusive_sum_packed(own1);
                    if (lane == kWave - 1) wsum[j][0][w] = inc0, wsum[j][1][w] = inc1;
                    ex0[j] = inc0 - own0, ex1[j] = inc1 - own1;
                }
            }
        } else if (lane < kTailBatch * 2) {
            wsum[lane >> 1][lane & 1][w] = 0;
        }
        __syncthreads();
        if (t < kTailBatch * 2 * kTailWaves) { // (round, half, wave): the 16 wave totals of a (round, half) scanned by 16 lanes
            const int j = t / (2 * kTailWaves), h = (t / kTailWaves) & 1, ww = t % kTailWaves;
            static_assert(kTailWaves == 16, "a (round, half) is scanned by a 16-lane segment");
            const uint64_t own = (uint32_t)j < nr ? wsum[j][h][ww] : 0ull;
            const uint64_t inc = row_inclusive_sum_packed(own); // (a segment of 16 lanes is a DPP row)
            if ((uint32_t)j < nr) {
                wsum[j][h][ww] = inc - own;
                if (ww == kTailWaves - 1) { // the round's totals of four buckets
#pragma unroll
                    for (int f = 0; f < 4; ++f) s_tot[j][2 * f + h] = (uint32_t)(inc >> (16 * f)) & 0xFFFFu;
                }
            }
        }
        __syncthreads();
        if (t < 8) { // the bucket's cursor before every round of the step
            uint32_t b = gbase[t];
#pragma unroll
            for (int j = 0; j < kTailBatch; ++j) {
                if ((uint32_t)j < nr) {
                    s_base[j][t] = b;
                    b = dir > 0 ? b + s_tot[j][t] : b - s_tot[j][t];
                }
            }
            gbase[t] = b;
            if ((uint32_t)t == c) { // the last round's entries for bucket c are the next range
                const uint32_t sb = s_base[nr - 1u][t], n_last = s_tot[nr - 1u][t];
                s_range[0] = dir > 0 ? sb : sb - n_last;
                s_range[1] = dir > 0 ? sb + n_last : sb;
            }
        }
        __syncthreads();
        // ---- scatter ---------------------------------------------------------------------------------------------------
        uint32_t live_next = 0;
        if (wave_live) {
#pragma unroll
        for (int j = 0; j < kTailBatch; ++j) // entries of earlier waves: all of the step's reads in flight together
            if ((uint32_t)j < nr) ex0[j] += wsum[j][0][w], ex1[j] += wsum[j][1][w];
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            if ((uint32_t)k >= per) break; // uniform
            const WT codes = wnd[k] >> kCntBits;
            const uint32_t wcnt = wnd_count<WT>(wnd[k]);
#pragma unroll
            for (int j = 0; j < kTailBatch; ++j) {
                if ((uint32_t)j < nr && ((emask[k] >> j) & 1u)) {
                    const uint32_t d = (((uint32_t)(codes >> (j * B)) & cmask) + 1u) & 7u;
                    const uint64_t exw = (d & 1u) ? ex1[j] : ex0[j];
                    const uint32_t r = ((uint32_t)(exw >> (16u * (d >> 1))) & 0xFFFFu) + ((emask[k] >> (8 + 3 * j)) & 7u);
                    const uint32_t sb = s_base[j][d];
                    const uint32_t dst = dir > 0 ? sb + r : sb - 1u - r;
                    const uint32_t pos = val[k] - (uint32_t)(j + 1);
                    WT nw = (((codes >> (j * B)) >> B) << kCntBits) | (WT)(wcnt - (uint32_t)(j + 1)); // j + 1 symbols popped
                    if (pos != 0 && wcnt == (uint32_t)(j + 1)) nw = wnd_fill<WT>(T, pos, cfg); // window ran dry: back to the text
                    SA[dst] = pos;
                    WN[dst] = nw;
                    BW[dst] = wnd_symbol<WT>(nw, cfg);
                    if ((uint32_t)j == nr - 1u && ((alive_after >> k) & 1u)) { // stays in bucket c: the entry of the next step
                        live_next |= 1u << k;
                        val[k] = pos;
                        wnd[k] = nw;
                    }
                }
            }
        }
        }
        live = live_next;
        it += nr;
        __syncthreads(); // (s_range, s_base and wsum are rewritten by the next step)
    }
    if (t < 8) cursor_nxt[t] = gbase[t];
    if (t == 0) {
        range_out[0] = s_range[0];
        range_out[1] = s_range[1];
        tail_report(s_range[0], s_range[1], c, poison, host_poison);
    }
}

// ---- the self rounds of a bucket, eight at a time (at most 8 buckets) ------------------------------------------
// Round k of bucket c reads what round k-1 appended to c, and on ordinary text every round is a quarter of the one
// before: after the first (large) round a bucket went through a dozen launches that moved next to nothing, each with
// its launch latency (1 GiB of DNA: 0.3 ms of 1.0 per bucket region).  An entry's window already says where its
// descendants go: with a = the number of symbols c immediately to its left, the descendants of rounds 0 .. a-1 stay in
// bucket c (position - 1 ... position - a) and the one of round a goes to the bucket of the first other symbol, if
// the type test accepts it (as induce_tail_small_kernel does inside one workgroup).  So kBatchRounds rounds are taken
// by one counting launch (per tile: outputs per round and bucket), one scan of the 64 count rows and one scatter:
// round j's outputs into bucket d lie behind those of rounds < j, tiles in order inside a round.  The last round's
// outputs into bucket c are the next range.  A window that shows only symbols c and is shorter than the rounds ahead
// is refilled from the text first (by both kernels alike).
constexpr int kBatchRounds = 8;
constexpr int kBatchRows = kBatchRounds * 8; // (round, bucket) count rows
#ifndef SX_BATCH_FROM
#define SX_BATCH_FROM (1u << 21)
#endif
constexpr uint32_t kBatchFrom = SX_BATCH_FROM;    // rounds expected to hold more entries than this are launches of their own

template <class WT> struct batch_plan {
    uint32_t a;    // descendants that stay in bucket c (rounds 0 .. a-1), at most kBatchRounds
    uint32_t tsym; // bucket of the round-a descendant, when `term`
    bool term;
};

// field index of the lowest set bit of x (fields of B bits)
__device__ __forceinline__ uint32_t batch_field_of(uint32_t bit, uint32_t B)
{
    return B == 2 ? bit >> 1 : (B == 1 ? bit : (B == 3 ? (bit * 171u) >> 9 : bit >> 2)); // (B uniform, bit < 64)
}

template <class WT, int MODE>
__device__ __forceinline__ batch_plan<WT> batch_chain(WT &w, uint32_t p, uint32_t c, const wnd_cfg &cfg, WT cpat,
                                                        const uint8_t *__restrict__ T)
{
    batch_plan<WT> pl;
    uint32_t cntw = wnd_count<WT>(w);
    WT x = (w >> kCntBits) ^ cpat;
    uint32_t r = x ? batch_field_of((uint32_t)(sizeof(WT) == 8 ? __builtin_ctzll((unsigned long long)x) : __builtin_ctz((uint32_t)x)), cfg.B)
                   : cfg.CW;
    if (r >= cntw && cntw <= (uint32_t)kBatchRounds && p > cntw) {
        // every symbol the window holds is c and the text goes on to the left: look further (rare)
        w = wnd_fill<WT>(T, p, cfg);
        cntw = wnd_count<WT>(w);
        x = (w >> kCntBits) ^ cpat;
        r = x ? batch_field_of((uint32_t)(sizeof(WT) == 8 ? __builtin_ctzll((unsigned long long)x) : __builtin_ctz((uint32_t)x)), cfg.B)
              : cfg.CW;
    }
    uint32_t a = r < cntw ? r : cntw;
    pl.term = a < (uint32_t)kBatchRounds && a < cntw;
    if (a > (uint32_t)kBatchRounds) a = (uint32_t)kBatchRounds;
    pl.a = a;
    pl.tsym = ((uint32_t)((w >> kCntBits) >> (a * cfg.B)) & cfg.mask) + 1u;
    pl.term = pl.term && induce_accept(pl.tsym, c, MODE) && pl.tsym < 8u;
    return pl;
}

// window of the descendant `depth` + 1 positions to the left (depth + 1 symbols popped), back to the text when it ran dry
template <class WT>
__device__ __forceinline__ WT batch_window(WT w, uint32_t depth, uint32_t pos, const wnd_cfg &cfg, const uint8_t *__restrict__ T)
{
    const uint32_t cntw = wnd_count<WT>(w);
    WT nw = ((((w >> kCntBits) >> (depth * cfg.B)) >> cfg.B) << kCntBits) | (WT)(cntw - (depth + 1u));
    if (pos != 0 && cntw == depth + 1u) nw = wnd_fill<WT>(T, pos, cfg);
    return nw;
}

template <class WT> __device__ __forceinline__ WT batch_cpat(uint32_t c, const wnd_cfg &cfg)
{
    WT pat = 0;
    for (uint32_t i = 0; i < cfg.CW; ++i) pat |= (WT)(c - 1u) << (i * cfg.B); // uniform
    return pat;
}

// the thread's 8 consecutive entries of the scan order (windows; positions too when wanted)
template <class WT, bool kRev, bool kWantP>
__device__ __forceinline__ void batch_load(const uint32_t *__restrict__ srcP, const WT *__restrict__ srcW, uint32_t lo, uint32_t len,
                                           uint32_t i0, uint32_t (&P)[kIndItems], WT (&W)[kIndItems])
{
    if (i0 + kIndItems <= len) {
        const uint32_t first = kRev ? lo + len - i0 - kIndItems : lo + i0;
        uint32_t Pm[kIndItems];
        WT Wm[kIndItems];
        if (kWantP) __builtin_memcpy(Pm, srcP + first, sizeof(Pm));
        __builtin_memcpy(Wm, srcW + first, sizeof(Wm));
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            P[k] = kWantP ? Pm[kRev ? kIndItems - 1 - k : k] : 0u;
            W[k] = Wm[kRev ? kIndItems - 1 - k : k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            const uint32_t i = i0 + (uint32_t)k;
            const uint32_t idx = i < len ? (kRev ? lo + len - 1u - i : lo + i) : lo;
            P[k] = (kWantP && i < len) ? srcP[idx] : 0u;
            W[k] = i < len ? srcW[idx] : (WT)0;
        }
    }
}

// per round: this thread's outputs per bucket, 8-bit fields (at most 8 entries a thread)
template <class WT, int MODE>
__device__ __forceinline__ void batch_tally(const batch_plan<WT> &pl, uint32_t c, uint64_t (&cnt)[kBatchRounds])
{
#pragma unroll
    for (int j = 0; j < kBatchRounds; ++j) {
        const uint64_t self = (uint32_t)j < pl.a ? 1ull << (8u * c) : 0ull;
        const uint64_t term = (pl.term && pl.a == (uint32_t)j) ? 1ull << (8u * pl.tsym) : 0ull;
        cnt[j] += self + term;
    }
}

template <class WT, int MODE>
__global__ __launch_bounds__(kBlock) void induce_batch_count_kernel(const uint32_t *__restrict__ srcP, const WT *__restrict__ srcW,
                                                                    const uint32_t *__restrict__ range_in, uint32_t c, wnd_cfg cfg,
                                                                    const uint8_t *__restrict__ T, uint32_t *__restrict__ hist /* [row][stride] */,
                                                                    uint32_t stride, uint32_t min_len)
{
    constexpr bool kRev = MODE == MODE_S_FROM_S;
    constexpr uint64_t kField16 = 0x00FF00FF00FF00FFull;
    __shared__ uint64_t wtot[kBatchRounds][2][kWavesPerBlock];
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t lo = range_in[0], len = range_in[1] - lo;
    if (len <= min_len) return;
    const uint32_t ntiles = (len + kIndTile - 1) / kIndTile;
    const WT cpat = batch_cpat<WT>(c, cfg);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // uniform per workgroup
        const uint32_t i0 = tile * (uint32_t)kIndTile + (uint32_t)t * kIndItems;
        uint32_t P[kIndItems];
        WT W[kIndItems];
        batch_load<WT, kRev, false>(srcP, srcW, lo, len, i0, P, W);
        uint64_t cnt[kBatchRounds];
#pragma unroll
        for (int j = 0; j < kBatchRounds; ++j) cnt[j] = 0;
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            const uint32_t cntw = wnd_count<WT>(W[k]);
            uint32_t p = cntw; // (the position matters only to a window that may need more symbols: fetched then)
            if (cntw != 0 && cntw <= (uint32_t)kBatchRounds) {
                const WT x = (W[k] >> kCntBits) ^ cpat;
                const WT low = cntw * cfg.B >= sizeof(WT) * 8 ? ~(WT)0 : (((WT)1 << (cntw * cfg.B)) - 1);
                if ((x & low) == 0) { // all of its symbols are c
                    const uint32_t i = i0 + (uint32_t)k;
                    p = srcP[kRev ? lo + len - 1u - i : lo + i];
                }
            }
            const batch_plan<WT> pl = batch_chain<WT, MODE>(W[k], p, c, cfg, cpat, T);
            batch_tally<WT, MODE>(pl, c, cnt);
        }
#pragma unroll
        for (int j = 0; j < kBatchRounds; ++j) {
            const uint64_t e = wave_total_packed(cnt[j] & kField16), o = wave_total_packed((cnt[j] >> 8) & kField16);
            if (lane == 0) wtot[j][0][w] = e, wtot[j][1][w] = o;
        }
        __syncthreads();
        if (t < kBatchRows) { // row (j, d): the tile's outputs of round j into bucket d
            const int j = t >> 3, d = t & 7;
            uint64_t sum = 0;
#pragma unroll
            for (int ww = 0; ww < kWavesPerBlock; ++ww) sum += wtot[j][d & 1][ww];
            hist[(uint64_t)t * stride + tile] = (uint32_t)(sum >> (16 * (d >> 1))) & 0xFFFFu;
        }
        __syncthreads();
    }
}

// one workgroup per (round, bucket) row: exclusive prefix over the tiles, the row's total aside
__global__ __launch_bounds__(kRowThreads) void induce_batch_offsets_kernel(uint32_t *__restrict__ hist, uint32_t stride,
                                                                      const uint32_t *__restrict__ range_in,
                                                                      uint32_t *__restrict__ totals, uint32_t nk, uint32_t min_len)
{
    __shared__ uint32_t lds[kRowPieces * kRowWaves];
    const uint32_t len = range_in[1] - range_in[0];
    if (len <= min_len) return;
    const uint32_t ntiles = (len + kIndTile - 1) / kIndTile;
    const uint32_t row = (blockIdx.x / nk) * 8u + blockIdx.x % nk;
    const uint32_t total = wide_scan_row_inplace(hist + (uint64_t)row * stride, ntiles, lds);
    if (threadIdx.x == 0) totals[row] = total;
}

template <class WT, int MODE>
__global__ __launch_bounds__(kBlock, 2) void induce_batch_scatter_kernel(
    const uint32_t *__restrict__ srcP, const WT *__restrict__ srcW, const uint32_t *__restrict__ range_in,
    uint32_t *__restrict__ range_out, uint32_t c, wnd_cfg cfg, const uint8_t *__restrict__ T, const uint32_t *__restrict__ offs,
    uint32_t stride, const uint32_t *__restrict__ totals, const uint32_t *__restrict__ cursor_cur, uint32_t *__restrict__ cursor_nxt,
    uint32_t *__restrict__ SA, WT *__restrict__ WN, uint8_t *__restrict__ BW, uint32_t nk, uint32_t min_len)
{
    constexpr bool kRev = MODE == MODE_S_FROM_S;
    constexpr uint64_t kField16 = 0x00FF00FF00FF00FFull;
    __shared__ uint64_t wsum[kBatchRounds][2][kWavesPerBlock];
    __shared__ uint32_t s_round0[kBatchRows]; // outputs of earlier rounds into the bucket (the round's first slot, relative)
    __shared__ uint32_t s_base[kBatchRows];   // destination of the tile's first output of (round, bucket)
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t lo = range_in[0], hi = range_in[1], len = hi - lo;
    if (len <= min_len) { // not a range for this form: the cursors and the range go on as they are
        if (blockIdx.x == 0) {
            if (t < 256) cursor_nxt[t] = cursor_cur[t];
            if (t == 0 && range_out) range_out[0] = lo, range_out[1] = hi;
        }
        return;
    }
    const uint32_t ntiles = (len + kIndTile - 1) / kIndTile;
    __shared__ uint32_t s_tot[kBatchRows];
    if (t < kBatchRows) s_tot[t] = (uint32_t)(t & 7) < nk ? totals[t] : 0u; // (one trip to memory for all of them)
    __syncthreads();
    if (t < kBatchRows) {
        const int j = t >> 3, d = t & 7;
        uint32_t before = 0;
        for (int jj = 0; jj < j; ++jj) before += s_tot[jj * 8 + d];
        s_round0[t] = before;
    }
    __syncthreads();
    if (blockIdx.x == 0 && t < 256) { // the cursors after all the rounds; the last round's outputs into bucket c are the next range
        uint32_t all = 0;
        if ((uint32_t)t < nk) all = s_round0[(kBatchRounds - 1) * 8 + t] + s_tot[(kBatchRounds - 1) * 8 + t];
        const uint32_t cur = cursor_cur[t];
        cursor_nxt[t] = kRev ? cur - all : cur + all;
        if ((uint32_t)t == c && range_out) {
            const uint32_t last = s_tot[(kBatchRounds - 1) * 8 + t];
            range_out[0] = kRev ? cur - all : cur + all - last;
            range_out[1] = kRev ? cur - all + last : cur + all;
        }
    }
    const uint32_t base_d = (uint32_t)(t & 7) < nk && t < kBatchRows ? cursor_cur[t & 7] : 0u;
    const WT cpat = batch_cpat<WT>(c, cfg);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // uniform per workgroup
        if (t < kBatchRows) {
            const uint32_t rel = s_round0[t] + (((uint32_t)(t & 7) < nk) ? offs[(uint64_t)t * stride + tile] : 0u);
            s_base[t] = kRev ? base_d - 1u - rel : base_d + rel;
        }
        const uint32_t i0 = tile * (uint32_t)kIndTile + (uint32_t)t * kIndItems;
        uint32_t P[kIndItems];
        WT W[kIndItems];
        batch_load<WT, kRev, true>(srcP, srcW, lo, len, i0, P, W);
        batch_plan<WT> pl[kIndItems];
        uint64_t cnt[kBatchRounds];
#pragma unroll
        for (int j = 0; j < kBatchRounds; ++j) cnt[j] = 0;
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            pl[k] = batch_chain<WT, MODE>(W[k], P[k], c, cfg, cpat, T);
            batch_tally<WT, MODE>(pl[k], c, cnt);
        }
        // outputs of earlier threads per round and bucket: two words of 16-bit fields a round, scanned over the workgroup
        uint64_t ex0[kBatchRounds], ex1[kBatchRounds];
#pragma unroll
        for (int j = 0; j < kBatchRounds; ++j) {
            const uint64_t own0 = cnt[j] & kField16, own1 = (cnt[j] >> 8) & kField16;
            const uint64_t inc0 = wave_inclusive_sum_packed(own0), inc1 = wave_inclusive_sum_packed(own1);
            if (lane == kWave - 1) wsum[j][0][w] = inc0, wsum[j][1][w] = inc1;
            ex0[j] = inc0 - own0, ex1[j] = inc1 - own1;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kBatchRounds; ++j) {
#pragma unroll
            for (int ww = 0; ww < kWavesPerBlock; ++ww)
                if (ww < w) ex0[j] += wsum[j][0][ww], ex1[j] += wsum[j][1][ww];
        }
        uint64_t run[kBatchRounds]; // outputs of this thread's earlier entries, 8-bit fields
#pragma unroll
        for (int j = 0; j < kBatchRounds; ++j) run[j] = 0;
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
#pragma unroll
            for (int j = 0; j < kBatchRounds; ++j) {
                const bool self = (uint32_t)j < pl[k].a, term = pl[k].term && pl[k].a == (uint32_t)j;
                if (self || term) {
                    const uint32_t d = self ? c : pl[k].tsym;
                    const uint64_t exw = (d & 1u) ? ex1[j] : ex0[j];
                    const uint32_t r = ((uint32_t)(exw >> (16u * (d >> 1))) & 0xFFFFu) + ((uint32_t)(run[j] >> (8u * d)) & 0xFFu);
                    const uint32_t sb = s_base[j * 8 + (int)d];
                    const uint32_t dst = kRev ? sb - r : sb + r;
                    const uint32_t pos = P[k] - (uint32_t)(j + 1);
                    const WT nw = batch_window<WT>(W[k], (uint32_t)j, pos, cfg, T);
                    SA[dst] = pos;
                    WN[dst] = nw;
                    BW[dst] = wnd_symbol<WT>(nw, cfg);
                    run[j] += 1ull << (8u * d);
                }
            }
        }
        __syncthreads(); // s_base and wsum are rewritten for the next tile
    }
}

} // namespace sx
