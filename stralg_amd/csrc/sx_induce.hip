// sx_induce.hip -- the forward (L) and backward (S) induced-sort passes.
//
// stralg/sa_is.c:220-242 induce_L scans SA left to right and appends
// j = SA[i]-1 to the head of bucket text[j] when j is L-type; sa_is.c:245-263
// induce_S mirrors it right to left for S-type.  The scan is loop-carried
// (entries written ahead of the cursor are read later), so the device version
// walks the buckets in the same order and splits each bucket's work into
// rounds whose entries are independent:
//
//   bucket c, round 0 : every entry induced into c from earlier buckets
//   bucket c, round k : the entries round k-1 induced into c itself
//                       (same symbol to the left: a run of c's)
//   then              : the bucket's other region (LMS seeds in the L pass,
//                       the L region in the S pass)
//
// A round is a stable multi-way split by text[SA[i]-1]: gather + per-tile
// histogram, per-bucket offsets, stable scatter to the bucket cursors.  The
// type test needs no type array: for an L-type entry p of bucket c, p-1 is
// L-type iff text[p-1] >= c; for an S-type entry, p-1 is S-type iff
// text[p-1] <= c (equal symbols share the type of their right neighbour).
//
// Cost: streaming.  Every entry carries a window of the symbols to its left
// (filled from the text once per LMS seed and again only when it runs dry), so
// a round reads (entry, window) pairs and writes them to <= sigma sequential
// streams; after the S pass the windows' first symbols are the BWT.
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_scan.hpp"
#include "sx_internal.hpp"
#include "sx_window.hpp"

namespace sx {

constexpr int kIndItems = 8;
constexpr int kIndTile = kBlock * kIndItems;
// Rounds of up to 8192 entries are left to the tail kernel (one workgroup of 1024 threads, many rounds per launch):
// a chained launch costs ~18 us whatever it holds, a round of the tail a few.  The rounds of a bucket shrink with the
// run length of its symbol, so texts with poly-A tracts and microsatellites spend hundreds of rounds at a few
// thousand entries (a genome-like 1 GiB text: 489 chained launches, 9 ms).  (Four 2048-entry tiles one after the other
// in a 256-thread workgroup were five times slower than the chained launches: every tile pays the load latency.)
constexpr int kTailBlock = 1024, kTailWaves = kTailBlock / kWave; // the tail kernel's workgroup: 16 waves, one tile
constexpr int kTailTile = kTailBlock * kIndItems;
constexpr uint32_t kTailEntries = (uint32_t)kTailTile;
constexpr uint32_t kTailMulti = 4; // more than 8 buckets: tiles of a round the tail kernel takes one after the other

enum { MODE_L_FROM_L = 0, MODE_L_FROM_LMS = 1, MODE_S_FROM_S = 2, MODE_S_FROM_L = 3 };
__device__ __forceinline__ void tail_report(uint32_t lo, uint32_t hi, uint32_t c, uint32_t *poison, uint32_t *host_poison);

__device__ __forceinline__ bool induce_accept(uint32_t ch, uint32_t c, int mode)
{
    switch (mode) {
    case MODE_L_FROM_L: return ch >= c;
    case MODE_L_FROM_LMS: return true;
    case MODE_S_FROM_S: return ch <= c;
    default: return ch < c;
    }
}

template <class WT>
__global__ __launch_bounds__(kBlock) void fill_windows_kernel(const uint8_t *__restrict__ T,
                                                              const uint32_t *__restrict__ pos, uint64_t count,
                                                              wnd_cfg cfg, WT *__restrict__ out)
{
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    const uint32_t p = pos[k];
    out[k] = p ? wnd_fill<WT>(T, p, cfg) : (WT)0;
}

// ---- large rounds: count, offsets, scatter ----------------------------------------------
// A round whose range is longer than chain_max entries is split in three launches (the
// entries are read twice); shorter rounds take the single chained launch below, whose
// look-back walk costs a few microseconds per tile and would dominate a long round.
// Both forms are queued for every round; each checks the range and returns at once when
// the round belongs to the other.
// (threshold: sx_ctx::chain_max_entries, default 256 tiles; SX_FLAG_CHAIN_MAX_ENTRIES)

// four consecutive entries from 16-byte loads
__device__ __forceinline__ void load_quad(const uint32_t *__restrict__ p, uint32_t (&o)[4])
{
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    o[0] = v.x, o[1] = v.y, o[2] = v.z, o[3] = v.w;
}
__device__ __forceinline__ void load_quad(const uint64_t *__restrict__ p, uint64_t (&o)[4])
{
    const uint4 v0 = *reinterpret_cast<const uint4 *>(p), v1 = *reinterpret_cast<const uint4 *>(p + 2);
    o[0] = pack64(v0.x, v0.y), o[1] = pack64(v0.z, v0.w), o[2] = pack64(v1.x, v1.y), o[3] = pack64(v1.z, v1.w);
}

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
            for (int k = 0; k < ITEMS; ++k) need[k] = (lpos[k] & 0x8000u) && val[k] != 0 && wnd_count<WT>(wnd[k]) <= refill_at;
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

// ---- one round = one launch -----------------------------------------------------------
// Stable multi-way split of the entries in range_in (read from device memory, so rounds
// can be queued without the host knowing their sizes): entry p with window w induces
// p-1 into bucket text[p-1] (= the window's first symbol) when the type test accepts
// it.  Tiles take tickets; per destination bucket the tile-local counts are chained
// across tiles by decoupled look-back (sx_device.hpp), so the entries are read once.
// The last tile publishes the advanced bucket cursors and the range appended to
// bucket c, which is the next round's input.
template <class WT, int BITS>
__global__ __launch_bounds__(kBlock) void induce_round_kernel(
    const uint32_t *__restrict__ srcP, const WT *__restrict__ srcW, const uint32_t *__restrict__ range_in,
    uint32_t *__restrict__ range_out, int rev, int mode, uint32_t c, wnd_cfg cfg, const uint8_t *__restrict__ T,
    const uint32_t *__restrict__ cursor_cur, uint32_t *__restrict__ cursor_nxt, int dir, uint32_t *__restrict__ SA,
    WT *__restrict__ WN, uint8_t *__restrict__ BW, uint32_t nkeys, uint64_t *__restrict__ status, uint32_t epoch,
    uint32_t *__restrict__ ticket,
    uint32_t chain_max /* rounds longer than this are left to the three-launch form; ~0u: take any round */,
    int tail_follows /* the batch ends with the tail kernel: rounds of up to kTailEntries entries are left to it */,
    int pass_large /* a round longer than chain_max is nobody's here: hand it on as it is (the host queues it again) */)
{
    __shared__ uint32_t wcount[kWavesPerBlock][256];
    __shared__ uint32_t gpos[256];  // entries of earlier tiles per bucket
    __shared__ uint32_t gbase[256]; // bucket cursors at the start of the round
    __shared__ uint32_t tcount[256]; // this tile's entries per bucket
    __shared__ uint32_t s_tile;
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t lo = range_in[0], hi = range_in[1];
    const uint32_t len = hi - lo;
    if (len > chain_max && !pass_large) return; // a large round: the three-launch form handles it
    if (len == 0 || (range_out && tail_follows && len <= (BITS > 3 ? kTailMulti * kTailEntries : kTailEntries)) || len > chain_max) {
        // nothing to do, or a round small enough for the tail kernel that ends the batch: carry the cursors over,
        // hand the range on as it is
        if (blockIdx.x == 0) {
            cursor_nxt[t] = cursor_cur[t];
            if (t == 0 && range_out) {
                range_out[0] = len ? lo : hi;
                range_out[1] = hi;
            }
        }
        return;
    }
    const uint32_t ntiles = (len + kIndTile - 1) / kIndTile;
    gbase[t] = cursor_cur[t];
    for (;;) {
        if (t == 0) s_tile = atomicAdd(ticket, 1u);
        for (int i = t; i < kWavesPerBlock * 256; i += kBlock) (&wcount[0][0])[i] = 0;
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= ntiles) break;
        const uint32_t wave0 = tile * (uint32_t)kIndTile + (uint32_t)w * (kWave * kIndItems);
        uint32_t val[kIndItems], dig[kIndItems], rnk[kIndItems];
        WT wnd[kIndItems];
        bool ok[kIndItems];
        // (all of the tile's loads are issued before the first is looked at: see wide_scatter_tile)
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            const uint32_t i = wave0 + (uint32_t)k * kWave + lane;
            const uint32_t idx = lo + (i < len ? (rev ? len - 1u - i : i) : 0u);
            val[k] = srcP[idx];
            wnd[k] = srcW[idx];
        }
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            const uint32_t i = wave0 + (uint32_t)k * kWave + lane;
            const uint32_t p = i < len ? val[k] : 0u;
            const WT ww = wnd[k];
            ok[k] = false;
            dig[k] = 0;
            val[k] = 0;
            wnd[k] = 0;
            if (p != 0) {
                const uint32_t ch = wnd_first<WT>(ww, cfg);
                ok[k] = induce_accept(ch, c, mode);
                dig[k] = ch;
                val[k] = p - 1u;
                wnd[k] = wnd_pop<WT>(ww, cfg);
            }
        }
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) rnk[k] = wave_rank_step<BITS>(dig[k], ok[k], wcount[w]);
        __syncthreads();
        {
            const uint32_t d = (uint32_t)t;
            uint32_t cnt = 0;
#pragma unroll
            for (int ww = 0; ww < kWavesPerBlock; ++ww) {
                const uint32_t x = wcount[ww][d];
                wcount[ww][d] = cnt;
                cnt += x;
            }
            uint32_t excl = 0;
            if (BITS > 3) { // one thread per bucket walks back on its own
                if (d < nkeys) excl = chain_exclusive_prefix(status, nkeys, tile, d, cnt, epoch);
                gpos[d] = excl;
                tcount[d] = cnt;
            } else {
                tcount[d] = cnt;
            }
        }
        if (BITS <= 3) { // <= 8 buckets: a whole wave walks back for each of them, 64 tiles a step
            __syncthreads();
            for (uint32_t d = (uint32_t)w; d < nkeys; d += kWavesPerBlock) {
                const uint32_t excl = chain_exclusive_prefix_wave(status, nkeys, tile, d, tcount[d], epoch);
                if (lane == 0) gpos[d] = excl;
            }
        }
        __syncthreads();
        if (tile == ntiles - 1) {
            const uint32_t d = (uint32_t)t;
            const uint32_t total = (d < nkeys ? gpos[d] : 0u) + tcount[d], cur = gbase[d];
            cursor_nxt[d] = dir > 0 ? cur + total : cur - total;
            if (d == c && range_out) {
                range_out[0] = dir > 0 ? cur : cur - total;
                range_out[1] = dir > 0 ? cur + total : cur;
            }
        }
        __syncthreads();
        {
            bool need[kIndItems]; // windows that ran dry: back to the text, all of a thread's reads in flight together
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) need[k] = ok[k] && val[k] != 0 && wnd_count<WT>(wnd[k]) == 0;
            refill_windows<WT, kIndItems>(T, val, need, cfg, wnd);
        }
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            if (ok[k]) {
                const uint32_t d = dig[k];
                const uint32_t r = gpos[d] + wcount[w][d] + rnk[k];
                const uint32_t dst = dir > 0 ? gbase[d] + r : gbase[d] - 1u - r;
                const WT nw = wnd[k];
                SA[dst] = val[k];
                WN[dst] = nw;
                BW[dst] = wnd_symbol<WT>(nw, cfg);
            }
        }
        __syncthreads(); // LDS is reused by the next tile
    }
}

// ---- the tail of a bucket's rounds: one workgroup, many rounds, one launch -------------
// Once a round fits one tile, its successors are smaller still (each keeps only the entries
// whose run of symbol c goes on), and a launch per round is all latency.  This kernel runs
// successive rounds of bucket c in a single workgroup -- read <= one tile, rank, scatter,
// advance the cursors held in LDS -- until the range is empty, grows beyond a tile (it
// cannot, but then the host's ordinary rounds take over) or max_iters rounds have run.
// Entries written in one iteration are read in the next by other waves of the same
// workgroup: the barrier's workgroup-scope fence orders them (the waves share the CU's L1).
template <class WT, int BITS>
__global__ __launch_bounds__(kTailBlock) void induce_tail_kernel(uint32_t *SA, WT *WN, uint8_t *BW, const uint32_t *__restrict__ range_in,
                                                             uint32_t *__restrict__ range_out, int rev, int mode,
                                                             uint32_t c, wnd_cfg cfg, const uint8_t *__restrict__ T,
                                                             const uint32_t *__restrict__ cursor_cur,
                                                             uint32_t *__restrict__ cursor_nxt, int dir,
                                                             uint32_t max_iters, uint32_t *poison, uint32_t *host_poison)
{
    constexpr int kDigits = BITS == 3 ? 8 : 256; // buckets that can receive anything
    __shared__ uint32_t wcount[kTailWaves][kDigits];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t s_range[2];
    __shared__ uint32_t s_flag;
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    if (t < 256) gbase[t] = cursor_cur[t];
    if (t == 0) {
        s_range[0] = range_in[0];
        s_range[1] = range_in[1];
    }
    for (int i = t; i < kTailWaves * kDigits; i += kTailBlock) (&wcount[0][0])[i] = 0;
    __syncthreads();
    // The entries a round appends to bucket c are the next round's input, in the order they were appended: the
    // threads that wrote them keep them (position, window) in registers, in place, so from the second round of a
    // launch on nothing is read back from memory (a round then costs its barriers, not two trips to L2).
    uint32_t val[kIndItems];
    WT wnd[kIndItems];
    bool live[kIndItems]; // entry k of this thread belongs to the current range (scan order: wave, k, lane)
    bool held = false;
    uint32_t prev_len = 0; // the range of the round before (for the jump)
    const uint32_t wave0 = (uint32_t)w * (kWave * kIndItems);
    for (uint32_t it = 0; it < max_iters; ++it) {
        const uint32_t lo = s_range[0], len = s_range[1] - lo;
        // (more than 8 buckets: a round of up to kTailMulti tiles is taken tile after tile -- the second round of a byte
        //  text's buckets, 8 - 16 thousand entries, was a chained launch of its own in front of this kernel: 10 us of
        //  the bucket's 70)
        if (len == 0 || len > (BITS == 3 ? kTailEntries : kTailMulti * kTailEntries)) break; // uniform
        const bool multi = len > kTailEntries; // uniform
        // ---- run jump -------------------------------------------------------------------
        // Inside a long run of symbol c every entry of the range induces its left neighbour
        // into bucket c again, round after round, in the same order.  If the L symbols to the
        // left of every entry are all c, the next L rounds are known: round j holds the same
        // entries minus j, in the next `len` slots.  They are written at once (L = 16 symbols
        // per checking thread; 4096 rounds a step for a single run) instead of one at a time.
        // Tried only when the last round kept every entry (the sign of runs): the check reads memory.
        if ((mode == MODE_L_FROM_L || mode == MODE_S_FROM_S) && len == prev_len && !multi) {
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
                __syncthreads();
                continue;
            }
        }
        prev_len = len;
        const uint32_t c_first = gbase[c]; // (where the round's appends to bucket c begin)
        for (uint32_t sub0 = 0; sub0 < len; sub0 += kTailEntries) { // uniform; one trip unless `multi`
        if (!held || multi) { // the range's entries from memory (the first round of a launch, after a jump, a round of several tiles)
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) { // (all loads issued before any is looked at)
                const uint32_t i = sub0 + wave0 + (uint32_t)k * kWave + lane;
                const uint32_t idx = lo + (i < len ? (rev ? len - 1u - i : i) : 0u);
                val[k] = SA[idx];
                wnd[k] = WN[idx];
            }
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) {
                const uint32_t i = sub0 + wave0 + (uint32_t)k * kWave + lane;
                live[k] = i < len;
                if (!live[k]) val[k] = 0, wnd[k] = 0;
            }
        }
        uint32_t dig[kIndItems], rnk[kIndItems];
        bool ok[kIndItems];
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            ok[k] = false;
            dig[k] = 0;
            if (live[k] && val[k] != 0) {
                const uint32_t ch = wnd_first<WT>(wnd[k], cfg);
                ok[k] = induce_accept(ch, c, mode);
                dig[k] = ch;
                val[k] -= 1u;
                wnd[k] = wnd_pop<WT>(wnd[k], cfg);
            }
        }
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) rnk[k] = wave_rank_step<BITS>(dig[k] & (uint32_t)(kDigits - 1), ok[k], wcount[w]);
        __syncthreads();
        uint32_t cnt = 0; // entries of this round for bucket t
        if (t < kDigits) {
#pragma unroll
            for (int ww = 0; ww < kTailWaves; ++ww) {
                const uint32_t x = wcount[ww][t];
                wcount[ww][t] = cnt;
                cnt += x;
            }
        }
        __syncthreads();
        {
            bool need[kIndItems];
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) need[k] = ok[k] && val[k] != 0 && wnd_count<WT>(wnd[k]) == 0;
            refill_windows<WT, kIndItems>(T, val, need, cfg, wnd);
        }
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            live[k] = ok[k] && dig[k] == c; // appended to bucket c itself: part of the next round
            if (ok[k]) {
                const uint32_t d = dig[k];
                const uint32_t r = wcount[w][d & (uint32_t)(kDigits - 1)] + rnk[k];
                const uint32_t dst = dir > 0 ? gbase[d] + r : gbase[d] - 1u - r;
                SA[dst] = val[k];
                WN[dst] = wnd[k];
                BW[dst] = wnd_symbol<WT>(wnd[k], cfg);
            }
        }
        held = !multi; // (the entries a round of several tiles appended lie with many threads' registers' worth each: from memory)
        __syncthreads();
        if (t < kDigits) {
            const uint32_t before = gbase[t];
            gbase[t] = dir > 0 ? before + cnt : before - cnt;
#pragma unroll
            for (int ww = 0; ww < kTailWaves; ++ww) wcount[ww][t] = 0;
        }
        __syncthreads();
        } // (tiles of the round)
        if ((uint32_t)t == c) { // what the round appended to bucket c is the next round's input
            const uint32_t now = gbase[c];
            s_range[0] = dir > 0 ? c_first : now;
            s_range[1] = dir > 0 ? now : c_first;
        }
        __syncthreads();
    }
    if (t < 256) cursor_nxt[t] = gbase[t];
    if (t == 0) {
        range_out[0] = s_range[0];
        range_out[1] = s_range[1];
        tail_report(s_range[0], s_range[1], c, poison, host_poison);
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
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    if (t < 8) gbase[t] = cursor_cur[t];
    if (t == 0) {
        s_range[0] = range_in[0];
        s_range[1] = range_in[1];
    }
    __syncthreads();
    const uint32_t B = cfg.B, cmask = cfg.mask;
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
                ++it;
                __syncthreads();
                continue;
            }
        }
        prev_len = len;
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
constexpr uint32_t kBatchFrom = 1u << 21;    // rounds expected to hold more entries than this are launches of their own

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

// ---- very long runs -------------------------------------------------------------------------------------
// The tail kernel's run jump writes 4096 rounds a step with one workgroup: 11 us a step, 45 ms for the 16 Mi
// symbols of a gap in a reference assembly (runs of N of up to 30 Mbp, one or more per chromosome, all in one
// bucket).  When a bucket's range is down to a handful of entries and the tail kernel has not finished them, the
// whole device takes over: `run_probe` finds how many symbols c lie immediately to the left of every entry (L, the
// minimum, looking kRunProbe symbols far), `run_fill` writes the L rounds (entries - 1, entries - 2, ...) into the
// next L x len slots of bucket c (each with its window and symbol byte), `run_commit` advances the cursor and
// leaves the last round as the range.  L = 0 changes nothing.
constexpr uint32_t kRunProbe = 1u << 26;
constexpr uint32_t kRunEntries = 64; // runs of c that are alive in the bucket at the same time (a gap per chromosome)
constexpr uint32_t kRunProbeChunk = (uint32_t)kBlock * 16u; // symbols a workgroup looks at per step
constexpr uint32_t kRunProbeGrid = 256;                    // workgroups per entry
// Workgroup x of entry y looks at the distances [k * chunk, (k + 1) * chunk), k = x, x + grid, ..., and stops as soon
// as the run is known to end nearer than where it would look next: the probe costs what the run is long, not the
// 64 Mi symbols it may look at most (with every workgroup reading its piece whatever the others found, a probe of 64
// entries read 4 GB: 5.5 ms, 14 probes in a genome-like 1 GiB text whose runs are a few dozen symbols long).
__global__ __launch_bounds__(kBlock) void run_probe_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ SA,
                                                          const uint32_t *__restrict__ range, uint32_t c, uint32_t from,
                                                          uint32_t look, uint32_t *__restrict__ run_len /* preset to ~0 */)
{
    const uint32_t lo = range[0], len = range[1] - lo;
    if (len == 0 || len > kRunEntries || blockIdx.y >= len) { // (uniform) not a handful of entries: nothing to jump over
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (len == 0 || len > kRunEntries)) atomicMin(run_len, 0u);
        return;
    }
    const uint32_t p = SA[lo + blockIdx.y]; // (the minimum over the entries does not depend on their order)
    for (uint32_t chunk0 = from + blockIdx.x * kRunProbeChunk; chunk0 < look; chunk0 += gridDim.x * kRunProbeChunk) { // uniform
        // (a relaxed agent-scope load: what another workgroup found becomes visible in time, never too early)
        if (__hip_atomic_load(run_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= chunk0) return;
        const uint32_t d0 = chunk0 + threadIdx.x * 16u; // this thread looks at p-d0-1 ... p-d0-16
        if (d0 >= p) {
            if (d0 < p + 16u) atomicMin(run_len, p); // the text starts here: at most p symbols to the left
            continue;
        }
        const uint32_t cnt = p - d0 < 16u ? p - d0 : 16u;
        uint32_t first_other = cnt; // symbols c in a row, going left from p - d0
        for (uint32_t e = 0; e < cnt; ++e)
            if (T[p - d0 - 1u - e] != (uint8_t)c) {
                first_other = e;
                break;
            }
        if (first_other < 16u) atomicMin(run_len, d0 + first_other); // (cnt < 16: the text starts there)
    }
}

// rounds the jump covers: every entry of the range has at least that many symbols c to its left
__device__ __forceinline__ uint32_t run_length(const uint32_t *run_len, uint32_t len)
{
    if (len == 0 || len > kRunEntries) return 0;
    const uint32_t L = *run_len, most = kRunProbe / len; // (at most kRunProbe entries a jump: fits 32-bit offsets)
    return L > most ? most : L;
}

template <class WT>
__global__ __launch_bounds__(kBlock) void run_fill_kernel(const uint8_t *__restrict__ T, uint32_t *SA, WT *__restrict__ WN,
                                                         uint8_t *__restrict__ BW, const uint32_t *__restrict__ range,
                                                         const uint32_t *__restrict__ cursor, uint32_t c, int rev, int dir,
                                                         wnd_cfg cfg, const uint32_t *__restrict__ run_len)
{
    const uint32_t lo = range[0], len = range[1] - lo;
    const uint64_t total = (uint64_t)run_length(run_len, len) * len;
    const uint32_t cur = cursor[c];
    // round j holds the range's entries minus j, in the same order, in the next len slots (as the tail kernel's jump)
    for (uint64_t o = (uint64_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (uint64_t)gridDim.x * kBlock) {
        const uint32_t j = (uint32_t)(o / len) + 1u, i = (uint32_t)(o % len);
        const uint32_t v = SA[lo + (rev ? len - 1u - i : i)] - j;
        const uint32_t dst = dir > 0 ? cur + (uint32_t)o : cur - 1u - (uint32_t)o;
        const WT nw = v ? wnd_fill<WT>(T, v, cfg) : (WT)0;
        SA[dst] = v;
        WN[dst] = nw;
        BW[dst] = wnd_symbol<WT>(nw, cfg);
    }
}

__global__ void run_commit_kernel(uint32_t *__restrict__ range, uint32_t *__restrict__ cursor, uint32_t c, int dir,
                                  const uint32_t *__restrict__ run_len)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t lo = range[0], len = range[1] - lo;
    const uint32_t total = run_length(run_len, len) * len;
    if (total == 0) return;
    const uint32_t cur = cursor[c];
    cursor[c] = dir > 0 ? cur + total : cur - total;
    range[0] = dir > 0 ? cur + total - len : cur - total; // the last round written: the next round's input
    range[1] = range[0] + len;
}

// range <- [lo, hi) given by the host, or [a, cursor[c]) / [cursor[c], b) for the first round of a bucket
// (and the tickets of the chained launches that follow are zeroed: one launch instead of a memset and a launch)
// (poison: a bucket earlier in this unattended pass did not come to its end -- see induce_typed --: the range is left
//  empty, and every launch over an empty range only carries the cursors on)
__global__ void set_range_kernel(uint32_t *range, uint32_t lo, uint32_t hi, const uint32_t *cursor, int c, int which,
                                 uint32_t *tickets, uint32_t ntickets, const uint32_t *poison)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (uint32_t i = 0; i < ntickets; ++i) tickets[i] = 0;
        if (which == 1) hi = cursor[c];      // L pass: [bucket begin, head cursor)
        else if (which == 2) lo = cursor[c]; // S pass: [tail cursor, bucket end)
        if (poison && poison[0]) lo = hi = 0;
        range[0] = lo;
        range[1] = hi;
    }
}

// the tail kernel's last word in an unattended pass: a range it could not finish (runs longer than its steps, or more
// entries than it holds) is recorded once -- bucket and range, on the device and in the host's pinned page -- and
// stops the rest of the pass (set_range_kernel)
__device__ __forceinline__ void tail_report(uint32_t lo, uint32_t hi, uint32_t c, uint32_t *poison, uint32_t *host_poison)
{
    if (!poison || lo == hi || poison[0]) return;
    poison[1] = c, poison[2] = lo, poison[3] = hi;
    poison[0] = 1;
    __hip_atomic_store(host_poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bwt[i] = text[SA[i]-1]: the first symbol of slot i's window; the one slot whose entry is
// position 0 has an empty window (count 0) and gets the sentinel (bwt.c:13-20)
template <class WT>
__global__ __launch_bounds__(kBlock) void bwt_from_windows_kernel(const WT *__restrict__ WN, uint64_t N, wnd_cfg cfg,
                                                                  uint8_t *__restrict__ bwt)
{
    // 16 slots per thread: 16-byte loads of the windows, one 16-byte store of the symbols
    const uint64_t i0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 16u;
    if (i0 >= N) return;
    if (i0 + 16u <= N && (((uintptr_t)WN | (uintptr_t)bwt) & 15u) == 0) {
        uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            WT w[4];
            load_quad(WN + i0 + 4 * q, w);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t b = wnd_count<WT>(w[e]) == 0 ? 0u : wnd_first<WT>(w[e], cfg);
                out[q] |= (b & 0xFFu) << (8 * e);
            }
        }
        uint4 v;
        v.x = out[0], v.y = out[1], v.z = out[2], v.w = out[3];
        *reinterpret_cast<uint4 *>(bwt + i0) = v;
    } else {
        for (uint64_t i = i0; i < N && i < i0 + 16u; ++i) {
            const WT w = WN[i];
            bwt[i] = wnd_count<WT>(w) == 0 ? (uint8_t)0 : (uint8_t)wnd_first<WT>(w, cfg);
        }
    }
}

// the sort's 32-bit seed windows (fewer symbols, same layout) as the 64-bit words the passes of a wide alphabet read
__global__ __launch_bounds__(kBlock) void widen_windows_kernel(const uint32_t *__restrict__ in, uint64_t count, uint64_t *__restrict__ out)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 4u;
    if (i0 >= count) return;
    if (i0 + 4u <= count && (((uintptr_t)in | (uintptr_t)out) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + i0);
        uint4 a, b;
        a.x = v.x, a.y = 0, a.z = v.y, a.w = 0;
        b.x = v.z, b.y = 0, b.z = v.w, b.w = 0;
        *reinterpret_cast<uint4 *>(out + i0) = a;
        *reinterpret_cast<uint4 *>(out + i0 + 2) = b;
    } else {
        for (uint64_t i = i0; i < count && i < i0 + 4u; ++i) out[i] = in[i];
    }
}

template <class WT>
__global__ void set_entry_kernel(uint32_t *SA, WT *WN, uint8_t *BW, uint32_t p, const uint8_t *T, wnd_cfg cfg)
{
    const WT w = p ? wnd_fill<WT>(T, p, cfg) : (WT)0;
    SA[0] = p;
    WN[0] = w;
    BW[0] = wnd_symbol<WT>(w, cfg);
}

} // namespace sx

using namespace sx;

size_t sx_induce_scratch_bytes(uint64_t N, uint32_t sigma)
{
    // windows for every SA slot (8 bytes worst case) + seed windows (N/2) + symbol bytes + control block
    const uint64_t ntiles = (N + kIndTile - 1) / kIndTile + 1;
    const uint64_t wtiles = N / 4096 + 4; // wide alphabets: [tile][256] counts of (at least) 4096-entry tiles + chunk sums
    // (at most 8 buckets: (round, bucket) count rows of the eight-rounds-at-a-time form, over the largest bucket's tiles)
    return (size_t)N * 8 + 256 + (size_t)(N / 2 + 2) * 8 + 256 + (size_t)N + 256 + (size_t)sigma * ntiles * 4 + 256 +
           (size_t)kBatchRows * (ntiles + 1) * 4 + 1024 +
           (sigma > 8 ? (size_t)(wtiles + wtiles / 256 + 4) * 1024 + 512 : 0) + 16384 +
           // the up-front rounds of more than 8 buckets: tile counts of all buckets' regions, the bigram matrix and its tables
           (sigma > 8 ? ((size_t)N / kWideTile + 520) * 1024 + 5 * 65536 * 4 + 16384 : 0);
}

namespace {
#ifndef SX_INDUCE_GRID_CAP
#define SX_INDUCE_GRID_CAP 16384
#endif
// workgroups of a round's launches (they loop over the round's tiles).  1 GiB DNA, induce_scatter per step: 4096
// workgroups 5.55 ms, 16384: 4.99, 65536: 5.06, 262144: 5.12 (a device copy is fastest with many short workgroups too).
constexpr uint32_t kInduceGridCap = SX_INDUCE_GRID_CAP;
constexpr int kMaxSpec = 16; // rounds queued per batch (then the tail kernel) before the host looks at the range

template <class WT> struct induce_state {
    sx_ctx *ctx;
    const uint8_t *T;
    uint32_t *SA;
    WT *WN;
    uint64_t N, m; // entries of (SA, WN, BW) and of the seed arrays
    uint8_t *BW; // text[SA[i] - 1] of every written slot (0 for position 0): what the counting launches read; the BWT in the end
    uint32_t *cursor[2]; // ping-pong: a round reads one, its last tile writes the other
    uint32_t *ranges;    // (kMaxSpec + 2) x {lo, hi}
    uint32_t *tickets;   // kMaxSpec + 2
    uint32_t *run_len;   // symbols a device-wide run jump covers
    uint64_t *status;
    uint32_t chain_max; // rounds up to this many entries take the chained launch
    uint32_t *hist;   // [nk][stride] tile counts of the three-launch form (at most 8 buckets)
    uint32_t *bhist;  // [round * 8 + bucket][stride] tile counts of the eight-rounds-at-a-time form
    uint32_t *btotals; // kBatchRows row totals
    int batch_on;
    int unattended; // the buckets are queued one behind the other without a look at a bucket's last range (see induce_typed)
    uint32_t *poison;      // device: {set, bucket, lo, hi} of the first bucket an unattended pass could not finish
    uint32_t *host_poison; // the same flag in the host's pinned page (the host looks at it between buckets, without a wait)
    uint32_t *whist;  // [tile][256] the same for wide alphabets, tiles of 8192 entries
    uint32_t *wsums;  // [chunk][256] column sums of chunks of 256 tiles
    uint32_t stride;
    uint32_t nk;
    int small_alphabet;
    int par; // which cursor buffer is current
    wnd_cfg cfg;
    // more than 8 buckets: the other-region rounds of all buckets are done up front (hoist_*_kernel, bucket_begin_kernel)
    int hoist;
    uint32_t *d_begin;       // bucket boundaries on the device (257)
    uint32_t *hoist_E;       // the current pass's group ends: EL or ES, [c][d]
    uint32_t *hoist_tot;     // the current pass's up-front entries from bucket c to bucket d, [c][d]
    uint32_t *hoist_err;     // set by bucket_begin_kernel when a cursor is not where the bigram counts put it
    uint32_t hoist_from;     // the last bucket of the pass that had rounds of its own (L pass: 0, S pass: nk - 1 before the first)
};

template <class WT>
void launch_round(induce_state<WT> &st, const uint32_t *srcP, const WT *srcW, int range_slot, int out_slot,
                  uint32_t tiles_bound, uint32_t tiles_likely, int rev, int mode, uint32_t c, int dir, int tail_follows,
                  int chained_only_up_to_chain_max = 0, int three_launch_only = 0)
{
    sx_ctx *ctx = st.ctx;
    uint32_t grid = tiles_bound < 1 ? 1 : tiles_bound;
    if (grid > kInduceGridCap) grid = kInduceGridCap; // (every kernel loops over its tiles: any grid size is correct)
    const uint32_t epoch = sx_chain_next_epoch(ctx);
    uint32_t *rin = st.ranges + 2 * range_slot;
    uint32_t *rout = out_slot >= 0 ? st.ranges + 2 * out_slot : nullptr;
    const uint32_t *cur = st.cursor[st.par];
    uint32_t *nxt = st.cursor[st.par ^ 1];
    // tiles_likely: what the round is expected to need (decides which forms are queued);
    // a round that turns out longer is still handled, by the chained form alone if need be.
    // three_launch_only: the round is large for sure (its size is known, or expected beyond doubt), no chained launch is
    // queued behind the three (they take a range of any length then, an empty one is carried on by the offsets launch)
    const bool only3 = three_launch_only != 0;
    const bool both = only3 || (!chained_only_up_to_chain_max && (uint64_t)tiles_likely * kIndTile > st.chain_max);
    const uint32_t chain_max = only3 ? 0u : ((both || chained_only_up_to_chain_max) ? st.chain_max : ~0u);
    const int pass_large = chained_only_up_to_chain_max;
    if (both && st.small_alphabet) {
        // the round may be a large one: queue the three-launch form as well
        // (entries of the suffix array have their symbol bytes next to them; the LMS seeds only their windows)
        const uint8_t *srcB = srcP == st.SA ? (const uint8_t *)st.BW : nullptr;
        const uint64_t src_len = srcP == st.SA ? st.N : st.m;
        if (srcB)
            sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, induce_count_bytes_kernel, dim3(grid), dim3(kBlock), srcB, (const uint32_t *)rin, rev,
                      mode, c, st.hist, st.stride, st.nk, chain_max, src_len);
        else
            sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, induce_count_kernel<WT, 3>, dim3(grid), dim3(kBlock), srcW, srcB,
                      (const uint32_t *)rin, rev, mode, c, st.cfg, st.hist, st.stride, st.nk, chain_max, src_len);
        sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)tiles_bound * st.nk * 8, induce_offsets_kernel, dim3(st.nk),
                  dim3(kRowThreads), st.hist, st.stride, (const uint32_t *)rin, rout, cur, nxt, dir, c, chain_max, only3 ? 1 : 0);
        {
#define SX_SCATTER_SMALL(M)                                                                                            \
    sx_launch(ctx, SX_KC_INDUCE_SCATTER, 0, induce_scatter_small_kernel<WT, M>, dim3(grid), dim3(kBlock), srcP, srcW, \
              (const uint32_t *)rin, c, st.cfg, st.T, (const uint32_t *)st.hist, st.stride, cur, st.SA, st.WN, st.BW, st.nk, \
              chain_max)
            switch (mode) { // mode fixes the scan direction (rev) and the side the buckets grow to (dir)
            case MODE_L_FROM_L: SX_SCATTER_SMALL(MODE_L_FROM_L); break;
            case MODE_L_FROM_LMS: SX_SCATTER_SMALL(MODE_L_FROM_LMS); break;
            case MODE_S_FROM_S: SX_SCATTER_SMALL(MODE_S_FROM_S); break;
            default: SX_SCATTER_SMALL(MODE_S_FROM_L); break;
            }
#undef SX_SCATTER_SMALL
        }
    }
    if (both && !st.small_alphabet) {
        // wide alphabets: the round as a radix pass over tiles of 8192 entries (count, offsets, scatter)
        const uint8_t *srcB = srcP == st.SA ? (const uint8_t *)st.BW : nullptr;
        const uint32_t wtiles = sx_div_up((uint64_t)(tiles_bound < 1 ? 1 : tiles_bound) * kIndTile, kWideTile);
        const uint32_t wgrid = wtiles > 2048 ? 2048 : wtiles;
        const int only = only3 ? 1 : 0;
        sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, induce_wide_count_kernel<WT>, dim3(wgrid), dim3(kWideThreads), srcW, srcB,
                  (const uint32_t *)rin, rev, mode, c, st.cfg, st.whist, chain_max);
        if (wtiles <= kWideOffMaxTiles) {
            sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)wtiles * 2048, induce_wide_offsets_kernel, dim3(256 / kWideOffCols), dim3(kWideOffThreads),
                      st.whist, (const uint32_t *)rin, rout, cur, nxt, dir, c, chain_max, only);
        } else {
            const uint32_t nchunks = sx_div_up(wtiles, kWideChunk);
            sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)wtiles * 1024, induce_wide_colsum_kernel, dim3(nchunks), dim3(kBlock),
                      (const uint32_t *)st.whist, (const uint32_t *)rin, st.wsums, chain_max);
            sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)nchunks * 2048, induce_wide_bases_kernel, dim3(1), dim3(kBlock), st.wsums,
                      (const uint32_t *)rin, rout, cur, nxt, dir, c, chain_max, only);
            sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)wtiles * 2048, induce_wide_apply_kernel, dim3(nchunks), dim3(kBlock), st.whist,
                      (const uint32_t *)rin, (const uint32_t *)st.wsums, chain_max);
        }
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, 0, induce_wide_scatter_kernel<WT, 8>, dim3(wgrid), dim3(kWideThreads),
                  srcP, srcW, (const uint32_t *)rin, rev, mode, c, st.cfg, st.T, (const uint32_t *)st.whist, cur, dir, st.SA, st.WN,
                  st.BW, chain_max);
    }
    uint32_t cgrid = grid > 1024 ? 1024 : grid;
    if (only3) {
    } else if (st.small_alphabet)
        sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, induce_round_kernel<WT, 3>, dim3(cgrid), dim3(kBlock), srcP, srcW,
                  (const uint32_t *)rin, rout, rev, mode, c, st.cfg, st.T, cur, nxt, dir, st.SA, st.WN, st.BW, st.nk, st.status,
                  epoch, st.tickets + range_slot, chain_max, tail_follows, pass_large);
    else if (st.nk <= 32)
        sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, induce_round_kernel<WT, 5>, dim3(cgrid), dim3(kBlock), srcP, srcW,
                  (const uint32_t *)rin, rout, rev, mode, c, st.cfg, st.T, cur, nxt, dir, st.SA, st.WN, st.BW, st.nk, st.status,
                  epoch, st.tickets + range_slot, chain_max, tail_follows, pass_large);
    else
        sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, induce_round_kernel<WT, 8>, dim3(cgrid), dim3(kBlock), srcP, srcW,
                  (const uint32_t *)rin, rout, rev, mode, c, st.cfg, st.T, cur, nxt, dir, st.SA, st.WN, st.BW, st.nk, st.status,
                  epoch, st.tickets + range_slot, chain_max, tail_follows, pass_large);
    st.par ^= 1;
    ctx->stats.induce_rounds++;
}

// steps of the tail kernel per launch: a run that outlasts them goes to the device-wide jump (run_fill).  In a pass
// queued as a whole nobody is there to start that jump, so the tail kernel gets more steps -- but not the 16 384 that
// would see any run the classification did not report (shorter than two tiles, 8191 symbols) to its end: runs of
// differing lengths just under that never meet the jump's condition (every entry of the round continued), and one
// workgroup then ground through thousands of dependent steps of a few microseconds each while the chip idled.  A
// bucket that outlasts these steps is reported (tail_report) and carried on attended, with the device-wide jump.
constexpr uint32_t kTailIters = 64, kTailItersUnattended = 1024;
template <class WT>
void launch_tail(induce_state<WT> &st, int range_slot, int out_slot, int rev, int mode, uint32_t c, int dir)
{
    sx_ctx *ctx = st.ctx;
    const uint32_t *cur = st.cursor[st.par];
    uint32_t *nxt = st.cursor[st.par ^ 1];
    if (st.small_alphabet)
        sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, induce_tail_small_kernel<WT>, dim3(1), dim3(kTailBlock), st.SA, st.WN, st.BW,
                  (const uint32_t *)(st.ranges + 2 * range_slot), st.ranges + 2 * out_slot, rev, mode, c, st.cfg, st.T,
                  cur, nxt, dir, st.unattended ? kTailItersUnattended : kTailIters, st.unattended ? st.poison : (uint32_t *)nullptr,
                  st.host_poison);
    else
        sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, induce_tail_kernel<WT, 8>, dim3(1), dim3(kTailBlock), st.SA, st.WN, st.BW,
                  (const uint32_t *)(st.ranges + 2 * range_slot), st.ranges + 2 * out_slot, rev, mode, c, st.cfg, st.T,
                  cur, nxt, dir, st.unattended ? kTailItersUnattended : kTailIters, st.unattended ? st.poison : (uint32_t *)nullptr,
                  st.host_poison);
    st.par ^= 1;
}

// kBatchRounds self rounds of bucket c in three launches (at most 8 buckets; ranges the tail kernel can take pass through)
template <class WT>
void launch_batch(induce_state<WT> &st, int range_slot, int out_slot, uint32_t tiles_bound, int mode, uint32_t c)
{
    sx_ctx *ctx = st.ctx;
    uint32_t grid = tiles_bound < 1 ? 1 : tiles_bound;
    if (grid > kInduceGridCap) grid = kInduceGridCap;
    const uint32_t *rin = st.ranges + 2 * range_slot;
    uint32_t *rout = st.ranges + 2 * out_slot;
    const uint32_t *cur = st.cursor[st.par];
    uint32_t *nxt = st.cursor[st.par ^ 1];
    const uint32_t min_len = ctx->induce_batch_min >= 0 ? (uint32_t)ctx->induce_batch_min : kTailEntries;
    if (mode == MODE_L_FROM_L) {
        sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, induce_batch_count_kernel<WT, MODE_L_FROM_L>, dim3(grid), dim3(kBlock),
                  (const uint32_t *)st.SA, (const WT *)st.WN, rin, c, st.cfg, st.T, st.bhist, st.stride, min_len);
    } else {
        sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, induce_batch_count_kernel<WT, MODE_S_FROM_S>, dim3(grid), dim3(kBlock),
                  (const uint32_t *)st.SA, (const WT *)st.WN, rin, c, st.cfg, st.T, st.bhist, st.stride, min_len);
    }
    sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)tiles_bound * st.nk * kBatchRounds * 8, induce_batch_offsets_kernel,
              dim3(kBatchRounds * st.nk), dim3(kRowThreads), st.bhist, st.stride, rin, st.btotals, st.nk, min_len);
    if (mode == MODE_L_FROM_L) {
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, 0, induce_batch_scatter_kernel<WT, MODE_L_FROM_L>, dim3(grid), dim3(kBlock),
                  (const uint32_t *)st.SA, (const WT *)st.WN, rin, rout, c, st.cfg, st.T, (const uint32_t *)st.bhist, st.stride,
                  (const uint32_t *)st.btotals, cur, nxt, st.SA, st.WN, st.BW, st.nk, min_len);
    } else {
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, 0, induce_batch_scatter_kernel<WT, MODE_S_FROM_S>, dim3(grid), dim3(kBlock),
                  (const uint32_t *)st.SA, (const WT *)st.WN, rin, rout, c, st.cfg, st.T, (const uint32_t *)st.bhist, st.stride,
                  (const uint32_t *)st.btotals, cur, nxt, st.SA, st.WN, st.BW, st.nk, min_len);
    }
    st.par ^= 1;
    ctx->stats.induce_rounds++;
}

// all rounds of one region of bucket c: the first range comes from the cursor, every
// round appends to bucket c what the next round reads; batches of queued rounds, one
// host look per batch
template <class WT>
int run_self_rounds(induce_state<WT> &st, uint32_t fixed_bound, uint32_t region_entries, int rev, int mode, uint32_t c,
                    int dir, int which, uint32_t *total_in_region, double share /* of symbol c in the text */,
                    const uint32_t *resume = nullptr /* {lo, hi}: the range an unattended pass left of this region: carry on from it */)
{
    sx_ctx *ctx = st.ctx;
    bool first = true;
    uint32_t bound_tiles = sx_div_up(region_entries ? region_entries : 1, kIndTile);
    // queued rounds per batch: until the expected round size (a run of c continues with
    // probability ~1/#symbols) is down to one tile; the tail kernel takes it from there
    int spec = 1;
    {
        const int sh = st.small_alphabet ? 2 : 6;
        while (spec < kMaxSpec && (bound_tiles >> (sh * spec)) >= 1) ++spec;
    }
    if (!st.small_alphabet) {
        // by the symbol's share of the text: rounds are queued until the one handed to the tail kernel is expected to hold
        // an eighth of what the kernel takes (a pass queued as a whole has nobody to queue one more: 1 GiB of 20 symbols,
        // round 3 of a bucket expected at 5400 entries, beyond 8192 in two buckets -- both passes ran twice).  Round 4: the
        // tail kernel takes rounds of up to kTailMulti tiles (32 768 entries), so an unattended bucket's rounds are queued
        // by the share alone (a byte text: one round, then the tail kernel; the chained launch in between is gone) -- a
        // bucket whose runs make the rounds shrink more slowly than its share says leaves word and is carried on attended,
        // with the longer queue.
        double expect = (double)region_entries * share;
        int by_share = 1;
        // (unattended: half of what the tail kernel takes -- a round's size is a sum of independent draws, and one that is
        //  too long after all is carried on attended; attended: an eighth, nobody queues one more)
        const double tail_takes = (double)(kTailMulti * kTailEntries) / ((st.unattended && !resume) ? 2.0 : 8.0);
        while (by_share < kMaxSpec && expect > tail_takes) ++by_share, expect *= share;
        if (st.unattended && !resume) spec = by_share;
        else if (by_share > spec) spec = by_share;
    }
    // Every batch ends with the tail kernel, which runs kTailIters rounds unless the range empties first, and a round
    // consumes one symbol of every run it follows: a bucket cannot need more batches than this (a device fault that
    // keeps the range alive must not keep the host here for ever).
    const uint64_t max_batches = 2 * (st.N / kTailIters) + 64; // (rounds too long for the tail consume > 8192 symbols each)
    uint32_t r[2] = {0, 0};
    bool resuming = resume != nullptr;
    if (resuming) r[0] = resume[0], r[1] = resume[1];
    for (uint64_t batch = 0;; ++batch) {
        if (batch > max_batches) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: the rounds of a bucket did not come to an end");
        if (!resuming) {
        if (first && st.hoist) {
            sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, bucket_begin_kernel, dim3(1), dim3(256), st.ranges, st.cursor[st.par],
                      (const uint32_t *)st.d_begin, (const uint32_t *)st.hoist_E, (const uint32_t *)st.hoist_tot, st.nk, c, st.hoist_from,
                      dir, st.tickets, (uint32_t)(kMaxSpec + 2), (const uint32_t *)(st.unattended ? st.poison : nullptr), st.hoist_err);
            st.hoist_from = c;
        } else if (first)
            sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, set_range_kernel, dim3(1), dim3(1), st.ranges, fixed_bound, fixed_bound,
                      (const uint32_t *)st.cursor[st.par], (int)c, which, st.tickets, (uint32_t)(kMaxSpec + 2),
                      (const uint32_t *)(st.unattended ? st.poison : nullptr));
        else
            SX_CHECK(hipMemsetAsync(st.tickets, 0, (kMaxSpec + 2) * sizeof(uint32_t), ctx->stream));
        const bool batched = st.small_alphabet && st.batch_on && (mode == MODE_L_FROM_L || mode == MODE_S_FROM_S);
        if (batched) {
            // First the rounds expected to be large, a launch each -- round k of a bucket holds about share^k of its region
            // (three launches, and no chained one behind them where the expectation is beyond doubt) --, then eight rounds by
            // one count / scan / scatter (the rounds that moved next to nothing and cost a launch chain each), the tail
            // kernel for what eight rounds leave of a range of a million.  A range the tail cannot hold (runs longer than the
            // rounds taken: poly-A, microsatellites) comes round again: eight more rounds, some chained ones, the tail.
            int slot = 0;
            if (first) {
                double expect = (double)region_entries;
                // (SX_FLAG_INDUCE_BATCH_MIN, tests: that bound here too, and every such round in the three-launch-only form)
                const double batch_from = ctx->induce_batch_min >= 0 ? (double)ctx->induce_batch_min : (double)kBatchFrom;
                for (int k = 0; k < 6 && expect > batch_from; ++k, expect *= share) {
                    uint32_t tb = sx_div_up((uint64_t)(expect * 2.0 < (double)region_entries ? expect * 2.0 : (double)region_entries), kIndTile);
                    if (tb < 256) tb = bound_tiles < 256 ? bound_tiles : 256;
                    const bool sure = ctx->induce_batch_min >= 0 || expect > 4.0 * (double)st.chain_max;
                    launch_round<WT>(st, st.SA, st.WN, slot, slot + 1, tb, tb, rev, mode, c, dir, 1, 0, sure ? 1 : 0);
                    ++slot;
                }
            }
            launch_batch<WT>(st, slot, slot + 1, bound_tiles, mode, c);
            ++slot;
            if (!first) {
                for (int k = 0; k < 3; ++k, ++slot)
                    launch_round<WT>(st, st.SA, st.WN, slot, slot + 1, bound_tiles < 256 ? bound_tiles : 256, 0, rev, mode, c, dir, 1, 1);
            }
            spec = slot;
        }
        for (int k = 0; k < spec && !batched; ++k) {
            uint32_t tb = bound_tiles >> k;
            const uint32_t floor_tiles = bound_tiles < 256 ? bound_tiles : 256;
            if (tb < floor_tiles) tb = floor_tiles;
            // a run of c's continues with the probability of c: expect round k to hold share^k of the region
            // (twice that, to be on the safe side, decides whether the three-launch form is queued as well: a launch
            // that finds nothing to do still costs 5 us, and there were 20 of them per bucket)
            double expect = 2.0 * (double)bound_tiles;
            for (int i = 0; i < k; ++i) expect *= share;
            const uint32_t likely = expect < (double)bound_tiles ? (uint32_t)expect : bound_tiles;
            // (the first round of a large region is large beyond doubt: the three-launch form alone)
            const bool sure = first && k == 0 && (uint64_t)region_entries > 16ull * st.chain_max;
            launch_round<WT>(st, st.SA, st.WN, k, k + 1, tb, first && k == 0 ? bound_tiles : likely, rev, mode, c, dir, 1, 0, sure ? 1 : 0);
        }
        launch_tail<WT>(st, spec, spec + 1, rev, mode, c, dir);
        if (st.unattended) { // the tail kernel ends nearly every bucket; one that it does not leaves word (tail_report)
            if (total_in_region) *total_in_region = 0xFFFFFFFFu;
            return 0;
        }
        SX_TRY(sx_readback(ctx, st.ranges + 2 * (spec + 1), 2, r));
        if (r[1] == r[0]) {
            if (total_in_region) *total_in_region = dir > 0 ? r[1] : r[0];
            return 0;
        }
        } // (!resuming)
        resuming = false;
        // a long run of symbol c: carry on from the last range
        sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, set_range_kernel, dim3(1), dim3(1), st.ranges, r[0], r[1],
                  (const uint32_t *)st.cursor[st.par], (int)c, 0, (uint32_t *)nullptr, 0u, (const uint32_t *)nullptr);
        if (r[1] - r[0] <= kRunEntries) {
            // a handful of entries deep inside runs: the device-wide jump, twice (a run may be longer than one probe looks)
            for (int rep = 0; rep < 2; ++rep) {
                SX_CHECK(hipMemsetAsync(st.run_len, 0xFF, sizeof(uint32_t), ctx->stream));
                const uint64_t look = st.N < (uint64_t)kRunProbe ? st.N : (uint64_t)kRunProbe; // (no run is longer than the text)
                // the nearest 4096 symbols first, by one workgroup per entry: most runs end there, and the workgroups of the
                // far probe then leave at their first look (started together they would all read their first piece)
                sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, run_probe_kernel, dim3(1, r[1] - r[0]), dim3(kBlock), st.T,
                          (const uint32_t *)st.SA, (const uint32_t *)st.ranges, c, 0u,
                          (uint32_t)(look < kRunProbeChunk ? look : kRunProbeChunk), st.run_len);
                if (look > kRunProbeChunk) {
                    const uint32_t far_chunks = sx_div_up(look - kRunProbeChunk, kRunProbeChunk);
                    sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, run_probe_kernel,
                              dim3(far_chunks < kRunProbeGrid ? far_chunks : kRunProbeGrid, r[1] - r[0]), dim3(kBlock), st.T,
                              (const uint32_t *)st.SA, (const uint32_t *)st.ranges, c, kRunProbeChunk, (uint32_t)look, st.run_len);
                }
                sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, run_fill_kernel<WT>, dim3(4096), dim3(kBlock), st.T, st.SA, st.WN, st.BW,
                          (const uint32_t *)st.ranges, (const uint32_t *)st.cursor[st.par], c, rev, dir, st.cfg,
                          (const uint32_t *)st.run_len);
                sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, run_commit_kernel, dim3(1), dim3(1), st.ranges, st.cursor[st.par], c, dir,
                          (const uint32_t *)st.run_len);
            }
        }
        first = false;
    }
}

template <class WT>
int induce_typed(sx_ctx *ctx, const sx_text_info &ti, uint32_t sigma, const uint32_t *sorted_lms,
                 const void *seed_windows, bool seed_windows_u32, uint32_t *SA, uint8_t *bwt_out, sx_arena &arena, wnd_cfg cfg)
{
    const uint64_t N = ti.N;
    // buckets that hold anything: 0 .. maxc
    const uint32_t nk = ti.maxc + 1 < sigma ? ti.maxc + 1 : sigma;
    induce_state<WT> st;
    st.ctx = ctx;
    st.T = ti.T;
    st.SA = SA;
    st.nk = nk;
    st.small_alphabet = nk <= 8;
    st.cfg = cfg;
    st.par = 0;
    st.WN = arena.take<WT>(N);
    st.BW = bwt_out ? bwt_out : arena.take<uint8_t>(N);
    st.N = N;
    st.m = ti.m;
    // (seed_windows_u32: the prefix-key sort's 32-bit words for a text whose windows are 64-bit: widened below)
    const bool widen = seed_windows && seed_windows_u32 && sizeof(WT) == 8;
    WT *seedW = seed_windows && !widen ? (WT *)seed_windows : arena.take<WT>(ti.m ? ti.m : 1);
    st.cursor[0] = arena.take<uint32_t>(256);
    st.cursor[1] = arena.take<uint32_t>(256);
    st.ranges = arena.take<uint32_t>(2 * (kMaxSpec + 3));
    st.tickets = arena.take<uint32_t>(kMaxSpec + 3);
    st.run_len = arena.take<uint32_t>(4);
    if (!st.WN || !st.BW || !seedW || !st.cursor[0] || !st.cursor[1] || !st.ranges || !st.tickets || !st.run_len)
        return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small");

    // bucket boundaries on the host (sa_is.c:176-201)
    uint32_t begin[257], lms_off[257];
    begin[0] = 0;
    lms_off[0] = 0;
    uint32_t largest = 1;
    for (uint32_t c = 0; c < 256; ++c) {
        begin[c + 1] = begin[c] + ti.h_all[c];
        lms_off[c + 1] = lms_off[c] + ti.h_lms[c];
        if (ti.h_all[c] > largest) largest = ti.h_all[c];
    }
    // look-back status words: one per (tile, bucket) of the largest round
    // The look-back walk costs a few microseconds per tile, so long rounds are better off with
    // the three launches: beyond 256 tiles when a wave walks back for each of <= 8 buckets,
    // With more than 8 buckets every round beyond four times what the tail kernel takes (32 768 entries, the second
    // round of a byte text's 4 M-entry buckets) goes to the radix-pass form (induce_wide_*): one look-back thread per
    // bucket and tile made a 2 M-entry round of a 255-symbol text cost 100 us and more, but a round of a few tiles is
    // one launch of 20 us where the three took 50 (1 GiB of bytes through the induction: 122 -> 111 ms).
    st.chain_max = ctx->chain_max_override >= 0 ? (uint32_t)ctx->chain_max_override
                                                : (st.small_alphabet ? 256u * (uint32_t)kIndTile : 4u * kTailEntries);
    const size_t status_words = ((size_t)sx_div_up(largest, kIndTile) + 2) * nk + kChainHeader; // any round may be chained
    st.stride = sx_div_up(largest, kIndTile) + 1;
    st.hist = st.whist = st.wsums = st.bhist = st.btotals = nullptr;
    st.batch_on = ctx->induce_batch_off ? 0 : 1;
    if (st.small_alphabet) {
        st.hist = arena.take<uint32_t>((size_t)nk * st.stride);
        st.bhist = arena.take<uint32_t>((size_t)kBatchRows * st.stride);
        st.btotals = arena.take<uint32_t>(kBatchRows);
        if (!st.hist || !st.bhist || !st.btotals) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small (tile counts)");
    } else {
        const size_t wt = (size_t)sx_div_up(largest, kWideTile) + 2;
        st.whist = arena.take<uint32_t>(wt * 256);
        st.wsums = arena.take<uint32_t>((wt / kWideChunk + 2) * 256);
        if (!st.whist || !st.wsums) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small (tile counts)");
    }
    SX_TRY(sx_chain_slab(ctx, SX_SLAB_CHAIN, status_words * 8));
    st.status = (uint64_t *)ctx->slab[SX_SLAB_CHAIN].p;
    SX_CHECK(hipMemsetAsync(st.status, 0, sizeof(uint64_t), ctx->stream)); // the time-out word

    // windows of the sorted LMS suffixes: the only systematic text access of both passes, unless
    // they already came along with the sort keys (sx_lmssort.hip)
    if (widen)
        sx_launch(ctx, SX_KC_INDUCE_GATHER, ti.m * 12, widen_windows_kernel, dim3(sx_div_up(ti.m, kBlock * 4)), dim3(kBlock),
                  (const uint32_t *)seed_windows, (uint64_t)ti.m, (uint64_t *)seedW);
    else if (!seed_windows)
        sx_launch(ctx, SX_KC_INDUCE_GATHER, ti.m * (4 + sizeof(WT) + 16), fill_windows_kernel<WT>,
                  dim3(sx_div_up(ti.m, kBlock)), dim3(kBlock), ti.T, sorted_lms, ti.m, cfg, seedW);
    // the sentinel suffix (sa_is.c:463: SA[0] = n)
    sx_launch(ctx, SX_KC_MISC, 0, set_entry_kernel<WT>, dim3(1), dim3(1), SA, st.WN, st.BW, (uint32_t)ti.n, ti.T, cfg);

    // Unattended passes.  After a bucket's queued rounds the host used to read the bucket's last range back and wait
    // (20 - 30 us of idle device: 16 times a build at 5 buckets, 1000 times at 256) -- almost always to learn that the tail
    // kernel had finished the bucket.  Now the buckets are queued one behind the other.  A tail kernel that cannot finish
    // its bucket (runs of a symbol longer than its steps and jumps reach, or more entries alive than it holds: thousands
    // of poly-A tracts) leaves word: the bucket and its last range, on the device and in the host's pinned page
    // (tail_report).  From then on every set_range_kernel leaves its range empty, and launches over an empty range only
    // carry the cursors on -- the device's state stays what it was when the bucket stopped.  The host looks at the
    // pinned word between buckets (a plain load, no wait), stops queuing, reads the record, carries that bucket on
    // attended (read-backs, device-wide run jumps) and goes on unattended behind it.  Texts in which the classification
    // saw a run fill a whole 4096-symbol tile are attended from the start.
    st.poison = arena.take<uint32_t>(4);
    if (!st.poison) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small");
    st.host_poison = ctx->h_pin + 1040;
    // More than 8 buckets: every bucket's other-region round (its LMS seeds in the L pass, its L-type entries in the S
    // pass) up front, all buckets in one count / offsets / scatter, placed by the text's bigram counts (hoist_*_kernel)
    st.hoist = (!st.small_alphabet && !ctx->induce_no_hoist) ? 1 : 0;
    st.d_begin = st.hoist_E = st.hoist_tot = st.hoist_err = nullptr;
    st.hoist_from = 0;
    uint32_t *hz_BG = nullptr, *hz_EL = nullptr, *hz_ES = nullptr, *hz_tot = nullptr, *hz_dbase = nullptr, *hz_hist = nullptr,
             *hz_desc = nullptr;
    uint32_t h_desc[2][768]; // per pass: lo[256], len[256], first hist row[256] of every bucket's region (uploaded; alive to the end)
    uint32_t hz_rows[2] = {0, 0}, hz_most[2] = {1, 1};
    if (st.hoist) {
        for (int pass = 0; pass < 2; ++pass) {
            uint32_t row = 0;
            for (uint32_t c = 0; c < 256; ++c) {
                const uint32_t len = c < nk ? (pass == 0 ? ti.h_lms[c] : ti.h_l[c]) : 0u;
                h_desc[pass][c] = pass == 0 ? lms_off[c] : begin[c];
                h_desc[pass][256 + c] = len;
                h_desc[pass][512 + c] = row;
                const uint32_t tiles = sx_div_up(len, kWideTile);
                row += tiles;
                if (tiles > hz_most[pass]) hz_most[pass] = tiles;
            }
            hz_rows[pass] = row;
        }
        st.d_begin = arena.take<uint32_t>(260);
        hz_BG = arena.take<uint32_t>(65536);
        hz_EL = arena.take<uint32_t>(65536);
        hz_ES = arena.take<uint32_t>(65536);
        hz_tot = arena.take<uint32_t>(65536);
        hz_dbase = arena.take<uint32_t>(65536);
        hz_desc = arena.take<uint32_t>(2 * 768);
        st.hoist_err = arena.take<uint32_t>(4);
        hz_hist = arena.take<uint32_t>(((size_t)(hz_rows[0] > hz_rows[1] ? hz_rows[0] : hz_rows[1]) + 2) * 256);
        if (!st.d_begin || !hz_BG || !hz_EL || !hz_ES || !hz_tot || !hz_dbase || !hz_desc || !st.hoist_err || !hz_hist)
            return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small (up-front rounds)");
        SX_CHECK(hipMemcpyAsync(st.d_begin, begin, 257 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        SX_CHECK(hipMemcpyAsync(hz_desc, h_desc, sizeof h_desc, hipMemcpyHostToDevice, ctx->stream));
        SX_CHECK(hipMemsetAsync(hz_BG, 0, 65536 * sizeof(uint32_t), ctx->stream));
        SX_CHECK(hipMemsetAsync(st.hoist_err, 0, 4 * sizeof(uint32_t), ctx->stream));
        // the text's bigram counts (one pass per 32768 / nk rows of the matrix), then where every bucket's groups end
        uint32_t bg_grid = (uint32_t)sx_div_up(sx_div_up(ti.n ? ti.n : 1, 16), (uint64_t)kBigramThreads * 16);
        if (bg_grid > kBigramGrid) bg_grid = kBigramGrid;
        sx_launch(ctx, SX_KC_INDUCE_GATHER, ti.n * (uint64_t)sx_div_up(nk, kBigramWords / nk), bigram_kernel, dim3(bg_grid),
                  dim3(kBigramThreads), ti.T, (uint64_t)ti.n, nk, hz_BG);
        sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, hoist_tables_kernel, dim3(1), dim3(256), (const uint32_t *)hz_BG, (const uint32_t *)st.d_begin, nk,
                  hz_EL, hz_ES);
    }
    // all buckets' other-region rounds of a pass: one count, one offsets, one scatter (queued at the start of the pass)
    auto hoisted_rounds = [&](int pass) -> int {
        const uint32_t *desc = hz_desc + pass * 768;
        const int rev = pass, mode = pass == 0 ? MODE_L_FROM_LMS : MODE_S_FROM_L, dir = pass == 0 ? +1 : -1;
        const uint32_t *srcP = pass == 0 ? sorted_lms : (const uint32_t *)SA;
        const WT *srcW = pass == 0 ? (const WT *)seedW : (const WT *)st.WN;
        const uint8_t *srcB = pass == 0 ? (const uint8_t *)nullptr : (const uint8_t *)st.BW;
        st.hoist_E = pass == 0 ? hz_EL : hz_ES;
        st.hoist_tot = hz_tot;
        st.hoist_from = pass == 0 ? 0u : nk - 1u;
        uint32_t gx = hz_most[pass];
        if (gx > kHoistGridX) gx = kHoistGridX;
        sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, hoist_count_kernel<WT>, dim3(gx, nk), dim3(kWideThreads), srcW, srcB, desc, rev, mode, st.cfg,
                  hz_hist);
        sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)hz_rows[pass] * 2048, hoist_offsets_kernel, dim3(nk), dim3(kBlock * kHoistOffGroups),
                  hz_hist, desc, (const uint32_t *)st.hoist_E, dir, hz_tot, hz_dbase);
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, 0, hoist_scatter_kernel<WT, 8>, dim3(gx, nk), dim3(kWideThreads), srcP, srcW, desc, rev,
                  mode, st.cfg, st.T, (const uint32_t *)hz_hist, (const uint32_t *)hz_dbase, dir, st.SA, st.WN, st.BW,
                  (uint32_t)(pass == 0 ? 1 : 0));
        ctx->stats.induce_rounds++;
        return 0;
    };
    auto cursors_as_counted = [&](bool &ok) -> int {
        uint32_t cur[256];
        SX_TRY(sx_readback(ctx, (const uint32_t *)st.cursor[st.par], nk, cur));
        ok = true;
        // both passes end with every cursor between its bucket's L and S suffixes (bucket 0 holds the sentinel's suffix
        // alone, which no pass induces)
        for (uint32_t c = 1; c < nk; ++c)
            if (ti.h_all[c] && cur[c] != begin[c] + ti.h_l[c]) ok = false;
        return 0;
    };
    const bool unattended_ok = ctx->induce_attended != 1 && ti.open_tiles == 0;
    auto stopped = [&]() -> bool { return st.unattended && *(volatile uint32_t *)st.host_poison != 0; };
    // ---- L pass: buckets ascending, cursors at the bucket heads; from bucket `from` on (resume: that bucket's L region
    // carries on from the range an unattended run left)
    auto pass_L = [&](uint32_t from, const uint32_t *resume) -> int {
        for (uint32_t c = from; c < nk; ++c) {
            if (ti.h_all[c] == 0) continue;
            if (stopped()) return 0;
            const bool carry_on = resume && c == from;
            if (carry_on) st.hoist_from = c; // (its rounds are carried on from where they stopped: no bucket_begin_kernel)
            if (ti.h_l[c]) {
                uint32_t head_end = 0;
                st.unattended = (unattended_ok && !carry_on) ? 1 : 0;
                SX_TRY(run_self_rounds<WT>(st, begin[c], ti.h_l[c], 0, MODE_L_FROM_L, c, +1, 1, &head_end, (double)ti.h_all[c] / (double)N,
                                           carry_on ? resume : nullptr));
                if (!st.unattended && head_end - begin[c] != ti.h_l[c])
                    return sx_fail_msg(ctx, SX_E_INTERNAL, "induce L: bucket did not receive its L-type count");
                st.unattended = unattended_ok ? 1 : 0;
            }
            if (ti.h_lms[c] && !st.hoist) {
                sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, set_range_kernel, dim3(1), dim3(1), st.ranges, lms_off[c], lms_off[c + 1],
                          (const uint32_t *)st.cursor[st.par], (int)c, 0, st.tickets, 1u, (const uint32_t *)(st.unattended ? st.poison : nullptr));
                // (the round's size is known: the one form that takes it, and no launch that finds nothing to do)
                launch_round<WT>(st, sorted_lms, seedW, 0, -1, sx_div_up(ti.h_lms[c], kIndTile), sx_div_up(ti.h_lms[c], kIndTile), 0,
                                 MODE_L_FROM_LMS, c, +1, 0, 0, ti.h_lms[c] > st.chain_max ? 1 : 0);
            }
        }
        return 0;
    };
    // ---- S pass: buckets descending, cursors at the bucket ends -------------------------
    auto pass_S = [&](uint32_t from, const uint32_t *resume) -> int {
        for (uint32_t cc = from + 1; cc-- > 0;) {
            const uint32_t c = cc;
            if (ti.h_all[c] == 0) continue;
            if (stopped()) return 0;
            const bool carry_on = resume && c == from;
            if (carry_on) st.hoist_from = c;
            const uint32_t n_s = ti.h_all[c] - ti.h_l[c];
            if (c > 0 && n_s) {
                uint32_t tail_end = 0;
                st.unattended = (unattended_ok && !carry_on) ? 1 : 0;
                SX_TRY(run_self_rounds<WT>(st, begin[c + 1], n_s, 1, MODE_S_FROM_S, c, -1, 2, &tail_end, (double)ti.h_all[c] / (double)N,
                                           carry_on ? resume : nullptr));
                if (!st.unattended && begin[c + 1] - tail_end != n_s)
                    return sx_fail_msg(ctx, SX_E_INTERNAL, "induce S: bucket did not receive its S-type count");
                st.unattended = unattended_ok ? 1 : 0;
            }
            if (ti.h_l[c] && !st.hoist) {
                sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, set_range_kernel, dim3(1), dim3(1), st.ranges, begin[c],
                          begin[c] + ti.h_l[c], (const uint32_t *)st.cursor[st.par], (int)c, 0, st.tickets, 1u,
                          (const uint32_t *)(st.unattended ? st.poison : nullptr));
                launch_round<WT>(st, SA, st.WN, 0, -1, sx_div_up(ti.h_l[c], kIndTile), sx_div_up(ti.h_l[c], kIndTile), 1,
                                 MODE_S_FROM_L, c, -1, 0, 0, ti.h_l[c] > st.chain_max ? 1 : 0);
            }
        }
        return 0;
    };
    ctx->stats.long_runs = ti.open_tiles ? 1u : 0u;
    for (int pass = 0; pass < 2; ++pass) {
        st.unattended = unattended_ok ? 1 : 0;
        if (pass == 1) SX_CHECK(hipStreamSynchronize(ctx->stream)); // (`begin`, the L pass's upload source, may still be in use)
        SX_CHECK(hipMemcpyAsync(st.cursor[st.par], pass == 0 ? begin : begin + 1, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        if (st.hoist) SX_TRY(hoisted_rounds(pass));
        uint32_t from = pass == 0 ? 0u : nk - 1u, rec[4] = {0, 0, 0, 0};
        const uint32_t *resume = nullptr;
        for (uint32_t attempt = 0;; ++attempt) {
            if (attempt > 2 * nk + 4) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: a pass did not come to its end");
            if (unattended_ok) {
                *(volatile uint32_t *)st.host_poison = 0;
                SX_CHECK(hipMemsetAsync(st.poison, 0, 4 * sizeof(uint32_t), ctx->stream));
            }
            SX_TRY(pass == 0 ? pass_L(from, resume) : pass_S(from, resume));
            if (!unattended_ok) break;
            SX_TRY(sx_readback(ctx, (const uint32_t *)st.poison, 4, rec));
            if (!rec[0]) break;
            // bucket rec[1] stopped with the range [rec[2], rec[3]) alive: carry it on attended, then the buckets behind it
            ctx->stats.induce_redo++;
            from = rec[1];
            resume = rec + 2;
        }
        bool ok = false;
        SX_TRY(cursors_as_counted(ok));
        if (!ok) return sx_fail_msg(ctx, SX_E_INTERNAL, pass == 0 ? "induce L: a bucket did not receive its L-type count"
                                                                    : "induce S: a bucket did not receive its S-type count");
    }

    {
        uint32_t timed_out[2] = {0, 0};
        SX_TRY(sx_readback(ctx, (const uint32_t *)st.status, 2, timed_out));
        if (timed_out[0]) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: a look-back wait timed out");
    }
    if (st.hoist) {
        uint32_t err = 0;
        SX_TRY(sx_readback(ctx, (const uint32_t *)st.hoist_err, 1, &err));
        if (err) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: a bucket's cursor is not where the text's bigram counts put it");
    }
    if (ctx->prof_on) {
        // Algorithmic bytes of the two passes (the launches themselves were queued with bounds, not
        // sizes): the L pass scans every L-type entry and every LMS seed, the S pass every entry but
        // the sentinel's; every suffix is written once.  The counting launches read the symbol bytes (windows for the seeds), the
        // scatter launches the (position, window) pairs and write a symbol byte along; the few entries that went through the chained
        // rounds are booked here too.
        uint64_t n_l = 0;
        for (uint32_t c = 0; c < nk; ++c) n_l += ti.h_l[c];
        const uint64_t scanned = n_l + ti.m + (N - 1);
        // (with events around one class only -- bench.py's timed region -- that class alone is booked)
        if (ctx->prof_only < 0 || ctx->prof_only == SX_KC_INDUCE_GATHER)
            ctx->kstat[SX_KC_INDUCE_GATHER].alg_bytes += (scanned - ti.m) + ti.m * sizeof(WT); // symbol bytes; seeds: windows
        if (ctx->prof_only < 0 || ctx->prof_only == SX_KC_INDUCE_SCATTER)
            ctx->kstat[SX_KC_INDUCE_SCATTER].alg_bytes += (scanned + N) * (4 + sizeof(WT)) + N;
    }
    // st.BW now holds text[SA[i]-1] for every slot: the BWT (bwt.c:13-20), written along with the entries
    return 0;
}
} // namespace

// BWT from one-symbol windows of all suffixes in suffix-array order (the direct sort of wide alphabets)
int sx_bwt_from_seed_windows(sx_ctx *ctx, const uint32_t *seedw, uint64_t N, uint32_t maxc, uint8_t *bwt_out)
{
    wnd_cfg cfg;
    (void)sx_window_cfg(maxc, cfg);
    cfg.CW = 1;
    sx_launch(ctx, SX_KC_BWT_GATHER, N * 5, bwt_from_windows_kernel<uint32_t>, dim3(sx_div_up(N, kBlock * 16)), dim3(kBlock),
              seedw, N, cfg, bwt_out);
    return 0;
}

int sx_induce(sx_ctx *ctx, const sx_text_info &ti, uint32_t sigma, const uint32_t *sorted_lms,
              const void *seed_windows, bool seed_windows_u32, uint32_t *SA, uint8_t *bwt_out, sx_arena &arena)
{
    if (ti.N > 0xFFFFFFFFull) return sx_fail_msg(ctx, SX_E_ARG, "induce: n exceeds 32-bit positions");
    wnd_cfg cfg;
    const bool wide = sx_window_cfg(ti.maxc, cfg);
    if (!wide) return induce_typed<uint32_t>(ctx, ti, sigma, sorted_lms, seed_windows, false, SA, bwt_out, arena, cfg);
    return induce_typed<uint64_t>(ctx, ti, sigma, sorted_lms, seed_windows, seed_windows_u32, SA, bwt_out, arena, cfg);
}
