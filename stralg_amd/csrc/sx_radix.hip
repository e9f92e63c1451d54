// sx_radix.hip -- stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits a pass.
//
// Used for the LMS-substring piece sort (role of stralg/sa_is.c:295-336's
// induced LMS-substring order; the pass structure is that of
// stralg/skew.c:53-99: count, prefix sum, stable scatter) and for the
// reduced-string suffix sort.
//
// Per pass:
//   radix_hist     tile digit counts -> hist[tile][digit]            (reads keys)
//   radix_colsum / radix_bases / radix_apply
//                  hist[tile][digit] <- first output index of (digit, tile): a prefix sum
//                  down every digit column (tiles in chunks of kRadixChunk), digits stacked in
//                  order.  The table stays tile-major so that every kernel touches whole
//                  1 KiB rows; a digit-major table costs a 64-byte sector per 4-byte count
//                  on both sides, a quarter of the pass's traffic.
//   radix_scatter  wave-striped load, ballot-based stable ranking, tile
//                  re-ordered in LDS so that each digit's run leaves the CU
//                  as contiguous stores.
// HBM-bound: 2 x (8 + 4) B per pair and pass for the scatter, 8 B for the
// histogram read.  No MFMA (integer indexing).
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_scan.hpp"
#include "sx_internal.hpp"
#include "sx_lmskey.hpp"

namespace sx {

// Tile shape (tools/sortbench.sh, 3e8 pairs, 40 key bits, ms for the whole sort): 256 threads x 16
// keys 15.4; 512 x 8 15.3; 512 x 12 14.4; 512 x 16 13.5; 512 x 20 18.3; 1024 x 16 14.8.  What counts
// is the run a digit leaves the tile with: 8192 keys over 256 digits = 256-byte key runs.  Round 2, with
// the per-wave counters aliased into the key image (66 KB of LDS, two workgroups a CU either way): the same
// 8192-key tile as 1024 threads x 8 keys needs 60 VGPRs instead of 100, so 32 waves a CU instead of 16, and
// the scatter of the 1 GiB DNA build takes 5.26 ms instead of 5.51 (same box, twice each; tools/r02_run26.sh).
#ifndef SX_RADIX_ITEMS
#define SX_RADIX_ITEMS 8
#endif
constexpr int kRadixItems = SX_RADIX_ITEMS;
#ifndef SX_LMS_RANK_INORDER
#define SX_LMS_RANK_INORDER 0 // 1: the keyed first passes rank their pairs in order inside a wave too (rounds 1 - 4; A/B)
#endif
#ifndef SX_HIST_GRID
#define SX_HIST_GRID 8192u // (workgroups of the digit histogram; one per tile: 0.435 ms for the three passes of 1 GiB of DNA, this: 0.40)
#endif
#ifndef SX_RADIX_MINWAVES
#define SX_RADIX_MINWAVES 1
#endif
#ifndef SX_RADIX_THREADS
#define SX_RADIX_THREADS 1024
#endif
constexpr int kRT = SX_RADIX_THREADS, kRW = kRT / kWave;
constexpr int kRadixTile = kRT * kRadixItems;
// The histogram kernels walk the same tiles with their own shape: 16 keys (or digits) a thread suits their wide loads.
constexpr int kHT = 512, kHI = kRadixTile / kHT;
// the per-wave counters live in the key image when they would not leave room for two workgroups a CU otherwise
template <int DB> struct radix_alias { static constexpr bool value = DB > 8 || kRW > 8; };

// Digit width of a pass: 8 bits (256 digits: the default), 9 or 10.  Wider digits mean fewer passes over the
// pairs (40 key bits: 4 passes of 10 instead of 5 of 8) and shorter runs per (tile, digit): 8192 pairs over 1024
// digits leave 64-byte key runs, which only stay whole cache lines because consecutive tiles are handled by the
// same XCD at about the same time (see the tile order in the scatter kernel).
template <int DB> struct radix_digits {
    static_assert(DB >= 8 && DB <= 10, "digit width");
    static constexpr int ND = 1 << DB;
};
template <int DB> struct radix_dig_type { typedef uint16_t type; };
template <> struct radix_dig_type<8> { typedef uint8_t type; };

template <int DB>
__global__ __launch_bounds__(kHT) void radix_hist_kernel(const uint64_t *__restrict__ keys, uint64_t n,
                                                            int shift, uint32_t mask,
                                                            uint32_t *__restrict__ hist, uint32_t ntiles)
{
    constexpr int ND = 1 << DB;
    __shared__ uint32_t h[ND];
    for (int i = (int)threadIdx.x; i < ND; i += kHT) h[i] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * kRadixTile;
#pragma unroll
    for (int k = 0; k < kHI; ++k) {
        const uint64_t i = base + (uint64_t)k * kHT + threadIdx.x;
        const uint32_t d = i < n ? (uint32_t)(keys[i] >> shift) & mask : 0u;
        // (a wave of equal digits adds once: see radix_hist_digits_kernel)
        const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
        if (__all((i < n && d == lead) ? 1 : 0)) {
            if (lane_id() == 0) atomicAdd(&h[d], (uint32_t)kWave);
        } else if (i < n) {
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < ND; i += kHT) hist[(uint64_t)blockIdx.x * ND + i] = h[i];
}

// key of a position of a text-keyed sort (sx_textkey) from the 16 bytes at the position (q0, q1) and the byte in front of
// it: the value all_keys16_kernel (sx_lmssort.hip) would have stored
__device__ __forceinline__ uint64_t textkey_from(const sx_textkey &tk, uint64_t q0, uint64_t q1, uint32_t before)
{
    uint64_t acc;
    if (tk.base == 256u) { // (uniform) bytes: the first C of them as a big-endian number
        acc = tk.C <= 8u ? __builtin_bswap64(q0) >> (64u - 8u * tk.C)
                         : (__builtin_bswap64(q0) << (8u * (tk.C - 8u))) | (__builtin_bswap64(q1) >> (64u - 8u * (tk.C - 8u)));
    } else { // Horner, three symbols at a time (base^3 <= 2^24: 24-bit multiply-adds; the accumulator once a group)
        acc = 0;
        uint32_t g = 0;
#pragma unroll
        for (uint32_t s = 0; s < 12; ++s) {
            if (s < tk.C) { // uniform
                g = __umul24(g, tk.base) + (uint32_t)(((s < 8 ? q0 : q1) >> (8u * (s & 7u))) & 0xFFull);
                if (s % 3 == 2) { // static
                    acc = acc * tk.pow3 + g;
                    g = 0;
                }
            }
        }
        if (tk.C % 3u) acc = acc * tk.powR + g; // uniform
    }
    // the symbol in front of the suffix as a one-symbol window (it becomes the BWT)
    if (tk.wnd && before) acc |= (uint64_t)(((before - 1u) << 4) | 1u) << tk.kbits;
    return acc;
}
__device__ __forceinline__ uint64_t textkey_at(const sx_textkey &tk, uint64_t p)
{
    uint64_t q0, q1;
    load_bytes16(tk.T, p, q0, q1);
    return textkey_from(tk, q0, q1, p ? (uint32_t)tk.T[p - 1] : 0u);
}

template <int DB>
__global__ __launch_bounds__(kHT) void radix_hist_text_kernel(sx_textkey tk, uint64_t n, int shift, uint32_t mask,
                                                              uint32_t *__restrict__ hist, uint32_t ntiles)
{
    constexpr int ND = 1 << DB;
    __shared__ uint32_t h[ND];
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // (workgroups walk over their tiles)
        for (int i = (int)threadIdx.x; i < ND; i += kHT) h[i] = 0;
        __syncthreads();
        const uint64_t base = (uint64_t)tile * kRadixTile;
        // bytes whose digit is a whole byte of the key: that byte of the text, no key to compute
        const bool byte_digit = tk.base == 256u && (shift & 7) == 0 && mask == 0xFFu && shift < (int)(8u * tk.C);
        const uint32_t byte_at = byte_digit ? tk.C - 1u - (uint32_t)(shift >> 3) : 0u;
        if (byte_digit && kHI == 16) { // (uniform) the digits are the text's own bytes: 16 consecutive ones a thread, one load
            const uint64_t i0 = base + (uint64_t)threadIdx.x * 16u;
            uint64_t q0 = 0, q1 = 0;
            if (i0 < n) load_bytes16(tk.T, i0 + byte_at, q0, q1);
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (i0 + (uint64_t)e < n) atomicAdd(&h[(uint32_t)((e < 8 ? q0 : q1) >> (8 * (e & 7))) & 0xFFu], 1u);
        } else {
#pragma unroll
        for (int k = 0; k < kHI; ++k) {
            const uint64_t i = base + (uint64_t)k * kHT + threadIdx.x;
            uint32_t d = 0;
            if (i < n) d = byte_digit ? (uint32_t)tk.T[i + byte_at] : (uint32_t)(textkey_at(tk, i) >> shift) & mask;
            const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
            if (__all((i < n && d == lead) ? 1 : 0)) {
                if (lane_id() == 0) atomicAdd(&h[d], (uint32_t)kWave);
            } else if (i < n) {
                atomicAdd(&h[d], 1u);
            }
        }
        }
        __syncthreads();
        for (int i = (int)threadIdx.x; i < ND; i += kHT) hist[(uint64_t)tile * ND + i] = h[i];
        __syncthreads();
    }
}

// The same from the digits the previous pass's scatter wrote next to its output (one byte per key for 8-bit
// digits, two for wider ones, in the order of that output): an eighth / a quarter of the key array's traffic.
template <int DB>
__global__ __launch_bounds__(kHT) void radix_hist_digits_kernel(const typename radix_dig_type<DB>::type *__restrict__ dig,
                                                                uint64_t n, uint32_t *__restrict__ hist, uint32_t ntiles)
{
    constexpr int ND = 1 << DB;
    // four copies of the counters, a lane adds to copy (lane & 3): digits of natural text are skewed (a third of the keys
    // of a word-like text share one), and lanes adding to one LDS word are served one after the other
    constexpr int kCopies = DB == 8 ? 4 : 1;
    __shared__ uint32_t hh[kCopies][ND];
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // (workgroups walk over their tiles)
    for (int i = (int)threadIdx.x; i < kCopies * ND; i += kHT) (&hh[0][0])[i] = 0;
    __syncthreads();
    uint32_t *h = hh[threadIdx.x & (kCopies - 1)];
    const uint64_t base = (uint64_t)tile * kRadixTile + (uint64_t)threadIdx.x * kHI;
    static_assert(kHI == 16 || kHI == 8, "16 or 8 digits per thread: wide loads");
    const bool full = base + kHI <= n;
    if (DB == 8) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (full) {
            if (kHI == 16) {
                const uint4 v = *reinterpret_cast<const uint4 *>(dig + base);
                w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
            } else {
                const uint2 v = *reinterpret_cast<const uint2 *>(dig + base);
                w[0] = v.x, w[1] = v.y, w[2] = v.x, w[3] = v.y; // (the second half repeats the first: tested, not counted)
            }
        }
        // Sorted or repetitive input (the keys of a tie-refinement round, a periodic text) has long runs of one
        // digit, and 64 lanes adding to one LDS word are served one after the other: a wave whose digits are
        // all the same adds once, a thread whose digits are adds once (250 us a pass for 10^8 equal keys before).
        // (The wave-wide tests are reached by every lane: no collective under a branch.)
        const uint32_t first = w[0] & 0xFFu, rep = first * 0x01010101u;
        const bool mono = full && w[0] == rep && w[1] == rep && w[2] == rep && w[3] == rep;
        const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        const bool wave_mono = __all((mono && first == lead) ? 1 : 0);
        if (wave_mono) {
            if (lane_id() == 0) atomicAdd(&h[first], (uint32_t)kHI * kWave);
        } else if (mono) {
            atomicAdd(&h[first], (uint32_t)kHI);
        } else if (full) {
#pragma unroll
            for (int k = 0; k < kHI; ++k) atomicAdd(&h[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu], 1u);
        } else {
            for (uint64_t i = base; i < n && i < base + kHI; ++i) atomicAdd(&h[(uint32_t)dig[i] & (uint32_t)(ND - 1)], 1u);
        }
    } else if (full && kHI == 16) {
        const uint4 v0 = *reinterpret_cast<const uint4 *>(dig + base), v1 = *reinterpret_cast<const uint4 *>(dig + base + 8);
        const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) atomicAdd(&h[(w[k >> 1] >> (16 * (k & 1))) & (uint32_t)(ND - 1)], 1u);
    } else {
        for (uint64_t i = base; i < n && i < base + kHI; ++i) atomicAdd(&h[(uint32_t)dig[i] & (uint32_t)(ND - 1)], 1u);
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < ND; i += kHT) {
        uint32_t sum = 0;
#pragma unroll
        for (int cpy = 0; cpy < kCopies; ++cpy) sum += hh[cpy][i];
        hist[(uint64_t)tile * ND + i] = sum;
    }
    __syncthreads(); // the counters are zeroed for the next tile
    }
}

constexpr uint32_t kRadixChunk = 256; // tiles per chunk of the column sums
constexpr int kColBatch = 16;         // independent loads in flight per thread

// The three table kernels run with one thread per digit (ND threads a workgroup).
// column sums of one chunk of tiles: sums[chunk][digit]
template <int ND>
__global__ __launch_bounds__(ND) void radix_colsum_kernel(const uint32_t *__restrict__ hist, uint32_t ntiles,
                                                          uint32_t *__restrict__ sums)
{
    const uint32_t t0 = blockIdx.x * kRadixChunk;
    const uint32_t t1 = t0 + kRadixChunk < ntiles ? t0 + kRadixChunk : ntiles;
    uint32_t s = 0;
    for (uint32_t tb = t0; tb < t1; tb += kColBatch) {
        uint32_t x[kColBatch];
#pragma unroll
        for (int i = 0; i < kColBatch; ++i) x[i] = tb + i < t1 ? hist[(uint64_t)(tb + i) * ND + threadIdx.x] : 0u;
#pragma unroll
        for (int i = 0; i < kColBatch; ++i) s += x[i];
    }
    sums[(uint64_t)blockIdx.x * ND + threadIdx.x] = s;
}

// one workgroup: sums[chunk][digit] <- entries of the digit in earlier chunks; digit_base[digit] <- keys with a smaller digit
template <int ND>
__global__ __launch_bounds__(ND) void radix_bases_kernel(uint32_t *__restrict__ sums, uint32_t nchunks,
                                                         uint32_t *__restrict__ digit_base)
{
    __shared__ uint32_t lds[ND / kWave];
    uint32_t run = 0;
    for (uint32_t cb = 0; cb < nchunks; cb += kColBatch) {
        uint32_t x[kColBatch];
#pragma unroll
        for (int i = 0; i < kColBatch; ++i) x[i] = cb + i < nchunks ? sums[(uint64_t)(cb + i) * ND + threadIdx.x] : 0u;
#pragma unroll
        for (int i = 0; i < kColBatch; ++i) {
            if (cb + i < nchunks) sums[(uint64_t)(cb + i) * ND + threadIdx.x] = run;
            run += x[i];
        }
    }
    // exclusive prefix over the ND digits (one per thread)
    const uint32_t inc = wave_inclusive_scan<OpAdd>(run);
    if (lane_id() == kWave - 1) lds[wave_id()] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < wave_id(); ++i) base += lds[i];
    digit_base[threadIdx.x] = base + inc - run;
}

// hist[tile][digit] <- first output index of the tile's keys with that digit
template <int ND>
__global__ __launch_bounds__(ND) void radix_apply_kernel(uint32_t *__restrict__ hist, uint32_t ntiles,
                                                         const uint32_t *__restrict__ sums,
                                                         const uint32_t *__restrict__ digit_base)
{
    const uint32_t t0 = blockIdx.x * kRadixChunk;
    const uint32_t t1 = t0 + kRadixChunk < ntiles ? t0 + kRadixChunk : ntiles;
    uint32_t run = sums[(uint64_t)blockIdx.x * ND + threadIdx.x] + digit_base[threadIdx.x];
    for (uint32_t tb = t0; tb < t1; tb += kColBatch) {
        uint32_t x[kColBatch];
#pragma unroll
        for (int i = 0; i < kColBatch; ++i) x[i] = tb + i < t1 ? hist[(uint64_t)(tb + i) * ND + threadIdx.x] : 0u;
#pragma unroll
        for (int i = 0; i < kColBatch; ++i) {
            if (tb + i < t1) hist[(uint64_t)(tb + i) * ND + threadIdx.x] = run;
            run += x[i];
        }
    }
}

// The three table kernels in one for sorts of at most kRadixChunk tiles (2 M pairs): one workgroup, a thread per
// digit.  A pass of a small sort is all launch and latency (five dependent launches: ~100 us for 10^6 pairs; the tie
// refinement of a repetitive text runs sixty such passes), so three of its launches become one.
template <int ND>
__global__ __launch_bounds__(ND) void radix_offsets_small_kernel(uint32_t *__restrict__ hist, uint32_t ntiles)
{
    __shared__ uint32_t lds[ND / kWave];
    const uint32_t d = threadIdx.x;
    uint32_t total = 0;
    for (uint32_t tb = 0; tb < ntiles; tb += kColBatch) {
        uint32_t x[kColBatch];
#pragma unroll
        for (int i = 0; i < kColBatch; ++i) x[i] = tb + i < ntiles ? hist[(uint64_t)(tb + i) * ND + d] : 0u;
#pragma unroll
        for (int i = 0; i < kColBatch; ++i) total += x[i];
    }
    const uint32_t inc = wave_inclusive_scan<OpAdd>(total);
    if (lane_id() == kWave - 1) lds[wave_id()] = inc;
    __syncthreads();
    uint32_t run = inc - total; // keys with a smaller digit
    for (int i = 0; i < wave_id(); ++i) run += lds[i];
    for (uint32_t tb = 0; tb < ntiles; tb += kColBatch) {
        uint32_t x[kColBatch];
#pragma unroll
        for (int i = 0; i < kColBatch; ++i) x[i] = tb + i < ntiles ? hist[(uint64_t)(tb + i) * ND + d] : 0u;
#pragma unroll
        for (int i = 0; i < kColBatch; ++i) {
            if (tb + i < ntiles) hist[(uint64_t)(tb + i) * ND + d] = run;
            run += x[i];
        }
    }
}

// One tile of the scatter.  FULL: the tile lies inside the input (all but the last one): no bound checks.
// IOTA: the values of the input are its indices 0, 1, 2, ... (the first pass of a sort of all positions): not read.
// wcount: kRW rows of ND per-wave digit counters.  For digits wider than 8 bits the rows live in the key image
// (they are dead before the first key is staged), so that two workgroups still fit a CU's LDS.
template <int DB, bool IOTA, bool FULL, bool TEXT>
__device__ __forceinline__ void radix_scatter_tile(const sx_textkey &tk, const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                   uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, uint64_t n,
                                                   int shift, uint32_t mask, uint32_t tile, const uint32_t *__restrict__ offs,
                                                   typename radix_dig_type<DB>::type *__restrict__ dig_out, int next_shift,
                                                   uint32_t next_mask, uint32_t *wcount, uint32_t *goff, uint32_t *scan_lds,
                                                   uint64_t *skey)
{
    constexpr int ND = 1 << DB;
    constexpr int R = ND > kRT ? ND / kRT : 1; // digits per thread in the per-digit steps (consecutive ones)
    typedef typename radix_dig_type<DB>::type dig_t;
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const bool has_digits = t * R < ND;
    uint32_t first_out[R]; // first output index of (tile, digit): asked for now, needed after the ranking
#pragma unroll
    for (int r = 0; r < R; ++r) first_out[r] = has_digits ? offs[(uint64_t)tile * ND + t * R + r] : 0u;
    const uint64_t tile0 = (uint64_t)tile * kRadixTile;
    const uint64_t wave0 = tile0 + (uint64_t)w * (kWave * kRadixItems);
    uint64_t key[kRadixItems];
    uint32_t lpos[kRadixItems]; // [12:0] rank within (wave, digit), then slot in the tile's digit order; [31:16] digit
    static_assert(kRadixTile <= 65536, "slot and digit share a register");
    // TEXT: the tile's text, bytes [tile0 - 16, tile0 + kRadixTile + 32), behind the per-wave counters in the key image
    // (both are dead before the first key is staged): every key is then a few LDS reads instead of an unaligned 16-byte
    // load from memory (8 of them a thread: the first version of this pass lost to the key kernel it replaced)
    uint8_t *stxt = reinterpret_cast<uint8_t *>(skey) + (size_t)kRW * ND * sizeof(uint32_t);
    if constexpr (TEXT) {
        static_assert((size_t)kRW * ND * 4 + kRadixTile + 64 <= sizeof(uint64_t) * kRadixTile, "text image fits behind the counters");
        for (uint32_t piece = (uint32_t)t; piece < (uint32_t)kRadixTile / 16u + 3u; piece += kRT) {
            const uint64_t p0 = tile0 + (uint64_t)piece * 16u; // the piece holds text[p0 - 16, p0)
            uint4 v;
            v.x = v.y = v.z = v.w = 0;
            if (p0 >= 16u && p0 - 16u < n + 64u) v = *reinterpret_cast<const uint4 *>(tk.T + p0 - 16u); // (T is 16-byte aligned, padded 128 bytes beyond n)
            *reinterpret_cast<uint4 *>(stxt + (size_t)piece * 16u) = v;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint64_t i = wave0 + (uint64_t)k * kWave + lane;
        // read once, never again: streaming loads leave L2 to the runs being written (10.1 -> 9.95 ms per sort)
        if (TEXT) { // the keys of a text-keyed sort's first pass, from the tile's text staged in LDS (below)
            uint64_t q0, q1;
            const uint32_t at = 16u + (uint32_t)(i - tile0);
            lds_bytes16(stxt, at, q0, q1);
            key[k] = (FULL || i < n) ? textkey_from(tk, q0, q1, i ? (uint32_t)stxt[at - 1u] : 0u) : ~0ull;
        } else {
            key[k] = (FULL || i < n) ? __builtin_nontemporal_load(kin + i) : ~0ull;
        }
    }
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint64_t i = wave0 + (uint64_t)k * kWave + lane;
        const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
        // (TEXT: the first pass of its sort -- pairs of one digit may leave a wave in any order, see lms_scatter_row)
        if (TEXT && !SX_LMS_RANK_INORDER) lpos[k] = ((FULL || i < n) ? atomicAdd(&wcount[w * ND + d], 1u) : 0u) | (d << 16);
        else lpos[k] = wave_rank_inorder<DB, FULL>(d, FULL || i < n, wcount + w * ND) | (d << 16);
    }
    __syncthreads();
    {
        uint32_t s[R], tot = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s[r] = 0;
            if (has_digits) {
                const int d = t * R + r;
#pragma unroll
                for (int ww = 0; ww < kRW; ++ww) {
                    const uint32_t x = wcount[ww * ND + d];
                    wcount[ww * ND + d] = s[r];
                    s[r] += x;
                }
            }
            tot += s[r];
        }
        const uint32_t inc = wave_inclusive_scan<OpAdd>(tot);
        if (lane == kWave - 1) scan_lds[w] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int ww = 0; ww < w; ++ww) base += scan_lds[ww];
        uint32_t ex = base + inc - tot; // first slot of the thread's first digit inside the tile
        if (has_digits) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int d = t * R + r;
#pragma unroll
                for (int ww = 0; ww < kRW; ++ww) wcount[ww * ND + d] += ex; // first slot of (wave, digit)
                goff[d] = first_out[r] - ex;
                ex += s[r];
            }
        }
    }
    __syncthreads();
    // A large tile keeps the runs per digit long, so the LDS image is used twice, for the keys
    // and then for the values, instead of holding both.
    const uint64_t left = n - tile0;
    const uint32_t cnt = FULL || left >= (uint64_t)kRadixTile ? (uint32_t)kRadixTile : (uint32_t)left;
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) lpos[k] = (lpos[k] & 0xFFFFu) + wcount[w * ND + (lpos[k] >> 16)];
    if (radix_alias<DB>::value) __syncthreads(); // the counters share the key image: every slot is known before the first key lands
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint64_t i = wave0 + (uint64_t)k * kWave + lane;
        if (FULL || i < n) skey[lpos[k]] = key[k];
    }
    __syncthreads();
    uint32_t dstv[kRadixItems]; // destinations of the slots this thread copies out
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint32_t i = (uint32_t)t + (uint32_t)k * kRT;
        dstv[k] = 0;
        if (FULL || i < cnt) {
            const uint64_t kk = skey[i];
            const uint32_t d = (uint32_t)(kk >> shift) & mask;
            dstv[k] = goff[d] + i;
            kout[dstv[k]] = kk; // (streaming stores here cost 20 %: the runs of neighbouring tiles meet in L2)
            if (dig_out) dig_out[dstv[k]] = (dig_t)((uint32_t)(kk >> next_shift) & next_mask); // uniform test
        }
    }
    __syncthreads();
    uint32_t *sval = reinterpret_cast<uint32_t *>(skey);
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint64_t i = wave0 + (uint64_t)k * kWave + lane;
        // the values are only read now: fewer live registers
        if (FULL || i < n) sval[lpos[k]] = IOTA ? (uint32_t)i : __builtin_nontemporal_load(vin + i);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint32_t i = (uint32_t)t + (uint32_t)k * kRT;
        if (FULL || i < cnt) vout[dstv[k]] = sval[i];
    }
}

// (second launch bound: workgroups per CU to plan registers for; LDS already limits a CU to two)
template <int DB, bool IOTA, bool TEXT = false>
__global__ __launch_bounds__(kRT, SX_RADIX_MINWAVES) void radix_scatter_kernel(
    sx_textkey tk, const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint64_t *__restrict__ kout,
    uint32_t *__restrict__ vout, uint64_t n, int shift, uint32_t mask, const uint32_t *__restrict__ offs,
    uint32_t ntiles, typename radix_dig_type<DB>::type *__restrict__ dig_out /* digits of the NEXT pass, or null */,
    int next_shift, uint32_t next_mask)
{
    constexpr int ND = 1 << DB;
    __shared__ uint64_t skey[kRadixTile]; // the tile in digit order: keys first, then reused for the values
    // per-wave digit counters, then the first slot of each (wave, digit)
    __shared__ uint32_t wcount_own[radix_alias<DB>::value ? 1 : kRW * ND];
    __shared__ uint32_t goff[ND];         // global offset of the digit minus its first slot inside the tile
    __shared__ uint32_t scan_lds[kRW];
    static_assert((size_t)kRW * ND * sizeof(uint32_t) <= sizeof(uint64_t) * kRadixTile, "the counters fit the key image");
    uint32_t *wcount = radix_alias<DB>::value ? reinterpret_cast<uint32_t *>(skey) : wcount_own;

    const int t = (int)threadIdx.x;
    // Workgroups are dealt round robin to the 8 XCDs, each with its own L2.  Tiles that follow each other write
    // runs that follow each other (per digit), so XCD x takes the x-th eighth of the tiles, in order: the two
    // halves of a cache line shared by neighbouring runs then meet in one L2 instead of leaving two as partial lines.
    const uint32_t per_xcd = (ntiles + 7u) / 8u;
    const uint32_t tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return; // uniform
    for (int i = t; i < kRW * ND; i += kRT) wcount[i] = 0;
    __syncthreads();
    if ((uint64_t)(tile + 1) * kRadixTile <= n) // uniform
        radix_scatter_tile<DB, IOTA, true, TEXT>(tk, kin, vin, kout, vout, n, shift, mask, tile, offs, dig_out, next_shift, next_mask,
                                                 wcount, goff, scan_lds, skey);
    else
        radix_scatter_tile<DB, IOTA, false, TEXT>(tk, kin, vin, kout, vout, n, shift, mask, tile, offs, dig_out, next_shift, next_mask,
                                                  wcount, goff, scan_lds, skey);
}

// ---- the first pass of a four-letter text's LMS sort, keys computed on the way (sx_lmskey) -----------------------------------
// The key kernel wrote 13 bytes an LMS suffix (key, position, first digit) that the first pass read back at once: 8.3 GB of
// the 1 GiB DNA build's 91.  Here the first pass's tiles are pieces of the TEXT: a workgroup lists its piece's LMS positions
// from the classification's bit array, computes their keys from the text staged in LDS (the value lms_tile_keys_kernel
// stores) and goes on as radix_scatter_tile does.  Tiles hold as many pairs as their piece has LMS suffixes, in text order,
// so the pass is stable like any other.  (Accounted under SX_KC_KEYS, like the key kernel it replaces and the direct sort's
// text-keyed first pass: key generation, here with the first scatter in it.)  1 GiB of DNA, same box: key kernel 1.16 + histogram 0.11 + scatter 1.70 ms before,
// histogram 0.42 + scatter 1.85 now; the whole build 21.3 - 21.4 -> 20.6 - 21.0 ms.
#if SX_RADIX_THREADS == 1024 && SX_RADIX_ITEMS == 8
#define SX_RADIX_LMS_PASS 1
// Tiles of the pass: the text in blocks of kLmsBlockCls classification tiles.  Four of them (16 384 positions) hold at most a
// radix tile's 8192 LMS suffixes -- no two are neighbours --, but only 5 000 on uniform DNA: a tile 61 % full, every
// per-slot step of the ranking paid for 8192.  Round 5: six (24 576 positions, 7 500 LMS suffixes of uniform DNA: a tile
// nearly as full as any other pass's).  A block may then hold more than a radix tile does (an LMS suffix at every other
// position), so every block owns TWO rows of the tile table: the second one stays empty unless the block is taken as two
// halves.  Whether it is, the workgroup sees from the LMS bits it has just counted -- round 4 looked the block's counts up
// in the classification's tile table before it asked for its text (one more dependent trip to memory a workgroup: 2.8
// against 2.1 ms); now the whole block's bits and text are asked for at once, and the rare split block is done again by
// halves.
#ifndef SX_LMS_BLOCK_CLS
#define SX_LMS_BLOCK_CLS 6
#endif
constexpr int kLmsBlockCls = SX_LMS_BLOCK_CLS;
#ifndef SX_LMS_SPLIT_GRID
#define SX_LMS_SPLIT_GRID 256u // (the CPU test harness: 2)
#endif
constexpr uint32_t kLmsSplitGrid = SX_LMS_SPLIT_GRID; // workgroups of the launch that takes the listed (split) blocks by halves
constexpr int kLmsRows = kLmsBlockCls * (kClsTile / 2) > kRadixTile ? 2 : 1; // rows of the tile table a block owns
constexpr int kLmsBlock = kLmsBlockCls * kClsTile; // text positions of a block
constexpr int kLmsPerThread = kLmsBlock / kRT;     // positions whose LMS bits a thread lists
static_assert((kLmsPerThread == 16 || kLmsPerThread == 24) && kLmsPerThread * kRT == kLmsBlock, "two or three bytes of the bit array a thread");
static_assert((kLmsBlockCls / kLmsRows) * (kClsTile / 2) <= kRadixTile && kLmsBlockCls % kLmsRows == 0, "a row's span cannot overflow a radix tile");
static_assert(kLmsBlock <= 65536, "positions inside a tile are kept as 16-bit offsets");
// The tile's text in LDS, two bits a symbol: word q holds text[pos0 - 16 + 16 q ...), sixteen symbols, the first one in the
// top bits, codes symbol - 1 (the sentinel and the padding behind it: 0).  Read as one big-endian bit stream, the dense key
// of suffix p (dense4_finish) is the 2 C bits from symbol p on and its window (sx_window.hpp: the nearest symbol in the
// lowest field) the 2 W bits in front of them: three words and two funnel shifts a suffix, where the byte image cost nine
// words, eight byte-align steps and eight dot products (a hundred instructions: this pass was bound by them).
constexpr int kLmsPackWords = (kLmsBlock + 64) / 16;
constexpr size_t kLmsImgAt = (size_t)kRW * 256 * sizeof(uint32_t);          // behind the per-wave counters
constexpr size_t kLmsPosAt = kLmsImgAt + (size_t)kLmsPackWords * sizeof(uint32_t) + 16; // the listed positions behind the text
static_assert(kLmsPosAt + 2 * (size_t)kRadixTile <= sizeof(uint64_t) * kRadixTile, "text and positions fit the key image");

struct lms_span {
    uint32_t pos0, npos; // first text position, positions (a multiple of kClsTile; 0: the row is empty)
};
// block b of the text: all of it (half < 0), or its first / second half (a split block's two rows)
__device__ __forceinline__ lms_span lms_block_span(const sx_lmskey &lk, uint32_t block, int half)
{
    lms_span sp = {0, 0};
    const uint32_t c0 = block * kLmsBlockCls;
    if (c0 >= lk.cls_tiles) return sp;
    const uint32_t c1 = c0 + kLmsBlockCls < lk.cls_tiles ? c0 + kLmsBlockCls : lk.cls_tiles;
    const uint32_t cm = c0 + kLmsBlockCls / 2 < lk.cls_tiles ? c0 + kLmsBlockCls / 2 : lk.cls_tiles;
    if (half < 0) sp = {c0 * (uint32_t)kClsTile, (c1 - c0) * (uint32_t)kClsTile};
    else if (half == 0) sp = {c0 * (uint32_t)kClsTile, (cm - c0) * (uint32_t)kClsTile};
    else sp = {cm * (uint32_t)kClsTile, (c1 - cm) * (uint32_t)kClsTile};
    return sp;
}

__device__ __forceinline__ uint32_t lms_pack16(const uint4 &v, bool zeros)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t x = w[k];
        if (zeros) x += ((x - 0x01010101u) & ~x & 0x80808080u) >> 7; // a zero byte counts as symbol 1: code 0
        out = (out << 8) | (__builtin_amdgcn_udot4(x, 0x01041040u, 0u, false) - 85u); // sum (symbol - 1) * 4^(3 - j)
    }
    return out;
}
template <int THREADS, bool FLIGHT>
__device__ __forceinline__ void lms_tile_pack(const sx_lmskey &lk, const lms_span &sp, uint32_t *pk)
{
    if (!FLIGHT) { // (the scatter kernel has no registers to spare for a second load in flight: two workgroups a CU at 64)
        const uint64_t text_end = (uint64_t)lk.cls_tiles * kClsTile + 128;
        for (uint32_t q = threadIdx.x; q < (sp.npos + 64u) / 16u; q += THREADS) {
            const uint64_t at = (uint64_t)sp.pos0 + 16ull * q;
            uint4 v = {0, 0, 0, 0};
            if (at >= 16 && at <= text_end) v = *reinterpret_cast<const uint4 *>(lk.T + at - 16);
            pk[q] = lms_pack16(v, at < 16 || at >= lk.dense_n);
        }
        return;
    }
    constexpr int ROUNDS = (kLmsPackWords + THREADS - 1) / THREADS;
    const uint64_t text_end = (uint64_t)lk.cls_tiles * kClsTile + 128; // the build's copy of the text is padded that far
    const uint32_t words = (sp.npos + 64u) / 16u;
    uint4 v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) { // (every load asked for before the first is packed)
        const uint32_t q = threadIdx.x + (uint32_t)r * THREADS;
        const uint64_t at = (uint64_t)sp.pos0 + 16ull * q; // the word holds text[at - 16, at)
        v[r] = {0, 0, 0, 0};
        if (q < words && at >= 16 && at <= text_end) v[r] = *reinterpret_cast<const uint4 *>(lk.T + at - 16);
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t q = threadIdx.x + (uint32_t)r * THREADS;
        const uint64_t at = (uint64_t)sp.pos0 + 16ull * q;
        if (q < words) pk[q] = lms_pack16(v[r], at < 16 || at >= lk.dense_n /* text[n] = 0 and the padding; in front of the text */);
    }
}

// (key, window) word of LMS suffix p -- what lms_key_static returns for dense keys of CS symbols and windows of WS two-bit
// codes --, WND false: without the window.  pos0: first position of the span that pk holds.
template <int CS, int WS, bool WND>
__device__ __forceinline__ uint64_t lms_key_packed(const uint32_t *pk, uint32_t pos0, uint32_t p, const sx_lmskey &lk)
{
    static_assert(CS + WS <= 32, "key and window inside 64 bits of the stream");
    if (p < (uint32_t)WS) // the first few positions of the text: symbol by symbol from memory
        return lms_key_static<CS, WS, 2>(nullptr, 0, p, lk.T, lk.kc, lk.kbits, lk.wcfg, lk.dense_n, lk.lenbits);
    const uint32_t s0 = p - (uint32_t)WS - pos0 + 16u; // symbol of the stream where the window begins
    const uint32_t w = s0 >> 4, sh = (s0 & 15u) * 2u;
    const uint32_t a = pk[w], b = pk[w + 1], c = pk[w + 2];
    const uint32_t hi = sh ? (a << sh) | (b >> (32u - sh)) : a, lo = sh ? (b << sh) | (c >> (32u - sh)) : b;
    const uint64_t X = (uint64_t)hi << 32 | lo; // symbols p - WS ... p - WS + 31
    const uint64_t D = (X >> (64 - 2 * (WS + CS))) & ((1ull << (2 * CS)) - 1ull);
    uint64_t key = (D << lk.lenbits) | (uint64_t)((uint32_t)CS - dense4_zeros(p, (uint32_t)CS, lk.dense_n - 1));
    if (WND) key |= (uint64_t)(((uint32_t)(X >> (64 - 2 * WS)) << kCntBits) | (uint32_t)WS) << lk.kbits;
    return key;
}

// Lists the span's LMS suffixes and computes the keys of this thread's slots (slot i of the tile: wave w, item k, lane l
// <-> i = 512 w + 64 k + l, radix_scatter_tile's order); pos16[k / 2] holds the offsets of items k, k + 1 inside the span.
// Returns their number -- when that is more than a radix tile holds, nothing else has been done.  Ends behind a barrier;
// scan_lds is free again.
template <int CS, int WS, int BS>
__device__ __forceinline__ uint32_t lms_tile_keys(const sx_lmskey &lk, const lms_span &sp, uint32_t *pk, uint16_t *spos, uint32_t *scan_lds,
                                                  uint64_t (&key)[kRadixItems], uint32_t (&pos16)[kRadixItems / 2])
{
    static_assert(BS == 2, "two-bit codes");
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    uint32_t bits = 0;
    if ((uint32_t)t * kLmsPerThread < sp.npos) { // the thread's positions: two or three bytes of the bit array
        if (kLmsPerThread == 16) {
            bits = (uint32_t)lk.lmsbits[(size_t)sp.pos0 / 16 + (uint32_t)t];
        } else {
            const uint8_t *by = reinterpret_cast<const uint8_t *>(lk.lmsbits) + (size_t)sp.pos0 / 8 + (kLmsPerThread / 8) * (uint32_t)t;
            bits = (uint32_t)by[0] | (uint32_t)by[1] << 8 | (uint32_t)by[2] << 16;
        }
        const uint32_t left = sp.npos - (uint32_t)t * kLmsPerThread; // (a span is whole classification tiles, not whole threads)
        if (left < (uint32_t)kLmsPerThread) bits &= (1u << left) - 1u;
    }
    lms_tile_pack<kRT, false>(lk, sp, pk);
    const uint32_t mine = (uint32_t)__popc(bits);
    const uint32_t inc = wave_inclusive_scan<OpAdd>(mine);
    if (lane == kWave - 1) scan_lds[w] = inc;
    __syncthreads();
    uint32_t at = inc - mine, cnt = 0;
#pragma unroll
    for (int ww = 0; ww < kRW; ++ww) {
        const uint32_t x = scan_lds[ww];
        if (ww < w) at += x;
        cnt += x;
    }
    if (cnt > (uint32_t)kRadixTile) { // (uniform) a whole block with more LMS suffixes than a radix tile holds: the caller takes it by halves
        __syncthreads();
        return cnt;
    }
    while (bits) {
        const int i = __ffs(bits) - 1;
        bits &= bits - 1u;
        if (at < (uint32_t)kRadixTile) spos[at] = (uint16_t)(t * kLmsPerThread + i);
        ++at;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint32_t i = (uint32_t)w * (kWave * kRadixItems) + (uint32_t)k * kWave + (uint32_t)lane;
        const uint32_t off = i < cnt ? (uint32_t)spos[i] : 0u;
        key[k] = ~0ull;
        if (i < cnt) key[k] = lms_key_packed<CS, WS, true>(pk, sp.pos0, sp.pos0 + off, lk);
        if (k & 1) pos16[k / 2] |= off << 16;
        else pos16[k / 2] = off;
    }
    __syncthreads(); // (text and positions are dead: the image is the scatter's from here)
    return cnt;
}

// The digit counts of every row of the tile table.  No order is needed here, so a workgroup is four waves and every thread
// takes the LMS positions of its own 64 (96) positions.
constexpr int kLmsHistThreads = 256;
template <int CS, int WS, int BS>
__global__ __launch_bounds__(kLmsHistThreads) void radix_hist_lms_kernel(sx_lmskey lk, int shift, uint32_t mask, uint32_t *__restrict__ hist, uint32_t nblocks)
{
    constexpr int ND = 256, kCopies = 4, kPer = kLmsBlock / kLmsHistThreads, kWords = kPer / 32; // words of the bit array a thread
    static_assert(kPer % 32 == 0 && ND == kLmsHistThreads, "a thread per digit, whole 32-bit words of LMS bits a thread");
    static_assert(kLmsRows == 1 || (kLmsBlock / 2) % kPer == 0, "a thread's positions lie in one half of the block");
    __shared__ uint32_t pk[kLmsPackWords];
    __shared__ uint32_t hh[kLmsRows][kCopies][ND];
    __shared__ uint32_t s_cnt;
    const uint32_t block = blockIdx.x;
    if (block >= nblocks) return; // uniform
    const int t = (int)threadIdx.x;
    const lms_span sp = lms_block_span(lk, block, -1);
    for (int i = t; i < kLmsRows * kCopies * ND; i += kLmsHistThreads) (&hh[0][0][0])[i] = 0;
    if (t == 0) s_cnt = 0;
    uint32_t bits[kWords];
#pragma unroll
    for (int j = 0; j < kWords; ++j) bits[j] = 0;
    if ((uint32_t)t * kPer < sp.npos) {
        const uint32_t *bw = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(lk.lmsbits) + (size_t)sp.pos0 / 8) + (uint32_t)kWords * (uint32_t)t;
        const uint32_t left = sp.npos - (uint32_t)t * kPer; // (a span is whole classification tiles, not whole threads)
#pragma unroll
        for (int j = 0; j < kWords; ++j) {
            bits[j] = bw[j];
            if (left < 32u * (j + 1)) bits[j] = left > 32u * j ? bits[j] & ((1u << (left - 32u * j)) - 1u) : 0u;
        }
    }
    lms_tile_pack<kLmsHistThreads, true>(lk, sp, pk);
    __syncthreads();
    uint32_t *h = hh[0][t & (kCopies - 1)];
    if (kLmsRows == 2) { // (static) more LMS suffixes than a radix tile holds: the block is two rows, its halves
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < kWords; ++j) mine += (uint32_t)__popc(bits[j]);
        const uint32_t inc = wave_inclusive_scan<OpAdd>(mine);
        if (lane_id() == kWave - 1) atomicAdd(&s_cnt, inc);
        __syncthreads();
        if (s_cnt > (uint32_t)kRadixTile && (uint32_t)t * kPer >= (uint32_t)(kLmsBlock / 2)) h = hh[kLmsRows - 1][t & (kCopies - 1)];
    }
    // A digit that lies inside the key's symbol fields is eight bits of the stream, 2 C - 8 - (shift - lenbits) bits behind
    // the suffix's first symbol, whatever the suffix (the codes behind the end of the text are 0, as in the key): the four
    // words that hold them for 32 positions stay in registers, and a suffix costs a select, a shift and the add -- no LDS
    // read per suffix to wait for (0.42 -> 0.3 ms at 1 GiB).  The first pass of the hybrid sort always qualifies.
    const int above = shift - (int)lk.lenbits; // bits of the symbol fields below the digit
    if (lk.dense_n != 0 && mask == 0xFFu && above >= 0 && above + 8 <= 2 * CS) { // uniform
        const uint32_t delta = (uint32_t)(2 * CS - 8 - above);
#pragma unroll
        for (int j = 0; j < kWords; ++j) {
            uint32_t bj = bits[j];
            const uint32_t *q = pk + (kPer / 16) * t + 2 * j + 1; // the word of position t * kPer + 32 j
            const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            while (bj) {
                const int i = __ffs(bj) - 1;
                bj &= bj - 1u;
                const uint32_t o = 2u * (uint32_t)i + delta, wi = o >> 5, r = o & 31u; // o <= 90: words 0 .. 2 and the next
                const uint32_t hi = wi == 0 ? q0 : (wi == 1 ? q1 : q2), lo = wi == 0 ? q1 : (wi == 1 ? q2 : q3);
                atomicAdd(&h[(uint32_t)((((uint64_t)hi << 32) | lo) >> (56u - r)) & 0xFFu], 1u);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < kWords; ++j) {
            uint32_t bj = bits[j];
            while (bj) {
                const int i = __ffs(bj) - 1;
                bj &= bj - 1u;
                const uint32_t p = sp.pos0 + (uint32_t)(t * kPer + 32 * j + i);
                atomicAdd(&h[(uint32_t)(lms_key_packed<CS, WS, false>(pk, sp.pos0, p, lk) >> shift) & mask], 1u);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kLmsRows; ++r) {
        uint32_t sum = 0;
#pragma unroll
        for (int cpy = 0; cpy < kCopies; ++cpy) sum += hh[r][cpy][t];
        hist[((uint64_t)block * kLmsRows + (uint32_t)r) * ND + t] = sum;
    }
}

// One row of the pass's tile table: the LMS suffixes of span sp, listed, keyed, ranked and written to their places.  false
// (and nothing written): the span holds more of them than a radix tile does.
template <int CS, int WS, int BS>
__device__ __forceinline__ bool lms_scatter_row(const sx_lmskey &lk, const lms_span &sp, uint32_t row, uint64_t *skey, uint32_t *goff,
                                                uint32_t *scan_lds, uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, int shift,
                                                uint32_t mask, const uint32_t *__restrict__ offs, uint8_t *__restrict__ dig_out,
                                                int next_shift, uint32_t next_mask)
{
    constexpr int ND = 256;
    uint32_t *wcount = reinterpret_cast<uint32_t *>(skey);
    uint32_t *pk = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(skey) + kLmsImgAt);
    uint16_t *spos = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(skey) + kLmsPosAt);
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    for (int i = t; i < kRW * ND; i += kRT) wcount[i] = 0;
    const uint32_t first_out = t < ND ? offs[(uint64_t)row * ND + t] : 0u; // asked for now, needed after the ranking
    uint64_t key[kRadixItems];
    uint32_t pos16[kRadixItems / 2];
    const uint32_t cnt = lms_tile_keys<CS, WS, BS>(lk, sp, pk, spos, scan_lds, key, pos16);
    if (cnt > (uint32_t)kRadixTile) return false; // uniform: a whole block with more LMS suffixes than a radix tile holds
    const uint32_t wave0 = (uint32_t)w * (kWave * kRadixItems);
    uint32_t lpos[kRadixItems]; // [12:0] rank within (wave, digit), then slot in the tile's digit order; [31:16] digit
    // This is the sort's FIRST pass: nothing it is handed has an order that a later pass relies on, so pairs of one digit may
    // leave a wave in any order (later passes are stable with respect to THIS pass's output; equal keys are ties whatever
    // their order) -- a pair's rank within (wave, digit) is the counter's value at its atomic add, one LDS operation where the
    // in-order ranking costs some fifty vector instructions a pair (round 5: SX_LMS_RANK_INORDER=1 keeps those, A/B).
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint32_t i = wave0 + (uint32_t)k * kWave + (uint32_t)lane;
        const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
        if (SX_LMS_RANK_INORDER) lpos[k] = wave_rank_inorder<8, false>(d, i < cnt, wcount + w * ND) | (d << 16);
        else lpos[k] = (i < cnt ? atomicAdd(&wcount[w * ND + d], 1u) : 0u) | (d << 16);
    }
    __syncthreads();
    {
        uint32_t s = 0;
        if (t < ND) {
#pragma unroll
            for (int ww = 0; ww < kRW; ++ww) {
                const uint32_t x = wcount[ww * ND + t];
                wcount[ww * ND + t] = s;
                s += x;
            }
        }
        const uint32_t inc = wave_inclusive_scan<OpAdd>(s);
        if (lane == kWave - 1) scan_lds[w] = inc;
        __syncthreads();
        uint32_t ex = inc - s; // first slot of the thread's digit inside the tile
        for (int ww = 0; ww < w; ++ww) ex += scan_lds[ww];
        if (t < ND) {
#pragma unroll
            for (int ww = 0; ww < kRW; ++ww) wcount[ww * ND + t] += ex; // first slot of (wave, digit)
            goff[t] = first_out - ex;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) lpos[k] = (lpos[k] & 0xFFFFu) + wcount[w * ND + (lpos[k] >> 16)];
    __syncthreads(); // the counters share the key image: every slot is known before the first key lands
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint32_t i = wave0 + (uint32_t)k * kWave + (uint32_t)lane;
        if (i < cnt) skey[lpos[k]] = key[k];
    }
    __syncthreads();
    uint32_t dstv[kRadixItems]; // destinations of the slots this thread copies out
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint32_t i = (uint32_t)t + (uint32_t)k * kRT;
        dstv[k] = 0;
        if (i < cnt) {
            const uint64_t kk = skey[i];
            const uint32_t d = (uint32_t)(kk >> shift) & mask;
            dstv[k] = goff[d] + i;
            kout[dstv[k]] = kk;
            if (dig_out) dig_out[dstv[k]] = (uint8_t)((uint32_t)(kk >> next_shift) & next_mask); // uniform test
        }
    }
    __syncthreads();
    uint32_t *sval = reinterpret_cast<uint32_t *>(skey);
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint32_t i = wave0 + (uint32_t)k * kWave + (uint32_t)lane;
        if (i < cnt) sval[lpos[k]] = sp.pos0 + ((pos16[k / 2] >> (16 * (k & 1))) & 0xFFFFu);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRadixItems; ++k) {
        const uint32_t i = (uint32_t)t + (uint32_t)k * kRT;
        if (i < cnt) vout[dstv[k]] = sval[i];
    }
    return true;
}

// split_list: [0] the number of blocks that hold more LMS suffixes than a radix tile (an LMS suffix at every other position:
// never on random text), [1 ...] those blocks.  HALVES false: a workgroup a block, the whole block as row kLmsRows * block of
// the tile table; a block that turns out too full is put on the list and left alone.  HALVES true (a few workgroups, queued
// behind: they find the list empty): the listed blocks as their two halves, two rows each.
template <int CS, int WS, int BS, bool HALVES>
__global__ __launch_bounds__(kRT) SX_WAVES_PER_EU(8) void radix_scatter_lms_kernel(
    sx_lmskey lk, uint64_t *__restrict__ kout, uint32_t *__restrict__ vout, int shift, uint32_t mask, const uint32_t *__restrict__ offs,
    uint32_t nblocks, uint8_t *__restrict__ dig_out /* digits of the NEXT pass, or null */, int next_shift, uint32_t next_mask,
    uint32_t *__restrict__ split_list)
{
    constexpr int ND = 256;
    __shared__ uint64_t skey[kRadixTile]; // counters, text, positions; then the tile in digit order: keys, then values
    __shared__ uint32_t goff[ND];
    __shared__ uint32_t scan_lds[kRW];
    static_assert(radix_alias<8>::value, "the per-wave counters live in the key image");
    if (!HALVES) {
        const uint32_t per_xcd = (nblocks + 7u) / 8u; // (tile order: see radix_scatter_kernel)
        const uint32_t block = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
        if (block >= nblocks) return; // uniform
        const lms_span sp = lms_block_span(lk, block, -1);
        if (sp.npos == 0) return;
        if (!lms_scatter_row<CS, WS, BS>(lk, sp, block * (uint32_t)kLmsRows, skey, goff, scan_lds, kout, vout, shift, mask, offs, dig_out,
                                         next_shift, next_mask)) {
            if (threadIdx.x == 0) split_list[1u + atomicAdd(&split_list[0], 1u)] = block;
        }
    } else {
        const uint32_t n_split = split_list[0];
        for (uint32_t i = blockIdx.x; i < n_split; i += gridDim.x) { // uniform
            const uint32_t block = split_list[1u + i];
            for (int h = 0; h < 2; ++h) {
                const lms_span sp = lms_block_span(lk, block, h);
                if (sp.npos)
                    (void)lms_scatter_row<CS, WS, BS>(lk, sp, block * (uint32_t)kLmsRows + (uint32_t)h, skey, goff, scan_lds, kout, vout, shift,
                                                      mask, offs, dig_out, next_shift, next_mask);
                __syncthreads(); // (the image is read by the row's last stores and written by the next row)
            }
        }
    }
}
#else
#define SX_RADIX_LMS_PASS 0
#endif

} // namespace sx

using namespace sx;

// workspace of a sort of n pairs with db-bit digits: tile table, chunk sums, digit bases, then one digit per pair
static size_t sort_workspace(uint64_t n, int db, uint32_t &ntiles, uint32_t &nchunks)
{
    const size_t nd = (size_t)1 << db;
    ntiles = sx_div_up(n, kRadixTile);
    nchunks = sx_div_up(ntiles, kRadixChunk);
    return ((size_t)ntiles + nchunks + 1) * nd * sizeof(uint32_t) + (((n + 255) & ~(uint64_t)255) + 256) * (db > 8 ? 2 : 1);
}

int sx_sort_digit_bits(const sx_ctx *ctx)
{
    return ctx->radix_digit_bits >= 8 && ctx->radix_digit_bits <= 10 ? ctx->radix_digit_bits : 8;
}

void *sx_sort_digit_buffer(sx_ctx *ctx, uint64_t n, int digit_bits)
{
    uint32_t ntiles, nchunks;
    const size_t bytes = sort_workspace(n, digit_bits, ntiles, nchunks);
    if (sx_slab_ensure(ctx, SX_SLAB_SORT, bytes) != 0) return nullptr;
    return (uint8_t *)ctx->slab[SX_SLAB_SORT].p + ((size_t)ntiles + nchunks + 1) * ((size_t)1 << digit_bits) * sizeof(uint32_t);
}

// The LMS-keyed first pass (sx_lmskey): its tile table -- kLmsRows rows per kLmsBlockCls classification tiles, not one per 8192 pairs --
// lies in the sort's first key array, which such a sort never reads.
bool sx_sort_lms_keys_applies(uint64_t m, uint32_t cls_tiles, uint32_t shape)
{
#if SX_RADIX_LMS_PASS
    const uint64_t ntiles1 = (uint64_t)kLmsRows * sx_div_up(cls_tiles, kLmsBlockCls), nchunks1 = sx_div_up(ntiles1, kRadixChunk);
    if ((ntiles1 + nchunks1 + 1) * 256 * sizeof(uint32_t) + (ntiles1 / kLmsRows + 2) * sizeof(uint32_t) > m * sizeof(uint64_t)) return false; // (+ the list of split blocks)
    switch (shape) {
    case lms_key_shape(15, 12, 2):
    case lms_key_shape(16, 11, 2):
    case lms_key_shape(17, 10, 2):
    case lms_key_shape(18, 9, 2): return true;
    default: return false;
    }
#else
    (void)m, (void)cls_tiles, (void)shape;
    return false;
#endif
}

template <int DB>
static int sort_pairs_db(sx_ctx *ctx, uint64_t *ka, uint32_t *va, uint64_t *kb, uint32_t *vb, uint64_t n, int begin_bit,
                         int end_bit, int *result_in_b, bool values_are_indices, bool first_digits_ready, const sx_textkey *text_keys,
                         const sx_lmskey *lms_keys)
{
    const sx_textkey no_text = {nullptr, 0, 0, 1, 1, 0, 0};
    constexpr int ND = 1 << DB;
    typedef typename radix_dig_type<DB>::type dig_t;
    uint32_t ntiles, nchunks;
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_SORT, sort_workspace(n, DB, ntiles, nchunks)));
    uint32_t *hist = (uint32_t *)ctx->slab[SX_SLAB_SORT].p;
    uint32_t *sums = hist + (size_t)ntiles * ND, *digit_base = sums + (size_t)nchunks * ND;
    dig_t *dig = (dig_t *)(digit_base + ND); // next pass's digits, written by every scatter but the last
    uint64_t *kin = ka, *kout = kb;
    uint32_t *vin = va, *vout = vb;
    int flips = 0;
    const uint64_t dig_bytes = sizeof(dig_t);
    for (int shift = begin_bit; shift < end_bit; shift += DB) {
        const int bits = end_bit - shift < DB ? end_bit - shift : DB;
        const uint32_t mask = (1u << bits) - 1u;
        const int next_shift = shift + DB;
        const bool has_next = next_shift < end_bit;
        const int next_bits = has_next ? (end_bit - next_shift < DB ? end_bit - next_shift : DB) : 0;
#if SX_RADIX_LMS_PASS
        if (shift == begin_bit && lms_keys) {
            if constexpr (DB == 8) {
                const sx_lmskey &lk = *lms_keys;
                const uint32_t nblocks1 = sx_div_up(lk.cls_tiles, kLmsBlockCls);
                const uint32_t ntiles1 = (uint32_t)kLmsRows * nblocks1, nchunks1 = sx_div_up(ntiles1, kRadixChunk);
                if (!sx_sort_lms_keys_applies(n, lk.cls_tiles, lk.shape) || lk.m != n) return sx_fail_msg(ctx, SX_E_INTERNAL, "sort: LMS-keyed first pass");
                uint32_t *hist1 = (uint32_t *)ka, *sums1 = hist1 + (size_t)ntiles1 * ND, *base1 = sums1 + (size_t)nchunks1 * ND;
                uint32_t *split1 = base1 + ND; // the list of blocks taken by halves: a count, then at most nblocks1 blocks
                SX_CHECK(hipMemsetAsync(split1, 0, sizeof(uint32_t), ctx->stream));
                const uint64_t text_bytes = (uint64_t)lk.cls_tiles * (kClsTile + kClsTile / 8);
                const uint8_t next_mask8 = has_next ? (uint8_t)((1u << next_bits) - 1u) : 0;
#define SX_LMS_PASS(CS, WS, BS)                                                                                                          \
    case lms_key_shape(CS, WS, BS):                                                                                                      \
        sx_launch(ctx, SX_KC_RADIX_HIST, text_bytes, radix_hist_lms_kernel<CS, WS, BS>, dim3(nblocks1), dim3(kLmsHistThreads), lk, shift, mask, hist1, \
                  nblocks1);                                                                                                             \
        break;
                switch (lk.shape) {
                    SX_LMS_PASS(15, 12, 2)
                    SX_LMS_PASS(16, 11, 2)
                    SX_LMS_PASS(17, 10, 2)
                    SX_LMS_PASS(18, 9, 2)
                }
#undef SX_LMS_PASS
                if (nchunks1 == 1) {
                    sx_launch(ctx, SX_KC_SCAN, (uint64_t)ntiles1 * ND * 12, radix_offsets_small_kernel<ND>, dim3(1), dim3(ND), hist1, ntiles1);
                } else {
                    sx_launch(ctx, SX_KC_SCAN, (uint64_t)ntiles1 * ND * 4, radix_colsum_kernel<ND>, dim3(nchunks1), dim3(ND),
                              (const uint32_t *)hist1, ntiles1, sums1);
                    sx_launch(ctx, SX_KC_SCAN, (uint64_t)nchunks1 * ND * 8, radix_bases_kernel<ND>, dim3(1), dim3(ND), sums1, nchunks1, base1);
                    sx_launch(ctx, SX_KC_SCAN, (uint64_t)ntiles1 * ND * 8, radix_apply_kernel<ND>, dim3(nchunks1), dim3(ND), hist1, ntiles1,
                              (const uint32_t *)sums1, (const uint32_t *)base1);
                }
#define SX_LMS_PASS(CS, WS, BS)                                                                                                          \
    case lms_key_shape(CS, WS, BS):                                                                                                      \
        sx_launch(ctx, SX_KC_KEYS, text_bytes + n * (12 + (has_next ? 1 : 0)), radix_scatter_lms_kernel<CS, WS, BS, false>,              \
                  dim3(((nblocks1 + 7) / 8) * 8), dim3(kRT), lk, kout, vout, shift, mask, (const uint32_t *)hist1, nblocks1,              \
                  has_next ? (uint8_t *)dig : (uint8_t *)nullptr, next_shift & 63, (uint32_t)next_mask8, split1);                         \
        if (kLmsRows == 2)                                                                                                               \
            sx_launch(ctx, SX_KC_KEYS, 0, radix_scatter_lms_kernel<CS, WS, BS, true>, dim3(kLmsSplitGrid), dim3(kRT), lk, kout, vout,    \
                      shift, mask, (const uint32_t *)hist1, nblocks1, has_next ? (uint8_t *)dig : (uint8_t *)nullptr, next_shift & 63,   \
                      (uint32_t)next_mask8, split1);                                                                                     \
        break;
                switch (lk.shape) {
                    SX_LMS_PASS(15, 12, 2)
                    SX_LMS_PASS(16, 11, 2)
                    SX_LMS_PASS(17, 10, 2)
                    SX_LMS_PASS(18, 9, 2)
                }
#undef SX_LMS_PASS
                uint64_t *tk = kin; kin = kout; kout = tk;
                uint32_t *tv = vin; vin = vout; vout = tv;
                ++flips;
                ctx->stats.sort_passes++;
                continue;
            } else {
                return sx_fail_msg(ctx, SX_E_INTERNAL, "sort: LMS-keyed first pass with digits wider than 8 bits");
            }
        }
#else
        if (lms_keys) return sx_fail_msg(ctx, SX_E_INTERNAL, "sort: LMS-keyed first pass not built");
#endif
        if (shift == begin_bit && text_keys)
            sx_launch(ctx, SX_KC_RADIX_HIST, n, radix_hist_text_kernel<DB>, dim3(ntiles < SX_HIST_GRID ? ntiles : SX_HIST_GRID), dim3(kHT),
                      *text_keys, n, shift, mask, hist, ntiles);
        else if (shift == begin_bit && !first_digits_ready)
            sx_launch(ctx, SX_KC_RADIX_HIST, n * 8, radix_hist_kernel<DB>, dim3(ntiles), dim3(kHT), (const uint64_t *)kin, n, shift,
                      mask, hist, ntiles);
        else
            sx_launch(ctx, SX_KC_RADIX_HIST, n * dig_bytes, radix_hist_digits_kernel<DB>, dim3(ntiles < SX_HIST_GRID ? ntiles : SX_HIST_GRID), dim3(kHT),
                      (const dig_t *)dig, n, hist, ntiles);
        if (nchunks == 1) {
            sx_launch(ctx, SX_KC_SCAN, (uint64_t)ntiles * ND * 12, radix_offsets_small_kernel<ND>, dim3(1), dim3(ND), hist, ntiles);
        } else {
            sx_launch(ctx, SX_KC_SCAN, (uint64_t)ntiles * ND * 4, radix_colsum_kernel<ND>, dim3(nchunks), dim3(ND),
                      (const uint32_t *)hist, ntiles, sums);
            sx_launch(ctx, SX_KC_SCAN, (uint64_t)nchunks * ND * 8, radix_bases_kernel<ND>, dim3(1), dim3(ND), sums, nchunks,
                      digit_base);
            sx_launch(ctx, SX_KC_SCAN, (uint64_t)ntiles * ND * 8, radix_apply_kernel<ND>, dim3(nchunks), dim3(ND), hist, ntiles,
                      (const uint32_t *)sums, (const uint32_t *)digit_base);
        }
        if (text_keys && shift == begin_bit) {
            if constexpr (DB == 8) // (the text image lies behind the counters in the key image: 8-bit digits only)
                sx_launch(ctx, SX_KC_KEYS, n * (13 + (has_next ? dig_bytes : 0)), radix_scatter_kernel<DB, true, true>,
                          dim3(((ntiles + 7) / 8) * 8), dim3(kRT), *text_keys, (const uint64_t *)kin, (const uint32_t *)vin, kout, vout, n,
                          shift, mask, (const uint32_t *)hist, ntiles, has_next ? dig : (dig_t *)nullptr, next_shift & 63,
                          has_next ? (1u << next_bits) - 1u : 0u);
            else
                return sx_fail_msg(ctx, SX_E_INTERNAL, "sort: text-keyed first pass with digits wider than 8 bits");
        } else if (values_are_indices && shift == begin_bit)
            sx_launch(ctx, SX_KC_RADIX_SCATTER, n * (20 + (has_next ? dig_bytes : 0)), radix_scatter_kernel<DB, true>,
                      dim3(((ntiles + 7) / 8) * 8), dim3(kRT), no_text, (const uint64_t *)kin, (const uint32_t *)vin, kout, vout, n, shift,
                      mask, (const uint32_t *)hist, ntiles, has_next ? dig : (dig_t *)nullptr, next_shift & 63,
                      has_next ? (1u << next_bits) - 1u : 0u);
        else
            sx_launch(ctx, SX_KC_RADIX_SCATTER, n * (24 + (has_next ? dig_bytes : 0)), radix_scatter_kernel<DB, false>,
                      dim3(((ntiles + 7) / 8) * 8), dim3(kRT), no_text, (const uint64_t *)kin, (const uint32_t *)vin, kout, vout, n, shift,
                      mask, (const uint32_t *)hist, ntiles, has_next ? dig : (dig_t *)nullptr, next_shift & 63,
                      has_next ? (1u << next_bits) - 1u : 0u);
        uint64_t *tk = kin; kin = kout; kout = tk;
        uint32_t *tv = vin; vin = vout; vout = tv;
        ++flips;
        ctx->stats.sort_passes++;
    }
    *result_in_b = flips & 1;
    return 0;
}

int sx_sort_pairs(sx_ctx *ctx, uint64_t *ka, uint32_t *va, uint64_t *kb, uint32_t *vb, uint64_t n,
                  int begin_bit, int end_bit, int *result_in_b, bool values_are_indices, bool first_digits_ready,
                  int digit_bits, const sx_textkey *text_keys, const sx_lmskey *lms_keys)
{
    *result_in_b = 0;
    if (n == 0 || end_bit <= begin_bit) return 0;
    if (n > 0xFFFFFFFFull) return sx_fail_msg(ctx, SX_E_ARG, "sort: n exceeds 32-bit positions");
    switch (digit_bits ? digit_bits : sx_sort_digit_bits(ctx)) {
    case 9: return sort_pairs_db<9>(ctx, ka, va, kb, vb, n, begin_bit, end_bit, result_in_b, values_are_indices, first_digits_ready, text_keys, lms_keys);
    case 10: return sort_pairs_db<10>(ctx, ka, va, kb, vb, n, begin_bit, end_bit, result_in_b, values_are_indices, first_digits_ready, text_keys, lms_keys);
    default: return sort_pairs_db<8>(ctx, ka, va, kb, vb, n, begin_bit, end_bit, result_in_b, values_are_indices, first_digits_ready, text_keys, lms_keys);
    }
}

extern "C" int sx_prim_sort_pairs_dev(sx_ctx *ctx, uint64_t *d_keys_a, uint32_t *d_vals_a, uint64_t *d_keys_b,
                                      uint32_t *d_vals_b, uint64_t n, int begin_bit, int end_bit,
                                      int *result_in_b)
{
    if (!ctx || !result_in_b || begin_bit < 0 || end_bit > 64) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    SX_TRY(sx_sort_pairs(ctx, d_keys_a, d_vals_a, d_keys_b, d_vals_b, n, begin_bit, end_bit, result_in_b, false, false));
    return sx_sync(ctx);
}

extern "C" int sx_prim_exclusive_sum_dev(sx_ctx *ctx, const uint32_t *d_in, uint32_t *d_out, uint64_t n,
                                         uint32_t *d_total)
{
    if (!ctx) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    SX_TRY((device_scan<OpAdd>(ctx, n, InU32{d_in}, OutExclusive{d_out}, d_total)));
    return sx_sync(ctx);
}
