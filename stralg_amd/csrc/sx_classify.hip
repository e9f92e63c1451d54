// sx_classify.hip -- level-0 streaming passes over the text.
//
//   cls_first / cls_resolve / cls_types   S/L classification as a backward
//        segmented scan (stralg/sa_is.c:134-153 classify_SL), LMS predicate
//        (sa_is.c:155-162), and the three per-symbol histograms: bucket sizes
//        (sa_is.c:164-174 compute_buckets), L-type counts, LMS counts.
//   samp_flags / samp_write               sample positions = LMS positions plus a
//        cut every W symbols inside LMS substrings longer than W, so that every
//        piece fits one 64-bit sort key.
//   piece_keys                            pieces -> keys (comparison semantics of
//        sa_is.c:265-292 equal_LMS extended to an order).
//
// All passes are HBM streaming: 16 text bytes per thread as one dwordx4 load,
// a workgroup owns 4096 consecutive positions, types are resolved with
// 64-lane ballots inside a wave and a 4-entry LDS hand-off between waves.
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_scan.hpp"
#include "sx_internal.hpp"

namespace sx {


__device__ __forceinline__ void load_chunk(const uint8_t *__restrict__ T, uint64_t p0, uint32_t (&c)[17])
{
    const uint4 v = *reinterpret_cast<const uint4 *>(T + p0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
    c[16] = T[p0 + 16];
}

// bit i of dmask: the type of position p0+i is decided by its right neighbour;
// bit i of vmask: ... and it is S.  The sentinel position n is S by definition.
__device__ __forceinline__ void decided_masks(const uint32_t (&c)[17], uint64_t p0, uint64_t n,
                                              uint32_t &dmask, uint32_t &vmask)
{
    dmask = 0;
    vmask = 0;
    if (p0 + 16 <= n) { // everywhere but at the very end of the text: no position needs the 64-bit bound checks
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dmask |= (c[i] != c[i + 1] ? 1u : 0u) << i;
            vmask |= (c[i] < c[i + 1] ? 1u : 0u) << i;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint64_t pos = p0 + i;
        bool dec, val;
        if (pos < n) {
            dec = c[i] != c[i + 1];
            val = c[i] < c[i + 1];
        } else {
            dec = pos == n;
            val = true;
        }
        dmask |= (dec ? 1u : 0u) << i;
        vmask |= ((dec && val) ? 1u : 0u) << i;
    }
}

// ---- pass 1: type of each tile's first position, if the tile decides it -------
// One wave per tile, 1024 positions a step: almost every tile decides in its first few symbols, so the wave
// stops after the first step and three quarters of the text are not read by this pass.
// (src / src_tiles: the first src_tiles tiles are read from the caller's text, which the build has not copied yet -- see
// cls_types_kernel; T holds the text's tail, the sentinel and the padding from the start)
__global__ __launch_bounds__(kWave) void cls_first_kernel(const uint8_t *__restrict__ Tcopy, uint64_t n,
                                                          uint8_t *__restrict__ tile_first, uint32_t *__restrict__ open_tiles,
                                                          const uint8_t *__restrict__ src, uint32_t src_tiles)
{
    const uint8_t *__restrict__ T = blockIdx.x < src_tiles ? src : Tcopy;
    const int lane = lane_id();
    for (uint32_t seg = 0; seg < (uint32_t)kClsTile / (kWave * kClsPerThread); ++seg) {
        const uint64_t p0 = (uint64_t)blockIdx.x * kClsTile + (uint64_t)seg * (kWave * kClsPerThread) +
                            (uint64_t)lane * kClsPerThread;
        uint32_t c[17], dmask, vmask;
        load_chunk(T, p0, c);
        decided_masks(c, p0, n, dmask, vmask);
        const uint64_t has = __ballot(dmask != 0 ? 1 : 0);
        if (has) { // uniform
            const int first = __ffsll((unsigned long long)has) - 1;
            const uint32_t mine = dmask ? (vmask >> (__ffs(dmask) - 1)) & 1u : 0u;
            const uint32_t val = __shfl(mine, first, kWave);
            if (lane == 0) tile_first[blockIdx.x] = (uint8_t)val;
            return;
        }
    }
    if (lane == 0) {
        tile_first[blockIdx.x] = (uint8_t)2;
        atomicAdd(open_tiles, 1u); // (rare: a run of one symbol through the whole tile and beyond)
    }
}

// ---- pass 2: tiles made of one symbol whose run continues take the type of the
// next tile to the right.  A workgroup resolves 4096 tiles; the type that enters its chunk from the right is the
// nearest decided tile beyond it, found by looking (almost always one tile far; whatever another workgroup has
// resolved there in the meantime is the same type).  One workgroup walking over all tiles took 150 us at 1 GiB.
__global__ __launch_bounds__(kBlock) void cls_resolve_kernel(uint8_t *__restrict__ tile_first, uint32_t ntiles)
{
    __shared__ uint32_t whas[kWavesPerBlock], wval[kWavesPerBlock], carry_s;
    const int lane = lane_id(), w = wave_id();
    // r = distance from the last tile; tile index = ntiles - 1 - r
    const uint64_t start = (uint64_t)blockIdx.x * kBlock * 16;
    if (threadIdx.x == 0) carry_s = 1; // nothing decided to the right: the sentinel's type
    __syncthreads();
    if (w == 0) { // the nearest decided tile at r < start, 256 tiles a step (four loads in flight per lane)
        bool found = false;
        for (uint64_t base = start; base > 0 && !found; base -= (base < 4ull * kWave ? base : 4ull * kWave)) { // uniform
            uint32_t v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t d = (uint64_t)q * kWave + (uint64_t)lane; // distance below base - 1
                v[q] = d < base ? tile_first[ntiles - 1 - (base - 1 - d)] : 2u;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t dm = __ballot(v[q] != 2u ? 1 : 0);
                if (dm && !found) { // uniform
                    const uint32_t val = __shfl(v[q], __ffsll((unsigned long long)dm) - 1, kWave);
                    if (lane == 0) carry_s = val;
                    found = true;
                }
            }
        }
    }
    __syncthreads();
    {
        const uint64_t r0 = start + (uint64_t)threadIdx.x * 16;
        uint32_t v[16];
        bool has = false;
        uint32_t last = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint64_t r = r0 + k;
            v[k] = r < ntiles ? tile_first[ntiles - 1 - r] : 2u;
            if (v[k] != 2u) {
                has = true;
                last = v[k];
            }
        }
        const uint64_t hm = __ballot(has ? 1 : 0);
        const uint64_t fm = __ballot((has && last) ? 1 : 0);
        if (lane == 0) {
            whas[w] = hm != 0;
            wval[w] = hm ? (uint32_t)((fm >> (63 - __clzll((unsigned long long)hm))) & 1ull) : 0u;
        }
        __syncthreads();
        uint32_t cin;
        const uint64_t lower = hm & lanemask_lt();
        if (lower) {
            cin = (uint32_t)((fm >> (63 - __clzll((unsigned long long)lower))) & 1ull);
        } else {
            cin = carry_s;
            for (int ww = 0; ww < w; ++ww)
                if (whas[ww]) cin = wval[ww];
        }
        uint32_t cur = cin;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint64_t r = r0 + k;
            if (v[k] != 2u) cur = v[k];
            if (r < ntiles) tile_first[ntiles - 1 - r] = (uint8_t)cur;
        }
    }
}

// Type of the first position to the right of this thread's chunk: the first
// decided position of a later thread of the tile, else the next tile's first type.
__device__ __forceinline__ uint32_t carry_from_right(bool has, uint32_t first_val, uint32_t tile_carry,
                                                     uint32_t *lds /* 2 * kWavesPerBlock */)
{
    const int lane = lane_id(), w = wave_id();
    const uint64_t hm = __ballot(has ? 1 : 0);
    const uint64_t fm = __ballot((has && first_val) ? 1 : 0);
    if (lane == 0) {
        lds[w] = hm != 0;
        lds[kWavesPerBlock + w] = hm ? (uint32_t)((fm >> (__ffsll((unsigned long long)hm) - 1)) & 1ull) : 0u;
    }
    __syncthreads();
    const uint64_t higher = lane == 63 ? 0ull : ((hm >> (lane + 1)) << (lane + 1));
    uint32_t cin;
    if (higher) {
        cin = (uint32_t)((fm >> (__ffsll((unsigned long long)higher) - 1)) & 1ull);
    } else {
        cin = tile_carry;
        for (int ww = kWavesPerBlock - 1; ww > w; --ww)
            if (lds[ww]) cin = lds[kWavesPerBlock + ww];
    }
    __syncthreads();
    return cin;
}

// ---- pass 3: types, LMS bits, histograms ------------------------------------------
// The kernel is bound by instruction issue, not by memory (500 instructions per thread and tile when every byte
// is extracted and compared on its own: 1.5 ms at 1 GiB), so the 16 bytes of a thread stay in their four words:
//  * "smaller than / different from the right neighbour" per byte without carries between bytes (the high bit of
//    each byte is handled apart), one three-input bit operation each, and the four flag bits of a word are
//    gathered by a dot product with the weights 1, 2, 4, 8 (16 ... 128 for the next word);
//  * histograms of symbols below 8 (all of DNA): the three bit planes of the symbols are gathered the same way,
//    the 16-bit "is symbol a" masks are one bit operation each, and the counts are popcounts of those masks
//    (and of their intersections with the type masks) that accumulate in registers over all the tiles a
//    workgroup walks; they are reduced over the wave once, at the end.
// The build works on a copy of the text with the sentinel and 16-byte-load padding behind it.  Copying 1 GiB is
// 0.38 ms of a 23 ms build, and this kernel reads every byte of the text anyway while it waits for its vector
// unit: it reads the first src_tiles tiles (all but the text's last 32 bytes or so) from the caller's buffer and
// stores them into the copy as it goes; only the tail was copied beforehand.  (src == nullptr: T is complete.)
__global__ __launch_bounds__(kBlock) void cls_types_kernel(
    uint8_t *__restrict__ T, uint64_t n, const uint8_t *__restrict__ tile_first, uint32_t ntiles,
    uint16_t *__restrict__ lmsbits, uint32_t *__restrict__ tile_lms, uint32_t *__restrict__ tile_last,
    uint32_t *__restrict__ g_hist /* 3 * 256 */, const uint8_t *__restrict__ src, uint32_t src_tiles)
{
    __shared__ uint32_t h[3][256];
    __shared__ uint32_t lds[2 * kWavesPerBlock];
    __shared__ uint32_t last_s[kBlock];
    const int t = (int)threadIdx.x;
    h[0][t] = 0;
    h[1][t] = 0;
    h[2][t] = 0;
    __syncthreads();
    uint32_t cnt_all[8], cnt_s[8], cnt_lms[8]; // symbols below 8: all / S-type / LMS positions of this thread so far
#pragma unroll
    for (int a = 0; a < 8; ++a) cnt_all[a] = cnt_s[a] = cnt_lms[a] = 0;
    // A workgroup walks over many tiles and adds its histograms to the global ones once at the end: a global
    // atomic per tile and symbol (262 144 tiles on 15 addresses at 1 GiB of DNA) serialises at the memory side
    // and was most of this kernel's time.
    // (the next tile's 17 bytes are asked for before this tile's barriers: SX_CLS_PREFETCH)
    uint4 v_next = {0u, 0u, 0u, 0u};
    uint32_t b_next = 0;
#ifndef SX_CLS_PREFETCH
#define SX_CLS_PREFETCH 1
#endif
    if (SX_CLS_PREFETCH && blockIdx.x < ntiles) {
        const uint64_t q0 = (uint64_t)blockIdx.x * kClsTile + (uint64_t)t * kClsPerThread;
        const uint8_t *__restrict__ S0 = blockIdx.x < src_tiles ? src : T;
        v_next = *reinterpret_cast<const uint4 *>(S0 + q0);
        b_next = (uint32_t)S0[q0 + 16];
    }
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // uniform
        const uint64_t tile0 = (uint64_t)tile * kClsTile;
        const uint64_t p0 = tile0 + (uint64_t)t * kClsPerThread;
        const uint8_t *__restrict__ S = tile < src_tiles ? src : T; // (uniform per workgroup and tile)
        uint4 v;
        uint32_t b16;
        if (SX_CLS_PREFETCH) {
            v = v_next, b16 = b_next;
            const uint32_t nt = tile + gridDim.x;
            if (nt < ntiles) {
                const uint64_t q0 = (uint64_t)nt * kClsTile + (uint64_t)t * kClsPerThread;
                const uint8_t *__restrict__ Sn = nt < src_tiles ? src : T;
                v_next = *reinterpret_cast<const uint4 *>(Sn + q0);
                b_next = (uint32_t)Sn[q0 + 16];
            }
        } else {
            v = *reinterpret_cast<const uint4 *>(S + p0);
            b16 = (uint32_t)S[p0 + 16];
        }
        const uint32_t w[5] = {v.x, v.y, v.z, v.w, b16};
        if (tile < src_tiles) *reinterpret_cast<uint4 *>(T + p0) = v; // the copy the rest of the build reads
        const bool inside = p0 + 16 <= n; // everywhere but at the very end of the text
        uint32_t dmask, vmask;
        if (inside) {
            uint32_t ne[4], lt[4];
            const uint32_t H = 0x80808080u, L = 0x7F7F7F7Fu;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t x = w[j], y = __builtin_amdgcn_alignbyte(w[j + 1], w[j], 1u); // y: the right neighbours
                const uint32_t d = (x | H) - (y & L); // per byte, no borrow: bit 7 = "low 7 bits of x >= those of y"
                // x < y: the high bits say so, or they agree and the low bits do
                lt[j] = __builtin_amdgcn_bitop3_b32(x, y, d, 0x4D) & H; // (~x & y) | (~(x ^ y) & ~d)
                const uint32_t e = x ^ y;
                ne[j] = (((e & L) + L) | e) & H;
            }
            dmask = gather16(ne[0], ne[1], ne[2], ne[3], 7);
            vmask = gather16(lt[0], lt[1], lt[2], lt[3], 7);
        } else {
            uint32_t c[17];
#pragma unroll
            for (int i = 0; i < 17; ++i) c[i] = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            decided_masks(c, p0, n, dmask, vmask);
        }
        const uint32_t tile_carry = tile + 1 < ntiles ? tile_first[tile + 1] : 1u;
        const bool has = dmask != 0;
        const uint32_t fv = has ? (vmask >> (__ffs(dmask) - 1)) & 1u : 0u;
        uint32_t cur = carry_from_right(has, fv, tile_carry, lds); // includes barriers
        // every position takes the type of the nearest decided position at or above it, else the carry: the decided
        // values spread downwards through the undecided bits in four doubling steps
        uint32_t smask = vmask & dmask, known = dmask;
#pragma unroll
        for (int k = 1; k < 16; k <<= 1) {
            smask |= (smask >> k) & ~known;
            known |= known >> k;
        }
        if (cur) smask |= ~known & 0xFFFFu;
        last_s[t] = (smask >> 15) & 1u;
        __syncthreads();
        uint32_t prev_s;
        if (t > 0) {
            prev_s = last_s[t - 1];
        } else if (tile0 == 0) {
            prev_s = 1; // position 0 is never LMS
        } else {
            const uint32_t a = (tile - 1 < src_tiles ? src : T)[tile0 - 1], b = w[0] & 0xFFu; // (the copy of the tile before may not be written yet)
            prev_s = a < b ? 1u : (a > b ? 0u : (smask & 1u));
        }
        // LMS: S-type whose left neighbour is L-type (sa_is.c:155-162)
        uint32_t lmsmask = smask & ~((smask << 1) | prev_s);
        uint32_t valid = 0xFFFFu;
        if (p0 > n) valid = 0;
        else if (n - p0 < 15) valid = (2u << (uint32_t)(n - p0)) - 1u; // positions p0 .. n
        lmsmask &= valid;
        if (inside && ((w[0] | w[1] | w[2] | w[3]) & 0xF8F8F8F8u) == 0) { // all symbols below 8 (DNA: always)
            const uint32_t one = 0x01010101u;
            const uint32_t b0 = gather16(w[0] & one, w[1] & one, w[2] & one, w[3] & one, 0);
            const uint32_t b1 = gather16(w[0] & (one << 1), w[1] & (one << 1), w[2] & (one << 1), w[3] & (one << 1), 1);
            const uint32_t b2 = gather16(w[0] & (one << 2), w[1] & (one << 2), w[2] & (one << 2), w[3] & (one << 2), 2);
            // positions holding symbol A: every plane agrees with A's bit (a truth table with a single one)
#define SX_CLS_COUNT(A)                                                                                                \
    {                                                                                                                  \
        const uint32_t eq = __builtin_amdgcn_bitop3_b32(b0, b1, b2, 1u << ((((A) & 1) << 2) | ((A) & 2) | (((A) >> 2) & 1))) & 0xFFFFu; \
        cnt_all[A] += (uint32_t)__popc(eq);                                                                            \
        cnt_s[A] += (uint32_t)__popc(eq & smask);                                                                      \
        cnt_lms[A] += (uint32_t)__popc(eq & lmsmask);                                                                  \
    }
            SX_CLS_COUNT(0) SX_CLS_COUNT(1) SX_CLS_COUNT(2) SX_CLS_COUNT(3)
            SX_CLS_COUNT(4) SX_CLS_COUNT(5) SX_CLS_COUNT(6) SX_CLS_COUNT(7)
#undef SX_CLS_COUNT
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((valid >> i) & 1u) {
                    const uint32_t ch = (p0 + i == n) ? 0u : (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                    atomicAdd(&h[0][ch], 1u);
                    if (!((smask >> i) & 1u)) atomicAdd(&h[1][ch], 1u);
                    if ((lmsmask >> i) & 1u) atomicAdd(&h[2][ch], 1u);
                }
            }
        }
        lmsbits[(uint64_t)tile * kBlock + t] = (uint16_t)lmsmask;
        const uint32_t cnt = (uint32_t)__popc(lmsmask);
        const uint32_t lastp1 = lmsmask ? (uint32_t)(p0 + (31 - __clz(lmsmask))) + 1u : 0u;
        // tile totals: only the sums are needed, so one wave reduction each and a single barrier
        uint32_t wsum = cnt, wmax = lastp1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            wsum += __shfl_xor(wsum, o, kWave);
            const uint32_t other = __shfl_xor(wmax, o, kWave);
            wmax = other > wmax ? other : wmax;
        }
        if (lane_id() == 0) {
            lds[wave_id()] = wsum;
            lds[kWavesPerBlock + wave_id()] = wmax;
        }
        __syncthreads();
        if (t == 0) {
            uint32_t tot_cnt = 0, tot_last = 0;
#pragma unroll
            for (int ww = 0; ww < kWavesPerBlock; ++ww) {
                tot_cnt += lds[ww];
                tot_last = lds[kWavesPerBlock + ww] > tot_last ? lds[kWavesPerBlock + ww] : tot_last;
            }
            tile_lms[tile] = tot_cnt;
            tile_last[tile] = tot_last;
        }
        __syncthreads(); // lds[] and last_s[] are rewritten by the next tile
    }
    // the register counts: over the wave, then one LDS add per wave and symbol (L-type = all - S-type)
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        uint32_t x = cnt_all[a], y = cnt_s[a], z = cnt_lms[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            x += __shfl_xor(x, o, kWave);
            y += __shfl_xor(y, o, kWave);
            z += __shfl_xor(z, o, kWave);
        }
        if (lane_id() == 0) {
            if (x) atomicAdd(&h[0][a], x);
            if (x - y) atomicAdd(&h[1][a], x - y);
            if (z) atomicAdd(&h[2][a], z);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t v = h[k][t];
        if (v) atomicAdd(&g_hist[k * 256 + t], v);
    }
}

#ifndef SX_CLS_GRID
#define SX_CLS_GRID 4096u // workgroups of cls_types_kernel (1 GiB of DNA: 1024 0.79 ms, 2048 0.75, 4096 0.655, 16384 0.66)
#endif
// ---- pass 4: sample flags ---------------------------------------------------------
__global__ __launch_bounds__(kBlock) void samp_flags_kernel(const uint16_t *__restrict__ lmsbits, uint64_t n,
                                                            const uint32_t *__restrict__ tile_prev, uint32_t W,
                                                            uint16_t *__restrict__ sampbits,
                                                            uint32_t *__restrict__ tile_samp)
{
    __shared__ uint32_t lds[kWavesPerBlock];
    const int t = (int)threadIdx.x;
    const uint64_t p0 = (uint64_t)blockIdx.x * kClsTile + (uint64_t)t * kClsPerThread;
    const uint32_t lmsmask = lmsbits[(uint64_t)blockIdx.x * kBlock + t];
    const uint32_t lastp1 = lmsmask ? (uint32_t)(p0 + (31 - __clz(lmsmask))) + 1u : 0u;
    uint32_t tot;
    uint32_t prev = block_exclusive_scan<OpMax>(lastp1, lds, tot);
    const uint32_t tp = tile_prev[blockIdx.x];
    prev = prev > tp ? prev : tp; // position + 1 of the nearest LMS to the left, 0 = none
    uint32_t smp = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint64_t pos = p0 + i;
        if (pos <= n) {
            if ((lmsmask >> i) & 1u) {
                prev = (uint32_t)pos + 1u;
                smp |= 1u << i;
            } else if (prev != 0 && pos < n && ((uint32_t)pos - (prev - 1u)) % W == 0) {
                smp |= 1u << i;
            }
        }
    }
    sampbits[(uint64_t)blockIdx.x * kBlock + t] = (uint16_t)smp;
    uint32_t tot_s;
    (void)block_exclusive_scan<OpAdd>((uint32_t)__popc(smp), lds, tot_s);
    if (t == 0) tile_samp[blockIdx.x] = tot_s;
}

// ---- pass 5: compaction of the sample positions -------------------------------------
__global__ __launch_bounds__(kBlock) void samp_write_kernel(const uint16_t *__restrict__ sampbits,
                                                            const uint16_t *__restrict__ lmsbits,
                                                            const uint32_t *__restrict__ tile_off,
                                                            uint32_t *__restrict__ pos_out,
                                                            uint8_t *__restrict__ islms_out)
{
    __shared__ uint32_t lds[kWavesPerBlock];
    const int t = (int)threadIdx.x;
    const uint64_t p0 = (uint64_t)blockIdx.x * kClsTile + (uint64_t)t * kClsPerThread;
    uint32_t smp = sampbits[(uint64_t)blockIdx.x * kBlock + t];
    const uint32_t lmsmask = lmsbits[(uint64_t)blockIdx.x * kBlock + t];
    uint32_t tot;
    uint32_t dst = block_exclusive_scan<OpAdd>((uint32_t)__popc(smp), lds, tot) + tile_off[blockIdx.x];
    while (smp) {
        const int i = __ffs(smp) - 1;
        smp &= smp - 1u;
        pos_out[dst] = (uint32_t)(p0 + i);
        islms_out[dst] = (uint8_t)((lmsmask >> i) & 1u);
        ++dst;
    }
}

// ---- piece keys ---------------------------------------------------------------------
// Piece k covers text[pos[k] .. pos[k+1]] (both ends included, <= slots symbols).
// Key, most significant first: the symbols (bits each) padded with all-ones,
// then (slots - length) so that of two pieces with equal padded symbols the
// longer is smaller, then 1 if the piece ends at an LMS position, 0 if it ends
// at a cut.  The sentinel piece gets key 0.
__global__ __launch_bounds__(kBlock) void piece_keys_kernel(const uint8_t *__restrict__ T,
                                                            const uint32_t *__restrict__ pos,
                                                            const uint8_t *__restrict__ is_lms, uint64_t M,
                                                            uint32_t bits, uint32_t slots, uint32_t lenbits,
                                                            uint64_t *__restrict__ keys,
                                                            uint32_t *__restrict__ vals)
{
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= M) return;
    uint64_t key = 0;
    if (k + 1 < M) {
        const uint32_t i = pos[k], e = pos[k + 1];
        const uint32_t len = e - i + 1u;
        const uint64_t ones = (1ull << bits) - 1ull;
        uint64_t acc = 0;
        for (uint32_t s = 0; s < slots; ++s) {
            const uint64_t code = s < len ? (uint64_t)T[(uint64_t)i + s] : ones;
            acc = (acc << bits) | code;
        }
        acc = (acc << lenbits) | (uint64_t)(slots - len);
        acc = (acc << 1) | (uint64_t)is_lms[k + 1];
        key = acc << (64u - (slots * bits + lenbits + 1u));
    }
    keys[k] = key;
    vals[k] = (uint32_t)k;
}

__global__ __launch_bounds__(kBlock) void expand_bits_kernel(const uint16_t *__restrict__ bits, uint64_t N,
                                                             uint8_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < N) out[i] = (uint8_t)((bits[i >> 4] >> (i & 15)) & 1u);
}

} // namespace sx

using namespace sx;

size_t sx_text_scratch_bytes(uint64_t n)
{
    const uint64_t N = n + 1;
    const uint64_t ntiles = (N + kClsTile - 1) / kClsTile;
    size_t b = 0;
    b += ntiles * kClsTile + 256;         // padded text copy
    b += 2 * (ntiles * kBlock * 2 + 256); // lmsbits, sampbits
    b += 5 * ntiles * 4 + 256;            // tile_u32
    b += ntiles + 256;                    // tile_first
    b += 3 * 256 * 4 + 256 + 256;         // hist, scalars
    return b + 4096;
}

// symbol counts alone (no types): enough to decide whether a wide-alphabet text takes the direct sort
namespace sx {
__global__ __launch_bounds__(kBlock) void symbol_hist_kernel(const uint8_t *__restrict__ T, uint64_t n,
                                                             uint32_t *__restrict__ g_hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q * 16 < n; q += (uint64_t)gridDim.x * kBlock) {
        // T is the context's padded copy: 16-byte aligned, readable past n
        const uint4 v = *reinterpret_cast<const uint4 *>(T + q * 16);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (q * 16 + k < n) atomicAdd(&h[(w[k >> 2] >> (8 * (k & 3))) & 0xFFu], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&g_hist[threadIdx.x], h[threadIdx.x]);
}
} // namespace sx

int sx_symbol_histogram(sx_ctx *ctx, const uint8_t *T, uint64_t n, uint32_t *d_scratch256, uint32_t h_out[256])
{
    SX_CHECK(hipMemsetAsync(d_scratch256, 0, 256 * sizeof(uint32_t), ctx->stream));
    uint32_t grid = sx_div_up(n, kBlock * 64);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    sx_launch(ctx, SX_KC_CLASSIFY, n, symbol_hist_kernel, dim3(grid), dim3(kBlock), T, n, d_scratch256);
    return sx_readback(ctx, d_scratch256, 256, h_out);
}

int sx_classify(sx_ctx *ctx, uint8_t *T, uint64_t n, sx_arena &arena, sx_text_info &ti, const uint8_t *src, uint32_t src_tiles)
{
    ti.T = T;
    ti.n = n;
    ti.N = n + 1;
    ti.ntiles = sx_div_up(ti.N, kClsTile);
    ti.lmsbits = arena.take<uint16_t>((size_t)ti.ntiles * kBlock);
    ti.sampbits = arena.take<uint16_t>((size_t)ti.ntiles * kBlock);
    ti.tile_u32 = arena.take<uint32_t>((size_t)ti.ntiles * 5);
    ti.tile_first = arena.take<uint8_t>(ti.ntiles);
    ti.d_hist = arena.take<uint32_t>(3 * 256 + 16); // (+ the count of open tiles)
    ti.d_scalar = arena.take<uint32_t>(16);
    if (!ti.lmsbits || !ti.sampbits || !ti.tile_u32 || !ti.tile_first || !ti.d_hist || !ti.d_scalar)
        return sx_fail_msg(ctx, SX_E_INTERNAL, "classify: arena too small");
    uint32_t *tile_lms = ti.tile_u32, *tile_last = ti.tile_u32 + ti.ntiles;
    SX_CHECK(hipMemsetAsync(ti.d_hist, 0, (3 * 256 + 1) * sizeof(uint32_t), ctx->stream));
    const dim3 grid(ti.ntiles), block(kBlock);
    sx_launch(ctx, SX_KC_CLASSIFY, ti.N / 4, cls_first_kernel, grid, dim3(kWave), (const uint8_t *)T, n, ti.tile_first, ti.d_hist + 3 * 256, src, src_tiles);
    sx_launch(ctx, SX_KC_CLASSIFY, ti.ntiles, cls_resolve_kernel, dim3(sx_div_up(ti.ntiles, kBlock * 16)), block, ti.tile_first,
              ti.ntiles);
    sx_launch(ctx, SX_KC_CLASSIFY, ti.N + ti.N / 8 + (src ? (uint64_t)src_tiles * kClsTile : 0), cls_types_kernel,
              dim3(ti.ntiles < SX_CLS_GRID ? ti.ntiles : SX_CLS_GRID), block, T, n, (const uint8_t *)ti.tile_first, ti.ntiles, ti.lmsbits, tile_lms,
              tile_last, ti.d_hist, src, src_tiles);
    // read the three histograms back: the host drives the bucket loop
    uint32_t h[3 * 256 + 1];
    SX_TRY(sx_readback(ctx, ti.d_hist, 3 * 256 + 1, h));
    ti.open_tiles = h[3 * 256];
    memcpy(ti.h_all, h, sizeof ti.h_all);
    memcpy(ti.h_l, h + 256, sizeof ti.h_l);
    memcpy(ti.h_lms, h + 512, sizeof ti.h_lms);
    ti.maxc = 0;
    ti.m = 0;
    uint64_t total = 0;
    for (int c = 0; c < 256; ++c) {
        if (ti.h_all[c]) ti.maxc = (uint32_t)c;
        ti.m += ti.h_lms[c];
        total += ti.h_all[c];
    }
    if (total != ti.N) return sx_fail_msg(ctx, SX_E_INTERNAL, "classify: histogram does not add up to n+1");
    if (ti.h_all[0] != 1) return sx_fail_msg(ctx, SX_E_ARG, "text contains the sentinel symbol 0");
    ti.M = 0;
    return 0;
}

int sx_sample_flags(sx_ctx *ctx, sx_text_info &ti, uint32_t W)
{
    uint32_t *tile_last = ti.tile_u32 + ti.ntiles, *tile_prev = ti.tile_u32 + 2 * (size_t)ti.ntiles,
             *tile_samp = ti.tile_u32 + 3 * (size_t)ti.ntiles, *tile_off = ti.tile_u32 + 4 * (size_t)ti.ntiles;
    // nearest LMS position (+1) in the tiles to the left of each tile
    SX_TRY((device_scan<OpMax>(ctx, ti.ntiles, InU32{tile_last}, OutExclusive{tile_prev}, nullptr)));
    sx_launch(ctx, SX_KC_SAMPLES, ti.N / 4, samp_flags_kernel, dim3(ti.ntiles), dim3(kBlock),
              (const uint16_t *)ti.lmsbits, ti.n, (const uint32_t *)tile_prev, W, ti.sampbits, tile_samp);
    SX_TRY((device_scan<OpAdd>(ctx, ti.ntiles, InU32{tile_samp}, OutExclusive{tile_off}, ti.d_scalar)));
    uint32_t M32 = 0;
    SX_TRY(sx_readback(ctx, ti.d_scalar, 1, &M32));
    ti.M = M32;
    if (ti.M < ti.m) return sx_fail_msg(ctx, SX_E_INTERNAL, "samples: fewer samples than LMS positions");
    return 0;
}

int sx_sample_write(sx_ctx *ctx, const sx_text_info &ti, uint32_t *pos, uint8_t *is_lms)
{
    const uint32_t *tile_off = ti.tile_u32 + 4 * (size_t)ti.ntiles;
    sx_launch(ctx, SX_KC_SAMPLES, ti.N / 4 + ti.M * 5, samp_write_kernel, dim3(ti.ntiles), dim3(kBlock),
              (const uint16_t *)ti.sampbits, (const uint16_t *)ti.lmsbits, tile_off, pos, is_lms);
    return 0;
}

int sx_piece_keys(sx_ctx *ctx, const sx_text_info &ti, const uint32_t *pos, const uint8_t *is_lms,
                  uint32_t bits, uint32_t slots, uint32_t lenbits, uint64_t *keys, uint32_t *vals)
{
    if (ti.M == 0) return 0;
    sx_launch(ctx, SX_KC_KEYS, ti.M * (4 + 1 + 12) + ti.N, piece_keys_kernel, dim3(sx_div_up(ti.M, kBlock)),
              dim3(kBlock), ti.T, pos, is_lms, ti.M, bits, slots, lenbits, keys, vals);
    return 0;
}

extern "C" int sx_prim_classify_dev(sx_ctx *ctx, const uint8_t *d_text, uint64_t n, uint8_t *d_lms_flags,
                                    uint32_t *d_hist_all, uint32_t *d_hist_l, uint32_t *d_hist_lms)
{
    if (!ctx) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_N, sx_text_scratch_bytes(n)));
    sx_arena arena;
    arena.base = (char *)ctx->slab[SX_SLAB_N].p;
    arena.cap = ctx->slab[SX_SLAB_N].cap;
    const uint64_t N = n + 1;
    const uint64_t padded = (uint64_t)sx_div_up(N, kClsTile) * kClsTile + 64;
    uint8_t *T = arena.take<uint8_t>(padded);
    if (!T) return sx_fail_msg(ctx, SX_E_INTERNAL, "arena");
    SX_CHECK(hipMemsetAsync(T, 0, padded, ctx->stream));
    if (n) SX_CHECK(hipMemcpyAsync(T, d_text, n, hipMemcpyDeviceToDevice, ctx->stream));
    sx_text_info ti;
    SX_TRY(sx_classify(ctx, T, n, arena, ti));
    sx_launch(ctx, SX_KC_MISC, 0, expand_bits_kernel, dim3(sx_div_up(N, kBlock)), dim3(kBlock),
              (const uint16_t *)ti.lmsbits, N, d_lms_flags);
    SX_CHECK(hipMemcpyAsync(d_hist_all, ti.d_hist, 256 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    SX_CHECK(hipMemcpyAsync(d_hist_l, ti.d_hist + 256, 256 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    SX_CHECK(hipMemcpyAsync(d_hist_lms, ti.d_hist + 512, 256 * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return sx_sync(ctx);
}
