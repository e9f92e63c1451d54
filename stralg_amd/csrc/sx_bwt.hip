// sx_bwt.hip -- BWT symbols, C table and O table (stralg/bwt.c:13-65).
//
//   bwt[i]  = SA[i] == 0 ? 0 : text[SA[i] - 1]                 bwt.c:13-20
//   C[a]    = #{symbols of text + sentinel that are < a}        bwt.c:35-45
//   O(a, i) = #{k < i : bwt[k] == a},  i = 0 .. N               bwt.c:47-65
//             stored position-major: o[i * sigma + a]           bwt.c:50-57
//
// The reference loops letter-outer / position-inner (sigma strided sweeps and
// sigma gathers of every bwt symbol).  Here each bwt symbol is gathered once
// (the only random access), tiles of rows get their starting counts from a
// per-symbol prefix over tile histograms, and every tile of O rows is
// assembled in LDS and leaves the CU as contiguous stores: the kernel is
// bound by the 4*sigma output bytes per position.
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_scan.hpp"
#include "sx_internal.hpp"
#include "sx_pager.hpp"

// workgroups of the wide O-table kernel, each looping over its tiles (sigma 21 at 1 GiB, 4.2 M tiles of 256 rows: 20.5 ms
// with a workgroup per tile, 17.6 with 65 536 workgroups, 18.1 with 16 384, 18.8 with 4096)
// ... and of the kernel for small alphabets (1 GiB of DNA, 1 M tiles of 1024 rows: 4.19 ms with a workgroup per tile,
// 3.92 with 262 144 workgroups, 3.87 with 65 536, 3.96 with 16 384)
#ifndef SX_SMALL_GRID
#define SX_SMALL_GRID 65536u
#endif
#ifndef SX_WIDE_GRID
#define SX_WIDE_GRID 65536u
#endif
namespace sx {

constexpr int kSmallSigma = 8;      // register-vector O kernel for sigma <= 8
// rows per thread: 8 for sigma <= 5 (DNA + sentinel: 2048-row tiles, 40 KiB of LDS), 4 up to sigma 8
template <int SIG> struct small_cfg {
    static constexpr int rows = 4;
    static constexpr int tile = kBlock * rows;
};
constexpr int kMaxSigmaO = 128;     // stralg/remap.h:14-18

// bwt symbols + per-tile symbol counts.  tile_rows rows per workgroup; row N
// (the last O row) has no symbol of its own.
__global__ __launch_bounds__(kBlock) void bwt_gather_kernel(const uint8_t *__restrict__ T,
                                                            const uint32_t *__restrict__ SA, uint64_t N,
                                                            uint32_t tile_rows, uint32_t sigma,
                                                            uint8_t *__restrict__ bwt,
                                                            uint32_t *__restrict__ tilehist, uint32_t ntiles)
{
    __shared__ uint32_t h[256];
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // (a grid of ntiles workgroups can exceed the launch limit)
        h[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t tile0 = (uint64_t)tile * tile_rows;
        for (uint32_t r = threadIdx.x; r < tile_rows; r += kBlock) {
            const uint64_t i = tile0 + r;
            if (i < N) {
                const uint32_t p = SA[i];
                // an entry outside [0, n] (a malformed sa) must not fault: count it as symbol 255
                const uint32_t b = p == 0 ? 0u : ((uint64_t)p < N ? (uint32_t)T[p - 1u] : 255u);
                bwt[i] = (uint8_t)b;
                atomicAdd(&h[b], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < sigma) tilehist[(uint64_t)threadIdx.x * ntiles + tile] = h[threadIdx.x];
    }
}

// per-tile symbol counts of a BWT that is already in memory (fused SA+BWT build)
__global__ __launch_bounds__(kBlock) void bwt_count_kernel(const uint8_t *__restrict__ bwt, uint64_t N,
                                                           uint32_t tile_rows, uint32_t sigma,
                                                           uint32_t *__restrict__ tilehist, uint32_t ntiles)
{
    __shared__ uint32_t h[256];
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        h[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t tile0 = (uint64_t)tile * tile_rows;
        for (uint32_t r = threadIdx.x; r < tile_rows; r += kBlock) {
            const uint64_t i = tile0 + r;
            if (i < N) atomicAdd(&h[bwt[i]], 1u);
        }
        __syncthreads();
        if (threadIdx.x < sigma) tilehist[(uint64_t)threadIdx.x * ntiles + tile] = h[threadIdx.x];
    }
}

// The same for tiles of 1024 rows (sigma <= 8): a wave per tile, 16 symbols per lane in one load, counts as popcounts
// of symbol masks gathered by dot products (sx_device.hpp: gather16), one reduction over the wave.  (With 8 symbols
// per thread in packed byte counters the launch ran at 1.8 TB/s of its one byte per row.)
__global__ __launch_bounds__(kBlock) void bwt_count_wave_kernel(const uint8_t *__restrict__ bwt, uint64_t N,
                                                                uint32_t sigma, uint32_t *__restrict__ tilehist,
                                                                uint32_t ntiles)
{
    const int lane = lane_id();
    // (whole waves; a wave walks over its tiles: 65 536 workgroups instead of one per four tiles)
    for (uint32_t tile = blockIdx.x * kWavesPerBlock + wave_id(); tile < ntiles; tile += gridDim.x * kWavesPerBlock) {
    const uint64_t r0 = (uint64_t)tile * 1024u + (uint64_t)lane * 16u;
    uint32_t S[4] = {0, 0, 0, 0};
    uint32_t inside = 0xFFFFu; // rows below N
    if (r0 + 16u <= N && ((uintptr_t)bwt & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(bwt + r0);
        S[0] = v.x, S[1] = v.y, S[2] = v.z, S[3] = v.w;
    } else {
        inside = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (r0 + e < N) {
                S[e >> 2] |= (uint32_t)bwt[r0 + e] << (8 * (e & 3));
                inside |= 1u << e;
            }
    }
    // (a symbol >= 8 would be taken for its low three bits here; the totals check of the caller does not see that,
    // so such bytes are counted as nothing: the sum then falls short of N and the caller reports the bad symbol)
    const uint32_t high = (S[0] | S[1] | S[2] | S[3]) & 0xF8F8F8F8u;
    if (high) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if ((S[e >> 2] >> (8 * (e & 3))) & 0xF8u) inside &= ~(1u << e);
    }
    const uint32_t one = 0x01010101u;
    const uint32_t b0 = gather16(S[0] & one, S[1] & one, S[2] & one, S[3] & one, 0);
    const uint32_t b1 = gather16(S[0] & (one << 1), S[1] & (one << 1), S[2] & (one << 1), S[3] & (one << 1), 1);
    const uint32_t b2 = gather16(S[0] & (one << 2), S[1] & (one << 2), S[2] & (one << 2), S[3] & (one << 2), 2);
    uint32_t n_of[8];
#define SX_BWT_COUNT(A) n_of[A] = (uint32_t)__popc(__builtin_amdgcn_bitop3_b32(b0, b1, b2, 1u << ((((A) & 1) << 2) | ((A) & 2) | (((A) >> 2) & 1))) & inside);
    SX_BWT_COUNT(0) SX_BWT_COUNT(1) SX_BWT_COUNT(2) SX_BWT_COUNT(3) SX_BWT_COUNT(4) SX_BWT_COUNT(5) SX_BWT_COUNT(6) SX_BWT_COUNT(7)
#undef SX_BWT_COUNT
    uint64_t even = (uint64_t)n_of[0] | (uint64_t)n_of[2] << 16 | (uint64_t)n_of[4] << 32 | (uint64_t)n_of[6] << 48; // 16-bit fields
    uint64_t odd = (uint64_t)n_of[1] | (uint64_t)n_of[3] << 16 | (uint64_t)n_of[5] << 32 | (uint64_t)n_of[7] << 48;
    even = wave_total_packed(even);
    odd = wave_total_packed(odd);
    if ((uint32_t)lane < sigma && lane < 8)
        tilehist[(uint64_t)lane * ntiles + tile] = (uint32_t)(((lane & 1) ? odd : even) >> (16 * (lane >> 1))) & 0xFFFFu;
    }
}

// The tile counts [sigma][ntiles] are scanned as one flat array (device_scan); the prefix of
// (symbol a, tile t) inside its row is flat[a*ntiles + t] - flat[a*ntiles], and the symbol
// totals -- hence the C table (bwt.c:35-45) -- are differences of the row starts.
__global__ void c_table_kernel(const uint32_t *__restrict__ flat, const uint32_t *__restrict__ grand_total,
                               uint32_t ntiles, uint32_t sigma, uint32_t *__restrict__ totals,
                               uint32_t *__restrict__ c_out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t a = 0; a < sigma; ++a) {
            const uint32_t start = flat[(uint64_t)a * ntiles];
            const uint32_t next = a + 1 < sigma ? flat[(uint64_t)(a + 1) * ntiles] : *grand_total;
            totals[a] = next - start;
            c_out[a] = acc;
            acc += next - start;
        }
    }
}

// copy a tile of finished O rows from LDS to its contiguous place in the table, 16 bytes a lane
__device__ __forceinline__ void store_rows(const uint32_t *__restrict__ rows, uint32_t *__restrict__ dst,
                                           uint32_t nwords)
{
    const uint32_t nvec = nwords >> 2;
    const uint4 *src4 = reinterpret_cast<const uint4 *>(rows);
    uint4 *dst4 = reinterpret_cast<uint4 *>(dst);
    // written once, never read by this kernel: streaming stores keep the rows out of L2's way
    for (uint32_t i = threadIdx.x; i < nvec; i += kBlock) stream_store16(dst4 + i, src4[i]);
    for (uint32_t i = (nvec << 2) + threadIdx.x; i < nwords; i += kBlock) dst[i] = rows[i];
}

// O rows for sigma <= SIG <= 8: every thread owns `rows` consecutive rows and keeps the
// running counts of all symbols in registers.
template <int SIG>
__global__ __launch_bounds__(kBlock) void otable_small_kernel(const uint8_t *__restrict__ bwt, uint64_t N,
                                                              uint32_t sigma,
                                                              const uint32_t *__restrict__ tilepre,
                                                              uint32_t ntiles, uint32_t *__restrict__ o_out)
{
    __shared__ uint64_t lds[2 * kWavesPerBlock];
    constexpr int kRows = small_cfg<SIG>::rows, kTile = small_cfg<SIG>::tile;
    __shared__ __attribute__((aligned(16))) uint32_t rows[kTile * SIG];
    const int t = (int)threadIdx.x;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { // SX_SMALL_GRID
    const uint64_t tile0 = (uint64_t)tile * kTile;
    const uint64_t r0 = tile0 + (uint64_t)t * kRows;
    // the tile's starting counts: asked for first, needed only after the block scan
    uint32_t pre[SIG];
#pragma unroll
    for (int a = 0; a < SIG; ++a)
        pre[a] = (uint32_t)a < sigma ? tilepre[(uint64_t)a * ntiles + tile] - tilepre[(uint64_t)a * ntiles] : 0u;
    uint32_t sym[kRows];
    // per-thread symbol counts as 16-bit fields of two u64 (symbols 0-3, 4-7): a tile holds at most
    // 1024 symbols, so one 64-bit block scan replaces four 32-bit ones
    uint64_t pk[2] = {0, 0};
    if (r0 + kRows <= N && ((uintptr_t)bwt & 7u) == 0) { // the thread's symbols in one load (kRows is 4 or 8)
        uint64_t word;
        if (kRows == 8) word = *reinterpret_cast<const uint64_t *>(bwt + r0);
        else word = *reinterpret_cast<const uint32_t *>(bwt + r0);
#pragma unroll
        for (int k = 0; k < kRows; ++k) sym[k] = (uint32_t)(word >> (8 * k)) & 0xFFu;
    } else {
#pragma unroll
        for (int k = 0; k < kRows; ++k) sym[k] = r0 + k < N ? (uint32_t)bwt[r0 + k] : 0xFFu;
    }
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const uint64_t one = 1ull << (16u * (sym[k] & 3u));
        if (sym[k] < 4u) pk[0] += one; // static indices: the pair stays in registers
        else if (sym[k] < (uint32_t)SIG) pk[1] += one;
    }
    uint32_t run[SIG];
    {
        uint64_t ex[2] = {pk[0], pk[1]};
        if (SIG > 4) {
            block_exclusive_sum64x2(ex[0], ex[1], lds); // both words behind one pair of barriers
        } else {
            uint64_t tot;
            ex[0] = block_exclusive_sum64(pk[0], lds, tot);
        }
#pragma unroll
        for (int a = 0; a < SIG; ++a) run[a] = (uint32_t)((ex[a >> 2] >> (16 * (a & 3))) & 0xFFFFull) + pre[a];
    }
    if (sigma == (uint32_t)SIG && (kRows * SIG) % 4 == 0) {
        // the thread's rows are kRows * SIG consecutive words of the tile: build them in
        // registers and hand them to LDS 16 bytes at a time (scalar stores at a 40-word lane
        // stride would hit 8-way bank conflicts; storing straight to memory from here was
        // measured slower: 9.2 vs 6.0 ms at 1 GiB)
        uint32_t out[kRows * SIG];
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
#pragma unroll
            for (int a = 0; a < SIG; ++a) {
                out[k * SIG + a] = run[a];
                run[a] += sym[k] == (uint32_t)a ? 1u : 0u;
            }
        }
        uint4 *dst4 = reinterpret_cast<uint4 *>(rows + (uint32_t)t * (kRows * SIG));
#pragma unroll
        for (int q = 0; q < kRows * SIG / 4; ++q) {
            uint4 v;
            v.x = out[4 * q];
            v.y = out[4 * q + 1];
            v.z = out[4 * q + 2];
            v.w = out[4 * q + 3];
            dst4[q] = v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kRows; ++k) {
            const uint32_t lr = (uint32_t)t * kRows + k;
#pragma unroll
            for (int a = 0; a < SIG; ++a) {
                if ((uint32_t)a < sigma) rows[lr * sigma + a] = run[a];
                run[a] += sym[k] == (uint32_t)a ? 1u : 0u;
            }
        }
    }
    __syncthreads();
    // rows tile0 .. min(tile0 + tile, N + 1) leave as one contiguous block
    const uint64_t rows_left = N + 1 - tile0;
    const uint32_t nrows = rows_left < (uint64_t)kTile ? (uint32_t)rows_left : (uint32_t)kTile;
    const uint32_t nwords = nrows * sigma;
    store_rows(rows, o_out + tile0 * sigma, nwords);
    __syncthreads(); // rows[] is reused by the next tile
    }
}

// O rows for 8 < sigma <= 128.  A tile is RG groups of 64 rows (RG = 4, 2, 1 for sigma <= 32, 64, 128: about
// 32 KiB of rows in LDS); wave w takes row group w % RG and the columns a with a % (4 / RG) == w / RG.  With one
// row per lane, the running count of column a at the lane's row is a ballot away:
//   O[r][a] = (count before the tile) + (count in the row groups above) + popcount(ballot(sym == a) & lanes below)
// The LDS row stride is made odd so that 64 lanes writing one column do not share a bank.
__device__ __forceinline__ uint32_t wide_row_groups(uint32_t sigma) { return sigma <= 32 ? 4u : (sigma <= 64 ? 2u : 1u); }

__global__ __launch_bounds__(kBlock) void otable_wide_kernel(const uint8_t *__restrict__ bwt, uint64_t N,
                                                             uint32_t sigma,
                                                             const uint32_t *__restrict__ tilepre,
                                                             uint32_t ntiles, uint32_t *__restrict__ o_out)
{
    constexpr uint32_t kRowWords = 4 * 64 * 33; // the largest of 4*64*33, 2*64*65 and 64*129 words: every (RG, sigma) pair fits
    __shared__ __attribute__((aligned(16))) uint32_t rows[kRowWords];
    __shared__ uint32_t gtot[4][kMaxSigmaO]; // symbol counts of every row group
    __shared__ uint32_t pre[kMaxSigmaO];     // symbol counts before the tile
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t RG = wide_row_groups(sigma), CP = kWavesPerBlock / RG;
    const uint32_t rg = (uint32_t)w % RG, cp = (uint32_t)w / RG;
    const uint32_t tile_rows = RG * kWave;
    const uint32_t stride = sigma | 1u;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t tile0 = (uint64_t)tile * tile_rows;
        // row r of the table counts the symbols before position r: the lane's own symbol is not in its row
        const uint64_t i = tile0 + (uint64_t)rg * kWave + lane;
        const uint32_t sym = i < N ? (uint32_t)bwt[i] : 0xFFFFu;
        if ((uint32_t)t < sigma) pre[t] = tilepre[(uint64_t)t * ntiles + tile] - tilepre[(uint64_t)t * ntiles];
        // symbol counts of every row group: one LDS add per lane (a ballot per column here would double the
        // kernel's vector work; with wide alphabets the adds spread over many words)
        for (uint32_t k = (uint32_t)t; k < RG * kMaxSigmaO; k += kBlock) (&gtot[0][0])[k] = 0;
        __syncthreads();
        if (cp == 0 && sym < sigma) atomicAdd(&gtot[rg][sym], 1u); // (one wave per row group does the counting)
        __syncthreads();
        // counts -> the count every row group starts from (one thread per column): a column step then reads one word
        if ((uint32_t)t < sigma) {
            uint32_t run = pre[t];
            for (uint32_t g = 0; g < RG; ++g) {
                const uint32_t c = gtot[g][t];
                gtot[g][t] = run;
                run += c;
            }
        }
        __syncthreads();
        for (uint32_t a = cp; a < sigma; a += CP) { // (four columns a step with their reads and ballots interleaved: no faster)
            const uint64_t m = __ballot(sym == a ? 1 : 0);
            rows[(rg * kWave + (uint32_t)lane) * stride + a] = gtot[rg][a] + (uint32_t)__popcll(m & lanemask_lt());
        }
        __syncthreads();
        const uint64_t rows_left = N + 1 - tile0;
        const uint32_t nrows = rows_left < (uint64_t)tile_rows ? (uint32_t)rows_left : tile_rows;
        uint32_t *dst = o_out + tile0 * sigma;
        if (stride == sigma) {
            store_rows(rows, dst, nrows * sigma);
        } else { // skip the pad word of every row
            const uint32_t nwords = nrows * sigma;
            uint32_t r = (uint32_t)t / sigma, a = (uint32_t)t % sigma;
            const uint32_t dr = kBlock / sigma, da = kBlock % sigma;
            for (uint32_t idx = (uint32_t)t; idx < nwords; idx += kBlock) {
                dst[idx] = rows[r * stride + a];
                r += dr, a += da;
                if (a >= sigma) a -= sigma, ++r;
            }
        }
        __syncthreads(); // rows[] and gtot[] are reused by the next tile
    }
}

} // namespace sx

using namespace sx;

// tables from (text, sa) or from a ready BWT (d_bwt_in != NULL)
static int bwt_tables_dev(sx_ctx *ctx, const uint8_t *d_text, const uint32_t *d_sa, const uint8_t *d_bwt_in,
                          uint64_t N, uint32_t sigma, uint32_t *d_c, uint32_t *d_o, uint8_t *d_bwt_out)
{
    if (N == 0 || N > 0xFFFFFFFFull) return sx_fail_msg(ctx, SX_E_ARG, "N must be in [1, 2^32 - 1]");
    if (sigma < 1 || sigma > 256) return sx_fail_msg(ctx, SX_E_ARG, "sigma must be in [1, 256]");
    if (d_o && sigma > kMaxSigmaO)
        return sx_fail_msg(ctx, SX_E_ARG, "the O table is defined for sigma <= 128 (stralg/remap.h:14-18)");
    const bool small = sigma <= kSmallSigma;
    const uint32_t tile_rows =
        small ? (uint32_t)small_cfg<8>::tile : (sigma <= 32 ? 256u : (sigma <= 64 ? 128u : 64u));
    const uint32_t ntiles = sx_div_up(N + 1, tile_rows);
    const uint32_t loop_grid = ntiles < (1u << 22) ? ntiles : (1u << 22); // kernels that loop over their tiles
    const size_t need = (size_t)N + 256 + (size_t)sigma * ntiles * 4 + 256 + 1024 + 4096;
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_BWT, need));
    sx_arena ar;
    ar.base = (char *)ctx->slab[SX_SLAB_BWT].p;
    ar.cap = ctx->slab[SX_SLAB_BWT].cap;
    uint8_t *bwt_own = d_bwt_in ? nullptr : (d_bwt_out ? d_bwt_out : ar.take<uint8_t>(N));
    const uint8_t *bwt = d_bwt_in ? d_bwt_in : bwt_own;
    uint32_t *tilehist = ar.take<uint32_t>((size_t)sigma * ntiles);
    uint32_t *totals = ar.take<uint32_t>(256 + 16); // per-symbol totals, then the grand total
    if (!bwt || !tilehist || !totals) return sx_fail_msg(ctx, SX_E_INTERNAL, "bwt: arena too small");

    static_assert(small_cfg<5>::tile == 1024 && small_cfg<8>::tile == 1024, "bwt_count_wave: a wave per 1024-row tile");
    if (d_bwt_in && small)
        sx_launch(ctx, SX_KC_BWT_GATHER, N, bwt_count_wave_kernel,
                  dim3(sx_div_up(ntiles, kWavesPerBlock) < SX_SMALL_GRID ? sx_div_up(ntiles, kWavesPerBlock) : SX_SMALL_GRID), dim3(kBlock), d_bwt_in,
                  N, sigma, tilehist, ntiles);
    else if (d_bwt_in)
        sx_launch(ctx, SX_KC_BWT_GATHER, N, bwt_count_kernel, dim3(loop_grid), dim3(kBlock), d_bwt_in, N, tile_rows, sigma,
                  tilehist, ntiles);
    else
        sx_launch(ctx, SX_KC_BWT_GATHER, N * 6, bwt_gather_kernel, dim3(loop_grid), dim3(kBlock), d_text, d_sa, N,
                  tile_rows, sigma, bwt_own, tilehist, ntiles);
    const uint64_t flat_n = (uint64_t)sigma * ntiles;
    SX_TRY((device_scan<OpAdd>(ctx, flat_n, InU32{tilehist}, OutExclusive{tilehist}, totals + 256, SX_KC_SCAN,
                               flat_n * 12)));
    sx_launch(ctx, SX_KC_MISC, 0, c_table_kernel, dim3(1), dim3(64), (const uint32_t *)tilehist,
              (const uint32_t *)(totals + 256), ntiles, sigma, totals, d_c);
    // every bwt symbol must be < sigma (the O kernels index rows by symbol)
    uint32_t h_tot[256];
    SX_TRY(sx_readback(ctx, totals, sigma, h_tot));
    uint64_t sum = 0;
    for (uint32_t a = 0; a < sigma; ++a) sum += h_tot[a];
    if (sum != N) return sx_fail_msg(ctx, SX_E_ARG, "text holds a symbol >= sigma, or sa is not over this text");
    if (d_o) {
        const uint64_t out_bytes = (N + 1) * (uint64_t)sigma * 4 + N;
        // small alphabets: the kernel whose row length is sigma itself assembles a thread's rows in registers and
        // hands them to LDS 16 bytes at a time (sigma = 6 through the 8-column form: 7.3 instead of 5.3 ms at 1 GiB)
#define SX_OTABLE_SMALL(SIG)                                                                                           \
    sx_launch(ctx, SX_KC_OTABLE, out_bytes, otable_small_kernel<SIG>, dim3(ntiles < SX_SMALL_GRID ? ntiles : SX_SMALL_GRID), dim3(kBlock), bwt, N, sigma,       \
              (const uint32_t *)tilehist, ntiles, d_o)
        if (small) {
            switch (sigma) {
            case 3: SX_OTABLE_SMALL(3); break;
            case 4: SX_OTABLE_SMALL(4); break;
            case 5: SX_OTABLE_SMALL(5); break;
            case 6: SX_OTABLE_SMALL(6); break;
            case 7: SX_OTABLE_SMALL(7); break;
            case 8: SX_OTABLE_SMALL(8); break;
            default: SX_OTABLE_SMALL(5); break; // sigma 1, 2
            }
        } else
#undef SX_OTABLE_SMALL
            sx_launch(ctx, SX_KC_OTABLE, out_bytes, otable_wide_kernel, dim3(loop_grid < SX_WIDE_GRID ? loop_grid : SX_WIDE_GRID), dim3(kBlock), bwt, N, sigma,
                      (const uint32_t *)tilehist, ntiles, d_o);
    }
    return 0;
}

int sx_tables_from_bwt_impl(sx_ctx *ctx, const uint8_t *d_bwt, uint64_t N, uint32_t sigma, uint32_t *d_c, uint32_t *d_o)
{
    return bwt_tables_dev(ctx, nullptr, nullptr, d_bwt, N, sigma, d_c, d_o, nullptr);
}

extern "C" {

int sx_bwt_tables_dev(sx_ctx *ctx, const uint8_t *d_text, const uint32_t *d_sa, uint64_t N, uint32_t sigma,
                      uint32_t *d_c_out, uint32_t *d_o_out, uint8_t *d_bwt_out)
{
    if (!ctx || !d_sa || !d_c_out || (N > 1 && !d_text)) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    SX_TRY(bwt_tables_dev(ctx, d_text, d_sa, nullptr, N, sigma, d_c_out, d_o_out, d_bwt_out));
    return sx_sync(ctx);
}

int sx_bwt_tables_from_bwt_dev(sx_ctx *ctx, const uint8_t *d_bwt, uint64_t N, uint32_t sigma, uint32_t *d_c_out,
                               uint32_t *d_o_out)
{
    if (!ctx || !d_bwt || !d_c_out) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    SX_TRY(bwt_tables_dev(ctx, nullptr, nullptr, d_bwt, N, sigma, d_c_out, d_o_out, nullptr));
    return sx_sync(ctx);
}

int sx_bwt_tables(sx_ctx *ctx, const uint8_t *text, const uint32_t *sa, uint64_t N, uint32_t sigma,
                  uint32_t *c_out, uint32_t *o_out)
{
    if (!ctx || !sa || !c_out || (N > 1 && !text)) return SX_E_ARG;
    if (N == 0 || N > 0xFFFFFFFFull) return sx_fail_msg(ctx, SX_E_ARG, "N must be in [1, 2^32 - 1]");
    if (sigma < 1 || sigma > 256) return sx_fail_msg(ctx, SX_E_ARG, "sigma must be in [1, 256]");
    SX_CHECK(hipSetDevice(ctx->device));
    const uint64_t n = N - 1;
    const size_t text_b = (n + 255) & ~(size_t)255, sa_b = (N * 4 + 255) & ~(size_t)255;
    const size_t o_b = o_out ? (((N + 1) * (size_t)sigma * 4 + 255) & ~(size_t)255) : 0;
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_IO, text_b + sa_b + o_b + 1024 + 1024));
    char *base = (char *)ctx->slab[SX_SLAB_IO].p;
    uint8_t *d_text = (uint8_t *)base;
    uint32_t *d_sa = (uint32_t *)(base + text_b + 256);
    uint32_t *d_c = (uint32_t *)(base + text_b + 256 + sa_b);
    uint32_t *d_o = o_out ? (uint32_t *)(base + text_b + 256 + sa_b + 1024) : nullptr;
    sx_host_pager pager; // the caller's O table is paged in by host threads meanwhile (sx_pager.hpp)
    const size_t o_bytes = (N + 1) * (size_t)sigma * 4;
    const size_t c_o = o_out ? pager.add(o_out, o_bytes) : 0;
    pager.start();
    if (n) SX_CHECK(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, ctx->stream));
    SX_CHECK(hipMemcpyAsync(d_sa, sa, N * 4, hipMemcpyHostToDevice, ctx->stream));
    SX_TRY(bwt_tables_dev(ctx, d_text, d_sa, nullptr, N, sigma, d_c, d_o, nullptr));
    SX_CHECK(hipMemcpyAsync(c_out, d_c, (size_t)sigma * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (o_out) SX_TRY(pager.download(ctx, c_o, o_out, d_o, o_bytes));
    return sx_sync(ctx);
}

} // extern "C"
