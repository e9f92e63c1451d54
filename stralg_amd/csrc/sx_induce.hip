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

namespace sx {

constexpr int kIndItems = 8;
constexpr int kIndTile = kBlock * kIndItems;

enum { MODE_L_FROM_L = 0, MODE_L_FROM_LMS = 1, MODE_S_FROM_S = 2, MODE_S_FROM_L = 3 };

__device__ __forceinline__ bool induce_accept(uint32_t ch, uint32_t c, int mode)
{
    switch (mode) {
    case MODE_L_FROM_L: return ch >= c;
    case MODE_L_FROM_LMS: return true;
    case MODE_S_FROM_S: return ch <= c;
    default: return ch < c;
    }
}

// ---- symbol windows ---------------------------------------------------------------
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
template <class WT> __device__ __forceinline__ WT wnd_pop(WT w, const wnd_cfg &c)
{
    const WT cnt = w & (WT)15;
    return (((w >> kCntBits) >> c.B) << kCntBits) | (cnt - 1);
}
// window of position p, read from the text (p >= 1)
template <class WT>
__device__ __forceinline__ WT wnd_fill(const uint8_t *__restrict__ T, uint32_t p, const wnd_cfg &c)
{
    const uint32_t cnt = p < c.CW ? p : c.CW;
    // text[p-cnt .. p-1]: two aligned 16-byte loads, bytes picked with static indices
    uint64_t lo, hi;
    load_bytes16(T, (uint64_t)(p - cnt), lo, hi);
    WT acc = 0;
#pragma unroll
    for (uint32_t i = 0; i < 15; ++i) {
        if (i < cnt) {
            const uint64_t byte = ((i < 8 ? lo : hi) >> (8u * (i & 7u))) & 0xFFull;
            acc = (acc << c.B) | (WT)(byte - 1u); // ends with text[p-1] in the lowest field
        }
    }
    return (acc << kCntBits) | (WT)cnt;
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

// count accepted entries per destination bucket (no text access: the symbol is in the window)
template <class WT>
__global__ __launch_bounds__(kBlock) void induce_count_kernel(const uint32_t *__restrict__ srcP,
                                                              const WT *__restrict__ srcW, uint32_t len, int rev,
                                                              int mode, uint32_t c, wnd_cfg cfg,
                                                              uint32_t *__restrict__ hist, uint32_t ntiles,
                                                              uint32_t nkeys)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile0 = blockIdx.x * (uint32_t)kIndTile;
#pragma unroll
    for (int k = 0; k < kIndItems; ++k) {
        const uint32_t i = tile0 + (uint32_t)k * kBlock + threadIdx.x;
        if (i < len) {
            const uint32_t idx = rev ? len - 1u - i : i;
            const uint32_t p = srcP[idx];
            if (p != 0) {
                const uint32_t ch = wnd_first<WT>(srcW[idx], cfg);
                if (induce_accept(ch, c, mode)) atomicAdd(&h[ch], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < nkeys) hist[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// one workgroup per destination bucket: exclusive prefix over the tiles, cursor update
__global__ __launch_bounds__(kBlock) void induce_offsets_kernel(uint32_t *__restrict__ hist, uint32_t ntiles,
                                                                uint32_t *__restrict__ cursor,
                                                                uint32_t *__restrict__ base, int dir, uint32_t c,
                                                                uint32_t *__restrict__ ctl)
{
    __shared__ uint32_t lds[kWavesPerBlock];
    const uint32_t key = blockIdx.x;
    const uint32_t carry = block_scan_row_inplace(hist + (uint64_t)key * ntiles, ntiles, lds);
    if (threadIdx.x == 0) {
        const uint32_t cur = cursor[key];
        base[key] = cur;
        cursor[key] = dir > 0 ? cur + carry : cur - carry;
        if (key == c) ctl[0] = carry;
    }
}

// stable scatter of p-1 (with its popped window) to the bucket cursors
template <class WT, int BITS>
__global__ __launch_bounds__(kBlock) void induce_scatter_kernel(
    const uint32_t *__restrict__ srcP, const WT *__restrict__ srcW, uint32_t len, int rev, int mode, uint32_t c,
    wnd_cfg cfg, const uint8_t *__restrict__ T, const uint32_t *__restrict__ offs, uint32_t ntiles,
    const uint32_t *__restrict__ base, int dir, uint32_t *__restrict__ SA, WT *__restrict__ WN, uint32_t nkeys)
{
    __shared__ uint32_t wcount[kWavesPerBlock][256];
    __shared__ uint32_t gpos[256]; // first destination index of the tile for each bucket
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    for (int i = t; i < kWavesPerBlock * 256; i += kBlock) (&wcount[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile0 = blockIdx.x * (uint32_t)kIndTile;
    const uint32_t wave0 = tile0 + (uint32_t)w * (kWave * kIndItems);
    uint32_t val[kIndItems], dig[kIndItems], rnk[kIndItems];
    WT wnd[kIndItems];
    bool ok[kIndItems];
#pragma unroll
    for (int k = 0; k < kIndItems; ++k) {
        const uint32_t i = wave0 + (uint32_t)k * kWave + lane;
        ok[k] = false;
        dig[k] = 0;
        val[k] = 0;
        wnd[k] = 0;
        if (i < len) {
            const uint32_t idx = rev ? len - 1u - i : i;
            const uint32_t p = srcP[idx];
            if (p != 0) {
                const WT ww = srcW[idx];
                const uint32_t ch = wnd_first<WT>(ww, cfg);
                ok[k] = induce_accept(ch, c, mode);
                dig[k] = ch;
                val[k] = p - 1u;
                wnd[k] = wnd_pop<WT>(ww, cfg);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kIndItems; ++k) rnk[k] = wave_rank_step<BITS>(dig[k], ok[k], wcount[w]);
    __syncthreads();
    {
        const uint32_t d = (uint32_t)t;
        uint32_t s = 0;
#pragma unroll
        for (int ww = 0; ww < kWavesPerBlock; ++ww) {
            const uint32_t x = wcount[ww][d];
            wcount[ww][d] = s;
            s += x;
        }
        gpos[d] = d < nkeys ? offs[(uint64_t)d * ntiles + blockIdx.x] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kIndItems; ++k) {
        if (ok[k]) {
            const uint32_t d = dig[k];
            const uint32_t r = gpos[d] + wcount[w][d] + rnk[k];
            const uint32_t dst = dir > 0 ? base[d] + r : base[d] - 1u - r;
            const uint32_t j = val[k];
            WT nw = wnd[k];
            if (j != 0 && wnd_count<WT>(nw) == 0) nw = wnd_fill<WT>(T, j, cfg); // window ran dry: back to the text
            SA[dst] = j;
            WN[dst] = nw;
        }
    }
}

template <class WT>
__global__ __launch_bounds__(kBlock) void bwt_from_windows_kernel(const uint32_t *__restrict__ SA,
                                                                  const WT *__restrict__ WN, uint64_t N, wnd_cfg cfg,
                                                                  uint8_t *__restrict__ bwt)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < N) bwt[i] = SA[i] == 0 ? (uint8_t)0 : (uint8_t)wnd_first<WT>(WN[i], cfg);
}

template <class WT> __global__ void set_entry_kernel(uint32_t *SA, WT *WN, uint32_t p, const uint8_t *T, wnd_cfg cfg)
{
    SA[0] = p;
    WN[0] = p ? wnd_fill<WT>(T, p, cfg) : (WT)0;
}

} // namespace sx

using namespace sx;

size_t sx_induce_scratch_bytes(uint64_t N, uint32_t sigma)
{
    const uint64_t ntiles = (N + kIndTile - 1) / kIndTile;
    // windows for every SA slot (8 bytes worst case) + seed windows (N/2) + tile histograms
    return (size_t)N * 8 + 256 + (size_t)(N / 2 + 2) * 8 + 256 + (size_t)sigma * ntiles * 4 + 256 + 3 * 1024 + 4096;
}

namespace {
template <class WT> struct induce_state {
    sx_ctx *ctx;
    const uint8_t *T;
    uint32_t *SA;
    WT *WN;
    uint32_t *hist, *cursor, *base, *ctl;
    uint32_t nk;
    int small_alphabet;
    wnd_cfg cfg;
};

// one stable multi-way split; *added_c = number of entries appended to bucket c
template <class WT>
int induce_round(induce_state<WT> &st, const uint32_t *srcP, const WT *srcW, uint32_t len, int rev, int mode,
                 uint32_t c, int dir, uint32_t *added_c)
{
    sx_ctx *ctx = st.ctx;
    const uint32_t ntiles = sx_div_up(len, kIndTile);
    const uint64_t eb = 4 + sizeof(WT);
    sx_launch(ctx, SX_KC_INDUCE_GATHER, (uint64_t)len * eb, induce_count_kernel<WT>, dim3(ntiles), dim3(kBlock), srcP,
              srcW, len, rev, mode, c, st.cfg, st.hist, ntiles, st.nk);
    sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)ntiles * st.nk * 8, induce_offsets_kernel, dim3(st.nk), dim3(kBlock),
              st.hist, ntiles, st.cursor, st.base, dir, c, st.ctl);
    if (st.small_alphabet)
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, (uint64_t)len * eb * 2, induce_scatter_kernel<WT, 3>, dim3(ntiles),
                  dim3(kBlock), srcP, srcW, len, rev, mode, c, st.cfg, st.T, (const uint32_t *)st.hist, ntiles,
                  (const uint32_t *)st.base, dir, st.SA, st.WN, st.nk);
    else
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, (uint64_t)len * eb * 2, induce_scatter_kernel<WT, 8>, dim3(ntiles),
                  dim3(kBlock), srcP, srcW, len, rev, mode, c, st.cfg, st.T, (const uint32_t *)st.hist, ntiles,
                  (const uint32_t *)st.base, dir, st.SA, st.WN, st.nk);
    ctx->stats.induce_rounds++;
    if (added_c) SX_TRY(sx_readback(ctx, st.ctl, 1, added_c));
    return 0;
}

template <class WT>
int induce_typed(sx_ctx *ctx, const sx_text_info &ti, uint32_t sigma, const uint32_t *sorted_lms, uint32_t *SA,
                 uint8_t *bwt_out, sx_arena &arena, wnd_cfg cfg)
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
    const uint64_t max_tiles = (N + kIndTile - 1) / kIndTile;
    st.WN = arena.take<WT>(N);
    WT *seedW = arena.take<WT>(ti.m ? ti.m : 1);
    st.hist = arena.take<uint32_t>((size_t)nk * max_tiles);
    st.cursor = arena.take<uint32_t>(256);
    st.base = arena.take<uint32_t>(256);
    st.ctl = arena.take<uint32_t>(16);
    if (!st.WN || !seedW || !st.hist || !st.cursor || !st.base || !st.ctl)
        return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small");

    // bucket boundaries on the host (sa_is.c:176-201)
    uint32_t begin[257], lms_off[257];
    begin[0] = 0;
    lms_off[0] = 0;
    for (uint32_t c = 0; c < 256; ++c) {
        begin[c + 1] = begin[c] + ti.h_all[c];
        lms_off[c + 1] = lms_off[c] + ti.h_lms[c];
    }

    // windows of the sorted LMS suffixes: the only systematic text access of both passes
    sx_launch(ctx, SX_KC_INDUCE_GATHER, ti.m * (4 + sizeof(WT) + 16), fill_windows_kernel<WT>,
              dim3(sx_div_up(ti.m, kBlock)), dim3(kBlock), ti.T, sorted_lms, ti.m, cfg, seedW);
    // the sentinel suffix (sa_is.c:463: SA[0] = n)
    sx_launch(ctx, SX_KC_MISC, 0, set_entry_kernel<WT>, dim3(1), dim3(1), SA, st.WN, (uint32_t)ti.n, ti.T, cfg);

    // ---- L pass: buckets ascending, cursors at the bucket heads ------------------------
    SX_CHECK(hipMemcpyAsync(st.cursor, begin, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    SX_CHECK(hipStreamSynchronize(ctx->stream));
    for (uint32_t c = 0; c < nk; ++c) {
        if (ti.h_all[c] == 0) continue;
        uint32_t head_c = begin[c];
        if (c > 0) SX_TRY(sx_readback(ctx, st.cursor + c, 1, &head_c));
        uint32_t lo = begin[c], hi = head_c;
        while (hi > lo) {
            uint32_t added = 0;
            SX_TRY(induce_round<WT>(st, SA + lo, st.WN + lo, hi - lo, 0, MODE_L_FROM_L, c, +1, &added));
            lo = hi;
            hi += added;
        }
        if (hi - begin[c] != ti.h_l[c])
            return sx_fail_msg(ctx, SX_E_INTERNAL, "induce L: bucket did not receive its L-type count");
        if (ti.h_lms[c])
            SX_TRY(induce_round<WT>(st, sorted_lms + lms_off[c], seedW + lms_off[c], ti.h_lms[c], 0, MODE_L_FROM_LMS, c,
                                    +1, nullptr));
    }

    // ---- S pass: buckets descending, cursors at the bucket ends -------------------------
    SX_CHECK(hipMemcpyAsync(st.cursor, begin + 1, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    SX_CHECK(hipStreamSynchronize(ctx->stream));
    for (uint32_t cc = nk; cc-- > 0;) {
        const uint32_t c = cc;
        if (ti.h_all[c] == 0) continue;
        const uint32_t end_c = begin[c + 1];
        uint32_t tail_c = end_c;
        SX_TRY(sx_readback(ctx, st.cursor + c, 1, &tail_c));
        uint32_t lo = tail_c, hi = end_c;
        while (hi > lo) {
            uint32_t added = 0;
            SX_TRY(induce_round<WT>(st, SA + lo, st.WN + lo, hi - lo, 1, MODE_S_FROM_S, c, -1, &added));
            hi = lo;
            lo -= added;
        }
        const uint32_t n_s = ti.h_all[c] - ti.h_l[c];
        if (c > 0 && end_c - lo != n_s)
            return sx_fail_msg(ctx, SX_E_INTERNAL, "induce S: bucket did not receive its S-type count");
        if (ti.h_l[c])
            SX_TRY(induce_round<WT>(st, SA + begin[c], st.WN + begin[c], ti.h_l[c], 1, MODE_S_FROM_L, c, -1, nullptr));
    }

    // the windows now hold text[SA[i]-1] for every slot: the BWT for free (bwt.c:13-20)
    if (bwt_out)
        sx_launch(ctx, SX_KC_BWT_GATHER, N * (4 + sizeof(WT) + 1), bwt_from_windows_kernel<WT>,
                  dim3(sx_div_up(N, kBlock)), dim3(kBlock), (const uint32_t *)SA, (const WT *)st.WN, N, cfg, bwt_out);
    return 0;
}
} // namespace

int sx_induce(sx_ctx *ctx, const sx_text_info &ti, uint32_t sigma, const uint32_t *sorted_lms, uint32_t *SA,
              uint8_t *bwt_out, sx_arena &arena)
{
    if (ti.N > 0xFFFFFFFFull) return sx_fail_msg(ctx, SX_E_ARG, "induce: n exceeds 32-bit positions");
    wnd_cfg cfg;
    cfg.B = (uint32_t)sx_bitlen(ti.maxc > 0 ? ti.maxc - 1 : 0);
    if (cfg.B < 1) cfg.B = 1;
    cfg.mask = (1u << cfg.B) - 1u;
    if (cfg.B <= 4) {
        cfg.CW = (32 - kCntBits) / cfg.B;
        if (cfg.CW > 15) cfg.CW = 15;
        return induce_typed<uint32_t>(ctx, ti, sigma, sorted_lms, SA, bwt_out, arena, cfg);
    }
    cfg.CW = (64 - kCntBits) / cfg.B;
    if (cfg.CW > 15) cfg.CW = 15;
    return induce_typed<uint64_t>(ctx, ti, sigma, sorted_lms, SA, bwt_out, arena, cfg);
}
