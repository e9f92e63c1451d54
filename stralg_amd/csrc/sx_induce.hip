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
// Cost: one random byte gather per scanned entry (the HBM-latency-bound part)
// and a 4-byte write per induced entry into <= sigma sequential streams.
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

// logical item i of a round -> entry of the source range (reversed for the S pass)
__device__ __forceinline__ uint32_t round_item(const uint32_t *__restrict__ src, uint32_t len, bool rev,
                                               uint32_t i)
{
    return src[rev ? len - 1u - i : i];
}

// gather text[p-1], remember it, count accepted entries per destination bucket
__global__ __launch_bounds__(kBlock) void induce_gather_kernel(const uint32_t *__restrict__ src, uint32_t len,
                                                               int rev, const uint8_t *__restrict__ T, int mode,
                                                               uint32_t c, uint8_t *__restrict__ tmpch,
                                                               uint32_t *__restrict__ hist, uint32_t ntiles,
                                                               uint32_t nkeys)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    // a symbol the accept test of this mode rejects (p == 0 has no predecessor)
    const uint32_t reject = mode == MODE_L_FROM_L ? 0u : 255u;
    const uint32_t tile0 = blockIdx.x * (uint32_t)kIndTile;
#pragma unroll
    for (int k = 0; k < kIndItems; ++k) {
        const uint32_t i = tile0 + (uint32_t)k * kBlock + threadIdx.x;
        if (i < len) {
            const uint32_t p = round_item(src, len, rev != 0, i);
            uint32_t ch = reject;
            if (p != 0) ch = T[p - 1u];
            const bool ok = p != 0 && induce_accept(ch, c, mode);
            tmpch[i] = (uint8_t)(ok ? ch : reject);
            if (ok) atomicAdd(&h[ch], 1u);
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

template <int BITS>
__global__ __launch_bounds__(kBlock) void induce_scatter_kernel(const uint32_t *__restrict__ src, uint32_t len,
                                                                int rev, const uint8_t *__restrict__ tmpch,
                                                                int mode, uint32_t c,
                                                                const uint32_t *__restrict__ offs, uint32_t ntiles,
                                                                const uint32_t *__restrict__ base, int dir,
                                                                uint32_t *__restrict__ SA, uint32_t nkeys)
{
    __shared__ uint32_t wcount[kWavesPerBlock][256];
    __shared__ uint32_t gpos[256]; // first destination index of the tile for each bucket
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    for (int i = t; i < kWavesPerBlock * 256; i += kBlock) (&wcount[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile0 = blockIdx.x * (uint32_t)kIndTile;
    const uint32_t wave0 = tile0 + (uint32_t)w * (kWave * kIndItems);
    uint32_t val[kIndItems], dig[kIndItems], rnk[kIndItems];
    bool ok[kIndItems];
#pragma unroll
    for (int k = 0; k < kIndItems; ++k) {
        const uint32_t i = wave0 + (uint32_t)k * kWave + lane;
        ok[k] = false;
        dig[k] = 0;
        val[k] = 0;
        if (i < len) {
            const uint32_t p = round_item(src, len, rev != 0, i);
            const uint32_t ch = tmpch[i];
            ok[k] = p != 0 && induce_accept(ch, c, mode);
            dig[k] = ch;
            val[k] = p - 1u;
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
            SA[dst] = val[k];
        }
    }
}

__global__ void set_u32_kernel(uint32_t *p, uint32_t v) { *p = v; }

} // namespace sx

using namespace sx;

size_t sx_induce_scratch_bytes(uint64_t N, uint32_t sigma)
{
    const uint64_t ntiles = (N + kIndTile - 1) / kIndTile;
    return (size_t)N + 256 + (size_t)sigma * ntiles * 4 + 256 + 3 * 1024 + 4096;
}

namespace {
struct induce_state {
    sx_ctx *ctx;
    const uint8_t *T;
    uint32_t *SA;
    uint8_t *tmpch;
    uint32_t *hist, *cursor, *base, *ctl;
    uint32_t sigma;
    int small_alphabet;
};

// one stable multi-way split; returns the number of entries appended to bucket c
int induce_round(induce_state &st, const uint32_t *src, uint32_t len, int rev, int mode, uint32_t c, int dir,
                 uint32_t *added_c)
{
    sx_ctx *ctx = st.ctx;
    const uint32_t ntiles = sx_div_up(len, kIndTile);
    sx_launch(ctx, SX_KC_INDUCE_GATHER, (uint64_t)len * 6, induce_gather_kernel, dim3(ntiles), dim3(kBlock), src,
              len, rev, st.T, mode, c, st.tmpch, st.hist, ntiles, st.sigma);
    sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)ntiles * st.sigma * 8, induce_offsets_kernel, dim3(st.sigma),
              dim3(kBlock), st.hist, ntiles, st.cursor, st.base, dir, c, st.ctl);
    if (st.small_alphabet)
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, (uint64_t)len * 9, induce_scatter_kernel<3>, dim3(ntiles),
                  dim3(kBlock), src, len, rev, (const uint8_t *)st.tmpch, mode, c, (const uint32_t *)st.hist,
                  ntiles, (const uint32_t *)st.base, dir, st.SA, st.sigma);
    else
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, (uint64_t)len * 9, induce_scatter_kernel<8>, dim3(ntiles),
                  dim3(kBlock), src, len, rev, (const uint8_t *)st.tmpch, mode, c, (const uint32_t *)st.hist,
                  ntiles, (const uint32_t *)st.base, dir, st.SA, st.sigma);
    ctx->stats.induce_rounds++;
    if (added_c) SX_TRY(sx_readback(ctx, st.ctl, 1, added_c));
    return 0;
}
} // namespace

int sx_induce(sx_ctx *ctx, const sx_text_info &ti, uint32_t sigma, const uint32_t *sorted_lms, uint32_t *SA,
              sx_arena &arena)
{
    const uint64_t N = ti.N;
    if (N > 0xFFFFFFFFull) return sx_fail_msg(ctx, SX_E_ARG, "induce: n exceeds 32-bit positions");
    // buckets that hold anything: 0 .. maxc
    const uint32_t nk = ti.maxc + 1 < sigma ? ti.maxc + 1 : sigma;
    induce_state st;
    st.ctx = ctx;
    st.T = ti.T;
    st.SA = SA;
    st.sigma = nk;
    st.small_alphabet = nk <= 8;
    const uint64_t max_tiles = (N + kIndTile - 1) / kIndTile;
    st.tmpch = arena.take<uint8_t>(N);
    st.hist = arena.take<uint32_t>((size_t)nk * max_tiles);
    st.cursor = arena.take<uint32_t>(256);
    st.base = arena.take<uint32_t>(256);
    st.ctl = arena.take<uint32_t>(16);
    if (!st.tmpch || !st.hist || !st.cursor || !st.base || !st.ctl)
        return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small");

    // bucket boundaries on the host (sa_is.c:176-201)
    uint32_t begin[257], lms_off[257];
    begin[0] = 0;
    lms_off[0] = 0;
    for (uint32_t c = 0; c < 256; ++c) {
        begin[c + 1] = begin[c] + ti.h_all[c];
        lms_off[c + 1] = lms_off[c] + ti.h_lms[c];
    }

    // the sentinel suffix (sa_is.c:463: SA[0] = n)
    sx_launch(ctx, SX_KC_MISC, 0, set_u32_kernel, dim3(1), dim3(1), SA, (uint32_t)ti.n);

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
            SX_TRY(induce_round(st, SA + lo, hi - lo, 0, MODE_L_FROM_L, c, +1, &added));
            lo = hi;
            hi += added;
        }
        if (hi - begin[c] != ti.h_l[c])
            return sx_fail_msg(ctx, SX_E_INTERNAL, "induce L: bucket did not receive its L-type count");
        if (ti.h_lms[c])
            SX_TRY(induce_round(st, sorted_lms + lms_off[c], ti.h_lms[c], 0, MODE_L_FROM_LMS, c, +1, nullptr));
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
            SX_TRY(induce_round(st, SA + lo, hi - lo, 1, MODE_S_FROM_S, c, -1, &added));
            hi = lo;
            lo -= added;
        }
        const uint32_t n_s = ti.h_all[c] - ti.h_l[c];
        if (c > 0 && end_c - lo != n_s)
            return sx_fail_msg(ctx, SX_E_INTERNAL, "induce S: bucket did not receive its S-type count");
        if (ti.h_l[c])
            SX_TRY(induce_round(st, SA + begin[c], ti.h_l[c], 1, MODE_S_FROM_L, c, -1, nullptr));
    }
    return 0;
}
