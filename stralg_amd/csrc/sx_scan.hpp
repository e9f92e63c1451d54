// sx_scan.hpp -- device-wide scan with fused input / output functors.
//
//   device_scan<Op>(ctx, n, in, out, d_total)
//     in(i)            -> uint32 value of element i        (i is uint64)
//     out(i, excl, v)  <- exclusive prefix of element i and its own value
//     d_total          <- Op over all elements (optional device pointer)
//
// Stream compaction is the same call with in = predicate and
// out = "if (v) dst[excl] = ...".  Three launches (tile reduce, one-workgroup
// scan of the tile totals, tile scan); every element's input functor is
// evaluated twice, which is cheaper than materialising flags in HBM.
#pragma once
#include "sx_common.hpp"
#include "sx_device.hpp"

namespace sx {

constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;

template <class Op, class In>
__global__ __launch_bounds__(kBlock) void scan_reduce_kernel(In in, uint64_t n, uint32_t *tile_tot)
{
    __shared__ uint32_t lds[kWavesPerBlock];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    uint32_t acc = Op::identity();
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint64_t i = base + k;
        if (i < n) acc = Op::apply(acc, in(i));
    }
    const uint32_t tot = block_reduce<Op>(acc, lds);
    if (threadIdx.x == 0) tile_tot[blockIdx.x] = tot;
}

// One workgroup turns tile totals into exclusive tile prefixes, in place.
template <class Op>
__global__ __launch_bounds__(kBlock) void scan_spine_kernel(uint32_t *tile_tot, uint32_t ntiles,
                                                            uint32_t *d_total)
{
    __shared__ uint32_t lds[kWavesPerBlock];
    constexpr int kPer = 16;
    uint32_t carry = Op::identity();
    for (uint64_t start = 0; start < ntiles; start += (uint64_t)kBlock * kPer) { // uniform trip count
        const uint64_t i0 = start + (uint64_t)threadIdx.x * kPer;
        uint32_t v[kPer];
        uint32_t acc = Op::identity();
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            v[k] = i0 + k < ntiles ? tile_tot[i0 + k] : Op::identity();
            acc = Op::apply(acc, v[k]);
        }
        uint32_t tot;
        uint32_t run = Op::apply(carry, block_exclusive_scan<Op>(acc, lds, tot));
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            if (i0 + k < ntiles) tile_tot[i0 + k] = run;
            run = Op::apply(run, v[k]);
        }
        carry = Op::apply(carry, tot);
    }
    if (threadIdx.x == 0 && d_total) *d_total = carry;
}

template <class Op, class In, class Out>
__global__ __launch_bounds__(kBlock) void scan_apply_kernel(In in, Out out, uint64_t n,
                                                            const uint32_t *tile_pre)
{
    __shared__ uint32_t lds[kWavesPerBlock];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t acc = Op::identity();
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint64_t i = base + k;
        v[k] = i < n ? in(i) : Op::identity();
        acc = Op::apply(acc, v[k]);
    }
    uint32_t tot;
    uint32_t run = block_exclusive_scan<Op>(acc, lds, tot);
    run = Op::apply(tile_pre[blockIdx.x], run);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint64_t i = base + k;
        if (i < n) out(i, run, v[k]);
        run = Op::apply(run, v[k]);
    }
}

// Scratch for the tile totals (the context's SCAN slab).
uint32_t *sx_scan_scratch(sx_ctx *ctx, uint32_t ntiles);

template <class Op, class In, class Out>
static inline int device_scan(sx_ctx *ctx, uint64_t n, In in, Out out, uint32_t *d_total,
                              int kclass = SX_KC_SCAN, uint64_t alg_bytes = 0)
{
    if (n == 0) {
        if (d_total) SX_CHECK(hipMemsetAsync(d_total, 0, sizeof(uint32_t), ctx->stream));
        return 0;
    }
    const uint32_t ntiles = sx_div_up(n, kScanTile);
    uint32_t *tile_tot = sx_scan_scratch(ctx, ntiles);
    if (!tile_tot) return sx_fail_msg(ctx, SX_E_NOMEM, "scan scratch");
    sx_launch(ctx, kclass, alg_bytes / 2, scan_reduce_kernel<Op, In>, dim3(ntiles), dim3(kBlock), in, n,
              tile_tot);
    sx_launch(ctx, kclass, 0, scan_spine_kernel<Op>, dim3(1), dim3(kBlock), tile_tot, ntiles, d_total);
    sx_launch(ctx, kclass, alg_bytes - alg_bytes / 2, scan_apply_kernel<Op, In, Out>, dim3(ntiles),
              dim3(kBlock), in, out, n, (const uint32_t *)tile_tot);
    return 0;
}

// ---- stream compaction of 0/1 flags -------------------------------------------------------
//   device_compact(ctx, n, flag, out, d_total)
//     flag(i)          -> bool
//     out(i, dst, f)   <- called for every i; dst = number of set flags before i
// Same three launches as device_scan, but the elements are visited in striped order (lane l of
// row k reads element tile + k*256 + l: coalesced), and because the values are single bits the
// prefix inside a row is a ballot and a popcount instead of a shuffle scan.
template <class Flag>
__global__ __launch_bounds__(kBlock) void compact_count_kernel(Flag flag, uint64_t n, uint32_t *tile_tot)
{
    __shared__ uint32_t lds[kWavesPerBlock];
    const uint64_t tile0 = (uint64_t)blockIdx.x * kScanTile;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint64_t i = tile0 + (uint64_t)k * kBlock + threadIdx.x;
        cnt += (uint32_t)__popcll(__ballot((i < n && flag(i)) ? 1 : 0));
    }
    if (lane_id() == 0) lds[wave_id()] = cnt; // every lane of a wave holds the wave's count
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) tot += lds[w];
        tile_tot[blockIdx.x] = tot;
    }
}

template <class Flag, class Out>
__global__ __launch_bounds__(kBlock) void compact_apply_kernel(Flag flag, Out out, uint64_t n,
                                                               const uint32_t *tile_pre)
{
    __shared__ uint32_t cnt[kScanItems][kWavesPerBlock];
    const uint64_t tile0 = (uint64_t)blockIdx.x * kScanTile;
    const int w = wave_id();
    bool f[kScanItems];
    uint32_t below[kScanItems];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint64_t i = tile0 + (uint64_t)k * kBlock + threadIdx.x;
        f[k] = i < n && flag(i);
        const uint64_t m = __ballot(f[k] ? 1 : 0);
        below[k] = (uint32_t)__popcll(m & lanemask_lt());
        if (lane_id() == 0) cnt[k][w] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    uint32_t run = tile_pre[blockIdx.x]; // set flags before (row k, wave w), rows and waves in order
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        uint32_t before_wave = 0, row_total = 0;
#pragma unroll
        for (int ww = 0; ww < kWavesPerBlock; ++ww) {
            const uint32_t c = cnt[k][ww];
            if (ww < w) before_wave += c;
            row_total += c;
        }
        const uint64_t i = tile0 + (uint64_t)k * kBlock + threadIdx.x;
        if (i < n) out(i, run + before_wave + below[k], f[k]);
        run += row_total;
    }
}

template <class Flag, class Out>
static inline int device_compact(sx_ctx *ctx, uint64_t n, Flag flag, Out out, uint32_t *d_total, int kclass = SX_KC_SCAN,
                                 uint64_t alg_bytes = 0)
{
    if (n == 0) {
        if (d_total) SX_CHECK(hipMemsetAsync(d_total, 0, sizeof(uint32_t), ctx->stream));
        return 0;
    }
    const uint32_t ntiles = sx_div_up(n, kScanTile);
    uint32_t *tile_tot = sx_scan_scratch(ctx, ntiles);
    if (!tile_tot) return sx_fail_msg(ctx, SX_E_NOMEM, "scan scratch");
    sx_launch(ctx, kclass, alg_bytes / 2, compact_count_kernel<Flag>, dim3(ntiles), dim3(kBlock), flag, n, tile_tot);
    sx_launch(ctx, kclass, 0, scan_spine_kernel<OpAdd>, dim3(1), dim3(kBlock), tile_tot, ntiles, d_total);
    sx_launch(ctx, kclass, alg_bytes - alg_bytes / 2, compact_apply_kernel<Flag, Out>, dim3(ntiles), dim3(kBlock), flag, out,
              n, (const uint32_t *)tile_tot);
    return 0;
}

// ---- common functors ------------------------------------------------------
struct InU32 {
    const uint32_t *p;
    __device__ __forceinline__ uint32_t operator()(uint64_t i) const { return p[i]; }
};
struct OutExclusive {
    uint32_t *p;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t excl, uint32_t) const { p[i] = excl; }
};

} // namespace sx
