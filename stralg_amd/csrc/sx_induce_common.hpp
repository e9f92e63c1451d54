// sx_induce_common.hpp -- what the kernels of the induced-sort passes share: tile sizes, the scan modes and their
// accept test, 16-byte entry loads, the seed-window fill.
#pragma once
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

} // namespace sx
