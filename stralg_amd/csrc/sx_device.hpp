// sx_device.hpp -- wave64 / workgroup building blocks shared by the kernels.
//
// gfx950 wavefronts are 64 lanes wide; everything here is written for that
// width (ballots are 64-bit, scans take 6 shuffle steps).  All collectives
// must be reached by every lane of a wave: callers predicate, never branch
// around them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sx {

constexpr int kWave = 64;
constexpr int kBlock = 256; // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ uint64_t lanemask_le() { return (2ull << lane_id()) - 1ull; }
// maximum over the wave's lanes (every lane gets it; all lanes must call)
__device__ __forceinline__ uint32_t wave_reduce_max(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_xor(v, d, 64);
        v = v > o ? v : o;
    }
    return v;
}
// OR over the wave's lanes (every lane gets it; all lanes must call)
__device__ __forceinline__ uint32_t wave_reduce_or(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v |= __shfl_xor(v, d, 64);
    return v;
}

struct OpAdd {
    __device__ __forceinline__ static uint32_t identity() { return 0u; }
    __device__ __forceinline__ static uint32_t apply(uint32_t a, uint32_t b) { return a + b; }
};
struct OpMax {
    __device__ __forceinline__ static uint32_t identity() { return 0u; }
    __device__ __forceinline__ static uint32_t apply(uint32_t a, uint32_t b) { return a > b ? a : b; }
};

// Inclusive scan over the 64 lanes of a wave in six data-parallel-primitive steps: inside every row of 16 lanes the
// value of the lane 1, 2, 4, 8 places to the left comes in through the DPP operand path (row_shr; lanes that have no
// such neighbour read the identity, 0), then lane 15 of rows 0 and 2 is broadcast into rows 1 and 3 (row_bcast:15) and
// lane 31 into rows 2 and 3 (row_bcast:31).  The compiler folds each step into one v_add_u32_dpp / v_max_u32_dpp:
// six instructions and no LDS traffic, against six ds_bpermute round trips with their address arithmetic for a scan
// by __shfl_up.  Both operators here have the identity 0.
template <class Op, int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_scan_step(uint32_t x)
{
    return Op::apply((uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, true), x);
}
template <class Op> __device__ __forceinline__ uint32_t row_inclusive_scan(uint32_t v) // inside rows of 16 lanes
{
    v = dpp_scan_step<Op, 0x111, 0xF>(v); // row_shr:1
    v = dpp_scan_step<Op, 0x112, 0xF>(v); // row_shr:2
    v = dpp_scan_step<Op, 0x114, 0xF>(v); // row_shr:4
    v = dpp_scan_step<Op, 0x118, 0xF>(v); // row_shr:8
    return v;
}
template <class Op> __device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    v = row_inclusive_scan<Op>(v);
    v = dpp_scan_step<Op, 0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
    v = dpp_scan_step<Op, 0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3
    return v;
}
// Sums of 16-bit fields packed in a 64-bit word (no field ever overflows, so the two halves never exchange a carry)
__device__ __forceinline__ uint64_t wave_inclusive_sum_packed(uint64_t v)
{
    const uint32_t lo = wave_inclusive_scan<OpAdd>((uint32_t)v), hi = wave_inclusive_scan<OpAdd>((uint32_t)(v >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// the wave's total of such a word, in every lane (the scan's last lane, read through a scalar register)
__device__ __forceinline__ uint64_t wave_total_packed(uint64_t v)
{
    const uint32_t lo = wave_inclusive_scan<OpAdd>((uint32_t)v), hi = wave_inclusive_scan<OpAdd>((uint32_t)(v >> 32));
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)lo, kWave - 1) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, kWave - 1) << 32);
}
__device__ __forceinline__ uint64_t row_inclusive_sum_packed(uint64_t v) // inside rows of 16 lanes
{
    const uint32_t lo = row_inclusive_scan<OpAdd>((uint32_t)v), hi = row_inclusive_scan<OpAdd>((uint32_t)(v >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// Exclusive scan over the kBlock threads of a workgroup.  `lds` needs
// kWavesPerBlock entries; the function ends with a barrier so `lds` can be
// reused by the caller straight away.
template <class Op>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t &total)
{
    const int lane = lane_id(), w = wave_id();
    uint32_t inc = wave_inclusive_scan<Op>(v);
    uint32_t prev = __shfl_up(inc, 1u, kWave);
    uint32_t exc = lane == 0 ? Op::identity() : prev;
    if (lane == kWave - 1) lds[w] = inc;
    __syncthreads();
    uint32_t base = Op::identity(), tot = Op::identity();
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) {
        uint32_t x = lds[i];
        if (i < w) base = Op::apply(base, x);
        tot = Op::apply(tot, x);
    }
    __syncthreads();
    total = tot;
    return Op::apply(base, exc);
}

// 64-bit sum variant (packed 16-bit counters: several small scans in one)
__device__ __forceinline__ uint64_t block_exclusive_sum64(uint64_t v, uint64_t *lds /* kWavesPerBlock */, uint64_t &total)
{
    const int lane = lane_id(), w = wave_id();
    const uint64_t inc = wave_inclusive_sum_packed(v); // (every caller's word holds 16-bit fields)
    if (lane == kWave - 1) lds[w] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) {
        const uint64_t x = lds[i];
        if (i < w) base += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// two packed words at once (one pair of barriers instead of two)
__device__ __forceinline__ void block_exclusive_sum64x2(uint64_t &a, uint64_t &b, uint64_t *lds /* 2 * kWavesPerBlock */)
{
    const int lane = lane_id(), w = wave_id();
    const uint64_t ia = wave_inclusive_sum_packed(a), ib = wave_inclusive_sum_packed(b);
    if (lane == kWave - 1) lds[w] = ia, lds[kWavesPerBlock + w] = ib;
    __syncthreads();
    uint64_t ba = 0, bb = 0;
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) {
        if (i < w) ba += lds[i], bb += lds[kWavesPerBlock + i];
    }
    __syncthreads();
    a = ba + ia - a;
    b = bb + ib - b;
}

template <class Op> __device__ __forceinline__ uint32_t block_reduce(uint32_t v, uint32_t *lds)
{
    uint32_t total;
    (void)block_exclusive_scan<Op>(v, lds, total);
    return total;
}

// Stable ranking step shared by the radix scatter and the induce scatter:
// items are presented wave by wave in (item k, lane) order; `counter` is this
// wave's LDS counter row (one u32 per digit).  Returns the rank of the item
// among the wave's items with the same digit seen so far.
// The lanes holding the same digit are found bit by bit (a ballot per bit).  The ranking of a radix pass is
// bound by vector instructions, not by memory, so each step is kept to four: bit b of the digit is spread over
// a register (x), compared (the ballot m), and each half of the peer mask keeps the lanes whose bit equals this
// lane's with one three-input bit operation, p & ~(m ^ x); the rank comes from mbcnt.  Every lane reads the
// digit's counter before the group's first lane adds the group's size: the LDS executes a wave's operations in
// order, so no value has to come back from the atomic and be passed around.  ALL: every lane is valid.
#ifndef SX_OPAQUE_VGPR // the CPU test harness defines it away
#define SX_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
#endif
// nothing is scheduled across this point (keeps the compiler from interleaving the unrolled copies of a register-hungry
// step -- eight window decodes at once spilled a thousand registers)
#ifndef SX_WAVES_PER_EU // exactly N waves a SIMD: the register budget of a kernel whose workgroups must fit a CU two at a time
#define SX_WAVES_PER_EU(N) __attribute__((amdgpu_waves_per_eu(N, N)))
#endif
#ifndef SX_SCHED_FENCE // (the CPU test harness defines it away as well)
#define SX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
template <int BITS, bool ALL>
__device__ __forceinline__ uint32_t wave_rank_inorder(uint32_t digit, bool valid, uint32_t *counter)
{
    uint32_t lo = ~0u, hi = ~0u;
    if (!ALL) {
        const uint64_t v = __ballot(valid ? 1 : 0);
        lo = (uint32_t)v;
        hi = (uint32_t)(v >> 32);
    }
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        uint32_t x = (uint32_t)((int32_t)(digit << (31 - b)) >> 31);
        SX_OPAQUE_VGPR(x); // compare x itself: the compiler would rather shift the digit once more for the test
        const uint64_t m = __ballot(x != 0u ? 1 : 0);
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)m, x, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(m >> 32), x, 0x90);
    }
    const uint32_t r = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    // (a relaxed atomic load: the compiler must keep it ordered with the adds to the same, run-time, address)
    const uint32_t old = __hip_atomic_load(&counter[(ALL || valid) ? digit : 0u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_wave_barrier();
    if ((ALL || valid) && r == 0) atomicAdd(&counter[digit], (uint32_t)(__popc(lo) + __popc(hi)));
    __builtin_amdgcn_wave_barrier();
    return old + r;
}
template <int BITS> __device__ __forceinline__ uint32_t wave_rank_step(uint32_t digit, bool valid, uint32_t *counter)
{
    return wave_rank_inorder<BITS, false>(digit, valid, counter);
}

// ---- chained scan across workgroups (decoupled look-back) ------------------------------
// One 8-byte status word per (tile, digit): [63:40] epoch of the launch, [39:38] state
// (1 = the tile's own count, 2 = inclusive prefix over tiles 0..tile), [31:0] value.
// The word is written and read whole with relaxed agent-scope 8-byte atomics, so the
// data is its own flag: no fence and no other memory needs to become visible
// (MI355X guide, inter-workgroup visibility: data-tagged granules).  Tiles take their
// index from an atomic ticket, so every tile a waiter polls has already started and
// never waits on a later one: the chain cannot deadlock whatever the residency.
// Epochs make stale words from earlier launches inert; no per-launch memset.
// The first kChainHeader words of the status buffer are a header: word 0 is set
// when a wait exceeds kChainSpinLimit polls (a bug, never expected); the waiter then
// gives up so that the grid always drains, and the host reports the error.
constexpr uint64_t kChainValueMask = 0xFFFFFFFFull;
constexpr uint32_t kChainHeader = 8;
constexpr uint32_t kChainSpinLimit = 1u << 22;
__device__ __forceinline__ uint64_t chain_pack(uint32_t epoch, uint32_t state, uint32_t value)
{
    return ((uint64_t)epoch << 40) | ((uint64_t)state << 38) | (uint64_t)value;
}

// Step 1, as early as the tile knows its counts: publish them (tile 0 publishes its inclusive
// prefix straight away).
__device__ __forceinline__ void chain_publish(uint64_t *__restrict__ status, uint32_t ndigits, uint32_t tile,
                                              uint32_t d, uint32_t count, uint32_t epoch)
{
    uint64_t *mine = status + kChainHeader + (uint64_t)tile * ndigits + d;
    __hip_atomic_store(mine, chain_pack(epoch, tile == 0 ? 2u : 1u, count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Step 2, by the same thread: the sum of `count` over tiles [0, tile), walking back over the
// predecessors BATCH status words at a time (the tiles of a launch that start together cannot
// see each other's inclusive prefixes yet, so walks are as long as the number of tiles in flight:
// without the batching every step would cost a full L2 round trip); then publish the inclusive
// prefix of this tile.
template <int BATCH>
__device__ __forceinline__ uint32_t chain_lookback(uint64_t *__restrict__ status, uint32_t ndigits, uint32_t tile,
                                                   uint32_t d, uint32_t count, uint32_t epoch)
{
    if (tile == 0) return 0u;
    uint64_t *words = status + kChainHeader;
    uint32_t prefix = 0;
    bool done = false;
    uint32_t j = tile; // predecessors still to visit: j-1, j-2, ...
    while (!done && j > 0) {
        uint64_t w[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const uint32_t idx = j > (uint32_t)i ? j - 1u - (uint32_t)i : 0u;
            w[i] = __hip_atomic_load(words + (uint64_t)idx * ndigits + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            if (!done && j > (uint32_t)i) {
                const uint64_t *theirs = words + (uint64_t)(j - 1u - (uint32_t)i) * ndigits + d;
                uint64_t x = w[i];
                uint32_t spins = 0;
                while ((uint32_t)(x >> 40) != epoch || ((x >> 38) & 3ull) == 0) {
                    if (++spins > kChainSpinLimit) { // give up: flag it, treat the predecessor as empty
                        __hip_atomic_store(status, (uint64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        x = chain_pack(epoch, 2u, 0u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    x = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                prefix += (uint32_t)(x & kChainValueMask);
                if (((x >> 38) & 3ull) == 2ull) done = true;
            }
        }
        j = j > (uint32_t)BATCH ? j - (uint32_t)BATCH : 0u;
    }
    __hip_atomic_store(words + (uint64_t)tile * ndigits + d, chain_pack(epoch, 2u, prefix + count), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    return prefix;
}

// Called by the thread that owns digit d of tile `tile`: publish, walk back, publish again.
__device__ __forceinline__ uint32_t chain_exclusive_prefix(uint64_t *__restrict__ status, uint32_t ndigits,
                                                           uint32_t tile, uint32_t d, uint32_t count,
                                                           uint32_t epoch)
{
    chain_publish(status, ndigits, tile, d, count, epoch);
    return chain_lookback<8>(status, ndigits, tile, d, count, epoch);
}

// Wave-wide form for small digit counts: the whole wave (all 64 lanes must call) looks
// back 64 predecessors of digit d at a time, so the walk over the tiles that started
// together with this one takes tiles/64 steps instead of tiles/8.
__device__ __forceinline__ uint32_t chain_exclusive_prefix_wave(uint64_t *__restrict__ status, uint32_t ndigits,
                                                                uint32_t tile, uint32_t d, uint32_t count,
                                                                uint32_t epoch)
{
    uint64_t *words = status + kChainHeader;
    uint64_t *mine = words + (uint64_t)tile * ndigits + d;
    const int lane = lane_id();
    if (tile == 0) {
        if (lane == 0) __hip_atomic_store(mine, chain_pack(epoch, 2u, count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0u;
    }
    if (lane == 0) __hip_atomic_store(mine, chain_pack(epoch, 1u, count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t prefix = 0;
    uint32_t base = tile; // predecessors still to visit: base-1, base-2, ...
    for (;;) {
        const bool valid = base > (uint32_t)lane;
        const uint64_t *theirs = words + (uint64_t)(valid ? base - 1u - (uint32_t)lane : 0u) * ndigits + d;
        uint64_t x = valid ? __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        uint32_t spins = 0;
        for (;;) {
            const bool ready = !valid || ((uint32_t)(x >> 40) == epoch && ((x >> 38) & 3ull) != 0);
            if (!__any(ready ? 0 : 1)) break;
            if (!ready) {
                if (++spins > kChainSpinLimit) { // give up: flag it, treat the predecessor as empty
                    __hip_atomic_store(status, (uint64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x = chain_pack(epoch, 2u, 0u);
                } else {
                    __builtin_amdgcn_s_sleep(1);
                    x = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        const bool incl = valid && ((x >> 38) & 3ull) == 2ull;
        const uint64_t incl_mask = __ballot(incl ? 1 : 0);
        // nearest predecessor holding an inclusive prefix = lowest lane in the mask
        const int stop = incl_mask ? (__ffsll((unsigned long long)incl_mask) - 1) : 63;
        uint32_t v = (valid && lane <= stop) ? (uint32_t)(x & kChainValueMask) : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
        prefix += v;
        if (incl_mask) break;
        base -= (uint32_t)kWave; // no inclusive word among 64 predecessors: base > 64 here (tile 0 is always inclusive)
    }
    if (lane == 0)
        __hip_atomic_store(mine, chain_pack(epoch, 2u, prefix + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return prefix;
}

// 16-byte store of data this kernel will not read again (streaming: no L2 allocation)
typedef unsigned int sx_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store16(uint4 *dst, uint4 v)
{
    const sx_v4u x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<sx_v4u *>(dst));
}

// ---- byte windows of the text at any offset ----------------------------------------------
// gfx950 serves global 16-byte loads at any byte address (the compiler emits global_load_dwordx4 for these
// copies); a per-lane 16-byte load costs the address unit the same whatever its alignment, so two unaligned
// loads replace the three aligned ones plus funnel shifts that a 32-byte window needs otherwise.
__device__ __forceinline__ uint64_t funnel64(uint64_t lo, uint64_t hi, uint32_t r)
{
    return r ? (lo >> r) | (hi << (64u - r)) : lo;
}
__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

// bytes [p, p+16) as two little-endian u64
__device__ __forceinline__ void load_bytes16(const uint8_t *__restrict__ T, uint64_t p, uint64_t &o0, uint64_t &o1)
{
    uint64_t q[2];
    __builtin_memcpy(q, T + p, 16);
    o0 = q[0];
    o1 = q[1];
}

// bytes [p, p+32) as four little-endian u64
__device__ __forceinline__ void load_bytes32(const uint8_t *__restrict__ T, uint64_t p, uint64_t (&o)[4])
{
    __builtin_memcpy(o, T + p, 32);
}

// The same from a 16-byte aligned image in LDS (a staged piece of the text): aligned 16-byte reads and funnel
// shifts, LDS has no unaligned wide reads.  The image must hold 16 readable bytes past the last one wanted.
__device__ __forceinline__ void lds_bytes16(const uint8_t *img, uint32_t p, uint64_t &o0, uint64_t &o1)
{
    const uint32_t base = p & ~15u, s = p & 15u;
    const uint4 v0 = *reinterpret_cast<const uint4 *>(img + base);
    const uint4 v1 = *reinterpret_cast<const uint4 *>(img + base + 16);
    const uint64_t q0 = pack64(v0.x, v0.y), q1 = pack64(v0.z, v0.w), q2 = pack64(v1.x, v1.y), q3 = pack64(v1.z, v1.w);
    const bool up = s >= 8;
    const uint64_t a = up ? q1 : q0, b = up ? q2 : q1, c = up ? q3 : q2;
    const uint32_t r = 8u * (s & 7u);
    o0 = funnel64(a, b, r);
    o1 = funnel64(b, c, r);
}
__device__ __forceinline__ void lds_bytes32(const uint8_t *img, uint32_t p, uint64_t (&o)[4])
{
    const uint32_t base = p & ~15u, s = p & 15u;
    const uint4 v0 = *reinterpret_cast<const uint4 *>(img + base);
    const uint4 v1 = *reinterpret_cast<const uint4 *>(img + base + 16);
    const uint4 v2 = *reinterpret_cast<const uint4 *>(img + base + 32);
    const uint64_t q0 = pack64(v0.x, v0.y), q1 = pack64(v0.z, v0.w), q2 = pack64(v1.x, v1.y), q3 = pack64(v1.z, v1.w),
                   q4 = pack64(v2.x, v2.y), q5 = pack64(v2.z, v2.w);
    const bool up = s >= 8;
    const uint64_t a = up ? q1 : q0, b = up ? q2 : q1, c = up ? q3 : q2, d = up ? q4 : q3, e = up ? q5 : q4;
    const uint32_t r = 8u * (s & 7u);
    o[0] = funnel64(a, b, r);
    o[1] = funnel64(b, c, r);
    o[2] = funnel64(c, d, r);
    o[3] = funnel64(d, e, r);
}

// Bit `bit` of each of the 16 bytes of four words as a 16-bit mask, byte 0 of m0 first; every byte of m0..m3 must
// hold 0 or 1 << bit.  Four dot products with the weights 1, 2, 4, 8 / 16, 32, 64, 128.
__device__ __forceinline__ uint32_t gather16(uint32_t m0, uint32_t m1, uint32_t m2, uint32_t m3, int bit)
{
    // bit `bit` of every byte of the four words -> 16 bits, byte 0 of m0 first
    const uint32_t lo = __builtin_amdgcn_udot4(m1, 0x80402010u, __builtin_amdgcn_udot4(m0, 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(m3, 0x80402010u, __builtin_amdgcn_udot4(m2, 0x08040201u, 0u, false), false);
    return ((lo >> bit) | (hi << (8 - bit))) & 0xFFFFu; // (a byte holds 0 or 1 << bit: the sums are the masks << bit)
}


// In-place exclusive prefix sum of row[0..count) by one workgroup; returns the total.  8192 entries a step: loaded
// and stored with the lanes on consecutive entries, turned through LDS (`stage`, kScanRowStage words; a word of
// padding per 32 keeps both views free of bank conflicts) so that a thread scans 32 consecutive ones.  (With each
// thread loading its own 32 entries a load instruction touched 64 cache lines: 100 us for the 131 072 tile counts
// of a large induce round, most of the time between that round's counting and scattering launches.)
constexpr int kScanRowPer = 32;
constexpr int kScanRowStage = kBlock * kScanRowPer + kBlock;
__device__ __forceinline__ uint32_t block_scan_row_inplace(uint32_t *__restrict__ row, uint64_t count,
                                                           uint32_t *lds, uint32_t *stage)
{
    constexpr int kPer = kScanRowPer;
    const uint32_t t = threadIdx.x;
    uint32_t carry = 0;
    for (uint64_t start = 0; start < count; start += (uint64_t)kBlock * kPer) { // uniform trip count
        uint32_t v[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint64_t i = start + (uint64_t)k * kBlock + t;
            v[k] = i < count ? row[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint32_t e = (uint32_t)k * kBlock + t;
            stage[e + (e >> 5)] = v[k];
        }
        __syncthreads();
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint32_t e = t * kPer + (uint32_t)k;
            v[k] = stage[e + (e >> 5)];
            acc += v[k];
        }
        uint32_t tot;
        uint32_t run = carry + block_exclusive_scan<OpAdd>(acc, lds, tot);
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint32_t e = t * kPer + (uint32_t)k;
            stage[e + (e >> 5)] = run;
            run += v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint64_t i = start + (uint64_t)k * kBlock + t;
            const uint32_t e = (uint32_t)k * kBlock + t;
            if (i < count) row[i] = stage[e + (e >> 5)];
        }
        carry += tot;
        __syncthreads(); // `stage` is refilled by the next step
    }
    return carry;
}

// The same by a workgroup of 1024 threads, for the rows of an induce round's tile counts (a launch of a handful of
// workgroups between the counting and the scattering launch: its latency is the round's).  A thread owns 8 consecutive
// counts of each of four 8192-count pieces and loads them as 16-byte pairs, all eight loads in flight together (a wave's
// load covers 2 KiB of consecutive counts, so nothing has to be turned through LDS); 32 768 counts an iteration: one
// trip to memory for the 33 000 tile counts of a 1 GiB text's largest round (the 256-thread form above: five).
#ifndef SX_ROW_THREADS
#define SX_ROW_THREADS 1024 // (the CPU test harness builds with 256: a fiber per thread and launch)
#endif
constexpr int kRowThreads = SX_ROW_THREADS, kRowWaves = kRowThreads / kWave, kRowPieces = 4, kRowPer = 8;
__device__ __forceinline__ uint32_t wide_scan_row_inplace(uint32_t *__restrict__ row, uint64_t count, uint32_t *lds /* kRowPieces * kRowWaves */)
{
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    uint32_t carry = 0;
    for (uint64_t start = 0; start < count; start += (uint64_t)kRowPieces * kRowThreads * kRowPer) { // uniform trip count
        uint32_t v[kRowPieces][kRowPer], s[kRowPieces], inc[kRowPieces];
#pragma unroll
        for (int q = 0; q < kRowPieces; ++q) {
            const uint64_t base = start + (uint64_t)q * kRowThreads * kRowPer + (uint64_t)t * kRowPer;
            if (base + kRowPer <= count) {
                __builtin_memcpy(v[q], row + base, sizeof(v[q]));
            } else {
#pragma unroll
                for (int k = 0; k < kRowPer; ++k) v[q][k] = base + k < count ? row[base + k] : 0u;
            }
        }
#pragma unroll
        for (int q = 0; q < kRowPieces; ++q) {
            s[q] = 0;
#pragma unroll
            for (int k = 0; k < kRowPer; ++k) s[q] += v[q][k];
            inc[q] = wave_inclusive_scan<OpAdd>(s[q]);
            if (lane == kWave - 1) lds[q * kRowWaves + w] = inc[q];
        }
        __syncthreads();
        uint32_t before[kRowPieces], run = carry;
#pragma unroll
        for (int q = 0; q < kRowPieces; ++q) {
            uint32_t mine = 0, all = 0;
#pragma unroll
            for (int ww = 0; ww < kRowWaves; ++ww) {
                const uint32_t x = lds[q * kRowWaves + ww];
                if ((uint32_t)ww < w) mine += x;
                all += x;
            }
            before[q] = run + mine + inc[q] - s[q];
            run += all;
        }
        carry = run;
        __syncthreads(); // `lds` is rewritten by the next iteration
#pragma unroll
        for (int q = 0; q < kRowPieces; ++q) {
            const uint64_t base = start + (uint64_t)q * kRowThreads * kRowPer + (uint64_t)t * kRowPer;
            uint32_t o[kRowPer], r = before[q];
#pragma unroll
            for (int k = 0; k < kRowPer; ++k) {
                o[k] = r;
                r += v[q][k];
            }
            if (base + kRowPer <= count) {
                __builtin_memcpy(row + base, o, sizeof(o));
            } else {
#pragma unroll
                for (int k = 0; k < kRowPer; ++k)
                    if (base + k < count) row[base + k] = o[k];
            }
        }
    }
    return carry;
}

} // namespace sx
