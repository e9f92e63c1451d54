// sx_localsort.hip -- the last step of the hybrid sort of the LMS suffixes' prefix keys.
//
// Role of stralg/sa_is.c:295-336 (the order reduce_SA reads the LMS substrings in) for the fast
// path of sx_lmssort.hip.  An LSD radix sort of (40-bit key, window, position) pairs moves
// 24 bytes per pair and pass through HBM, five times.  Here only the key's top kTopBits go
// through HBM passes (three 8-bit passes of sx_radix.hip, stable, lowest digit first): after
// them the pairs are ordered by their top 24 bits, i.e. they sit in *sub-buckets* of equal top
// bits -- a few hundred pairs each on a text whose prefixes are spread out -- and the
// remaining low bits only have to be ordered inside each sub-bucket.  That is done by this
// kernel in LDS: a workgroup takes the whole sub-buckets that start in its span of the
// array, orders them by (sub-bucket, low bits) with 8-bit LSD passes that never leave the CU,
// and writes what the rest of the build needs: the positions in suffix order, the symbol
// windows that rode in the keys' spare bits, and the members of groups of equal keys (the
// ties the refinement rounds work on) -- the sorted keys themselves are never written again.
// Per pair: 12 B read, 8 B written, against 3 x 25 B for the passes it replaces plus the
// 13 B of the pass that looked for equal neighbours.
//
// A sub-bucket that does not fit a workgroup's LDS (a text with many copies of one 10-symbol
// prefix: a repeat family, an AT-rich prefix of a genome) is left out and put on a list: the
// members of all such long sub-buckets are gathered into one array, ordered by (sub-bucket,
// low bits) with HBM passes of their own and written to the same outputs (sx_long_subbuckets).
// Without a list (or with more long sub-buckets than it holds) the kernel raises a flag; the
// caller then finishes with plain LSD passes.
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_scan.hpp"
#include "sx_internal.hpp"

namespace sx {

constexpr int kLsThreads = 512, kLsWaves = kLsThreads / kWave, kLsItems = 12;
constexpr int kLsCap = kLsThreads * kLsItems; // pairs a workgroup can hold: 6144 (72 KiB of LDS, two workgroups a CU)
// (span 3584: 4.10 ms, 4096: 3.88, 4608: 3.64, 5120: 3.40 at 1 GiB of DNA: the per-workgroup costs -- zeroing and
// scanning the 8192 counters -- are spread over more pairs; what is left of the reach bounds the sub-bucket that fits)
#ifndef SX_LS_SPAN
#define SX_LS_SPAN 5120
#endif
constexpr int kLsSpan = SX_LS_SPAN;           // a workgroup owns the sub-buckets that start in its span of the array
constexpr int kLsWords = kLsCap / 32;
constexpr int kLsBins = 8192;  // bins of the counting pass: 16 KiB of packed 16-bit counters in the (then unused) key image
constexpr int kLsMaxBin = 16;  // a bin with more pairs than this: stable passes instead (equal keys crowd one bin)
static_assert(kLsBins / 2 % kLsThreads == 0 && (kLsBins / 2 + 1) * 8 <= kLsCap * 8, "counter words per thread; both counter sets fit the key image");

// One workgroup: local indices i = global index - g0, g0 = blockIdx.x * kLsSpan.
//   s = first sub-bucket start at i >= 0, e = first sub-bucket start (or the end of the array) at i >= kLsSpan;
//   the workgroup owns [s, e).  Every sub-bucket starts in exactly one span, so the owned ranges tile the array.
// kin/vin: pairs ordered by (key & kmask) >> L; the key bits from kbits on are payload (the symbol window).
__global__ __launch_bounds__(kLsThreads, 4) void local_sort_kernel(
    const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint64_t m, uint32_t L, uint32_t kbits,
    uint32_t *__restrict__ vout, uint32_t *__restrict__ seedw /* or null */, uint32_t *__restrict__ tile_start,
    uint32_t *__restrict__ tile_cnt, uint2 *__restrict__ stage, uint8_t *__restrict__ stage_head,
    uint32_t *__restrict__ fail, uint32_t span /* kLsSpan, or more where the sub-buckets are known to be short */,
    uint32_t *__restrict__ long_list /* or null */, uint32_t *__restrict__ long_count, uint32_t long_cap)
{
    __shared__ uint64_t K[kLsCap]; // keys; then (payload << 32 | sub-bucket rank << L | low bits); per-wave counters during a ranking
    __shared__ uint32_t V[kLsCap]; // positions
    __shared__ uint32_t bnd[kLsWords], bpre[kLsWords]; // sub-bucket starts as bits; starts before each word
    __shared__ uint32_t s_first, s_end, s_last, s_scan[kLsWaves], s_kprev[2], s_max;
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t g0 = (uint64_t)blockIdx.x * span;
    const uint64_t kmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
    const uint32_t avail = m - g0 < (uint64_t)kLsCap ? (uint32_t)(m - g0) : (uint32_t)kLsCap; // pairs in reach
    const bool end_in_reach = g0 + avail == m;
    constexpr uint32_t kNone = 0xFFFFFFFFu;

    if (t == 0) {
        s_first = kNone;
        s_end = kNone;
        s_last = 0;
        const uint64_t kp = g0 ? kin[g0 - 1] : 0ull;
        s_kprev[0] = (uint32_t)kp, s_kprev[1] = (uint32_t)(kp >> 32);
    }
    for (int i = t; i < kLsWords; i += kLsThreads) bnd[i] = 0;
    // ---- load the keys in reach (index order: coalesced) ------------------------------------------------------------------
    {
        // (all loads of a thread in flight together: keys and positions of the whole reach, not only of the range the
        // workgroup turns out to own -- a loop over that range would wait for every load before issuing the next; a
        // smaller first load followed by the rest only when a sub-bucket crossed the span's end was no faster)
        uint64_t kk[kLsItems];
        uint32_t vv[kLsItems];
#pragma unroll
        for (int k = 0; k < kLsItems; ++k) {
            const uint32_t i = (uint32_t)t + (uint32_t)k * kLsThreads;
            kk[k] = i < avail ? __builtin_nontemporal_load(kin + g0 + i) : 0ull;
            vv[k] = i < avail ? __builtin_nontemporal_load(vin + g0 + i) : 0u;
        }
#pragma unroll
        for (int k = 0; k < kLsItems; ++k) {
            const uint32_t i = (uint32_t)t + (uint32_t)k * kLsThreads;
            if (i < avail) K[i] = kk[k], V[i] = vv[k];
        }
    }
    __syncthreads();
    const uint64_t kprev = pack64(s_kprev[0], s_kprev[1]);
    // sub-bucket starts among the keys in reach
    for (uint32_t i = (uint32_t)t; i < avail; i += kLsThreads) {
        const uint64_t idc = (K[i] & kmask) >> L;
        const bool start = i == 0 ? (g0 == 0 || ((kprev & kmask) >> L) != idc) : ((K[i - 1] & kmask) >> L) != idc;
        if (start) {
            atomicOr(&bnd[i >> 5], 1u << (i & 31u));
            if (i < span) atomicMin(&s_first, i);
            else atomicMin(&s_end, i);
        }
    }
    __syncthreads();
    uint32_t s = s_first, e = s_end;
    if (e == kNone && end_in_reach) e = avail; // (a last workgroup whose reach ends inside the span)
    if (s == kNone || s >= e) { // no sub-bucket starts in the span: nothing of its own (uniform)
        if (t == 0) tile_start[blockIdx.x] = 0, tile_cnt[blockIdx.x] = 0;
        return;
    }
    if (e == kNone) { // the sub-bucket across the span's end -- the last one that starts in the span -- is longer than the reach (uniform)
        if (t == 0) { // (rare: the last start below the span's end, from the bit array)
            uint32_t zz = s;
            for (int wd = (int)((span - 1u) >> 5); wd >= (int)(s >> 5); --wd) {
                uint32_t bits = bnd[wd];
                if ((uint32_t)wd == ((span - 1u) >> 5) && ((span - 1u) & 31u) != 31u) bits &= (2u << ((span - 1u) & 31u)) - 1u;
                if (bits) {
                    zz = (uint32_t)wd * 32u + (31u - (uint32_t)__clz(bits));
                    break;
                }
            }
            s_last = zz;
        }
        __syncthreads();
        const uint32_t z = s_last;
        if (t == 0) {
            // on the list of long sub-buckets (sx_long_subbuckets orders them); no list, or a full one: the whole sort falls back
            const uint32_t at = long_list ? atomicAdd(long_count, 1u) : long_cap;
            if (at < long_cap) long_list[at] = (uint32_t)(g0 + z);
            else atomicOr(fail, 1u);
            if (z == s) tile_start[blockIdx.x] = 0, tile_cnt[blockIdx.x] = 0;
        }
        if (z == s) return;
        e = z; // the sub-buckets in front of it are this workgroup's as ever
    }
    const uint32_t n_loc = e - s;
    // ---- sub-bucket rank of every owned pair: starts in (s, i], from the bit array ---------------------------------
    if (w == 0) { // starts before each word: one wave scans the kLsWords word counts
        uint32_t run = 0;
        for (int c0 = 0; c0 < kLsWords; c0 += kWave) {
            const int i = c0 + lane;
            const uint32_t cnt = i < kLsWords ? (uint32_t)__popc(bnd[i]) : 0u;
            const uint32_t inc = wave_inclusive_scan<OpAdd>(cnt);
            if (i < kLsWords) bpre[i] = run + inc - cnt;
            run += __shfl(inc, kWave - 1, kWave);
        }
    }
    __syncthreads();
    const uint32_t starts_to_s = bpre[s >> 5] + (uint32_t)__popc(bnd[s >> 5] & ((2u << (s & 31u)) - 1u)); // incl. the one at s
    const uint32_t last = e - 1u;
    const uint32_t nseg = bpre[last >> 5] + (uint32_t)__popc(bnd[last >> 5] & ((2u << (last & 31u)) - 1u)) - starts_to_s + 1u;
    // order by (rank, low L bits): that many bits, 8 a pass
    uint32_t sort_bits = L;
    for (uint32_t v = nseg - 1u; v; v >>= 1) ++sort_bits;
    const uint32_t npass = sort_bits ? (sort_bits + 7u) / 8u : 0u;
    const uint64_t lowmask = (1ull << L) - 1ull;
    // items: K[i] <- payload << 32 | rank << L | low bits (rank < 2^13, L <= 19: 32 bits), V[i] <- position
#pragma unroll
    for (int k = 0; k < kLsItems; ++k) {
        const uint32_t i = (uint32_t)t + (uint32_t)k * kLsThreads;
        if (i >= s && i < e) {
            const uint64_t key = K[i];
            const uint32_t rank = bpre[i >> 5] + (uint32_t)__popc(bnd[i >> 5] & ((2u << (i & 31u)) - 1u)) - starts_to_s;
            K[i] = ((key >> kbits) << 32) | ((uint64_t)rank << L) | (key & lowmask);
        }
    }
    __syncthreads();
    // ---- the owned pairs into registers, striped: q = wave * 768 + k * 64 + lane --------------------------------------
    const uint32_t q0 = (uint32_t)w * (kWave * kLsItems) + (uint32_t)lane;
    uint64_t key[kLsItems];
    uint32_t val[kLsItems];
#pragma unroll
    for (int k = 0; k < kLsItems; ++k) {
        const uint32_t q = q0 + (uint32_t)k * kWave;
        key[k] = q < n_loc ? K[s + q] : 0ull;
        val[k] = q < n_loc ? V[s + q] : 0u;
    }
    __syncthreads(); // K and V are free until the pairs are written back
    // ---- one counting pass + insertion inside the bins ---------------------------------------------------------------
    // The sort field's top bits (sub-bucket rank, then as many of the low bits as keep the number of bins within
    // kLsBins) spread the pairs over ~8192 bins: on a text whose prefixes are spread out a bin gets less than one pair
    // on average.  So the pairs are dropped into their bins in any order (LDS atomics: a count, a scan, a cursor each)
    // and every pair then finds its place among the one or two others of its bin by looking at their sort fields
    // (all 12 pairs of a thread step through their bins together, so that the LDS reads of a step overlap).  About a
    // third of the instructions of three stable 8-bit passes (the kernel is bound by instruction issue).  A
    // workgroup that finds a crowded bin (equal keys: the ties of a repeat) takes the stable passes below instead.
    uint32_t bb = 0;
    while (bb < L && ((uint64_t)nseg << (bb + 1u)) <= (uint64_t)kLsBins) ++bb;
    const uint32_t nb = nseg << bb, bshift = L - bb, nwords = nb / 2u + 1u; // (one bin more: the end of the last one)
    uint32_t *cw = reinterpret_cast<uint32_t *>(K);  // first slot of every bin, two 16-bit fields a word (a workgroup holds < 2^16 pairs)
    uint32_t *cur = cw + (kLsBins / 2 + 1);           // the same, moving: the cursors
    uint32_t *R = V;                                  // the sort fields in bin order
    bool counted = nb <= (uint32_t)kLsBins; // (uniform; false only with more sub-buckets than bins)
    if (counted) {
        for (uint32_t i = (uint32_t)t; i < nwords; i += kLsThreads) cw[i] = 0;
        if (t == 0) s_max = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kLsItems; ++k) {
            const uint32_t q = q0 + (uint32_t)k * kWave;
            if (q < n_loc) {
                const uint32_t bin = (uint32_t)key[k] >> bshift;
                atomicAdd(&cw[bin >> 1], 1u << (16u * (bin & 1u)));
            }
        }
        __syncthreads();
        { // counts -> first slots, 16 bins a thread; the fullest bin
            constexpr int kPer = kLsBins / 2 / kLsThreads;
            uint32_t wv[kPer], sum = 0, mx = 0;
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const uint32_t idx = (uint32_t)t * kPer + (uint32_t)j;
                wv[j] = idx < nwords ? cw[idx] : 0u;
                const uint32_t lo = wv[j] & 0xFFFFu, hi = wv[j] >> 16;
                mx = mx > lo ? mx : lo;
                mx = mx > hi ? mx : hi;
                sum += lo + hi;
            }
            const uint32_t inc = wave_inclusive_scan<OpAdd>(sum);
            if (lane == kWave - 1) s_scan[w] = inc;
            if (mx > (uint32_t)kLsMaxBin) atomicMax(&s_max, mx);
            __syncthreads();
            uint32_t run = inc - sum;
            for (int ww = 0; ww < w; ++ww) run += s_scan[ww];
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const uint32_t idx = (uint32_t)t * kPer + (uint32_t)j;
                const uint32_t lo = wv[j] & 0xFFFFu, hi = wv[j] >> 16;
                const uint32_t packed = run | ((run + lo) << 16);
                if (idx < nwords) cw[idx] = packed, cur[idx] = packed;
                run += lo + hi;
            }
            if (t == kLsThreads - 1 && (uint32_t)kLsThreads * kPer == nwords - 1u) cw[nwords - 1u] = run; // (nb == kLsBins: the end word)
        }
        __syncthreads();
        counted = s_max <= (uint32_t)kLsMaxBin; // uniform
    }
    if (counted) {
        uint32_t slot[kLsItems], span[kLsItems]; // where the pair was dropped; its bin: first slot | pairs << 16
#pragma unroll
        for (int k = 0; k < kLsItems; ++k) {
            const uint32_t q = q0 + (uint32_t)k * kWave;
            slot[k] = 0, span[k] = 0;
            if (q < n_loc) {
                const uint32_t bin = (uint32_t)key[k] >> bshift, sh = 16u * (bin & 1u);
                const uint32_t first = (cw[bin >> 1] >> sh) & 0xFFFFu;
                const uint32_t next = (cw[(bin + 1u) >> 1] >> (16u * ((bin + 1u) & 1u))) & 0xFFFFu;
                span[k] = first | ((next - first) << 16);
                slot[k] = (atomicAdd(&cur[bin >> 1], 1u << sh) >> sh) & 0xFFFFu;
            }
        }
#pragma unroll
        for (int k = 0; k < kLsItems; ++k) {
            const uint32_t q = q0 + (uint32_t)k * kWave;
            if (q < n_loc) R[slot[k]] = (uint32_t)key[k];
        }
        __syncthreads();
        // a pair's final slot: first slot of its bin + the pairs of the bin that sort before it (equal fields: by slot)
        uint32_t fin[kLsItems];
#pragma unroll
        for (int k = 0; k < kLsItems; ++k) fin[k] = span[k] & 0xFFFFu;
        for (uint32_t step = 0; step < (uint32_t)kLsMaxBin; ++step) {
            bool more = false;
#pragma unroll
            for (int k = 0; k < kLsItems; ++k) {
                const uint32_t len = span[k] >> 16;
                if (step < len) {
                    const uint32_t j = (span[k] & 0xFFFFu) + step, r2 = R[j], rel = (uint32_t)key[k];
                    fin[k] += (r2 < rel || (r2 == rel && j < slot[k])) ? 1u : 0u;
                    more = more || step + 1u < len;
                }
            }
            if (!__any(more ? 1 : 0)) break;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kLsItems; ++k) {
            const uint32_t q = q0 + (uint32_t)k * kWave;
            if (q < n_loc) K[fin[k]] = key[k], V[fin[k]] = val[k];
        }
        __syncthreads();
    } else {
        // ---- stable LSD passes inside the CU, 8 bits each (any distribution of the keys) ----------------------------
        uint32_t *wcount = reinterpret_cast<uint32_t *>(K); // kLsWaves x 256 counters: the key image is dead while the pairs are in registers
        if (t == 0) atomicOr(fail, 2u); // (statistics: some workgroup took the stable passes)
        for (uint32_t pass = 0; pass < npass; ++pass) { // uniform
            uint32_t lpos[kLsItems];
            if (pass > 0) {
#pragma unroll
                for (int k = 0; k < kLsItems; ++k) {
                    const uint32_t q = q0 + (uint32_t)k * kWave;
                    key[k] = q < n_loc ? K[q] : 0ull;
                    val[k] = q < n_loc ? V[q] : 0u;
                }
            }
            __syncthreads();
            for (int i = t; i < kLsWaves * 256; i += kLsThreads) wcount[i] = 0;
            __syncthreads();
            const uint32_t shift = 8u * pass;
#pragma unroll
            for (int k = 0; k < kLsItems; ++k) {
                const uint32_t q = q0 + (uint32_t)k * kWave;
                const uint32_t d = ((uint32_t)key[k] >> shift) & 0xFFu;
                lpos[k] = wave_rank_inorder<8, false>(d, q < n_loc, wcount + w * 256) | (d << 16);
            }
            __syncthreads();
            {
                uint32_t tot = 0;
                if (t < 256) {
#pragma unroll
                    for (int ww = 0; ww < kLsWaves; ++ww) {
                        const uint32_t x = wcount[ww * 256 + t];
                        wcount[ww * 256 + t] = tot;
                        tot += x;
                    }
                }
                const uint32_t inc = wave_inclusive_scan<OpAdd>(tot);
                if (lane == kWave - 1) s_scan[w] = inc;
                __syncthreads();
                uint32_t base = 0;
                for (int ww = 0; ww < w; ++ww) base += s_scan[ww];
                const uint32_t ex = base + inc - tot;
                if (t < 256) {
#pragma unroll
                    for (int ww = 0; ww < kLsWaves; ++ww) wcount[ww * 256 + t] += ex;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kLsItems; ++k) lpos[k] = (lpos[k] & 0xFFFFu) + wcount[w * 256 + (lpos[k] >> 16)];
            __syncthreads(); // the counters are part of the key image
#pragma unroll
            for (int k = 0; k < kLsItems; ++k) {
                const uint32_t q = q0 + (uint32_t)k * kWave;
                if (q < n_loc) K[lpos[k]] = key[k], V[lpos[k]] = val[k];
            }
            __syncthreads();
        }
        if (npass == 0) { // (a single pair, or all sort fields empty: back as they came)
#pragma unroll
            for (int k = 0; k < kLsItems; ++k) {
                const uint32_t q = q0 + (uint32_t)k * kWave;
                if (q < n_loc) K[q] = key[k], V[q] = val[k];
            }
            __syncthreads();
        }
    }
    const uint32_t base_i = 0;
    // ---- out: positions, windows, and the members of groups of equal keys (compacted in order) ---------------------
    const uint64_t gs = g0 + s;
    uint32_t tied_mask = 0, head_mask = 0, wave_total = 0; // bit k: pair k of this lane is tied / opens its group
    uint32_t pos[kLsItems];
#pragma unroll
    for (int k = 0; k < kLsItems; ++k) {
        const uint32_t q = q0 + (uint32_t)k * kWave;
        bool tied = false, head = false;
        pos[k] = 0;
        if (q < n_loc) {
            const uint64_t it = K[base_i + q];
            const uint32_t rel = (uint32_t)it;
            pos[k] = V[base_i + q];
            const bool eq_prev = q > 0 && (uint32_t)K[base_i + q - 1] == rel;
            const bool eq_next = q + 1 < n_loc && (uint32_t)K[base_i + q + 1] == rel;
            tied = eq_prev || eq_next;
            head = !eq_prev;
            vout[gs + q] = pos[k];
            if (seedw) seedw[gs + q] = (uint32_t)(it >> 32);
        }
        if (tied) tied_mask |= 1u << k;
        if (head) head_mask |= 1u << k;
        wave_total += (uint32_t)__popcll(__ballot(tied ? 1 : 0));
    }
    if (lane == 0) s_scan[w] = wave_total;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
#pragma unroll
    for (int ww = 0; ww < kLsWaves; ++ww) {
        const uint32_t x = s_scan[ww];
        if (ww < w) wave_base += x;
        total += x;
    }
    if (t == 0) tile_start[blockIdx.x] = (uint32_t)gs, tile_cnt[blockIdx.x] = total;
    if (total) { // uniform
        uint32_t run = wave_base;
#pragma unroll
        for (int k = 0; k < kLsItems; ++k) {
            const bool tied = (tied_mask >> k) & 1u;
            const uint64_t bal = __ballot(tied ? 1 : 0);
            if (tied) {
                const uint32_t q = q0 + (uint32_t)k * kWave;
                const uint64_t at = gs + run + (uint32_t)__popcll(bal & lanemask_lt());
                uint2 ent;
                ent.x = (uint32_t)(gs + q), ent.y = pos[k];
                stage[at] = ent;
                stage_head[at] = (uint8_t)((head_mask >> k) & 1u);
            }
            run += (uint32_t)__popcll(bal);
        }
    }
}

// every workgroup's tied members -> their place in the global list (the offsets are a scan of tile_cnt)
__global__ __launch_bounds__(kBlock) void local_tied_gather_kernel(const uint32_t *__restrict__ tile_start,
                                                                   const uint32_t *__restrict__ tile_cnt,
                                                                   const uint32_t *__restrict__ tile_off,
                                                                   const uint2 *__restrict__ stage,
                                                                   const uint8_t *__restrict__ stage_head,
                                                                   uint32_t *__restrict__ apos, uint32_t *__restrict__ ap,
                                                                   uint8_t *__restrict__ ahead, uint32_t cap)
{
    const uint32_t tile = blockIdx.x, cnt = tile_cnt[tile], off = tile_off[tile];
    const uint64_t start = tile_start[tile];
    for (uint32_t i = threadIdx.x; i < cnt; i += kBlock) {
        const uint32_t slot = off + i;
        if (slot >= cap) break;
        const uint2 ent = stage[start + i];
        apos[slot] = ent.x;
        ap[slot] = ent.y;
        ahead[slot] = stage_head[start + i];
    }
}

// ---- long sub-buckets --------------------------------------------------------------------------------------------------
// where each listed sub-bucket ends: the pairs are ordered by their top bits, so the first pair behind `start` with other
// top bits is found by bisection (a thread a sub-bucket)
__global__ __launch_bounds__(kBlock) void long_ends_kernel(const uint64_t *__restrict__ kin, uint64_t m, uint32_t L, uint32_t kbits,
                                                           const uint32_t *__restrict__ starts, uint32_t n_long,
                                                           uint32_t *__restrict__ lens)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_long) return;
    const uint64_t kmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
    const uint64_t start = starts[i], id = (kin[start] & kmask) >> L;
    uint64_t lo = start + 1, hi = m; // the end lies in [lo, hi]
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (((kin[mid] & kmask) >> L) == id) lo = mid + 1;
        else hi = mid;
    }
    lens[i] = (uint32_t)(lo - start);
}

// the sub-bucket of slot e of the gathered array: the last one whose offset is <= e
__device__ __forceinline__ uint32_t long_find(const uint32_t *__restrict__ offs, uint32_t n_long, uint32_t e)
{
    uint32_t lo = 0, hi = n_long; // offs[lo] <= e < offs[hi] (offs[n_long] = the total, not stored)
    while (hi - lo > 1u) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (offs[mid] <= e) lo = mid;
        else hi = mid;
    }
    return lo;
}

// members of the long sub-buckets, one behind the other; the key's top bits (equal inside a sub-bucket) make way for the
// sub-bucket's number on the list, so that HBM passes over (number, low bits) order every sub-bucket by its low bits
__global__ __launch_bounds__(kBlock) void long_gather_kernel(const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                             uint32_t L, uint32_t kbits, const uint32_t *__restrict__ starts,
                                                             const uint32_t *__restrict__ offs, uint32_t n_long, uint32_t total,
                                                             uint64_t *__restrict__ ck, uint32_t *__restrict__ cv)
{
    const uint32_t e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= total) return;
    const uint32_t idx = long_find(offs, n_long, e);
    const uint64_t g = (uint64_t)starts[idx] + (e - offs[idx]);
    const uint64_t kmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull), lowmask = (1ull << L) - 1ull;
    const uint64_t key = kin[g];
    ck[e] = (key & ~kmask) | ((uint64_t)idx << L) | (key & lowmask);
    cv[e] = vin[g];
}

// what local_sort_kernel writes for its pairs, from the gathered array in (number, low bits) order: positions, windows,
// and the members of groups of equal keys behind the `tied0` members the workgroups listed (device_compact's functors)
struct InLongTied {
    const uint64_t *k;
    uint64_t total, cmp; // cmp: the bits of (number, low bits)
    __device__ __forceinline__ bool operator()(uint64_t i) const
    {
        const uint64_t x = k[i] & cmp;
        return (i > 0 && (k[i - 1] & cmp) == x) || (i + 1 < total && (k[i + 1] & cmp) == x);
    }
};
struct OutLongMember {
    const uint64_t *k;
    const uint32_t *v, *starts, *offs;
    uint64_t cmp;
    uint32_t L, kbits, idmask, tied0, cap;
    uint32_t *vout, *seedw, *apos, *ap;
    uint8_t *ahead;
    __device__ __forceinline__ void operator()(uint64_t i, uint32_t dst, bool tied) const
    {
        const uint64_t key = k[i];
        const uint32_t idx = (uint32_t)(key >> L) & idmask, pos = v[i];
        const uint64_t g = (uint64_t)starts[idx] + ((uint32_t)i - offs[idx]);
        vout[g] = pos;
        if (seedw) seedw[g] = (uint32_t)(key >> kbits);
        if (tied) {
            const uint64_t slot = (uint64_t)tied0 + dst;
            if (slot < cap) {
                apos[slot] = (uint32_t)g;
                ap[slot] = pos;
                ahead[slot] = (uint8_t)((i > 0 && (k[i - 1] & cmp) == (key & cmp)) ? 0u : 1u);
            }
        }
    }
};

} // namespace sx

using namespace sx;

uint32_t sx_local_sort_tiles(uint64_t m) { return sx_div_up(m, kLsSpan); }

// The long sub-buckets local_sort_kernel listed (n_long starts in long_list; long_list holds 3 * long_cap words: starts,
// lengths, offsets).  *done = 0 when they hold more pairs than `max_pairs` (the scratch arrays' size; the caller falls back
// to plain passes); else every output of sx_local_sort is complete, *tied_total = tied0 + their tied members.
int sx_long_subbuckets(sx_ctx *ctx, const uint64_t *kin, const uint32_t *vin, uint64_t m, int kbits, int top_bits, uint32_t n_long,
                       uint32_t *long_list, uint32_t long_cap, uint64_t *ck_a, uint64_t *ck_b, uint32_t *cv_a, uint32_t *cv_b,
                       uint32_t max_pairs, uint32_t *vout, uint32_t *seedw, uint32_t *apos, uint32_t *ap, uint8_t *ahead, uint32_t cap,
                       uint32_t tied0, uint32_t *d_scalar2 /* two words of device scratch */, uint32_t *tied_total, uint32_t *pairs,
                       int *done)
{
    *done = 0;
    *pairs = 0;
    *tied_total = tied0;
    if (n_long == 0 || n_long > long_cap) return 0;
    const uint32_t L = (uint32_t)(kbits - top_bits);
    uint32_t *starts = long_list, *lens = long_list + long_cap, *offs = long_list + 2 * (size_t)long_cap;
    sx_launch(ctx, SX_KC_LOCAL_SORT, (uint64_t)n_long * 256, long_ends_kernel, dim3(sx_div_up(n_long, kBlock)), dim3(kBlock), kin, m, L,
              (uint32_t)kbits, (const uint32_t *)starts, n_long, lens);
    SX_TRY((device_scan<OpAdd>(ctx, n_long, InU32{lens}, OutExclusive{offs}, d_scalar2, SX_KC_LOCAL_SORT, 0)));
    uint32_t total = 0;
    SX_TRY(sx_readback(ctx, d_scalar2, 1, &total));
    *pairs = total;
    if (total == 0 || total > max_pairs) return 0;
    sx_launch(ctx, SX_KC_LOCAL_SORT, (uint64_t)total * 24, long_gather_kernel, dim3(sx_div_up(total, kBlock)), dim3(kBlock), kin, vin, L,
              (uint32_t)kbits, (const uint32_t *)starts, (const uint32_t *)offs, n_long, total, ck_a, cv_a);
    const int idbits = n_long > 1 ? sx_bitlen(n_long - 1) : 1;
    int f = 0;
    SX_TRY(sx_sort_pairs(ctx, ck_a, cv_a, ck_b, cv_b, total, 0, (int)L + idbits, &f));
    const uint64_t *ks = f ? ck_b : ck_a;
    const uint32_t *vs = f ? cv_b : cv_a;
    const uint64_t cmp = (1ull << (L + (uint32_t)idbits)) - 1ull;
    SX_TRY((device_compact(ctx, total, InLongTied{ks, total, cmp},
                           OutLongMember{ks, vs, starts, offs, cmp, L, (uint32_t)kbits, (1u << idbits) - 1u, tied0, cap, vout, seedw, apos, ap, ahead},
                           d_scalar2 + 1, SX_KC_LOCAL_SORT, (uint64_t)total * 28)));
    uint32_t tied = 0;
    SX_TRY(sx_readback(ctx, d_scalar2 + 1, 1, &tied));
    *tied_total = tied0 + tied;
    *done = 1;
    return 0;
}

bool sx_local_sort_applies(uint64_t m, int kbits, int top_bits)
{
    // the top `top_bits` go through HBM passes, the low L = kbits - top_bits bits (with up to 13 bits of sub-bucket rank)
    // must fit the 32-bit sort field of an LDS item, the payload the other 32
    return m >= 1 && kbits >= 32 && kbits - top_bits >= 1 && kbits - top_bits <= 19;
}

int sx_local_sort(sx_ctx *ctx, const uint64_t *kin, const uint32_t *vin, uint64_t m, int kbits, int top_bits, uint32_t *vout,
                  uint32_t *seedw, uint32_t *tile_start, uint32_t *tile_cnt, uint32_t *tile_off, uint2 *stage,
                  uint8_t *stage_head, uint32_t *apos, uint32_t *ap, uint8_t *ahead, uint32_t cap,
                  uint32_t *d_total_and_fail /* [0] <- tied members, [1] <- bit 0: a sub-bucket did not fit, bit 1: stable passes were used,
                                                [2] <- long sub-buckets listed */,
                  uint32_t longest_expected, uint32_t *long_list, uint32_t long_cap)
{
    // The span: a workgroup's costs that do not depend on its pairs (zeroing and scanning 8192 counters, the barriers) are
    // spread over more pairs the longer it is (1 GiB of DNA, dense keys: span 5120 2.79 ms, 5632 2.60, 5888 2.52), but a
    // sub-bucket that starts at the span's last pair must end within the reach of kLsCap pairs, or the whole sort falls back
    // to plain passes.  So the span grows only where the caller knows how long sub-buckets get (four-letter texts with
    // dense keys: uniform symbols at a known rate; longest_expected = 0: not known), with a factor of safety.
    uint32_t span = (uint32_t)kLsSpan;
    if (longest_expected > 0 && 3 * (uint64_t)longest_expected <= 2 * ((uint64_t)kLsCap - 5632u)) span = 5632u;
    else if (longest_expected > 0 && 3 * (uint64_t)longest_expected <= 2 * ((uint64_t)kLsCap - 5376u)) span = 5376u;
    if (span < (uint32_t)kLsSpan) span = (uint32_t)kLsSpan;
    const uint32_t tiles = sx_div_up(m, span);
    const uint32_t L = (uint32_t)(kbits - top_bits);
    SX_CHECK(hipMemsetAsync(d_total_and_fail + 1, 0, 2 * sizeof(uint32_t), ctx->stream));
    sx_launch(ctx, SX_KC_LOCAL_SORT, m * (12 + 4 + (seedw ? 4 : 0)), local_sort_kernel, dim3(tiles), dim3(kLsThreads), kin, vin, m, L,
              (uint32_t)kbits, vout, seedw, tile_start, tile_cnt, stage, stage_head, d_total_and_fail + 1, span, long_list,
              d_total_and_fail + 2, long_list ? long_cap : 0u);
    SX_TRY((device_scan<OpAdd>(ctx, tiles, InU32{tile_cnt}, OutExclusive{tile_off}, d_total_and_fail, SX_KC_NAMES, 0)));
    sx_launch(ctx, SX_KC_NAMES, 0, local_tied_gather_kernel, dim3(tiles), dim3(kBlock), (const uint32_t *)tile_start,
              (const uint32_t *)tile_cnt, (const uint32_t *)tile_off, (const uint2 *)stage, (const uint8_t *)stage_head, apos, ap,
              ahead, cap);
    return 0;
}
