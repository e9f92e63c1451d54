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
constexpr int kLsWords = kLsCap / 32;
#ifndef SX_LS_BINS
#define SX_LS_BINS 8192
#endif
constexpr int kLsBins = SX_LS_BINS;  // bins of the counting pass: 16 KiB of packed 16-bit counters in the (then unused) key image
#ifndef SX_LS_MAXBIN
#define SX_LS_MAXBIN 16
#endif
constexpr int kLsMaxBin = SX_LS_MAXBIN;  // a bin with more pairs than this: stable passes instead (equal keys crowd one bin)
static_assert(kLsBins / 2 % kLsThreads == 0 && (kLsBins / 2 + 1) * 8 <= kLsCap * 8, "counter words per thread; both counter sets fit the key image");

// One workgroup: local indices i = global index - g0, g0 = blockIdx.x * kLsSpan.
//   s = first sub-bucket start at i >= 0, e = first sub-bucket start (or the end of the array) at i >= kLsSpan;
//   the workgroup owns [s, e).  Every sub-bucket starts in exactly one span, so the owned ranges tile the array.
// kin/vin: pairs ordered by (key & kmask) >> L; the key bits from kbits on are payload (the symbol window).
// Round 5: this is the kernel of rounds 3 and 4, kept for the workgroups whose pairs the lean kernel below cannot order
// (crowded bins: many equal keys, a repeat's ties; more sub-buckets than bins): those leave kLsRedo in tile_start and are
// done again here, with the stable passes, by a launch the host queues only when some workgroup asked for it.
constexpr uint32_t kLsRedo = 0xFFFFFFFFu;
__global__ __launch_bounds__(kLsThreads, 4) void local_sort_redo_kernel(
    const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint64_t m, uint32_t L, uint32_t kbits,
    uint32_t *__restrict__ vout, uint32_t *__restrict__ seedw /* or null */, uint32_t *__restrict__ tile_start,
    uint32_t *__restrict__ tile_cnt, uint2 *__restrict__ stage, uint8_t *__restrict__ stage_head,
    uint32_t *__restrict__ fail, uint32_t span /* kLsSpan, or more where the sub-buckets are known to be short */,
    uint32_t *__restrict__ long_list /* or null */, uint32_t *__restrict__ long_count, uint32_t long_cap)
{
    // (a long sub-bucket across this span's end has been listed by the lean kernel already: long_cap == 0 here means
    //  "do not list", not "no list")
    if (tile_start[blockIdx.x] != kLsRedo) return; // (uniform: the lean kernel finished this workgroup's pairs)
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
            if (long_cap != 0 || !long_list) { // (see above)
                const uint32_t at = long_list ? atomicAdd(long_count, 1u) : long_cap;
                if (at < long_cap) long_list[at] = (uint32_t)(g0 + z);
                else atomicOr(fail, 1u);
            }
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


// ---- the lean kernel (round 5) ------------------------------------------------------------------------------------------
// The same job with the pairs kept in registers from the load to the stores.  Rounds 3 and 4 wrote the keys to LDS, found
// the sub-bucket starts there, rewrote every key as (payload, rank, low bits), read the pairs back striped, and after the
// ranking wrote them to LDS once more to read them a last time in order: 14 barriers, ~25 LDS operations a pair, 72 KiB a
// workgroup (two a CU, four waves a SIMD), 2.4 TB/s -- bound by the latencies between its barriers, not by instruction issue
// (2300 vector instructions a wave) nor by memory.  Here:
//   * a thread loads its pairs in the order it keeps them (wave w: indices w * 64 * ITEMS + k * 64 + lane), a pair's
//     neighbour is the lane beside it: sub-bucket starts are a compare with a DPP-shifted copy, their bit array twelve ballots
//     in scalar registers, a pair's sub-bucket rank a popcount;
//   * one array of 16-bit counters: counts, their scan, and cursors that the drops advance -- after the drops counter b holds
//     the end of bin b, which is the start of bin b + 1;
//   * a pair learns its final slot from the sort fields of its bin's other pairs (R, 4 bytes a pair: the only LDS image) and
//     whether it is tied in the same look; positions and windows go straight to their places in HBM (a wave's 64 stores land
//     in the few sub-buckets its lanes belong to: the same cache lines as stores in order);
//   * tied pairs (under one in a hundred) leave their position in R at their final slot and a bit in two small bit arrays, from
//     which they are listed in order.
//   * a crowded bin (more than kLsMaxBin pairs: a repeat's equal and nearly equal keys) is ordered by waves, 64 of its pairs
//     each, counting per distinct value through ballots (below) -- a genome's repeat families put one into four workgroups of ten.
// 56 KiB of LDS, 8 barriers (9 with crowded bins).  A workgroup that meets a bin of more than SX_LS2_TEAM_MAX pairs or has more
// sub-buckets than bins leaves its pairs to the kernel above (kLsRedo in tile_start, bit 2 of *fail: the host queues that launch).
#ifdef SX_LS_PROBE // (diagnostic build: cycles of the lean kernel's phases, summed over the workgroups' first waves; tools/ab_macros.sh)
__device__ unsigned long long sx_ls_probe[24];
#define LS_PROBE(i)                                                                  \
    do {                                                                             \
        const unsigned long long now_ = (unsigned long long)clock64();               \
        probe_d[i] = (unsigned)(now_ - probe_t);                                     \
        probe_t = now_;                                                              \
    } while (0)
#else
#define LS_PROBE(i) ((void)0)
#endif
#ifndef SX_LS2_THREADS
#define SX_LS2_THREADS 1024
#endif
#ifndef SX_LS2_CAP
#define SX_LS2_CAP 6144 // pairs in a workgroup's reach (A/B: 3072 with 512 threads and 4096 bins -- four workgroups a CU)
#endif
#ifndef SX_LS2_BINS
#define SX_LS2_BINS SX_LS_BINS
#endif
#ifndef SX_LS2_UNROLL
#define SX_LS2_UNROLL 2 // bin members an owner compares itself with a step (1 GiB of DNA: one 1.87 ms, two 1.75, four 1.77)
#endif
#ifndef SX_LS2_TEAM_MAX
#define SX_LS2_TEAM_MAX 512 // the fullest bin a workgroup orders itself (by waves, below); beyond: left to the other kernel
#endif
constexpr int kL2Cap = SX_LS2_CAP, kL2Words = kL2Cap / 32, kL2Bins = SX_LS2_BINS, kL2Unroll = SX_LS2_UNROLL;
constexpr int kL2TeamMax = SX_LS2_TEAM_MAX, kL2Jobs = kL2Cap / (kLsMaxBin + 1) + kL2Cap / kWave + 1; // (a job: 64 members of a crowded bin)
constexpr int kL2Threads = SX_LS2_THREADS, kL2Waves = kL2Threads / kWave, kL2Items = kL2Cap / kL2Threads, kL2PerWave = kWave * kL2Items;
#ifndef SX_LS2_MINWAVES
#define SX_LS2_MINWAVES (SX_LS2_THREADS >= 1024 ? 8 : 4)
#endif
constexpr int kL2MinWaves = SX_LS2_MINWAVES; // (waves a SIMD the register budget is cut for)
static_assert(kL2Threads * kL2Items == kL2Cap && kL2Bins / 2 % kL2Threads == 0 && kL2Cap <= kLsCap, "the lean kernel's shape");
static_assert(kL2Cap < (1 << 13) && kL2TeamMax <= 64 * kWave, "a job word: 13 bits of first slot, 13 of members, 6 of piece");
__global__ __launch_bounds__(kL2Threads, kL2MinWaves) void local_sort_kernel(
    const uint64_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint64_t m, uint32_t L, uint32_t kbits,
    uint32_t *__restrict__ vout, uint32_t *__restrict__ seedw /* or null */, uint32_t *__restrict__ tile_start,
    uint32_t *__restrict__ tile_cnt, uint2 *__restrict__ stage, uint8_t *__restrict__ stage_head,
    uint32_t *__restrict__ fail, uint32_t span, uint32_t *__restrict__ long_list /* or null */, uint32_t *__restrict__ long_count,
    uint32_t long_cap)
{
    __shared__ uint32_t cw[kL2Bins / 2 + 2];  // packed 16-bit counters: counts, then first slots, then (after the drops) ends
    __shared__ uint32_t R[kL2Cap + kL2Unroll]; // sort fields in bin order; then the positions of tied pairs at their final slots
    __shared__ uint16_t team_fin[kL2Cap];     // crowded bins: final slot | tied << 13 | not the group's head << 14, by the slot a pair dropped into
    __shared__ uint32_t jobs[kL2Jobs], s_jobs; // crowded bins in pieces of 64 members: first slot | members << 13 | piece << 26
    __shared__ uint32_t tiedb[kL2Words], headb[kL2Words], tpre[kL2Words];
    __shared__ uint32_t s_first, s_end, s_last, s_max, s_scan[kL2Waves], s_starts[kL2Waves];
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t g0 = (uint64_t)blockIdx.x * span;
    const uint64_t kmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
    const uint32_t avail = m - g0 < (uint64_t)kL2Cap ? (uint32_t)(m - g0) : (uint32_t)kL2Cap; // pairs in reach
    const bool end_in_reach = g0 + avail == m;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const uint32_t i0 = (uint32_t)w * kL2PerWave + (uint32_t)lane; // index (from g0) of this thread's pair 0; pair k: + 64 k
#ifdef SX_LS_PROBE
    unsigned long long probe_t = (unsigned long long)clock64();
    unsigned probe_d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // ---- the pairs, and the key in front of the wave's first ----------------------------------------------------------------
    uint64_t key[kL2Items];
#pragma unroll
    for (int k = 0; k < kL2Items; ++k) {
        const uint32_t i = i0 + (uint32_t)k * kWave;
        key[k] = i < avail ? __builtin_nontemporal_load(kin + g0 + i) : 0ull;
    }
    const uint64_t wave_at = g0 + (uint64_t)w * kL2PerWave;
    const uint64_t kfront = (wave_at > 0 && wave_at <= m) ? kin[wave_at - 1] : 0ull;
    // (while the loads are in flight: the counters and the bit arrays)
    for (int i = t; i < kL2Bins / 2 + 2; i += kL2Threads) cw[i] = 0;
    for (int i = t; i < kL2Words; i += kL2Threads) tiedb[i] = 0, headb[i] = 0;
    if (t == 0) s_first = kNone, s_end = kNone, s_last = 0, s_max = 0, s_jobs = 0;
    __syncthreads();
    LS_PROBE(0);
    // ---- sub-bucket starts: ballots ------------------------------------------------------------------------------------------
    uint64_t sm[kL2Items]; // bit l of sm[k]: pair k of lane l opens a sub-bucket (uniform over the wave)
    uint32_t n_starts = 0;
    {
        uint32_t before = (uint32_t)((kfront & kmask) >> L); // the sub-bucket of the pair in front of lane 0's
#pragma unroll
        for (int k = 0; k < kL2Items; ++k) {
            const uint32_t i = i0 + (uint32_t)k * kWave;
            const uint32_t idc = (uint32_t)((key[k] & kmask) >> L);
            uint32_t prev = __shfl_up(idc, 1u);
            if (lane == 0) prev = before;
            const bool start = i < avail && (g0 + i == 0 || prev != idc);
            sm[k] = __ballot(start ? 1 : 0);
            n_starts += (uint32_t)__popcll(sm[k]);
            before = (uint32_t)__builtin_amdgcn_readlane((int)idc, kWave - 1);
        }
    }
    if (lane == 0) {
        s_starts[w] = n_starts;
        // the first start in the span, the first at or behind the span's end (this wave's candidates)
        uint32_t fs = kNone, fe = kNone;
#pragma unroll
        for (int k = 0; k < kL2Items; ++k) {
            const uint32_t base = (uint32_t)w * kL2PerWave + (uint32_t)k * kWave;
            uint64_t below = sm[k], above = sm[k];
            if (base >= span) below = 0;
            else if (span - base < 64u) below &= (1ull << (span - base)) - 1ull, above &= ~((1ull << (span - base)) - 1ull);
            else above = 0;
            if (below && fs == kNone) fs = base + (uint32_t)__ffsll((unsigned long long)below) - 1u;
            if (above && fe == kNone) fe = base + (uint32_t)__ffsll((unsigned long long)above) - 1u;
        }
        if (fs != kNone) atomicMin(&s_first, fs);
        if (fe != kNone) atomicMin(&s_end, fe);
    }
    __syncthreads();
    LS_PROBE(1);
    const uint32_t s = s_first;
    uint32_t e = s_end;
    if (e == kNone && end_in_reach) e = avail; // (a last workgroup whose reach ends inside the span)
    if (s == kNone || s >= e) { // no sub-bucket starts in the span: nothing of its own (uniform)
        if (t == 0) tile_start[blockIdx.x] = 0, tile_cnt[blockIdx.x] = 0;
        return;
    }
    if (e == kNone) { // the sub-bucket across the span's end is longer than the reach (uniform; rare): the last start below the span's end
        if (lane == 0) {
            uint32_t z1 = 0; // 1 + the wave's last start below the span's end
#pragma unroll
            for (int k = 0; k < kL2Items; ++k) {
                const uint32_t base = (uint32_t)w * kL2PerWave + (uint32_t)k * kWave;
                uint64_t below = sm[k];
                if (base >= span) below = 0;
                else if (span - base < 64u) below &= (1ull << (span - base)) - 1ull;
                if (below) z1 = base + 64u - (uint32_t)__clzll((unsigned long long)below);
            }
            if (z1) atomicMax(&s_last, z1);
        }
        __syncthreads();
        const uint32_t z = s_last - 1u; // (s itself is such a start: s_last >= s + 1)
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
    // ---- sort fields: (sub-bucket rank, low bits); the rank of a pair = the starts up to it, less the one at s ------------------
    uint32_t starts_before = 0, starts_all = 0;
#pragma unroll
    for (int ww = 0; ww < kL2Waves; ++ww) {
        const uint32_t x = s_starts[ww];
        if (ww < w) starts_before += x;
        starts_all += x;
    }
    // (bins by the starts in the whole reach, an upper bound of the owned sub-buckets: at most a few more than those)
    const uint32_t nseg = starts_all;
    uint32_t bb = 0;
    while (bb < L && ((uint64_t)nseg << (bb + 1u)) <= (uint64_t)kL2Bins) ++bb;
    const uint32_t nb = nseg << bb, bshift = L - bb;
    if (nb > (uint32_t)kL2Bins) { // more sub-buckets than bins (uniform): the other kernel's stable passes
        if (t == 0) tile_start[blockIdx.x] = kLsRedo, tile_cnt[blockIdx.x] = 0, atomicOr(fail, 4u);
#ifdef SX_LS_PROBE
        if (t == 0) atomicAdd(&sx_ls_probe[8], 1ull);
#endif
        return;
    }
    const uint32_t lowmask = L >= 32 ? ~0u : ((1u << L) - 1u);
    uint32_t f[kL2Items], pay[kL2Items], own = 0; // sort field, window, bit k: pair k is this workgroup's
    {
        uint32_t run = starts_before;
#pragma unroll
        for (int k = 0; k < kL2Items; ++k) {
            const uint32_t i = i0 + (uint32_t)k * kWave;
            const uint32_t rank = run + (uint32_t)__popcll(sm[k] & lanemask_le()) - 1u;
            f[k] = (rank << L) | ((uint32_t)key[k] & lowmask);
            pay[k] = (uint32_t)(key[k] >> kbits);
            if (i >= s && i < e) own |= 1u << k;
            run += (uint32_t)__popcll(sm[k]);
        }
    }
    // ---- counts -> first slots ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < kL2Items; ++k)
        if ((own >> k) & 1u) {
            const uint32_t bin = f[k] >> bshift;
            atomicAdd(&cw[bin >> 1], 1u << (16u * (bin & 1u)));
        }
    __syncthreads();
    LS_PROBE(2);
    {
        constexpr int kPer = kL2Bins / 2 / kL2Threads;
        uint32_t wv[kPer], sum = 0, mx = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            wv[j] = cw[(uint32_t)t * kPer + (uint32_t)j];
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
            const uint32_t lo = wv[j] & 0xFFFFu, hi = wv[j] >> 16;
            cw[(uint32_t)t * kPer + (uint32_t)j] = run | ((run + lo) << 16);
            if (mx > (uint32_t)kLsMaxBin) { // (rare) crowded bins: equal keys, a repeat's ties -- jobs for the waves, 64 members each
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t len = h ? hi : lo, first = h ? run + lo : run;
                    if (len > (uint32_t)kLsMaxBin) {
                        const uint32_t pieces = (len + kWave - 1u) / kWave, at = atomicAdd(&s_jobs, pieces);
#ifdef SX_LS_PROBE
                        atomicAdd(&sx_ls_probe[len <= 32 ? 11 : len <= 64 ? 12 : len <= 128 ? 13 : len <= 256 ? 14 : 15], (unsigned long long)len | (1ull << 40));
#endif
                        for (uint32_t pc = 0; pc < pieces && at + pc < (uint32_t)kL2Jobs; ++pc) jobs[at + pc] = first | (len << 13) | (pc << 26);
                    }
                }
            }
            run += lo + hi;
        }
    }
    __syncthreads();
    LS_PROBE(3);
    if (s_max > (uint32_t)kL2TeamMax) { // a bin too crowded for that too: the other kernel's stable passes (uniform)
        if (t == 0) tile_start[blockIdx.x] = kLsRedo, tile_cnt[blockIdx.x] = 0, atomicOr(fail, 4u);
#ifdef SX_LS_PROBE
        if (t == 0) atomicAdd(&sx_ls_probe[9], 1ull), atomicMax(&sx_ls_probe[10], (unsigned long long)s_max);
#endif
        return;
    }
    // ---- the pairs drop into their bins (any order) ------------------------------------------------------------------------
    uint32_t sf[kL2Items]; // where the pair was dropped | its final slot << 16 (a workgroup holds < 2^16 pairs)
#pragma unroll
    for (int k = 0; k < kL2Items; ++k) {
        sf[k] = 0;
        if ((own >> k) & 1u) {
            const uint32_t bin = f[k] >> bshift, sh = 16u * (bin & 1u);
            sf[k] = (atomicAdd(&cw[bin >> 1], 1u << sh) >> sh) & 0xFFFFu;
        }
    }
#pragma unroll
    for (int k = 0; k < kL2Items; ++k)
        if ((own >> k) & 1u) R[sf[k]] = f[k];
    __syncthreads();
    LS_PROBE(4);
    // (the positions are asked for now -- a third of the bytes, needed for the stores only: they arrive while the slots are found)
    uint32_t val[kL2Items];
#pragma unroll
    for (int k = 0; k < kL2Items; ++k) {
        const uint32_t i = i0 + (uint32_t)k * kWave;
        val[k] = ((own >> k) & 1u) ? __builtin_nontemporal_load(vin + g0 + i) : 0u;
    }
    // ---- a pair's final slot: the first slot of its bin + the pairs of the bin that sort before it (equal fields: by slot) ----
    uint32_t bspan[kL2Items], tied = 0, nothead = 0, team = 0; // the pair's bin: first slot | pairs << 16
#pragma unroll
    for (int k = 0; k < kL2Items; ++k) {
        bspan[k] = 0;
        if ((own >> k) & 1u) {
            const uint32_t bin = f[k] >> bshift;
            const uint32_t end = (cw[bin >> 1] >> (16u * (bin & 1u))) & 0xFFFFu; // (the cursor has reached the bin's end)
            const uint32_t first = bin ? (cw[(bin - 1u) >> 1] >> (16u * ((bin - 1u) & 1u))) & 0xFFFFu : 0u;
            if (end - first > (uint32_t)kLsMaxBin) team |= 1u << k; // a crowded bin: placed by the waves' jobs below
            else bspan[k] = first | ((end - first) << 16), sf[k] |= first << 16;
        }
    }
#pragma unroll 1
    for (uint32_t step = 0; step < (uint32_t)kLsMaxBin; step += (uint32_t)kL2Unroll) {
        bool more = false;
#pragma unroll
        for (int k = 0; k < kL2Items; ++k) {
            const uint32_t len = bspan[k] >> 16;
            if (step < len) {
                const uint32_t j0 = (bspan[k] & 0xFFFFu) + step, dropped = sf[k] & 0xFFFFu;
                uint32_t r2[kL2Unroll];
#pragma unroll
                for (int u = 0; u < kL2Unroll; ++u) r2[u] = R[j0 + (uint32_t)u]; // (R is padded: what lies behind the bin is masked below)
#pragma unroll
                for (int u = 0; u < kL2Unroll; ++u) {
                    const uint32_t j = j0 + (uint32_t)u;
                    const bool in = step + (uint32_t)u < len, eq = in && r2[u] == f[k];
                    sf[k] += ((in && r2[u] < f[k]) || (eq && j < dropped)) ? 0x10000u : 0u;
                    if (eq && j != dropped) tied |= 1u << k;
                    if (eq && j < dropped) nothead |= 1u << k;
                }
                more = more || step + (uint32_t)kL2Unroll < len;
            }
        }
        if (!__any(more ? 1 : 0)) break;
    }
    // Crowded bins (more than kLsMaxBin members: equal and nearly equal keys of a repeat family -- the genome-like 1 GiB text
    // has one in 42 % of its workgroups, 68 000 bins of 33 ... 128 members mostly; a member stepping through its bin alone
    // takes a step a member): a wave takes 64 members of the bin and, for every distinct value among them, counts over the whole
    // bin -- 64 members a ballot -- the members below it, the equal ones and the equal ones in front of its piece; equal
    // members keep the order of their slots.  (Every member against every member, passed from lane to lane: the same result,
    // but 64 / (distinct values) times the work, and workgroups of nothing but crowded bins took ten times the usual.)
    const uint32_t n_jobs = s_jobs < (uint32_t)kL2Jobs ? s_jobs : (uint32_t)kL2Jobs;
    if (n_jobs) { // (uniform)
        const uint32_t inbin = bshift >= 32 ? ~0u : ((1u << bshift) - 1u); // (a bin's members differ below bit bshift <= 19 only)
        constexpr uint32_t kNoKey = 0xFFFFFFFFu;                           // (beyond the bin's end: above every key, equal to none)
        for (uint32_t jb = (uint32_t)w; jb < n_jobs; jb += (uint32_t)kL2Waves) {
            const uint32_t job = (uint32_t)__builtin_amdgcn_readfirstlane((int)jobs[jb]); // (uniform: the loops below stay scalar)
            const uint32_t first = job & 0x1FFFu, len = (job >> 13) & 0x1FFFu, piece0 = (job >> 26) * kWave, mine = piece0 + (uint32_t)lane;
            const bool act = mine < len;
            const uint32_t ck = act ? R[first + mine] & inbin : 0u;
            // (the bin's first 128 members stay in registers: few bins hold more)
            const uint32_t c0 = (uint32_t)lane < len ? R[first + (uint32_t)lane] & inbin : kNoKey;
            const uint32_t c1 = kWave + (uint32_t)lane < len ? R[first + kWave + (uint32_t)lane] & inbin : kNoKey;
            uint64_t todo = __ballot(act ? 1 : 0);
            while (todo) {
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)ck, __ffsll((unsigned long long)todo) - 1);
                const uint64_t mym = __ballot(act && ck == v ? 1 : 0), e0 = __ballot(c0 == v ? 1 : 0), e1 = __ballot(c1 == v ? 1 : 0);
                uint32_t below = (uint32_t)__popcll(__ballot(c0 < v ? 1 : 0)) + (uint32_t)__popcll(__ballot(c1 < v ? 1 : 0));
                uint32_t equal = (uint32_t)__popcll(e0) + (uint32_t)__popcll(e1);
                uint32_t front = (piece0 >= kWave ? (uint32_t)__popcll(e0) : 0u) + (piece0 >= 2u * kWave ? (uint32_t)__popcll(e1) : 0u);
                for (uint32_t j0 = 2u * kWave; j0 < len; j0 += kWave) {
                    const uint32_t c = j0 + (uint32_t)lane < len ? R[first + j0 + (uint32_t)lane] & inbin : kNoKey;
                    const uint32_t e = (uint32_t)__popcll(__ballot(c == v ? 1 : 0));
                    below += (uint32_t)__popcll(__ballot(c < v ? 1 : 0));
                    equal += e;
                    front += j0 < piece0 ? e : 0u;
                }
                if ((mym >> lane) & 1ull) {
                    const uint32_t infront = front + (uint32_t)__popcll(mym & lanemask_lt());
                    team_fin[first + mine] = (uint16_t)((first + below + infront) | (equal > 1u ? 0x2000u : 0u) | (infront ? 0x4000u : 0u));
                }
                todo &= ~mym;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kL2Items; ++k)
            if ((team >> k) & 1u) {
                const uint32_t r = team_fin[sf[k] & 0xFFFFu];
                sf[k] |= (r & 0x1FFFu) << 16;
                tied |= ((r >> 13) & 1u) << k;
                nothead |= ((r >> 14) & 1u) << k;
            }
    }
    LS_PROBE(5);
    // ---- out: positions and windows to their places --------------------------------------------------------------------------
    const uint64_t gs = g0 + s;
    // (Measured and dropped: both arrays staged through LDS and stored as whole 16-byte pieces in order -- 2.04 against 1.84 ms;
    //  the positions' loads issued before the counting instead of before the placing -- 2.15: a spill, and the counting waits.)
#pragma unroll
    for (int k = 0; k < kL2Items; ++k)
        if ((own >> k) & 1u) {
            vout[gs + (sf[k] >> 16)] = val[k];
            if (seedw) seedw[gs + (sf[k] >> 16)] = pay[k];
        }
    // ---- the members of groups of equal keys, listed in order ------------------------------------------------------------------
    __syncthreads(); // (every wave has read the sort fields it needs: R is free)
    LS_PROBE(6);
#pragma unroll
    for (int k = 0; k < kL2Items; ++k)
        if ((tied >> k) & 1u) {
            const uint32_t fin = sf[k] >> 16;
            R[fin] = val[k];
            atomicOr(&tiedb[fin >> 5], 1u << (fin & 31u));
            if (!((nothead >> k) & 1u)) atomicOr(&headb[fin >> 5], 1u << (fin & 31u));
        }
    __syncthreads();
    if (w == 0) { // tied members before each word: one wave scans the kL2Words word counts
        uint32_t run = 0;
        for (int c0 = 0; c0 < kL2Words; c0 += kWave) {
            const int i = c0 + lane;
            const uint32_t cnt = i < kL2Words ? (uint32_t)__popc(tiedb[i]) : 0u;
            const uint32_t inc = wave_inclusive_scan<OpAdd>(cnt);
            if (i < kL2Words) tpre[i] = run + inc - cnt;
            run += __shfl(inc, kWave - 1, kWave);
        }
        if (lane == 0) tile_start[blockIdx.x] = (uint32_t)gs, tile_cnt[blockIdx.x] = run;
    }
    __syncthreads();
    for (int wd = t; wd < kL2Words; wd += kL2Threads) {
        uint32_t bits = tiedb[wd], at = tpre[wd];
        const uint32_t heads = headb[wd];
        while (bits) {
            const uint32_t b = (uint32_t)__ffs(bits) - 1u, q = (uint32_t)wd * 32u + b;
            bits &= bits - 1u;
            uint2 ent;
            ent.x = (uint32_t)(gs + q), ent.y = R[q];
            stage[gs + at] = ent;
            stage_head[gs + at] = (uint8_t)((heads >> b) & 1u);
            ++at;
        }
    }
    LS_PROBE(7);
#ifdef SX_LS_PROBE
    if (t == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&sx_ls_probe[i], (unsigned long long)probe_d[i]);
    if (t == 0) { // the slowest workgroup, its insert phase, and how many took more than 3 times / 10 times 60 000 cycles
        unsigned long long all = 0;
        for (int i = 0; i < 8; ++i) all += probe_d[i];
        atomicMax(&sx_ls_probe[16], all);
        atomicMax(&sx_ls_probe[17], (unsigned long long)probe_d[5]);
        if (all > 180000ull) atomicAdd(&sx_ls_probe[18], 1ull);
        if (all > 600000ull) atomicAdd(&sx_ls_probe[19], 1ull);
    }
#endif
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

uint32_t sx_local_sort_tiles(uint64_t m) { return sx_div_up(m, kL2Cap - 1024); } // (the shortest span a launch may take)

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
                  uint32_t longest_expected, uint32_t *long_list, uint32_t long_cap, uint32_t res[3], bool crowded_expected)
{
    // The span: a workgroup's costs that do not depend on its pairs (zeroing and scanning 8192 counters, the barriers) are
    // spread over more pairs the longer it is (1 GiB of DNA, dense keys: span 5120 2.79 ms, 5632 2.60, 5888 2.52), but a
    // sub-bucket that starts at the span's last pair must end within the reach of kLsCap pairs, or the whole sort falls back
    // to plain passes.  So the span grows only where the caller knows how long sub-buckets get (four-letter texts with
    // dense keys: uniform symbols at a known rate; longest_expected = 0: not known), with a factor of safety.
    // (a span is the lean kernel's reach less a margin of 1024, 768 or 512 pairs; the other kernel, whose reach is at least
    //  as long, takes the same tiling)
    uint32_t span = (uint32_t)kL2Cap - 1024u;
    if (longest_expected > 0 && 3 * (uint64_t)longest_expected <= 2 * 512u) span = (uint32_t)kL2Cap - 512u;
    else if (longest_expected > 0 && 3 * (uint64_t)longest_expected <= 2 * 768u) span = (uint32_t)kL2Cap - 768u;
    const uint32_t tiles = sx_div_up(m, span);
    const uint32_t L = (uint32_t)(kbits - top_bits);
    uint32_t res_local[3];
    if (!res) res = res_local;
    SX_CHECK(hipMemsetAsync(d_total_and_fail + 1, 0, 2 * sizeof(uint32_t), ctx->stream));
#ifndef SX_LS2_SKEWED
#define SX_LS2_SKEWED 1 // texts with skewed symbol counts (repeat families crowd bins): 1 the lean kernel too, 0 the other one at once
#endif
    // (1 GiB of uniform DNA 1.75 - 1.8 ms against the other kernel's 2.59; the genome-like 1 GiB text, whose repeat family crowds a bin
    //  in 42 % of the workgroups: 3.07 against 4.1 -- the lean kernel 2.2, the 2 % of the workgroups it leaves 0.2, the tied lists)
    if (ctx->local_sort_lean_off || (crowded_expected && !SX_LS2_SKEWED)) { // SX_FLAG_LOCAL_SORT_LEAN_OFF (tests, A/B): rounds 3 and 4's kernel for every workgroup
        SX_CHECK(hipMemsetAsync(tile_start, 0xFF, (size_t)tiles * sizeof(uint32_t), ctx->stream));
        sx_launch(ctx, SX_KC_LOCAL_SORT, m * (12 + 4 + (seedw ? 4 : 0)), local_sort_redo_kernel, dim3(tiles), dim3(kLsThreads), kin, vin, m, L,
                  (uint32_t)kbits, vout, seedw, tile_start, tile_cnt, stage, stage_head, d_total_and_fail + 1, span, long_list,
                  d_total_and_fail + 2, long_list ? long_cap : 0u);
    } else {
        sx_launch(ctx, SX_KC_LOCAL_SORT, m * (12 + 4 + (seedw ? 4 : 0)), local_sort_kernel, dim3(tiles), dim3(kL2Threads), kin, vin, m, L,
                  (uint32_t)kbits, vout, seedw, tile_start, tile_cnt, stage, stage_head, d_total_and_fail + 1, span, long_list,
                  d_total_and_fail + 2, long_list ? long_cap : 0u);
    }
#ifdef SX_LS_PROBE
    {
        unsigned long long h[24];
        SX_CHECK(hipStreamSynchronize(ctx->stream));
        SX_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(sx_ls_probe), sizeof h));
        fprintf(stderr, "local_sort phases (cycles a workgroup, %u workgroups): load+zero %llu starts %llu count %llu scan %llu drop %llu insert %llu stores %llu list %llu\n",
                tiles, h[0] / tiles, h[1] / tiles, h[2] / tiles, h[3] / tiles, h[4] / tiles, h[5] / tiles, h[6] / tiles, h[7] / tiles);
        fprintf(stderr, "local_sort crowded bins (bins / members): <=32 %llu / %llu, <=64 %llu / %llu, <=128 %llu / %llu, <=256 %llu / %llu, more %llu / %llu\n",
                h[11] >> 40, h[11] & 0xFFFFFFFFFFull, h[12] >> 40, h[12] & 0xFFFFFFFFFFull, h[13] >> 40, h[13] & 0xFFFFFFFFFFull, h[14] >> 40,
                h[14] & 0xFFFFFFFFFFull, h[15] >> 40, h[15] & 0xFFFFFFFFFFull);
        fprintf(stderr, "local_sort slowest workgroup %llu cycles (insert phase at most %llu); %llu took > 180 000, %llu > 600 000\n", h[16], h[17], h[18], h[19]);
        fprintf(stderr, "local_sort left to the other kernel: %llu workgroups with more sub-buckets than bins, %llu with a crowded bin (fullest %llu)\n", h[8], h[9], h[10]);
        memset(h, 0, sizeof h);
        SX_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(sx_ls_probe), h, sizeof h));
    }
#endif
    for (int again = 0; again < 2; ++again) {
        SX_TRY((device_scan<OpAdd>(ctx, tiles, InU32{tile_cnt}, OutExclusive{tile_off}, d_total_and_fail, SX_KC_NAMES, 0)));
        sx_launch(ctx, SX_KC_NAMES, 0, local_tied_gather_kernel, dim3(tiles), dim3(kBlock), (const uint32_t *)tile_start,
                  (const uint32_t *)tile_cnt, (const uint32_t *)tile_off, (const uint2 *)stage, (const uint8_t *)stage_head, apos, ap,
                  ahead, cap);
        SX_TRY(sx_readback(ctx, d_total_and_fail, 3, res));
        if (again || !(res[1] & 4u) || (res[1] & 1u)) break;
        // some workgroups left their pairs to the kernel with the stable passes (crowded bins: a repeat's ties): that launch,
        // then the list of tied members once more, now with theirs
        sx_launch(ctx, SX_KC_LOCAL_SORT, 0, local_sort_redo_kernel, dim3(tiles), dim3(kLsThreads), kin, vin, m, L, (uint32_t)kbits, vout,
                  seedw, tile_start, tile_cnt, stage, stage_head, d_total_and_fail + 1, span, long_list, d_total_and_fail + 2, 0u);
    }
    return 0;
}
