// sx_induce_chain.hpp -- small rounds of any alphabet: the chained single launch (decoupled look-back), the general one-workgroup tail kernel, device-wide jumps over very long runs, the passes' bookkeeping kernels
// (included by sx_induce.hip, which holds the passes' host side; one translation unit)
#pragma once
#include "sx_induce_common.hpp"

namespace sx {

// ---- one round = one launch -----------------------------------------------------------
// Stable multi-way split of the entries in range_in (read from device memory, so rounds
// can be queued without the host knowing their sizes): entry p with window w induces
// p-1 into bucket text[p-1] (= the window's first symbol) when the type test accepts
// it.  Tiles take tickets; per destination bucket the tile-local counts are chained
// across tiles by decoupled look-back (sx_device.hpp), so the entries are read once.
// The last tile publishes the advanced bucket cursors and the range appended to
// bucket c, which is the next round's input.
template <class WT, int BITS>
__global__ __launch_bounds__(kBlock) void induce_round_kernel(
    const uint32_t *__restrict__ srcP, const WT *__restrict__ srcW, const uint32_t *__restrict__ range_in,
    uint32_t *__restrict__ range_out, int rev, int mode, uint32_t c, wnd_cfg cfg, const uint8_t *__restrict__ T,
    const uint32_t *__restrict__ cursor_cur, uint32_t *__restrict__ cursor_nxt, int dir, uint32_t *__restrict__ SA,
    WT *__restrict__ WN, uint8_t *__restrict__ BW, uint32_t nkeys, uint64_t *__restrict__ status, uint32_t epoch,
    uint32_t *__restrict__ ticket,
    uint32_t chain_max /* rounds longer than this are left to the three-launch form; ~0u: take any round */,
    int tail_follows /* the batch ends with the tail kernel: rounds of up to kTailEntries entries are left to it */,
    int pass_large /* a round longer than chain_max is nobody's here: hand it on as it is (the host queues it again) */)
{
    __shared__ uint32_t wcount[kWavesPerBlock][256];
    __shared__ uint32_t gpos[256];  // entries of earlier tiles per bucket
    __shared__ uint32_t gbase[256]; // bucket cursors at the start of the round
    __shared__ uint32_t tcount[256]; // this tile's entries per bucket
    __shared__ uint32_t s_tile;
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t lo = range_in[0], hi = range_in[1];
    const uint32_t len = hi - lo;
    if (len > chain_max && !pass_large) return; // a large round: the three-launch form handles it
    if (len == 0 || (range_out && tail_follows && len <= (BITS > 3 ? kTailMulti * kTailEntries : kTailEntries)) || len > chain_max) {
        // nothing to do, or a round small enough for the tail kernel that ends the batch: carry the cursors over,
        // hand the range on as it is
        if (blockIdx.x == 0) {
            cursor_nxt[t] = cursor_cur[t];
            if (t == 0 && range_out) {
                range_out[0] = len ? lo : hi;
                range_out[1] = hi;
            }
        }
        return;
    }
    const uint32_t ntiles = (len + kIndTile - 1) / kIndTile;
    gbase[t] = cursor_cur[t];
    for (;;) {
        if (t == 0) s_tile = atomicAdd(ticket, 1u);
        for (int i = t; i < kWavesPerBlock * 256; i += kBlock) (&wcount[0][0])[i] = 0;
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= ntiles) break;
        const uint32_t wave0 = tile * (uint32_t)kIndTile + (uint32_t)w * (kWave * kIndItems);
        uint32_t val[kIndItems], dig[kIndItems], rnk[kIndItems];
        WT wnd[kIndItems];
        bool ok[kIndItems];
        // (all of the tile's loads are issued before the first is looked at: see wide_scatter_tile)
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            const uint32_t i = wave0 + (uint32_t)k * kWave + lane;
            const uint32_t idx = lo + (i < len ? (rev ? len - 1u - i : i) : 0u);
            val[k] = srcP[idx];
            wnd[k] = srcW[idx];
        }
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            const uint32_t i = wave0 + (uint32_t)k * kWave + lane;
            const uint32_t p = i < len ? val[k] : 0u;
            const WT ww = wnd[k];
            ok[k] = false;
            dig[k] = 0;
            val[k] = 0;
            wnd[k] = 0;
            if (p != 0) {
                const uint32_t ch = wnd_first<WT>(ww, cfg);
                ok[k] = induce_accept(ch, c, mode);
                dig[k] = ch;
                val[k] = p - 1u;
                wnd[k] = wnd_pop<WT>(ww, cfg);
            }
        }
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) rnk[k] = wave_rank_step<BITS>(dig[k], ok[k], wcount[w]);
        __syncthreads();
        {
            const uint32_t d = (uint32_t)t;
            uint32_t cnt = 0;
#pragma unroll
            for (int ww = 0; ww < kWavesPerBlock; ++ww) {
                const uint32_t x = wcount[ww][d];
                wcount[ww][d] = cnt;
                cnt += x;
            }
            uint32_t excl = 0;
            if (BITS > 3) { // one thread per bucket walks back on its own
                if (d < nkeys) excl = chain_exclusive_prefix(status, nkeys, tile, d, cnt, epoch);
                gpos[d] = excl;
                tcount[d] = cnt;
            } else {
                tcount[d] = cnt;
            }
        }
        if (BITS <= 3) { // <= 8 buckets: a whole wave walks back for each of them, 64 tiles a step
            __syncthreads();
            for (uint32_t d = (uint32_t)w; d < nkeys; d += kWavesPerBlock) {
                const uint32_t excl = chain_exclusive_prefix_wave(status, nkeys, tile, d, tcount[d], epoch);
                if (lane == 0) gpos[d] = excl;
            }
        }
        __syncthreads();
        if (tile == ntiles - 1) {
            const uint32_t d = (uint32_t)t;
            const uint32_t total = (d < nkeys ? gpos[d] : 0u) + tcount[d], cur = gbase[d];
            cursor_nxt[d] = dir > 0 ? cur + total : cur - total;
            if (d == c && range_out) {
                range_out[0] = dir > 0 ? cur : cur - total;
                range_out[1] = dir > 0 ? cur + total : cur;
            }
        }
        __syncthreads();
        {
            bool need[kIndItems]; // windows that ran dry: back to the text, all of a thread's reads in flight together
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) need[k] = ok[k] && val[k] != 0 && wnd_count<WT>(wnd[k]) == 0;
            refill_windows<WT, kIndItems>(T, val, need, cfg, wnd);
        }
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            if (ok[k]) {
                const uint32_t d = dig[k];
                const uint32_t r = gpos[d] + wcount[w][d] + rnk[k];
                const uint32_t dst = dir > 0 ? gbase[d] + r : gbase[d] - 1u - r;
                const WT nw = wnd[k];
                SA[dst] = val[k];
                WN[dst] = nw;
                BW[dst] = wnd_symbol<WT>(nw, cfg);
            }
        }
        __syncthreads(); // LDS is reused by the next tile
    }
}

// ---- the tail of a bucket's rounds: one workgroup, many rounds, one launch -------------
// Once a round fits one tile, its successors are smaller still (each keeps only the entries
// whose run of symbol c goes on), and a launch per round is all latency.  This kernel runs
// successive rounds of bucket c in a single workgroup -- read <= one tile, rank, scatter,
// advance the cursors held in LDS -- until the range is empty, grows beyond a tile (it
// cannot, but then the host's ordinary rounds take over) or max_iters rounds have run.
// Entries written in one iteration are read in the next by other waves of the same
// workgroup: the barrier's workgroup-scope fence orders them (the waves share the CU's L1).
// what the tail kernel of bucket c needs to open the next bucket's rounds (c < 0: nothing to do)
struct tail_next {
    int c;                   // the next bucket of the pass with rounds of its own
    uint32_t nk;
    const uint32_t *begin;   // bucket boundaries, 257
    const uint32_t *E;       // the pass's group ends [c][d]
    const uint32_t *tot;     // the pass's up-front entries from bucket c to bucket d, [c][d]
    uint32_t *range;         // the first range of the next bucket's rounds
    uint32_t *tickets;
    uint32_t ntickets;
    uint32_t *err;
};
template <class WT, int BITS>
__global__ __launch_bounds__(kTailBlock) void induce_tail_kernel(uint32_t *SA, WT *WN, uint8_t *BW, const uint32_t *__restrict__ range_in,
                                                             uint32_t *__restrict__ range_out, int rev, int mode,
                                                             uint32_t c, wnd_cfg cfg, const uint8_t *__restrict__ T,
                                                             const uint32_t *__restrict__ cursor_cur,
                                                             uint32_t *__restrict__ cursor_nxt, int dir,
                                                             uint32_t max_iters, uint32_t *poison, uint32_t *host_poison,
                                                             tail_next nb)
{
    constexpr int kDigits = BITS == 3 ? 8 : 256; // buckets that can receive anything
    __shared__ uint32_t wcount[kTailWaves][kDigits];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t s_range[2];
    __shared__ uint32_t s_flag;
    // More than 8 buckets: a round of thousands of entries leaves the workgroup in bucket order -- staged in LDS, every
    // bucket's entries next to each other, then stored as runs.  Stored from the registers that hold them, entry by entry
    // into 256 buckets, a wave's store touches 64 lines: the first round of a byte text's bucket (4 - 8 thousand entries,
    // three stores each) spent 12 of the kernel's 25 us handing its lines to the memory system one at a time (clock64
    // around the phases; the window refills, suspected first, were not it).
    constexpr int kStageCap = BITS > 3 ? kTailTile : 1;
#ifndef SX_TAIL_STAGE_FROM
#define SX_TAIL_STAGE_FROM 512u // (the CPU test harness: 8, so that short texts' rounds take this form too)
#endif
    constexpr uint32_t kStageFrom = SX_TAIL_STAGE_FROM; // entries of a round (a tile of it) from which the stores are staged
    __shared__ WT st_w[kStageCap];
    __shared__ uint32_t st_v[kStageCap];
    __shared__ uint8_t st_d[kStageCap];
    __shared__ uint32_t st_off[BITS > 3 ? 256 : 1], st_ex[BITS > 3 ? 256 : 1], st_scan[kTailWaves];
    const int t = (int)threadIdx.x, lane = lane_id(), w = wave_id();
    if (t < 256) gbase[t] = cursor_cur[t];
    if (t == 0) {
        s_range[0] = range_in[0];
        s_range[1] = range_in[1];
    }
    for (int i = t; i < kTailWaves * kDigits; i += kTailBlock) (&wcount[0][0])[i] = 0;
    __syncthreads();
    // The entries a round appends to bucket c are the next round's input, in the order they were appended: the
    // threads that wrote them keep them (position, window) in registers, in place, so from the second round of a
    // launch on nothing is read back from memory (a round then costs its barriers, not two trips to L2).
    uint32_t val[kIndItems];
    WT wnd[kIndItems];
    bool live[kIndItems]; // entry k of this thread belongs to the current range (scan order: wave, k, lane)
    bool held = false;
    uint32_t prev_len = 0; // the range of the round before (for the jump)
    const uint32_t wave0 = (uint32_t)w * (kWave * kIndItems);
    for (uint32_t it = 0; it < max_iters; ++it) {
        const uint32_t lo = s_range[0], len = s_range[1] - lo;
        // (more than 8 buckets: a round of up to kTailMulti tiles is taken tile after tile -- the second round of a byte
        //  text's buckets, 8 - 16 thousand entries, was a chained launch of its own in front of this kernel: 10 us of
        //  the bucket's 70)
        if (len == 0 || len > (BITS == 3 ? kTailEntries : kTailMulti * kTailEntries)) break; // uniform
        const bool multi = len > kTailEntries; // uniform
        // ---- run jump -------------------------------------------------------------------
        // Inside a long run of symbol c every entry of the range induces its left neighbour
        // into bucket c again, round after round, in the same order.  If the L symbols to the
        // left of every entry are all c, the next L rounds are known: round j holds the same
        // entries minus j, in the next `len` slots.  They are written at once (L = 16 symbols
        // per checking thread; 4096 rounds a step for a single run) instead of one at a time.
        // Tried only when the last round kept every entry (the sign of runs): the check reads memory.
        if ((mode == MODE_L_FROM_L || mode == MODE_S_FROM_S) && len == prev_len && !multi) {
            const uint32_t G = len <= (uint32_t)kTailBlock ? (uint32_t)kTailBlock / len : 1u; // threads per entry
            const uint64_t cpat = 0x0101010101010101ull * (uint64_t)c;
            // thread (i, q) looks at the q-th 16 symbols to the left of entry i; the nearest piece of any entry that is
            // not all c bounds the rounds that can be written at once (a run of 3000 symbols: 2992 rounds in one step,
            // where all G pieces had to be c before -- 16 384 symbols for a single run -- and shorter runs went round by round)
            uint32_t first_other = G;
            for (uint32_t e = (uint32_t)t; e < len * G; e += kTailBlock) {
                const uint32_t i = e / G, q = e % G;
                const uint32_t p = SA[lo + (rev ? len - 1u - i : i)];
                bool all_c = false;
                if (p >= 16u * (q + 1u)) {
                    uint64_t o0, o1;
                    load_bytes16(T, (uint64_t)(p - 16u * (q + 1u)), o0, o1);
                    all_c = o0 == cpat && o1 == cpat;
                }
                if (!all_c && q < first_other) first_other = q;
            }
            if (t == 0) s_flag = G;
            __syncthreads();
            if (first_other < G) atomicMin(&s_flag, first_other);
            __syncthreads();
            const uint32_t L = 16u * s_flag;
            __syncthreads(); // (s_flag is set again by the next step)
            if (L) { // uniform
                const uint32_t cur = gbase[c], total = L * len;
                for (uint32_t o = (uint32_t)t; o < total; o += kTailBlock) {
                    const uint32_t j = o / len + 1u, i = o % len;
                    const uint32_t v = SA[lo + (rev ? len - 1u - i : i)] - j;
                    const uint32_t dst = dir > 0 ? cur + o : cur - 1u - o;
                    const WT nw = v ? wnd_fill<WT>(T, v, cfg) : (WT)0;
                    SA[dst] = v;
                    WN[dst] = nw;
                    BW[dst] = wnd_symbol<WT>(nw, cfg);
                }
                __syncthreads();
                if ((uint32_t)t == c) {
                    gbase[c] = dir > 0 ? cur + total : cur - total;
                    s_range[0] = dir > 0 ? cur + total - len : cur - total;
                    s_range[1] = dir > 0 ? cur + total : cur - total + len;
                }
                held = false; // the range is now what the jump wrote last
                __syncthreads();
                continue;
            }
        }
        prev_len = len;
        const uint32_t c_first = gbase[c]; // (where the round's appends to bucket c begin)
        for (uint32_t sub0 = 0; sub0 < len; sub0 += kTailEntries) { // uniform; one trip unless `multi`
        if (!held || multi) { // the range's entries from memory (the first round of a launch, after a jump, a round of several tiles)
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) { // (all loads issued before any is looked at)
                const uint32_t i = sub0 + wave0 + (uint32_t)k * kWave + lane;
                const uint32_t idx = lo + (i < len ? (rev ? len - 1u - i : i) : 0u);
                val[k] = SA[idx];
                wnd[k] = WN[idx];
            }
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) {
                const uint32_t i = sub0 + wave0 + (uint32_t)k * kWave + lane;
                live[k] = i < len;
                if (!live[k]) val[k] = 0, wnd[k] = 0;
            }
        }
        uint32_t dig[kIndItems], rnk[kIndItems];
        bool ok[kIndItems];
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            ok[k] = false;
            dig[k] = 0;
            if (live[k] && val[k] != 0) {
                const uint32_t ch = wnd_first<WT>(wnd[k], cfg);
                ok[k] = induce_accept(ch, c, mode);
                dig[k] = ch;
                val[k] -= 1u;
                wnd[k] = wnd_pop<WT>(wnd[k], cfg);
            }
        }
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) rnk[k] = wave_rank_step<BITS>(dig[k] & (uint32_t)(kDigits - 1), ok[k], wcount[w]);
        __syncthreads();
        uint32_t cnt = 0; // entries of this round for bucket t
        if (t < kDigits) {
#pragma unroll
            for (int ww = 0; ww < kTailWaves; ++ww) {
                const uint32_t x = wcount[ww][t];
                wcount[ww][t] = cnt;
                cnt += x;
            }
        }
        __syncthreads();
        {
            bool need[kIndItems], any = false;
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) any |= need[k] = ok[k] && val[k] != 0 && wnd_count<WT>(wnd[k]) == 0;
            // (a wave without a dry window skips the block: the scatter that fed this round filled them up, kTailAhead)
            if (__any(any ? 1 : 0)) refill_windows<WT, kIndItems>(T, val, need, cfg, wnd);
        }
        const bool staged = BITS > 3 && (len - sub0 < kTailEntries ? len - sub0 : kTailEntries) >= kStageFrom; // uniform
        if (staged) {
            // first staged slot of every bucket: the buckets' counts of this round, summed up in bucket order
            const uint32_t inc = wave_inclusive_scan<OpAdd>(cnt);
            if (lane == kWave - 1) st_scan[w] = inc;
            __syncthreads();
            uint32_t ex = inc - cnt, produced = 0;
            for (int ww = 0; ww < kTailWaves; ++ww) {
                const uint32_t x = st_scan[ww];
                if (ww < w) ex += x;
                produced += x;
            }
            if (t < kDigits) {
                st_ex[t & 255] = ex;
                st_off[t & 255] = dir > 0 ? gbase[t] - ex : gbase[t] - 1u + ex; // staged slot i of bucket t lands at st_off + i (L pass) / - i (S pass)
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) {
                live[k] = ok[k] && dig[k] == c; // appended to bucket c itself: part of the next round
                if (ok[k]) {
                    const uint32_t d = dig[k] & 255u;
                    const uint32_t slot = (st_ex[d] + wcount[w][d & (uint32_t)(kDigits - 1)] + rnk[k]) & (uint32_t)(kStageCap - 1);
                    st_v[slot] = val[k];
                    st_w[slot] = wnd[k];
                    st_d[slot] = (uint8_t)d;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kIndItems; ++k) {
                const uint32_t i = (uint32_t)t + (uint32_t)k * kTailBlock;
                if (i < produced) {
                    const uint32_t g = st_off[st_d[i & (uint32_t)(kStageCap - 1)]];
                    const uint32_t dst = dir > 0 ? g + i : g - i;
                    const WT nw = st_w[i & (uint32_t)(kStageCap - 1)];
                    SA[dst] = st_v[i & (uint32_t)(kStageCap - 1)];
                    WN[dst] = nw;
                    BW[dst] = wnd_symbol<WT>(nw, cfg);
                }
            }
        } else {
#pragma unroll
        for (int k = 0; k < kIndItems; ++k) {
            live[k] = ok[k] && dig[k] == c; // appended to bucket c itself: part of the next round
            if (ok[k]) {
                const uint32_t d = dig[k];
                const uint32_t r = wcount[w][d & (uint32_t)(kDigits - 1)] + rnk[k];
                const uint32_t dst = dir > 0 ? gbase[d] + r : gbase[d] - 1u - r;
                SA[dst] = val[k];
                WN[dst] = wnd[k];
                BW[dst] = wnd_symbol<WT>(wnd[k], cfg);
            }
        }
        }
        held = !multi; // (the entries a round of several tiles appended lie with many threads' registers' worth each: from memory)
        __syncthreads();
        if (t < kDigits) {
            const uint32_t before = gbase[t];
            gbase[t] = dir > 0 ? before + cnt : before - cnt;
#pragma unroll
            for (int ww = 0; ww < kTailWaves; ++ww) wcount[ww][t] = 0;
        }
        __syncthreads();
        } // (tiles of the round)
        if ((uint32_t)t == c) { // what the round appended to bucket c is the next round's input
            const uint32_t now = gbase[c];
            s_range[0] = dir > 0 ? c_first : now;
            s_range[1] = dir > 0 ? now : c_first;
        }
        __syncthreads();
    }
    // The head of the next bucket's rounds in a pass whose other-region rounds were done up front (what bucket_begin_kernel
    // does in a launch of its own -- 4.4 us and a launch boundary a bucket, 510 times a byte text's build --; sx_induce_wide.hpp
    // has the bookkeeping): the cursors jump over the up-front entries of the buckets c .. next - 1 to the group starts, checked
    // against the bigram counts' table; the first range is what lies in front of the next bucket's own group.  Not when this
    // bucket is left unfinished (or an earlier one was): later launches then find an empty range.
    uint32_t mine = t < 256 ? gbase[t] : 0u;
    if (nb.c >= 0) { // uniform
        if (t == 0) {
            s_flag = (s_range[0] != s_range[1] || (poison && poison[0])) ? 1u : 0u;
            for (uint32_t i = 0; i < nb.ntickets; ++i) nb.tickets[i] = 0;
        }
        __syncthreads();
        const uint32_t d = (uint32_t)t, cn = (uint32_t)nb.c;
        if (s_flag) {
            if (t == 0) nb.range[0] = nb.range[1] = 0;
        } else if (d < nb.nk && (dir > 0 ? d >= cn : d <= cn)) {
            uint32_t want, have = mine;
            if (dir > 0) {
                want = cn == 0 ? nb.begin[d] : nb.E[(uint64_t)(cn - 1) * 256 + d];
                for (uint32_t k = c; k < cn; ++k) have += nb.tot[(uint64_t)k * 256 + d];
            } else {
                want = cn + 1 >= nb.nk ? nb.begin[d + 1] : nb.E[(uint64_t)(cn + 1) * 256 + d];
                for (uint32_t k = c; k > cn; --k) have -= nb.tot[(uint64_t)k * 256 + d];
            }
            if (have != want) atomicOr(nb.err, dir > 0 ? 1u : 2u);
            mine = want;
            if (d == cn) {
                nb.range[0] = dir > 0 ? nb.begin[cn] : want;
                nb.range[1] = dir > 0 ? want : nb.begin[cn + 1];
            }
        }
    }
    if (t < 256) cursor_nxt[t] = mine;
    if (t == 0) {
        range_out[0] = s_range[0];
        range_out[1] = s_range[1];
        tail_report(s_range[0], s_range[1], c, poison, host_poison);
    }
}

// ---- very long runs -------------------------------------------------------------------------------------
// The tail kernel's run jump writes 4096 rounds a step with one workgroup: 11 us a step, 45 ms for the 16 Mi
// symbols of a gap in a reference assembly (runs of N of up to 30 Mbp, one or more per chromosome, all in one
// bucket).  When a bucket's range is down to a handful of entries and the tail kernel has not finished them, the
// whole device takes over: `run_probe` finds how many symbols c lie immediately to the left of every entry (L, the
// minimum, looking kRunProbe symbols far), `run_fill` writes the L rounds (entries - 1, entries - 2, ...) into the
// next L x len slots of bucket c (each with its window and symbol byte), `run_commit` advances the cursor and
// leaves the last round as the range.  L = 0 changes nothing.
constexpr uint32_t kRunProbe = 1u << 26;
constexpr uint32_t kRunEntries = 64; // runs of c that are alive in the bucket at the same time (a gap per chromosome)
constexpr uint32_t kRunProbeChunk = (uint32_t)kBlock * 16u; // symbols a workgroup looks at per step
constexpr uint32_t kRunProbeGrid = 256;                    // workgroups per entry
// Workgroup x of entry y looks at the distances [k * chunk, (k + 1) * chunk), k = x, x + grid, ..., and stops as soon
// as the run is known to end nearer than where it would look next: the probe costs what the run is long, not the
// 64 Mi symbols it may look at most (with every workgroup reading its piece whatever the others found, a probe of 64
// entries read 4 GB: 5.5 ms, 14 probes in a genome-like 1 GiB text whose runs are a few dozen symbols long).
__global__ __launch_bounds__(kBlock) void run_probe_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ SA,
                                                          const uint32_t *__restrict__ range, uint32_t c, uint32_t from,
                                                          uint32_t look, uint32_t *__restrict__ run_len /* preset to ~0 */)
{
    const uint32_t lo = range[0], len = range[1] - lo;
    if (len == 0 || len > kRunEntries || blockIdx.y >= len) { // (uniform) not a handful of entries: nothing to jump over
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (len == 0 || len > kRunEntries)) atomicMin(run_len, 0u);
        return;
    }
    const uint32_t p = SA[lo + blockIdx.y]; // (the minimum over the entries does not depend on their order)
    for (uint32_t chunk0 = from + blockIdx.x * kRunProbeChunk; chunk0 < look; chunk0 += gridDim.x * kRunProbeChunk) { // uniform
        // (a relaxed agent-scope load: what another workgroup found becomes visible in time, never too early)
        if (__hip_atomic_load(run_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= chunk0) return;
        const uint32_t d0 = chunk0 + threadIdx.x * 16u; // this thread looks at p-d0-1 ... p-d0-16
        if (d0 >= p) {
            if (d0 < p + 16u) atomicMin(run_len, p); // the text starts here: at most p symbols to the left
            continue;
        }
        const uint32_t cnt = p - d0 < 16u ? p - d0 : 16u;
        uint32_t first_other = cnt; // symbols c in a row, going left from p - d0
        for (uint32_t e = 0; e < cnt; ++e)
            if (T[p - d0 - 1u - e] != (uint8_t)c) {
                first_other = e;
                break;
            }
        if (first_other < 16u) atomicMin(run_len, d0 + first_other); // (cnt < 16: the text starts there)
    }
}

// rounds the jump covers: every entry of the range has at least that many symbols c to its left
__device__ __forceinline__ uint32_t run_length(const uint32_t *run_len, uint32_t len)
{
    if (len == 0 || len > kRunEntries) return 0;
    const uint32_t L = *run_len, most = kRunProbe / len; // (at most kRunProbe entries a jump: fits 32-bit offsets)
    return L > most ? most : L;
}

template <class WT>
__global__ __launch_bounds__(kBlock) void run_fill_kernel(const uint8_t *__restrict__ T, uint32_t *SA, WT *__restrict__ WN,
                                                         uint8_t *__restrict__ BW, const uint32_t *__restrict__ range,
                                                         const uint32_t *__restrict__ cursor, uint32_t c, int rev, int dir,
                                                         wnd_cfg cfg, const uint32_t *__restrict__ run_len)
{
    const uint32_t lo = range[0], len = range[1] - lo;
    const uint64_t total = (uint64_t)run_length(run_len, len) * len;
    const uint32_t cur = cursor[c];
    // round j holds the range's entries minus j, in the same order, in the next len slots (as the tail kernel's jump)
    for (uint64_t o = (uint64_t)blockIdx.x * kBlock + threadIdx.x; o < total; o += (uint64_t)gridDim.x * kBlock) {
        const uint32_t j = (uint32_t)(o / len) + 1u, i = (uint32_t)(o % len);
        const uint32_t v = SA[lo + (rev ? len - 1u - i : i)] - j;
        const uint32_t dst = dir > 0 ? cur + (uint32_t)o : cur - 1u - (uint32_t)o;
        const WT nw = v ? wnd_fill<WT>(T, v, cfg) : (WT)0;
        SA[dst] = v;
        WN[dst] = nw;
        BW[dst] = wnd_symbol<WT>(nw, cfg);
    }
}

__global__ void run_commit_kernel(uint32_t *__restrict__ range, uint32_t *__restrict__ cursor, uint32_t c, int dir,
                                  const uint32_t *__restrict__ run_len)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t lo = range[0], len = range[1] - lo;
    const uint32_t total = run_length(run_len, len) * len;
    if (total == 0) return;
    const uint32_t cur = cursor[c];
    cursor[c] = dir > 0 ? cur + total : cur - total;
    range[0] = dir > 0 ? cur + total - len : cur - total; // the last round written: the next round's input
    range[1] = range[0] + len;
}

// range <- [lo, hi) given by the host, or [a, cursor[c]) / [cursor[c], b) for the first round of a bucket
// (and the tickets of the chained launches that follow are zeroed: one launch instead of a memset and a launch)
// (poison: a bucket earlier in this unattended pass did not come to its end -- see induce_typed --: the range is left
//  empty, and every launch over an empty range only carries the cursors on)
__global__ void set_range_kernel(uint32_t *range, uint32_t lo, uint32_t hi, const uint32_t *cursor, int c, int which,
                                 uint32_t *tickets, uint32_t ntickets, const uint32_t *poison)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (uint32_t i = 0; i < ntickets; ++i) tickets[i] = 0;
        if (which == 1) hi = cursor[c];      // L pass: [bucket begin, head cursor)
        else if (which == 2) lo = cursor[c]; // S pass: [tail cursor, bucket end)
        if (poison && poison[0]) lo = hi = 0;
        range[0] = lo;
        range[1] = hi;
    }
}

// the tail kernel's last word in an unattended pass: a range it could not finish (runs longer than its steps, or more
// entries than it holds) is recorded once -- bucket and range, on the device and in the host's pinned page -- and
// stops the rest of the pass (set_range_kernel)
__device__ __forceinline__ void tail_report(uint32_t lo, uint32_t hi, uint32_t c, uint32_t *poison, uint32_t *host_poison)
{
    if (!poison || lo == hi || poison[0]) return;
    poison[1] = c, poison[2] = lo, poison[3] = hi;
    poison[0] = 1;
    __hip_atomic_store(host_poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bwt[i] = text[SA[i]-1]: the first symbol of slot i's window; the one slot whose entry is
// position 0 has an empty window (count 0) and gets the sentinel (bwt.c:13-20)
template <class WT>
__global__ __launch_bounds__(kBlock) void bwt_from_windows_kernel(const WT *__restrict__ WN, uint64_t N, wnd_cfg cfg,
                                                                  uint8_t *__restrict__ bwt)
{
    // 16 slots per thread: 16-byte loads of the windows, one 16-byte store of the symbols
    const uint64_t i0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 16u;
    if (i0 >= N) return;
    if (i0 + 16u <= N && (((uintptr_t)WN | (uintptr_t)bwt) & 15u) == 0) {
        uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            WT w[4];
            load_quad(WN + i0 + 4 * q, w);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t b = wnd_count<WT>(w[e]) == 0 ? 0u : wnd_first<WT>(w[e], cfg);
                out[q] |= (b & 0xFFu) << (8 * e);
            }
        }
        uint4 v;
        v.x = out[0], v.y = out[1], v.z = out[2], v.w = out[3];
        *reinterpret_cast<uint4 *>(bwt + i0) = v;
    } else {
        for (uint64_t i = i0; i < N && i < i0 + 16u; ++i) {
            const WT w = WN[i];
            bwt[i] = wnd_count<WT>(w) == 0 ? (uint8_t)0 : (uint8_t)wnd_first<WT>(w, cfg);
        }
    }
}

// the sort's 32-bit seed windows (fewer symbols, same layout) as the 64-bit words the passes of a wide alphabet read
__global__ __launch_bounds__(kBlock) void widen_windows_kernel(const uint32_t *__restrict__ in, uint64_t count, uint64_t *__restrict__ out)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 4u;
    if (i0 >= count) return;
    if (i0 + 4u <= count && (((uintptr_t)in | (uintptr_t)out) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + i0);
        uint4 a, b;
        a.x = v.x, a.y = 0, a.z = v.y, a.w = 0;
        b.x = v.z, b.y = 0, b.z = v.w, b.w = 0;
        *reinterpret_cast<uint4 *>(out + i0) = a;
        *reinterpret_cast<uint4 *>(out + i0 + 2) = b;
    } else {
        for (uint64_t i = i0; i < count && i < i0 + 4u; ++i) out[i] = in[i];
    }
}

template <class WT>
__global__ void set_entry_kernel(uint32_t *SA, WT *WN, uint8_t *BW, uint32_t p, const uint8_t *T, wnd_cfg cfg)
{
    const WT w = p ? wnd_fill<WT>(T, p, cfg) : (WT)0;
    SA[0] = p;
    WN[0] = w;
    BW[0] = wnd_symbol<WT>(w, cfg);
}

} // namespace sx
