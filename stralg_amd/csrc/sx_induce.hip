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
#include "sx_induce_common.hpp"
#include "sx_induce_small.hpp"
#include "sx_induce_wide.hpp"
#include "sx_induce_chain.hpp"

using namespace sx;

size_t sx_induce_scratch_bytes(uint64_t N, uint32_t sigma)
{
    // windows for every SA slot (8 bytes worst case) + seed windows (N/2) + symbol bytes + control block
    const uint64_t ntiles = (N + kIndTile - 1) / kIndTile + 1;
    const uint64_t wtiles = N / 4096 + 4; // wide alphabets: [tile][256] counts of (at least) 4096-entry tiles + chunk sums
    // (at most 8 buckets: (round, bucket) count rows of the eight-rounds-at-a-time form, over the largest bucket's tiles)
    return (size_t)N * 8 + 256 + (size_t)(N / 2 + 2) * 8 + 256 + (size_t)N + 256 + (size_t)sigma * ntiles * 4 + 256 +
           (size_t)kBatchRows * (ntiles + 1) * 4 + 1024 +
           (sigma > 8 ? (size_t)(wtiles + wtiles / 256 + 4) * 1024 + 512 : 0) + 16384 +
           // the up-front rounds of more than 8 buckets: tile counts of all buckets' regions, the bigram matrix and its tables
           (sigma > 8 ? ((size_t)N / kWideTile + 520) * 1024 + 5 * 65536 * 4 + 16384 : 0);
}

namespace {
#ifndef SX_INDUCE_GRID_CAP
#define SX_INDUCE_GRID_CAP 16384
#endif
// workgroups of a round's launches (they loop over the round's tiles).  1 GiB DNA, induce_scatter per step: 4096
// workgroups 5.55 ms, 16384: 4.99, 65536: 5.06, 262144: 5.12 (a device copy is fastest with many short workgroups too).
constexpr uint32_t kInduceGridCap = SX_INDUCE_GRID_CAP;
constexpr int kMaxSpec = 16; // rounds queued per batch (then the tail kernel) before the host looks at the range

template <class WT> struct induce_state {
    sx_ctx *ctx;
    const uint8_t *T;
    uint32_t *SA;
    WT *WN;
    uint64_t N, m; // entries of (SA, WN, BW) and of the seed arrays
    uint8_t *BW; // text[SA[i] - 1] of every written slot (0 for position 0): what the counting launches read; the BWT in the end
    uint32_t *cursor[2]; // ping-pong: a round reads one, its last tile writes the other
    uint32_t *ranges;    // (kMaxSpec + 2) x {lo, hi}
    uint32_t *tickets;   // kMaxSpec + 2
    uint32_t *run_len;   // symbols a device-wide run jump covers
    uint64_t *status;
    uint32_t chain_max; // rounds up to this many entries take the chained launch
    uint32_t *hist;   // [nk][stride] tile counts of the three-launch form (at most 8 buckets)
    uint32_t *bhist;  // [round * 8 + bucket][stride] tile counts of the eight-rounds-at-a-time form
    uint32_t *btotals; // kBatchRows row totals
    int batch_on;
    int unattended; // the buckets are queued one behind the other without a look at a bucket's last range (see induce_typed)
    uint32_t *poison;      // device: {set, bucket, lo, hi} of the first bucket an unattended pass could not finish
    uint32_t *host_poison; // the same flag in the host's pinned page (the host looks at it between buckets, without a wait)
    uint32_t *whist;  // [tile][256] the same for wide alphabets, tiles of 8192 entries
    uint32_t *wsums;  // [chunk][256] column sums of chunks of 256 tiles
    uint32_t stride;
    uint32_t nk;
    int small_alphabet;
    int par; // which cursor buffer is current
    wnd_cfg cfg;
    // more than 8 buckets: the other-region rounds of all buckets are done up front (hoist_*_kernel, bucket_begin_kernel)
    int hoist;
    uint32_t *d_begin;       // bucket boundaries on the device (257)
    uint32_t *hoist_E;       // the current pass's group ends: EL or ES, [c][d]
    uint32_t *hoist_tot;     // the current pass's up-front entries from bucket c to bucket d, [c][d]
    uint32_t *hoist_err;     // set by bucket_begin_kernel when a cursor is not where the bigram counts put it
    uint32_t hoist_from;     // the last bucket of the pass that had rounds of its own (L pass: 0, S pass: nk - 1 before the first)
    int next_c;              // unattended pass: the next bucket with rounds of its own, whose head the tail kernel of this one takes (-1: none)
    int begun_c;             // the bucket whose rounds the last tail kernel has opened (-1: none)
};

template <class WT>
void launch_round(induce_state<WT> &st, const uint32_t *srcP, const WT *srcW, int range_slot, int out_slot,
                  uint32_t tiles_bound, uint32_t tiles_likely, int rev, int mode, uint32_t c, int dir, int tail_follows,
                  int chained_only_up_to_chain_max = 0, int three_launch_only = 0)
{
    sx_ctx *ctx = st.ctx;
    uint32_t grid = tiles_bound < 1 ? 1 : tiles_bound;
    if (grid > kInduceGridCap) grid = kInduceGridCap; // (every kernel loops over its tiles: any grid size is correct)
    const uint32_t epoch = sx_chain_next_epoch(ctx);
    uint32_t *rin = st.ranges + 2 * range_slot;
    uint32_t *rout = out_slot >= 0 ? st.ranges + 2 * out_slot : nullptr;
    const uint32_t *cur = st.cursor[st.par];
    uint32_t *nxt = st.cursor[st.par ^ 1];
    // tiles_likely: what the round is expected to need (decides which forms are queued);
    // a round that turns out longer is still handled, by the chained form alone if need be.
    // three_launch_only: the round is large for sure (its size is known, or expected beyond doubt), no chained launch is
    // queued behind the three (they take a range of any length then, an empty one is carried on by the offsets launch)
    const bool only3 = three_launch_only != 0;
    const bool both = only3 || (!chained_only_up_to_chain_max && (uint64_t)tiles_likely * kIndTile > st.chain_max);
    const uint32_t chain_max = only3 ? 0u : ((both || chained_only_up_to_chain_max) ? st.chain_max : ~0u);
    const int pass_large = chained_only_up_to_chain_max;
    if (both && st.small_alphabet) {
        // the round may be a large one: queue the three-launch form as well
        // (entries of the suffix array have their symbol bytes next to them; the LMS seeds only their windows)
        const uint8_t *srcB = srcP == st.SA ? (const uint8_t *)st.BW : nullptr;
        const uint64_t src_len = srcP == st.SA ? st.N : st.m;
        if (srcB)
            sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, induce_count_bytes_kernel, dim3(grid), dim3(kBlock), srcB, (const uint32_t *)rin, rev,
                      mode, c, st.hist, st.stride, st.nk, chain_max, src_len);
        else
            sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, induce_count_kernel<WT, 3>, dim3(grid), dim3(kBlock), srcW, srcB,
                      (const uint32_t *)rin, rev, mode, c, st.cfg, st.hist, st.stride, st.nk, chain_max, src_len);
        sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)tiles_bound * st.nk * 8, induce_offsets_kernel, dim3(st.nk),
                  dim3(kRowThreads), st.hist, st.stride, (const uint32_t *)rin, rout, cur, nxt, dir, c, chain_max, only3 ? 1 : 0);
        {
#define SX_SCATTER_SMALL(M)                                                                                            \
    sx_launch(ctx, SX_KC_INDUCE_SCATTER, 0, induce_scatter_small_kernel<WT, M>, dim3(grid), dim3(kBlock), srcP, srcW, \
              (const uint32_t *)rin, c, st.cfg, st.T, (const uint32_t *)st.hist, st.stride, cur, st.SA, st.WN, st.BW, st.nk, \
              chain_max)
            switch (mode) { // mode fixes the scan direction (rev) and the side the buckets grow to (dir)
            case MODE_L_FROM_L: SX_SCATTER_SMALL(MODE_L_FROM_L); break;
            case MODE_L_FROM_LMS: SX_SCATTER_SMALL(MODE_L_FROM_LMS); break;
            case MODE_S_FROM_S: SX_SCATTER_SMALL(MODE_S_FROM_S); break;
            default: SX_SCATTER_SMALL(MODE_S_FROM_L); break;
            }
#undef SX_SCATTER_SMALL
        }
    }
    if (both && !st.small_alphabet) {
        // wide alphabets: the round as a radix pass over tiles of 8192 entries (count, offsets, scatter)
        const uint8_t *srcB = srcP == st.SA ? (const uint8_t *)st.BW : nullptr;
        const uint32_t wtiles = sx_div_up((uint64_t)(tiles_bound < 1 ? 1 : tiles_bound) * kIndTile, kWideTile);
        const uint32_t wgrid = wtiles > 2048 ? 2048 : wtiles;
        const int only = only3 ? 1 : 0;
        sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, induce_wide_count_kernel<WT>, dim3(wgrid), dim3(kWideThreads), srcW, srcB,
                  (const uint32_t *)rin, rev, mode, c, st.cfg, st.whist, chain_max);
        if (wtiles <= kWideOffMaxTiles) {
            sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)wtiles * 2048, induce_wide_offsets_kernel, dim3(256 / kWideOffCols), dim3(kWideOffThreads),
                      st.whist, (const uint32_t *)rin, rout, cur, nxt, dir, c, chain_max, only);
        } else {
            const uint32_t nchunks = sx_div_up(wtiles, kWideChunk);
            sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)wtiles * 1024, induce_wide_colsum_kernel, dim3(nchunks), dim3(kBlock),
                      (const uint32_t *)st.whist, (const uint32_t *)rin, st.wsums, chain_max);
            sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)nchunks * 2048, induce_wide_bases_kernel, dim3(1), dim3(kBlock), st.wsums,
                      (const uint32_t *)rin, rout, cur, nxt, dir, c, chain_max, only);
            sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)wtiles * 2048, induce_wide_apply_kernel, dim3(nchunks), dim3(kBlock), st.whist,
                      (const uint32_t *)rin, (const uint32_t *)st.wsums, chain_max);
        }
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, 0, induce_wide_scatter_kernel<WT, 8>, dim3(wgrid), dim3(kWideThreads),
                  srcP, srcW, (const uint32_t *)rin, rev, mode, c, st.cfg, st.T, (const uint32_t *)st.whist, cur, dir, st.SA, st.WN,
                  st.BW, chain_max);
    }
    uint32_t cgrid = grid > 1024 ? 1024 : grid;
    if (only3) {
    } else if (st.small_alphabet)
        sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, induce_round_kernel<WT, 3>, dim3(cgrid), dim3(kBlock), srcP, srcW,
                  (const uint32_t *)rin, rout, rev, mode, c, st.cfg, st.T, cur, nxt, dir, st.SA, st.WN, st.BW, st.nk, st.status,
                  epoch, st.tickets + range_slot, chain_max, tail_follows, pass_large);
    else if (st.nk <= 32)
        sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, induce_round_kernel<WT, 5>, dim3(cgrid), dim3(kBlock), srcP, srcW,
                  (const uint32_t *)rin, rout, rev, mode, c, st.cfg, st.T, cur, nxt, dir, st.SA, st.WN, st.BW, st.nk, st.status,
                  epoch, st.tickets + range_slot, chain_max, tail_follows, pass_large);
    else
        sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, induce_round_kernel<WT, 8>, dim3(cgrid), dim3(kBlock), srcP, srcW,
                  (const uint32_t *)rin, rout, rev, mode, c, st.cfg, st.T, cur, nxt, dir, st.SA, st.WN, st.BW, st.nk, st.status,
                  epoch, st.tickets + range_slot, chain_max, tail_follows, pass_large);
    st.par ^= 1;
    ctx->stats.induce_rounds++;
}

// steps of the tail kernel per launch: a run that outlasts them goes to the device-wide jump (run_fill).  In a pass
// queued as a whole nobody is there to start that jump, so the tail kernel gets more steps -- but not the 16 384 that
// would see any run the classification did not report (shorter than two tiles, 8191 symbols) to its end: runs of
// differing lengths just under that never meet the jump's condition (every entry of the round continued), and one
// workgroup then ground through thousands of dependent steps of a few microseconds each while the chip idled.  A
// bucket that outlasts these steps is reported (tail_report) and carried on attended, with the device-wide jump.
#ifndef SX_TAIL_ITERS_UNATTENDED
#define SX_TAIL_ITERS_UNATTENDED 1024u // (the CPU test harness: 96, so that short texts reach the report too)
#endif
constexpr uint32_t kTailIters = 64, kTailItersUnattended = SX_TAIL_ITERS_UNATTENDED;
template <class WT>
void launch_tail(induce_state<WT> &st, int range_slot, int out_slot, int rev, int mode, uint32_t c, int dir)
{
    sx_ctx *ctx = st.ctx;
    const uint32_t *cur = st.cursor[st.par];
    uint32_t *nxt = st.cursor[st.par ^ 1];
    if (st.small_alphabet)
        sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, induce_tail_small_kernel<WT>, dim3(1), dim3(kTailBlock), st.SA, st.WN, st.BW,
                  (const uint32_t *)(st.ranges + 2 * range_slot), st.ranges + 2 * out_slot, rev, mode, c, st.cfg, st.T,
                  cur, nxt, dir, st.unattended ? kTailItersUnattended : kTailIters, st.unattended ? st.poison : (uint32_t *)nullptr,
                  st.host_poison);
    else {
        tail_next nb = {-1, st.nk, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, nullptr};
        if (st.hoist && st.unattended && st.next_c >= 0)
            nb = {st.next_c, st.nk, (const uint32_t *)st.d_begin, (const uint32_t *)st.hoist_E, (const uint32_t *)st.hoist_tot, st.ranges,
                  st.tickets, (uint32_t)(kMaxSpec + 2), st.hoist_err};
        sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, induce_tail_kernel<WT, 8>, dim3(1), dim3(kTailBlock), st.SA, st.WN, st.BW,
                  (const uint32_t *)(st.ranges + 2 * range_slot), st.ranges + 2 * out_slot, rev, mode, c, st.cfg, st.T,
                  cur, nxt, dir, st.unattended ? kTailItersUnattended : kTailIters, st.unattended ? st.poison : (uint32_t *)nullptr,
                  st.host_poison, nb);
        st.begun_c = nb.c;
    }
    st.par ^= 1;
}

// kBatchRounds self rounds of bucket c in three launches (at most 8 buckets; ranges the tail kernel can take pass through)
template <class WT>
void launch_batch(induce_state<WT> &st, int range_slot, int out_slot, uint32_t tiles_bound, int mode, uint32_t c)
{
    sx_ctx *ctx = st.ctx;
    uint32_t grid = tiles_bound < 1 ? 1 : tiles_bound;
    if (grid > kInduceGridCap) grid = kInduceGridCap;
    const uint32_t *rin = st.ranges + 2 * range_slot;
    uint32_t *rout = st.ranges + 2 * out_slot;
    const uint32_t *cur = st.cursor[st.par];
    uint32_t *nxt = st.cursor[st.par ^ 1];
    const uint32_t min_len = ctx->induce_batch_min >= 0 ? (uint32_t)ctx->induce_batch_min : kTailEntries;
    if (mode == MODE_L_FROM_L) {
        sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, induce_batch_count_kernel<WT, MODE_L_FROM_L>, dim3(grid), dim3(kBlock),
                  (const uint32_t *)st.SA, (const WT *)st.WN, rin, c, st.cfg, st.T, st.bhist, st.stride, min_len);
    } else {
        sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, induce_batch_count_kernel<WT, MODE_S_FROM_S>, dim3(grid), dim3(kBlock),
                  (const uint32_t *)st.SA, (const WT *)st.WN, rin, c, st.cfg, st.T, st.bhist, st.stride, min_len);
    }
    sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)tiles_bound * st.nk * kBatchRounds * 8, induce_batch_offsets_kernel,
              dim3(kBatchRounds * st.nk), dim3(kRowThreads), st.bhist, st.stride, rin, st.btotals, st.nk, min_len);
    if (mode == MODE_L_FROM_L) {
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, 0, induce_batch_scatter_kernel<WT, MODE_L_FROM_L>, dim3(grid), dim3(kBlock),
                  (const uint32_t *)st.SA, (const WT *)st.WN, rin, rout, c, st.cfg, st.T, (const uint32_t *)st.bhist, st.stride,
                  (const uint32_t *)st.btotals, cur, nxt, st.SA, st.WN, st.BW, st.nk, min_len);
    } else {
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, 0, induce_batch_scatter_kernel<WT, MODE_S_FROM_S>, dim3(grid), dim3(kBlock),
                  (const uint32_t *)st.SA, (const WT *)st.WN, rin, rout, c, st.cfg, st.T, (const uint32_t *)st.bhist, st.stride,
                  (const uint32_t *)st.btotals, cur, nxt, st.SA, st.WN, st.BW, st.nk, min_len);
    }
    st.par ^= 1;
    ctx->stats.induce_rounds++;
}

// all rounds of one region of bucket c: the first range comes from the cursor, every
// round appends to bucket c what the next round reads; batches of queued rounds, one
// host look per batch
template <class WT>
int run_self_rounds(induce_state<WT> &st, uint32_t fixed_bound, uint32_t region_entries, int rev, int mode, uint32_t c,
                    int dir, int which, uint32_t *total_in_region, double share /* of symbol c in the text */,
                    const uint32_t *resume = nullptr /* {lo, hi}: the range an unattended pass left of this region: carry on from it */)
{
    sx_ctx *ctx = st.ctx;
    bool first = true;
    uint32_t bound_tiles = sx_div_up(region_entries ? region_entries : 1, kIndTile);
    // queued rounds per batch: until the expected round size (a run of c continues with
    // probability ~1/#symbols) is down to one tile; the tail kernel takes it from there
    int spec = 1;
    {
        const int sh = st.small_alphabet ? 2 : 6;
        while (spec < kMaxSpec && (bound_tiles >> (sh * spec)) >= 1) ++spec;
    }
    if (!st.small_alphabet) {
        // by the symbol's share of the text: rounds are queued until the one handed to the tail kernel is expected to hold
        // an eighth of what the kernel takes (a pass queued as a whole has nobody to queue one more: 1 GiB of 20 symbols,
        // round 3 of a bucket expected at 5400 entries, beyond 8192 in two buckets -- both passes ran twice).  Round 4: the
        // tail kernel takes rounds of up to kTailMulti tiles (32 768 entries), so an unattended bucket's rounds are queued
        // by the share alone (a byte text: one round, then the tail kernel; the chained launch in between is gone) -- a
        // bucket whose runs make the rounds shrink more slowly than its share says leaves word and is carried on attended,
        // with the longer queue.
        double expect = (double)region_entries * share;
        int by_share = 1;
        // (unattended: half of what the tail kernel takes -- a round's size is a sum of independent draws, and one that is
        //  too long after all is carried on attended; attended: an eighth, nobody queues one more)
        const double tail_takes = (double)(kTailMulti * kTailEntries) / ((st.unattended && !resume) ? 2.0 : 8.0);
        while (by_share < kMaxSpec && expect > tail_takes) ++by_share, expect *= share;
        if (st.unattended && !resume) spec = by_share;
        else if (by_share > spec) spec = by_share;
    }
    // Every batch ends with the tail kernel, which runs kTailIters rounds unless the range empties first, and a round
    // consumes one symbol of every run it follows: a bucket cannot need more batches than this (a device fault that
    // keeps the range alive must not keep the host here for ever).
    const uint64_t max_batches = 2 * (st.N / kTailIters) + 64; // (rounds too long for the tail consume > 8192 symbols each)
    uint32_t r[2] = {0, 0};
    bool resuming = resume != nullptr;
    if (resuming) r[0] = resume[0], r[1] = resume[1];
    for (uint64_t batch = 0;; ++batch) {
        if (batch > max_batches) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: the rounds of a bucket did not come to an end");
        if (!resuming) {
        if (first && st.hoist) {
            // (the tail kernel of the bucket before has done it, in an unattended pass: launch_tail)
            if (st.begun_c != (int)c)
                sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, bucket_begin_kernel, dim3(1), dim3(256), st.ranges, st.cursor[st.par],
                          (const uint32_t *)st.d_begin, (const uint32_t *)st.hoist_E, (const uint32_t *)st.hoist_tot, st.nk, c, st.hoist_from,
                          dir, st.tickets, (uint32_t)(kMaxSpec + 2), (const uint32_t *)(st.unattended ? st.poison : nullptr), st.hoist_err);
            st.begun_c = -1;
            st.hoist_from = c;
        } else if (first)
            sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, set_range_kernel, dim3(1), dim3(1), st.ranges, fixed_bound, fixed_bound,
                      (const uint32_t *)st.cursor[st.par], (int)c, which, st.tickets, (uint32_t)(kMaxSpec + 2),
                      (const uint32_t *)(st.unattended ? st.poison : nullptr));
        else
            SX_CHECK(hipMemsetAsync(st.tickets, 0, (kMaxSpec + 2) * sizeof(uint32_t), ctx->stream));
        const bool batched = st.small_alphabet && st.batch_on && (mode == MODE_L_FROM_L || mode == MODE_S_FROM_S);
        if (batched) {
            // First the rounds expected to be large, a launch each -- round k of a bucket holds about share^k of its region
            // (three launches, and no chained one behind them where the expectation is beyond doubt) --, then eight rounds by
            // one count / scan / scatter (the rounds that moved next to nothing and cost a launch chain each), the tail
            // kernel for what eight rounds leave of a range of a million.  A range the tail cannot hold (runs longer than the
            // rounds taken: poly-A, microsatellites) comes round again: eight more rounds, some chained ones, the tail.
            int slot = 0;
            if (first) {
                double expect = (double)region_entries;
                // (SX_FLAG_INDUCE_BATCH_MIN, tests: that bound here too, and every such round in the three-launch-only form)
                const double batch_from = ctx->induce_batch_min >= 0 ? (double)ctx->induce_batch_min : (double)kBatchFrom;
                for (int k = 0; k < 6 && expect > batch_from; ++k, expect *= share) {
                    uint32_t tb = sx_div_up((uint64_t)(expect * 2.0 < (double)region_entries ? expect * 2.0 : (double)region_entries), kIndTile);
                    if (tb < 256) tb = bound_tiles < 256 ? bound_tiles : 256;
                    const bool sure = ctx->induce_batch_min >= 0 || expect > 4.0 * (double)st.chain_max;
                    launch_round<WT>(st, st.SA, st.WN, slot, slot + 1, tb, tb, rev, mode, c, dir, 1, 0, sure ? 1 : 0);
                    ++slot;
                }
            }
            launch_batch<WT>(st, slot, slot + 1, bound_tiles, mode, c);
            ++slot;
            if (!first) {
                for (int k = 0; k < 3; ++k, ++slot)
                    launch_round<WT>(st, st.SA, st.WN, slot, slot + 1, bound_tiles < 256 ? bound_tiles : 256, 0, rev, mode, c, dir, 1, 1);
            }
            spec = slot;
        }
        for (int k = 0; k < spec && !batched; ++k) {
            uint32_t tb = bound_tiles >> k;
            const uint32_t floor_tiles = bound_tiles < 256 ? bound_tiles : 256;
            if (tb < floor_tiles) tb = floor_tiles;
            // a run of c's continues with the probability of c: expect round k to hold share^k of the region
            // (twice that, to be on the safe side, decides whether the three-launch form is queued as well: a launch
            // that finds nothing to do still costs 5 us, and there were 20 of them per bucket)
            double expect = 2.0 * (double)bound_tiles;
            for (int i = 0; i < k; ++i) expect *= share;
            const uint32_t likely = expect < (double)bound_tiles ? (uint32_t)expect : bound_tiles;
            // (the first round of a large region is large beyond doubt: the three-launch form alone)
            const bool sure = first && k == 0 && (uint64_t)region_entries > 16ull * st.chain_max;
            launch_round<WT>(st, st.SA, st.WN, k, k + 1, tb, first && k == 0 ? bound_tiles : likely, rev, mode, c, dir, 1, 0, sure ? 1 : 0);
        }
        launch_tail<WT>(st, spec, spec + 1, rev, mode, c, dir);
        if (st.unattended) { // the tail kernel ends nearly every bucket; one that it does not leaves word (tail_report)
            if (total_in_region) *total_in_region = 0xFFFFFFFFu;
            return 0;
        }
        SX_TRY(sx_readback(ctx, st.ranges + 2 * (spec + 1), 2, r));
        if (r[1] == r[0]) {
            if (total_in_region) *total_in_region = dir > 0 ? r[1] : r[0];
            return 0;
        }
        } // (!resuming)
        resuming = false;
        // a long run of symbol c: carry on from the last range
        sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, set_range_kernel, dim3(1), dim3(1), st.ranges, r[0], r[1],
                  (const uint32_t *)st.cursor[st.par], (int)c, 0, (uint32_t *)nullptr, 0u, (const uint32_t *)nullptr);
        if (r[1] - r[0] <= kRunEntries) {
            // a handful of entries deep inside runs: the device-wide jump, twice (a run may be longer than one probe looks)
            for (int rep = 0; rep < 2; ++rep) {
                SX_CHECK(hipMemsetAsync(st.run_len, 0xFF, sizeof(uint32_t), ctx->stream));
                const uint64_t look = st.N < (uint64_t)kRunProbe ? st.N : (uint64_t)kRunProbe; // (no run is longer than the text)
                // the nearest 4096 symbols first, by one workgroup per entry: most runs end there, and the workgroups of the
                // far probe then leave at their first look (started together they would all read their first piece)
                sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, run_probe_kernel, dim3(1, r[1] - r[0]), dim3(kBlock), st.T,
                          (const uint32_t *)st.SA, (const uint32_t *)st.ranges, c, 0u,
                          (uint32_t)(look < kRunProbeChunk ? look : kRunProbeChunk), st.run_len);
                if (look > kRunProbeChunk) {
                    const uint32_t far_chunks = sx_div_up(look - kRunProbeChunk, kRunProbeChunk);
                    sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, run_probe_kernel,
                              dim3(far_chunks < kRunProbeGrid ? far_chunks : kRunProbeGrid, r[1] - r[0]), dim3(kBlock), st.T,
                              (const uint32_t *)st.SA, (const uint32_t *)st.ranges, c, kRunProbeChunk, (uint32_t)look, st.run_len);
                }
                sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, run_fill_kernel<WT>, dim3(4096), dim3(kBlock), st.T, st.SA, st.WN, st.BW,
                          (const uint32_t *)st.ranges, (const uint32_t *)st.cursor[st.par], c, rev, dir, st.cfg,
                          (const uint32_t *)st.run_len);
                sx_launch(ctx, SX_KC_INDUCE_CHAIN, 0, run_commit_kernel, dim3(1), dim3(1), st.ranges, st.cursor[st.par], c, dir,
                          (const uint32_t *)st.run_len);
            }
        }
        first = false;
    }
}

template <class WT>
int induce_typed(sx_ctx *ctx, const sx_text_info &ti, uint32_t sigma, const uint32_t *sorted_lms,
                 const void *seed_windows, bool seed_windows_u32, uint32_t *SA, uint8_t *bwt_out, sx_arena &arena, wnd_cfg cfg)
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
    st.par = 0;
    st.WN = arena.take<WT>(N);
    st.BW = bwt_out ? bwt_out : arena.take<uint8_t>(N);
    st.N = N;
    st.m = ti.m;
    // (seed_windows_u32: the prefix-key sort's 32-bit words for a text whose windows are 64-bit: widened below)
    const bool widen = seed_windows && seed_windows_u32 && sizeof(WT) == 8;
    WT *seedW = seed_windows && !widen ? (WT *)seed_windows : arena.take<WT>(ti.m ? ti.m : 1);
    st.cursor[0] = arena.take<uint32_t>(256);
    st.cursor[1] = arena.take<uint32_t>(256);
    st.ranges = arena.take<uint32_t>(2 * (kMaxSpec + 3));
    st.tickets = arena.take<uint32_t>(kMaxSpec + 3);
    st.run_len = arena.take<uint32_t>(4);
    if (!st.WN || !st.BW || !seedW || !st.cursor[0] || !st.cursor[1] || !st.ranges || !st.tickets || !st.run_len)
        return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small");

    // bucket boundaries on the host (sa_is.c:176-201)
    uint32_t begin[257], lms_off[257];
    begin[0] = 0;
    lms_off[0] = 0;
    uint32_t largest = 1;
    for (uint32_t c = 0; c < 256; ++c) {
        begin[c + 1] = begin[c] + ti.h_all[c];
        lms_off[c + 1] = lms_off[c] + ti.h_lms[c];
        if (ti.h_all[c] > largest) largest = ti.h_all[c];
    }
    // look-back status words: one per (tile, bucket) of the largest round
    // The look-back walk costs a few microseconds per tile, so long rounds are better off with
    // the three launches: beyond 256 tiles when a wave walks back for each of <= 8 buckets,
    // With more than 8 buckets every round beyond four times what the tail kernel takes (32 768 entries, the second
    // round of a byte text's 4 M-entry buckets) goes to the radix-pass form (induce_wide_*): one look-back thread per
    // bucket and tile made a 2 M-entry round of a 255-symbol text cost 100 us and more, but a round of a few tiles is
    // one launch of 20 us where the three took 50 (1 GiB of bytes through the induction: 122 -> 111 ms).
    st.chain_max = ctx->chain_max_override >= 0 ? (uint32_t)ctx->chain_max_override
                                                : (st.small_alphabet ? 256u * (uint32_t)kIndTile : 4u * kTailEntries);
    const size_t status_words = ((size_t)sx_div_up(largest, kIndTile) + 2) * nk + kChainHeader; // any round may be chained
    st.stride = sx_div_up(largest, kIndTile) + 1;
    st.hist = st.whist = st.wsums = st.bhist = st.btotals = nullptr;
    st.batch_on = ctx->induce_batch_off ? 0 : 1;
    if (st.small_alphabet) {
        st.hist = arena.take<uint32_t>((size_t)nk * st.stride);
        st.bhist = arena.take<uint32_t>((size_t)kBatchRows * st.stride);
        st.btotals = arena.take<uint32_t>(kBatchRows);
        if (!st.hist || !st.bhist || !st.btotals) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small (tile counts)");
    } else {
        const size_t wt = (size_t)sx_div_up(largest, kWideTile) + 2;
        st.whist = arena.take<uint32_t>(wt * 256);
        st.wsums = arena.take<uint32_t>((wt / kWideChunk + 2) * 256);
        if (!st.whist || !st.wsums) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small (tile counts)");
    }
    SX_TRY(sx_chain_slab(ctx, SX_SLAB_CHAIN, status_words * 8));
    st.status = (uint64_t *)ctx->slab[SX_SLAB_CHAIN].p;
    SX_CHECK(hipMemsetAsync(st.status, 0, sizeof(uint64_t), ctx->stream)); // the time-out word

    // windows of the sorted LMS suffixes: the only systematic text access of both passes, unless
    // they already came along with the sort keys (sx_lmssort.hip)
    if (widen)
        sx_launch(ctx, SX_KC_INDUCE_GATHER, ti.m * 12, widen_windows_kernel, dim3(sx_div_up(ti.m, kBlock * 4)), dim3(kBlock),
                  (const uint32_t *)seed_windows, (uint64_t)ti.m, (uint64_t *)seedW);
    else if (!seed_windows)
        sx_launch(ctx, SX_KC_INDUCE_GATHER, ti.m * (4 + sizeof(WT) + 16), fill_windows_kernel<WT>,
                  dim3(sx_div_up(ti.m, kBlock)), dim3(kBlock), ti.T, sorted_lms, ti.m, cfg, seedW);
    // the sentinel suffix (sa_is.c:463: SA[0] = n)
    sx_launch(ctx, SX_KC_MISC, 0, set_entry_kernel<WT>, dim3(1), dim3(1), SA, st.WN, st.BW, (uint32_t)ti.n, ti.T, cfg);

    // Unattended passes.  After a bucket's queued rounds the host used to read the bucket's last range back and wait
    // (20 - 30 us of idle device: 16 times a build at 5 buckets, 1000 times at 256) -- almost always to learn that the tail
    // kernel had finished the bucket.  Now the buckets are queued one behind the other.  A tail kernel that cannot finish
    // its bucket (runs of a symbol longer than its steps and jumps reach, or more entries alive than it holds: thousands
    // of poly-A tracts) leaves word: the bucket and its last range, on the device and in the host's pinned page
    // (tail_report).  From then on every set_range_kernel leaves its range empty, and launches over an empty range only
    // carry the cursors on -- the device's state stays what it was when the bucket stopped.  The host looks at the
    // pinned word between buckets (a plain load, no wait), stops queuing, reads the record, carries that bucket on
    // attended (read-backs, device-wide run jumps) and goes on unattended behind it.  Texts in which the classification
    // saw a run fill a whole 4096-symbol tile are attended from the start.
    st.poison = arena.take<uint32_t>(4);
    if (!st.poison) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small");
    st.host_poison = ctx->h_pin + 1040;
    // More than 8 buckets: every bucket's other-region round (its LMS seeds in the L pass, its L-type entries in the S
    // pass) up front, all buckets in one count / offsets / scatter, placed by the text's bigram counts (hoist_*_kernel)
    st.hoist = (!st.small_alphabet && !ctx->induce_no_hoist) ? 1 : 0;
    st.d_begin = st.hoist_E = st.hoist_tot = st.hoist_err = nullptr;
    st.hoist_from = 0;
    st.next_c = st.begun_c = -1;
    uint32_t *hz_BG = nullptr, *hz_EL = nullptr, *hz_ES = nullptr, *hz_tot = nullptr, *hz_dbase = nullptr, *hz_hist = nullptr,
             *hz_desc = nullptr;
    uint32_t h_desc[2][768]; // per pass: lo[256], len[256], first hist row[256] of every bucket's region (uploaded; alive to the end)
    uint32_t hz_rows[2] = {0, 0}, hz_most[2] = {1, 1};
    if (st.hoist) {
        for (int pass = 0; pass < 2; ++pass) {
            uint32_t row = 0;
            for (uint32_t c = 0; c < 256; ++c) {
                const uint32_t len = c < nk ? (pass == 0 ? ti.h_lms[c] : ti.h_l[c]) : 0u;
                h_desc[pass][c] = pass == 0 ? lms_off[c] : begin[c];
                h_desc[pass][256 + c] = len;
                h_desc[pass][512 + c] = row;
                const uint32_t tiles = sx_div_up(len, kWideTile);
                row += tiles;
                if (tiles > hz_most[pass]) hz_most[pass] = tiles;
            }
            hz_rows[pass] = row;
        }
        st.d_begin = arena.take<uint32_t>(260);
        hz_BG = arena.take<uint32_t>(65536);
        hz_EL = arena.take<uint32_t>(65536);
        hz_ES = arena.take<uint32_t>(65536);
        hz_tot = arena.take<uint32_t>(65536);
        hz_dbase = arena.take<uint32_t>(65536);
        hz_desc = arena.take<uint32_t>(2 * 768);
        st.hoist_err = arena.take<uint32_t>(4);
        hz_hist = arena.take<uint32_t>(((size_t)(hz_rows[0] > hz_rows[1] ? hz_rows[0] : hz_rows[1]) + 2) * 256);
        if (!st.d_begin || !hz_BG || !hz_EL || !hz_ES || !hz_tot || !hz_dbase || !hz_desc || !st.hoist_err || !hz_hist)
            return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: arena too small (up-front rounds)");
        SX_CHECK(hipMemcpyAsync(st.d_begin, begin, 257 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        SX_CHECK(hipMemcpyAsync(hz_desc, h_desc, sizeof h_desc, hipMemcpyHostToDevice, ctx->stream));
        SX_CHECK(hipMemsetAsync(hz_BG, 0, 65536 * sizeof(uint32_t), ctx->stream));
        SX_CHECK(hipMemsetAsync(st.hoist_err, 0, 4 * sizeof(uint32_t), ctx->stream));
        // the text's bigram counts (one pass per 32768 / nk rows of the matrix), then where every bucket's groups end
        uint32_t bg_grid = (uint32_t)sx_div_up(sx_div_up(ti.n ? ti.n : 1, 16), (uint64_t)kBigramThreads * 16);
        if (bg_grid > kBigramGrid) bg_grid = kBigramGrid;
        sx_launch(ctx, SX_KC_INDUCE_GATHER, ti.n * (uint64_t)sx_div_up(nk, kBigramWords / nk), bigram_kernel, dim3(bg_grid),
                  dim3(kBigramThreads), ti.T, (uint64_t)ti.n, nk, hz_BG);
        sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, hoist_tables_kernel, dim3(1), dim3(256), (const uint32_t *)hz_BG, (const uint32_t *)st.d_begin, nk,
                  hz_EL, hz_ES);
    }
    // all buckets' other-region rounds of a pass: one count, one offsets, one scatter (queued at the start of the pass)
    auto hoisted_rounds = [&](int pass) -> int {
        const uint32_t *desc = hz_desc + pass * 768;
        const int rev = pass, mode = pass == 0 ? MODE_L_FROM_LMS : MODE_S_FROM_L, dir = pass == 0 ? +1 : -1;
        const uint32_t *srcP = pass == 0 ? sorted_lms : (const uint32_t *)SA;
        const WT *srcW = pass == 0 ? (const WT *)seedW : (const WT *)st.WN;
        const uint8_t *srcB = pass == 0 ? (const uint8_t *)nullptr : (const uint8_t *)st.BW;
        st.hoist_E = pass == 0 ? hz_EL : hz_ES;
        st.hoist_tot = hz_tot;
        st.hoist_from = pass == 0 ? 0u : nk - 1u;
        uint32_t gx = hz_most[pass];
        if (gx > kHoistGridX) gx = kHoistGridX;
        sx_launch(ctx, SX_KC_INDUCE_GATHER, 0, hoist_count_kernel<WT>, dim3(gx, nk), dim3(kWideThreads), srcW, srcB, desc, rev, mode, st.cfg,
                  hz_hist);
        sx_launch(ctx, SX_KC_INDUCE_SCAN, (uint64_t)hz_rows[pass] * 2048, hoist_offsets_kernel, dim3(nk), dim3(kBlock * kHoistOffGroups),
                  hz_hist, desc, (const uint32_t *)st.hoist_E, dir, hz_tot, hz_dbase);
        sx_launch(ctx, SX_KC_INDUCE_SCATTER, 0, hoist_scatter_kernel<WT, 8>, dim3(gx, nk), dim3(kWideThreads), srcP, srcW, desc, rev,
                  mode, st.cfg, st.T, (const uint32_t *)hz_hist, (const uint32_t *)hz_dbase, dir, st.SA, st.WN, st.BW,
                  (uint32_t)(pass == 0 ? 1 : 0));
        ctx->stats.induce_rounds++;
        return 0;
    };
    // what a pass leaves for the host, in one read-back: the unattended run's stop record (4 words), the cursors (nk), and
    // behind the S pass the look-back time-out word and the hoisted rounds' error word
    auto pass_end = [&](int pass, bool with_stop, uint32_t (&rec)[4], uint32_t (&cur)[256], uint32_t (&tail)[3]) -> int {
        const uint32_t *src[4];
        uint32_t cnt[4], page[4 + 256 + 3];
        int k = 0;
        if (with_stop) src[k] = (const uint32_t *)st.poison, cnt[k++] = 4;
        src[k] = (const uint32_t *)st.cursor[st.par], cnt[k++] = nk;
        if (pass == 1) {
            src[k] = (const uint32_t *)st.status, cnt[k++] = 2;
            if (st.hoist) src[k] = (const uint32_t *)st.hoist_err, cnt[k++] = 1;
        }
        SX_TRY(sx_readback_ranges(ctx, src, cnt, k, page));
        const uint32_t *q = page;
        if (with_stop) memcpy(rec, q, sizeof rec), q += 4;
        memcpy(cur, q, nk * sizeof(uint32_t)), q += nk;
        tail[0] = tail[1] = tail[2] = 0;
        if (pass == 1) {
            tail[0] = q[0], tail[1] = q[1];
            if (st.hoist) tail[2] = q[2];
        }
        return 0;
    };
    // both passes end with every cursor between its bucket's L and S suffixes (bucket 0 holds the sentinel's suffix alone,
    // which no pass induces)
    auto cursors_as_counted = [&](const uint32_t (&cur)[256]) -> bool {
        bool ok = true;
        for (uint32_t c = 1; c < nk; ++c)
            if (ti.h_all[c] && cur[c] != begin[c] + ti.h_l[c]) ok = false;
        return ok;
    };
    const bool unattended_ok = ctx->induce_attended != 1 && ti.open_tiles == 0;
    auto stopped = [&]() -> bool { return st.unattended && *(volatile uint32_t *)st.host_poison != 0; };
    // ---- L pass: buckets ascending, cursors at the bucket heads; from bucket `from` on (resume: that bucket's L region
    // carries on from the range an unattended run left)
    auto pass_L = [&](uint32_t from, const uint32_t *resume) -> int {
        for (uint32_t c = from; c < nk; ++c) {
            if (ti.h_all[c] == 0) continue;
            if (stopped()) return 0;
            const bool carry_on = resume && c == from;
            if (carry_on) st.hoist_from = c; // (its rounds are carried on from where they stopped: no bucket_begin_kernel)
            if (ti.h_l[c]) {
                uint32_t head_end = 0;
                st.unattended = (unattended_ok && !carry_on) ? 1 : 0;
                st.next_c = -1; // (the next bucket with an L region: its rounds are opened by this bucket's tail kernel)
                for (uint32_t c2 = c + 1; c2 < nk && st.next_c < 0; ++c2)
                    if (ti.h_all[c2] && ti.h_l[c2]) st.next_c = (int)c2;
                SX_TRY(run_self_rounds<WT>(st, begin[c], ti.h_l[c], 0, MODE_L_FROM_L, c, +1, 1, &head_end, (double)ti.h_all[c] / (double)N,
                                           carry_on ? resume : nullptr));
                if (!st.unattended && head_end - begin[c] != ti.h_l[c])
                    return sx_fail_msg(ctx, SX_E_INTERNAL, "induce L: bucket did not receive its L-type count");
                st.unattended = unattended_ok ? 1 : 0;
            }
            if (ti.h_lms[c] && !st.hoist) {
                sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, set_range_kernel, dim3(1), dim3(1), st.ranges, lms_off[c], lms_off[c + 1],
                          (const uint32_t *)st.cursor[st.par], (int)c, 0, st.tickets, 1u, (const uint32_t *)(st.unattended ? st.poison : nullptr));
                // (the round's size is known: the one form that takes it, and no launch that finds nothing to do)
                launch_round<WT>(st, sorted_lms, seedW, 0, -1, sx_div_up(ti.h_lms[c], kIndTile), sx_div_up(ti.h_lms[c], kIndTile), 0,
                                 MODE_L_FROM_LMS, c, +1, 0, 0, ti.h_lms[c] > st.chain_max ? 1 : 0);
            }
        }
        return 0;
    };
    // ---- S pass: buckets descending, cursors at the bucket ends -------------------------
    auto pass_S = [&](uint32_t from, const uint32_t *resume) -> int {
        for (uint32_t cc = from + 1; cc-- > 0;) {
            const uint32_t c = cc;
            if (ti.h_all[c] == 0) continue;
            if (stopped()) return 0;
            const bool carry_on = resume && c == from;
            if (carry_on) st.hoist_from = c;
            const uint32_t n_s = ti.h_all[c] - ti.h_l[c];
            if (c > 0 && n_s) {
                uint32_t tail_end = 0;
                st.unattended = (unattended_ok && !carry_on) ? 1 : 0;
                st.next_c = -1; // (the next bucket with an S region)
                for (uint32_t c2 = c; c2-- > 1 && st.next_c < 0;)
                    if (ti.h_all[c2] && ti.h_all[c2] - ti.h_l[c2]) st.next_c = (int)c2;
                SX_TRY(run_self_rounds<WT>(st, begin[c + 1], n_s, 1, MODE_S_FROM_S, c, -1, 2, &tail_end, (double)ti.h_all[c] / (double)N,
                                           carry_on ? resume : nullptr));
                if (!st.unattended && begin[c + 1] - tail_end != n_s)
                    return sx_fail_msg(ctx, SX_E_INTERNAL, "induce S: bucket did not receive its S-type count");
                st.unattended = unattended_ok ? 1 : 0;
            }
            if (ti.h_l[c] && !st.hoist) {
                sx_launch(ctx, SX_KC_INDUCE_SCAN, 0, set_range_kernel, dim3(1), dim3(1), st.ranges, begin[c],
                          begin[c] + ti.h_l[c], (const uint32_t *)st.cursor[st.par], (int)c, 0, st.tickets, 1u,
                          (const uint32_t *)(st.unattended ? st.poison : nullptr));
                launch_round<WT>(st, SA, st.WN, 0, -1, sx_div_up(ti.h_l[c], kIndTile), sx_div_up(ti.h_l[c], kIndTile), 1,
                                 MODE_S_FROM_L, c, -1, 0, 0, ti.h_l[c] > st.chain_max ? 1 : 0);
            }
        }
        return 0;
    };
    ctx->stats.long_runs = ti.open_tiles ? 1u : 0u;
    for (int pass = 0; pass < 2; ++pass) {
        st.unattended = unattended_ok ? 1 : 0;
        if (pass == 1) SX_CHECK(hipStreamSynchronize(ctx->stream)); // (`begin`, the L pass's upload source, may still be in use)
        SX_CHECK(hipMemcpyAsync(st.cursor[st.par], pass == 0 ? begin : begin + 1, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        if (st.hoist) SX_TRY(hoisted_rounds(pass));
        uint32_t from = pass == 0 ? 0u : nk - 1u, rec[4] = {0, 0, 0, 0}, cur[256], tail[3];
        const uint32_t *resume = nullptr;
        for (uint32_t attempt = 0;; ++attempt) {
            if (attempt > 2 * nk + 4) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: a pass did not come to its end");
            if (unattended_ok) {
                *(volatile uint32_t *)st.host_poison = 0;
                SX_CHECK(hipMemsetAsync(st.poison, 0, 4 * sizeof(uint32_t), ctx->stream));
            }
            st.next_c = st.begun_c = -1;
            SX_TRY(pass == 0 ? pass_L(from, resume) : pass_S(from, resume));
            SX_TRY(pass_end(pass, unattended_ok, rec, cur, tail));
            if (!unattended_ok || !rec[0]) break;
            // bucket rec[1] stopped with the range [rec[2], rec[3]) alive: carry it on attended, then the buckets behind it
            ctx->stats.induce_redo++;
            from = rec[1];
            resume = rec + 2;
        }
        if (!cursors_as_counted(cur))
            return sx_fail_msg(ctx, SX_E_INTERNAL, pass == 0 ? "induce L: a bucket did not receive its L-type count"
                                                             : "induce S: a bucket did not receive its S-type count");
        if (pass == 1 && tail[0]) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: a look-back wait timed out");
        if (pass == 1 && tail[2]) return sx_fail_msg(ctx, SX_E_INTERNAL, "induce: a bucket's cursor is not where the text's bigram counts put it");
    }

    if (ctx->prof_on) {
        // Algorithmic bytes of the two passes (the launches themselves were queued with bounds, not
        // sizes): the L pass scans every L-type entry and every LMS seed, the S pass every entry but
        // the sentinel's; every suffix is written once.  The counting launches read the symbol bytes (windows for the seeds), the
        // scatter launches the (position, window) pairs and write a symbol byte along; the few entries that went through the chained
        // rounds are booked here too.
        uint64_t n_l = 0;
        for (uint32_t c = 0; c < nk; ++c) n_l += ti.h_l[c];
        const uint64_t scanned = n_l + ti.m + (N - 1);
        // (with events around one class only -- bench.py's timed region -- that class alone is booked)
        if (ctx->prof_only < 0 || ctx->prof_only == SX_KC_INDUCE_GATHER)
            ctx->kstat[SX_KC_INDUCE_GATHER].alg_bytes += (scanned - ti.m) + ti.m * sizeof(WT); // symbol bytes; seeds: windows
        if (ctx->prof_only < 0 || ctx->prof_only == SX_KC_INDUCE_SCATTER)
            ctx->kstat[SX_KC_INDUCE_SCATTER].alg_bytes += (scanned + N) * (4 + sizeof(WT)) + N;
    }
    // st.BW now holds text[SA[i]-1] for every slot: the BWT (bwt.c:13-20), written along with the entries
    return 0;
}
} // namespace

// BWT from one-symbol windows of all suffixes in suffix-array order (the direct sort of wide alphabets)
int sx_bwt_from_seed_windows(sx_ctx *ctx, const uint32_t *seedw, uint64_t N, uint32_t maxc, uint8_t *bwt_out)
{
    wnd_cfg cfg;
    (void)sx_window_cfg(maxc, cfg);
    cfg.CW = 1;
    sx_launch(ctx, SX_KC_BWT_GATHER, N * 5, bwt_from_windows_kernel<uint32_t>, dim3(sx_div_up(N, kBlock * 16)), dim3(kBlock),
              seedw, N, cfg, bwt_out);
    return 0;
}

int sx_induce(sx_ctx *ctx, const sx_text_info &ti, uint32_t sigma, const uint32_t *sorted_lms,
              const void *seed_windows, bool seed_windows_u32, uint32_t *SA, uint8_t *bwt_out, sx_arena &arena)
{
    if (ti.N > 0xFFFFFFFFull) return sx_fail_msg(ctx, SX_E_ARG, "induce: n exceeds 32-bit positions");
    wnd_cfg cfg;
    const bool wide = sx_window_cfg(ti.maxc, cfg);
    if (!wide) return induce_typed<uint32_t>(ctx, ti, sigma, sorted_lms, seed_windows, false, SA, bwt_out, arena, cfg);
    return induce_typed<uint64_t>(ctx, ti, sigma, sorted_lms, seed_windows, seed_windows_u32, SA, bwt_out, arena, cfg);
}
