// sx_reduce.hip -- names, reduced string, and the suffix sort of the reduced string.
//
// Role of stralg/sa_is.c:295-336 (reduce_SA: name LMS substrings in sorted
// order, emit the reduced string) and of the recursion sa_is.c:370-387.  The
// reference recurses into sort_SA; level >= 1 alphabets have 10^4..10^6
// symbols, which does not fit the LDS-histogram
// design, so the reduced problem is solved by prefix doubling on top of the
// radix sort instead: rank pairs (rank[i], rank[i+h]) are sorted, groups are
// refined, h doubles; only suffixes in groups of more than one stay active.
// The suffix array is unique, so the result equals the reference's.
#include <vector>
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_scan.hpp"
#include "sx_internal.hpp"
#include "sx_window.hpp"

namespace sx {

// ---- names ---------------------------------------------------------------------
struct InKeyBoundary {
    const uint64_t *ks;
    __device__ __forceinline__ uint32_t operator()(uint64_t j) const
    {
        return (j > 0 && ks[j] != ks[j - 1]) ? 1u : 0u;
    }
};
struct OutNames {
    const uint32_t *vs;
    uint32_t *R;
    __device__ __forceinline__ void operator()(uint64_t j, uint32_t excl, uint32_t v) const
    {
        R[vs[j]] = excl + v; // inclusive count of boundaries = dense name
    }
};

struct OutNamesInOrder { // (the names in sorted order: the two-pass scatter takes them to their places)
    uint32_t *names;
    __device__ __forceinline__ void operator()(uint64_t j, uint32_t excl, uint32_t v) const { names[j] = excl + v; }
};

// ---- prefix doubling -------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pack_names_kernel(const uint32_t *__restrict__ R, uint64_t M,
                                                            uint32_t b, uint32_t q, uint64_t *__restrict__ keys,
                                                            uint32_t *__restrict__ vals)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    uint64_t acc = 0;
    for (uint32_t s = 0; s < q; ++s) {
        const uint64_t sym = i + s < M ? (uint64_t)R[i + s] : 0ull;
        acc = (acc << b) | sym;
    }
    keys[i] = acc;
    vals[i] = (uint32_t)i;
}

// (every kernel that sets head flags also counts them: the host chooses the next round's form by the mean group size.
// One add per wave, spread over kHeadBins words: five million adds to one word took half a second.)
constexpr uint32_t kHeadBins = 1024; // (one read-back of a page)
__device__ __forceinline__ void count_heads(bool is_head, uint32_t *__restrict__ n_heads)
{
    const uint64_t b = __ballot(is_head ? 1 : 0);
    if (b && lane_id() == __ffsll((unsigned long long)b) - 1)
        atomicAdd(&n_heads[(blockIdx.x * (uint32_t)kWavesPerBlock + (uint32_t)wave_id()) & (kHeadBins - 1u)], (uint32_t)__popcll(b));
}
__global__ __launch_bounds__(kBlock) void init_groups_kernel(const uint64_t *__restrict__ ks,
                                                             const uint32_t *__restrict__ vs, uint64_t M,
                                                             uint32_t *__restrict__ sa_r, uint32_t *__restrict__ pos,
                                                             uint8_t *__restrict__ head, uint32_t *__restrict__ n_heads)
{
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    bool hd = false;
    if (j < M) {
        sa_r[j] = vs[j];
        pos[j] = (uint32_t)j;
        hd = j == 0 || ks[j] != ks[j - 1];
        head[j] = hd ? 1 : 0;
    }
    count_heads(hd, n_heads);
}

// group id = SA position of the group's first member, carried by a max-scan
struct InHeadPos {
    const uint8_t *head;
    const uint32_t *pos;
    __device__ __forceinline__ uint32_t operator()(uint64_t t) const { return head[t] ? pos[t] + 1u : 0u; }
};
struct OutGidRank {
    const uint32_t *sa;
    uint32_t *gid, *rank;
    __device__ __forceinline__ void operator()(uint64_t t, uint32_t excl, uint32_t v) const
    {
        const uint32_t g = (excl > v ? excl : v) - 1u;
        gid[t] = g;
        rank[sa[t]] = g;
    }
};

struct OutGidOnly { // (the ranks follow by the two-pass scatter)
    uint32_t *gid;
    __device__ __forceinline__ void operator()(uint64_t t, uint32_t excl, uint32_t v) const { gid[t] = (excl > v ? excl : v) - 1u; }
};
// the same after a round: a member's rank is written only where its group's number changed.  A group's number is a
// list position inside the group (any one: the intervals of different groups do not overlap, so the numbers order the
// groups); the sub-group that holds the old number's position keeps it, and with it the ranks of all its members --
// in a collection of near-identical sequences most rounds only peel one member off a group, and the scattered write
// of every member's rank was the most expensive step of a round (8.7 of 20 ms at 3.1e8 names).  Groups split by the
// radix sorts are numbered by their first member's position: the first sub-group keeps the number, if the old one was
// the group's first position.  oldkeys: (old group number << rb | rank ahead), slot by slot
struct OutGidRankChanged {
    const uint32_t *sa;
    uint32_t *gid, *rank;
    const uint64_t *oldkeys;
    uint32_t rb;
    __device__ __forceinline__ void operator()(uint64_t t, uint32_t excl, uint32_t v) const
    {
        const uint32_t g = (excl > v ? excl : v) - 1u;
        gid[t] = g;
        if ((uint32_t)(oldkeys[t] >> rb) != g) rank[sa[t]] = g;
    }
};
// the members the radix sorts ordered inside a wave-tier round (sorted sub-list slot j = list slot sub_t[j])
struct InSubHeadPos {
    const uint64_t *k2s;
    const uint32_t *sub_t, *pos2;
    __device__ __forceinline__ uint32_t operator()(uint64_t j) const
    {
        return (j == 0 || k2s[j] != k2s[j - 1]) ? pos2[sub_t[j]] + 1u : 0u;
    }
};
struct OutSubGidRank {
    const uint64_t *k2s;
    const uint32_t *sas, *sub_t;
    uint32_t *gid, *rank;
    uint32_t rb;
    __device__ __forceinline__ void operator()(uint64_t j, uint32_t excl, uint32_t v) const
    {
        const uint32_t g = (excl > v ? excl : v) - 1u;
        gid[sub_t[j]] = g;
        if ((uint32_t)(k2s[j] >> rb) != g) rank[sas[j]] = g;
    }
};

// keep the members of groups with more than one element
struct InKeep {
    const uint8_t *head;
    uint64_t A;
    __device__ __forceinline__ uint32_t operator()(uint64_t t) const
    {
        const bool single = head[t] && (t + 1 == A || head[t + 1]);
        return single ? 0u : 1u;
    }
};
struct OutKeep {
    const uint32_t *pos, *sa, *gid, *rank;
    uint32_t *pos2, *val2;
    uint64_t *key2;
    uint64_t h, M;
    uint32_t rb;
    __device__ __forceinline__ void operator()(uint64_t t, uint32_t excl, uint32_t v) const
    {
        if (!v) return;
        const uint32_t s = sa[t];
        pos2[excl] = pos[t];
        val2[excl] = s;
        // s + h < M for every member of a group of two or more (the sentinel name is unique)
        const uint64_t ahead = (uint64_t)s + h;
        key2[excl] = ((uint64_t)gid[t] << rb) | (uint64_t)(ahead < M ? rank[ahead] : 0u);
    }
};

__global__ __launch_bounds__(kBlock) void regroup_kernel(const uint64_t *__restrict__ k2s,
                                                         const uint32_t *__restrict__ sas,
                                                         const uint32_t *__restrict__ pos2, uint64_t A,
                                                         uint32_t *__restrict__ sa_r, uint8_t *__restrict__ head2,
                                                         uint32_t *__restrict__ n_heads)
{
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    bool hd = false;
    if (t < A) {
        hd = t == 0 || k2s[t] != k2s[t - 1];
        head2[t] = hd ? 1 : 0;
        sa_r[pos2[t]] = sas[t];
    }
    count_heads(hd, n_heads);
}

// Small groups in one wave.  The active list is in suffix order, so a round only has to order every group's members
// by the rank h symbols ahead: the group part of the (group, rank) key is sorted already, and re-sorting all of it
// through eight radix passes is what a collection of near-identical sequences pays round after round (every suffix
// in a group of as many members as there are copies).  A wave takes a window of 64 consecutive list slots, windows
// start every kWaveStride slots; it owns the groups whose first member lies in the window's first kWaveStride slots
// and that end inside the window -- every group of up to 64 - kWaveStride + 1 members is owned by exactly one wave,
// longer ones by at most one.  A member's place is its group's first slot + the members with a smaller rank + the
// members with the same rank to its left, counted by reading the window's 64 (group, rank) pairs lane by lane.
// Members of groups no wave owns stay unmarked in done[] and take the radix sorts.
constexpr int kWaveStride = 48;
constexpr uint64_t kWaveTierMeanGroup = 24; // the waves take a round when its groups hold at most this many members on average
struct InNotDone {
    const uint8_t *done;
    __device__ __forceinline__ uint32_t operator()(uint64_t t) const { return done[t] ? 0u : 1u; }
};
struct OutNotDone {
    const uint64_t *key2;
    const uint32_t *val2;
    uint64_t *ksub;
    uint32_t *vsub, *sub_t;
    __device__ __forceinline__ void operator()(uint64_t t, uint32_t excl, uint32_t v) const
    {
        if (!v) return;
        ksub[excl] = key2[t];
        vsub[excl] = val2[t];
        sub_t[excl] = (uint32_t)t;
    }
};
__global__ __launch_bounds__(kBlock) void doubling_wave_groups_kernel(const uint64_t *__restrict__ key2,
                                                                      const uint32_t *__restrict__ val2,
                                                                      const uint32_t *__restrict__ pos2, uint64_t A,
                                                                      uint32_t rb, uint32_t *__restrict__ v_out,
                                                                      uint8_t *__restrict__ head2,
                                                                      uint32_t *__restrict__ sa_r,
                                                                      uint8_t *__restrict__ done,
                                                                      uint32_t *__restrict__ n_heads,
                                                                      uint32_t *__restrict__ gid_out,
                                                                      uint32_t *__restrict__ rank)
{
    const uint64_t base = ((uint64_t)blockIdx.x * kWavesPerBlock + (uint64_t)wave_id()) * kWaveStride;
    if (base >= A) return; // (the whole wave)
    const int l = lane_id();
    const uint64_t t = base + (uint64_t)l;
    const bool valid = t < A;
    const uint64_t key = valid ? key2[t] : ~0ull;
    const uint32_t val = valid ? val2[t] : 0u, ps = valid ? pos2[t] : 0u;
    constexpr uint32_t kNoGroup = 0xFFFFFFFFu; // (group ids are list positions: below 2^32 - 1)
    const uint32_t g = valid ? (uint32_t)(key >> rb) : kNoGroup;
    const uint32_t r = (uint32_t)(key & ((1ull << rb) - 1ull));
    // the slot before the window and the one after it, for the groups that cross its ends
    const uint32_t g_before = base > 0 ? (uint32_t)(key2[base - 1] >> rb) : kNoGroup;
    const uint32_t g_after = base + kWave < A ? (uint32_t)(key2[base + kWave] >> rb) : kNoGroup;
    uint32_t g_left = __shfl_up(g, 1, kWave);
    if (l == 0) g_left = g_before;
    // a slot opens a group when its group differs from its left neighbour's; the slot after the last one closes the list
    const uint64_t heads = __ballot((valid && g != g_left) || t == A ? 1 : 0);
    const uint64_t upto = heads & lanemask_le();
    const int first = upto ? 63 - __clzll((unsigned long long)upto) : -1; // lane of the group's first member
    const uint64_t later = heads & ~lanemask_le();
    const bool open_end = later == 0ull && g == g_after; // the group goes on behind the window
    const bool owned = valid && first >= 0 && first < kWaveStride && !open_end;
    // members of the group to the left and to the right, one lane further every step, as far as the longest owned
    // group reaches (a collection of 16 copies: 15 steps; a long group: the whole window, lane by lane)
    const int my_end = later ? __ffsll((unsigned long long)later) - 1 : kWave; // lane after the group's last member
    uint32_t reach = owned ? (uint32_t)(my_end - first) : 0u;
    reach = wave_reduce_max(reach);
    uint32_t less = 0, same_left = 0, same_right = 0;
    if (reach <= 24u) { // uniform
        for (uint32_t o = 1; o < reach; ++o) {
            const uint32_t gu = __shfl_up(g, o, kWave), ru = __shfl_up(r, o, kWave);
            const uint32_t gd = __shfl_down(g, o, kWave), rd = __shfl_down(r, o, kWave);
            const bool mate_u = (uint32_t)l >= o && gu == g, mate_d = (uint32_t)l + o < (uint32_t)kWave && gd == g;
            less += (mate_u && ru < r) ? 1u : 0u;
            same_left += (mate_u && ru == r) ? 1u : 0u;
            less += (mate_d && rd < r) ? 1u : 0u;
            same_right += (mate_d && rd == r) ? 1u : 0u;
        }
    } else {
#pragma unroll
        for (int d = 0; d < kWave; ++d) {
            const uint32_t gd = (uint32_t)__builtin_amdgcn_readlane((int)g, d), rd = (uint32_t)__builtin_amdgcn_readlane((int)r, d);
            const bool mate = gd == g;
            less += (mate && rd < r) ? 1u : 0u;
            same_left += (mate && rd == r && d < l) ? 1u : 0u;
            same_right += (mate && rd == r && d > l) ? 1u : 0u;
        }
    }
    const int at = first + (int)(less + same_left); // lane whose slot the member moves to
    const uint32_t p_at = __shfl(ps, at < 0 ? 0 : at, kWave);
    const uint32_t p_first = __shfl(ps, first < 0 ? 0 : first, kWave); // (a group's list positions are consecutive)
    if (owned) {
        const uint64_t slot = base + (uint64_t)at;
        v_out[slot] = val;
        head2[slot] = same_left == 0 ? 1 : 0; // first of its run of equal ranks (the group's first member included)
        sa_r[p_at] = val;
        done[slot] = 1;
        // the members of equal rank ahead form a sub-group at positions p_first + less ... + same: it keeps the group's
        // number if that position is one of its own (nothing to write then), and takes its middle position otherwise
        const uint32_t same = same_left + same_right + 1u, at_g = g - p_first;
        const uint32_t g_new = (at_g >= less && at_g < less + same) ? g : p_first + less + ((same - 1u) >> 1);
        gid_out[slot] = g_new;
        if (g_new != g) rank[val] = g_new;
    }
    count_heads(owned && same_left == 0, n_heads);
}

// the members ordered by the radix sorts go back to their slots: sorted slot j of the sub-list is list slot sub_t[j]
__global__ __launch_bounds__(kBlock) void regroup_sub_kernel(const uint64_t *__restrict__ k2s,
                                                             const uint32_t *__restrict__ sas,
                                                             const uint32_t *__restrict__ sub_t,
                                                             const uint32_t *__restrict__ pos2, uint64_t A3,
                                                             uint32_t *__restrict__ v_out, uint32_t *__restrict__ sa_r,
                                                             uint8_t *__restrict__ head2, uint32_t *__restrict__ n_heads)
{
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    bool hd = false;
    if (j < A3) {
        const uint32_t t = sub_t[j], s = sas[j];
        hd = j == 0 || k2s[j] != k2s[j - 1];
        head2[t] = hd ? 1 : 0;
        v_out[t] = s;
        sa_r[pos2[t]] = s;
    }
    count_heads(hd, n_heads);
}

// ---- sorted LMS suffixes -----------------------------------------------------------
struct InIsLms {
    const uint32_t *sa_r;
    const uint8_t *is_lms;
    __device__ __forceinline__ uint32_t operator()(uint64_t j) const { return is_lms[sa_r[j]]; }
};
struct OutLmsPos {
    const uint32_t *sa_r, *pos;
    uint32_t *out;
    __device__ __forceinline__ void operator()(uint64_t j, uint32_t excl, uint32_t v) const
    {
        if (v) out[excl] = pos[sa_r[j]];
    }
};

__global__ __launch_bounds__(kBlock) void gather_pos_kernel(const uint32_t *__restrict__ sa_r,
                                                            const uint32_t *__restrict__ pos, uint64_t M,
                                                            uint32_t *__restrict__ out, uint32_t *__restrict__ d_total)
{
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < M) out[j] = pos[sa_r[j]];
    if (j == 0) *d_total = (uint32_t)M;
}

// ---- sorted LMS suffixes with their windows: one random read a suffix ----------------------------------------------------
// The induction starts from (position, window of the symbols to its left) of every sorted LMS suffix.  pos[sa_r[j]]
// and then the text around that position were two random reads a suffix (gather_pos + fill_windows: 13.6 ms at
// 3.1e8).  The windows are cheap in *text* order (the samples' positions ascend: the text is read front to back),
// so they are made there, next to the position (cuts: no position), and one gather in suffix order fetches both.
constexpr uint32_t kNoLms = 0xFFFFFFFFu;
template <class WT> struct pos_wnd;
template <> struct pos_wnd<uint32_t> { uint32_t p, w; };                       // 8 bytes
template <> struct __attribute__((aligned(16))) pos_wnd<uint64_t> { uint32_t p, pad; uint64_t w; }; // 16 bytes

template <class WT>
__global__ __launch_bounds__(kBlock) void sample_windows_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ pos,
                                                               const uint8_t *__restrict__ is_lms, uint64_t M, wnd_cfg cfg,
                                                               pos_wnd<WT> *__restrict__ pw)
{
    const uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= M) return;
    const uint32_t p = pos[k];
    pos_wnd<WT> e = {};
    e.p = is_lms[k] ? p : kNoLms;
    e.w = (is_lms[k] && p) ? wnd_fill<WT>(T, p, cfg) : (WT)0;
    pw[k] = e;
}
// every sample is an LMS position: straight to the outputs
template <class WT>
__global__ __launch_bounds__(kBlock) void gather_pos_windows_kernel(const uint32_t *__restrict__ sa_r, const pos_wnd<WT> *__restrict__ pw,
                                                                   uint64_t M, uint32_t *__restrict__ out_pos, WT *__restrict__ out_w,
                                                                   pos_wnd<WT> *__restrict__ out_pw /* or null */, uint32_t *__restrict__ d_total)
{
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j < M) {
        const pos_wnd<WT> e = pw[sa_r[j]];
        if (out_pw) out_pw[j] = e;
        else out_pos[j] = e.p, out_w[j] = e.w;
    }
    if (j == 0 && d_total) *d_total = (uint32_t)M;
}
template <class WT> struct InHasLms {
    const pos_wnd<WT> *g;
    __device__ __forceinline__ uint32_t operator()(uint64_t j) const { return g[j].p != kNoLms ? 1u : 0u; }
};
template <class WT> struct OutPosWindow {
    const pos_wnd<WT> *g;
    uint32_t *out_pos;
    WT *out_w;
    __device__ __forceinline__ void operator()(uint64_t j, uint32_t excl, uint32_t v) const
    {
        if (v) out_pos[excl] = g[j].p, out_w[excl] = g[j].w;
    }
};

} // namespace sx

using namespace sx;

template <class WT>
static int sorted_lms_windows_typed(sx_ctx *ctx, const sx_text_info &ti, const uint32_t *sa_r, const uint32_t *pos, const uint8_t *is_lms,
                                    uint64_t M, uint64_t m, void *buf_a, void *buf_b, uint32_t *sorted_lms, const void **seed_windows,
                                    uint32_t *d_total, const wnd_cfg &cfg)
{
    pos_wnd<WT> *pw = (pos_wnd<WT> *)buf_a;
    sx_launch(ctx, SX_KC_DOUBLING, M * (5 + sizeof(pos_wnd<WT>)) + ti.N, sample_windows_kernel<WT>, dim3(sx_div_up(M, kBlock)), dim3(kBlock),
              ti.T, pos, is_lms, M, cfg, pw);
    if (M == m) {
        WT *seed = (WT *)buf_b;
        sx_launch(ctx, SX_KC_DOUBLING, M * (4 + 64 + 4 + sizeof(WT)), gather_pos_windows_kernel<WT>, dim3(sx_div_up(M, kBlock)), dim3(kBlock),
                  sa_r, (const pos_wnd<WT> *)pw, M, sorted_lms, seed, (pos_wnd<WT> *)nullptr, d_total);
        *seed_windows = seed;
        return 0;
    }
    // cuts among the samples: both fields in suffix order first, then the compaction reads them front to back
    pos_wnd<WT> *g = (pos_wnd<WT> *)buf_b;
    sx_launch(ctx, SX_KC_DOUBLING, M * (4 + 64 + sizeof(pos_wnd<WT>)), gather_pos_windows_kernel<WT>, dim3(sx_div_up(M, kBlock)), dim3(kBlock),
              sa_r, (const pos_wnd<WT> *)pw, M, (uint32_t *)nullptr, (WT *)nullptr, g, (uint32_t *)nullptr);
    WT *seed = (WT *)buf_a; // (the text-order pairs are done with)
    SX_TRY((device_compact(ctx, M, InHasLms<WT>{g}, OutPosWindow<WT>{g, sorted_lms, seed}, d_total, SX_KC_DOUBLING,
                           M * 2 * sizeof(pos_wnd<WT>))));
    *seed_windows = seed;
    return 0;
}

// buf_a, buf_b: M x 16 bytes each (M x 8 for texts of at most 16 symbols' windows, which are 32-bit words)
int sx_sorted_lms_windows(sx_ctx *ctx, const sx_text_info &ti, const uint32_t *sa_r, const uint32_t *pos, const uint8_t *is_lms, uint64_t M,
                          uint64_t m, void *buf_a, void *buf_b, uint32_t *sorted_lms, const void **seed_windows, uint32_t *d_total)
{
    wnd_cfg cfg;
    const bool wide = sx_window_cfg(ti.maxc, cfg);
    if (!wide) return sorted_lms_windows_typed<uint32_t>(ctx, ti, sa_r, pos, is_lms, M, m, buf_a, buf_b, sorted_lms, seed_windows, d_total, cfg);
    return sorted_lms_windows_typed<uint64_t>(ctx, ti, sa_r, pos, is_lms, M, m, buf_a, buf_b, sorted_lms, seed_windows, d_total, cfg);
}

int sx_name_pieces(sx_ctx *ctx, const uint64_t *ks, const uint32_t *vs, uint64_t M, sx_reduce_bufs &rb,
                   uint64_t *n_names)
{
    const bool two_pass = sx_scatter_permutation_applies(M) && M > sx_scatter_permutation_cursor_words() + 64;
    if (two_pass) {
        // R[vs[j]] = name of sorted piece j is one random 4-byte store a sample (3.1e8: 11.5 ms); the names in sorted
        // order first, then the two passes of sx_extras.hip (the sort buffer the pieces did not end in holds the pairs)
        SX_TRY((device_compact(ctx, M, InKeyBoundary{ks}, OutNamesInOrder{rb.rank}, rb.d_scalar, SX_KC_NAMES, M * (2 * 8 + 4))));
        uint32_t *cursor = rb.sub_t, *bad = rb.sub_t + sx_scatter_permutation_cursor_words();
        SX_CHECK(hipMemsetAsync(bad, 0, sizeof(uint32_t), ctx->stream));
        SX_TRY(sx_scatter_permutation(ctx, vs, rb.rank, M, rb.R, ks == rb.ka ? (void *)rb.kb : (void *)rb.ka, cursor, bad, SX_KC_NAMES));
    } else {
        SX_TRY((device_compact(ctx, M, InKeyBoundary{ks}, OutNames{vs, rb.R}, rb.d_scalar, SX_KC_NAMES,
                               M * (2 * 8 + 4 + 4))));
    }
    uint32_t boundaries = 0;
    SX_TRY(sx_readback(ctx, rb.d_scalar, 1, &boundaries));
    *n_names = (uint64_t)boundaries + 1;
    return 0;
}

int sx_reduced_suffix_sort(sx_ctx *ctx, uint64_t M, uint64_t n_names, sx_reduce_bufs &rb)
{
    const uint32_t b = (uint32_t)(sx_bitlen(n_names - 1) > 0 ? sx_bitlen(n_names - 1) : 1);
    uint32_t q = 64 / b;
    if (q < 1) q = 1;
    const uint32_t rbits = (uint32_t)(sx_bitlen(M - 1) > 0 ? sx_bitlen(M - 1) : 1);
    const dim3 block(kBlock);

    sx_launch(ctx, SX_KC_DOUBLING, M * (4 + 12), pack_names_kernel, dim3(sx_div_up(M, kBlock)), block,
              (const uint32_t *)rb.R, M, b, q, rb.ka, rb.va);
    int in_b = 0;
    SX_TRY(sx_sort_pairs(ctx, rb.ka, rb.va, rb.kb, rb.vb, M, 0, (int)(b * q), &in_b));
    uint64_t *ks = in_b ? rb.kb : rb.ka, *kfree = in_b ? rb.ka : rb.kb;
    uint32_t *vs = in_b ? rb.vb : rb.va, *vfree = in_b ? rb.va : rb.vb;

    uint32_t *pos = rb.pos_a, *pos2 = rb.pos_b;
    uint8_t *head = rb.head_a, *head2 = rb.head_b;
    uint32_t *n_heads = rb.head_bins; // groups of the list a round leaves (singletons included), in kHeadBins partial counts
    std::vector<uint32_t> h_bins(kHeadBins);
    SX_CHECK(hipMemsetAsync(n_heads, 0, kHeadBins * sizeof(uint32_t), ctx->stream));
    sx_launch(ctx, SX_KC_DOUBLING, M * (12 + 9), init_groups_kernel, dim3(sx_div_up(M, kBlock)), block,
              (const uint64_t *)ks, (const uint32_t *)vs, M, rb.sa_r, pos, head, n_heads);
    uint32_t *sa = vs; // active suffixes in sorted order
    uint64_t A = M;
    uint64_t h = q;
    bool waves_pay = true; // until a round leaves more than a quarter of its members to the radix sorts all the same
    // a. group numbers and ranks of all suffixes; the rounds keep both up to date (a rank is written again only where
    // the group's number changed).  The numbers travel with the list in one of two arrays, the other is a round's scratch
    // (the reduced string itself is not read again once its names are packed)
    uint32_t *gid = rb.gid, *gid2 = rb.R;
    if (sx_scatter_permutation_applies(M) && M > sx_scatter_permutation_cursor_words() + 64) {
        // (every suffix's first rank: 3.1e8 random 4-byte stores took 13.6 ms; the numbers in list order, then the two
        // passes of sx_extras.hip, the pairs in the free key buffer)
        SX_TRY((device_scan<OpMax>(ctx, A, InHeadPos{head, pos}, OutGidOnly{gid}, nullptr, SX_KC_DOUBLING, A * (1 + 4 + 4))));
        uint32_t *cursor = rb.sub_t, *bad = rb.sub_t + sx_scatter_permutation_cursor_words();
        SX_CHECK(hipMemsetAsync(bad, 0, sizeof(uint32_t), ctx->stream));
        SX_TRY(sx_scatter_permutation(ctx, sa, gid, M, rb.rank, kfree, cursor, bad, SX_KC_DOUBLING));
    } else {
        SX_TRY((device_scan<OpMax>(ctx, A, InHeadPos{head, pos}, OutGidRank{sa, gid, rb.rank}, nullptr, SX_KC_DOUBLING,
                                   A * (1 + 4 + 4 + 4 + 4))));
    }
    for (int round = 0; round < 64; ++round) {
        // b+d. drop singleton groups; build (group, rank[i+h]) keys for the rest
        SX_TRY((device_compact(ctx, A, InKeep{head, A},
                                   OutKeep{pos, sa, gid, rb.rank, pos2, vfree, kfree, h, M, rbits},
                                   rb.d_scalar, SX_KC_DOUBLING, A * (2 + 4 + 4 + 4 + 4 + 16))));
        uint32_t A2 = 0;
        SX_TRY(sx_readback(ctx, rb.d_scalar, 1, &A2));
        if (A2 == 0) return 0;
        if (h >= M) return sx_fail_msg(ctx, SX_E_INTERNAL, "doubling: span exceeds the string with active groups");
        ctx->stats.doubling_rounds++;
        // every dropped element was a group of its own: the kept list holds the other groups
        SX_TRY(sx_readback(ctx, n_heads, kHeadBins, h_bins.data()));
        uint64_t heads = 0;
        for (uint32_t i = 0; i < kHeadBins; ++i) heads += h_bins[i];
        const uint64_t dropped = A - A2, groups = heads > dropped ? heads - dropped : 1;
        // Small groups (a collection of near-identical sequences: as many members as copies) are ordered by one wave
        // each; the others, and all of them when the groups are long on average, by radix sorts of (group, rank).
        // (SX_FLAG_SORT_MODE 1: plain passes only.)
        const bool wave_tier = ctx->sort_mode != 1 && waves_pay && (uint64_t)A2 <= groups * kWaveTierMeanGroup;
        SX_CHECK(hipMemsetAsync(n_heads, 0, kHeadBins * sizeof(uint32_t), ctx->stream));
        if (wave_tier) {
            // e. (the old arrays are free now: sa receives the new order, the old head flags serve as done[])
            ctx->stats.refine_tiers |= 4u;
            uint8_t *done = head;
            SX_CHECK(hipMemsetAsync(done, 0, A2, ctx->stream));
            const uint64_t waves = sx_div_up(A2, kWaveStride);
            sx_launch(ctx, SX_KC_DOUBLING, (uint64_t)A2 * (8 + 4 + 4 + 4 + 4 + 2), doubling_wave_groups_kernel,
                      dim3(sx_div_up(waves, kWavesPerBlock)), block, (const uint64_t *)kfree, (const uint32_t *)vfree,
                      (const uint32_t *)pos2, (uint64_t)A2, rbits, sa, head2, rb.sa_r, done, n_heads, gid2, rb.rank);
            // the members of groups no wave owned: radix sorts of the compacted sub-list, then back to their slots
            uint64_t *k_in = ks;     // (the sorted keys are not needed once the groups are marked)
            uint32_t *v_in = gid;    // (free after the compaction above)
            SX_TRY((device_compact(ctx, A2, InNotDone{done}, OutNotDone{kfree, vfree, k_in, v_in, rb.sub_t},
                                   rb.d_scalar + 1, SX_KC_DOUBLING, (uint64_t)A2 * 2)));
            uint32_t A3 = 0;
            SX_TRY(sx_readback(ctx, rb.d_scalar + 1, 1, &A3));
            if ((uint64_t)A3 * 4 > (uint64_t)A2) waves_pay = false; // (a few long groups hold much of the list: words of a vocabulary)
            if (A3) {
                ctx->stats.refine_tiers |= 8u;
                uint64_t *k_other = kfree; // (copied into the sub-list: free)
                uint32_t *v_other = pos;   // (the old slots: free after the compaction above)
                SX_TRY(sx_sort_pairs(ctx, k_in, v_in, k_other, v_other, A3, 0, (int)(2 * rbits), &in_b));
                sx_launch(ctx, SX_KC_DOUBLING, (uint64_t)A3 * (8 + 4 + 4 + 4 + 4 + 1), regroup_sub_kernel,
                          dim3(sx_div_up(A3, kBlock)), block, (const uint64_t *)(in_b ? k_other : k_in),
                          (const uint32_t *)(in_b ? v_other : v_in), (const uint32_t *)rb.sub_t, (const uint32_t *)pos2,
                          (uint64_t)A3, sa, rb.sa_r, head2, n_heads);
                const uint64_t *k2s = in_b ? k_other : k_in;
                const uint32_t *sas = in_b ? v_other : v_in;
                SX_TRY((device_scan<OpMax>(ctx, A3, InSubHeadPos{k2s, rb.sub_t, pos2},
                                           OutSubGidRank{k2s, sas, rb.sub_t, gid2, rb.rank, rbits}, nullptr, SX_KC_DOUBLING,
                                           (uint64_t)A3 * (8 + 4 + 4 + 4 + 4 + 4))));
            }
        } else {
            // e. sort the active suffixes inside their groups by the rank h symbols ahead
            ctx->stats.refine_tiers |= 8u;
            uint64_t *k_in = kfree, *k_other = ks;
            uint32_t *v_in = vfree, *v_other = vs;
            SX_TRY(sx_sort_pairs(ctx, k_in, v_in, k_other, v_other, A2, 0, (int)(2 * rbits), &in_b));
            ks = in_b ? k_other : k_in;
            kfree = in_b ? k_in : k_other;
            vs = in_b ? v_other : v_in;
            vfree = in_b ? v_in : v_other;
            // f. new group boundaries, write the refined order back
            sx_launch(ctx, SX_KC_DOUBLING, (uint64_t)A2 * (8 + 4 + 4 + 4 + 1), regroup_kernel,
                      dim3(sx_div_up(A2, kBlock)), block, (const uint64_t *)ks, (const uint32_t *)vs,
                      (const uint32_t *)pos2, (uint64_t)A2, rb.sa_r, head2, n_heads);
            sa = vs;
            SX_TRY((device_scan<OpMax>(ctx, A2, InHeadPos{head2, pos2}, OutGidRankChanged{sa, gid2, rb.rank, ks, rbits},
                                       nullptr, SX_KC_DOUBLING, (uint64_t)A2 * (1 + 4 + 4 + 8 + 4 + 4))));
        }
        A = A2;
        h *= 2;
        uint32_t *tp = pos; pos = pos2; pos2 = tp;
        uint8_t *th = head; head = head2; head2 = th;
        uint32_t *tg = gid; gid = gid2; gid2 = tg;
    }
    return sx_fail_msg(ctx, SX_E_INTERNAL, "doubling: did not converge");
}

int sx_sorted_lms(sx_ctx *ctx, const uint32_t *sa_r, const uint32_t *pos, const uint8_t *is_lms, uint64_t M,
                  uint64_t m, uint32_t *sorted_lms, uint32_t *d_total)
{
    if (M == m) {
        // no cut points: every sample is an LMS position, one gather does it
        sx_launch(ctx, SX_KC_DOUBLING, M * 12, gather_pos_kernel, dim3(sx_div_up(M, kBlock)), dim3(kBlock), sa_r, pos,
                  M, sorted_lms, d_total);
        return 0;
    }
    return device_compact(ctx, M, InIsLms{sa_r, is_lms}, OutLmsPos{sa_r, pos, sorted_lms}, d_total,
                              SX_KC_DOUBLING, M * (2 * (4 + 1) + 8));
}
