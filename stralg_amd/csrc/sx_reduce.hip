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
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_scan.hpp"
#include "sx_internal.hpp"

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

__global__ __launch_bounds__(kBlock) void init_groups_kernel(const uint64_t *__restrict__ ks,
                                                             const uint32_t *__restrict__ vs, uint64_t M,
                                                             uint32_t *__restrict__ sa_r, uint32_t *__restrict__ pos,
                                                             uint8_t *__restrict__ head)
{
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= M) return;
    sa_r[j] = vs[j];
    pos[j] = (uint32_t)j;
    head[j] = (j == 0 || ks[j] != ks[j - 1]) ? 1 : 0;
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
                                                         uint32_t *__restrict__ sa_r, uint8_t *__restrict__ head2)
{
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= A) return;
    head2[t] = (t == 0 || k2s[t] != k2s[t - 1]) ? 1 : 0;
    sa_r[pos2[t]] = sas[t];
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

} // namespace sx

using namespace sx;

int sx_name_pieces(sx_ctx *ctx, const uint64_t *ks, const uint32_t *vs, uint64_t M, sx_reduce_bufs &rb,
                   uint64_t *n_names)
{
    SX_TRY((device_compact(ctx, M, InKeyBoundary{ks}, OutNames{vs, rb.R}, rb.d_scalar, SX_KC_NAMES,
                               M * (2 * 8 + 4 + 4))));
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
    sx_launch(ctx, SX_KC_DOUBLING, M * (12 + 9), init_groups_kernel, dim3(sx_div_up(M, kBlock)), block,
              (const uint64_t *)ks, (const uint32_t *)vs, M, rb.sa_r, pos, head);
    uint32_t *sa = vs; // active suffixes in sorted order
    uint64_t A = M;
    uint64_t h = q;
    for (int round = 0; round < 64; ++round) {
        // a. group ids and ranks of the active suffixes
        SX_TRY((device_scan<OpMax>(ctx, A, InHeadPos{head, pos}, OutGidRank{sa, rb.gid, rb.rank}, nullptr,
                                   SX_KC_DOUBLING, A * (1 + 4 + 4 + 4 + 4))));
        // b+d. drop singleton groups; build (group, rank[i+h]) keys for the rest
        SX_TRY((device_compact(ctx, A, InKeep{head, A},
                                   OutKeep{pos, sa, rb.gid, rb.rank, pos2, vfree, kfree, h, M, rbits},
                                   rb.d_scalar, SX_KC_DOUBLING, A * (2 + 4 + 4 + 4 + 4 + 16))));
        uint32_t A2 = 0;
        SX_TRY(sx_readback(ctx, rb.d_scalar, 1, &A2));
        if (A2 == 0) return 0;
        if (h >= M) return sx_fail_msg(ctx, SX_E_INTERNAL, "doubling: span exceeds the string with active groups");
        ctx->stats.doubling_rounds++;
        // e. sort the active suffixes inside their groups by the rank h symbols ahead
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
                  (const uint32_t *)pos2, (uint64_t)A2, rb.sa_r, head2);
        sa = vs;
        A = A2;
        h *= 2;
        uint32_t *tp = pos; pos = pos2; pos2 = tp;
        uint8_t *th = head; head = head2; head2 = th;
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
