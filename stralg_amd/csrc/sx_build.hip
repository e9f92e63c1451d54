// sx_build.hip -- suffix-array construction driver and its C-ABI entry points.
//
// Pipeline (each step cites the reference pass whose role it takes):
//   1 classify            sa_is.c:134-174   types, LMS flags, bucket sizes
//   2 samples + keys      sa_is.c:265-292   LMS-substring pieces as 64-bit keys
//   3 radix sort, names   sa_is.c:295-336   sorted LMS substrings -> reduced string
//   4 reduced suffix sort sa_is.c:370-387   (recursion) -> order of the LMS suffixes
//   5 sorted LMS          sa_is.c:443-464   remap_LMS
//   6 induce L, induce S  sa_is.c:220-263,397-398
// The output is the unique suffix array, hence bit-identical to
// sa_is_construction / sa_is_mem_construction / skew_sa_construction.
#include "sx_common.hpp"
#include "sx_device.hpp"
#include "sx_internal.hpp"
#include "sx_window.hpp"
#include "sx_pager.hpp"

#include <chrono>
#include <cmath>

namespace sx {

__global__ void write_u32_kernel(uint32_t *p, uint32_t v) { *p = v; }

// names of the reduced string as symbols of a byte text (at most 255 names besides the sentinel's 0)
__global__ __launch_bounds__(kBlock) void names_to_bytes_kernel(const uint32_t *__restrict__ R, uint64_t count, uint8_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < count) out[i] = (uint8_t)R[i];
}

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(kBlock) void synth_kernel(uint8_t *__restrict__ out, uint64_t n, uint32_t span,
                                                       uint64_t seed)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[i] = (uint8_t)(1u + (uint32_t)((splitmix64_at(seed, i) >> 33) % span));
}

} // namespace sx

using namespace sx;

// a sample's tied share under the longest prefix key from which both attempts of the prefix-key sort are skipped (the sort
// gives up at a quarter of its suffixes tied; the sample's share is a lower bound of the whole's)
static constexpr double kDoomedTiedShare = 0.30;

// bits per symbol, symbols per key, bits of the length field (tests/model.py key_layout)
static void key_layout(uint32_t maxc, uint32_t &bits, uint32_t &slots, uint32_t &lenbits)
{
    bits = (uint32_t)sx_bitlen(maxc);
    if (bits < 1) bits = 1;
    slots = 2;
    while ((slots + 1) * bits + (uint32_t)sx_bitlen(slots + 1) + 1 <= 64) ++slots;
    lenbits = (uint32_t)sx_bitlen(slots);
}

static size_t reduce_bytes(uint64_t M, uint64_t m)
{
    const size_t a = 256;
    size_t b = 0;
    b += M * 4 + a;       // pos
    b += M + a;           // is_lms
    b += 2 * (M * 8 + a); // keys
    b += 2 * (M * 4 + a); // vals
    b += 3 * (M * 4 + a); // R, rank, sa_r
    b += 2 * (M * 4 + a); // active positions
    b += 2 * (M * 4 + a); // gid, sub-list slots
    b += 2 * (M + a);     // heads
    b += m * 4 + a;       // sorted lms
    b += 4096 * 4 + a;    // partial head counts
    b += 4096;
    return b;
}

// bwt[i] = text[sa[i] - 1], 0 for the suffix that starts the text (bwt.c:13-20); only the direct sort needs
// this gather, the induction hands the BWT over with the suffix array
__global__ __launch_bounds__(kBlock) void bwt_of_sa_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ SA,
                                                           uint64_t N, uint8_t *__restrict__ bwt)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const uint32_t p = SA[i];
    bwt[i] = p ? T[p - 1u] : (uint8_t)0;
}

int sx_sa_build_impl(sx_ctx *ctx, const uint8_t *d_text, uint64_t n, uint32_t sigma, uint32_t *d_sa, uint8_t *d_bwt)
{
    const auto t0 = std::chrono::steady_clock::now();
    memset(&ctx->stats, 0, sizeof ctx->stats);
    ctx->stats.n = n;
    if (sigma < 1 || sigma > 256) return sx_fail_msg(ctx, SX_E_ARG, "alphabet_size must be in [1, 256]");
    if (n > 0xFFFFFFFEull) return sx_fail_msg(ctx, SX_E_ARG, "n must be at most 2^32 - 2");
    if (n > 0 && sigma < 2) return sx_fail_msg(ctx, SX_E_ARG, "non-empty text needs alphabet_size >= 2");
    if (n == 0) { // sa_is.c:413-417
        sx_launch(ctx, SX_KC_MISC, 0, write_u32_kernel, dim3(1), dim3(1), d_sa, 0u);
        if (d_bwt) SX_CHECK(hipMemsetAsync(d_bwt, 0, 1, ctx->stream));
        return sx_sync(ctx);
    }
    const uint64_t N = n + 1;
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_N, sx_text_scratch_bytes(n) + sx_induce_scratch_bytes(N, sigma)));
    sx_arena an;
    an.base = (char *)ctx->slab[SX_SLAB_N].p;
    an.cap = ctx->slab[SX_SLAB_N].cap;
    const uint64_t padded = (uint64_t)sx_div_up(N, kClsTile) * kClsTile + 128;
    uint8_t *T = an.take<uint8_t>(padded);
    if (!T) return sx_fail_msg(ctx, SX_E_INTERNAL, "arena: text");
    // The copy of the text: where the classification comes first (no direct sort of all suffixes to be tried) it makes
    // the copy itself while it reads the text -- all but the last tile or two, which are copied here --; only the tail
    // needs zeroing: sentinel + padding.
    // Short records of few symbols (round 4): classification + LMS sort + induced passes are 160 dependent launches whatever
    // the text's length -- 0.9 ms at 2^22 symbols, 0.44 ms at eleven --, the direct sort of all suffixes a third of that,
    // and on a short text its extra bytes cost less than the launches it saves.
    // (one MI355X, uniform symbols, suffix array + BWT, tools/small_direct.py: four letters 2^10 0.455 -> 0.144 ms, 2^22 0.79 -> 0.39,
    //  2^24 1.20 -> 0.78, 2^25 1.56 -> 1.47, 2^26 2.04 -> 2.75; seven letters 2^24 1.55 -> 0.74, 2^26 2.50 -> 2.46; fifteen 2^26
    //  4.76 -> 2.43)
    // The limit by the largest symbol, below where the two ways cross over (2^25, 2^26, beyond 2^27 for 4, 7, 11 - 15 letters:
    // 15 letters 2^27 6.9 -> 4.5 ms; at 2^30 the induction wins up to 16 symbols, see below).
    auto small_direct_limit = [&](uint32_t maxc) -> uint64_t {
        if (ctx->small_direct_max >= 0) return (uint64_t)ctx->small_direct_max;
        return (1ull << (maxc <= 4 ? 24 : (maxc <= 7 ? 25 : 27))) + 1ull;
    };
    // (before the symbols are counted: by the alphabet size the caller gave; checked again below with the largest symbol)
    const bool small_direct = sigma <= 16 && N <= small_direct_limit(sigma - 1) && N >= 4 && !ctx->no_direct && !ctx->force_general;
    uint32_t src_tiles = 0;
    if (!small_direct && !ctx->copy_text_first && sigma <= 16 && ((uintptr_t)d_text & 15u) == 0 && n >= (uint64_t)kClsTile + 64) src_tiles = (uint32_t)((n - 64) / kClsTile);
    const uint64_t copied_from = (uint64_t)src_tiles * kClsTile;
    SX_CHECK(hipMemcpyAsync(T + copied_from, d_text + copied_from, n - copied_from, hipMemcpyDeviceToDevice, ctx->stream));
    SX_CHECK(hipMemsetAsync(T + n, 0, padded - n, ctx->stream));

    // Wide alphabets: the induction visits the buckets one after the other, a few dependent launches per bucket
    // (sigma = 256: 1500 rounds, most of the build), while the first few symbols already tell nearly all suffixes
    // apart.  When the symbol statistics say that a 40-bit prefix key leaves only a few per cent of ALL suffixes
    // tied, the suffixes are sorted directly with the machinery of the LMS sort (radix sort by prefix key, tie
    // refinement); the result is the same suffix array.  Skewed or repetitive texts fail the tie bound inside and
    // continue on the usual path.  Only symbol counts are needed to decide, so this comes before the classification.
    if ((sigma > 16 || small_direct) && !ctx->no_direct && !ctx->force_general) {
        sx_text_info td;
        memset(&td, 0, sizeof td);
        td.T = T;
        td.n = n;
        td.N = N;
        sx_arena probe = an; // scratch of the probe is handed back whatever happens
        uint32_t *d_h = probe.take<uint32_t>(256);
        if (!d_h) return sx_fail_msg(ctx, SX_E_INTERNAL, "arena: histogram");
        SX_TRY(sx_symbol_histogram(ctx, T, n, d_h, td.h_all));
        td.h_all[0] += 1; // the sentinel
        if (td.h_all[0] != 1) return sx_fail_msg(ctx, SX_E_ARG, "text contains the sentinel symbol 0");
        double sum_p2 = 0.0;
        for (int c = 0; c < 256; ++c) {
            if (td.h_all[c]) td.maxc = (uint32_t)c;
            const double p = (double)td.h_all[c] / (double)N;
            sum_p2 += p * p;
        }
        if (td.maxc >= sigma) return sx_fail_msg(ctx, SX_E_ARG, "text holds a symbol >= alphabet_size");
        const double eff = 1.0 / sum_p2; // the alphabet size a uniform text with the same collision rate would have
        uint32_t need = 1;
        for (double v = eff; v < 16.0 * (double)N && need < 64; v *= eff) ++need; // <= ~6 % of the suffixes tied
        // (measured on uniform symbols, 1 GiB: the induction wins up to 16 symbols -- 4-bit window fields, 8-byte
        //  entries: 42 against 52 ms --, the direct sort from 20 symbols on: 51 ms flat against 53 ... 114 ms)
        if ((td.maxc >= 17 || (small_direct && td.maxc >= 2 && N <= small_direct_limit(td.maxc))) && (double)need * log2((double)td.maxc + 1.0) <= 40.0) {
            SX_TRY(sx_slab_ensure(ctx, SX_SLAB_M, sx_lms_prefix_bytes(N) + 1024));
            sx_arena am;
            am.base = (char *)ctx->slab[SX_SLAB_M].p;
            am.cap = ctx->slab[SX_SLAB_M].cap;
            const uint32_t *sorted = nullptr;
            const void *prev_symbols = nullptr; // one-symbol windows, when the keys had room for them
            int resolved = 0;
            double tied_share = -1.0; // (a look at a sample first: a word text ties most suffixes on any prefix)
            SX_TRY(sx_prefix_ties_sampled(ctx, td, am, true, &tied_share));
            ctx->stats.sample_tied_permille = tied_share < 0 ? 0u : (uint32_t)(tied_share * 1000.0) + 1u;
            if (tied_share < kDoomedTiedShare) SX_TRY(sx_sort_lms_by_prefix(ctx, td, am, &sorted, &prev_symbols, &resolved, true));
            if (resolved) {
                ctx->stats.lms_path = 3;
                ctx->stats.n_samples = N;
                SX_CHECK(hipMemcpyAsync(d_sa, sorted, N * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
                if (d_bwt && prev_symbols) // the sort carried every suffix's preceding symbol along
                    SX_TRY(sx_bwt_from_seed_windows(ctx, (const uint32_t *)prev_symbols, N, td.maxc, d_bwt));
                else if (d_bwt)
                    sx_launch(ctx, SX_KC_BWT_GATHER, N * 6, bwt_of_sa_kernel, dim3(sx_div_up(N, kBlock)), dim3(kBlock),
                              (const uint8_t *)T, (const uint32_t *)d_sa, N, d_bwt);
                SX_TRY(sx_sync(ctx));
                ctx->stats.ms_total =
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                return 0;
            }
            ctx->stats.doubling_rounds = 0;
            ctx->stats.sort_passes = 0;
        }
    }

    sx_text_info ti;
    SX_TRY(sx_classify(ctx, T, n, an, ti, src_tiles ? d_text : nullptr, src_tiles));
    if (ti.maxc >= sigma) return sx_fail_msg(ctx, SX_E_ARG, "text holds a symbol >= alphabet_size");
    ctx->stats.n_lms = ti.m;

    const uint32_t *sorted_lms = nullptr;
    const void *seed_windows = nullptr; // windows of the sorted LMS suffixes, when the sort carried them
    bool seed_windows_u32 = false;      // ... as 32-bit words (the prefix-key sort's), whatever the text's window width
    if (ti.m <= 1) {
        // only the sentinel is LMS: it alone seeds the induction
        uint32_t *one = an.take<uint32_t>(1);
        if (!one) return sx_fail_msg(ctx, SX_E_INTERNAL, "arena");
        sx_launch(ctx, SX_KC_MISC, 0, write_u32_kernel, dim3(1), dim3(1), one, (uint32_t)n);
        sorted_lms = one;
        ctx->stats.n_samples = 1;
        ctx->stats.n_names = 1;
    }
    if (ti.m > 1 && !ctx->force_general) {
        // fast path: radix sort of the LMS suffixes by their first 64/b symbols (sx_lmssort.hip)
        SX_TRY(sx_slab_ensure(ctx, SX_SLAB_M, sx_lms_prefix_bytes(ti.m) + 1024));
        sx_arena am;
        am.base = (char *)ctx->slab[SX_SLAB_M].p;
        am.cap = ctx->slab[SX_SLAB_M].cap;
        int resolved = 0;
        double tied_share = -1.0;
        SX_TRY(sx_prefix_ties_sampled(ctx, ti, am, false, &tied_share));
        if (tied_share >= 0) ctx->stats.sample_tied_permille = (uint32_t)(tied_share * 1000.0) + 1u;
        if (tied_share < kDoomedTiedShare) SX_TRY(sx_sort_lms_by_prefix(ctx, ti, am, &sorted_lms, &seed_windows, &resolved));
        if (resolved) {
            ctx->stats.lms_path = 1;
            ctx->stats.n_samples = ti.m;
            seed_windows_u32 = seed_windows != nullptr;
        } else {
            sorted_lms = nullptr;
            seed_windows = nullptr;
        }
    }
    if (ti.m > 1 && !sorted_lms) {
        ctx->stats.lms_path = 2;
        ctx->stats.doubling_rounds = 0;
        uint32_t bits, slots, lenbits;
        key_layout(ti.maxc, bits, slots, lenbits);
        ctx->stats.key_bits = bits;
        ctx->stats.key_slots = slots;
        SX_TRY(sx_sample_flags(ctx, ti, slots - 1));
        const uint64_t M = ti.M;
        ctx->stats.n_samples = M;
        SX_TRY(sx_slab_ensure(ctx, SX_SLAB_M, reduce_bytes(M, ti.m)));
        sx_arena am;
        am.base = (char *)ctx->slab[SX_SLAB_M].p;
        am.cap = ctx->slab[SX_SLAB_M].cap;
        uint32_t *pos = am.take<uint32_t>(M);
        uint8_t *is_lms = am.take<uint8_t>(M);
        sx_reduce_bufs rb;
        rb.ka = am.take<uint64_t>(M);
        rb.kb = am.take<uint64_t>(M);
        rb.va = am.take<uint32_t>(M);
        rb.vb = am.take<uint32_t>(M);
        rb.R = am.take<uint32_t>(M);
        rb.rank = am.take<uint32_t>(M);
        rb.sa_r = am.take<uint32_t>(M);
        rb.pos_a = am.take<uint32_t>(M);
        rb.pos_b = am.take<uint32_t>(M);
        rb.gid = am.take<uint32_t>(M);
        rb.sub_t = am.take<uint32_t>(M);
        rb.head_a = am.take<uint8_t>(M);
        rb.head_b = am.take<uint8_t>(M);
        uint32_t *slms = am.take<uint32_t>(ti.m);
        rb.d_scalar = am.take<uint32_t>(16);
        rb.head_bins = am.take<uint32_t>(4096);
        if (!rb.head_bins) return sx_fail_msg(ctx, SX_E_INTERNAL, "arena: reduce buffers");
        if (!pos || !is_lms || !rb.ka || !rb.kb || !rb.va || !rb.vb || !rb.R || !rb.rank || !rb.sa_r || !rb.pos_a ||
            !rb.pos_b || !rb.gid || !rb.sub_t || !rb.head_a || !rb.head_b || !slms || !rb.d_scalar)
            return sx_fail_msg(ctx, SX_E_INTERNAL, "arena: reduce buffers");

        SX_TRY(sx_sample_write(ctx, ti, pos, is_lms));
        SX_TRY(sx_piece_keys(ctx, ti, pos, is_lms, bits, slots, lenbits, rb.ka, rb.va));
        const int used = (int)(slots * bits + lenbits + 1);
        int in_b = 0;
        SX_TRY(sx_sort_pairs(ctx, rb.ka, rb.va, rb.kb, rb.vb, M, 64 - used, 64, &in_b));
        const uint64_t *ks = in_b ? rb.kb : rb.ka;
        const uint32_t *vs = in_b ? rb.vb : rb.va;
        uint64_t n_names = 0;
        SX_TRY(sx_name_pieces(ctx, ks, vs, M, rb, &n_names));
        ctx->stats.n_names = n_names;
        const uint32_t *sa_r;
        const uint64_t recurse_min = ctx->recurse_min >= 0 ? (uint64_t)ctx->recurse_min : (1ull << 20);
        if (n_names == M) {
            sa_r = vs; // every piece is unique: sorted pieces == sorted suffixes (sa_is.c:423-428)
        } else if (n_names <= 256 && M - 1 >= recurse_min && ctx->depth < 48) {
            // The recursion of sa_is.c:370-387, where it costs nothing new: a reduced string of at most 255 names (and
            // the sentinel's name 0) *is* a remapped byte text, and this pipeline sorts those -- in a child context with
            // its own workspace, the general path forced (a text that got here has too many ties for a prefix sort).
            // These are the texts prefix doubling is worst on -- periodic and Fibonacci strings, a handful of distinct
            // LMS substrings however long the text: log n rounds over all samples -- and every level is a third to a half
            // as long as the one above, so the levels together cost what the first costs twice over.
            uint8_t *rbytes = reinterpret_cast<uint8_t *>(rb.kb); // (the key buffers are free once the pieces have their names)
            sx_launch(ctx, SX_KC_NAMES, M * 5, names_to_bytes_kernel, dim3(sx_div_up(M, kBlock)), dim3(kBlock), (const uint32_t *)rb.R,
                      M - 1, rbytes);
            SX_TRY(sx_sync(ctx));
            sx_ctx *child = nullptr;
            SX_TRY(sx_child_begin(ctx, &child));
            const int rc = sx_sa_build_impl(child, rbytes, M - 1, (uint32_t)n_names, rb.sa_r, nullptr);
            sx_child_end(ctx, child);
            if (rc != 0) return sx_fail_msg(ctx, rc, sx_last_error(child));
            ctx->stats.recursion_levels = 1 + child->stats.recursion_levels;
            ctx->stats.doubling_rounds += child->stats.doubling_rounds;
            ctx->stats.sort_passes += child->stats.sort_passes;
            sa_r = rb.sa_r;
        } else {
            SX_TRY(sx_reduced_suffix_sort(ctx, M, n_names, rb));
            sa_r = rb.sa_r;
        }
        {
            // positions and windows of the sorted LMS suffixes from one gather; the scratch: the key buffers (32-bit
            // windows), or the key buffers as one and the four list arrays as the other (64-bit windows: 16 bytes a sample)
            wnd_cfg wc;
            const bool wide = sx_window_cfg(ti.maxc, wc);
            void *buf_a = rb.ka, *buf_b = wide ? (void *)rb.pos_a : (void *)rb.kb;
            const bool room = !wide || ((char *)rb.kb + M * 8 >= (char *)rb.ka + M * 16 && (char *)rb.sub_t + M * 4 >= (char *)rb.pos_a + M * 16 &&
                                        (char *)rb.kb > (char *)rb.ka && (char *)rb.sub_t > (char *)rb.pos_a);
            if (room)
                SX_TRY(sx_sorted_lms_windows(ctx, ti, sa_r, pos, is_lms, M, ti.m, buf_a, buf_b, slms, &seed_windows, rb.d_scalar));
            else
                SX_TRY(sx_sorted_lms(ctx, sa_r, pos, is_lms, M, ti.m, slms, rb.d_scalar));
        }
        uint32_t got = 0;
        SX_TRY(sx_readback(ctx, rb.d_scalar, 1, &got));
        if (got != ti.m) return sx_fail_msg(ctx, SX_E_INTERNAL, "sorted LMS count differs from the LMS count");
        sorted_lms = slms;
    }

    SX_TRY(sx_induce(ctx, ti, sigma, sorted_lms, seed_windows, seed_windows_u32, d_sa, d_bwt, an));
    SX_TRY(sx_sync(ctx));
    ctx->stats.ms_total =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

extern "C" {

int sx_sa_build_dev(sx_ctx *ctx, const uint8_t *d_text, uint64_t n, uint32_t alphabet_size, uint32_t *d_sa_out)
{
    if (!ctx || !d_sa_out || (n && !d_text)) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    return sx_sa_build_impl(ctx, d_text, n, alphabet_size, d_sa_out, nullptr);
}

int sx_sa_bwt_build_dev(sx_ctx *ctx, const uint8_t *d_text, uint64_t n, uint32_t alphabet_size, uint32_t *d_sa_out,
                        uint8_t *d_bwt_out)
{
    if (!ctx || !d_sa_out || (n && !d_text)) return SX_E_ARG;
    SX_CHECK(hipSetDevice(ctx->device));
    return sx_sa_build_impl(ctx, d_text, n, alphabet_size, d_sa_out, d_bwt_out);
}

int sx_sa_build(sx_ctx *ctx, const uint8_t *text, uint64_t n, uint32_t alphabet_size, uint32_t *sa_out)
{
    if (!ctx || !sa_out || (n && !text)) return SX_E_ARG;
    if (n > 0xFFFFFFFEull) return sx_fail_msg(ctx, SX_E_ARG, "n must be at most 2^32 - 2");
    SX_CHECK(hipSetDevice(ctx->device));
    const uint64_t N = n + 1;
    const size_t text_bytes = (n + 255) & ~(size_t)255;
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_IO, text_bytes + N * sizeof(uint32_t) + 512));
    uint8_t *d_text = (uint8_t *)ctx->slab[SX_SLAB_IO].p;
    uint32_t *d_sa = (uint32_t *)((char *)ctx->slab[SX_SLAB_IO].p + text_bytes + 256);
    // the caller's array is paged in by host threads while the GPU builds (sx_pager.hpp)
    sx_host_pager pager;
    const size_t c_sa = pager.add(sa_out, N * sizeof(uint32_t));
    pager.start();
    if (n) SX_CHECK(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, ctx->stream));
    SX_TRY(sx_sa_build_impl(ctx, d_text, n, alphabet_size, d_sa, nullptr));
    SX_TRY(pager.download(ctx, c_sa, sa_out, d_sa, N * sizeof(uint32_t)));
    return sx_sync(ctx);
}

int sx_build_tables(sx_ctx *ctx, const uint8_t *text, uint64_t n, uint32_t sigma, uint32_t *sa_out, uint32_t *c_out,
                    uint32_t *o_out)
{
    if (!ctx || (n && !text) || !c_out) return SX_E_ARG;
    if (n > 0xFFFFFFFEull) return sx_fail_msg(ctx, SX_E_ARG, "n must be at most 2^32 - 2");
    if (sigma < 1 || sigma > 256) return sx_fail_msg(ctx, SX_E_ARG, "sigma must be in [1, 256]");
    SX_CHECK(hipSetDevice(ctx->device));
    const uint64_t N = n + 1;
    const size_t text_b = (n + 255) & ~(size_t)255, sa_b = (N * 4 + 255) & ~(size_t)255, bwt_b = (N + 255) & ~(size_t)255;
    const size_t o_b = o_out ? (((N + 1) * (size_t)sigma * 4 + 255) & ~(size_t)255) : 0;
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_IO, text_b + sa_b + bwt_b + o_b + 4096));
    char *base = (char *)ctx->slab[SX_SLAB_IO].p;
    uint8_t *d_text = (uint8_t *)base;
    uint32_t *d_sa = (uint32_t *)(base + text_b + 256);
    uint8_t *d_bwt = (uint8_t *)(base + text_b + 256 + sa_b);
    uint32_t *d_c = (uint32_t *)(base + text_b + 256 + sa_b + bwt_b);
    uint32_t *d_o = o_out ? (uint32_t *)(base + text_b + 256 + sa_b + bwt_b + 1024) : nullptr;
    // the caller's arrays (4 + 4 sigma bytes per suffix) are paged in by host threads while the GPU builds, in the
    // order the copies want them (sx_pager.hpp)
    sx_host_pager pager;
    const size_t o_bytes = (N + 1) * (size_t)sigma * 4;
    const size_t c_sa = sa_out ? pager.add(sa_out, N * sizeof(uint32_t)) : 0;
    const size_t c_o = o_out ? pager.add(o_out, o_bytes) : 0;
    pager.start();
    if (n) SX_CHECK(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, ctx->stream));
    SX_TRY(sx_sa_build_impl(ctx, d_text, n, sigma, d_sa, d_bwt));
    SX_TRY(sx_tables_from_bwt_impl(ctx, d_bwt, N, sigma, d_c, d_o));
    if (sa_out) SX_TRY(pager.download(ctx, c_sa, sa_out, d_sa, N * sizeof(uint32_t)));
    SX_CHECK(hipMemcpyAsync(c_out, d_c, (size_t)sigma * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (o_out) SX_TRY(pager.download(ctx, c_o, o_out, d_o, o_bytes));
    return sx_sync(ctx);
}

// device -> sink in chunks through two pinned staging buffers: the copy of chunk k+1 runs while the sink
// (typically fwrite) consumes chunk k; nothing of the array's size exists on the host
static int stream_out(sx_ctx *ctx, int section, const void *d_src, size_t bytes, sx_sink_fn sink, void *user)
{
    constexpr size_t kChunk = (size_t)32 << 20;
    if (!ctx->h_stage[0]) {
        for (int b = 0; b < 2; ++b)
            if (hipHostMalloc((void **)&ctx->h_stage[b], kChunk, hipHostMallocDefault) != hipSuccess)
                return sx_fail_msg(ctx, SX_E_NOMEM, "pinned staging buffers");
    }
    const char *src = (const char *)d_src;
    size_t off = 0, pending = 0;
    int cur = 0;
    if (bytes) {
        pending = bytes < kChunk ? bytes : kChunk;
        SX_CHECK(hipMemcpyAsync(ctx->h_stage[0], src, pending, hipMemcpyDeviceToHost, ctx->stream));
    }
    while (pending) {
        SX_CHECK(hipStreamSynchronize(ctx->stream));
        const size_t have = pending;
        off += have;
        const size_t next = bytes - off < kChunk ? bytes - off : kChunk;
        if (next) SX_CHECK(hipMemcpyAsync(ctx->h_stage[cur ^ 1], src + off, next, hipMemcpyDeviceToHost, ctx->stream));
        if (sink(user, section, ctx->h_stage[cur], have) != 0) {
            (void)hipStreamSynchronize(ctx->stream);
            return sx_fail_msg(ctx, SX_E_ARG, "the sink refused a chunk");
        }
        pending = next;
        cur ^= 1;
    }
    return 0;
}

int sx_build_tables_stream(sx_ctx *ctx, const uint8_t *text, uint64_t n, uint32_t sigma, int want_sa, sx_sink_fn sink,
                           void *user)
{
    if (!ctx || (n && !text) || !sink) return SX_E_ARG;
    if (n > 0xFFFFFFFEull) return sx_fail_msg(ctx, SX_E_ARG, "n must be at most 2^32 - 2");
    if (sigma < 1 || sigma > 128) return sx_fail_msg(ctx, SX_E_ARG, "sigma must be in [1, 128] for the O table");
    SX_CHECK(hipSetDevice(ctx->device));
    const uint64_t N = n + 1;
    const size_t text_b = (n + 255) & ~(size_t)255, sa_b = (N * 4 + 255) & ~(size_t)255, bwt_b = (N + 255) & ~(size_t)255;
    const size_t o_bytes = (N + 1) * (size_t)sigma * 4;
    SX_TRY(sx_slab_ensure(ctx, SX_SLAB_IO, text_b + sa_b + bwt_b + ((o_bytes + 255) & ~(size_t)255) + 4096));
    char *base = (char *)ctx->slab[SX_SLAB_IO].p;
    uint8_t *d_text = (uint8_t *)base;
    uint32_t *d_sa = (uint32_t *)(base + text_b + 256);
    uint8_t *d_bwt = (uint8_t *)(base + text_b + 256 + sa_b);
    uint32_t *d_c = (uint32_t *)(base + text_b + 256 + sa_b + bwt_b);
    uint32_t *d_o = (uint32_t *)(base + text_b + 256 + sa_b + bwt_b + 1024);
    if (n) SX_CHECK(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, ctx->stream));
    SX_TRY(sx_sa_build_impl(ctx, d_text, n, sigma, d_sa, d_bwt));
    SX_TRY(sx_tables_from_bwt_impl(ctx, d_bwt, N, sigma, d_c, d_o));
    if (want_sa) SX_TRY(stream_out(ctx, SX_SECTION_SA, d_sa, N * sizeof(uint32_t), sink, user));
    SX_TRY(stream_out(ctx, SX_SECTION_C, d_c, (size_t)sigma * 4, sink, user));
    SX_TRY(stream_out(ctx, SX_SECTION_O, d_o, o_bytes, sink, user));
    return sx_sync(ctx);
}

int sx_synth_dev(sx_ctx *ctx, uint8_t *d_out, uint64_t n, uint32_t sigma, uint64_t seed)
{
    if (!ctx || sigma < 2 || sigma > 256) return SX_E_ARG;
    if (n == 0) return 0;
    SX_CHECK(hipSetDevice(ctx->device));
    uint32_t grid = sx_div_up(n, kBlock);
    if (grid > 8192) grid = 8192;
    sx_launch(ctx, SX_KC_MISC, n, synth_kernel, dim3(grid), dim3(kBlock), d_out, n, sigma - 1, seed);
    return sx_sync(ctx);
}

} // extern "C"
