// sx_common.hpp -- context, workspace slabs, launch + profiling helpers.
//
// One sx_ctx per (host thread, GPU): it owns a HIP stream, grow-only device
// slabs and a pinned read-back page.  Nothing here is process-global, so N
// host threads can drive N GPUs concurrently (threading).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/stralg_amd.h"

#define SX_CHECK(expr)                                                         \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) return sx_fail(ctx, (int)e_, #expr, __FILE__, __LINE__); \
    } while (0)

#define SX_TRY(expr)                                                           \
    do {                                                                       \
        int rc_ = (expr);                                                      \
        if (rc_ != 0) return rc_;                                              \
    } while (0)

struct sx_slab {
    void *p = nullptr;
    size_t cap = 0;
};

// bump allocator over one slab; offsets are 256-byte aligned
struct sx_arena {
    char *base = nullptr;
    size_t cap = 0, off = 0;
    template <class T> T *take(size_t count)
    {
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (off + bytes > cap) return nullptr;
        T *r = reinterpret_cast<T *>(base + off);
        off += bytes;
        return r;
    }
};

struct sx_event_pair {
    hipEvent_t a, b;
    int kclass;
};

// grow-only device slabs of a context
enum {
    SX_SLAB_N = 0,     // text copy, bit arrays, windows, induce control: proportional to n
    SX_SLAB_M = 1,     // LMS-suffix sort / reduced problem: proportional to the LMS count
    SX_SLAB_BWT = 2,   // bwt bytes and tile counts of the table build
    SX_SLAB_SCAN = 3,  // tile totals of the device scan in flight
    SX_SLAB_SORT = 4,  // radix tile histograms
    SX_SLAB_IO = 5,    // staging of the host-buffer entry points
    SX_SLAB_CHAIN = 6, // look-back status words
    SX_NSLABS = 7
};

struct sx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    sx_slab slab[SX_NSLABS];
    uint32_t *h_pin = nullptr; // pinned read-back page (4 KiB) + word 1024: the sequence number of the last read-back
    uint32_t readback_seq = 0;
    char *h_stage[2] = {nullptr, nullptr}; // pinned staging of the streaming downloads (allocated on first use)
    // profiling
    uint32_t chain_epoch = 0; // look-back status epoch (24 bits), see sx_device.hpp
    int64_t chain_max_override = -1; // SX_FLAG_CHAIN_MAX_ENTRIES; -1 = choose by alphabet size
    int prefix_symbols = 0;          // SX_FLAG_PREFIX_SYMBOLS; 0 = choose by the number of suffixes
    int force_general = 0; // SX_FLAG_FORCE_GENERAL_PATH
    int no_direct = 0;     // SX_FLAG_NO_DIRECT_SORT
    int radix_digit_bits = 0; // SX_FLAG_RADIX_DIGIT_BITS; 0 = 8
    int sort_mode = 0;        // SX_FLAG_SORT_MODE
    int induce_batch_off = 0; // SX_FLAG_INDUCE_BATCH_OFF
    int induce_attended = 0;  // SX_FLAG_INDUCE_ATTENDED
    int induce_no_hoist = 0;  // SX_FLAG_INDUCE_NO_HOIST
    int text_keys_off = 0;    // SX_FLAG_TEXT_KEYS_OFF
    int long_subbuckets_off = 0; // SX_FLAG_LONG_SUBBUCKETS_OFF
    int local_sort_lean_off = 0; // SX_FLAG_LOCAL_SORT_LEAN_OFF
    int64_t small_direct_max = -1; // SX_FLAG_SMALL_DIRECT_MAX; -1 = the default
    int copy_text_first = 0;  // SX_FLAG_COPY_TEXT_FIRST
    int64_t sample_min = -1;  // SX_FLAG_SAMPLE_MIN; -1 = texts of 2^20 suffixes and more get the look at a sample
    int64_t recurse_min = -1; // SX_FLAG_RECURSE_MIN; -1 = the default length from which a reduced string of <= 255 names recurses
    sx_ctx *child = nullptr;  // the context a reduced string over a byte alphabet is sorted in (sx_build.hip), created on first use
    int depth = 0;            // 0 for a caller's context, 1 + the parent's for a child
    int64_t induce_batch_min = -1; // SX_FLAG_INDUCE_BATCH_MIN; -1 = ranges the tail kernel holds pass the batch form by
    int prof_on = 0;
    int prof_only = -1; // >= 0: only launches of this kernel class are bracketed with events
    // first launch that the runtime refused (a bad grid, ...): reported by the next sx_sync / sx_readback
    hipError_t launch_err = hipSuccess;
    int launch_err_class = 0;
    std::vector<sx_event_pair> ev_used;
    std::vector<sx_event_pair> ev_free;
    sx_kernel_stat kstat[SX_KC_COUNT];
    sx_build_stats stats;
};

int sx_fail(sx_ctx *ctx, int code, const char *what, const char *file, int line);
int sx_fail_msg(sx_ctx *ctx, int code, const char *msg);
int sx_slab_ensure(sx_ctx *ctx, int which, size_t bytes);
int sx_sync(sx_ctx *ctx);
// device -> pinned host copy of `count` u32 followed by a stream sync
int sx_readback(sx_ctx *ctx, const uint32_t *d_src, size_t count, uint32_t *h_dst);
// the same for one to four arrays (1024 words together), one behind the other in h_dst
int sx_readback_ranges(sx_ctx *ctx, const uint32_t *const *d_src, const uint32_t *counts, int ranges, uint32_t *h_dst);
// look-back status memory: a slab of (epoch-tagged) status words, zeroed whenever it is (re)allocated
int sx_chain_slab(sx_ctx *ctx, int which, size_t bytes);
// a fresh epoch for one chained launch (24 bits; every status slab is zeroed when they wrap)
uint32_t sx_chain_next_epoch(sx_ctx *ctx);

// the child context of ctx (created on first use; it takes over the behaviour switches and the profiling state), and the
// end of a build in it: its event times and launch counts are added to the parent's
int sx_child_begin(sx_ctx *ctx, sx_ctx **child);
void sx_child_end(sx_ctx *ctx, sx_ctx *child);

// a pair of events for one timed launch of class kclass (nullptr: none to be had), and its entry in the class's table
sx_event_pair *sx_prof_pair(sx_ctx *ctx, int kclass);
void sx_prof_count(sx_ctx *ctx, int kclass, uint64_t alg_bytes);

// Launch, optionally timed by HIP events on the context's stream: the two events ride on the dispatch itself
// (hipExtLaunchKernelGGL: they take the kernel's own start and end times).  Events recorded in front of and behind the
// launch (hipEventRecord, rounds 1 - 5) are packets of their own: 5.6 us of idle queue each, 0.39 ms of a 20 ms step for
// the 35 launches of the one class the bench times inside its timed region.
// alg_bytes = algorithmic bytes moved by this launch (per kernel).
template <class... P, class... A>
static inline void sx_launch(sx_ctx *ctx, int kclass, uint64_t alg_bytes, void (*kernel)(P...),
                             dim3 grid, dim3 block, A... args)
{
    const bool timed = ctx->prof_on && (ctx->prof_only < 0 || ctx->prof_only == kclass);
    sx_event_pair *ep = timed ? sx_prof_pair(ctx, kclass) : nullptr;
    if (ep) hipExtLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, ep->a, ep->b, 0u, static_cast<P>(args)...);
    else hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, args...);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && ctx->launch_err == hipSuccess) ctx->launch_err = e, ctx->launch_err_class = kclass;
    if (ep) sx_prof_count(ctx, kclass, alg_bytes);
}

static inline uint32_t sx_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
static inline int sx_bitlen(uint64_t v)
{
    int b = 0;
    while (v) {
        ++b;
        v >>= 1;
    }
    return b;
}
