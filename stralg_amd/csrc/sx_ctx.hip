// sx_ctx.hip -- context lifetime, workspace slabs, HIP-event profiler.
#include <atomic>
#include <chrono>
#include "sx_common.hpp"
#include "sx_scan.hpp"

#include <new>

static const char *kClassNames[SX_KC_COUNT] = {
    "classify", "samples", "keys", "radix_hist", "radix_scatter", "scan", "names",
    "doubling", "induce_gather", "induce_scan", "induce_scatter", "induce_chain", "bwt_gather", "otable", "misc",
    "fasta", "remap", "lcp", "search", "local_sort",
};

int sx_fail(sx_ctx *ctx, int code, const char *what, const char *file, int line)
{
    if (ctx) snprintf(ctx->err, sizeof ctx->err, "%s failed: %s (%d) at %s:%d", what,
                      hipGetErrorString((hipError_t)code), code, file, line);
    return code ? code : SX_E_INTERNAL;
}

int sx_fail_msg(sx_ctx *ctx, int code, const char *msg)
{
    if (ctx) snprintf(ctx->err, sizeof ctx->err, "%s", msg);
    return code;
}

int sx_slab_ensure(sx_ctx *ctx, int which, size_t bytes)
{
    sx_slab &s = ctx->slab[which];
    if (s.cap >= bytes) return 0;
    if (s.p) {
        SX_CHECK(hipStreamSynchronize(ctx->stream));
        SX_CHECK(hipFree(s.p));
        s.p = nullptr;
        s.cap = 0;
    }
    // round up so that repeated calls with slowly growing inputs do not thrash
    size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    SX_CHECK(hipMalloc(&s.p, want));
    s.cap = want;
    return 0;
}

uint32_t *sx::sx_scan_scratch(sx_ctx *ctx, uint32_t ntiles)
{
    // tile totals of the scan in flight; a slab of its own so that growing it
    // never moves a caller's data
    const size_t need = (size_t)ntiles * sizeof(uint32_t);
    if (ctx->slab[SX_SLAB_SCAN].cap < need) {
        if (sx_slab_ensure(ctx, SX_SLAB_SCAN, need) != 0) return nullptr;
    }
    return (uint32_t *)ctx->slab[SX_SLAB_SCAN].p;
}

int sx_chain_slab(sx_ctx *ctx, int which, size_t bytes)
{
    const size_t before = ctx->slab[which].cap;
    SX_TRY(sx_slab_ensure(ctx, which, bytes));
    if (ctx->slab[which].cap != before) // new memory holds arbitrary bits: epoch 0 never matches a launch
        SX_CHECK(hipMemsetAsync(ctx->slab[which].p, 0, ctx->slab[which].cap, ctx->stream));
    return 0;
}

uint32_t sx_chain_next_epoch(sx_ctx *ctx)
{
    if (ctx->chain_epoch + 1 >= (1u << 24)) { // about to wrap: retire every old status word
        if (ctx->slab[SX_SLAB_CHAIN].p)
            (void)hipMemsetAsync(ctx->slab[SX_SLAB_CHAIN].p, 0, ctx->slab[SX_SLAB_CHAIN].cap, ctx->stream);
        ctx->chain_epoch = 0;
    }
    return ++ctx->chain_epoch;
}

static int launch_error(sx_ctx *ctx)
{
    const hipError_t e = ctx->launch_err;
    ctx->launch_err = hipSuccess;
    snprintf(ctx->err, sizeof ctx->err, "a %s kernel launch failed: %s", sx_kernel_class_name(ctx->launch_err_class),
             hipGetErrorString(e));
    return (int)e;
}

int sx_sync(sx_ctx *ctx)
{
    SX_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->launch_err != hipSuccess) return launch_error(ctx);
    return 0;
}

// The values go to the pinned page by a one-workgroup kernel that then stores a sequence number behind them (system
// scope: the page is host-coherent), and the host polls that word instead of sleeping in hipStreamSynchronize: the
// stream is in order, so the word's arrival says that everything queued before has run.  A build reads a handful of
// words back some twenty times (one per bucket region of the induced-sort passes); the wake-up through the runtime
// cost 10 - 15 us of idle device each time.  The poll gives up after two seconds and falls back to the synchronize
// (which also reports a device fault).
namespace sx {
__global__ void readback_kernel(const uint32_t *__restrict__ src, uint32_t count, uint32_t *__restrict__ page, uint32_t seq)
{
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) __hip_atomic_store(&page[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(&page[1024], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// the same for up to four arrays, one behind the other on the page (the induced-sort passes read a pass's stop record, its
// cursors and, at the end, two error words: one wake-up of the host instead of four)
struct readback_ranges {
    const uint32_t *src[4];
    uint32_t count[4];
};
__global__ void readback_ranges_kernel(readback_ranges r, uint32_t *__restrict__ page, uint32_t seq)
{
    uint32_t at = 0;
    for (int k = 0; k < 4; ++k) {
        for (uint32_t i = threadIdx.x; i < r.count[k]; i += blockDim.x)
            __hip_atomic_store(&page[at + i], r.src[k][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        at += r.count[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(&page[1024], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
} // namespace sx

constexpr int kReadbackSpinUs = 150;
static int readback_wait(sx_ctx *ctx, uint32_t seq, size_t count, uint32_t *h_dst);
int sx_readback(sx_ctx *ctx, const uint32_t *d_src, size_t count, uint32_t *h_dst)
{
    if (count > 1024) return sx_fail_msg(ctx, SX_E_INTERNAL, "readback too large");
    const uint32_t seq = ++ctx->readback_seq ? ctx->readback_seq : ++ctx->readback_seq; // (never 0: the page starts zeroed)
    hipLaunchKernelGGL(sx::readback_kernel, dim3(1), dim3(256), 0, ctx->stream, d_src, (uint32_t)count, ctx->h_pin, seq);
    if (hipGetLastError() != hipSuccess) return sx_fail_msg(ctx, SX_E_INTERNAL, "readback launch refused");
    return readback_wait(ctx, seq, count, h_dst);
}

int sx_readback_ranges(sx_ctx *ctx, const uint32_t *const *d_src, const uint32_t *counts, int ranges, uint32_t *h_dst)
{
    sx::readback_ranges r = {{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}};
    size_t total = 0;
    if (ranges < 1 || ranges > 4) return sx_fail_msg(ctx, SX_E_INTERNAL, "readback: one to four arrays");
    for (int k = 0; k < ranges; ++k) r.src[k] = d_src[k], r.count[k] = counts[k], total += counts[k];
    if (total > 1024) return sx_fail_msg(ctx, SX_E_INTERNAL, "readback too large");
    const uint32_t seq = ++ctx->readback_seq ? ctx->readback_seq : ++ctx->readback_seq;
    hipLaunchKernelGGL(sx::readback_ranges_kernel, dim3(1), dim3(256), 0, ctx->stream, r, ctx->h_pin, seq);
    if (hipGetLastError() != hipSuccess) return sx_fail_msg(ctx, SX_E_INTERNAL, "readback launch refused");
    return readback_wait(ctx, seq, total, h_dst);
}

static int readback_wait(sx_ctx *ctx, uint32_t seq, size_t count, uint32_t *h_dst)
{
    // The word usually arrives within tens of microseconds (the device is a launch or two behind the host), which is
    // what the poll is for: hipStreamSynchronize sleeps through that.  A wait that lasts longer -- milliseconds of queued
    // work in front of the read-back -- is left to the runtime: a farm runs several workers a device, all pinned to the
    // CPUs of one NUMA node beside the pager and remap helpers, and a spinning worker takes a core from those.
    volatile uint32_t *flag = ctx->h_pin + 1024;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (*flag != seq) {
        __builtin_ia32_pause();
        if ((++spins & 0xFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kReadbackSpinUs)) break;
    }
    if (*flag != seq) SX_CHECK(hipStreamSynchronize(ctx->stream)); // (a long wait, or a fault: let the runtime say)
    std::atomic_thread_fence(std::memory_order_acquire);
    if (*flag != seq) return sx_fail_msg(ctx, SX_E_INTERNAL, "readback: the sequence word did not arrive");
    if (ctx->launch_err != hipSuccess) return launch_error(ctx);
    memcpy(h_dst, ctx->h_pin, count * sizeof(uint32_t));
    return 0;
}

sx_event_pair *sx_prof_pair(sx_ctx *ctx, int kclass)
{
    sx_event_pair ep;
    if (!ctx->ev_free.empty()) {
        ep = ctx->ev_free.back();
        ctx->ev_free.pop_back();
    } else {
        if (hipEventCreate(&ep.a) != hipSuccess || hipEventCreate(&ep.b) != hipSuccess) return nullptr;
    }
    ep.kclass = kclass;
    ctx->ev_used.push_back(ep);
    return &ctx->ev_used.back();
}

void sx_prof_count(sx_ctx *ctx, int kclass, uint64_t alg_bytes)
{
    ctx->kstat[kclass].launches += 1;
    ctx->kstat[kclass].alg_bytes += alg_bytes;
}

static void prof_drain(sx_ctx *ctx)
{
    if (ctx->ev_used.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (sx_event_pair &ep : ctx->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) ctx->kstat[ep.kclass].ms += ms;
        ctx->ev_free.push_back(ep);
    }
    ctx->ev_used.clear();
}

int sx_child_begin(sx_ctx *ctx, sx_ctx **out)
{
    if (!ctx->child) {
        int rc = sx_ctx_create(ctx->device, &ctx->child);
        if (rc != 0) return sx_fail_msg(ctx, rc, "cannot create the context of the reduced string");
        ctx->child->depth = ctx->depth + 1;
    }
    sx_ctx *c = ctx->child;
    c->chain_max_override = ctx->chain_max_override;
    c->radix_digit_bits = ctx->radix_digit_bits;
    c->sort_mode = ctx->sort_mode;
    c->induce_batch_off = ctx->induce_batch_off;
    c->induce_batch_min = ctx->induce_batch_min;
    c->induce_attended = ctx->induce_attended;
    c->induce_no_hoist = ctx->induce_no_hoist;
    c->text_keys_off = ctx->text_keys_off;
    c->long_subbuckets_off = ctx->long_subbuckets_off;
    c->local_sort_lean_off = ctx->local_sort_lean_off;
    c->small_direct_max = ctx->small_direct_max;
    c->copy_text_first = ctx->copy_text_first;
    c->recurse_min = ctx->recurse_min;
    c->sample_min = ctx->sample_min;
    c->no_direct = 1;      // (a reduced string that got here has too many ties for any prefix sort)
    c->force_general = 1;
    c->prefix_symbols = 0;
    c->prof_on = ctx->prof_on;
    c->prof_only = ctx->prof_only;
    *out = c;
    return 0;
}

void sx_child_end(sx_ctx *ctx, sx_ctx *c)
{
    prof_drain(c);
    for (int k = 0; k < SX_KC_COUNT; ++k) {
        ctx->kstat[k].launches += c->kstat[k].launches;
        ctx->kstat[k].ms += c->kstat[k].ms;
        ctx->kstat[k].alg_bytes += c->kstat[k].alg_bytes;
    }
    memset(c->kstat, 0, sizeof c->kstat);
    c->prof_on = 0;
}

extern "C" {

int sx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sx_device_numa_node(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return -1;
    for (char *p = bus; *p; ++p)
        if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a'); // sysfs spells bus ids in lower case
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

static std::atomic<int> g_live_contexts{0};
extern "C" int sx_ctx_live_count(void) { return g_live_contexts.load(); }

int sx_ctx_create(int device, sx_ctx **out)
{
    if (!out) return SX_E_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        fprintf(stderr, "stralg_amd: no usable HIP device (%s); this library has no CPU fallback\n",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return e != hipSuccess ? (int)e : SX_E_INTERNAL;
    }
    if (device < 0 || device >= ndev) return SX_E_ARG;
    sx_ctx *ctx = new (std::nothrow) sx_ctx();
    if (!ctx) return SX_E_NOMEM;
    ctx->device = device;
    memset(ctx->kstat, 0, sizeof ctx->kstat);
    memset(&ctx->stats, 0, sizeof ctx->stats);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void **)&ctx->h_pin, 8192, hipHostMallocDefault) != hipSuccess) {
        fprintf(stderr, "stralg_amd: cannot initialise device %d\n", device);
        delete ctx;
        return SX_E_INTERNAL;
    }
    memset(ctx->h_pin, 0, 8192); // (the read-back sequence word starts at 0)
    g_live_contexts.fetch_add(1);
    *out = ctx;
    return 0;
}

void sx_ctx_trim(sx_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->child) sx_ctx_trim(ctx->child);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < SX_NSLABS; ++i) {
        if (ctx->slab[i].p) (void)hipFree(ctx->slab[i].p);
        ctx->slab[i].p = nullptr;
        ctx->slab[i].cap = 0;
    }
}

void sx_ctx_destroy(sx_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->child) sx_ctx_destroy(ctx->child);
    ctx->child = nullptr;
    sx_ctx_trim(ctx);
    prof_drain(ctx);
    for (sx_event_pair &ep : ctx->ev_free) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    for (char *b : ctx->h_stage)
        if (b) (void)hipHostFree(b);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    g_live_contexts.fetch_sub(1);
    delete ctx;
}

int sx_ctx_set_flag(sx_ctx *ctx, int flag, int value)
{
    if (!ctx) return SX_E_ARG;
    if (flag == SX_FLAG_FORCE_GENERAL_PATH) {
        ctx->force_general = value ? 1 : 0;
        return 0;
    }
    if (flag == SX_FLAG_NO_DIRECT_SORT) {
        ctx->no_direct = value ? 1 : 0;
        return 0;
    }
    if (flag == SX_FLAG_PREFIX_SYMBOLS) {
        ctx->prefix_symbols = value > 0 ? value : 0;
        return 0;
    }
    if (flag == SX_FLAG_SORT_MODE) {
        if (value < 0 || value > 3) return SX_E_ARG;
        ctx->sort_mode = value;
        return 0;
    }
    if (flag == SX_FLAG_RADIX_DIGIT_BITS) {
        if (value != 0 && (value < 8 || value > 10)) return SX_E_ARG;
        ctx->radix_digit_bits = value;
        return 0;
    }
    if (flag == SX_FLAG_INDUCE_BATCH_OFF) {
        ctx->induce_batch_off = value ? 1 : 0;
        return 0;
    }
    if (flag == SX_FLAG_SAMPLE_MIN) {
        ctx->sample_min = value < 0 ? -1 : (int64_t)value;
        return 0;
    }
    if (flag == SX_FLAG_RECURSE_MIN) {
        ctx->recurse_min = value < 0 ? -1 : (int64_t)value;
        return 0;
    }
    if (flag == SX_FLAG_COPY_TEXT_FIRST) {
        ctx->copy_text_first = value ? 1 : 0;
        return 0;
    }
    if (flag == SX_FLAG_INDUCE_ATTENDED) {
        if (value < 0 || value > 2) return SX_E_ARG;
        ctx->induce_attended = value == 1 ? 1 : 0; // (2: what 0 is now)
        return 0;
    }
    if (flag == SX_FLAG_TEXT_KEYS_OFF) {
        ctx->text_keys_off = value ? 1 : 0;
        return 0;
    }
    if (flag == SX_FLAG_SMALL_DIRECT_MAX) {
        ctx->small_direct_max = value < 0 ? -1 : (int64_t)value;
        return 0;
    }
    if (flag == SX_FLAG_LONG_SUBBUCKETS_OFF) {
        ctx->long_subbuckets_off = value ? 1 : 0;
        return 0;
    }
    if (flag == SX_FLAG_LOCAL_SORT_LEAN_OFF) {
        ctx->local_sort_lean_off = value ? 1 : 0;
        return 0;
    }
    if (flag == SX_FLAG_INDUCE_NO_HOIST) {
        ctx->induce_no_hoist = value ? 1 : 0;
        return 0;
    }
    if (flag == SX_FLAG_INDUCE_BATCH_MIN) {
        ctx->induce_batch_min = value < 0 ? -1 : (int64_t)value;
        return 0;
    }
    if (flag == SX_FLAG_CHAIN_MAX_ENTRIES) {
        ctx->chain_max_override = value < 0 ? -1 : (int64_t)value; // negative: back to the default
        return 0;
    }
    return SX_E_ARG;
}

const char *sx_last_error(const sx_ctx *ctx) { return ctx ? ctx->err : "no context"; }

int sx_profile_enable(sx_ctx *ctx, int on)
{
    if (!ctx) return SX_E_ARG;
    if (!on) prof_drain(ctx);
    ctx->prof_on = on ? 1 : 0;
    return 0;
}

int sx_profile_only(sx_ctx *ctx, int kclass)
{
    if (!ctx || kclass >= SX_KC_COUNT) return SX_E_ARG;
    ctx->prof_only = kclass < 0 ? -1 : kclass;
    return 0;
}

int sx_profile_reset(sx_ctx *ctx)
{
    if (!ctx) return SX_E_ARG;
    prof_drain(ctx);
    memset(ctx->kstat, 0, sizeof ctx->kstat);
    return 0;
}

int sx_profile_read(sx_ctx *ctx, sx_kernel_stat *out)
{
    if (!ctx || !out) return SX_E_ARG;
    prof_drain(ctx);
    memcpy(out, ctx->kstat, sizeof ctx->kstat);
    return 0;
}

const char *sx_kernel_class_name(int kclass)
{
    return kclass >= 0 && kclass < SX_KC_COUNT ? kClassNames[kclass] : "?";
}

int sx_last_stats(const sx_ctx *ctx, sx_build_stats *out)
{
    if (!ctx || !out) return SX_E_ARG;
    *out = ctx->stats;
    return 0;
}

} // extern "C"
