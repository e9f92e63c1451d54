/*
 * tests/emu -- CPU execution harness for the HIP kernel sources in
 * stralg_amd/csrc.  TEST INFRASTRUCTURE ONLY: it lets the build container
 * (which has no GPU) run the very same .hip files under g++ / ASan so that
 * kernel logic and out-of-bounds accesses are caught before a kernel ever
 * reaches a real MI355X.  It is not shipped, not a fallback, and the product
 * loader (stralg_amd/_lib.py) never loads a library built with it.
 *
 * Model: one OS thread; every HIP thread of a workgroup is a ucontext fiber;
 * workgroups run one after another.  __syncthreads() and the 64-lane wave
 * collectives (__ballot, __shfl*, ...) are rendezvous points between fibers.
 * A collective that not every live lane of the wave reaches is reported as a
 * deadlock (on hardware it would be a divergence bug).
 */
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <type_traits>
#include <utility>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };

extern uint3_ threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 64;

typedef int hipError_t;
typedef struct emuStream *hipStream_t;
typedef struct emuEvent *hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
struct hipDeviceProp_t { char name[256]; size_t totalGlobalMem; int multiProcessorCount; char gcnArchName[256]; };

hipError_t hipMalloc(void **p, size_t n);
template <class T> hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags = 0);
template <class T> hipError_t hipHostMalloc(T **p, size_t n, unsigned flags = 0) { return hipHostMalloc((void **)p, n, flags); }
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned f);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipDeviceGetPCIBusId(char *bus, int len, int device);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);

/* ---- scheduler hooks (hip_emu.cpp) ---- */
void emu_launch(void (*tramp)(void *), void *args, dim3 grid, dim3 block);
void emu_syncthreads();
/* deposits v, waits for the wave, returns pointer to the 64 deposited values
 * (valid until this lane's next collective) and the live-lane mask */
const unsigned long long *emu_wave_gather(unsigned long long v, unsigned long long *live_mask);
int emu_lane();

template <class K, class Tup, size_t... I>
static void emu_apply(K k, Tup &t, std::index_sequence<I...>) { k(std::get<I>(t)...); }

template <class... P, class... A>
static void hipLaunchKernelGGL(void (*kernel)(P...), dim3 grid, dim3 block, size_t /*lds*/, hipStream_t /*st*/, A... args)
{
    struct Pack { void (*k)(P...); std::tuple<P...> a; };
    Pack pk{kernel, std::tuple<P...>(static_cast<P>(args)...)};
    emu_launch([](void *p) { Pack *q = (Pack *)p; emu_apply(q->k, q->a, std::index_sequence_for<P...>{}); }, &pk, grid, block);
}
/* hip_ext.h: the launch whose two events take the dispatch's own start and end times */
template <class... P, class... A>
static void hipExtLaunchKernelGGL(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t st, hipEvent_t start,
                                  hipEvent_t stop, unsigned /*flags*/, A... args)
{
    if (start) (void)hipEventRecord(start, st);
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
    if (stop) (void)hipEventRecord(stop, st);
}

static inline void __syncthreads() { emu_syncthreads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}
static inline int __lane_id() { return emu_lane(); }

static inline unsigned long long __ballot(int pred)
{
    unsigned long long live;
    const unsigned long long *v = emu_wave_gather(pred ? 1ull : 0ull, &live);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (((live >> l) & 1) && v[l]) m |= 1ull << l;
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) { unsigned long long live; const unsigned long long *v = emu_wave_gather(pred ? 1ull : 0ull, &live); for (int l = 0; l < 64; ++l) if (((live >> l) & 1) && !v[l]) return 0; return 1; }

template <class T> static inline unsigned long long emu_bits(T v) { unsigned long long b = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&b, &v, sizeof(T)); return b; }
template <class T> static inline T emu_unbits(unsigned long long b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define ext_vector_type(n) vector_size((n) * 4) /* g++ spelling of clang's 4-byte-element vectors */
#define __builtin_nontemporal_load(p) (*(p))
static inline unsigned int __umul24(unsigned int a, unsigned int b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
template <class T> static inline T __shfl(T var, int src, int width = 64)
{
    unsigned long long live; const unsigned long long *v = emu_wave_gather(emu_bits(var), &live);
    int lane = emu_lane(); int base = lane & ~(width - 1); int s = base + (src & (width - 1));
    return emu_unbits<T>(v[s]);
}
template <class T> static inline T __shfl_up(T var, unsigned delta, int width = 64)
{
    unsigned long long live; const unsigned long long *v = emu_wave_gather(emu_bits(var), &live);
    int lane = emu_lane(); int base = lane & ~(width - 1); int s = lane - (int)delta;
    return emu_unbits<T>(s < base ? v[lane] : v[s]);
}
template <class T> static inline T __shfl_down(T var, unsigned delta, int width = 64)
{
    unsigned long long live; const unsigned long long *v = emu_wave_gather(emu_bits(var), &live);
    int lane = emu_lane(); int base = lane & ~(width - 1); int s = lane + (int)delta;
    return emu_unbits<T>(s >= base + width ? v[lane] : v[s]);
}
template <class T> static inline T __shfl_xor(T var, int mask, int width = 64)
{
    unsigned long long live; const unsigned long long *v = emu_wave_gather(emu_bits(var), &live);
    int lane = emu_lane(); int s = lane ^ mask; (void)width;
    return emu_unbits<T>(s >= 64 ? v[lane] : v[s]);
}
static inline int __builtin_amdgcn_readfirstlane(int x)
{
    unsigned long long live; const unsigned long long *v = emu_wave_gather((unsigned long long)(unsigned)x, &live);
    return (int)(unsigned)v[__builtin_ctzll(live)];
}

static inline int __builtin_amdgcn_readlane(int x, int lane)
{
    unsigned long long live; const unsigned long long *v = emu_wave_gather((unsigned long long)(unsigned)x, &live);
    return (int)(unsigned)v[lane];
}
// data-parallel-primitive moves used by the wave scans (sx_device.hpp): row_shr:n (0x110 + n), row_bcast:15 (0x142),
// row_bcast:31 (0x143); old where the row / bank mask disables the lane; 0 (bound_ctrl) or old where there is no source
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    unsigned long long live; const unsigned long long *v = emu_wave_gather((unsigned long long)(unsigned)src, &live);
    const int lane = emu_lane(), row = lane >> 4, bank = (lane >> 2) & 3;
    if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; if ((lane & 15) >= n) from = lane - n; }
    else if (ctrl == 0x142) { if (row >= 1) from = row * 16 - 1; }
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }
    if (from < 0 || !((live >> from) & 1)) return bound_ctrl ? 0 : old;
    return (int)(unsigned)v[from];
}

// wave-level helpers of the ranking code: a scheduling barrier is a rendezvous of the wave's fibers here
static inline void __builtin_amdgcn_wave_barrier() { (void)__ballot(0); }
static inline unsigned emu_bitop3(unsigned a, unsigned b, unsigned c, unsigned tt)
{
    unsigned r = 0;
    for (int i = 0; i < 32; ++i) {
        const unsigned idx = (((a >> i) & 1u) << 2) | (((b >> i) & 1u) << 1) | ((c >> i) & 1u);
        r |= ((tt >> idx) & 1u) << i;
    }
    return r;
}
#define __builtin_amdgcn_bitop3_b32(a, b, c, tt) emu_bitop3((a), (b), (c), (tt))
static inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool)
{
    for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
    return c;
}
static inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh)
{
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (sh & 3u)));
}
#define SX_OPAQUE_VGPR(x) ((void)0) /* a register-allocation hint on the GPU */
#define SX_SCHED_FENCE() ((void)0)  /* a scheduling fence on the GPU */
#define SX_WAVES_PER_EU(N)          /* a register budget on the GPU */
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned m, unsigned add)
{
    const int l = emu_lane();
    return add + (unsigned)__builtin_popcount(l >= 32 ? m : (m & ((1u << l) - 1u)));
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned m, unsigned add)
{
    const int l = emu_lane();
    return add + (l > 32 ? (unsigned)__builtin_popcount(m & ((1u << (l - 32)) - 1u)) : 0u);
}

static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }

#define __ATOMIC_RELAXED_EMU 0
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
#ifndef __HIP_MEMORY_SCOPE_SYSTEM
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
#ifndef __HIP_MEMORY_SCOPE_WORKGROUP
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#endif
template <class T> static inline T __hip_atomic_load(const T *p, int, int) { return *p; }
template <class T, class V> static inline void __hip_atomic_store(T *p, V v, int, int) { *p = (T)v; }
static inline void __builtin_amdgcn_s_sleep(int) {}

template <class T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicSub(T *p, T v) { T o = *p; *p = o - v; return o; }
template <class T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
