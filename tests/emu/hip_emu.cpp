/* tests/emu/hip_emu.cpp -- fiber scheduler behind tests/emu/include/hip/hip_runtime.h.
 * TEST INFRASTRUCTURE ONLY (see that header). */
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <mutex>

// Minimal x86-64 context switch (callee-saved registers + stack pointer); glibc's
// swapcontext makes a signal-mask system call per switch, far too slow here.
extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(".text\n.globl emu_switch\n.type emu_switch,@function\nemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size emu_switch,.-emu_switch\n");

uint3_ threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
const size_t kStack = 256 * 1024;
const int kMaxThreads = 1024;

struct Fiber { void *sp; bool done; };
struct Wave {
    int arrived, nlive; unsigned gen; unsigned long long live;
    unsigned long long vals[2][64]; unsigned long long done_live[2];
};

Fiber g_fib[kMaxThreads];
char *g_stacks = nullptr;
void *g_sched_sp;
Wave g_wave[kMaxThreads / 64];
int g_nthreads, g_cur, g_block_live, g_bar_arrived;
unsigned g_bar_gen;
unsigned long long g_progress;
void (*g_tramp)(void *);
void *g_args;

void set_tid(int t)
{
    g_cur = t;
    threadIdx.x = t % blockDim.x;
    threadIdx.y = (t / blockDim.x) % blockDim.y;
    threadIdx.z = t / (blockDim.x * blockDim.y);
}

void yield_() { int me = g_cur; emu_switch(&g_fib[me].sp, g_sched_sp); }

void complete_wave(Wave &w)
{
    unsigned g = w.gen;
    w.done_live[g & 1] = w.live;
    w.arrived = 0;
    w.gen++;
    g_progress++;
}

void fiber_main()
{
    g_tramp(g_args);
    int me = g_cur;
    g_fib[me].done = true;
    g_progress++;
    Wave &w = g_wave[me / 64];
    w.nlive--;
    w.live &= ~(1ull << (me % 64));
    if (w.nlive > 0 && w.arrived == w.nlive) complete_wave(w);
    g_block_live--;
    if (g_block_live > 0 && g_bar_arrived == g_block_live) { g_bar_arrived = 0; g_bar_gen++; g_progress++; }
    emu_switch(&g_fib[me].sp, g_sched_sp);
    abort(); // a finished fiber is never resumed
}
} // namespace

int emu_lane() { return g_cur % 64; }

void emu_syncthreads()
{
    unsigned g = g_bar_gen;
    g_bar_arrived++;
    g_progress++;
    if (g_bar_arrived == g_block_live) { g_bar_arrived = 0; g_bar_gen++; }
    else while (g_bar_gen == g) yield_();
}

const unsigned long long *emu_wave_gather(unsigned long long v, unsigned long long *live_mask)
{
    Wave &w = g_wave[g_cur / 64];
    unsigned g = w.gen;
    w.vals[g & 1][g_cur % 64] = v;
    w.arrived++;
    g_progress++;
    if (w.arrived == w.nlive) complete_wave(w);
    else while (w.gen == g) yield_();
    *live_mask = w.done_live[g & 1];
    return w.vals[g & 1];
}

void emu_launch(void (*tramp)(void *), void *args, dim3 grid, dim3 block)
{
    // one kernel at a time: the scheduler's state, threadIdx & co. and the kernels' __shared__ arrays (plain statics
    // here) are process-wide, and the host layer may launch from several threads (one context each)
    static std::mutex launch_mu;
    std::lock_guard<std::mutex> lock(launch_mu);
    int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || nt > kMaxThreads) { fprintf(stderr, "emu: bad block size %d\n", nt); abort(); }
    if ((unsigned long long)grid.x * grid.y * grid.z == 0) { fprintf(stderr, "emu: empty grid\n"); abort(); }
    if (!g_stacks) {
        g_stacks = (char *)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == (char *)MAP_FAILED) { perror("mmap"); abort(); }
    }
    g_tramp = tramp; g_args = args; g_nthreads = nt;
    blockDim = block; gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
        g_block_live = nt; g_bar_arrived = 0; g_bar_gen = 0;
        for (int w = 0; w * 64 < nt; ++w) {
            Wave &wv = g_wave[w];
            int nl = nt - w * 64 < 64 ? nt - w * 64 : 64;
            wv.arrived = 0; wv.nlive = nl; wv.gen = 0;
            wv.live = nl == 64 ? ~0ull : ((1ull << nl) - 1);
        }
        for (int t = 0; t < nt; ++t) {
            Fiber &f = g_fib[t];
            f.done = false;
            // initial frame: six zeroed callee-saved registers, then the entry point as
            // return address; 16-byte stack alignment at fiber_main's entry
            void **top = (void **)(g_stacks + (size_t)(t + 1) * kStack);
            top[-1] = nullptr;
            top[-2] = (void *)fiber_main;
            for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
            f.sp = (void *)(top - 8);
        }
        int remaining = nt;
        while (remaining > 0) {
            unsigned long long before = g_progress;
            remaining = 0;
            for (int t = 0; t < nt; ++t) {
                if (g_fib[t].done) continue;
                set_tid(t);
                emu_switch(&g_sched_sp, g_fib[t].sp);
                if (!g_fib[t].done) remaining++;
            }
            if (remaining > 0 && g_progress == before) {
                fprintf(stderr, "emu: DEADLOCK in block (%u,%u,%u): %d threads blocked at a barrier or wave collective "
                                "that not every live lane reaches\n", bx, by, bz, remaining);
                abort();
            }
        }
    }
}

/* ---- runtime API ---- */
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char *bus, int len, int) { if (len > 0) bus[0] = 0; return hipErrorInvalidValue; } // (no PCI device behind the harness)
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof *p); strcpy(p->name, "cpu-emu"); strcpy(p->gcnArchName, "emu"); p->multiProcessorCount = 4; p->totalGlobalMem = 8ull << 30; return hipSuccess; }
hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = 8ull << 30; *t = 8ull << 30; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emu error"; }
