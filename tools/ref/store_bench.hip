// store patterns against hipMemset on the same buffer: which launch shape reaches the fill rate (tools/ref: measurement aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
// chunk bytes per workgroup iteration, threads, loop over chunks by grid stride
template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void chunk_store(uint4 *dst, uint64_t nvec, uint32_t chunk_vec)
{
    const uint64_t chunks = (nvec + chunk_vec - 1) / chunk_vec;
    for (uint64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint64_t v0 = c * chunk_vec;
        const uint32_t cnt = nvec - v0 < chunk_vec ? (uint32_t)(nvec - v0) : chunk_vec;
        for (uint32_t i = threadIdx.x; i < cnt; i += THREADS) {
            const v4u x = {1u, 2u, 3u, (uint32_t)i};
            if (NT) __builtin_nontemporal_store(x, reinterpret_cast<v4u *>(dst + v0 + i));
            else *reinterpret_cast<v4u *>(dst + v0 + i) = x;
        }
    }
}
int main()
{
    const uint64_t bytes = (uint64_t)((1ull << 30) + 2) * 20; // the O table of 1 GiB of DNA
    const uint64_t nvec = bytes / 16;
    uint4 *d;
    CK(hipMalloc(&d, bytes + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto fn) {
        fn(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0, 0); fn(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-58s %7.3f ms  %6.2f TB/s\n", name, best, bytes / best / 1e9);
    };
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); });
    timeit("hipMemsetD32Async", [&] { hipMemsetD32Async((hipDeviceptr_t)d, 7, bytes / 4, 0); });
    struct V { const char *n; int threads; bool nt; uint32_t chunk_vec; uint32_t grid; };
    const uint32_t all = 0xFFFFFFFFu;
    std::vector<V> vs = {
        {"256 thr, 20 KiB chunks, grid 65536, plain", 256, false, 1280, 65536}, {"256 thr, 20 KiB chunks, grid 65536, nontemporal", 256, true, 1280, 65536},
        {"256 thr, 20 KiB chunks, no loop, plain", 256, false, 1280, all},       {"256 thr, 4 KiB chunks, no loop, plain", 256, false, 256, all},
        {"256 thr, 4 KiB chunks, no loop, nontemporal", 256, true, 256, all},    {"256 thr, 16 KiB chunks, no loop, plain", 256, false, 1024, all},
        {"256 thr, 64 KiB chunks, no loop, plain", 256, false, 4096, all},       {"256 thr, 4 KiB chunks, grid 16384, plain", 256, false, 256, 16384},
        {"1024 thr, 16 KiB chunks, no loop, plain", 1024, false, 1024, all},     {"1024 thr, 64 KiB chunks, grid 4096, plain", 1024, false, 4096, 4096},
        {"512 thr, 8 KiB chunks, no loop, plain", 512, false, 512, all},         {"64 thr, 1 KiB chunks, no loop, plain", 64, false, 64, all},
        {"1024 thr, 20 KiB chunks, no loop, plain", 1024, false, 1280, all},     {"640 thr, 10 KiB chunks, no loop, plain", 640, false, 640, all},
        {"320 thr, 5 KiB chunks, no loop, plain", 320, false, 320, all},         {"256 thr, 8 KiB chunks, no loop, plain", 256, false, 512, all},
        {"256 thr, 12 KiB chunks, no loop, plain", 256, false, 768, all},        {"640 thr, 10 KiB chunks, no loop, nontemporal", 641, true, 640, all},
        {"320 thr, 5 KiB chunks, no loop, nontemporal", 321, true, 320, all},    {"1024 thr, 20 KiB chunks, grid 65536, plain", 1024, false, 1280, 65536},
    };
    for (const V &v : vs) {
        const uint64_t chunks = (nvec + v.chunk_vec - 1) / v.chunk_vec;
        const uint32_t grid = v.grid == all ? (uint32_t)chunks : v.grid;
        timeit(v.n, [&] {
            if (v.threads == 256 && !v.nt) chunk_store<256, false><<<grid, 256>>>(d, nvec, v.chunk_vec);
            else if (v.threads == 256) chunk_store<256, true><<<grid, 256>>>(d, nvec, v.chunk_vec);
            else if (v.threads == 1024) chunk_store<1024, false><<<grid, 1024>>>(d, nvec, v.chunk_vec);
            else if (v.threads == 512) chunk_store<512, false><<<grid, 512>>>(d, nvec, v.chunk_vec);
            else if (v.threads == 640) chunk_store<640, false><<<grid, 640>>>(d, nvec, v.chunk_vec);
            else if (v.threads == 641) chunk_store<640, true><<<grid, 640>>>(d, nvec, v.chunk_vec);
            else if (v.threads == 320) chunk_store<320, false><<<grid, 320>>>(d, nvec, v.chunk_vec);
            else if (v.threads == 321) chunk_store<320, true><<<grid, 320>>>(d, nvec, v.chunk_vec);
            else chunk_store<64, false><<<grid, 64>>>(d, nvec, v.chunk_vec);
        });
    }
    return 0;
}
