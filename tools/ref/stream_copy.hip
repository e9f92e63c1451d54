// copy-rate reference points for the roofline discussion: device copy kernels with different numbers of 16-byte
// loads in flight per thread, with and without streaming (non-temporal) loads / stores.
//   hipcc --offload-arch=gfx950 -O3 -o stream_copy stream_copy.hip && ./stream_copy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int K, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_kernel(const v4u *__restrict__ src, v4u *__restrict__ dst, uint64_t n16)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i0 = (uint64_t)blockIdx.x * 256 * K + threadIdx.x; i0 < n16; i0 += stride * K) {
        v4u v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t i = i0 + (uint64_t)k * 256;
            if (i < n16) v[k] = NTL ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t i = i0 + (uint64_t)k * 256;
            if (i < n16) {
                if (NTS) __builtin_nontemporal_store(v[k], dst + i);
                else dst[i] = v[k];
            }
        }
    }
}
template <int K, bool NTL, bool NTS> static void run(const v4u *s, v4u *d, uint64_t n16, int grid, const char *name)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    copy_kernel<K, NTL, NTS><<<grid, 256>>>(s, d, n16);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) copy_kernel<K, NTL, NTS><<<grid, 256>>>(s, d, n16);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    printf("%-28s grid %6d: %.3f ms  %.2f TB/s (read + written)\n", name, grid, ms, (double)n16 * 32 / ms / 1e9);
}
int main()
{
    const uint64_t bytes = 8ull << 30, n16 = bytes / 16;
    v4u *s, *d;
    hipMalloc(&s, bytes), hipMalloc(&d, bytes);
    hipMemset(s, 1, bytes), hipMemset(d, 0, bytes);
    for (int grid : {2048, 8192, 65536}) {
        run<1, false, false>(s, d, n16, grid, "1 load, plain");
        run<4, false, false>(s, d, n16, grid, "4 loads, plain");
        run<8, false, false>(s, d, n16, grid, "8 loads, plain");
        run<4, true, false>(s, d, n16, grid, "4 loads nt, stores plain");
        run<4, false, true>(s, d, n16, grid, "4 loads plain, stores nt");
        run<4, true, true>(s, d, n16, grid, "4 loads nt, stores nt");
        run<8, true, true>(s, d, n16, grid, "8 loads nt, stores nt");
    }
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("hipMemcpy device to device: %.3f ms  %.2f TB/s\n", ms / 5, (double)bytes * 2 / (ms / 5) / 1e9);
    hipMemsetAsync(d, 0, bytes, 0);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipMemsetAsync(d, 0, bytes, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("hipMemset: %.3f ms  %.2f TB/s written\n", ms / 5, (double)bytes / (ms / 5) / 1e9);
    return 0;
}
