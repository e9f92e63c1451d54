// Measurement reference only (not linked into the product): rocPRIM's device radix sort on the
// same shape as the LMS-suffix sort (u64 keys, u32 values, 40 key bits), to know what the
// library shipped with ROCm reaches on this GPU.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void fill(uint64_t *k, uint32_t *v, size_t n, int kbits)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = 12345 + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    k[i] = z & ((1ull << kbits) - 1);
    v[i] = (uint32_t)i;
}

int main(int argc, char **argv)
{
    size_t n = argc > 1 ? (size_t)atof(argv[1]) : 300000000;
    int kbits = argc > 2 ? atoi(argv[2]) : 40;
    uint64_t *ki, *ko;
    uint32_t *vi, *vo;
    hipMalloc(&ki, n * 8); hipMalloc(&ko, n * 8); hipMalloc(&vi, n * 4); hipMalloc(&vo, n * 4);
    size_t tmp_bytes = 0;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, ki, ko, vi, vo, n, 0, kbits);
    void *tmp; hipMalloc(&tmp, tmp_bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) {
        fill<<<(n + 255) / 256, 256>>>(ki, vi, n, kbits);
        hipEventRecord(a);
        rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, n, 0, kbits);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("rocprim radix_sort_pairs n=%zu kbits=%d: %.2f ms (tmp %.1f MB)\n", n, kbits, ms, tmp_bytes / 1e6);
    }
    return 0;
}
