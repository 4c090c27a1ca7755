// mall_probe.hip -- does a per-workgroup multi-pass scatter stay in L2 / Infinity Cache?  (design probe, round 2)
// Every workgroup (1024 threads, one per CU) owns a private region of S bytes and ping-pongs P passes over it: read 16 K-key
// chunks sequentially, write them as 512 runs of 32 keys (256 B) to 512 bucket cursors inside the region -- the access
// pattern of an LSD radix pass confined to one barcode's keys.  Prints GB/s (read + write) per region size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(1024) void k_probe(uint64_t *a, uint64_t *b, uint64_t keys_per_wg, int passes) {
    uint64_t *src = a + (uint64_t)blockIdx.x * keys_per_wg, *dst = b + (uint64_t)blockIdx.x * keys_per_wg;
    const uint64_t n_chunks = keys_per_wg / 16384, cap = keys_per_wg / 512;
    for (int p = 0; p < passes; p++) {
        for (uint64_t c = 0; c < n_chunks; c++) {
            uint64_t v[16];
#pragma unroll
            for (int it = 0; it < 16; it++) v[it] = src[c * 16384 + it * 1024 + threadIdx.x];
#pragma unroll
            for (int it = 0; it < 16; it++) {
                const uint32_t k = it * 1024 + threadIdx.x, r = k >> 5, j = k & 31;
                dst[(uint64_t)r * cap + c * 32 + j] = v[it] + 1;
            }
        }
        __syncthreads();
        __threadfence();
        uint64_t *t = src; src = dst; dst = t;
    }
}
__global__ void k_copy(const uint4 *a, uint4 *b, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const uint64_t total_max = 4ull << 30;
    uint64_t *a, *b;
    CHECK(hipMalloc(&a, total_max)); CHECK(hipMalloc(&b, total_max));
    CHECK(hipMemset(a, 1, total_max)); CHECK(hipMemset(b, 2, total_max));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {
        const uint64_t n = total_max / 16;
        k_copy<<<4096, 256>>>((uint4 *)a, (uint4 *)b, n);
        hipEventRecord(e0); k_copy<<<4096, 256>>>((uint4 *)a, (uint4 *)b, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy 4 GiB uint4: %.3f ms  %.0f GB/s (r+w)\n", ms, 2.0 * total_max / ms / 1e6);
    }
    const int passes = 6;
    for (int grid : {256, 512}) for (uint64_t S : {128ull << 10, 256ull << 10, 512ull << 10, 1ull << 20, 2ull << 20, 4ull << 20, 8ull << 20}) {
        const uint64_t keys = S / 8;
        if (keys * 8 * grid > total_max) continue;
        k_probe<<<grid, 1024>>>(a, b, keys, 2);
        hipEventRecord(e0);
        k_probe<<<grid, 1024>>>(a, b, keys, passes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        CHECK(hipGetLastError());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %d region %5llu KB/WG (footprint %6.0f MB x2): %7.3f ms for %d passes  %.0f GB/s (r+w)\n", grid,
               (unsigned long long)(S >> 10), (double)S * grid / 1e6, ms, passes, 2.0 * S * grid * passes / ms / 1e6);
    }
    return 0;
}
