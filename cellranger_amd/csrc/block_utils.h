// block_utils.h -- workgroup-level helpers (wave64 shuffles + a few LDS words).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// Exclusive prefix of `v` over the 256 threads of a workgroup (4 waves); *total_out = block sum.
// `lds` must hold >= 8 uint32.  Contains two barriers; every thread must call it.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *lds, uint32_t *total_out) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63u) lds[wave] = x;
    __syncthreads();
    uint32_t wpre = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t t = lds[w];
        if (w < wave) wpre += t;
        tot += t;
    }
    __syncthreads();  // lds may be reused by the caller's next call
    if (total_out) *total_out = tot;
    return wpre + x - v;
}

// Same for a workgroup of THREADS threads (a multiple of 64, at most 1024).  `lds` >= THREADS/64 uint32.
template <uint32_t THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *lds, uint32_t *total_out) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63u) lds[wave] = x;
    __syncthreads();
    uint32_t wpre = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < THREADS / 64u; w++) {
        const uint32_t t = lds[w];
        if (w < wave) wpre += t;
        tot += t;
    }
    __syncthreads();
    if (total_out) *total_out = tot;
    return wpre + x - v;
}

// Reserve `my_count` consecutive output slots for this thread with ONE global atomic per workgroup
// call (a single hot counter serialises at the memory side: one atomic per wave is ~300 k same-
// address atomics for 20 M reads).  `lds` >= 10 uint32 (8-byte aligned).  Every thread must call it.
__device__ __forceinline__ unsigned long long block_reserve_256(uint32_t my_count, unsigned long long *counter,
                                                                uint32_t *lds) {
    uint32_t total;
    const uint32_t pre = block_excl_scan_256(my_count, lds, &total);
    unsigned long long *base_s = reinterpret_cast<unsigned long long *>(lds + 8);
    if (threadIdx.x == 0) *base_s = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
    __syncthreads();
    const unsigned long long base = *base_s;
    __syncthreads();
    return base + pre;
}
