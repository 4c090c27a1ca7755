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
// TRAILING_BARRIER = false: the caller guarantees a barrier of its own before anybody writes `lds` again.
template <uint32_t THREADS, bool TRAILING_BARRIER = true>
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
    if (TRAILING_BARRIER) __syncthreads();
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

// Stable rank of a BITS-bit digit inside a wave (ballot multisplit): returns wc[d] before this call plus
// the number of lower lanes holding the same digit, and advances wc[d] by the size of the digit's group.
// wc = this wave's BITS-bit counter array in LDS.  Per digit bit: one v_bfe_i32 + one ballot + xnor/and on
// the two mask halves; ranks from v_mbcnt; the running count is a plain LDS read by every lane followed by
// a write from the lowest lane of each group (a wave runs in lockstep and its LDS operations complete in
// order, so no atomic or cross-lane shuffle is needed).  FULL: every lane holds a live element.
template <int BITS, bool FULL, typename C = uint32_t>
__device__ __forceinline__ uint32_t wave_multisplit_rank(uint32_t d, bool ok, C *__restrict__ wc) {
    uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;
    if (!FULL) {
        const unsigned long long live = __ballot(ok);
        lo = (uint32_t)live;
        hi = (uint32_t)(live >> 32);
    }
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        const int32_t nb = (int32_t)(d << (31 - b)) >> 31;  // 0 or -1 (v_bfe_i32)
        const unsigned long long bal = __ballot(nb != 0);
        lo &= ~((uint32_t)bal ^ (uint32_t)nb);          // lanes whose bit b equals mine
        hi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)nb);
    }
    const uint32_t below = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    const uint32_t cnt = (uint32_t)__popc(lo) + (uint32_t)__popc(hi);
    uint32_t prev = 0;
    if (FULL || ok) {
        prev = wc[d];
        if (below == 0) wc[d] = (C)(prev + cnt);
    }
    return prev + below;
}

// Decoupled look-back by one full wave (all 64 lanes must call it): `status` holds one word per tile, flag in the top two
// bits (0 = nothing yet, 1 = the tile's own count, 2 = inclusive prefix up to and including the tile), value in the low 62.
// Returns the sum of the counts of all tiles before `tile`.  64 predecessors are read per round trip -- a single lane
// walking back one word at a time falls behind as soon as a walk takes longer than the stagger between tiles, and then
// every walk gets long (measured: 60 us per 2048-key tile).
// Watchdog (abort_word nullable): tiles are handed out by tickets, so every predecessor is held by a workgroup that is running
// or done and the chain always moves on -- as long as the device schedules those workgroups.  Should that ever not hold (CU
// masking, a shared device), a waiter gives up after ~2^22 polls, raises *abort_word and returns WAVE_LOOKBACK_ABORTED; every
// other waiter sees the word within 4096 polls and leaves, too: no hung GPU, the host reports an error.
#define WAVE_LOOKBACK_ABORTED (~0ull)
__device__ __forceinline__ unsigned long long wave_lookback(const unsigned long long *status, uint64_t tile,
                                                            uint32_t *abort_word = nullptr) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long excl = 0;
    uint64_t end = tile;  // predecessors [.., end) are still to be added
    uint32_t polls = 0;
    for (;;) {
        const bool valid = end > lane;
        unsigned long long sv = 2ull << 62;  // before tile 0: an inclusive prefix of 0
        if (valid) sv = __hip_atomic_load(&status[end - 1 - lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t flag = (uint32_t)(sv >> 62);
        const unsigned long long inc = __ballot(flag == 2u), pending = __ballot(flag == 0u);
        const uint32_t first = inc ? (uint32_t)(__ffsll((long long)inc) - 1) : 64u;
        const unsigned long long need = first >= 63u ? ~0ull : ((2ull << first) - 1ull);  // lanes 0 .. first
        if (pending & need) {
            __builtin_amdgcn_s_sleep(1);
            if (abort_word && (++polls & 0xFFFu) == 0u) {  // uniform: every lane counts the same polls
                if (polls >= (1u << 22) && lane == 0) atomicExch(abort_word, 1u);
                if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return WAVE_LOOKBACK_ABORTED;
            }
            continue;  // somebody in front of the first inclusive prefix has not published yet: same window again
        }
        unsigned long long v = lane <= first ? (sv & ((1ull << 62) - 1ull)) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        excl += v;
        if (first < 64u) return excl;
        end -= 64;
    }
}
