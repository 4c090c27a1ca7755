// umi_correct.h -- UMI correction kernel (correct_umis, tx_annotation/src/mark_dups.rs:19-59).
// Included by dedup.hip after KL / lowmask / segment_bounds / run_count are defined.
//
// For every distinct key (barcode, feature, library, UMI) with read count c: among the EXISTING keys of
// the same (barcode, feature, library) segment whose UMI is exactly one base away, pick the maximum by
// (count, UMI) -- the reference's loop over 3L candidates keeps the candidate when
// `count > best || (count == best && umi > best_umi)`, which is an order-independent arg-max -- and
// move to it when it beats (c, own UMI).  One step only, never transitive.
//
// The distinct keys are sorted, so a segment is a contiguous range.  A workgroup stages a tile of
// UC_TILE consecutive keys (+1 halo on each side) in LDS (lane-interleaved, so neighbouring lanes touch
// neighbouring LDS words), derives every key's segment bounds from per-64-key ballots of the
// segment-head flags, and searches out of LDS: all pairs for short
// segments, 3L binary searches for long ones.  Only segments that cross a tile edge go through global
// memory (galloping bounds + binary searches).
#pragma once

#define UC_ITEMS 8
#define UC_TILE (256 * UC_ITEMS)
#define UC_SMALL 32
#define UC_OPEN 0xFFFFu

__device__ __forceinline__ bool hd1(uint32_t a, uint32_t b) {
    const uint32_t x = a ^ b;
    const uint32_t y = (x | (x >> 1)) & 0x55555555u;
    return y != 0u && (y & (y - 1u)) == 0u;
}

// global-memory path for one key (segments that are not fully inside a tile)
__device__ uint32_t correct_one_global(const KL &kl, const uint64_t *__restrict__ ukey, const uint32_t *__restrict__ upos,
                                       uint64_t nd, uint64_t n_keys, uint64_t k, uint32_t my_cnt) {
    const uint64_t umi_mask = lowmask(kl.bits_umi);
    const uint64_t key = ukey[k];
    uint64_t s, e;
    segment_bounds(ukey, nd, k, kl.sh_lib, s, e);
    if (e - s <= 1) return NONE32;
    const uint32_t my_umi = (uint32_t)((key >> kl.sh_umi) & umi_mask);
    uint32_t best_cnt = my_cnt, best_umi = my_umi;
    uint64_t best_idx = k;
    if (e - s <= UC_SMALL) {
        for (uint64_t j = s; j < e; j++) {
            if (j == k) continue;
            const uint32_t u = (uint32_t)((ukey[j] >> kl.sh_umi) & umi_mask);
            if (!hd1(u, my_umi)) continue;
            const uint32_t c = run_count(upos, nd, n_keys, j);
            if (c > best_cnt || (c == best_cnt && u > best_umi)) {
                best_cnt = c;
                best_umi = u;
                best_idx = j;
            }
        }
    } else {
        const uint64_t pre = (key >> kl.sh_lib) << kl.bits_umi;
        for (uint32_t pos = 0; pos < kl.umi_len; pos++) {
            const uint32_t sh = 2u * (kl.umi_len - 1u - pos);
            const uint32_t orig = (my_umi >> sh) & 3u;
            for (uint32_t b = 0; b < 4; b++) {
                if (b == orig) continue;
                const uint32_t u = (my_umi & ~(3u << sh)) | (b << sh);
                const uint64_t want = pre | u;  // == ukey >> sh_umi of the probed key
                uint64_t lo = s, hi = e;
                while (lo < hi) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if ((ukey[mid] >> kl.sh_umi) < want) lo = mid + 1; else hi = mid;
                }
                if (lo < e && (ukey[lo] >> kl.sh_umi) == want) {
                    const uint32_t c = run_count(upos, nd, n_keys, lo);
                    if (c > best_cnt || (c == best_cnt && u > best_umi)) {
                        best_cnt = c;
                        best_umi = u;
                        best_idx = lo;
                    }
                }
            }
        }
    }
    return best_idx != k ? (uint32_t)best_idx : NONE32;
}

#define UC_BLOCKS (UC_TILE / 64)

__global__ __launch_bounds__(256) void k_correct_umis_tiled(const KL kl, const uint64_t *__restrict__ ukey,
                                                            const uint32_t *__restrict__ upos, uint64_t nd,
                                                            uint64_t n_keys, uint32_t *__restrict__ corr,
                                                            uint32_t *__restrict__ inc1, uint32_t *__restrict__ inc_all) {
    __shared__ uint64_t s_pre[UC_TILE + 2];  // segment id (key >> sh_lib) of positions -1 .. UC_TILE
    __shared__ uint32_t s_umi[UC_TILE];
    __shared__ uint32_t s_cnt[UC_TILE];
    __shared__ unsigned long long s_heads[UC_BLOCKS];  // bit l of entry b: position 64*b+l starts a segment
    __shared__ int s_carry_start[UC_BLOCKS];           // last segment start in blocks < b, or -1
    __shared__ int s_carry_end[UC_BLOCKS];             // first segment start in blocks > b (UC_TILE = halo), or INT_MAX
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint64_t umi_mask = lowmask(kl.bits_umi);
    const uint64_t n_tiles = (nd + UC_TILE - 1) / UC_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t t0 = tile * UC_TILE;
        const uint32_t tn = nd - t0 < UC_TILE ? (uint32_t)(nd - t0) : UC_TILE;
        // ---- stage (coalesced; lane l of round r holds position 256*r + tid) ----
#pragma unroll
        for (int r = 0; r < UC_ITEMS; r++) {
            const uint32_t p = (uint32_t)r * 256u + tid;
            if (p < tn) {
                const uint64_t key = ukey[t0 + p];
                s_pre[p + 1] = key >> kl.sh_lib;
                s_umi[p] = (uint32_t)((key >> kl.sh_umi) & umi_mask);
                const uint32_t end = t0 + p + 1 < nd ? upos[t0 + p + 1] : (uint32_t)n_keys;
                s_cnt[p] = end - upos[t0 + p];
            } else {
                s_pre[p + 1] = ~0ull;  // never equals a real segment id (a real one has < 63 bits)
            }
        }
        if (tid == 0) {
            s_pre[0] = t0 > 0 ? (ukey[t0 - 1] >> kl.sh_lib) : ~0ull;
            s_pre[UC_TILE + 1] = t0 + UC_TILE < nd ? (ukey[t0 + UC_TILE] >> kl.sh_lib) : ~0ull;
        }
        __syncthreads();
        // ---- segment-head bitmask of every 64-position block ----
#pragma unroll
        for (int r = 0; r < UC_ITEMS; r++) {
            const uint32_t p = (uint32_t)r * 256u + tid;
            const unsigned long long m = __ballot(s_pre[p + 1] != s_pre[p]);
            if (lane == 0) s_heads[p >> 6] = m;
        }
        __syncthreads();
        if (tid < UC_BLOCKS) {
            // serial prefix / suffix over 32 blocks (one thread each direction would do; all 32 threads
            // compute their own entry by walking, it is 32 steps at most)
            int cs = -1;
            for (int b = 0; b < (int)tid; b++) {
                const unsigned long long m = s_heads[b];
                if (m) cs = b * 64 + 63 - __clzll((long long)m);
            }
            s_carry_start[tid] = cs;
            int ce = 0x7FFFFFFF;
            if (s_pre[UC_TILE + 1] != s_pre[UC_TILE]) ce = UC_TILE;  // the halo starts a new segment
            for (int b = UC_BLOCKS - 1; b > (int)tid; b--) {
                const unsigned long long m = s_heads[b];
                if (m) ce = b * 64 + (__ffsll((long long)m) - 1);
            }
            s_carry_end[tid] = ce;
        }
        __syncthreads();
        // ---- one key per lane per round ----
#pragma unroll 1
        for (int r = 0; r < UC_ITEMS; r++) {
            const uint32_t p = (uint32_t)r * 256u + tid;
            if (p >= tn) continue;
            const uint32_t b = p >> 6;
            const unsigned long long heads = s_heads[b];
            const unsigned long long le = heads & ((2ull << lane) - 1ull);  // heads at or before p
            const int start = le ? (int)(b * 64u + 63u - (uint32_t)__clzll((long long)le)) : s_carry_start[b];
            const unsigned long long gt = lane < 63u ? (heads >> (lane + 1u)) : 0ull;  // heads after p
            const int end = gt ? (int)(p + (uint32_t)__ffsll((long long)gt)) : s_carry_end[b];
            const uint64_t k = t0 + p;
            const uint32_t my_umi = s_umi[p], my_cnt = s_cnt[p];
            const uint32_t lib = (uint32_t)(s_pre[p + 1] & lowmask(kl.bits_lib));
            uint32_t target = NONE32;
            if (!((kl.mux_mask >> lib) & 1u)) {  // UmiCorrection::Disable for Multiplexing Capture (aligner.rs:315-318)
                if (start < 0 || end > (int)UC_TILE) {
                    target = correct_one_global(kl, ukey, upos, nd, n_keys, k, my_cnt);
                } else if (end - start > 1) {
                    const uint32_t s = (uint32_t)start, e = (uint32_t)end;
                    uint32_t best_cnt = my_cnt, best_umi = my_umi, best_p = p;
                    if (e - s <= UC_SMALL) {
                        for (uint32_t q = s; q < e; q++) {
                            const uint32_t u = s_umi[q];
                            if (!hd1(u, my_umi)) continue;
                            const uint32_t c = s_cnt[q];
                            if (c > best_cnt || (c == best_cnt && u > best_umi)) {
                                best_cnt = c;
                                best_umi = u;
                                best_p = q;
                            }
                        }
                    } else {
                        for (uint32_t pos = 0; pos < kl.umi_len; pos++) {
                            const uint32_t sh = 2u * (kl.umi_len - 1u - pos);
                            const uint32_t orig = (my_umi >> sh) & 3u;
                            for (uint32_t bb = 0; bb < 4; bb++) {
                                if (bb == orig) continue;
                                const uint32_t u = (my_umi & ~(3u << sh)) | (bb << sh);
                                uint32_t lo = s, hi = e;
                                while (lo < hi) {
                                    const uint32_t mid = (lo + hi) >> 1;
                                    if (s_umi[mid] < u) lo = mid + 1; else hi = mid;
                                }
                                if (lo < e && s_umi[lo] == u) {
                                    const uint32_t c = s_cnt[lo];
                                    if (c > best_cnt || (c == best_cnt && u > best_umi)) {
                                        best_cnt = c;
                                        best_umi = u;
                                        best_p = lo;
                                    }
                                }
                            }
                        }
                    }
                    if (best_p != p) target = (uint32_t)(t0 + best_p);
                }
            }
            corr[k] = target;
            if (target != NONE32) {
                atomicAdd(&inc1[target], 1u);          // phase 1 moves one read (mark_dups.rs:228-232)
                atomicAdd(&inc_all[target], my_cnt);   // phases 1+2 move them all (:242-246)
            }
        }
        __syncthreads();
    }
}
