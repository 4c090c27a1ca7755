// umi_correct.h -- UMI correction kernels (correct_umis, tx_annotation/src/mark_dups.rs:19-59).
// Included by dedup.hip after KL / lowmask / segment_bounds / run_count are defined.
//
// For every distinct key (barcode, feature, library, UMI) with read count c: among the EXISTING keys of
// the same (barcode, feature, library) segment whose UMI is exactly one base away, pick the maximum by
// (count, UMI) -- the reference's loop over 3L candidates keeps the candidate when
// `count > best || (count == best && umi > best_umi)`, which is an order-independent arg-max -- and
// move to it when it beats (c, own UMI).  One step only, never transitive.
//
// The distinct keys are sorted, so a segment is a contiguous range.  Two kernels cover every key once:
//
//  k_correct_umis_tiled  A workgroup stages a tile of UC_TILE consecutive keys in LDS (lane-interleaved)
//      and derives every key's segment bounds from per-64-key ballots of the segment-head flags.
//      Segments that lie completely inside the tile are finished here: short ones (<= UC_SMALL keys) by
//      all pairs, long ones by the reference's 3L probes against an LDS hash set of the tile's
//      long-segment keys.  Keys of segments that cross a tile edge are left alone.
//  k_correct_umis_edges  One workgroup per tile boundary that falls strictly inside a segment (the
//      first such boundary owns the segment): the whole segment is loaded into LDS with a hash set and
//      finished the same way; segments larger than the LDS budget fall back to binary searches in
//      global memory.
//
// 3L probes into a hash set cost ~1.3 LDS reads each; the same probes as binary searches cost
// log2(m) dependent reads each and were the dominant cost of the whole count stage.
#pragma once

#define UC_ITEMS 8
#define UC_TILE (256 * UC_ITEMS)
#define UC_BLOCKS (UC_TILE / 64)
#define UC_SMALL 192  // all pairs cost ~8 ops per member, a hashed 3L-probe search ~1500: break-even near 200
#define UC_OPEN 0xFFFFu
#define UC_BUCKETS 1024u  // tile hash set: 8-slot buckets (32 B); at most UC_TILE keys => load <= 0.25
#define UC_POSBITS 11u    // slot = (fingerprint << POSBITS) | position
#define UC_EMPTY 0xFFFFFFFFu
#define UC_NOCORR 0x80000000u  // flag inside the staged count: UMI correction disabled for the key's library

#define UE_THREADS 512
#define UE_CAP 4096u      // keys of one edge segment held in LDS
#define UE_BUCKETS 2048u  // its hash set: load <= 0.25
#define UE_POSBITS 12u

__device__ __forceinline__ bool hd1(uint32_t a, uint32_t b) {
    const uint32_t x = a ^ b;
    const uint32_t y = (x | (x >> 1)) & 0x55555555u;
    return y != 0u && (y & (y - 1u)) == 0u;
}

// global-memory path for one key of the segment [s, e)
__device__ uint32_t correct_one_global(const KL &kl, const uint64_t *__restrict__ ukey, const uint32_t *__restrict__ upos,
                                       uint64_t nd, uint64_t n_keys, uint64_t k, uint32_t my_cnt, uint64_t s, uint64_t e) {
    const uint64_t umi_mask = lowmask(kl.bits_umi);
    const uint64_t key = ukey[k];
    if (e - s <= 1) return NONE32;
    const uint32_t my_umi = (uint32_t)((key >> kl.sh_umi) & umi_mask);
    uint32_t best_cnt = my_cnt, best_umi = my_umi;
    uint64_t best_idx = k;
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
    return best_idx != k ? (uint32_t)best_idx : NONE32;
}

__device__ __forceinline__ uint32_t uc_hash(uint32_t seg_start, uint32_t umi, uint32_t mask) {
    uint32_t h = umi * 0x9E3779B1u ^ (seg_start * 0x85EBCA6Bu);
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h & mask;
}

// ---- bucketised LDS hash set ------------------------------------------------------------------------
// A probe chain is the enemy here: all 64 lanes of a wave wait for the lane with the longest chain, at
// each of the 3L probes.  Buckets of 8 four-byte slots (32 B, two ds_read_b128) at load <= 0.25 answer a
// probe with ONE wide read: slots of a bucket fill front to back, so a bucket with a free slot proves
// absence and only a full bucket without a match (P ~ 1e-3) continues to the next bucket.
// slot = (fingerprint << POSBITS) | position; a fingerprint hit is verified against the staged UMI / tag.
template <uint32_t BUCKETS>
__device__ __forceinline__ uint32_t uc_bucket(uint32_t tag, uint32_t umi, uint32_t &fp_out, uint32_t posbits) {
    // one multiply: the top bits pick the bucket, the low bits are the fingerprint
    const uint32_t x = (umi ^ (tag * 0x85EBCA6Bu)) * 0x9E3779B1u;
    uint32_t f = (x ^ (x >> 15)) & ((1u << (32u - posbits)) - 1u);
    if (f == (1u << (32u - posbits)) - 1u) f = 0;  // all ones is reserved for EMPTY
    fp_out = f;
    return (x >> 20) & (BUCKETS - 1u);
}
template <uint32_t BUCKETS, uint32_t POSBITS>
__device__ __forceinline__ void uc_insert(uint32_t *s_hash, uint32_t tag, uint32_t umi, uint32_t pos) {
    uint32_t fp;
    uint32_t b = uc_bucket<BUCKETS>(tag, umi, fp, POSBITS);
    const uint32_t v = (fp << POSBITS) | pos;
    for (;;) {
        for (uint32_t j = 0; j < 8; j++)
            if (atomicCAS(&s_hash[b * 8u + j], UC_EMPTY, v) == UC_EMPTY) return;
        b = (b + 1u) & (BUCKETS - 1u);
    }
}
struct UcBucket {
    uint4 lo, hi;
};
__device__ __forceinline__ UcBucket uc_load(const uint32_t *s_hash, uint32_t b) {
    const uint4 *p = reinterpret_cast<const uint4 *>(s_hash + b * 8u);
    UcBucket r;
    r.lo = p[0];
    r.hi = p[1];
    return r;
}
// position of (tag, umi) or 0xFFFFFFFF; `bk` is the already loaded bucket b
template <uint32_t BUCKETS, uint32_t POSBITS>
__device__ __forceinline__ uint32_t uc_resolve(const uint32_t *s_hash, const uint32_t *s_umi, const uint16_t *s_tag,
                                               UcBucket bk, uint32_t b, uint32_t fp, uint32_t tag, uint32_t umi) {
    for (;;) {
        const uint32_t w[8] = {bk.lo.x, bk.lo.y, bk.lo.z, bk.lo.w, bk.hi.x, bk.hi.y, bk.hi.z, bk.hi.w};
        bool has_free = false;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            has_free |= w[j] == UC_EMPTY;
            if ((w[j] >> POSBITS) == fp && w[j] != UC_EMPTY) {
                const uint32_t q = w[j] & ((1u << POSBITS) - 1u);
                if (s_umi[q] == umi && (s_tag == nullptr || s_tag[q] == (uint16_t)tag)) return q;
            }
        }
        if (has_free) return 0xFFFFFFFFu;
        b = (b + 1u) & (BUCKETS - 1u);
        bk = uc_load(s_hash, b);
    }
}

// best Hamming-1 neighbour of (my_umi, my_cnt) among the keys of segment [s, e) staged in LDS.
// `seg_tag` distinguishes segments inside one hash set.  Returns the LDS position or `self`.
template <uint32_t BUCKETS, uint32_t POSBITS>
__device__ __forceinline__ uint32_t best_neighbour_lds(const uint32_t *s_umi, const uint32_t *s_cnt, const uint32_t *s_hash,
                                                       const uint16_t *s_tag, uint32_t s, uint32_t e, uint32_t seg_tag,
                                                       uint32_t self, uint32_t my_umi, uint32_t my_cnt, uint32_t umi_len) {
    uint32_t best_cnt = my_cnt, best_umi = my_umi, best_p = self;
    if (e - s <= UC_SMALL) {
        for (uint32_t q = s; q < e; q++) {
            const uint32_t u = s_umi[q];
            if (!hd1(u, my_umi)) continue;
            const uint32_t c = s_cnt[q] & ~UC_NOCORR;
            if (c > best_cnt || (c == best_cnt && u > best_umi)) {
                best_cnt = c;
                best_umi = u;
                best_p = q;
            }
        }
    } else {
        for (uint32_t pos = 0; pos < umi_len; pos++) {
            const uint32_t sh = 2u * (umi_len - 1u - pos);
            const uint32_t orig = (my_umi >> sh) & 3u;
            // the three substitutions of this position: their bucket reads are independent
            uint32_t u[3], b[3], fp[3];
            UcBucket bk[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const uint32_t bb = (orig + 1u + (uint32_t)i) & 3u;
                u[i] = (my_umi & ~(3u << sh)) | (bb << sh);
                b[i] = uc_bucket<BUCKETS>(seg_tag, u[i], fp[i], POSBITS);
            }
#pragma unroll
            for (int i = 0; i < 3; i++) bk[i] = uc_load(s_hash, b[i]);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const uint32_t q = uc_resolve<BUCKETS, POSBITS>(s_hash, s_umi, s_tag, bk[i], b[i], fp[i], seg_tag, u[i]);
                if (q != 0xFFFFFFFFu) {
                    const uint32_t c = s_cnt[q] & ~UC_NOCORR;
                    if (c > best_cnt || (c == best_cnt && u[i] > best_umi)) {
                        best_cnt = c;
                        best_umi = u[i];
                        best_p = q;
                    }
                }
            }
        }
    }
    return best_p;
}

__global__ __launch_bounds__(256) void k_correct_umis_tiled(const KL kl, const uint64_t *__restrict__ ukey,
                                                            const uint32_t *__restrict__ upos, uint64_t nd,
                                                            uint64_t n_keys, uint32_t *__restrict__ corr,
                                                            uint32_t *__restrict__ inc1, uint32_t *__restrict__ inc_all,
                                                            int ablate) {
    __shared__ uint32_t s_umi[UC_TILE];
    __shared__ uint32_t s_cnt[UC_TILE];    // read count | UC_NOCORR
    __shared__ uint16_t s_start[UC_TILE];  // segment start inside the tile, UC_OPEN = not fully inside the tile
    __shared__ uint16_t s_end[UC_TILE];    // exclusive end
    __shared__ __attribute__((aligned(16))) uint32_t s_hash[UC_BUCKETS * 8];  // bucketised hash set
    __shared__ unsigned long long s_heads[UC_BLOCKS];  // bit l of entry b: position 64*b+l starts a segment
    __shared__ int s_carry_start[UC_BLOCKS];           // last segment start in blocks < b, or -1
    __shared__ int s_carry_end[UC_BLOCKS];             // first segment start in blocks > b, or INT_MAX
    __shared__ uint32_t s_halo_head;                   // position UC_TILE starts a new segment
    __shared__ uint32_t s_any_long;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint64_t umi_mask = lowmask(kl.bits_umi);
    const uint64_t n_tiles = (nd + UC_TILE - 1) / UC_TILE;
    for (uint32_t h = tid; h < UC_BUCKETS * 8; h += 256) s_hash[h] = UC_EMPTY;
    if (tid == 0) s_any_long = 0;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t t0 = tile * UC_TILE;
        const uint32_t tn = nd - t0 < UC_TILE ? (uint32_t)(nd - t0) : UC_TILE;
        // ---- stage (coalesced; lane l of round r holds position 256*r + tid) + head flags ----
#pragma unroll
        for (int r = 0; r < UC_ITEMS; r++) {
            const uint32_t p = (uint32_t)r * 256u + tid;
            bool head = false;
            if (p < tn) {
                const uint64_t key = ukey[t0 + p];
                const uint64_t pre = key >> kl.sh_lib;
                // the previous key is the neighbouring lane's load (same cache line)
                head = (t0 + p == 0) || (ukey[t0 + p - 1] >> kl.sh_lib) != pre;
                s_umi[p] = (uint32_t)((key >> kl.sh_umi) & umi_mask);
                const uint32_t end = t0 + p + 1 < nd ? upos[t0 + p + 1] : (uint32_t)n_keys;
                const uint32_t lib = (uint32_t)(pre & lowmask(kl.bits_lib));
                // UmiCorrection::Disable for Multiplexing Capture (aligner.rs:315-318)
                s_cnt[p] = (end - upos[t0 + p]) | (((kl.mux_mask >> lib) & 1u) ? UC_NOCORR : 0u);
            } else {
                head = p == tn;  // the padding behind the last key closes the last segment
            }
            const unsigned long long m = __ballot(head);
            if (lane == 0) s_heads[p >> 6] = m;
        }
        if (tid == 0) {
            bool hh = true;  // the end of the array closes the segment
            if (t0 + UC_TILE < nd) hh = (ukey[t0 + UC_TILE] >> kl.sh_lib) != (ukey[t0 + UC_TILE - 1] >> kl.sh_lib);
            s_halo_head = hh ? 1u : 0u;
        }
        __syncthreads();
        if (tid < UC_BLOCKS) {
            int cs = -1;
            for (int b = 0; b < (int)tid; b++) {
                const unsigned long long m = s_heads[b];
                if (m) cs = b * 64 + 63 - __clzll((long long)m);
            }
            s_carry_start[tid] = cs;
            int ce = s_halo_head ? (int)UC_TILE : 0x7FFFFFFF;
            for (int b = UC_BLOCKS - 1; b > (int)tid; b--) {
                const unsigned long long m = s_heads[b];
                if (m) ce = b * 64 + (__ffsll((long long)m) - 1);
            }
            s_carry_end[tid] = ce;
        }
        __syncthreads();
        // ---- segment bounds of every key; long in-tile segments enter the hash set ----
        bool inserted = false;
#pragma unroll
        for (int r = 0; r < UC_ITEMS; r++) {
            const uint32_t p = (uint32_t)r * 256u + tid;
            if (p >= tn) continue;
            const uint32_t b = p >> 6;
            const unsigned long long heads = s_heads[b];
            const unsigned long long le = heads & ((2ull << lane) - 1ull);  // heads at or before p
            const int start = le ? (int)(b * 64u + 63u - (uint32_t)__clzll((long long)le)) : s_carry_start[b];
            const unsigned long long gt = lane < 63u ? (heads >> (lane + 1u)) : 0ull;  // heads after p
            const int end = gt ? (int)(p + (uint32_t)__ffsll((long long)gt)) : s_carry_end[b];
            const bool closed = start >= 0 && end <= (int)UC_TILE;
            s_start[p] = closed ? (uint16_t)start : (uint16_t)UC_OPEN;
            s_end[p] = closed ? (uint16_t)end : (uint16_t)UC_OPEN;
            if (closed && end - start > UC_SMALL && !(ablate & 16)) {
                uc_insert<UC_BUCKETS, UC_POSBITS>(s_hash, (uint32_t)start, s_umi[p], p);
                inserted = true;
            }
        }
        if (inserted) s_any_long = 1;
        __syncthreads();
        // ---- one key per lane per round; segments that cross a tile edge belong to k_correct_umis_edges ----
#pragma unroll 1
        for (int r = 0; r < UC_ITEMS; r++) {
            const uint32_t p = (uint32_t)r * 256u + tid;
            if (p >= tn) continue;
            const uint32_t s = s_start[p], e = s_end[p];
            if (s == UC_OPEN) continue;
            const uint64_t k = t0 + p;
            const uint32_t cw = s_cnt[p];
            const uint32_t my_cnt = cw & ~UC_NOCORR;
            uint32_t target = NONE32;
            if (!(cw & UC_NOCORR) && e - s > 1 && !(ablate & 1) && !((ablate & 4) && e - s > UC_SMALL) &&
                !((ablate & 8) && e - s <= UC_SMALL)) {
                const uint32_t bp = best_neighbour_lds<UC_BUCKETS, UC_POSBITS>(s_umi, s_cnt, s_hash, s_start, s, e, s, p, s_umi[p], my_cnt, kl.umi_len);
                if (bp != p) target = (uint32_t)(t0 + bp);
            }
            corr[k] = target;
            if (target != NONE32 && !(ablate & 2)) {
                atomicAdd(&inc1[target], 1u);          // phase 1 moves one read (mark_dups.rs:228-232)
                atomicAdd(&inc_all[target], my_cnt);   // phases 1+2 move them all (:242-246)
            }
        }
        __syncthreads();
        if (s_any_long) {
            for (uint32_t h = tid; h < UC_BUCKETS * 8; h += 256) s_hash[h] = UC_EMPTY;
            __syncthreads();
            if (tid == 0) s_any_long = 0;
        }
    }
}

// Segments that contain a tile boundary strictly inside.  Boundary t (position t*UC_TILE) owns the
// segment when the previous boundary is not inside the same segment.
__global__ __launch_bounds__(UE_THREADS) void k_correct_umis_edges(const KL kl, const uint64_t *__restrict__ ukey,
                                                                   const uint32_t *__restrict__ upos, uint64_t nd,
                                                                   uint64_t n_keys, uint32_t *__restrict__ corr,
                                                                   uint32_t *__restrict__ inc1,
                                                                   uint32_t *__restrict__ inc_all) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *s_umi = smem;                  // UE_CAP
    uint32_t *s_cnt = smem + UE_CAP;         // UE_CAP
    uint32_t *s_hash = smem + 2 * UE_CAP;    // UE_BUCKETS * 8 slots (16-byte aligned: UE_CAP is a multiple of 4)
    __shared__ unsigned long long s_bounds[2];
    const uint32_t tid = threadIdx.x;
    const uint64_t umi_mask = lowmask(kl.bits_umi);
    const uint64_t n_tiles = (nd + UC_TILE - 1) / UC_TILE;
    for (uint64_t t = (uint64_t)blockIdx.x + 1; t < n_tiles; t += gridDim.x) {
        const uint64_t x = t * UC_TILE;
        const uint64_t pre = ukey[x] >> kl.sh_lib;
        if ((ukey[x - 1] >> kl.sh_lib) != pre) continue;  // the boundary is a segment head: nothing crosses it
        if (tid == 0) {
            uint64_t s, e;
            segment_bounds(ukey, nd, x, kl.sh_lib, s, e);
            s_bounds[0] = s;
            s_bounds[1] = e;
        }
        __syncthreads();
        const uint64_t s = s_bounds[0], e = s_bounds[1];
        __syncthreads();
        if (x - UC_TILE > s) continue;  // the previous boundary is inside the same segment and owns it
        const uint64_t m = e - s;
        const uint32_t lib = (uint32_t)(pre & lowmask(kl.bits_lib));
        const bool nocorr = (kl.mux_mask >> lib) & 1u;  // UmiCorrection::Disable (aligner.rs:315-318)
        if (nocorr) {
            for (uint64_t k = s + tid; k < e; k += UE_THREADS) corr[k] = NONE32;
            continue;
        }
        if (m <= UE_CAP) {
            const uint32_t mm = (uint32_t)m;
            const bool use_hash = mm > UC_SMALL;
            if (use_hash)
                for (uint32_t h = tid; h < UE_BUCKETS * 8; h += UE_THREADS) s_hash[h] = UC_EMPTY;
            for (uint32_t p = tid; p < mm; p += UE_THREADS) {
                const uint64_t k = s + p;
                s_umi[p] = (uint32_t)((ukey[k] >> kl.sh_umi) & umi_mask);
                s_cnt[p] = run_count(upos, nd, n_keys, k);
            }
            __syncthreads();
            if (use_hash) {
                for (uint32_t p = tid; p < mm; p += UE_THREADS) uc_insert<UE_BUCKETS, UE_POSBITS>(s_hash, 0u, s_umi[p], p);
                __syncthreads();
            }
            for (uint32_t p = tid; p < mm; p += UE_THREADS) {
                const uint32_t my_cnt = s_cnt[p];
                const uint32_t bp = best_neighbour_lds<UE_BUCKETS, UE_POSBITS>(s_umi, s_cnt, s_hash, nullptr, 0u, mm, 0u, p, s_umi[p], my_cnt, kl.umi_len);
                const uint32_t target = bp != p ? (uint32_t)(s + bp) : NONE32;
                corr[s + p] = target;
                if (target != NONE32) {
                    atomicAdd(&inc1[target], 1u);
                    atomicAdd(&inc_all[target], my_cnt);
                }
            }
            __syncthreads();
        } else {
            // larger than the LDS budget: the reference's 3L probes as binary searches in global memory
            for (uint64_t k = s + tid; k < e; k += UE_THREADS) {
                const uint32_t my_cnt = run_count(upos, nd, n_keys, k);
                const uint32_t target = correct_one_global(kl, ukey, upos, nd, n_keys, k, my_cnt, s, e);
                corr[k] = target;
                if (target != NONE32) {
                    atomicAdd(&inc1[target], 1u);
                    atomicAdd(&inc_all[target], my_cnt);
                }
            }
        }
    }
}
