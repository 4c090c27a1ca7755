// umi_correct.h -- UMI correction kernels (correct_umis, tx_annotation/src/mark_dups.rs:19-59).
// Included by dedup.hip after KL / lowmask / segment_bounds / run_count and the per-key state helpers (st_*) are defined.
//
// For every distinct key (barcode, feature, library, UMI) with read count c: among the EXISTING keys of
// the same (barcode, feature, library) segment whose UMI is exactly one base away, pick the maximum by
// (count, UMI) -- the reference's loop over 3L candidates keeps the candidate when
// `count > best || (count == best && umi > best_umi)`, which is an order-independent arg-max -- and
// move to it when it beats (c, own UMI).  One step only, never transitive.
//
// Pigeonhole: split the UMI into a high half (first L - L/2 bases) and a low half (last L/2 bases).
// A Hamming-1 neighbour either
//   (1) shares my high half -> the distinct keys are sorted by UMI, so it sits in my own contiguous run of
//       equal high halves: scan outwards from my position (the run has ~m/4^(L-L/2) members), or
//   (2) shares my low half  -> ONE lookup in a hash set keyed by (segment, low half) lists every member
//       with that low half; check their high halves.
// That is two short scans per key instead of the reference's 3L = 36 hash probes; the 36 probes (as LDS
// hash probes, and before that as binary searches) were the dominant cost of the count stage.
//
//  k_correct_umis_tiled  A workgroup stages a tile of UC_TILE consecutive keys in LDS (lane-interleaved)
//      and derives every key's segment bounds from per-64-key ballots of the segment-head flags.  Segments
//      completely inside the tile are finished here: <= UC_SMALL members by all pairs, longer ones by the
//      pigeonhole search.  Keys of segments that cross a tile edge are left to
//      The kernel also records, per tile, the position of its first and last segment head.
//  k_correct_umis_edges  one workgroup per tile boundary that falls strictly inside a segment (the first
//      such boundary owns the segment); the segment's bounds come straight from the per-tile head
//      positions, no search.  Two instantiations share the work by segment size: up to UES_CAP keys
//      (256 threads, 24 KB of LDS, six workgroups per CU -- nearly all edge segments are this small and the
//      kernel is latency bound, so residency is what counts) and above (512 threads, 96 KB).  Segments up
//      to UE_CAP keys are staged in LDS whole; larger ones are processed in chunks of UE_CAP table entries
//      against which every key of the segment is probed.
#pragma once

// 1024-key tiles: 28 KB of LDS, five workgroups per CU.  The kernel waits at its barriers most of the time
// (SQ_WAIT_ANY 73 % of the wave cycles), so residency beats tile size: 2048-key tiles (57 KB, two per CU) were
// 12 % slower on the whole dedup stage even though fewer segments cross a tile edge.
#ifndef UC_ITEMS
#define UC_ITEMS 4
#endif
#define UC_TILE (256 * UC_ITEMS)
#ifndef UC_MIN_WG
#define UC_MIN_WG 1   // second launch-bounds argument: workgroups per CU the register allocation has to allow
#endif
#define UC_BLOCKS (UC_TILE / 64)
#ifndef UC_SMALL
#define UC_SMALL 32
#endif
#define UC_OPEN 0xFFFFu
#ifndef UC_BUCKETS
#define UC_BUCKETS 512u  // tile hash set: 8-slot buckets (32 B); at most UC_TILE entries => load <= 0.25
#endif
#define UC_POSBITS 11u    // slot = (fingerprint << POSBITS) | position
#define UC_EMPTY 0xFFFFFFFFu
#define UC_NOCORR 0x80000000u  // flag inside the staged count: UMI correction disabled for the key's library

#define UE_THREADS 512
#define UE_CAP 4096u      // table entries of one edge-segment chunk held in LDS
#define UE_BUCKETS 1024u  // load <= 0.5: 64 KB of LDS in all, two workgroups per CU
#define UE_POSBITS 12u
#define UES_THREADS 256
#define UES_CAP 1024u     // the small-segment instantiation
#define UES_BUCKETS 512u
#define UES_POSBITS 10u
#define UC_NOHEAD 0xFFFFFFFFu

struct GiantItem {
    uint32_t s, e, c0;  // segment [s, e) and the first key of this item's chunk of UE_CAP keys
};

template <bool SMALL>
struct EdgeCfg;
template <>
struct EdgeCfg<true> {
    static constexpr uint32_t THREADS = UES_THREADS, CAP = UES_CAP, BUCKETS = UES_BUCKETS, POSBITS = UES_POSBITS;
};
template <>
struct EdgeCfg<false> {
    static constexpr uint32_t THREADS = UE_THREADS, CAP = UE_CAP, BUCKETS = UE_BUCKETS, POSBITS = UE_POSBITS;
};

__device__ __forceinline__ bool hd1(uint32_t a, uint32_t b) {
    const uint32_t x = a ^ b;
    const uint32_t y = (x | (x >> 1)) & 0x55555555u;
    return y != 0u && (y & (y - 1u)) == 0u;
}

struct UmiSplit {
    uint32_t lo_bits, lo_mask;
};
__device__ __forceinline__ UmiSplit umi_split(uint32_t umi_len) {
    UmiSplit s;
    s.lo_bits = 2u * (umi_len / 2u);
    s.lo_mask = s.lo_bits >= 32u ? 0xFFFFFFFFu : ((1u << s.lo_bits) - 1u);
    return s;
}

// ---- bucketised LDS hash set keyed by (segment tag, low half) --------------------------------------------
// Buckets of 8 four-byte slots (32 B, two ds_read_b128), slot = (fingerprint << POSBITS) | position.  Slots
// of a bucket fill front to back, so a bucket with a free slot ends the search; only a full bucket
// continues to the next one.  Several members may share one (tag, low half): a lookup visits all of them.
template <uint32_t BUCKETS>
__device__ __forceinline__ uint32_t uc_bucket(uint32_t tag, uint32_t lo, uint32_t &fp_out, uint32_t posbits) {
    const uint32_t x = (lo ^ (tag * 0x85EBCA6Bu)) * 0x9E3779B1u;
    uint32_t f = (x ^ (x >> 15)) & ((1u << (32u - posbits)) - 1u);
    if (f == (1u << (32u - posbits)) - 1u) f = 0;  // all ones is reserved for EMPTY
    fp_out = f;
    return (x >> 20) & (BUCKETS - 1u);
}
template <uint32_t BUCKETS, uint32_t POSBITS>
__device__ __forceinline__ void uc_insert(uint32_t *s_hash, uint32_t tag, uint32_t lo, uint32_t pos) {
    uint32_t fp;
    uint32_t b = uc_bucket<BUCKETS>(tag, lo, fp, POSBITS);
    const uint32_t v = (fp << POSBITS) | pos;
    for (;;) {
        for (uint32_t j = 0; j < 8; j++)
            if (atomicCAS(&s_hash[b * 8u + j], UC_EMPTY, v) == UC_EMPTY) return;
        b = (b + 1u) & (BUCKETS - 1u);
    }
}
// f(position) for every table entry whose fingerprint matches (tag, lo); the caller verifies exactly
template <uint32_t BUCKETS, uint32_t POSBITS, typename F>
__device__ __forceinline__ void uc_for_each(const uint32_t *s_hash, uint32_t tag, uint32_t lo, F f) {
    uint32_t fp;
    uint32_t b = uc_bucket<BUCKETS>(tag, lo, fp, POSBITS);
    for (;;) {
        const uint4 *p = reinterpret_cast<const uint4 *>(s_hash + b * 8u);
        const uint4 x = p[0], y = p[1];
        const uint32_t w[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
        bool has_free = false;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            has_free |= w[j] == UC_EMPTY;
            if (w[j] != UC_EMPTY && (w[j] >> POSBITS) == fp) f(w[j] & ((1u << POSBITS) - 1u));
        }
        if (has_free) return;
        b = (b + 1u) & (BUCKETS - 1u);
    }
}

struct Best {
    uint32_t cnt, umi, pos;
    __device__ __forceinline__ void offer(uint32_t c, uint32_t u, uint32_t p) {
        if (c > cnt || (c == cnt && u > umi)) {
            cnt = c;
            umi = u;
            pos = p;
        }
    }
};

// best Hamming-1 neighbour of the key at LDS position `self` among the members of segment [s, e) staged in
// LDS (s_umi sorted ascending inside the segment).  Returns the LDS position or `self`.
template <uint32_t BUCKETS, uint32_t POSBITS>
__device__ __forceinline__ uint32_t best_neighbour_lds(const uint32_t *s_umi, const uint32_t *s_cnt, const uint32_t *s_hash,
                                                       const uint16_t *s_tag, uint32_t s, uint32_t e, uint32_t seg_tag,
                                                       uint32_t self, uint32_t my_umi, uint32_t my_cnt, UmiSplit sp) {
    Best best{my_cnt, my_umi, self};
    if (e - s <= UC_SMALL) {
        for (uint32_t q = s; q < e; q++) {
            const uint32_t u = s_umi[q];
            if (hd1(u, my_umi)) best.offer(s_cnt[q] & ~UC_NOCORR, u, q);
        }
        return best.pos;
    }
    const uint32_t my_hi = my_umi >> sp.lo_bits, my_lo = my_umi & sp.lo_mask;
    // (1) same high half: my own run in the sorted segment
    for (uint32_t q = self; q > s;) {
        --q;
        const uint32_t u = s_umi[q];
        if ((u >> sp.lo_bits) != my_hi) break;
        if (hd1(u & sp.lo_mask, my_lo)) best.offer(s_cnt[q] & ~UC_NOCORR, u, q);
    }
    for (uint32_t q = self + 1; q < e; q++) {
        const uint32_t u = s_umi[q];
        if ((u >> sp.lo_bits) != my_hi) break;
        if (hd1(u & sp.lo_mask, my_lo)) best.offer(s_cnt[q] & ~UC_NOCORR, u, q);
    }
    // (2) same low half: every member listed under (segment, low half)
    uc_for_each<BUCKETS, POSBITS>(s_hash, seg_tag, my_lo, [&](uint32_t q) {
        const uint32_t u = s_umi[q];
        if ((u & sp.lo_mask) != my_lo) return;                      // fingerprint collision
        if (s_tag != nullptr && s_tag[q] != (uint16_t)seg_tag) return;  // another segment of the tile
        if (hd1(u >> sp.lo_bits, my_hi)) best.offer(s_cnt[q] & ~UC_NOCORR, u, q);
    });
    return best.pos;
}

__global__ __launch_bounds__(256, UC_MIN_WG) void k_correct_umis_tiled(const KL kl, const uint64_t *__restrict__ ukey,
                                                            const uint32_t *__restrict__ upos, uint64_t nd,
                                                            uint64_t n_keys, uint32_t *__restrict__ tile_first,
                                                            uint32_t *__restrict__ tile_last, uint32_t *__restrict__ corr,
                                                            uint16_t *__restrict__ st, uint32_t *__restrict__ inc_all,
                                                            uint32_t *__restrict__ minidx) {
    __shared__ uint32_t s_umi[UC_TILE];
    __shared__ uint32_t s_cnt[UC_TILE];    // read count | UC_NOCORR
    __shared__ uint16_t s_start[UC_TILE];  // segment start inside the tile, UC_OPEN = not fully inside the tile
    __shared__ __attribute__((aligned(16))) uint32_t s_hash[UC_BUCKETS * 8];
    __shared__ unsigned long long s_heads[UC_BLOCKS];  // bit l of entry b: position 64*b+l starts a segment
    __shared__ int s_carry_start[UC_BLOCKS];           // last segment start in blocks < b, or -1
    __shared__ int s_carry_end[UC_BLOCKS];             // first segment start in blocks > b, or INT_MAX
    __shared__ uint32_t s_halo_head;                   // position UC_TILE starts a new segment
    __shared__ uint32_t s_any_long;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint64_t umi_mask = lowmask(kl.bits_umi);
    const UmiSplit sp = umi_split(kl.umi_len);
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
                const uint32_t lib = (uint32_t)((pre >> kl.bits_ulen) & lowmask(kl.bits_lib));
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
        if (tid < 64) {
            // one wave, lane b = block b of 64 keys: the last segment head before each block and the first one behind it
            // (prefix maximum / suffix minimum over the blocks by shuffles -- sixteen lanes looping over sixteen masks
            // held every tile's other 240 threads at the barrier for a couple of microseconds), and the tile's first / last
            // head for k_correct_umis_edges
            static_assert(UC_BLOCKS <= 64, "one lane per block");
            const unsigned long long m = tid < UC_BLOCKS ? s_heads[tid] : 0ull;
            const int last = m ? (int)(tid * 64u + 63u - (uint32_t)__clzll((long long)m)) : -1;
            const int first = m ? (int)(tid * 64u + (uint32_t)__ffsll((long long)m) - 1u) : 0x7FFFFFFF;
            int pm = last, sm = first;
#pragma unroll
            for (int d = 1; d < UC_BLOCKS; d <<= 1) {
                const int y = __shfl_up(pm, d), z = __shfl_down(sm, d);
                if ((int)tid >= d) pm = pm > y ? pm : y;
                if ((int)tid + d < (int)UC_BLOCKS) sm = sm < z ? sm : z;
            }
            int cs = __shfl_up(pm, 1), ce = __shfl_down(sm, 1);
            if (tid == 0) cs = -1;
            const int behind = s_halo_head ? (int)UC_TILE : 0x7FFFFFFF;
            if (tid + 1 >= UC_BLOCKS) ce = 0x7FFFFFFF;
            ce = ce < behind ? ce : behind;
            if (tid < UC_BLOCKS) {
                s_carry_start[tid] = cs;
                s_carry_end[tid] = ce;
            }
            // heads at positions < tn only (the padding head at tn closes the last segment but is no key)
            unsigned long long mk = m;
            if (tid * 64u >= tn) mk = 0ull;
            else if (tn - tid * 64u < 64u) mk &= (1ull << (tn - tid * 64u)) - 1ull;
            uint32_t tf = mk ? tid * 64u + (uint32_t)__ffsll((long long)mk) - 1u : UC_NOHEAD;   // UC_NOHEAD is the maximum
            int tl = mk ? (int)(tid * 64u + 63u - (uint32_t)__clzll((long long)mk)) : -1;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t y = __shfl_xor(tf, d);
                const int z = __shfl_xor(tl, d);
                tf = tf < y ? tf : y;
                tl = tl > z ? tl : z;
            }
            if (tid == 0) {
                tile_first[tile] = tf;
                tile_last[tile] = tl >= 0 ? (uint32_t)tl : UC_NOHEAD;
            }
        }
        __syncthreads();
        // ---- segment bounds of every key; members of long in-tile segments enter the hash set ----
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
            if (closed && end - start > UC_SMALL && !(s_cnt[p] & UC_NOCORR)) {
                uc_insert<UC_BUCKETS, UC_POSBITS>(s_hash, (uint32_t)start, s_umi[p] & sp.lo_mask, p);
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
            const uint32_t s = s_start[p];
            if (s == UC_OPEN) continue;
            // the (exclusive) end again from the head masks: two LDS words instead of a 2 KB array (the sixth workgroup per CU)
            const uint32_t eb = p >> 6;
            const unsigned long long gt = lane < 63u ? (s_heads[eb] >> (lane + 1u)) : 0ull;
            const uint32_t e = gt ? p + (uint32_t)__ffsll((long long)gt) : (uint32_t)s_carry_end[eb];
            const uint64_t k = t0 + p;
            const uint32_t cw = s_cnt[p];
            const uint32_t my_cnt = cw & ~UC_NOCORR;
            uint32_t target = NONE32;
            if (!(cw & UC_NOCORR) && e - s > 1) {
                const uint32_t bp = best_neighbour_lds<UC_BUCKETS, UC_POSBITS>(s_umi, s_cnt, s_hash, s_start, s, e, s, p,
                                                                              s_umi[p], my_cnt, sp);
                if (bp != p) target = (uint32_t)(t0 + bp);
            }
            if (target != NONE32) move_reads(corr, st, inc_all, minidx, k, target, my_cnt);
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
// segment when the previous boundary is not inside the same segment, i.e. when tile t-1 holds a head.
template <bool SMALL>
__global__ __launch_bounds__(EdgeCfg<SMALL>::THREADS) void k_correct_umis_edges(
    const KL kl, const uint64_t *__restrict__ ukey, const uint32_t *__restrict__ upos, uint64_t nd, uint64_t n_keys,
    const uint32_t *__restrict__ tile_first, const uint32_t *__restrict__ tile_last, uint32_t *__restrict__ corr,
    uint16_t *__restrict__ st, uint32_t *__restrict__ inc_all, uint32_t *__restrict__ minidx, GiantItem *__restrict__ giant_items,
    uint32_t *__restrict__ n_giant) {
    constexpr uint32_t UE_T = EdgeCfg<SMALL>::THREADS, CAP = EdgeCfg<SMALL>::CAP, BUCKETS = EdgeCfg<SMALL>::BUCKETS,
                       POSBITS = EdgeCfg<SMALL>::POSBITS;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *s_umi = smem;             // CAP
    uint32_t *s_cnt = smem + CAP;       // CAP
    uint32_t *s_hash = smem + 2 * CAP;  // BUCKETS * 8 slots (16-byte aligned: CAP is a multiple of 4)
    const uint32_t tid = threadIdx.x;
    const uint64_t umi_mask = lowmask(kl.bits_umi);
    const UmiSplit sp = umi_split(kl.umi_len);
    const uint64_t n_tiles = (nd + UC_TILE - 1) / UC_TILE;
    for (uint64_t t = (uint64_t)blockIdx.x + 1; t < n_tiles; t += gridDim.x) {
        // every thread reads the same few words: the branches below are uniform across the workgroup
        const uint32_t tf = tile_first[t];
        if (tf == 0u) continue;  // the boundary is a segment head: nothing crosses it
        const uint32_t tl = tile_last[t - 1];
        if (tl == UC_NOHEAD) continue;  // the previous boundary is inside the same segment and owns it
        const uint64_t s = (t - 1) * UC_TILE + tl;
        uint64_t e = nd;
        if (tf != UC_NOHEAD) {
            e = t * UC_TILE + tf;
        } else {
            for (uint64_t u = t + 1; u < n_tiles; u++) {
                const uint32_t f = tile_first[u];
                if (f != UC_NOHEAD) {
                    e = u * UC_TILE + f;
                    break;
                }
            }
        }
        const uint64_t m = e - s;
        if (SMALL ? m > UES_CAP : m <= UES_CAP) continue;  // the other instantiation's segment
        const uint32_t lib = (uint32_t)((ukey[s] >> kl.sh_libid) & lowmask(kl.bits_lib));
        if ((kl.mux_mask >> lib) & 1u) continue;  // UmiCorrection::Disable (aligner.rs:315-318): nothing moves
        if (m <= CAP) {
            const uint32_t mm = (uint32_t)m;
            const bool use_hash = mm > UC_SMALL;
            if (use_hash)
                for (uint32_t h = tid; h < BUCKETS * 8; h += UE_T) s_hash[h] = UC_EMPTY;
            for (uint32_t p = tid; p < mm; p += UE_T) {
                const uint64_t k = s + p;
                s_umi[p] = (uint32_t)((ukey[k] >> kl.sh_umi) & umi_mask);
                s_cnt[p] = run_count(upos, nd, n_keys, k);
            }
            __syncthreads();
            if (use_hash) {
                for (uint32_t p = tid; p < mm; p += UE_T)
                    uc_insert<BUCKETS, POSBITS>(s_hash, 0u, s_umi[p] & sp.lo_mask, p);
                __syncthreads();
            }
            for (uint32_t p = tid; p < mm; p += UE_T) {
                const uint32_t my_cnt = s_cnt[p];
                const uint32_t bp = best_neighbour_lds<BUCKETS, POSBITS>(s_umi, s_cnt, s_hash, nullptr, 0u, mm, 0u, p,
                                                                              s_umi[p], my_cnt, sp);
                if (bp != p) move_reads(corr, st, inc_all, minidx, s + p, (uint32_t)(s + bp), my_cnt);
            }
            __syncthreads();
            continue;
        }
        // ---- larger than the LDS budget: one work item per chunk of CAP keys for the k_giant_* kernels, so that a
        // segment of tens of thousands of keys is spread over many workgroups instead of being this one's long pole
        {
            __shared__ uint32_t s_item0;
            const uint32_t n_chunks = (uint32_t)((m + CAP - 1) / CAP);
            if (tid == 0) s_item0 = atomicAdd(n_giant, n_chunks);
            __syncthreads();
            const uint32_t item0 = s_item0;
            for (uint32_t c = tid; c < n_chunks; c += UE_T)
                giant_items[item0 + c] = GiantItem{(uint32_t)s, (uint32_t)e, (uint32_t)(s + (uint64_t)c * CAP)};
            __syncthreads();
        }
    }
}

// ---- giant segments (more than UE_CAP keys) -------------------------------------------------------------------
// best[k] = (count << 32) | index of the best key found so far for k (its own to start with).  Inside a segment
// the index order is the UMI order, so the packed value orders exactly like (count, UMI) and atomicMax merges
// the findings of the workgroups that probe different table chunks.
#define GI_HALO 512u  // keys staged in front of and behind the chunk
__global__ __launch_bounds__(UE_THREADS) void k_giant_init(const KL kl, const uint64_t *__restrict__ ukey,
                                                           const uint32_t *__restrict__ upos, uint64_t nd, uint64_t n_keys,
                                                           const GiantItem *__restrict__ items,
                                                           const uint32_t *__restrict__ n_items_ptr,
                                                           unsigned long long *__restrict__ best) {
    // The runs of equal high halves around the chunk's keys are walked in LDS: the chunk and GI_HALO keys on either side are
    // staged once, coalesced (UMI and read count); only a run that leaves the staged window goes on in global memory.  (Every
    // step of the walk used to be two dependent global loads: 0.7 ms for a few million keys.)
    __shared__ uint32_t s_u[UE_CAP + 2 * GI_HALO], s_c[UE_CAP + 2 * GI_HALO];
    const uint32_t n_items = *n_items_ptr;
    const uint64_t umi_mask = lowmask(kl.bits_umi);
    const UmiSplit sp = umi_split(kl.umi_len);
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const GiantItem g = items[it];
        const uint64_t s = g.s, e = g.e;
        const uint64_t c1 = (uint64_t)g.c0 + UE_CAP < e ? (uint64_t)g.c0 + UE_CAP : e;
        const uint64_t w0 = (uint64_t)g.c0 >= s + GI_HALO ? (uint64_t)g.c0 - GI_HALO : s;  // staged window [w0, w1) inside the segment
        const uint64_t w1 = c1 + GI_HALO < e ? c1 + GI_HALO : e;
        __syncthreads();
        for (uint64_t k = w0 + threadIdx.x; k < w1; k += UE_THREADS) {
            s_u[k - w0] = (uint32_t)((ukey[k] >> kl.sh_umi) & umi_mask);
            s_c[k - w0] = run_count(upos, nd, n_keys, k);
        }
        __syncthreads();
        // same-high-half neighbours: the run around each key in the sorted array
        for (uint64_t k = g.c0 + threadIdx.x; k < c1; k += UE_THREADS) {
            const uint32_t my_umi = s_u[k - w0];
            const uint32_t my_hi = my_umi >> sp.lo_bits, my_lo = my_umi & sp.lo_mask;
            unsigned long long b = ((unsigned long long)s_c[k - w0] << 32) | (uint32_t)k;
            bool open = true;  // the run may go on in front of the window
            for (uint64_t q = k; q > w0;) {
                --q;
                const uint32_t u = s_u[q - w0];
                if ((u >> sp.lo_bits) != my_hi) {
                    open = false;
                    break;
                }
                if (hd1(u & sp.lo_mask, my_lo)) {
                    const unsigned long long c = ((unsigned long long)s_c[q - w0] << 32) | (uint32_t)q;
                    b = c > b ? c : b;
                }
            }
            if (open)
                for (uint64_t q = w0; q > s;) {
                    --q;
                    const uint32_t u = (uint32_t)((ukey[q] >> kl.sh_umi) & umi_mask);
                    if ((u >> sp.lo_bits) != my_hi) break;
                    if (hd1(u & sp.lo_mask, my_lo)) {
                        const unsigned long long c = ((unsigned long long)run_count(upos, nd, n_keys, q) << 32) | (uint32_t)q;
                        b = c > b ? c : b;
                    }
                }
            open = true;
            for (uint64_t q = k + 1; q < w1; q++) {
                const uint32_t u = s_u[q - w0];
                if ((u >> sp.lo_bits) != my_hi) {
                    open = false;
                    break;
                }
                if (hd1(u & sp.lo_mask, my_lo)) {
                    const unsigned long long c = ((unsigned long long)s_c[q - w0] << 32) | (uint32_t)q;
                    b = c > b ? c : b;
                }
            }
            if (open)
                for (uint64_t q = w1; q < e; q++) {
                    const uint32_t u = (uint32_t)((ukey[q] >> kl.sh_umi) & umi_mask);
                    if ((u >> sp.lo_bits) != my_hi) break;
                    if (hd1(u & sp.lo_mask, my_lo)) {
                        const unsigned long long c = ((unsigned long long)run_count(upos, nd, n_keys, q) << 32) | (uint32_t)q;
                        b = c > b ? c : b;
                    }
                }
            best[k] = b;
        }
    }
}

// same-low-half neighbours: the item's chunk of UE_CAP keys becomes an LDS table, every key of the segment probes it
__global__ __launch_bounds__(UE_THREADS) void k_giant_probe(const KL kl, const uint64_t *__restrict__ ukey,
                                                            const uint32_t *__restrict__ upos, uint64_t nd, uint64_t n_keys,
                                                            const GiantItem *__restrict__ items,
                                                            const uint32_t *__restrict__ n_items_ptr,
                                                            unsigned long long *__restrict__ best) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *s_umi = smem;                // UE_CAP
    uint32_t *s_cnt = smem + UE_CAP;       // UE_CAP
    uint32_t *s_hash = smem + 2 * UE_CAP;  // UE_BUCKETS * 8 slots
    const uint32_t n_items = *n_items_ptr;
    const uint32_t tid = threadIdx.x;
    const uint64_t umi_mask = lowmask(kl.bits_umi);
    const UmiSplit sp = umi_split(kl.umi_len);
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const GiantItem g = items[it];
        const uint64_t s = g.s, e = g.e, c0 = g.c0;
        const uint32_t cn = e - c0 < UE_CAP ? (uint32_t)(e - c0) : UE_CAP;
        for (uint32_t h = tid; h < UE_BUCKETS * 8; h += UE_THREADS) s_hash[h] = UC_EMPTY;
        for (uint32_t p = tid; p < cn; p += UE_THREADS) {
            s_umi[p] = (uint32_t)((ukey[c0 + p] >> kl.sh_umi) & umi_mask);
            s_cnt[p] = run_count(upos, nd, n_keys, c0 + p);
        }
        __syncthreads();
        for (uint32_t p = tid; p < cn; p += UE_THREADS)
            uc_insert<UE_BUCKETS, UE_POSBITS>(s_hash, 0u, s_umi[p] & sp.lo_mask, p);
        __syncthreads();
        for (uint64_t k = s + tid; k < e; k += UE_THREADS) {
            const uint32_t my_umi = (uint32_t)((ukey[k] >> kl.sh_umi) & umi_mask);
            const uint32_t my_hi = my_umi >> sp.lo_bits, my_lo = my_umi & sp.lo_mask;
            unsigned long long b = 0ull;
            uc_for_each<UE_BUCKETS, UE_POSBITS>(s_hash, 0u, my_lo, [&](uint32_t q) {
                const uint32_t u = s_umi[q];
                if ((u & sp.lo_mask) != my_lo) return;
                if (hd1(u >> sp.lo_bits, my_hi)) {
                    const unsigned long long c = ((unsigned long long)s_cnt[q] << 32) | (uint32_t)(c0 + q);
                    b = c > b ? c : b;
                }
            });
            if (b) atomicMax(&best[k], b);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(UE_THREADS) void k_giant_final(const uint32_t *__restrict__ upos, uint64_t nd, uint64_t n_keys,
                                                            const GiantItem *__restrict__ items,
                                                            const uint32_t *__restrict__ n_items_ptr,
                                                            const unsigned long long *__restrict__ best,
                                                            uint32_t *__restrict__ corr, uint16_t *__restrict__ st,
                                                            uint32_t *__restrict__ inc_all, uint32_t *__restrict__ minidx) {
    const uint32_t n_items = *n_items_ptr;
    for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {
        const GiantItem g = items[it];
        const uint64_t c1 = (uint64_t)g.c0 + UE_CAP < g.e ? (uint64_t)g.c0 + UE_CAP : g.e;
        for (uint64_t k = g.c0 + threadIdx.x; k < c1; k += UE_THREADS) {
            const uint32_t target = (uint32_t)best[k];
            if (target != (uint32_t)k) move_reads(corr, st, inc_all, minidx, k, target, run_count(upos, nd, n_keys, k));
        }
    }
}
