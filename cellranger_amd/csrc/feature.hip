// feature.hip -- K3: feature-barcode matching with 1-mismatch posterior correction for one
// tethered pattern (one capture per read).
//
// Replaces FeatureExtractor::find_closest and correct_feature_barcode
// (cr_types/src/reference/feature_extraction.rs:443-470 and :34-117; constants :21-22).
// The feature table (<= a few thousand packed sequences) is sorted and binary-searched; it lives
// in L1/L2.  f64 arithmetic in the reference's order (position-major, A<C<G<T), no FMA contraction.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"

struct PatView {
    const uint32_t *seq;
    const uint32_t *index;
    const double *dist;  // nullptr => exact matches only
    uint32_t n, len;
};

__device__ __forceinline__ int pat_find(const PatView &p, uint32_t key) {
    uint32_t lo = 0, hi = p.n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (p.seq[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < p.n && p.seq[lo] == key) ? (int)lo : -1;
}

__global__ __launch_bounds__(256) void k_match_features(const PatView p, const double *__restrict__ pedit,
                                                        const uint32_t *__restrict__ seq, const uint8_t *__restrict__ qualn,
                                                        uint64_t n, uint32_t *__restrict__ feature_out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t L = p.len;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t key = seq[i];
        uint32_t nmask = 0;
        for (uint32_t k = 0; k < L; k++) nmask |= (uint32_t)(qualn[i * L + k] >> 7) << k;
        uint32_t result = CRGPU_NO_FEATURE;
        int exact = nmask ? -1 : pat_find(p, key);
        if (exact >= 0) {
            result = p.index[exact];  // find_closest fast path (:452-457)
        } else if (p.dist && __popc(nmask) <= 1) {
            double sum = 0.0, mx = -1.0;
            int best = -1;
            for (uint32_t pos = 0; pos < L; pos++) {
                if (nmask && !((nmask >> pos) & 1u)) continue;  // other positions still hold the N
                const uint32_t sh = 2u * (L - 1u - pos);
                const uint32_t orig = (key >> sh) & 3u;
                const bool is_n = (nmask >> pos) & 1u;
                for (uint32_t b = 0; b < 4; b++) {
                    if (!is_n && b == orig) continue;
                    const int f = pat_find(p, (key & ~(3u << sh)) | (b << sh));
                    if (f < 0) continue;
                    // qv = min(qual - 33, FEATURE_MAX_QV) with u8 wrapping (:43-44)
                    uint32_t qv = (uint8_t)((qualn[i * L + pos] & 0x7Fu) - 33u);
                    qv = qv < 33u ? qv : 33u;
                    const double like = p.dist[f] * pedit[qv];
                    sum += like;
                    if (like > mx) {
                        mx = like;
                        best = f;
                    }
                }
            }
            if (best >= 0 && (mx / sum) >= 0.975) result = p.index[best];  // FEATURE_CONF_THRESHOLD
        }
        feature_out[i] = result;
    }
}

// ---- the same with the feature table as an LDS hash set and the quality rows staged through LDS -------------
// k_match_features above does a binary search in global memory per lookup (8-13 dependent loads, 45 lookups for a
// capture that needs correcting) and reads the L quality bytes of a read one by one: 6 G reads/s.  Here the table
// (<= FM_MAX_FEATURES sequences) lives in LDS as an open-addressing set of (sequence, position) pairs, and every wave
// copies the 64 * L quality bytes of its reads with coalesced dword loads into a private LDS area from which each
// lane assembles its row.
#define FM_MAX_FEATURES 4096u
#define FM_EMPTY 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ uint32_t fm_hash(uint32_t key, uint32_t mask) { return ((key * 0x9E3779B1u) >> 12) & mask; }
__device__ __forceinline__ int fm_find(const unsigned long long *tab, uint32_t mask, uint32_t key) {
    uint32_t s = fm_hash(key, mask);
    for (;;) {
        const unsigned long long e = tab[s];
        if (e == FM_EMPTY) return -1;
        if ((uint32_t)e == key) return (int)(e >> 32);
        s = (s + 1u) & mask;
    }
}

// the 1-mismatch posterior of one capture (correct_feature_barcode, feature_extraction.rs:34-117)
__device__ __forceinline__ uint32_t fm_correct(const PatView &p, const double *__restrict__ pedit, const unsigned long long *tab,
                                               uint32_t slot_mask, uint32_t key, unsigned long long qlo,
                                               unsigned long long qhi, uint32_t nmask) {
    const uint32_t L = p.len;
    double sum = 0.0, mx = -1.0;
    int best = -1;
    for (uint32_t pos = 0; pos < L; pos++) {
        if (nmask && !((nmask >> pos) & 1u)) continue;  // other positions still hold the N
        const uint32_t shp = 2u * (L - 1u - pos);
        const uint32_t orig = (key >> shp) & 3u;
        const bool is_n = (nmask >> pos) & 1u;
        for (uint32_t b = 0; b < 4; b++) {
            if (!is_n && b == orig) continue;
            const int f = fm_find(tab, slot_mask, (key & ~(3u << shp)) | (b << shp));
            if (f < 0) continue;
            // qv = min(qual - 33, FEATURE_MAX_QV) with u8 wrapping (:43-44)
            const uint32_t qb = (uint32_t)((pos < 8u ? qlo : qhi) >> (8u * (pos & 7u))) & 0x7Fu;
            uint32_t qv = (uint8_t)(qb - 33u);
            qv = qv < 33u ? qv : 33u;
            const double like = p.dist[f] * pedit[qv];
            sum += like;
            if (like > mx) {
                mx = like;
                best = f;
            }
        }
    }
    return (best >= 0 && (mx / sum) >= 0.975) ? p.index[best] : CRGPU_NO_FEATURE;  // FEATURE_CONF_THRESHOLD
}

struct FmPending {  // a capture that matched no feature exactly and has at most one N
    unsigned long long qlo, qhi;
    uint32_t i_lo, i_hi, key, nmask;
};

template <uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void k_match_features_lds(const PatView p, const double *__restrict__ pedit,
                                                            const uint32_t *__restrict__ seq, const uint8_t *__restrict__ qualn,
                                                            uint64_t n, uint32_t *__restrict__ feature_out, uint32_t slot_mask) {
    // LDS: the table (slot_mask + 1 entries), 272 u32 of quality staging per wave, the queue of 2 * THREADS captures
    // to correct (small tables: 256 threads, several workgroups per CU; large ones: one workgroup of 1024 per CU).
    // Only ~10 % of the captures need the 3L-candidate posterior; run inline it kept whole waves waiting for a handful
    // of lanes, so those captures are queued per workgroup and corrected 256 at a time with every lane busy.
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_tab[];
    uint32_t *s_q = reinterpret_cast<uint32_t *>(s_tab + slot_mask + 1) + (threadIdx.x >> 6) * 272u;
    FmPending *s_pend = reinterpret_cast<FmPending *>(reinterpret_cast<uint32_t *>(s_tab + slot_mask + 1) + (THREADS / 64u) * 272u);
    __shared__ uint32_t s_npend;
    const uint32_t L = p.len, lane = threadIdx.x & 63u, tid = threadIdx.x;
    for (uint32_t s = tid; s <= slot_mask; s += THREADS) s_tab[s] = FM_EMPTY;
    if (tid == 0) s_npend = 0;
    __syncthreads();
    for (uint32_t f = tid; f < p.n; f += THREADS) {
        const unsigned long long e = ((unsigned long long)f << 32) | p.seq[f];
        uint32_t s = fm_hash(p.seq[f], slot_mask);
        while (atomicCAS(&s_tab[s], FM_EMPTY, e) != FM_EMPTY) s = (s + 1u) & slot_mask;
    }
    __syncthreads();
    auto drain = [&](uint32_t first, uint32_t count) {  // entries [first, first + count), count <= THREADS
        if (tid < count) {
            const FmPending e = s_pend[first + tid];
            const uint64_t i = ((uint64_t)e.i_hi << 32) | e.i_lo;
            feature_out[i] = fm_correct(p, pedit, s_tab, slot_mask, e.key, e.qlo, e.qhi, e.nmask);
        }
    };
    const uint64_t total_bytes = n * L;
    for (uint64_t base = (uint64_t)blockIdx.x * THREADS; base < n; base += (uint64_t)gridDim.x * THREADS) {
        const uint64_t i = base + tid;
        // the wave's 64 rows (64 * L bytes, a multiple of 4, starting at a multiple of 4) -> LDS
        const uint64_t wbyte = (base + (tid & ~63u)) * L;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t d = lane; d < 16u * L; d += 64) {
            const uint64_t b = wbyte + 4ull * d;
            uint32_t v = 0;
            if (b + 4 <= total_bytes)
                v = *reinterpret_cast<const uint32_t *>(qualn + b);
            else
                for (uint32_t k = 0; k < 4; k++)
                    if (b + k < total_bytes) v |= (uint32_t)qualn[b + k] << (8 * k);
            s_q[d] = v;
        }
        __builtin_amdgcn_wave_barrier();
        // this lane's row as two 64-bit words: byte k of (qlo, qhi) = position k
        const uint32_t o = lane * L, d0 = o >> 2, sh = (o & 3u) * 8u;
        const uint32_t w0 = s_q[d0], w1 = s_q[d0 + 1], w2 = s_q[d0 + 2], w3 = s_q[d0 + 3], w4 = s_q[d0 + 4];
        unsigned long long qlo = ((unsigned long long)w1 << 32) | w0, qmid = ((unsigned long long)w3 << 32) | w2;
        unsigned long long qhi;
        if (sh) {
            qlo = (qlo >> sh) | (qmid << (64u - sh));
            qhi = (qmid >> sh) | ((unsigned long long)w4 << (64u - sh));
        } else {
            qhi = qmid;
        }
        if (L < 8u) qlo &= (1ull << (8u * L)) - 1ull;
        if (L <= 8u) qhi = 0ull; else if (L < 16u) qhi &= (1ull << (8u * (L - 8u))) - 1ull;
        if (i < n) {
            const uint32_t key = seq[i];
            const uint32_t nmask = (uint32_t)(((qlo & 0x8080808080808080ull) * 0x0002040810204081ull) >> 56) |
                                   ((uint32_t)(((qhi & 0x8080808080808080ull) * 0x0002040810204081ull) >> 56) << 8);
            const int exact = nmask ? -1 : fm_find(s_tab, slot_mask, key);
            if (exact >= 0) {
                feature_out[i] = p.index[exact];  // find_closest fast path (:452-457)
            } else if (p.dist && __popc(nmask) <= 1) {
                const uint32_t slot = atomicAdd(&s_npend, 1u);  // < 2 * THREADS: fewer than THREADS left over + THREADS new
                s_pend[slot] = FmPending{qlo, qhi, (uint32_t)i, (uint32_t)(i >> 32), key, nmask};
            } else {
                feature_out[i] = CRGPU_NO_FEATURE;
            }
        }
        __syncthreads();
        const uint32_t np = s_npend;
        if (np >= THREADS) {  // uniform
            drain(np - THREADS, THREADS);  // the newest ones: the older ones stay at the front
            __syncthreads();
            if (tid == 0) s_npend = np - THREADS;
        }
        __syncthreads();
    }
    drain(0u, s_npend);  // fewer than THREADS left
}

extern "C" int crgpu_set_feature_pattern(crgpu_ctx *ctx, int pattern, const char *feat_seqs, uint32_t n_feat, uint32_t len,
                                         const uint32_t *feat_index, const double *feat_dist) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, pattern >= 0 && pattern < CRGPU_MAX_LIB, CRGPU_EINVAL, "pattern id %d out of range", pattern);
    CR_REQUIRE(ctx, feat_seqs && feat_index && n_feat > 0, CRGPU_EINVAL, "empty feature pattern");
    CR_REQUIRE(ctx, len >= 1 && len <= 16, CRGPU_ERANGE, "feature barcode length %u unsupported (<= 16 bases)", len);
    struct E {
        uint32_t seq, index;
        double dist;
    };
    std::vector<E> es(n_feat);
    for (uint32_t i = 0; i < n_feat; i++) {
        uint32_t k = 0;
        for (uint32_t j = 0; j < len; j++) {
            uint32_t c;
            switch (feat_seqs[(size_t)i * len + j]) {
                case 'A': c = 0; break;
                case 'C': c = 1; break;
                case 'G': c = 2; break;
                case 'T': c = 3; break;
                default:
                    return cr_fail(ctx, CRGPU_EINVAL, "feature %u has a non-ACGT base (N in feature sequences is unsupported)", i);
            }
            k = (k << 2) | c;
        }
        es[i] = {k, feat_index[i], feat_dist ? feat_dist[i] : 0.0};
    }
    std::sort(es.begin(), es.end(), [](const E &a, const E &b) { return a.seq < b.seq; });
    for (uint32_t i = 1; i < n_feat; i++)
        CR_REQUIRE(ctx, es[i].seq != es[i - 1].seq, CRGPU_EINVAL,
                   "two features share one barcode sequence in this pattern (feature_extraction.rs:152-163)");
    std::vector<uint32_t> s(n_feat), ix(n_feat);
    std::vector<double> d(n_feat);
    for (uint32_t i = 0; i < n_feat; i++) {
        s[i] = es[i].seq;
        ix[i] = es[i].index;
        d[i] = es[i].dist;
    }
    FeaturePattern &P = ctx->pat[pattern];
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(P.d_seq);
    (void)hipFree(P.d_index);
    (void)hipFree(P.d_dist);
    P = FeaturePattern();
    CR_HIP(ctx, hipMalloc((void **)&P.d_seq, n_feat * sizeof(uint32_t)));
    CR_HIP(ctx, hipMalloc((void **)&P.d_index, n_feat * sizeof(uint32_t)));
    CR_HIP(ctx, hipMalloc((void **)&P.d_dist, n_feat * sizeof(double)));
    CR_HIP(ctx, hipMemcpy(P.d_seq, s.data(), n_feat * sizeof(uint32_t), hipMemcpyHostToDevice));
    CR_HIP(ctx, hipMemcpy(P.d_index, ix.data(), n_feat * sizeof(uint32_t), hipMemcpyHostToDevice));
    CR_HIP(ctx, hipMemcpy(P.d_dist, d.data(), n_feat * sizeof(double), hipMemcpyHostToDevice));
    P.n = n_feat;
    P.len = len;
    P.has_dist = feat_dist != nullptr;
    P.set = true;
    return CRGPU_OK;
}

extern "C" int crgpu_match_features_dev(crgpu_ctx *ctx, int pattern, const uint32_t *d_seq, const uint8_t *d_qualn,
                                        uint64_t n, uint32_t *d_feature_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, pattern >= 0 && pattern < CRGPU_MAX_LIB && ctx->pat[pattern].set, CRGPU_ESTATE,
               "feature pattern %d not set", pattern);
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_seq && d_qualn && d_feature_out, CRGPU_EINVAL, "crgpu_match_features: NULL buffer");
    cr_invalidate_range(ctx, d_feature_out, n * sizeof(uint32_t));  // a caller buffer is written: by-products about it go
    const FeaturePattern &P = ctx->pat[pattern];
    // p_edit[qv] = 10^(-qv/10) for qv = 0..33, host libm as in feature_extraction.rs:45
    double pe[34];
    for (int q = 0; q < 34; q++) pe[q] = std::pow(10.0, -(double)q / 10.0);
    double *d_pe = (double *)(ctx->d_scalars + 128);  // 34 doubles inside the 4 KB scalar page
    CR_HIP(ctx, hipMemcpyAsync(d_pe, pe, sizeof(pe), hipMemcpyHostToDevice, ctx->stream));
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pe is a stack buffer
    PatView v{P.d_seq, P.d_index, P.has_dist ? P.d_dist : nullptr, P.n, P.len};
    CrTimer t(ctx, CRGPU_T_FEATURE, n);
    if (P.n <= FM_MAX_FEATURES && (uintptr_t)d_qualn % 4 == 0 && !getenv("CRGPU_FEATURES_GLOBAL")) {
        uint32_t slots = 64;
        while (slots < 2u * P.n) slots <<= 1;  // load factor <= 0.5
        if (P.n <= 1024u) {
            const size_t lds = (size_t)slots * sizeof(unsigned long long) + 4 * 272 * sizeof(uint32_t) + 2 * 256 * sizeof(FmPending);
            cr_allow_lds(ctx, (const void *)k_match_features_lds<256>, lds);
            hipLaunchKernelGGL(k_match_features_lds<256>, dim3(cr_grid(n, 256, 256u * 4u)), dim3(256), lds, ctx->stream, v, d_pe,
                               d_seq, d_qualn, n, d_feature_out, slots - 1u);
        } else {
            const size_t lds = (size_t)slots * sizeof(unsigned long long) + 16 * 272 * sizeof(uint32_t) + 2 * 1024 * sizeof(FmPending);
            cr_allow_lds(ctx, (const void *)k_match_features_lds<1024>, lds);
            hipLaunchKernelGGL(k_match_features_lds<1024>, dim3(cr_grid(n, 1024, 256u)), dim3(1024), lds, ctx->stream, v, d_pe,
                               d_seq, d_qualn, n, d_feature_out, slots - 1u);
        }
    } else {
        hipLaunchKernelGGL(k_match_features, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, v, d_pe, d_seq, d_qualn, n,
                           d_feature_out);
    }
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}
