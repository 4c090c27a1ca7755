// feature.hip -- K3: feature-barcode matching with 1-mismatch posterior correction for one
// tethered pattern (one capture per read).
//
// Replaces FeatureExtractor::find_closest and correct_feature_barcode
// (cr_types/src/reference/feature_extraction.rs:443-470 and :34-117; constants :21-22).
// The feature table (<= a few thousand packed sequences) is sorted and binary-searched; it lives
// in L1/L2.  f64 arithmetic in the reference's order (position-major, A<C<G<T), no FMA contraction.
#include <algorithm>
#include <cmath>

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

extern "C" int crgpu_set_feature_pattern(crgpu_ctx *ctx, int pattern, const char *feat_seqs, uint32_t n_feat, uint32_t len,
                                         const uint32_t *feat_index, const double *feat_dist) {
    if (!ctx) return CRGPU_EINVAL;
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
    CR_REQUIRE(ctx, pattern >= 0 && pattern < CRGPU_MAX_LIB && ctx->pat[pattern].set, CRGPU_ESTATE,
               "feature pattern %d not set", pattern);
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_seq && d_qualn && d_feature_out, CRGPU_EINVAL, "crgpu_match_features: NULL buffer");
    const FeaturePattern &P = ctx->pat[pattern];
    // p_edit[qv] = 10^(-qv/10) for qv = 0..33, host libm as in feature_extraction.rs:45
    double pe[34];
    for (int q = 0; q < 34; q++) pe[q] = std::pow(10.0, -(double)q / 10.0);
    double *d_pe = (double *)(ctx->d_scalars + 128);  // 34 doubles inside the 4 KB scalar page
    CR_HIP(ctx, hipMemcpyAsync(d_pe, pe, sizeof(pe), hipMemcpyHostToDevice, ctx->stream));
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pe is a stack buffer
    PatView v{P.d_seq, P.d_index, P.has_dist ? P.d_dist : nullptr, P.n, P.len};
    CrTimer t(ctx, CRGPU_T_MATCH, n);
    hipLaunchKernelGGL(k_match_features, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, v, d_pe, d_seq, d_qualn, n,
                       d_feature_out);
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}
