// metrics.hip -- MAKE_SHARD's per-read quality metrics over the barcode and UMI parts of the reads, as one fused
// streaming reduction over the packed arrays the hot path already holds.
// Reference: MakeShardVisitor::visit_processed_read, cr_lib/src/make_shard_metrics.rs:263-332; frac_n_bases /
// frac_q30_bases :355-392; thresholds :20-23; RnaRead::barcode_min_qual / umi_min_qual cr_types/src/rna_read.rs:738-749;
// UmiInfo validity umi/src/info.rs:20-37.  PercentMetrics are returned as numerator / denominator counts.
#include "common.h"

#define SM_FIELDS 17

// N flags and quality predicates of a row of `len` quality bytes (bit 7 = the base was N)
struct RowStats {
    uint32_t n_bases, q30, q30_den, min_q;
    bool any_low;  // some (q - 33) as u8 below 10: the per-base rule of UmiInfo::new
};
__device__ __forceinline__ void row_byte(RowStats &r, uint32_t b) {
    const uint32_t v = b & 0x7Fu;
    r.n_bases += b >> 7;
    if (v > 2u + 33u) {
        r.q30_den++;
        r.q30 += v >= 30u + 33u;
    }
    r.min_q = v < r.min_q ? v : r.min_q;
    r.any_low |= (uint8_t)(v - 33u) < 10u;
}
__device__ __forceinline__ RowStats row_stats(const uint8_t *__restrict__ q, uint32_t len) {
    RowStats r{0, 0, 0, 255u, false};
    if ((len & 3u) == 0u && ((uintptr_t)q & 3u) == 0u) {
        // rows of 4k bytes: k dword loads (12- and 16-byte rows of neighbouring lanes coalesce)
        const uint32_t *__restrict__ w = reinterpret_cast<const uint32_t *>(q);
        uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            if (k < (len >> 2)) d[k] = w[k];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            if (k < (len >> 2))
#pragma unroll
                for (uint32_t b = 0; b < 4; b++) row_byte(r, (d[k] >> (8u * b)) & 0xFFu);
    } else {
        for (uint32_t k = 0; k < len; k++) row_byte(r, q[k]);
    }
    return r;
}
// every adjacent pair of bases equal (an N only equals an N): packed 2-bit codes + the N flags of the quality bytes
__device__ __forceinline__ bool is_homopolymer(uint32_t packed, const uint8_t *__restrict__ q, uint32_t len, uint32_t n_bases) {
    if (n_bases == len) return true;
    if (n_bases != 0u) return false;
    const uint32_t bits = 2u * len;
    const uint32_t m = bits >= 32u ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    const uint32_t adj = bits >= 2u ? ((packed ^ (packed >> 2)) & (m >> 2)) : 0u;
    return adj == 0u;
}

__global__ __launch_bounds__(256) void k_shard_metrics(const uint32_t *__restrict__ cb, const uint8_t *__restrict__ cbq,
                                                       uint32_t cb_len, const uint32_t *__restrict__ umi,
                                                       const uint8_t *__restrict__ umiq, uint32_t umi_len,
                                                       const uint32_t *__restrict__ idx, uint64_t n,
                                                       unsigned long long *__restrict__ out) {
    __shared__ unsigned long long s[4][SM_FIELDS];
    unsigned long long a[SM_FIELDS];
    for (int f = 0; f < SM_FIELDS; f++) a[f] = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const RowStats b = row_stats(cbq + i * cb_len, cb_len);
        const RowStats u = row_stats(umiq + i * umi_len, umi_len);
        const bool bc_homo = is_homopolymer(cb[i], cbq + i * cb_len, cb_len, b.n_bases);
        const bool umi_homo = is_homopolymer(umi[i], umiq + i * umi_len, umi_len, u.n_bases);
        a[0] += 1;
        a[1] += b.n_bases;
        a[2] += cb_len;
        a[3] += u.n_bases;
        a[4] += umi_len;
        a[5] += b.q30;
        a[6] += b.q30_den;
        a[7] += u.q30;
        a[8] += u.q30_den;
        a[9] += !(u.n_bases != 0u || umi_homo || u.any_low);       // good_umi
        a[10] += b.n_bases != 0u;                                  // has_n_barcode_property
        a[11] += u.n_bases != 0u;                                  // has_n_umi_property
        a[12] += bc_homo;
        a[13] += umi_homo;
        a[14] += (uint8_t)(b.min_q - 33u) < 10u;                   // low_min_qual_barcode_property
        a[15] += (uint8_t)(u.min_q - 33u) < 10u;                   // low_min_qual_umi_property
        if (idx) a[16] += idx[i] == CRGPU_MISS;                    // miss_whitelist_barcode_property
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int f = 0; f < SM_FIELDS; f++) {
        unsigned long long x = a[f];
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if (lane == 0) s[wave][f] = x;
    }
    __syncthreads();
    if (threadIdx.x < SM_FIELDS) {
        const unsigned long long t = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
        if (t) atomicAdd(&out[threadIdx.x * 16], t);  // one 128-byte line per counter
    }
}

extern "C" int crgpu_shard_metrics_dev(crgpu_ctx *ctx, const uint32_t *d_cb, const uint8_t *d_cb_qualn, uint32_t cb_len,
                                       const uint32_t *d_umi, const uint8_t *d_umi_qualn, uint32_t umi_len,
                                       const uint32_t *d_idx, uint64_t n, crgpu_shard_metrics *out) {
    if (!ctx || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    memset(out, 0, sizeof(*out));
    static_assert(sizeof(crgpu_shard_metrics) == SM_FIELDS * sizeof(uint64_t), "field count");
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_cb && d_cb_qualn && d_umi && d_umi_qualn, CRGPU_EINVAL, "crgpu_shard_metrics: NULL buffer");
    CR_REQUIRE(ctx, cb_len >= 1 && cb_len <= 16 && umi_len >= 1 && umi_len <= 16, CRGPU_ERANGE,
               "crgpu_shard_metrics: sequences of 1..16 bases");
    unsigned long long *d_acc = nullptr;
    CR_TRY(cr_pool_alloc(ctx, (void **)&d_acc, SM_FIELDS * 16 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_acc, 0, SM_FIELDS * 16 * sizeof(unsigned long long), ctx->stream);
    {
        CrTimer t(ctx, CRGPU_T_PACK, n);
        hipLaunchKernelGGL(k_shard_metrics, dim3(cr_grid(n, 256, 256u * 8u)), dim3(256), 0, ctx->stream, d_cb, d_cb_qualn, cb_len,
                           d_umi, d_umi_qualn, umi_len, d_idx, n, d_acc);
    }
    if (e == hipSuccess) e = hipGetLastError();
    unsigned long long h[SM_FIELDS * 16];
    int rc = e == hipSuccess ? crgpu_memcpy_d2h(ctx, h, d_acc, sizeof(h)) : CRGPU_EHIP;
    cr_pool_free(ctx, d_acc);
    if (e != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "crgpu_shard_metrics: %s", hipGetErrorString(e));
    CR_TRY(rc);
    uint64_t *o = reinterpret_cast<uint64_t *>(out);
    for (int f = 0; f < SM_FIELDS; f++) o[f] = h[f * 16];
    return CRGPU_OK;
}
