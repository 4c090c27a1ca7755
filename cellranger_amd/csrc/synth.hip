// synth.hip -- synthetic workload generation on the device (bench) and on the host (tests / CPU baseline).
#include <thread>
#include <vector>

#include "common.h"
#include "synth_core.h"

__global__ __launch_bounds__(256) void k_synth(const crgpu_synth_params p, uint64_t first, uint64_t n,
                                               const crgpu_synth_out o) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        CrSynthRead r;
        cr_synth_read(p, first + k, r);
        if (o.cb) o.cb[k] = r.cb;
        if (o.umi) o.umi[k] = r.umi;
        if (o.feature) o.feature[k] = r.feature;
        if (o.flags) o.flags[k] = r.flags;
        if (o.cb_qualn)
            for (uint32_t j = 0; j < p.cb_len; j++) o.cb_qualn[k * p.cb_len + j] = r.cbq[j];
        if (o.umi_qualn)
            for (uint32_t j = 0; j < p.umi_len; j++) o.umi_qualn[k * p.umi_len + j] = r.umiq[j];
    }
}

static int check_params(crgpu_ctx *ctx, const crgpu_synth_params *p) {
    if (!p) return cr_fail(ctx, CRGPU_EINVAL, "synth: NULL params");
    if (p->cb_len < 1 || p->cb_len > 16 || p->umi_len < 1 || p->umi_len > 16)
        return cr_fail(ctx, CRGPU_ERANGE, "synth: cb_len/umi_len must be 1..16");
    if (!p->wl_packed || p->n_wl == 0 || !p->cell_wl_pos || !p->cell_cdf || p->n_cells == 0)
        return cr_fail(ctx, CRGPU_EINVAL, "synth: whitelist / cell tables missing");
    if (p->n_ambient && !p->ambient_wl_pos) return cr_fail(ctx, CRGPU_EINVAL, "synth: ambient table missing");
    if (p->n_genes && !p->gene_cdf) return cr_fail(ctx, CRGPU_EINVAL, "synth: gene table missing");
    if (p->cell_cdf[p->n_cells - 1] != (1ull << 63) || (p->n_genes && p->gene_cdf[p->n_genes - 1] != (1ull << 63)))
        return cr_fail(ctx, CRGPU_EINVAL, "synth: the last CDF entry must be 2^63");
    return CRGPU_OK;
}

template <typename T>
static int to_dev(crgpu_ctx *ctx, const T *h, size_t n, const T **d_out, std::vector<void *> &owned) {
    *d_out = nullptr;
    if (!h || !n) return CRGPU_OK;
    void *d = nullptr;
    CR_HIP(ctx, hipMalloc(&d, n * sizeof(T)));
    owned.push_back(d);
    CR_HIP(ctx, hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    *d_out = (const T *)d;
    return CRGPU_OK;
}

extern "C" int crgpu_synth_dev(crgpu_ctx *ctx, const crgpu_synth_params *p, uint64_t first, uint64_t n,
                               const crgpu_synth_out *d_out) {
    if (!ctx || !d_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_TRY(check_params(ctx, p));
    if (n == 0) return CRGPU_OK;
    cr_invalidate(ctx);  // the output buffers may be ones a kept by-product describes
    crgpu_synth_params dp = *p;
    std::vector<void *> owned;
    int rc = CRGPU_OK;
    do {
        if ((rc = to_dev(ctx, p->wl_packed, p->n_wl, &dp.wl_packed, owned))) break;
        if ((rc = to_dev(ctx, p->cell_wl_pos, p->n_cells, &dp.cell_wl_pos, owned))) break;
        if ((rc = to_dev(ctx, p->cell_cdf, p->n_cells, &dp.cell_cdf, owned))) break;
        if ((rc = to_dev(ctx, p->ambient_wl_pos, p->n_ambient, &dp.ambient_wl_pos, owned))) break;
        if ((rc = to_dev(ctx, p->gene_cdf, p->n_genes, &dp.gene_cdf, owned))) break;
        {
            CrTimer t(ctx, CRGPU_T_SYNTH, n);
            hipLaunchKernelGGL(k_synth, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, dp, first, n, *d_out);
        }
        if (hipGetLastError() != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "k_synth launch failed");
    } while (0);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    for (void *d : owned) (void)hipFree(d);
    if (rc == CRGPU_OK && e != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "synth: %s", hipGetErrorString(e));
    return rc;
}

extern "C" int crgpu_synth_host(const crgpu_synth_params *p, uint64_t first, uint64_t n, const crgpu_synth_out *h_out) {
    if (!h_out) return CRGPU_EINVAL;
    CR_TRY(check_params(nullptr, p));
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
    if (n < 65536) nt = 1;
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t k = lo; k < hi; k++) {
            CrSynthRead r;
            cr_synth_read(*p, first + k, r);
            if (h_out->cb) h_out->cb[k] = r.cb;
            if (h_out->umi) h_out->umi[k] = r.umi;
            if (h_out->feature) h_out->feature[k] = r.feature;
            if (h_out->flags) h_out->flags[k] = r.flags;
            if (h_out->cb_qualn)
                for (uint32_t j = 0; j < p->cb_len; j++) h_out->cb_qualn[k * p->cb_len + j] = r.cbq[j];
            if (h_out->umi_qualn)
                for (uint32_t j = 0; j < p->umi_len; j++) h_out->umi_qualn[k * p->umi_len + j] = r.umiq[j];
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    for (auto &t : th) t.join();
    return CRGPU_OK;
}

// ---- Feature Barcoding read rows (synth_core.h: cr_synth_row) -------------------------------------------------------------
__global__ __launch_bounds__(256) void k_synth_rows(uint64_t seed, uint64_t first, uint64_t n, const uint32_t *__restrict__ feature,
                                                    const uint64_t *__restrict__ feat_seq, uint32_t n_feat, uint32_t L, uint32_t offset,
                                                    uint32_t row_stride, uint32_t err, uint32_t n_rate, uint8_t *__restrict__ seq,
                                                    uint8_t *__restrict__ qual) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
        cr_synth_row(seed, first + k, feature ? feature[k] : CRGPU_NO_FEATURE, feat_seq, n_feat, L, offset, row_stride, err, n_rate,
                     seq + k * row_stride, qual + k * row_stride);
}

static int check_rows(crgpu_ctx *ctx, const uint64_t *feat_seq, uint32_t n_feat, uint32_t L, uint32_t offset, uint32_t row_stride) {
    if (!feat_seq || !n_feat || L < 1 || L > 32 || offset + L > row_stride)
        return cr_fail(ctx, CRGPU_EINVAL, "synth rows: %u features of %u bases at offset %u in rows of %u bytes", n_feat, L, offset, row_stride);
    return CRGPU_OK;
}

extern "C" int crgpu_synth_rows_dev(crgpu_ctx *ctx, uint64_t seed, uint64_t first, uint64_t n, const uint32_t *d_feature,
                                    const uint64_t *feat_seq, uint32_t n_feat, uint32_t L, uint32_t offset, uint32_t row_stride,
                                    uint32_t err_per_2_16, uint32_t n_per_2_20, uint8_t *d_seq_rows, uint8_t *d_qual_rows) {
    if (!ctx || !d_seq_rows || !d_qual_rows) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_TRY(check_rows(ctx, feat_seq, n_feat, L, offset, row_stride));
    if (n == 0) return CRGPU_OK;
    cr_invalidate(ctx);
    void *d_fs = nullptr;
    CR_TRY(cr_pool_alloc(ctx, &d_fs, n_feat * sizeof(uint64_t)));
    int rc = crgpu_memcpy_h2d(ctx, d_fs, feat_seq, n_feat * sizeof(uint64_t));
    if (rc == CRGPU_OK) {
        CrTimer t(ctx, CRGPU_T_SYNTH, n);
        hipLaunchKernelGGL(k_synth_rows, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, seed, first, n, d_feature,
                           (const uint64_t *)d_fs, n_feat, L, offset, row_stride, err_per_2_16, n_per_2_20, d_seq_rows, d_qual_rows);
        if (hipGetLastError() != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "k_synth_rows launch failed");
    }
    cr_pool_free(ctx, d_fs);
    return rc;
}

extern "C" int crgpu_synth_rows_host(uint64_t seed, uint64_t first, uint64_t n, const uint32_t *feature, const uint64_t *feat_seq,
                                     uint32_t n_feat, uint32_t L, uint32_t offset, uint32_t row_stride, uint32_t err_per_2_16,
                                     uint32_t n_per_2_20, uint8_t *seq_rows, uint8_t *qual_rows) {
    if (!seq_rows || !qual_rows) return CRGPU_EINVAL;
    CR_TRY(check_rows(nullptr, feat_seq, n_feat, L, offset, row_stride));
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 16) nt = 16;
    if (n < 65536) nt = 1;
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t k = lo; k < hi; k++)
            cr_synth_row(seed, first + k, feature ? feature[k] : CRGPU_NO_FEATURE, feat_seq, n_feat, L, offset, row_stride, err_per_2_16,
                         n_per_2_20, seq_rows + k * row_stride, qual_rows + k * row_stride);
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    for (auto &t : th) t.join();
    return CRGPU_OK;
}
