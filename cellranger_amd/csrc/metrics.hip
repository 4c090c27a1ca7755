// metrics.hip -- MAKE_SHARD's per-read quality metrics over the barcode and UMI parts of the reads, as one fused
// streaming reduction over the packed arrays the hot path already holds.
// Reference: MakeShardVisitor::visit_processed_read, cr_lib/src/make_shard_metrics.rs:263-332; frac_n_bases /
// frac_q30_bases :355-392; thresholds :20-23; RnaRead::barcode_min_qual / umi_min_qual cr_types/src/rna_read.rs:738-749;
// UmiInfo validity umi/src/info.rs:20-37.  PercentMetrics are returned as numerator / denominator counts.
#include "common.h"

#define SM_FIELDS 18

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
        // polyt_suffix_umi_property: the last UMI_POLYT_SUFFIX_LENGTH = 5 bases are T (code 3), none of them an N
        if (umi_len >= 5u) {
            bool n5 = false;
            for (uint32_t k = umi_len - 5u; k < umi_len; k++) n5 |= (umiq[i * umi_len + k] & 0x80u) != 0u;
            a[17] += !n5 && (umi[i] & 0x3FFu) == 0x3FFu;
        }
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

// ------------------------------------------------------------------------------------------------------------------------
// whole-read metrics over FASTQ rows (make_shard_metrics.rs:266-300,355-392)
// ------------------------------------------------------------------------------------------------------------------------
// one wave per row, lanes stride over the bytes (coalesced); every lane adds up over all the rows it touches, one
// reduction at the end
__global__ __launch_bounds__(256) void k_rows_metrics(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                                      const uint32_t *__restrict__ len, uint64_t n, uint32_t stride,
                                                      unsigned long long *__restrict__ out) {
    __shared__ unsigned long long s[4][4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t w0 = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    unsigned long long a[4] = {0, 0, 0, 0};
    for (uint64_t r = w0; r < n; r += nw) {
        uint32_t L = len ? len[r] : stride;
        L = L < stride ? L : stride;
        const uint8_t *sr = seq + r * stride, *qr = qual + r * stride;
        for (uint32_t p = lane; p < L; p += 64) {
            const uint32_t b = sr[p], q = qr[p];
            a[0] += b == 'N';
            a[1] += 1;
            if (q > 2u + 33u) {
                a[3] += 1;
                a[2] += q >= 30u + 33u;
            }
        }
    }
    for (int f = 0; f < 4; f++) {
        unsigned long long x = a[f];
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if (lane == 0) s[wave][f] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long t = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
        if (t) atomicAdd(&out[threadIdx.x * 16], t);
    }
}

// bit x of the result: the row holds run_len equal bases `base` somewhere.  Windows of 64 bytes overlapping by
// run_len - 1; inside a window a ballot of (byte == base) and the shift-and trick for "run_len ones in a row".
__device__ __forceinline__ uint32_t row_homopolymers(const uint8_t *__restrict__ row, uint32_t L, uint32_t run_len, uint32_t lane) {
    uint32_t found = 0;
    if (L < run_len) return 0;
    const uint32_t step = 64u - (run_len - 1u);
    for (uint32_t w = 0; w + run_len <= L; w += step) {
        const uint32_t p = w + lane;
        const uint32_t b = p < L ? row[p] : 0u;
        const char bases[4] = {'A', 'C', 'G', 'T'};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            unsigned long long m = __ballot(b == (uint32_t)bases[k]);
            uint32_t have = 1;  // m marks the starts of runs of `have` ones
            while (m && have < run_len) {
                const uint32_t sh = have < run_len - have ? have : run_len - have;
                m &= m >> sh;
                have += sh;
            }
            if (m) found |= 1u << k;
        }
    }
    return found;
}
__global__ __launch_bounds__(256) void k_homopolymer_metrics(const uint8_t *__restrict__ r1, uint32_t s1, const uint32_t *__restrict__ l1,
                                                             const uint8_t *__restrict__ r2, uint32_t s2, const uint32_t *__restrict__ l2,
                                                             uint64_t n, uint32_t run_len, unsigned long long *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t w0 = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    uint32_t c[4] = {0, 0, 0, 0};
    for (uint64_t r = w0; r < n; r += nw) {
        uint32_t La = l1 ? l1[r] : s1;
        La = La < s1 ? La : s1;
        uint32_t f = row_homopolymers(r1 + r * s1, La, run_len, lane);
        if (r2) {
            uint32_t Lb = l2 ? l2[r] : s2;
            Lb = Lb < s2 ? Lb : s2;
            f |= row_homopolymers(r2 + r * s2, Lb, run_len, lane);
        }
        for (int k = 0; k < 4; k++) c[k] += (f >> k) & 1u;   // uniform over the wave
    }
    if (lane == 0)
        for (int k = 0; k < 4; k++)
            if (c[k]) atomicAdd(&out[k * 16], (unsigned long long)c[k]);
}

static int fetch_counters(crgpu_ctx *ctx, unsigned long long *d_acc, int n_fields, uint64_t *o) {
    unsigned long long h[32 * 16];
    CR_TRY(crgpu_memcpy_d2h(ctx, h, d_acc, (size_t)n_fields * 16 * sizeof(unsigned long long)));
    for (int f = 0; f < n_fields; f++) o[f] = h[f * 16];
    return CRGPU_OK;
}

extern "C" int crgpu_rows_metrics_dev(crgpu_ctx *ctx, const uint8_t *d_seq_rows, const uint8_t *d_qual_rows, const uint32_t *d_len,
                                      uint64_t n, uint32_t row_stride, crgpu_rows_metrics *out) {
    if (!ctx || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    memset(out, 0, sizeof(*out));
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_seq_rows && d_qual_rows && row_stride >= 1, CRGPU_EINVAL, "crgpu_rows_metrics: NULL buffer");
    unsigned long long *d_acc = nullptr;
    CR_TRY(cr_pool_alloc(ctx, (void **)&d_acc, 4 * 16 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_acc, 0, 4 * 16 * sizeof(unsigned long long), ctx->stream);
    {
        CrTimer t(ctx, CRGPU_T_PACK, n);
        hipLaunchKernelGGL(k_rows_metrics, dim3(cr_grid(n, 4, 256u * 8u)), dim3(256), 0, ctx->stream, d_seq_rows, d_qual_rows, d_len, n,
                           row_stride, d_acc);
    }
    if (e == hipSuccess) e = hipGetLastError();
    uint64_t o[4] = {0, 0, 0, 0};
    const int rc = e == hipSuccess ? fetch_counters(ctx, d_acc, 4, o) : CRGPU_EHIP;
    cr_pool_free(ctx, d_acc);
    if (e != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "crgpu_rows_metrics: %s", hipGetErrorString(e));
    CR_TRY(rc);
    out->n_bases = o[0];
    out->bases = o[1];
    out->q30_bases = o[2];
    out->q30_den = o[3];
    return CRGPU_OK;
}

extern "C" int crgpu_homopolymer_metrics_dev(crgpu_ctx *ctx, const uint8_t *d_r1_rows, uint32_t r1_stride, const uint32_t *d_r1_len,
                                             const uint8_t *d_r2_rows, uint32_t r2_stride, const uint32_t *d_r2_len, uint64_t n,
                                             uint32_t run_len, uint64_t *out4) {
    if (!ctx || !out4) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    for (int k = 0; k < 4; k++) out4[k] = 0;
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_r1_rows && r1_stride >= 1 && (!d_r2_rows || r2_stride >= 1), CRGPU_EINVAL, "crgpu_homopolymer_metrics: NULL buffer");
    CR_REQUIRE(ctx, run_len >= 2 && run_len <= 64, CRGPU_ERANGE, "crgpu_homopolymer_metrics: run length 2..64");
    unsigned long long *d_acc = nullptr;
    CR_TRY(cr_pool_alloc(ctx, (void **)&d_acc, 4 * 16 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_acc, 0, 4 * 16 * sizeof(unsigned long long), ctx->stream);
    {
        CrTimer t(ctx, CRGPU_T_PACK, n);
        hipLaunchKernelGGL(k_homopolymer_metrics, dim3(cr_grid(n, 4, 256u * 8u)), dim3(256), 0, ctx->stream, d_r1_rows, r1_stride, d_r1_len,
                           d_r2_rows, r2_stride, d_r2_len, n, run_len, d_acc);
    }
    if (e == hipSuccess) e = hipGetLastError();
    const int rc = e == hipSuccess ? fetch_counters(ctx, d_acc, 4, out4) : CRGPU_EHIP;
    cr_pool_free(ctx, d_acc);
    if (e != hipSuccess) return cr_fail(ctx, CRGPU_EHIP, "crgpu_homopolymer_metrics: %s", hipGetErrorString(e));
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------------
// FASTQ text -> rows
// ------------------------------------------------------------------------------------------------------------------------
#define FQ_TILE 4096u  // bytes per workgroup round: 256 threads x 16 bytes
__global__ __launch_bounds__(256) void k_nl_count(const uint8_t *__restrict__ text, uint64_t n, uint64_t tile, uint32_t *__restrict__ block_counts) {
    __shared__ uint32_t ws[4];
    const uint64_t lo = (uint64_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
    uint32_t c = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) c += text[i] == '\n';
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63u) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// positions of the line feeds, ascending: rounds of 256 consecutive bytes, stable inside a round by ballot ranks
__global__ __launch_bounds__(256) void k_nl_write(const uint8_t *__restrict__ text, uint64_t n, uint64_t tile,
                                                  const uint32_t *__restrict__ block_offs, uint32_t *__restrict__ pos) {
    __shared__ uint32_t ws[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * tile, hi = lo + tile < n ? lo + tile : n;
    uint32_t run = block_offs[blockIdx.x];
    for (uint64_t base = lo; base < hi; base += 256) {
        const uint64_t i = base + threadIdx.x;
        const bool f = i < hi && text[i] == '\n';
        const unsigned long long m = __ballot(f);
        if (lane == 0) ws[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (uint32_t w = 0; w < 4; w++) {
            if (w < wave) pre += ws[w];
            tot += ws[w];
        }
        if (f) pos[run + pre + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
        run += tot;
        __syncthreads();
    }
}
// record r = lines 4r .. 4r+3; line k spans (pos[k-1] + 1 .. pos[k]) (pos[-1] = -1; a missing final line feed = n)
__global__ __launch_bounds__(256) void k_fastq_rows(const uint8_t *__restrict__ text, uint64_t n_bytes, const uint32_t *__restrict__ pos,
                                                    uint64_t n_lines_lf, uint64_t n_records, uint32_t stride, uint8_t *__restrict__ seq,
                                                    uint8_t *__restrict__ qual, uint32_t *__restrict__ len_out, uint32_t *__restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t w0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = w0; r < n_records; r += nw) {
        uint64_t s[4], e[4];
        for (int k = 0; k < 4; k++) {
            const uint64_t ln = 4 * r + k;
            s[k] = ln == 0 ? 0 : (uint64_t)pos[ln - 1] + 1;
            e[k] = ln < n_lines_lf ? pos[ln] : n_bytes;
            if (e[k] > s[k] && text[e[k] - 1] == '\r') e[k]--;  // CRLF
        }
        const uint32_t Ls = (uint32_t)(e[1] - s[1]), Lq = (uint32_t)(e[3] - s[3]);
        if (lane == 0) {
            if (e[0] == s[0] || text[s[0]] != '@' || e[2] == s[2] || text[s[2]] != '+' || Ls != Lq) *bad = 1u;
            if (len_out) len_out[r] = Ls;
        }
        for (uint32_t p = lane; p < stride; p += 64) {
            seq[r * stride + p] = p < Ls ? text[s[1] + p] : 0;
            qual[r * stride + p] = p < Lq ? text[s[3] + p] : 0;
        }
    }
}

int cr_scan_small(crgpu_ctx *ctx, uint32_t *d_data, uint64_t n, uint32_t *d_total_out);

extern "C" int crgpu_fastq_to_rows_dev(crgpu_ctx *ctx, const uint8_t *d_text, uint64_t n_bytes, uint32_t row_stride,
                                       uint64_t max_records, uint8_t *d_seq_rows, uint8_t *d_qual_rows, uint32_t *d_len_out,
                                       uint64_t *n_records_out) {
    if (!ctx || !n_records_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *n_records_out = 0;
    if (n_bytes == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_text && d_seq_rows && d_qual_rows && row_stride >= 1, CRGPU_EINVAL, "crgpu_fastq_to_rows: NULL buffer");
    CR_REQUIRE(ctx, n_bytes < 0xFFFFFFFFull, CRGPU_ERANGE, "crgpu_fastq_to_rows: at most 2^32-2 bytes of text per call");
    cr_invalidate(ctx);
    uint64_t nb = (n_bytes + 4 * FQ_TILE - 1) / (4 * FQ_TILE);
    nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
    uint64_t tile = (n_bytes + nb - 1) / nb;
    tile = (tile + 255) / 256 * 256;
    nb = (n_bytes + tile - 1) / tile;
    uint32_t *d_block = nullptr, *d_pos = nullptr, *d_total = ctx->d_scalars + 16, *d_bad = ctx->d_scalars + 60;
    CR_TRY(cr_pool_alloc(ctx, (void **)&d_block, (nb + 1) * sizeof(uint32_t)));
    uint32_t n_lf = 0;
    int rc = CRGPU_OK;
    {
        CrTimer t(ctx, CRGPU_T_PACK, n_bytes);
        hipLaunchKernelGGL(k_nl_count, dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_text, n_bytes, tile, d_block);
        rc = cr_scan_small(ctx, d_block, nb, d_total);
    }
    if (rc == CRGPU_OK) rc = crgpu_memcpy_d2h(ctx, &n_lf, d_total, sizeof(n_lf));
    uint8_t last = '\n';
    if (rc == CRGPU_OK) rc = crgpu_memcpy_d2h(ctx, &last, d_text + n_bytes - 1, 1);
    const uint64_t n_lines = (uint64_t)n_lf + (last != '\n' ? 1 : 0);
    if (rc == CRGPU_OK && n_lines % 4 != 0)
        rc = cr_fail(ctx, CRGPU_EINVAL, "crgpu_fastq_to_rows: %llu lines are not whole 4-line records", (unsigned long long)n_lines);
    const uint64_t n_rec = n_lines / 4;
    if (rc == CRGPU_OK && n_rec > max_records)
        rc = cr_fail(ctx, CRGPU_ERANGE, "crgpu_fastq_to_rows: %llu records, room for %llu", (unsigned long long)n_rec,
                     (unsigned long long)max_records);
    if (rc == CRGPU_OK) rc = cr_pool_alloc(ctx, (void **)&d_pos, ((uint64_t)n_lf + 1) * sizeof(uint32_t));
    uint32_t bad = 0;
    if (rc == CRGPU_OK) {
        CrTimer t(ctx, CRGPU_T_PACK);
        hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(uint32_t), ctx->stream);
        hipLaunchKernelGGL(k_nl_write, dim3((unsigned)nb), dim3(256), 0, ctx->stream, d_text, n_bytes, tile, d_block, d_pos);
        if (n_rec)
            hipLaunchKernelGGL(k_fastq_rows, dim3(cr_grid(n_rec * 64, 256, 256u * 8u)), dim3(256), 0, ctx->stream, d_text, n_bytes, d_pos,
                               (uint64_t)n_lf, n_rec, row_stride, d_seq_rows, d_qual_rows, d_len_out, d_bad);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) rc = cr_fail(ctx, CRGPU_EHIP, "crgpu_fastq_to_rows: %s", hipGetErrorString(e));
    }
    if (rc == CRGPU_OK) rc = crgpu_memcpy_d2h(ctx, &bad, d_bad, sizeof(bad));
    cr_pool_free(ctx, d_block);
    cr_pool_free(ctx, d_pos);
    CR_TRY(rc);
    CR_REQUIRE(ctx, !bad, CRGPU_EINVAL,
               "crgpu_fastq_to_rows: malformed record (header without '@', separator without '+', or sequence and quality "
               "of different lengths)");
    *n_records_out = n_rec;
    return CRGPU_OK;
}
