// barcode.hip -- pack, K1 (exact match + valid histogram) and K2 (posterior 1-mismatch correction).
//
// K1 replaces Whitelist::check_and_update (barcode/src/whitelist.rs:494-517) per read and
//    MakeShardHistograms::observe (cr_lib/src/make_shard_metrics.rs:171-188).
// K2 replaces Posterior::correct_barcode (barcode/src/corrector.rs:111-165) driven by
//    correct_barcode_in_read (cr_lib/src/stages/barcode_correction.rs:76-99).
//
// Integer / f64 work, HBM- and cache-latency bound: no MFMA.  Built with -ffp-contract=off so the
// f64 multiply and the running sum are never fused (the reference's rustc flags have no +fma,
// lib/rust/.cargo/config.toml:5-8).
#include "common.h"
#include "block_utils.h"
#include "wl_view.h"

int cr_make_views(crgpu_ctx *ctx, WlView *views);

struct WlViewSet {
    WlView v[CRGPU_MAX_LIB];
};

// ------------------------------------------------------------------------------------------------
// pack: ASCII bases + ASCII qualities -> 2-bit word + N-flagged quality bytes
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t base_code(uint32_t c, bool &is_n) {
    // 'A' 0x41 'C' 0x43 'G' 0x47 'T' 0x54 : (c>>1)&3 = 0,1,3,2 ; swap the last two
    uint32_t x = (c >> 1) & 3u;
    x ^= x >> 1;
    is_n = !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
    return is_n ? 0u : x;
}

__global__ __launch_bounds__(256) void k_pack(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                              uint64_t n, uint32_t len, uint32_t *__restrict__ packed,
                                              uint8_t *__restrict__ qualn, uint8_t *__restrict__ flags) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t key = 0;
        bool any_n = false;
        if (len == 16) {
            const uint4 s4 = *reinterpret_cast<const uint4 *>(seq + i * 16);
            const uint4 q4 = *reinterpret_cast<const uint4 *>(qual + i * 16);
            const uint32_t sw[4] = {s4.x, s4.y, s4.z, s4.w};
            uint32_t qw[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t outq = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t c = (sw[w] >> (8 * b)) & 0xFFu;
                    uint32_t q = (qw[w] >> (8 * b)) & 0xFFu;
                    bool is_n;
                    const uint32_t code = base_code(c, is_n);
                    key = (key << 2) | code;
                    any_n |= is_n;
                    q = q > 127u ? 127u : q;
                    outq |= (q | (is_n ? 0x80u : 0u)) << (8 * b);
                }
                qw[w] = outq;
            }
            *reinterpret_cast<uint4 *>(qualn + i * 16) = make_uint4(qw[0], qw[1], qw[2], qw[3]);
        } else {
            for (uint32_t j = 0; j < len; j++) {
                const uint32_t c = seq[i * len + j];
                uint32_t q = qual[i * len + j];
                bool is_n;
                const uint32_t code = base_code(c, is_n);
                key = (key << 2) | code;
                any_n |= is_n;
                q = q > 127u ? 127u : q;
                qualn[i * len + j] = (uint8_t)(q | (is_n ? 0x80u : 0u));
            }
        }
        packed[i] = key;
        if (flags && any_n) flags[i] |= CRGPU_FLAG_CB_HAS_N;
    }
}

extern "C" int crgpu_pack_dev(crgpu_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n, uint32_t len,
                              uint32_t *d_packed_out, uint8_t *d_qualn_out, uint8_t *d_flags_inout) {
    if (!ctx) return CRGPU_EINVAL;
    CR_REQUIRE(ctx, len >= 1 && len <= 16, CRGPU_ERANGE, "sequence length %u unsupported (<= 16)", len);
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_seq && d_qual && d_packed_out && d_qualn_out, CRGPU_EINVAL, "crgpu_pack_dev: NULL buffer");
    if (len == 16)
        CR_REQUIRE(ctx, ((uintptr_t)d_seq | (uintptr_t)d_qual | (uintptr_t)d_qualn_out) % 16 == 0, CRGPU_EINVAL,
                   "crgpu_pack_dev: 16-base buffers must be 16-byte aligned");
    CrTimer t(ctx, CRGPU_T_PACK, n);
    hipLaunchKernelGGL(k_pack, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, d_seq, d_qual, n, len, d_packed_out,
                       d_qualn_out, d_flags_inout);
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// K1: exact match + valid-barcode histogram
// ------------------------------------------------------------------------------------------------
template <bool UNIFORM>
__global__ __launch_bounds__(256) void k_match(const WlViewSet vs, const uint32_t *__restrict__ cb,
                                               const uint8_t *__restrict__ flags, uint64_t n,
                                               uint32_t *__restrict__ idx_out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t key = cb[i];
        const uint32_t f = flags ? flags[i] : 0u;
        uint32_t rank = CRGPU_MISS;
        const uint32_t lib = f & CRGPU_FLAG_LIB_MASK;
        if (!(f & CRGPU_FLAG_CB_HAS_N)) {
            if (UNIFORM) {
                if (lib == 0) rank = wl_lookup(vs.v[0], key);
            } else {
                if (vs.v[lib].n) rank = wl_lookup(vs.v[lib], key);
            }
        }
        idx_out[i] = rank;
        if (rank != CRGPU_MISS) atomicAdd(UNIFORM ? &vs.v[0].valid[rank] : &vs.v[lib].valid[rank], 1u);
    }
}

static bool uniform_lib0(const crgpu_ctx *ctx) {
    if (!ctx->wl[0].set) return false;
    for (int l = 1; l < CRGPU_MAX_LIB; l++)
        if (ctx->wl[l].set) return false;
    return true;
}

extern "C" int crgpu_match_and_count_dev(crgpu_ctx *ctx, const uint32_t *d_cb, const uint8_t *d_flags, uint64_t n,
                                         uint32_t *d_idx_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_match_and_count: no whitelist set");
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_cb && d_idx_out, CRGPU_EINVAL, "crgpu_match_and_count: NULL buffer");
    WlViewSet vs;
    CR_TRY(cr_make_views(ctx, vs.v));
    CrTimer t(ctx, CRGPU_T_MATCH, n);
    const dim3 grid(cr_grid(n, 256)), block(256);
    if (uniform_lib0(ctx))
        hipLaunchKernelGGL(k_match<true>, grid, block, 0, ctx->stream, vs, d_cb, d_flags, n, d_idx_out);
    else
        hipLaunchKernelGGL(k_match<false>, grid, block, 0, ctx->stream, vs, d_cb, d_flags, n, d_idx_out);
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// K2: posterior correction of the reads that missed
// ------------------------------------------------------------------------------------------------
#define MISS_ITEMS 8
// Compact the indices of the reads that missed.  One global atomic per 2048-read chunk: a single hot
// counter serialises, so the reservation is aggregated over the whole workgroup.
__global__ __launch_bounds__(256) void k_collect_miss(const uint32_t *__restrict__ idx, uint64_t n,
                                                      uint32_t *__restrict__ miss_list,
                                                      unsigned long long *__restrict__ n_miss) {
    __shared__ __attribute__((aligned(8))) uint32_t lds[10];
    const uint64_t chunk = 256ull * MISS_ITEMS;
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    for (uint64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        uint32_t mask = 0;
#pragma unroll
        for (int j = 0; j < MISS_ITEMS; j++) {
            const uint64_t i = c * chunk + (uint64_t)j * 256 + threadIdx.x;
            if (i < n && idx[i] == CRGPU_MISS) mask |= 1u << j;
        }
        unsigned long long o = block_reserve_256((uint32_t)__popc(mask), n_miss, lds);
#pragma unroll
        for (int j = 0; j < MISS_ITEMS; j++)
            if (mask & (1u << j)) miss_list[o++] = (uint32_t)(c * chunk + (uint64_t)j * 256 + threadIdx.x);
    }
}

// exactly one differing 2-bit group between a and b (both < 2^16)?  returns the bit offset of that
// group (even) or -1.
__device__ __forceinline__ int one_base_diff(uint32_t a, uint32_t b) {
    const uint32_t x = a ^ b;
    const uint32_t y = (x | (x >> 1)) & 0x5555u;
    if (y == 0u || (y & (y - 1u)) != 0u) return -1;
    return __ffs((int)y) - 1;
}

template <bool UNIFORM>
__global__ __launch_bounds__(256) void k_correct(const WlViewSet vs, const uint32_t *__restrict__ cb,
                                                 const uint8_t *__restrict__ qualn, const uint8_t *__restrict__ flags,
                                                 const uint32_t *__restrict__ miss_list,
                                                 const unsigned long long *__restrict__ n_miss_ptr, uint32_t len,
                                                 const double *__restrict__ ptab, double max_expected, double thresh,
                                                 bool check_expected,
                                                 uint32_t *__restrict__ idx_inout, uint8_t *__restrict__ corrected_out) {
    const uint32_t n_miss = (uint32_t)*n_miss_ptr;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_miss; j += stride) {
        const uint64_t i = miss_list[j];
        const uint32_t key = cb[i];
        const uint32_t f = flags ? flags[i] : 0u;
        const uint32_t lib = UNIFORM ? 0u : (f & CRGPU_FLAG_LIB_MASK);
        if (UNIFORM && (f & CRGPU_FLAG_LIB_MASK) != 0u) continue;
        const WlView &w = vs.v[lib];
        if (!UNIFORM && w.n == 0) continue;

        // qualities (bit 7 = N), kept in two 64-bit registers: byte k of (qlo,qhi) = position k
        unsigned long long qlo, qhi;
        if (qualn) {
            if (len == 16) {
                const uint4 q4 = *reinterpret_cast<const uint4 *>(qualn + i * 16);
                qlo = (unsigned long long)q4.x | ((unsigned long long)q4.y << 32);
                qhi = (unsigned long long)q4.z | ((unsigned long long)q4.w << 32);
            } else {
                qlo = 0ull;
                qhi = 0ull;
                for (uint32_t k = 0; k < len; k++) {
                    const unsigned long long b = qualn[i * len + k];
                    if (k < 8) qlo |= b << (8 * k); else qhi |= b << (8 * (k - 8));
                }
            }
        } else {
            qlo = qhi = 0x4242424242424242ull;      // BC_MAX_QV = 66, corrector.rs:126 map_or
            if (f & CRGPU_FLAG_CB_HAS_N) continue;   // N position unknown without qualities
        }
        // movemask of the N bits: bit k = position k
        const uint32_t nmask = (uint32_t)(((qlo & 0x8080808080808080ull) * 0x0002040810204081ull) >> 56) |
                               ((uint32_t)(((qhi & 0x8080808080808080ull) * 0x0002040810204081ull) >> 56) << 8);
        const int n_n = __popc(nmask);

        // candidate slots: bit (pos*4 + base)
        unsigned long long cand = 0ull;
        if (n_n == 0) {
            const uint32_t head = key >> w.bitsB;
            const uint32_t tail = key & ((1u << w.bitsB) - 1u);
            const uint32_t hA = w.bitsA >> 1;
            // mutation in the tail: same head -> bin A
            for (uint32_t p = w.offA[head], e = w.offA[head + 1]; p < e; ++p) {
                const uint32_t t = w.tailA[p];
                const int bo = one_base_diff(t, tail);
                if (bo >= 0) {
                    const uint32_t pos = len - 1u - (uint32_t)(bo >> 1);
                    cand |= 1ull << (pos * 4u + ((t >> bo) & 3u));
                }
            }
            // mutation in the head: same tail -> bin B
            for (uint32_t p = w.offB[tail], e = w.offB[tail + 1]; p < e; ++p) {
                const uint32_t h = w.headB[p];
                const int bo = one_base_diff(h, head);
                if (bo >= 0) {
                    const uint32_t pos = hA - 1u - (uint32_t)(bo >> 1);
                    cand |= 1ull << (pos * 4u + ((h >> bo) & 3u));
                }
            }
        } else if (n_n == 1) {
            // the N is "observed": all four bases are tried at its position (corrector.rs:128-131);
            // a candidate built at any other position still contains the N and cannot match.
            const uint32_t pos = (uint32_t)__ffs((int)nmask) - 1u;
            cand = 0xFull << (pos * 4u);
        }

        bool have_best = false;
        double best_like = 0.0, total = 0.0;
        uint32_t best_rank = 0;
        while (cand) {
            const uint32_t slot = (uint32_t)__ffsll((long long)cand) - 1u;
            cand &= cand - 1ull;
            const uint32_t pos = slot >> 2, base = slot & 3u;
            const uint32_t sh = 2u * (len - 1u - pos);
            const uint32_t ckey = (key & ~(3u << sh)) | (base << sh);
            const uint32_t r = wl_lookup(w, ckey);
            if (r == CRGPU_MISS) continue;
            uint32_t qv = (uint32_t)((pos < 8u ? qlo : qhi) >> (8u * (pos & 7u))) & 0x7Fu;
            qv = qv < 66u ? qv : 66u;                                  // corrector.rs:126
            const long long bc_count = 1ll + (long long)w.prior[r];    // Laplace smoothing, :138-139
            const double like = ptab[qv] * (double)bc_count;           // :140-141
            if (!have_best) {
                have_best = true;
                best_like = like;
                best_rank = r;
            } else if (like > best_like || (like == best_like && r >= best_rank)) {
                // Ord::max on (NotNan, BarcodeSegment): ties go to the larger sequence == larger rank
                best_like = like;
                best_rank = r;
            }
            total += like;  // pos-major, A<C<G<T order (:146)
        }
        if (!have_best) continue;
        double expected = 0.0;  // :154, uncapped qualities, in order; 0.0 without qualities
        if (check_expected)
            for (uint32_t k = 0; k < len; k++)
                expected += ptab[(uint32_t)((k < 8u ? qlo : qhi) >> (8u * (k & 7u))) & 0x7Fu];
        if (expected < max_expected && best_like / total >= thresh) {
            idx_inout[i] = best_rank;
            if (corrected_out) corrected_out[i] = 1;
            atomicAdd(&w.corrected[best_rank], 1u);
        }
    }
}

extern "C" int crgpu_set_posterior(crgpu_ctx *ctx, double max_expected_barcode_errors, double bc_confidence_threshold) {
    if (!ctx) return CRGPU_EINVAL;
    ctx->max_expected_errors = max_expected_barcode_errors;
    ctx->confidence_threshold = bc_confidence_threshold;
    return CRGPU_OK;
}

// fake_quals: the quality bytes only carry the N flags (host path without qualities): the
// expected-error sum is 0.0 as in corrector.rs:154 map_or.
static int correct_dev_impl(crgpu_ctx *ctx, const uint32_t *d_cb, const uint8_t *d_qualn, const uint8_t *d_flags,
                            uint64_t n, uint32_t *d_idx_inout, uint8_t *d_corrected_out, bool fake_quals) {
    if (!ctx) return CRGPU_EINVAL;
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_correct: no whitelist set");
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_cb && d_idx_inout, CRGPU_EINVAL, "crgpu_correct: NULL buffer");
    CR_REQUIRE(ctx, n < 0xFFFFFFFFull, CRGPU_ERANGE, "crgpu_correct: batches are limited to 2^32-2 reads");
    if (ctx->cb_len == 16 && d_qualn)
        CR_REQUIRE(ctx, (uintptr_t)d_qualn % 16 == 0, CRGPU_EINVAL, "crgpu_correct: quality buffer must be 16-byte aligned");
    // NotNan::try_from(threshold).ok()? (corrector.rs:152): a NaN threshold corrects nothing
    if (ctx->confidence_threshold != ctx->confidence_threshold) return CRGPU_OK;
    WlViewSet vs;
    CR_TRY(cr_make_views(ctx, vs.v));
    void *ws;
    CR_TRY(cr_scratch(ctx, n * sizeof(uint32_t), &ws));
    uint32_t *miss_list = (uint32_t *)ws;
    unsigned long long *n_miss = (unsigned long long *)ctx->d_scalars;
    CrTimer t(ctx, CRGPU_T_CORRECT, n);
    CR_HIP(ctx, hipMemsetAsync(n_miss, 0, sizeof(unsigned long long), ctx->stream));
    if (d_corrected_out) CR_HIP(ctx, hipMemsetAsync(d_corrected_out, 0, n, ctx->stream));
    hipLaunchKernelGGL(k_collect_miss, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, d_idx_inout, n, miss_list, n_miss);
    // K2 is launched for the worst case and loops over the device-side count: no host round trip
    const dim3 grid(cr_grid(n / 8 + 1, 256)), block(256);
    // expected_errors < f64::MAX is always true for a finite sum: skip the sum for the default
    const bool check_expected = d_qualn && !fake_quals && ctx->max_expected_errors < 1.7976931348623157e308;
    if (uniform_lib0(ctx))
        hipLaunchKernelGGL(k_correct<true>, grid, block, 0, ctx->stream, vs, d_cb, d_qualn, d_flags, miss_list, n_miss,
                           ctx->cb_len, ctx->d_ptab, ctx->max_expected_errors, ctx->confidence_threshold, check_expected,
                           d_idx_inout, d_corrected_out);
    else
        hipLaunchKernelGGL(k_correct<false>, grid, block, 0, ctx->stream, vs, d_cb, d_qualn, d_flags, miss_list, n_miss,
                           ctx->cb_len, ctx->d_ptab, ctx->max_expected_errors, ctx->confidence_threshold, check_expected,
                           d_idx_inout, d_corrected_out);
    CR_HIP(ctx, hipGetLastError());
    return CRGPU_OK;
}

extern "C" int crgpu_correct_dev(crgpu_ctx *ctx, const uint32_t *d_cb, const uint8_t *d_qualn, const uint8_t *d_flags,
                                 uint64_t n, uint32_t *d_idx_inout, uint8_t *d_corrected_out) {
    return correct_dev_impl(ctx, d_cb, d_qualn, d_flags, n, d_idx_inout, d_corrected_out, false);
}

// ------------------------------------------------------------------------------------------------
// host-buffer convenience entry points (SURVEY.md 8b signatures)
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_u8(uint8_t *p, uint64_t n, uint8_t v) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

struct HostBatch {
    uint8_t *d_seq = nullptr, *d_qual = nullptr, *d_qualn = nullptr, *d_flags = nullptr;
    uint32_t *d_cb = nullptr, *d_idx = nullptr;
    uint8_t *d_corr = nullptr;
    ~HostBatch() {
        hipFree(d_seq);
        hipFree(d_qual);
        hipFree(d_qualn);
        hipFree(d_flags);
        hipFree(d_cb);
        hipFree(d_idx);
        hipFree(d_corr);
    }
};

static int stage_host_batch(crgpu_ctx *ctx, int lib, const uint8_t *seq, const uint8_t *qual, uint64_t n, HostBatch &b) {
    const uint32_t len = ctx->cb_len;
    CR_REQUIRE(ctx, lib >= 0 && lib < CRGPU_MAX_LIB && ctx->wl[lib].set, CRGPU_ESTATE, "library %d has no whitelist", lib);
    CR_HIP(ctx, hipMalloc((void **)&b.d_seq, n * len));
    CR_HIP(ctx, hipMalloc((void **)&b.d_qual, n * len));
    CR_HIP(ctx, hipMalloc((void **)&b.d_qualn, n * len));
    CR_HIP(ctx, hipMalloc((void **)&b.d_flags, n));
    CR_HIP(ctx, hipMalloc((void **)&b.d_cb, n * sizeof(uint32_t)));
    CR_HIP(ctx, hipMalloc((void **)&b.d_idx, n * sizeof(uint32_t)));
    CR_HIP(ctx, hipMalloc((void **)&b.d_corr, n));
    CR_HIP(ctx, hipMemcpyAsync(b.d_seq, seq, n * len, hipMemcpyHostToDevice, ctx->stream));
    if (qual) {
        CR_HIP(ctx, hipMemcpyAsync(b.d_qual, qual, n * len, hipMemcpyHostToDevice, ctx->stream));
    } else {
        // no qualities (corrector.rs:126 map_or): every base gets BC_MAX_QV
        hipLaunchKernelGGL(k_fill_u8, dim3(cr_grid(n * len, 256)), dim3(256), 0, ctx->stream, b.d_qual, n * len, (uint8_t)66);
    }
    hipLaunchKernelGGL(k_fill_u8, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, b.d_flags, n, (uint8_t)lib);
    CR_TRY(crgpu_pack_dev(ctx, b.d_seq, b.d_qual, n, len, b.d_cb, b.d_qualn, b.d_flags));
    return CRGPU_OK;
}

extern "C" int crgpu_match_and_count(crgpu_ctx *ctx, int lib, const uint8_t *seq, const uint8_t *qual, uint64_t n,
                                     uint32_t *idx_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_match_and_count: no whitelist set");
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, seq && idx_out, CRGPU_EINVAL, "crgpu_match_and_count: NULL buffer");
    HostBatch b;
    CR_TRY(stage_host_batch(ctx, lib, seq, qual, n, b));
    CR_TRY(crgpu_match_and_count_dev(ctx, b.d_cb, b.d_flags, n, b.d_idx));
    CR_TRY(crgpu_memcpy_d2h(ctx, idx_out, b.d_idx, n * sizeof(uint32_t)));
    return CRGPU_OK;
}

extern "C" int crgpu_correct(crgpu_ctx *ctx, int lib, const uint8_t *seq, const uint8_t *qual, uint64_t n,
                             uint32_t *idx_inout, uint8_t *corrected_flag_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_correct: no whitelist set");
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, seq && idx_inout, CRGPU_EINVAL, "crgpu_correct: NULL buffer");
    HostBatch b;
    CR_TRY(stage_host_batch(ctx, lib, seq, qual, n, b));
    CR_HIP(ctx, hipMemcpyAsync(b.d_idx, idx_inout, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    CR_TRY(correct_dev_impl(ctx, b.d_cb, b.d_qualn, b.d_flags, n, b.d_idx, b.d_corr, qual == nullptr));
    CR_TRY(crgpu_memcpy_d2h(ctx, idx_inout, b.d_idx, n * sizeof(uint32_t)));
    if (corrected_flag_out) CR_TRY(crgpu_memcpy_d2h(ctx, corrected_flag_out, b.d_corr, n));
    return CRGPU_OK;
}
