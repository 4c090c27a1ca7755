// ctx.hip -- context, memory helpers and the HIP-event timing ledger of libcrgpu.
#include <cmath>

#include <cstdio>
#include <cstdlib>

#include "common.h"

static thread_local std::string g_thread_err;

void cr_set_thread_error(const char *msg) { g_thread_err = msg; }

int cr_fail(crgpu_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_thread_err = buf;
    return code;
}

extern "C" int crgpu_abi_version(void) { return CRGPU_ABI_VERSION; }

// ---- layout of the public structs, for bindings to check their own declarations against (include/crgpu.h) -------------
#include <cstddef>
namespace {
struct AbiField {
    uint32_t offset, size;
};
struct AbiStruct {
    const char *name;
    uint32_t size, align;
    std::vector<AbiField> fields;
};
#define ABI_F(T, f) AbiField{(uint32_t)offsetof(T, f), (uint32_t)sizeof(((T *)nullptr)->f)}
#define ABI_S(T, ...) AbiStruct{#T, (uint32_t)sizeof(T), (uint32_t)alignof(T), {__VA_ARGS__}}
const std::vector<AbiStruct> &abi_table() {
    static const std::vector<AbiStruct> t = {
        ABI_S(crgpu_records, ABI_F(crgpu_records, n), ABI_F(crgpu_records, umi_len), ABI_F(crgpu_records, d_bc_idx),
              ABI_F(crgpu_records, d_umi), ABI_F(crgpu_records, d_umi_qualn), ABI_F(crgpu_records, d_feature),
              ABI_F(crgpu_records, d_flags), ABI_F(crgpu_records, d_umi_len), ABI_F(crgpu_records, d_probe_idx)),
        ABI_S(crgpu_matrix, ABI_F(crgpu_matrix, n_barcodes), ABI_F(crgpu_matrix, nnz), ABI_F(crgpu_matrix, n_features),
              ABI_F(crgpu_matrix, cb_len), ABI_F(crgpu_matrix, barcode_rank), ABI_F(crgpu_matrix, barcode_seq),
              ABI_F(crgpu_matrix, indptr), ABI_F(crgpu_matrix, indices), ABI_F(crgpu_matrix, data), ABI_F(crgpu_matrix, gem_group),
              ABI_F(crgpu_matrix, barcode_seq_hi)),
        ABI_S(crgpu_matrix_dev, ABI_F(crgpu_matrix_dev, n_barcodes), ABI_F(crgpu_matrix_dev, nnz),
              ABI_F(crgpu_matrix_dev, d_barcode_rank), ABI_F(crgpu_matrix_dev, d_indptr), ABI_F(crgpu_matrix_dev, d_indices),
              ABI_F(crgpu_matrix_dev, d_data)),
        ABI_S(crgpu_dupinfo, ABI_F(crgpu_dupinfo, processed_umi), ABI_F(crgpu_dupinfo, read_count), ABI_F(crgpu_dupinfo, flags),
              ABI_F(crgpu_dupinfo, reserved)),
        ABI_S(crgpu_barcode_summary_row, ABI_F(crgpu_barcode_summary_row, barcode_rank), ABI_F(crgpu_barcode_summary_row, library),
              ABI_F(crgpu_barcode_summary_row, reads), ABI_F(crgpu_barcode_summary_row, umis),
              ABI_F(crgpu_barcode_summary_row, candidate_dup_reads), ABI_F(crgpu_barcode_summary_row, umi_corrected_reads)),
        ABI_S(crgpu_bc_correction_metrics, ABI_F(crgpu_bc_correction_metrics, valid_reads),
              ABI_F(crgpu_bc_correction_metrics, corrected_reads), ABI_F(crgpu_bc_correction_metrics, barcodes_detected),
              ABI_F(crgpu_bc_correction_metrics, effective_barcode_diversity)),
        ABI_S(crgpu_shard_metrics, ABI_F(crgpu_shard_metrics, sequenced_reads), ABI_F(crgpu_shard_metrics, bc_n_bases),
              ABI_F(crgpu_shard_metrics, bc_bases), ABI_F(crgpu_shard_metrics, umi_n_bases), ABI_F(crgpu_shard_metrics, umi_bases),
              ABI_F(crgpu_shard_metrics, bc_q30_bases), ABI_F(crgpu_shard_metrics, bc_q30_den), ABI_F(crgpu_shard_metrics, umi_q30_bases),
              ABI_F(crgpu_shard_metrics, umi_q30_den), ABI_F(crgpu_shard_metrics, good_umi), ABI_F(crgpu_shard_metrics, has_n_barcode),
              ABI_F(crgpu_shard_metrics, has_n_umi), ABI_F(crgpu_shard_metrics, homopolymer_barcode),
              ABI_F(crgpu_shard_metrics, homopolymer_umi), ABI_F(crgpu_shard_metrics, low_min_qual_barcode),
              ABI_F(crgpu_shard_metrics, low_min_qual_umi), ABI_F(crgpu_shard_metrics, miss_whitelist_barcode),
              ABI_F(crgpu_shard_metrics, polyt_suffix_umi)),
        ABI_S(crgpu_rows_metrics, ABI_F(crgpu_rows_metrics, n_bases), ABI_F(crgpu_rows_metrics, bases),
              ABI_F(crgpu_rows_metrics, q30_bases), ABI_F(crgpu_rows_metrics, q30_den)),
        ABI_S(crgpu_feature_def, ABI_F(crgpu_feature_def, pattern), ABI_F(crgpu_feature_def, sequence),
              ABI_F(crgpu_feature_def, index), ABI_F(crgpu_feature_def, read)),
        ABI_S(crgpu_synth_params, ABI_F(crgpu_synth_params, seed), ABI_F(crgpu_synth_params, cb_len), ABI_F(crgpu_synth_params, umi_len),
              ABI_F(crgpu_synth_params, n_wl), ABI_F(crgpu_synth_params, wl_packed), ABI_F(crgpu_synth_params, n_cells),
              ABI_F(crgpu_synth_params, cell_wl_pos), ABI_F(crgpu_synth_params, cell_cdf), ABI_F(crgpu_synth_params, n_ambient),
              ABI_F(crgpu_synth_params, ambient_wl_pos), ABI_F(crgpu_synth_params, n_genes), ABI_F(crgpu_synth_params, gene_cdf),
              ABI_F(crgpu_synth_params, ambient_per_2_16), ABI_F(crgpu_synth_params, cb_err_per_2_16),
              ABI_F(crgpu_synth_params, umi_err_per_2_16), ABI_F(crgpu_synth_params, n_per_2_20),
              ABI_F(crgpu_synth_params, no_feature_per_2_16), ABI_F(crgpu_synth_params, reads_per_umi),
              ABI_F(crgpu_synth_params, n_total), ABI_F(crgpu_synth_params, n_libs)),
        ABI_S(crgpu_synth_out, ABI_F(crgpu_synth_out, cb), ABI_F(crgpu_synth_out, cb_qualn), ABI_F(crgpu_synth_out, umi),
              ABI_F(crgpu_synth_out, umi_qualn), ABI_F(crgpu_synth_out, feature), ABI_F(crgpu_synth_out, flags)),
    };
    return t;
}
}  // namespace

extern "C" int crgpu_abi_layout(const char *struct_name, uint32_t *out, uint32_t cap) {
    if (!struct_name) return CRGPU_EINVAL;
    for (const AbiStruct &s : abi_table()) {
        if (strcmp(s.name, struct_name) != 0) continue;
        std::vector<uint32_t> w = {s.size, s.align, (uint32_t)s.fields.size()};
        for (const AbiField &f : s.fields) {
            w.push_back(f.offset);
            w.push_back(f.size);
        }
        for (uint32_t i = 0; i < w.size() && i < cap && out; i++) out[i] = w[i];
        return (int)w.size();
    }
    return cr_fail(nullptr, CRGPU_EINVAL, "crgpu_abi_layout: no public struct named %s", struct_name);
}

extern "C" const char *crgpu_last_error(const crgpu_ctx *ctx) {
    return ctx ? ctx->err.c_str() : g_thread_err.c_str();
}

extern "C" int crgpu_create(crgpu_ctx **out, int device_id, int n_ranks, int rank, const void *unique_id) {
    if (!out) return cr_fail(nullptr, CRGPU_EINVAL, "crgpu_create: out is NULL");
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return cr_fail(nullptr, CRGPU_EINVAL, "crgpu_create: rank %d of %d ranks", rank, n_ranks);
    if (n_ranks > 1 && !unique_id)
        return cr_fail(nullptr, CRGPU_EINVAL, "crgpu_create: %d ranks need a unique_id (crgpu_get_unique_id / crgpu_local_group_id)",
                       n_ranks);
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return cr_fail(nullptr, CRGPU_ENODEV,
                       "crgpu_create: no HIP device visible (%s); libcrgpu has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device_id < 0 || device_id >= n_dev)
        return cr_fail(nullptr, CRGPU_EINVAL, "crgpu_create: device_id %d out of range [0,%d)", device_id, n_dev);
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return cr_fail(nullptr, CRGPU_ENODEV, "hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return cr_fail(nullptr, CRGPU_ENODEV, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return cr_fail(nullptr, CRGPU_ENODEV, "device %d is %s; libcrgpu is built for gfx950 (MI355X) only", device_id,
                       prop.gcnArchName);

    crgpu_ctx *ctx = new (std::nothrow) crgpu_ctx();
    if (!ctx) return cr_fail(nullptr, CRGPU_ENOMEM, "crgpu_create: out of host memory");
    ctx->device = device_id;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return cr_fail(nullptr, CRGPU_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    // the second stream carries the branch of the count stage that is NOT the critical path (candidate filter + hash sort beside
    // the UMI correction): lowest priority, so that its kernels fill what the main stream leaves (CRGPU_STREAM2_PRIORITY=same: A/B)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const char *sp = getenv("CRGPU_STREAM2_PRIORITY");
    const bool low2 = !(sp && strcmp(sp, "same") == 0);
    if ((low2 ? hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_least)
              : hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking)) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();  // the count stage then stays on the one stream
        if (ctx->stream2) hipStreamDestroy(ctx->stream2);
        ctx->stream2 = nullptr;
    }
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) ctx->pool_budget = (uint64_t)total_b / 2;
    }
    // probability(q) = 10^(-(q-33)/10) computed on the HOST with libm pow, exactly as the
    // reference does per call (corrector.rs:167-171), for every 7-bit quality character.
    double ptab[128];
    for (int q = 0; q < 128; q++) ptab[q] = std::pow(10.0, -((double)q - 33.0) / 10.0);
    if (hipMalloc(&ctx->d_ptab, sizeof(ptab)) != hipSuccess ||
        hipMemcpy(ctx->d_ptab, ptab, sizeof(ptab), hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&ctx->d_scalars, 4096) != hipSuccess || hipMemset(ctx->d_scalars, 0, 4096) != hipSuccess ||
        hipMalloc(&ctx->d_sort_hist, sizeof(uint32_t) * (256 * 2048 + 512)) != hipSuccess) {
        crgpu_destroy(ctx);
        return cr_fail(nullptr, CRGPU_ENOMEM, "crgpu_create: device allocation failed");
    }
    if (unique_id) {
        // blocks until every rank has arrived; the message of a failure is kept for crgpu_last_error(NULL)
        const int rc = cr_comm_init(ctx, n_ranks, rank, unique_id);
        if (rc != CRGPU_OK) {
            const std::string msg = ctx->err;
            crgpu_destroy(ctx);
            return cr_fail(nullptr, rc, "%s", msg.c_str());
        }
    }
    *out = ctx;
    return CRGPU_OK;
}

extern "C" int crgpu_comm_info(crgpu_ctx *ctx, uint32_t *n_ranks_out, uint32_t *rank_out) {
    if (!ctx) return CRGPU_EINVAL;
    if (n_ranks_out) *n_ranks_out = (uint32_t)ctx->n_ranks;
    if (rank_out) *rank_out = (uint32_t)ctx->rank;
    return CRGPU_OK;
}

extern "C" int crgpu_set_option(crgpu_ctx *ctx, int option, int64_t value) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    switch (option) {
        case CRGPU_OPT_BUFFERS_UNCHANGED_BETWEEN_CALLS:
            ctx->trust_buffers = value != 0;
            cr_invalidate(ctx);
            return CRGPU_OK;
        case CRGPU_OPT_DENSE_BARCODE_KEYS:
            cr_invalidate(ctx);
            cr_dense_drop(ctx);
            ctx->dense.on = value != 0;
            return CRGPU_OK;
        default:
            return cr_fail(ctx, CRGPU_EINVAL, "crgpu_set_option: unknown option %d", option);
    }
}

extern "C" int crgpu_get_stat(crgpu_ctx *ctx, int which, uint64_t *value_out) {
    if (!ctx || !value_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    switch (which) {
        case CRGPU_STAT_SORT_FALLBACKS:
            *value_out = ctx->sort_fallbacks;
            return CRGPU_OK;
        case CRGPU_STAT_SORT_REFINISHED:
            *value_out = ctx->sort_refinished;
            return CRGPU_OK;
        case CRGPU_STAT_K1_SPLIT_ROUNDS:
            *value_out = ctx->k1_split_rounds;
            return CRGPU_OK;
        case CRGPU_STAT_COMM_BYTES_C1:
        case CRGPU_STAT_COMM_BYTES_C2:
        case CRGPU_STAT_COMM_BYTES_C3:
            *value_out = ctx->comm_bytes[which - CRGPU_STAT_COMM_BYTES_C1];
            return CRGPU_OK;
        case CRGPU_STAT_FEATURE_RESUMED_READS:
            *value_out = ctx->feature_resumed_reads;
            return CRGPU_OK;
        case CRGPU_STAT_FEATURE_FAST_LAUNCHES:
            *value_out = ctx->feature_fast_launches;
            return CRGPU_OK;
        case CRGPU_STAT_MISS_RECORD_SETS: {
            uint64_t k = 0;
            for (const MissRecords &r : ctx->recs) k += r.valid ? 1u : 0u;
            *value_out = k;
            return CRGPU_OK;
        }
        case CRGPU_STAT_DISTINCT_KEYS:
            *value_out = ctx->last_distinct_keys;
            return CRGPU_OK;
        case CRGPU_STAT_LOW_SUPPORT_CANDIDATES:
            *value_out = ctx->last_low_support_candidates;
            return CRGPU_OK;
        case CRGPU_STAT_FEATURE_READS_REQUEUED:
            *value_out = ctx->feature_reads_requeued;
            return CRGPU_OK;
        default:
            return cr_fail(ctx, CRGPU_EINVAL, "crgpu_get_stat: unknown counter %d", which);
    }
}

void cr_invalidate(crgpu_ctx *ctx) {
    cr_drop_miss_records(ctx);
    cr_drop_feature_pending(ctx);
    ctx->ghist.valid = false;
}
// the same for a call that writes [p, p + bytes) of the caller's memory and nothing else: only the by-products that describe
// (a buffer overlapping) that range go -- the feature extraction between pass A and pass B writes the feature indices, not the
// barcodes pass A's miss records are about
void cr_invalidate_range(crgpu_ctx *ctx, const void *p, uint64_t bytes) {
    if (!p || !bytes) return;
    const uintptr_t a = (uintptr_t)p, b = a + bytes;
    auto hits = [&](const void *q, uint64_t len) { return q && (uintptr_t)q < b && (uintptr_t)q + len > a; };
    for (MissRecords &r : ctx->recs)
        if (r.valid && (hits(r.d_cb, r.n * 4) || hits(r.d_flags, r.n) || hits(r.d_idx, r.n * 4))) cr_drop_miss_records(ctx, r);
    const FxPendingSet &f = ctx->fxp;
    // (rows: at most stride bytes each; the outputs: 4 bytes per read and id / capture -- bounded generously)
    if (f.valid && (hits(f.d_seq, f.n * (uint64_t)(f.stride ? f.stride : 1)) || hits(f.d_qual, f.n * (uint64_t)(f.stride ? f.stride : 1)) ||
                    hits(f.d_len, f.n * 4) || hits(f.d_feature_out, f.n * 4) || hits(f.d_n_ids_out, f.n * 4) ||
                    hits(f.d_capture_out, f.n * 4)))
        cr_drop_feature_pending(ctx);
    const KeyHistograms &g = ctx->ghist;
    if (g.valid && hits(g.d_keys, g.n * 8)) ctx->ghist.valid = false;
}
void cr_dense_drop(crgpu_ctx *ctx) {
    if (!ctx->dense.valid) return;
    ctx->dense.valid = false;
    ctx->dense.V = 0;
    if (ctx->layout.set) ctx->layout.bits_bc = ctx->dense.canon_bits;
    ctx->ghist.valid = false;
}
void cr_dense_free(crgpu_ctx *ctx) {
    cr_dense_drop(ctx);
    (void)hipFree(ctx->dense.d_fwd);
    (void)hipFree(ctx->dense.d_back);
    ctx->dense.d_fwd = nullptr;
    ctx->dense.d_back = nullptr;
}

extern "C" int crgpu_invalidate(crgpu_ctx *ctx) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    cr_invalidate(ctx);
    cr_dense_drop(ctx);  // the host may have written a histogram table through crgpu_counts_dev (rebuilt from the tables on demand)
    return CRGPU_OK;
}

static void free_wl(WlTables &w) {
    hipFree(w.d_offA);
    hipFree(w.d_tailA);
    hipFree(w.d_valA);
    hipFree(w.d_offB);
    hipFree(w.d_offE);
    hipFree(w.d_headB);
    hipFree(w.d_valB);
    hipFree(w.d_key_of_rank);
    hipFree(w.d_valid);
    hipFree(w.d_corrected);
    hipFree(w.d_prior_override);
    w = WlTables();
}

void cr_free_wl(WlTables &w) { free_wl(w); }

extern "C" void crgpu_destroy(crgpu_ctx *ctx) {
    if (!ctx) return;
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    cr_comm_destroy(ctx);
    for (auto &w : ctx->wl) free_wl(w);
    for (auto &p : ctx->pat) {
        hipFree(p.d_seq);
        hipFree(p.d_index);
        hipFree(p.d_dist);
    }
    cr_feature_extractors_free(ctx);
    cr_dense_free(ctx);
    hipFree(ctx->d_on_target);
    hipFree(ctx->d_canon_keys);
    hipFree(ctx->d_hot_image);
    hipFree(ctx->d_ptab);
    hipFree(ctx->d_scalars);
    hipFree(ctx->d_sort_hist);
    hipFree(ctx->d_scratch);
    cr_pool_release_all(ctx);
    for (auto &s : ctx->spans) {
        hipEventDestroy(s.start);
        hipEventDestroy(s.stop);
    }
    for (auto ev : ctx->event_pool) hipEventDestroy(ev);
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    if (ctx->ev_aux) hipEventDestroy(ctx->ev_aux);
    if (ctx->stream2) hipStreamDestroy(ctx->stream2);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
}

extern "C" int crgpu_synchronize(crgpu_ctx *ctx) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRGPU_OK;
}

extern "C" void *crgpu_stream(crgpu_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int crgpu_malloc(crgpu_ctx *ctx, void **d_out, uint64_t bytes) {
    if (!ctx || !d_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    return cr_pool_alloc(ctx, d_out, bytes);
}

extern "C" int crgpu_free(crgpu_ctx *ctx, void *d_ptr) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    cr_invalidate(ctx);  // a by-product must not outlive the buffer it describes (the pool hands the address out again)
    cr_pool_free(ctx, d_ptr);
    return CRGPU_OK;
}

extern "C" int crgpu_trim(crgpu_ctx *ctx) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->pool.size();) {
        if (!ctx->pool[i].in_use) {
            (void)hipFree(ctx->pool[i].p);
            ctx->pool.erase(ctx->pool.begin() + i);
        } else {
            i++;
        }
    }
    return CRGPU_OK;
}

extern "C" int crgpu_memcpy_h2d(crgpu_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    if (!bytes) return CRGPU_OK;
    cr_invalidate_range(ctx, d_dst, bytes);  // the destination may be a buffer a kept by-product describes
    CR_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRGPU_OK;
}

extern "C" int crgpu_memcpy_d2h(crgpu_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    if (!bytes) return CRGPU_OK;
    CR_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CRGPU_OK;
}

extern "C" int crgpu_memset(crgpu_ctx *ctx, void *d_dst, int value, uint64_t bytes) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    if (!bytes) return CRGPU_OK;
    cr_invalidate_range(ctx, d_dst, bytes);
    CR_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return CRGPU_OK;
}

int cr_scratch(crgpu_ctx *ctx, uint64_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_scratch) CR_HIP(ctx, hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        uint64_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipMalloc(&ctx->d_scratch, want);
        if (e != hipSuccess)
            return cr_fail(ctx, CRGPU_ENOMEM, "workspace hipMalloc(%llu): %s", (unsigned long long)want, hipGetErrorString(e));
        ctx->scratch_bytes = want;
    }
    *out = ctx->d_scratch;
    return CRGPU_OK;
}

// ---- caching pool ----------------------------------------------------------------------------------

int cr_pool_alloc(crgpu_ctx *ctx, void **out, uint64_t bytes) {
    *out = nullptr;
    if (bytes == 0) bytes = 8;
    bytes = (bytes + 255) & ~255ull;
    // A free block of exactly this size if there is one.  Otherwise a new block, as long as the pool stays below
    // its budget: a step asks for the same sizes in the same order every time, so after ONE step every request
    // finds its own block (with a waste-tolerant best fit a request could take the block a later request needed,
    // and the second and third steps still paid multi-millisecond hipMallocs).  Above the budget: best fit among
    // the free blocks that waste at most 25 %.
    int best = -1;
    uint64_t pooled = 0;
    for (size_t i = 0; i < ctx->pool.size(); i++) {
        const auto &b = ctx->pool[i];
        pooled += b.bytes;
        if (!b.in_use && b.bytes == bytes && best < 0) best = (int)i;
    }
    if (best < 0 && pooled + bytes > ctx->pool_budget) {
        for (size_t i = 0; i < ctx->pool.size(); i++) {
            const auto &b = ctx->pool[i];
            if (b.in_use || b.bytes < bytes || b.bytes - bytes > bytes / 4 + 4096) continue;
            if (best < 0 || b.bytes < ctx->pool[best].bytes) best = (int)i;
        }
    }
    if (best >= 0) {
        ctx->pool[best].in_use = true;
        *out = ctx->pool[best].p;
        return CRGPU_OK;
    }
    void *p = nullptr;
    static const bool pool_debug = getenv("CRGPU_POOL_DEBUG") != nullptr;
    if (pool_debug) fprintf(stderr, "[crgpu pool] hipMalloc %llu bytes (%zu blocks cached)\n", (unsigned long long)bytes, ctx->pool.size());
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // memory pressure: drop every cached free block and retry once
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        for (size_t i = 0; i < ctx->pool.size();) {
            if (!ctx->pool[i].in_use) {
                (void)hipFree(ctx->pool[i].p);
                ctx->pool.erase(ctx->pool.begin() + i);
            } else {
                i++;
            }
        }
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess)
        return cr_fail(ctx, CRGPU_ENOMEM, "device pool hipMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    ctx->pool.push_back({p, bytes, true});
    *out = p;
    return CRGPU_OK;
}

void cr_pool_free(crgpu_ctx *ctx, void *p) {
    if (!p || !ctx) return;
    for (auto &b : ctx->pool)
        if (b.p == p) {
            b.in_use = false;
            return;
        }
    (void)hipFree(p);  // not from the pool
}

void cr_pool_release_all(crgpu_ctx *ctx) {
    for (auto &b : ctx->pool) (void)hipFree(b.p);
    ctx->pool.clear();
}

// ---- timing ------------------------------------------------------------------------------------

hipEvent_t cr_take_event(crgpu_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

CrTimer::CrTimer(crgpu_ctx *c, int s, uint64_t u) : ctx(c), slot(s), units(u) {
    if (!ctx->timing) return;
    start = cr_take_event(ctx);
    stop = cr_take_event(ctx);
    hipEventRecord(start, ctx->stream);
}

CrTimer::~CrTimer() {
    if (!start) return;
    hipEventRecord(stop, ctx->stream);
    ctx->spans.push_back({slot, start, stop, units});
}

static int drain_spans(crgpu_ctx *ctx) {
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &s : ctx->spans) {
        float ms = 0.f;
        CR_HIP(ctx, hipEventElapsedTime(&ms, s.start, s.stop));
        ctx->acc_ms[s.slot] += ms;
        ctx->acc_launches[s.slot] += 1;
        ctx->acc_units[s.slot] += s.units;
        ctx->event_pool.push_back(s.start);
        ctx->event_pool.push_back(s.stop);
    }
    ctx->spans.clear();
    return CRGPU_OK;
}

extern "C" int crgpu_timing_enable(crgpu_ctx *ctx, int on) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_TRY(drain_spans(ctx));
    ctx->timing = on != 0;
    return CRGPU_OK;
}

extern "C" int crgpu_timing_reset(crgpu_ctx *ctx) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_TRY(drain_spans(ctx));
    for (int i = 0; i < CRGPU_T_NSLOTS; i++) {
        ctx->acc_ms[i] = 0;
        ctx->acc_launches[i] = 0;
        ctx->acc_units[i] = 0;
    }
    return CRGPU_OK;
}

extern "C" int crgpu_timing_get(crgpu_ctx *ctx, double *ms_out, uint64_t *launches_out, uint64_t *units_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_TRY(drain_spans(ctx));
    for (int i = 0; i < CRGPU_T_NSLOTS; i++) {
        if (ms_out) ms_out[i] = ctx->acc_ms[i];
        if (launches_out) launches_out[i] = ctx->acc_launches[i];
        if (units_out) units_out[i] = ctx->acc_units[i];
    }
    return CRGPU_OK;
}
