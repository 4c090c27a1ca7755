// whitelist.hip -- host-side construction of the per-library whitelist tables and the per-library
// histograms.  Replaces Whitelist::construct (barcode/src/whitelist.rs:313-330,468-472).
#include <algorithm>
#include <numeric>

#include "common.h"
#include "wl_view.h"

void cr_free_wl(WlTables &w);

static int pack_ascii(crgpu_ctx *ctx, const char *s, uint32_t n, uint32_t len, std::vector<uint32_t> &out,
                      const char *what) {
    out.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t k = 0;
        for (uint32_t j = 0; j < len; j++) {
            uint32_t c;
            switch (s[(size_t)i * len + j]) {
                case 'A': c = 0; break;
                case 'C': c = 1; break;
                case 'G': c = 2; break;
                case 'T': c = 3; break;
                default:
                    return cr_fail(ctx, CRGPU_EINVAL, "%s entry %u has a non-ACGT character at position %u", what, i, j);
            }
            k = (k << 2) | c;
        }
        out[i] = k;
    }
    return CRGPU_OK;
}

template <typename T>
static int upload(crgpu_ctx *ctx, T **d, const std::vector<T> &h) {
    *d = nullptr;
    // +32 bytes of padding: the lookups read whole dwords / 16-byte groups starting inside the table
    hipError_t e = hipMalloc((void **)d, std::max<size_t>(h.size(), 1) * sizeof(T) + 32);
    if (e != hipSuccess) return cr_fail(ctx, CRGPU_ENOMEM, "hipMalloc whitelist table: %s", hipGetErrorString(e));
    if (!h.empty()) CR_HIP(ctx, hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return CRGPU_OK;
}

extern "C" int crgpu_set_whitelist_packed(crgpu_ctx *ctx, int lib, const uint32_t *keys, uint32_t n, uint32_t len,
                                          const uint32_t *canon, uint32_t n_canon, const uint32_t *translate_to) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    cr_dense_drop(ctx);
    CR_REQUIRE(ctx, lib >= 0 && lib < CRGPU_MAX_LIB, CRGPU_EINVAL, "library id %d out of range", lib);
    CR_REQUIRE(ctx, keys && canon && n > 0 && n_canon > 0, CRGPU_EINVAL, "empty whitelist");
    CR_REQUIRE(ctx, len >= 1 && len <= 16, CRGPU_ERANGE,
               "barcode length %u unsupported: this engine packs barcodes of <= 16 bases in 32 bits", len);
    const uint64_t space = len == 16 ? (1ull << 32) : (1ull << (2 * len));
    for (uint32_t i = 0; i < n; i++)
        CR_REQUIRE(ctx, keys[i] < space, CRGPU_EINVAL, "whitelist key %u does not fit %u bases", i, len);
    CR_HIP(ctx, hipSetDevice(ctx->device));
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cr_drop_miss_records(ctx);

    // canonical space: ascending packed order == byte-lexicographic order (barcode/src/lib.rs:119-124)
    std::vector<uint32_t> order(n_canon);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return canon[a] < canon[b]; });
    std::vector<uint32_t> sorted(n_canon);
    for (uint32_t r = 0; r < n_canon; r++) sorted[r] = canon[order[r]];
    for (uint32_t r = 1; r < n_canon; r++)
        CR_REQUIRE(ctx, sorted[r] != sorted[r - 1], CRGPU_EINVAL, "canonical barcode list has duplicates");
    if (!ctx->canon_set) {
        ctx->canon_sorted = sorted;
        ctx->canon_order = order;
        ctx->n_canon = n_canon;
        ctx->cb_len = len;
        ctx->canon_set = true;
        CR_TRY(upload(ctx, &ctx->d_canon_keys, sorted));
    } else {
        CR_REQUIRE(ctx, ctx->n_canon == n_canon && ctx->cb_len == len && ctx->canon_sorted == sorted, CRGPU_EINVAL,
                   "all libraries of a context must share one canonical barcode list");
    }
    std::vector<uint32_t> rank_of_pos(n_canon);
    for (uint32_t r = 0; r < n_canon; r++) rank_of_pos[order[r]] = r;

    // (raw key, canonical rank) pairs; HashMap::collect keeps the LAST duplicate (whitelist.rs:301-311)
    struct KV {
        uint32_t key, val, seq;
    };
    std::vector<KV> kv(n);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t val;
        if (translate_to) {
            CR_REQUIRE(ctx, translate_to[i] < n_canon, CRGPU_EINVAL, "translate_to[%u] out of range", i);
            val = rank_of_pos[translate_to[i]];
        } else {
            auto it = std::lower_bound(sorted.begin(), sorted.end(), keys[i]);
            CR_REQUIRE(ctx, it != sorted.end() && *it == keys[i], CRGPU_EINVAL,
                       "plain whitelist key %u is not in the canonical list", i);
            val = (uint32_t)(it - sorted.begin());
        }
        kv[i] = {keys[i], val, i};
    }
    std::sort(kv.begin(), kv.end(), [](const KV &a, const KV &b) { return a.key != b.key ? a.key < b.key : a.seq < b.seq; });
    std::vector<KV> uniq;
    uniq.reserve(n);
    for (uint32_t i = 0; i < n; i++) {
        if (!uniq.empty() && uniq.back().key == kv[i].key)
            uniq.back() = kv[i];
        else
            uniq.push_back(kv[i]);
    }
    const uint32_t m = (uint32_t)uniq.size();

    const uint32_t hA = len / 2, hB = len - hA;
    const uint32_t bitsA = 2 * hA, bitsB = 2 * hB;
    const uint32_t maskB = (uint32_t)((1ull << bitsB) - 1);
    std::vector<uint32_t> offA((1u << bitsA) + 1, 0), offB((1u << bitsB) + 1, 0), valA(m);
    std::vector<uint16_t> tailA(m), headB(m);
    std::vector<uint32_t> valB(m);
    bool identity = true;
    for (uint32_t p = 0; p < m; p++) {
        const uint32_t head = (uint32_t)((uint64_t)uniq[p].key >> bitsB), tail = uniq[p].key & maskB;
        offA[head + 1]++;
        offB[tail + 1]++;
        tailA[p] = (uint16_t)tail;
        valA[p] = uniq[p].val;
        if (uniq[p].val != p) identity = false;
    }
    for (size_t i = 1; i < offA.size(); i++) offA[i] += offA[i - 1];
    for (size_t i = 1; i < offB.size(); i++) offB[i] += offB[i - 1];
    {
        // table B: stable counting sort by tail keeps heads ascending inside a bin
        std::vector<uint32_t> cur(offB.begin(), offB.end() - 1);
        for (uint32_t p = 0; p < m; p++) {
            const uint32_t head = (uint32_t)((uint64_t)uniq[p].key >> bitsB), tail = uniq[p].key & maskB;
            valB[cur[tail]] = uniq[p].val;
            headB[cur[tail]++] = (uint16_t)head;
        }
    }

    // exact-lookup index over the same sorted tails: ~1-2 keys per bin, at least as fine as the head
    const uint32_t key_bits = 2 * len;
    uint32_t bitsE = cr_ceil_log2(m) > 0 ? cr_ceil_log2(m) - 1 : 0;
    if (bitsE < bitsA) bitsE = bitsA;
    if (bitsE > key_bits) bitsE = key_bits;
    if (bitsE > 26) bitsE = 26;
    const uint32_t shiftE = key_bits - bitsE;
    std::vector<uint32_t> offE(((size_t)1 << bitsE) + 1, 0);
    for (uint32_t p = 0; p < m; p++) offE[(size_t)((uint64_t)uniq[p].key >> shiftE) + 1]++;
    for (size_t i = 1; i < offE.size(); i++) offE[i] += offE[i - 1];

    WlTables &w = ctx->wl[lib];
    cr_free_wl(w);
    w.n = m;
    w.shiftE = shiftE;
    CR_TRY(upload(ctx, &w.d_offE, offE));
    w.bitsA = bitsA;
    w.bitsB = bitsB;
    CR_TRY(upload(ctx, &w.d_offA, offA));
    CR_TRY(upload(ctx, &w.d_tailA, tailA));
    if (!identity) {
        CR_TRY(upload(ctx, &w.d_valA, valA));
        // rank -> a key of THIS list with that rank (a translated list: the partner's key; a subset: some ranks have none): what
        // pass A's table of frequent barcodes is built from (on a plain full list the canonical keys serve)
        std::vector<uint32_t> key_of_rank(n_canon, 0xFFFFFFFFu);
        for (uint32_t p = 0; p < m; p++) key_of_rank[uniq[p].val] = uniq[p].key;
        CR_TRY(upload(ctx, &w.d_key_of_rank, key_of_rank));
    }
    CR_TRY(upload(ctx, &w.d_offB, offB));
    CR_TRY(upload(ctx, &w.d_headB, headB));
    CR_TRY(upload(ctx, &w.d_valB, valB));
    std::vector<uint32_t> zeros(n_canon, 0);
    CR_TRY(upload(ctx, &w.d_valid, zeros));
    CR_TRY(upload(ctx, &w.d_corrected, zeros));
    w.set = true;
    return CRGPU_OK;
}

extern "C" int crgpu_set_whitelist(crgpu_ctx *ctx, int lib, const char *keys, uint32_t n, uint32_t len,
                                   const char *canon, uint32_t n_canon, const uint32_t *translate_to) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, keys && canon, CRGPU_EINVAL, "NULL whitelist");
    CR_REQUIRE(ctx, len >= 1 && len <= 16, CRGPU_ERANGE,
               "barcode length %u unsupported: this engine packs barcodes of <= 16 bases in 32 bits", len);
    std::vector<uint32_t> pk, pc;
    CR_TRY(pack_ascii(ctx, keys, n, len, pk, "whitelist"));
    CR_TRY(pack_ascii(ctx, canon, n_canon, len, pc, "canonical list"));
    return crgpu_set_whitelist_packed(ctx, lib, pk.data(), n, len, pc.data(), n_canon, translate_to);
}

void cr_rank_to_seq(const crgpu_ctx *ctx, uint32_t rank, uint32_t *lo, uint32_t *hi) {
    if (ctx->n_segments == 0) {
        *lo = ctx->canon_sorted[rank];
        *hi = 0;
        return;
    }
    // mixed radix, first segment most significant -> the concatenated sequence as up to 64 bits
    uint32_t r[CRGPU_MAX_SEGMENTS];
    for (int s = (int)ctx->n_segments - 1; s >= 0; s--) {
        r[s] = rank % ctx->seg_n[s];
        rank /= ctx->seg_n[s];
    }
    uint64_t full = 0;
    for (uint32_t s = 0; s < ctx->n_segments; s++) full = (full << (2 * ctx->seg_len[s])) | ctx->seg_seq[s][r[s]];
    if (ctx->cb_len <= 16) {
        *lo = (uint32_t)full;
        *hi = 0;
    } else {
        const uint32_t n_hi = ctx->cb_len - 16;
        *lo = (uint32_t)(full >> (2 * n_hi));
        *hi = (uint32_t)(full & ((1ull << (2 * n_hi)) - 1ull));
    }
}

extern "C" int crgpu_set_barcode_segments(crgpu_ctx *ctx, int lib, uint32_t n_segments, const uint32_t *seg_n,
                                          const uint32_t *seg_len, const uint32_t *const *seg_seqs) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    cr_dense_drop(ctx);
    CR_REQUIRE(ctx, lib >= 0 && lib < CRGPU_MAX_LIB, CRGPU_EINVAL, "library id %d out of range", lib);
    CR_REQUIRE(ctx, n_segments >= 1 && n_segments <= CRGPU_MAX_SEGMENTS && seg_n && seg_len && seg_seqs, CRGPU_EINVAL,
               "crgpu_set_barcode_segments: 1..%d segments", CRGPU_MAX_SEGMENTS);
    uint64_t space = 1;
    uint32_t total_len = 0;
    for (uint32_t s = 0; s < n_segments; s++) {
        CR_REQUIRE(ctx, seg_n[s] > 0 && seg_seqs[s] && seg_len[s] >= 1 && seg_len[s] <= 16, CRGPU_EINVAL,
                   "segment %u: empty whitelist or more than 16 bases", s);
        const uint64_t lim = seg_len[s] == 16 ? (1ull << 32) : (1ull << (2 * seg_len[s]));
        for (uint32_t i = 0; i < seg_n[s]; i++) {
            CR_REQUIRE(ctx, seg_seqs[s][i] < lim, CRGPU_EINVAL, "segment %u: sequence %u does not fit %u bases", s, i, seg_len[s]);
            CR_REQUIRE(ctx, i == 0 || seg_seqs[s][i] > seg_seqs[s][i - 1], CRGPU_EINVAL,
                       "segment %u: the sequences must be ascending and distinct (a segment context's canonical order)", s);
        }
        space *= seg_n[s];
        total_len += seg_len[s];
        CR_REQUIRE(ctx, space < (1ull << 31), CRGPU_ERANGE, "the product of the segment whitelists exceeds 2^31 barcodes");
    }
    CR_REQUIRE(ctx, total_len <= 32, CRGPU_ERANGE, "segmented barcodes of %u bases unsupported (<= 32)", total_len);
    if (ctx->canon_set) {
        bool same = ctx->n_segments == n_segments && ctx->n_canon == (uint32_t)space;
        for (uint32_t s = 0; same && s < n_segments; s++)
            same = ctx->seg_n[s] == seg_n[s] && ctx->seg_len[s] == seg_len[s] &&
                   std::equal(ctx->seg_seq[s].begin(), ctx->seg_seq[s].end(), seg_seqs[s]);
        CR_REQUIRE(ctx, same, CRGPU_EINVAL, "all libraries of a context must share one canonical barcode space");
    } else {
        ctx->n_segments = n_segments;
        for (uint32_t s = 0; s < n_segments; s++) {
            ctx->seg_n[s] = seg_n[s];
            ctx->seg_len[s] = seg_len[s];
            ctx->seg_seq[s].assign(seg_seqs[s], seg_seqs[s] + seg_n[s]);
        }
        ctx->n_canon = (uint32_t)space;
        ctx->cb_len = total_len;
        ctx->canon_set = true;
    }
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cr_drop_miss_records(ctx);
    WlTables &w = ctx->wl[lib];
    cr_free_wl(w);
    std::vector<uint32_t> zeros(ctx->n_canon, 0);
    CR_TRY(upload(ctx, &w.d_valid, zeros));
    CR_TRY(upload(ctx, &w.d_corrected, zeros));
    w.n = 0;  // no lookup tables: the barcode stage runs on the segment contexts
    w.set = true;
    return CRGPU_OK;
}

extern "C" int crgpu_whitelist_info(crgpu_ctx *ctx, uint32_t *n_canon_out, uint32_t *len_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "no whitelist set");
    if (n_canon_out) *n_canon_out = ctx->n_canon;
    if (len_out) *len_out = ctx->cb_len;
    return CRGPU_OK;
}

extern "C" int crgpu_get_canon_order(crgpu_ctx *ctx, uint32_t *order_out, uint32_t *seqs_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "no whitelist set");
    CR_REQUIRE(ctx, ctx->n_segments == 0, CRGPU_ESTATE, "a segmented barcode space has no sequence list: ranks are mixed radix");
    if (order_out) memcpy(order_out, ctx->canon_order.data(), sizeof(uint32_t) * ctx->n_canon);
    if (seqs_out) memcpy(seqs_out, ctx->canon_sorted.data(), sizeof(uint32_t) * ctx->n_canon);
    return CRGPU_OK;
}

// ---- histograms --------------------------------------------------------------------------------

static int table_ptr(crgpu_ctx *ctx, int lib, int which, bool for_write, uint32_t **out) {
    CR_REQUIRE(ctx, lib >= 0 && lib < CRGPU_MAX_LIB && ctx->wl[lib].set, CRGPU_ESTATE, "library %d has no whitelist", lib);
    WlTables &w = ctx->wl[lib];
    switch (which) {
        case CRGPU_COUNTS_VALID: *out = w.d_valid; break;
        case CRGPU_COUNTS_CORRECTED: *out = w.d_corrected; break;
        case CRGPU_COUNTS_PRIOR:
            if (for_write && !w.d_prior_override) {
                CR_HIP(ctx, hipMalloc((void **)&w.d_prior_override, sizeof(uint32_t) * ctx->n_canon));
                CR_HIP(ctx, hipMemsetAsync(w.d_prior_override, 0, sizeof(uint32_t) * ctx->n_canon, ctx->stream));
            }
            *out = w.d_prior_override ? w.d_prior_override : w.d_valid;
            break;
        default: return cr_fail(ctx, CRGPU_EINVAL, "unknown counts table %d", which);
    }
    return CRGPU_OK;
}

extern "C" int crgpu_get_counts(crgpu_ctx *ctx, int lib, int which, uint32_t *counts_out) {
    if (!ctx || !counts_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    uint32_t *d;
    CR_TRY(table_ptr(ctx, lib, which, false, &d));
    return crgpu_memcpy_d2h(ctx, counts_out, d, sizeof(uint32_t) * ctx->n_canon);
}

extern "C" int crgpu_set_counts(crgpu_ctx *ctx, int lib, int which, const uint32_t *counts) {
    if (!ctx || !counts) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    uint32_t *d;
    CR_TRY(table_ptr(ctx, lib, which, true, &d));
    cr_dense_drop(ctx);
    return crgpu_memcpy_h2d(ctx, d, counts, sizeof(uint32_t) * ctx->n_canon);
}

extern "C" int crgpu_counts_dev(crgpu_ctx *ctx, int lib, int which, uint32_t **d_out) {
    if (!ctx || !d_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    return table_ptr(ctx, lib, which, which == CRGPU_COUNTS_PRIOR ? false : false, d_out);
}

extern "C" int crgpu_reset_counts(crgpu_ctx *ctx) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    cr_invalidate(ctx);
    cr_dense_drop(ctx);
    for (auto &w : ctx->wl)
        if (w.set) {
            CR_HIP(ctx, hipMemsetAsync(w.d_valid, 0, sizeof(uint32_t) * ctx->n_canon, ctx->stream));
            CR_HIP(ctx, hipMemsetAsync(w.d_corrected, 0, sizeof(uint32_t) * ctx->n_canon, ctx->stream));
            if (w.d_prior_override) {
                CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
                CR_HIP(ctx, hipFree(w.d_prior_override));
                w.d_prior_override = nullptr;
            }
        }
    return CRGPU_OK;
}

// ---- BARCODE_CORRECTION join outputs derived from the tables (barcode_correction.rs:372-448) ------------------------------
// Bookkeeping over 2 x n_libs tables of n_canon u32 (a few MB): done on the host after one download per table.
extern "C" int crgpu_barcode_correction_metrics(crgpu_ctx *ctx, int lib, crgpu_bc_correction_metrics *out) {
    if (!ctx || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, lib >= 0 && lib < CRGPU_MAX_LIB && ctx->wl[lib].set, CRGPU_ESTATE, "library %d has no whitelist", lib);
    const uint32_t W = ctx->n_canon;
    std::vector<uint32_t> v(W), c(W);
    CR_TRY(crgpu_memcpy_d2h(ctx, v.data(), ctx->wl[lib].d_valid, sizeof(uint32_t) * W));
    CR_TRY(crgpu_memcpy_d2h(ctx, c.data(), ctx->wl[lib].d_corrected, sizeof(uint32_t) * W));
    memset(out, 0, sizeof(*out));
    // effective_diversity (metric/src/histogram.rs:161-171) sums f64 over a HashMap in arbitrary order; here the sums are
    // exact integers converted once (the reference's own result varies in the last bits with its iteration order)
    unsigned __int128 s2 = 0;
    uint64_t s = 0;
    for (uint32_t r = 0; r < W; r++) {
        out->valid_reads += v[r];
        out->corrected_reads += c[r];
        const uint64_t t = (uint64_t)v[r] + c[r];  // bc_counts_corrected = raw valid merged with corrected (:401-404)
        if (t) {
            out->barcodes_detected++;
            s += t;
            s2 += (unsigned __int128)t * t;
        }
    }
    out->effective_barcode_diversity = s2 ? ((double)s * (double)s) / (double)s2 : 0.0 / 0.0;  // 0/0 = NaN as in the reference
    return CRGPU_OK;
}

extern "C" int crgpu_total_barcode_counts(crgpu_ctx *ctx, int64_t min_reads_to_report_bc, uint32_t *rank_out, uint64_t *count_out,
                                          uint64_t cap, uint64_t *n_out) {
    if (!ctx || !n_out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_total_barcode_counts: no whitelist set");
    const uint32_t W = ctx->n_canon;
    std::vector<uint64_t> corr(W, 0), total(W, 0);
    std::vector<uint32_t> tmp(W);
    for (int l = 0; l < CRGPU_MAX_LIB; l++) {
        if (!ctx->wl[l].set) continue;
        // join (:380-390): every library's raw valid counts, each filtered by the threshold on its own
        CR_TRY(crgpu_memcpy_d2h(ctx, tmp.data(), ctx->wl[l].d_valid, sizeof(uint32_t) * W));
        for (uint32_t r = 0; r < W; r++)
            if ((int64_t)tmp[r] >= min_reads_to_report_bc) total[r] += tmp[r];
        // chunk (:345,360): the barcodes of the corrected reads, all libraries in one histogram
        CR_TRY(crgpu_memcpy_d2h(ctx, tmp.data(), ctx->wl[l].d_corrected, sizeof(uint32_t) * W));
        for (uint32_t r = 0; r < W; r++) corr[r] += tmp[r];
    }
    uint64_t n = 0;
    for (uint32_t r = 0; r < W; r++) {
        if (corr[r] && (int64_t)corr[r] >= min_reads_to_report_bc) total[r] += corr[r];
        if (!total[r]) continue;
        if (rank_out && count_out && n < cap) {
            rank_out[n] = r;
            count_out[n] = total[r];
        }
        n++;
    }
    *n_out = n;
    CR_REQUIRE(ctx, !(rank_out && count_out) || n <= cap, CRGPU_ERANGE, "crgpu_total_barcode_counts: %llu barcodes, room for %llu",
               (unsigned long long)n, (unsigned long long)cap);
    return CRGPU_OK;
}

// fill the device views used by the kernels
int cr_make_views(crgpu_ctx *ctx, WlView *views) {
    for (int l = 0; l < CRGPU_MAX_LIB; l++) {
        const WlTables &w = ctx->wl[l];
        WlView v;
        memset(&v, 0, sizeof(v));
        if (w.set) {
            v.offA = w.d_offA;
            v.tailA = w.d_tailA;
            v.valA = w.d_valA;
            v.offB = w.d_offB;
            v.offE = w.d_offE;
            v.shiftE = w.shiftE;
            v.headB = w.d_headB;
            v.valB = w.d_valB;
            v.valid = w.d_valid;
            v.corrected = w.d_corrected;
            v.prior = w.d_prior_override ? w.d_prior_override : w.d_valid;
            v.bitsA = w.bitsA;
            v.bitsB = w.bitsB;
            v.n = w.n;
        }
        views[l] = v;
    }
    return CRGPU_OK;
}
