// matrix.hip -- barcode index, CSC assembly and Matrix Market text (host side of the boundary).
//
// Replaces BarcodeIndex::new / from_iter (cr_types/src/barcode_index.rs:20-53),
// write_matrix_h5_helper's CSC construction (cr_h5/src/count_matrix.rs:382-448) and
// write_matrix_mtx (cr_lib/src/stages/write_matrix_market.rs:80-122).  The .h5 container itself
// stays with the unchanged Rust host (no HDF5 library in this image): it receives exactly the
// arrays it writes today (data i32, indices -> i64 on disk, indptr i64, barcodes, shape).
#include <algorithm>
#include <cstdio>

#include "common.h"

// a barcode of cb_len <= 32 bases as text: lo = the first min(16, cb_len) bases packed, hi = the rest
static void format_barcode(char *buf, uint32_t lo, uint32_t hi, uint32_t cb_len) {
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    const uint32_t n_lo = cb_len < 16 ? cb_len : 16, n_hi = cb_len - n_lo;
    for (uint32_t p = 0; p < n_lo; p++) buf[p] = acgt[(lo >> (2 * (n_lo - 1 - p))) & 3u];
    for (uint32_t p = 0; p < n_hi; p++) buf[n_lo + p] = acgt[(hi >> (2 * (n_hi - 1 - p))) & 3u];
    buf[cb_len] = 0;
}

struct MatrixImpl {
    crgpu_matrix view;
    std::vector<uint32_t> rank, seq, seq_hi;  // seq_hi: bases 17.. of a segmented construct's barcodes, else empty
    std::vector<int64_t> indptr;
    std::vector<int32_t> indices, data;
    std::vector<uint16_t> gem_group;
};

// BarcodeIndex (cr_types/src/barcode_index.rs:20-53) as a map rank -> column, 0xFFFFFFFF for unseen barcodes
static int column_of_rank(crgpu_ctx *ctx, std::vector<uint32_t> &col_of_rank) {
    const uint32_t W = ctx->n_canon;
    std::vector<uint8_t> seen(W, 0);
    std::vector<uint32_t> tmp(W);
    for (int l = 0; l < CRGPU_MAX_LIB; l++) {
        if (!ctx->wl[l].set) continue;
        for (int which = 0; which < 2; which++) {
            CR_TRY(crgpu_memcpy_d2h(ctx, tmp.data(), which ? ctx->wl[l].d_corrected : ctx->wl[l].d_valid, sizeof(uint32_t) * W));
            for (uint32_t r = 0; r < W; r++) seen[r] |= tmp[r] != 0;
        }
    }
    col_of_rank.assign(W, 0xFFFFFFFFu);
    uint32_t c = 0;
    for (uint32_t r = 0; r < W; r++)
        if (seen[r]) col_of_rank[r] = c++;
    return CRGPU_OK;
}

extern "C" int crgpu_counts_molecule_info(crgpu_ctx *ctx, const crgpu_counts *c, uint16_t gem_group, uint16_t *gem_group_out,
                                          uint64_t *barcode_idx_out, uint32_t *feature_idx_out, uint16_t *library_idx_out,
                                          uint32_t *umi_out, uint32_t *count_out, uint32_t *umi_type_out) {
    if (!ctx || !c) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_counts_molecule_info: no whitelist set");
    uint64_t nt = 0, nm = 0;
    CR_TRY(crgpu_counts_info(ctx, c, &nt, &nm));
    if (!nm) return CRGPU_OK;
    std::vector<uint32_t> bc(nm);
    std::vector<uint8_t> lib(nm), ut(nm);
    CR_TRY(crgpu_counts_molecules(ctx, c, bc.data(), lib.data(), feature_idx_out, umi_out, count_out, ut.data()));
    std::vector<uint32_t> col;
    CR_TRY(column_of_rank(ctx, col));
    for (uint64_t i = 0; i < nm; i++) {
        CR_REQUIRE(ctx, col[bc[i]] != 0xFFFFFFFFu, CRGPU_ESTATE,
                   "crgpu_counts_molecule_info: a molecule's barcode has no read in the context's histograms");
        if (gem_group_out) gem_group_out[i] = gem_group;
        if (barcode_idx_out) barcode_idx_out[i] = col[bc[i]];
        if (library_idx_out) library_idx_out[i] = lib[i];
        if (umi_type_out) umi_type_out[i] = ut[i];
    }
    return CRGPU_OK;
}

extern "C" int crgpu_assemble_matrix(crgpu_ctx *ctx, const uint32_t *bc, const uint32_t *feature, const uint32_t *count,
                                     uint64_t n_triplets, uint32_t n_features, crgpu_matrix **out) {
    if (!ctx || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_assemble_matrix: no whitelist set");
    CR_REQUIRE(ctx, n_triplets == 0 || (bc && feature && count), CRGPU_EINVAL, "crgpu_assemble_matrix: NULL triplets");
    const uint32_t W = ctx->n_canon;

    // BarcodeIndex: sorted, dedup'd union over libraries of the barcodes in corrected_barcode_counts
    // (= valid counts merged with corrected counts, barcode_correction.rs:401-407)
    std::vector<uint8_t> seen(W, 0);
    std::vector<uint32_t> tmp(W);
    for (int l = 0; l < CRGPU_MAX_LIB; l++) {
        if (!ctx->wl[l].set) continue;
        for (int which = 0; which < 2; which++) {
            CR_TRY(crgpu_memcpy_d2h(ctx, tmp.data(), which ? ctx->wl[l].d_corrected : ctx->wl[l].d_valid, sizeof(uint32_t) * W));
            for (uint32_t r = 0; r < W; r++) seen[r] |= tmp[r] != 0;
        }
    }
    MatrixImpl *m = new (std::nothrow) MatrixImpl();
    if (!m) return cr_fail(ctx, CRGPU_ENOMEM, "out of host memory");
    std::vector<uint32_t> col_of_rank(W, 0xFFFFFFFFu);
    for (uint32_t r = 0; r < W; r++)
        if (seen[r]) {
            col_of_rank[r] = (uint32_t)m->rank.size();
            m->rank.push_back(r);
            uint32_t lo, hi;
            cr_rank_to_seq(ctx, r, &lo, &hi);
            m->seq.push_back(lo);
            if (ctx->cb_len > 16) m->seq_hi.push_back(hi);
        }
    const uint64_t V = m->rank.size();

    // FeatureBarcodeCount stream in BarcodeThenFeatureOrder (types.rs:121-137)
    std::vector<uint64_t> order(n_triplets);
    for (uint64_t i = 0; i < n_triplets; i++) order[i] = i;
    bool sorted = true;
    for (uint64_t i = 1; i < n_triplets && sorted; i++)
        sorted = bc[i - 1] < bc[i] || (bc[i - 1] == bc[i] && feature[i - 1] <= feature[i]);
    if (!sorted)
        std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
            return bc[a] != bc[b] ? bc[a] < bc[b] : feature[a] < feature[b];
        });

    std::vector<int64_t> percol(V, 0);
    m->indices.reserve(n_triplets);
    m->data.reserve(n_triplets);
    for (uint64_t i = 0; i < n_triplets;) {
        const uint64_t a = order[i];
        if (bc[a] >= W || col_of_rank[bc[a]] == 0xFFFFFFFFu || feature[a] >= n_features) {
            delete m;
            return cr_fail(ctx, CRGPU_EINVAL,
                           "triplet %llu (barcode rank %u, feature %u) is outside the barcode index / feature space",
                           (unsigned long long)a, bc[a], feature[a]);
        }
        // count_matrix.rs:409-414: entries with equal (barcode, feature) are summed
        uint64_t j = i;
        uint64_t sum = 0;
        while (j < n_triplets && bc[order[j]] == bc[a] && feature[order[j]] == feature[a]) sum += count[order[j++]];
        m->indices.push_back((int32_t)feature[a]);
        m->data.push_back((int32_t)sum);
        percol[col_of_rank[bc[a]]] += 1;
        i = j;
    }
    m->indptr.resize(V + 1);
    m->indptr[0] = 0;
    for (uint64_t c = 0; c < V; c++) m->indptr[c + 1] = m->indptr[c] + percol[c];

    m->view.n_barcodes = V;
    m->view.nnz = m->data.size();
    m->view.n_features = n_features;
    m->view.cb_len = ctx->cb_len;
    m->view.barcode_rank = m->rank.data();
    m->view.barcode_seq = m->seq.data();
    m->view.barcode_seq_hi = m->seq_hi.empty() ? nullptr : m->seq_hi.data();
    m->view.indptr = m->indptr.data();
    m->view.indices = m->indices.data();
    m->view.data = m->data.data();
    m->view.gem_group = nullptr;
    *out = &m->view;
    return CRGPU_OK;
}

extern "C" int crgpu_concat_matrices(crgpu_ctx *ctx, const crgpu_matrix *const *mats, const uint16_t *gem_groups,
                                     uint32_t n_mats, crgpu_matrix **out) {
    if (!ctx || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    CR_REQUIRE(ctx, n_mats >= 1 && mats && gem_groups, CRGPU_EINVAL, "crgpu_concat_matrices: nothing to merge");
    for (uint32_t i = 0; i < n_mats; i++) {
        CR_REQUIRE(ctx, mats[i] != nullptr, CRGPU_EINVAL, "crgpu_concat_matrices: NULL matrix %u", i);
        CR_REQUIRE(ctx, mats[i]->gem_group == nullptr, CRGPU_EINVAL, "crgpu_concat_matrices: matrix %u is already a merge", i);
        CR_REQUIRE(ctx, mats[i]->n_features == mats[0]->n_features && mats[i]->cb_len == mats[0]->cb_len, CRGPU_EINVAL,
                   "crgpu_concat_matrices: matrix %u has another feature space or barcode length", i);
        CR_REQUIRE(ctx, i == 0 || gem_groups[i] > gem_groups[i - 1], CRGPU_EINVAL,
                   "crgpu_concat_matrices: gem groups must be strictly ascending");
    }
    MatrixImpl *m = new (std::nothrow) MatrixImpl();
    if (!m) return cr_fail(ctx, CRGPU_ENOMEM, "out of host memory");
    m->indptr.push_back(0);
    for (uint32_t i = 0; i < n_mats; i++) {
        const crgpu_matrix &a = *mats[i];
        const int64_t shift = (int64_t)m->data.size();
        m->rank.insert(m->rank.end(), a.barcode_rank, a.barcode_rank + a.n_barcodes);
        m->seq.insert(m->seq.end(), a.barcode_seq, a.barcode_seq + a.n_barcodes);
        if (a.barcode_seq_hi) m->seq_hi.insert(m->seq_hi.end(), a.barcode_seq_hi, a.barcode_seq_hi + a.n_barcodes);
        m->gem_group.insert(m->gem_group.end(), a.n_barcodes, gem_groups[i]);
        for (uint64_t c = 0; c < a.n_barcodes; c++) m->indptr.push_back(shift + a.indptr[c + 1]);
        m->indices.insert(m->indices.end(), a.indices, a.indices + a.nnz);
        m->data.insert(m->data.end(), a.data, a.data + a.nnz);
    }
    m->view.n_barcodes = m->rank.size();
    m->view.nnz = m->data.size();
    m->view.n_features = mats[0]->n_features;
    m->view.cb_len = mats[0]->cb_len;
    m->view.barcode_rank = m->rank.data();
    m->view.barcode_seq = m->seq.data();
    m->view.barcode_seq_hi = m->seq_hi.empty() ? nullptr : m->seq_hi.data();
    m->view.indptr = m->indptr.data();
    m->view.indices = m->indices.data();
    m->view.data = m->data.data();
    m->view.gem_group = m->gem_group.data();
    *out = &m->view;
    return CRGPU_OK;
}

static void finish_view(MatrixImpl *m, uint32_t n_features, uint32_t cb_len) {
    m->view.n_barcodes = m->rank.size();
    m->view.nnz = m->data.size();
    m->view.n_features = n_features;
    m->view.cb_len = cb_len;
    m->view.barcode_rank = m->rank.data();
    m->view.barcode_seq = m->seq.data();
    m->view.barcode_seq_hi = m->seq_hi.empty() ? nullptr : m->seq_hi.data();
    m->view.indptr = m->indptr.data();
    m->view.indices = m->indices.data();
    m->view.data = m->data.data();
    m->view.gem_group = m->gem_group.empty() ? nullptr : m->gem_group.data();
}

// CountMatrix.merge (lib/python/cellranger/matrix.py:479-482, merge_matrices :1319-1329): `self.m += other.m` on two
// matrices of the same shape (same features, same barcodes in the same order) -- the element-wise sum, canonical CSC.
extern "C" int crgpu_sum_matrices(crgpu_ctx *ctx, const crgpu_matrix *a, const crgpu_matrix *b, crgpu_matrix **out) {
    if (!ctx || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    CR_REQUIRE(ctx, a && b, CRGPU_EINVAL, "crgpu_sum_matrices: NULL matrix");
    CR_REQUIRE(ctx, a->n_features == b->n_features && a->n_barcodes == b->n_barcodes && a->cb_len == b->cb_len, CRGPU_EINVAL,
               "crgpu_sum_matrices: shapes differ (%u x %llu vs %u x %llu)", a->n_features, (unsigned long long)a->n_barcodes,
               b->n_features, (unsigned long long)b->n_barcodes);
    for (uint64_t c = 0; c < a->n_barcodes; c++)
        CR_REQUIRE(ctx, a->barcode_seq[c] == b->barcode_seq[c] &&
                            (a->barcode_seq_hi ? a->barcode_seq_hi[c] : 0) == (b->barcode_seq_hi ? b->barcode_seq_hi[c] : 0) &&
                            (a->gem_group ? a->gem_group[c] : 0) == (b->gem_group ? b->gem_group[c] : 0),
                   CRGPU_EINVAL, "crgpu_sum_matrices: column %llu holds different barcodes", (unsigned long long)c);
    MatrixImpl *m = new (std::nothrow) MatrixImpl();
    if (!m) return cr_fail(ctx, CRGPU_ENOMEM, "out of host memory");
    m->rank.assign(a->barcode_rank, a->barcode_rank + a->n_barcodes);
    m->seq.assign(a->barcode_seq, a->barcode_seq + a->n_barcodes);
    if (a->barcode_seq_hi) m->seq_hi.assign(a->barcode_seq_hi, a->barcode_seq_hi + a->n_barcodes);
    if (a->gem_group) m->gem_group.assign(a->gem_group, a->gem_group + a->n_barcodes);
    m->indptr.push_back(0);
    for (uint64_t c = 0; c < a->n_barcodes; c++) {
        int64_t i = a->indptr[c], j = b->indptr[c];
        const int64_t ie = a->indptr[c + 1], je = b->indptr[c + 1];
        while (i < ie || j < je) {  // merge of two index-sorted columns
            int32_t row, v;
            if (j >= je || (i < ie && a->indices[i] < b->indices[j])) {
                row = a->indices[i];
                v = a->data[i++];
            } else if (i >= ie || b->indices[j] < a->indices[i]) {
                row = b->indices[j];
                v = b->data[j++];
            } else {
                row = a->indices[i];
                v = a->data[i++] + b->data[j++];
            }
            m->indices.push_back(row);
            m->data.push_back(v);
        }
        m->indptr.push_back((int64_t)m->data.size());
    }
    finish_view(m, a->n_features, a->cb_len);
    *out = &m->view;
    return CRGPU_OK;
}

// CountMatrix.select_barcodes (matrix.py:860-875): the columns `cols` in the given order.
extern "C" int crgpu_select_barcodes(crgpu_ctx *ctx, const crgpu_matrix *a, const uint64_t *cols, uint64_t n_cols,
                                     crgpu_matrix **out) {
    if (!ctx || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    CR_REQUIRE(ctx, a && (cols || n_cols == 0), CRGPU_EINVAL, "crgpu_select_barcodes: NULL argument");
    for (uint64_t k = 0; k < n_cols; k++)
        CR_REQUIRE(ctx, cols[k] < a->n_barcodes, CRGPU_EINVAL, "crgpu_select_barcodes: column %llu out of range",
                   (unsigned long long)cols[k]);
    MatrixImpl *m = new (std::nothrow) MatrixImpl();
    if (!m) return cr_fail(ctx, CRGPU_ENOMEM, "out of host memory");
    m->indptr.push_back(0);
    for (uint64_t k = 0; k < n_cols; k++) {
        const uint64_t c = cols[k];
        m->rank.push_back(a->barcode_rank[c]);
        m->seq.push_back(a->barcode_seq[c]);
        if (a->barcode_seq_hi) m->seq_hi.push_back(a->barcode_seq_hi[c]);
        if (a->gem_group) m->gem_group.push_back(a->gem_group[c]);
        m->indices.insert(m->indices.end(), a->indices + a->indptr[c], a->indices + a->indptr[c + 1]);
        m->data.insert(m->data.end(), a->data + a->indptr[c], a->data + a->indptr[c + 1]);
        m->indptr.push_back((int64_t)m->data.size());
    }
    finish_view(m, a->n_features, a->cb_len);
    *out = &m->view;
    return CRGPU_OK;
}

extern "C" void crgpu_matrix_free(crgpu_ctx *, crgpu_matrix *mv) {
    if (!mv) return;
    delete reinterpret_cast<MatrixImpl *>(mv);  // view is the first member
}

extern "C" int crgpu_write_mtx(crgpu_ctx *ctx, const crgpu_matrix *m, const char *metadata_line, const char *mtx_path,
                               const char *barcodes_tsv_path, uint16_t gem_group) {
    if (!ctx || !m) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    if (mtx_path) {
        FILE *f = fopen(mtx_path, "wb");
        if (!f) return cr_fail(ctx, CRGPU_EINVAL, "cannot open %s", mtx_path);
        // write_matrix_market.rs:96-118 (the reference gzips exactly this text)
        fprintf(f, "%%%%MatrixMarket matrix coordinate integer general\n");
        fprintf(f, "%s\n", metadata_line ? metadata_line : "%metadata_json: {}");
        fprintf(f, "%u %llu %llu\n", m->n_features, (unsigned long long)m->n_barcodes, (unsigned long long)m->nnz);
        for (uint64_t c = 0; c < m->n_barcodes; c++)
            for (int64_t k = m->indptr[c]; k < m->indptr[c + 1]; k++)
                fprintf(f, "%d %llu %d\n", 1 + m->indices[k], (unsigned long long)(1 + c), m->data[k]);
        fclose(f);
    }
    if (barcodes_tsv_path) {
        FILE *f = fopen(barcodes_tsv_path, "wb");
        if (!f) return cr_fail(ctx, CRGPU_EINVAL, "cannot open %s", barcodes_tsv_path);
        char buf[40];
        for (uint64_t c = 0; c < m->n_barcodes; c++) {
            format_barcode(buf, m->barcode_seq[c], m->barcode_seq_hi ? m->barcode_seq_hi[c] : 0u, m->cb_len);
            fprintf(f, "%s-%u\n", buf, (unsigned)(m->gem_group ? m->gem_group[c] : gem_group));  // Barcode Display: "{content}-{gem_group}" (barcode/src/lib.rs:197-201)
        }
        fclose(f);
    }
    return CRGPU_OK;
}

extern "C" int crgpu_count(crgpu_ctx *ctx, const crgpu_records *recs, uint32_t n_features, crgpu_matrix **out) {
    if (!ctx || !recs || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    CR_REQUIRE(ctx, ctx->layout.set && ctx->layout.n_features == n_features, CRGPU_ESTATE,
               "crgpu_count: call crgpu_set_key_layout with the same n_features first");
    uint64_t *d_keys = nullptr;
    CR_TRY(cr_pool_alloc(ctx, (void **)&d_keys, recs->n * sizeof(uint64_t)));
    uint64_t n_keys = 0;
    crgpu_counts *c = nullptr;
    int rc = crgpu_build_keys_dev(ctx, recs, d_keys, &n_keys);
    if (rc == CRGPU_OK) rc = crgpu_count_keys_dev(ctx, d_keys, n_keys, &c);
    cr_pool_free(ctx, d_keys);
    if (rc != CRGPU_OK) return rc;
    uint64_t nt = 0;
    crgpu_counts_info(ctx, c, &nt, nullptr);
    std::vector<uint32_t> bc(nt), ft(nt), ct(nt);
    rc = crgpu_counts_triplets(ctx, c, bc.data(), ft.data(), ct.data());
    crgpu_counts_free(ctx, c);
    if (rc != CRGPU_OK) return rc;
    CrTimer t(ctx, CRGPU_T_MATRIX);
    return crgpu_assemble_matrix(ctx, bc.data(), ft.data(), ct.data(), nt, n_features, out);
}

// The count entry of SURVEY 8(b) for host-resident records: upload -> crgpu_count_records_dev (per-read DupInfo wanted) or
// keys + crgpu_count_keys_dev -> triplets -> matrix; DupInfo comes back as an array of structs (mark_dups.rs:61-72).
extern "C" int crgpu_count_host(crgpu_ctx *ctx, const crgpu_records *recs_host, uint32_t n_features, crgpu_matrix **out,
                                crgpu_dupinfo *per_read, crgpu_counts **counts_out) {
    if (!ctx || !recs_host || !out) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    *out = nullptr;
    if (counts_out) *counts_out = nullptr;
    CR_REQUIRE(ctx, ctx->layout.set && ctx->layout.n_features == n_features, CRGPU_ESTATE,
               "crgpu_count_host: call crgpu_set_key_layout with the same n_features first");
    const crgpu_records &h = *recs_host;
    const uint64_t n = h.n;
    CR_REQUIRE(ctx, n <= 0x7FFFFFFFull, CRGPU_ERANGE, "crgpu_count_host: at most 2^31-1 records per call");
    CR_REQUIRE(ctx, n == 0 || (h.d_bc_idx && h.d_umi && h.d_umi_qualn && h.d_feature), CRGPU_EINVAL, "crgpu_count_host: NULL buffer");
    struct Up {  // pooled device copy of one host array
        crgpu_ctx *c;
        void *p = nullptr;
        ~Up() { cr_pool_free(c, p); }
        int put(const void *src, uint64_t bytes) {
            if (!src || !bytes) return CRGPU_OK;
            CR_TRY(cr_pool_alloc(c, &p, bytes));
            return crgpu_memcpy_h2d(c, p, src, bytes);
        }
        int room(uint64_t bytes) { return bytes ? cr_pool_alloc(c, &p, bytes) : CRGPU_OK; }
    };
    Up bc{ctx}, umi{ctx}, uq{ctx}, ft{ctx}, fl{ctx}, ul{ctx}, pb{ctx}, o_umi{ctx}, o_cnt{ctx}, o_fl{ctx};
    CR_TRY(bc.put(h.d_bc_idx, n * 4));
    CR_TRY(umi.put(h.d_umi, n * 4));
    CR_TRY(uq.put(h.d_umi_qualn, n * (uint64_t)h.umi_len));
    CR_TRY(ft.put(h.d_feature, n * 4));
    CR_TRY(fl.put(h.d_flags, n));
    CR_TRY(ul.put(h.d_umi_len, n));
    CR_TRY(pb.put(h.d_probe_idx, n * 4));
    crgpu_records d{n, h.umi_len, (const uint32_t *)bc.p, (const uint32_t *)umi.p, (const uint8_t *)uq.p, (const uint32_t *)ft.p,
                    (const uint8_t *)fl.p, (const uint8_t *)ul.p, (const int32_t *)pb.p};
    crgpu_counts *c = nullptr;
    const bool want_reads = per_read != nullptr || h.d_probe_idx != nullptr;
    if (n == 0) {
        CR_TRY(crgpu_count_keys_dev(ctx, nullptr, 0, &c));
    } else if (want_reads) {
        if (per_read) {
            CR_TRY(o_umi.room(n * 4));
            CR_TRY(o_cnt.room(n * 4));
            CR_TRY(o_fl.room(n));
        }
        CR_TRY(crgpu_count_records_dev(ctx, &d, &c, (uint32_t *)o_umi.p, (uint32_t *)o_cnt.p, (uint8_t *)o_fl.p));
    } else {
        Up keys{ctx};
        CR_TRY(keys.room((n ? n : 1) * sizeof(uint64_t)));
        uint64_t n_keys = 0;
        CR_TRY(crgpu_build_keys_dev(ctx, &d, (uint64_t *)keys.p, &n_keys));
        CR_TRY(crgpu_count_keys_dev(ctx, (uint64_t *)keys.p, n_keys, &c));
    }
    struct CountsGuard {
        crgpu_ctx *ctx;
        crgpu_counts *c;
        ~CountsGuard() {
            if (c) crgpu_counts_free(ctx, c);
        }
    } guard{ctx, c};
    uint64_t nt = 0;
    CR_TRY(crgpu_counts_info(ctx, c, &nt, nullptr));
    std::vector<uint32_t> tb(nt), tf(nt), tc(nt);
    CR_TRY(crgpu_counts_triplets(ctx, c, tb.data(), tf.data(), tc.data()));
    if (per_read && n) {
        std::vector<uint32_t> pu(n), rc(n);
        std::vector<uint8_t> df(n);
        CR_TRY(crgpu_memcpy_d2h(ctx, pu.data(), o_umi.p, n * 4));
        CR_TRY(crgpu_memcpy_d2h(ctx, rc.data(), o_cnt.p, n * 4));
        CR_TRY(crgpu_memcpy_d2h(ctx, df.data(), o_fl.p, n));
        for (uint64_t i = 0; i < n; i++) per_read[i] = crgpu_dupinfo{pu[i], rc[i], df[i], {0, 0, 0}};
    }
    {
        CrTimer t(ctx, CRGPU_T_MATRIX);
        CR_TRY(crgpu_assemble_matrix(ctx, tb.data(), tf.data(), tc.data(), nt, n_features, out));
    }
    if (counts_out) {
        *counts_out = c;
        guard.c = nullptr;
    }
    return CRGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// barcode_summary.csv (ALIGN_AND_COUNT join, cr_lib/src/stages/align_and_count.rs:806-817)
// ------------------------------------------------------------------------------------------------
extern "C" int crgpu_write_barcode_summary_csv(crgpu_ctx *ctx, const crgpu_barcode_summary_row *rows, uint64_t n_rows,
                                               uint16_t gem_group, const uint32_t *library_type_order,
                                               const char *const *library_type_name, uint32_t n_libs, const char *path) {
    if (!ctx || !path || (n_rows && !rows)) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, ctx->canon_set, CRGPU_ESTATE, "crgpu_write_barcode_summary_csv: no whitelist set");
    CR_REQUIRE(ctx, library_type_order && library_type_name && n_libs >= 1 && n_libs <= CRGPU_MAX_LIB, CRGPU_EINVAL,
               "crgpu_write_barcode_summary_csv: library types missing");
    // one BarcodeSummary per (library type, barcode): libraries of one type are visited by the same
    // AlignAndCountVisitor (align_metrics.rs:704-719).  Same gem group everywhere, so the String order of
    // "SEQ-gg" is the order of the sequences = the rank order.
    struct Row {
        uint32_t order, rank, lib;
        uint64_t v[4];
    };
    std::vector<Row> merged;
    merged.reserve(n_rows);
    for (uint64_t i = 0; i < n_rows; i++) {
        const crgpu_barcode_summary_row &r = rows[i];
        CR_REQUIRE(ctx, r.library < n_libs && r.barcode_rank < ctx->n_canon, CRGPU_EINVAL,
                   "crgpu_write_barcode_summary_csv: row %llu out of range", (unsigned long long)i);
        merged.push_back(Row{library_type_order[r.library], r.barcode_rank, r.library,
                             {r.reads, r.umis, r.candidate_dup_reads, r.umi_corrected_reads}});
    }
    std::stable_sort(merged.begin(), merged.end(),
                     [](const Row &a, const Row &b) { return a.order != b.order ? a.order < b.order : a.rank < b.rank; });
    FILE *f = fopen(path, "wb");
    if (!f) return cr_fail(ctx, CRGPU_EINVAL, "cannot open %s", path);
    fprintf(f, "library_type,barcode,reads,umis,candidate_dup_reads,umi_corrected_reads\n");
    char buf[40];
    for (size_t i = 0; i < merged.size();) {
        Row acc = merged[i];
        size_t j = i + 1;
        for (; j < merged.size() && merged[j].order == acc.order && merged[j].rank == acc.rank; j++)
            for (int k = 0; k < 4; k++) acc.v[k] += merged[j].v[k];
        uint32_t lo, hi;
        cr_rank_to_seq(ctx, acc.rank, &lo, &hi);
        format_barcode(buf, lo, hi, ctx->cb_len);
        fprintf(f, "%s,%s-%u,%llu,%llu,%llu,%llu\n", library_type_name[acc.lib], buf, (unsigned)gem_group,
                (unsigned long long)acc.v[0], (unsigned long long)acc.v[1], (unsigned long long)acc.v[2],
                (unsigned long long)acc.v[3]);
        i = j;
    }
    fclose(f);
    return CRGPU_OK;
}
