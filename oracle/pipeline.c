/*
 * pipeline.c -- oracle restatement of the stage glue around the per-read / per-barcode
 * functions: make_shard pass A, barcode_correction pass B, align_and_count's per-barcode
 * dedup/count, BarcodeIndex and CSC assembly, MTX text.
 * TEST INFRASTRUCTURE ONLY (see cr_oracle.h).
 *
 * Follows:
 *   cr_types/src/rna_read.rs:352-366          segment -> Whitelist::check_and_update
 *   cr_lib/src/make_shard_metrics.rs:171-188  valid_bc_counts / valid_bc_segment_counts
 *   cr_lib/src/stages/barcode_correction.rs:76-99,328-345   correct invalid segments, count
 *   cr_lib/src/stages/barcode_correction.rs:246-263         chunk shape (parallel oracle)
 *   cr_lib/src/aligner.rs:283-334             one DupBuilder per library type inside a barcode
 *   cr_lib/src/stages/align_and_count.rs:312-333   umi_counts.sort(); feature_counts()
 *   cr_types/src/types.rs:152-160,180-188     UmiCount Ord; BcUmiInfo::feature_counts
 *   cr_types/src/barcode_index.rs:20-53       sorted + dedup'd union of barcodes
 *   cr_h5/src/count_matrix.rs:382-448         CSC (data, indices, indptr over ALL barcodes)
 *   cr_lib/src/stages/write_matrix_market.rs:80-122   MTX text
 */
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "cr_oracle.h"

#define MAX_LIB 16

/* wall-clock spans of the last oracle_barcode_stage / oracle_count_stage call (bench.py's cpu_baseline reports the
 * stages separately): 0 pass A, 1 histogram join, 2 pass B, 3 corrected-histogram join, 4 barcode index + bringing the
 * valid reads into barcode order (the reference gets that order from shardio's sorted shards, outside its hot path),
 * 5 per-barcode dedup + counting, 6 matrix assembly */
static double g_timing[8];
#include <time.h>
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void oracle_get_timing(double *out8) { memcpy(out8, g_timing, sizeof(g_timing)); }

static int clamp_threads(int n_threads) {
    if (n_threads < 1) n_threads = 1;
#ifndef _OPENMP
    n_threads = 1;
#endif
    return n_threads;
}

int oracle_barcode_stage(const oracle_reads *reads, const oracle_whitelist *const *wl,
                         oracle_hist **valid_hist, oracle_hist **corrected_hist,
                         const oracle_hist *const *prior_override, double max_expected_errors,
                         double threshold, int n_threads, oracle_bc_result *out) {
    const uint64_t n = reads->n;
    const uint32_t L = reads->cb_len;
    n_threads = clamp_threads(n_threads);

    /* pass A: exact whitelist check (make_shard) */
    typedef oracle_hist *hist_row[MAX_LIB];
    hist_row *tl_valid = (hist_row *)calloc((size_t)n_threads, sizeof(hist_row));
    hist_row *tl_corr = (hist_row *)calloc((size_t)n_threads, sizeof(hist_row));
    double t0 = now_s();
#pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        uint64_t lo = n * (uint64_t)t / (uint64_t)n_threads, hi = n * (uint64_t)(t + 1) / (uint64_t)n_threads;
        for (uint64_t i = lo; i < hi; i++) {
            int lib = reads->lib ? reads->lib[i] : 0;
            const oracle_whitelist *w = wl[lib];
            char *dst = out->corrected_cb + i * L;
            if (w && oracle_whitelist_check_and_update(w, reads->cb + i * L, L, dst)) {
                out->bc_state[i] = 1; /* ValidBeforeCorrection */
                if (!tl_valid[t][lib]) tl_valid[t][lib] = oracle_hist_new();
                oracle_hist_observe_by(tl_valid[t][lib], dst, L, 1);
            } else {
                out->bc_state[i] = 0;
                memcpy(dst, reads->cb + i * L, L);
            }
        }
    }
    g_timing[0] = now_s() - t0;
    t0 = now_s();
    /* make_shard join: Metric::merge of the per-chunk histograms (make_shard.rs:343-358) */
    for (int t = 0; t < n_threads; t++)
        for (int lib = 0; lib < MAX_LIB; lib++)
            if (tl_valid[t][lib]) {
                uint64_t m = oracle_hist_size(tl_valid[t][lib]);
                char *seqs = (char *)malloc(m * L + 1);
                int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (m + 1));
                oracle_hist_dump_sorted(tl_valid[t][lib], L, seqs, cnt);
                for (uint64_t k = 0; k < m; k++) oracle_hist_observe_by(valid_hist[lib], seqs + k * L, L, cnt[k]);
                free(seqs);
                free(cnt);
                oracle_hist_free(tl_valid[t][lib]);
            }

    g_timing[1] = now_s() - t0;
    t0 = now_s();
    /* pass B: posterior correction of the invalid reads with the GLOBAL prior */
#pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        uint64_t lo = n * (uint64_t)t / (uint64_t)n_threads, hi = n * (uint64_t)(t + 1) / (uint64_t)n_threads;
        for (uint64_t i = lo; i < hi; i++) {
            if (out->bc_state[i] != 0) continue;
            int lib = reads->lib ? reads->lib[i] : 0;
            const oracle_whitelist *w = wl[lib];
            if (!w) continue;
            const oracle_hist *prior = (prior_override && prior_override[lib]) ? prior_override[lib] : valid_hist[lib];
            char fixed[ORACLE_MAX_SEQ];
            const uint8_t *q = reads->cb_qual ? reads->cb_qual + i * L : NULL;
            if (oracle_posterior_correct(w, prior, reads->cb + i * L, q, L, max_expected_errors, threshold, fixed)) {
                memcpy(out->corrected_cb + i * L, fixed, L);
                out->bc_state[i] = 2; /* ValidAfterCorrection */
                if (!tl_corr[t][lib]) tl_corr[t][lib] = oracle_hist_new();
                oracle_hist_observe_by(tl_corr[t][lib], fixed, L, 1);
            }
        }
    }
    g_timing[2] = now_s() - t0;
    t0 = now_s();
    for (int t = 0; t < n_threads; t++)
        for (int lib = 0; lib < MAX_LIB; lib++)
            if (tl_corr[t][lib]) {
                uint64_t m = oracle_hist_size(tl_corr[t][lib]);
                char *seqs = (char *)malloc(m * L + 1);
                int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (m + 1));
                oracle_hist_dump_sorted(tl_corr[t][lib], L, seqs, cnt);
                for (uint64_t k = 0; k < m; k++)
                    oracle_hist_observe_by(corrected_hist[lib], seqs + k * L, L, cnt[k]);
                free(seqs);
                free(cnt);
                oracle_hist_free(tl_corr[t][lib]);
            }
    g_timing[3] = now_s() - t0;
    free(tl_valid);
    free(tl_corr);
    return 0;
}

/* ---------------------------------------------------------------------------------------- */

typedef struct {
    uint64_t key; /* order-preserving 2-bit code of the barcode (A<C<G<T == byte order) */
    uint64_t idx;
} bcref;

static int cmp_bcref(const void *a, const void *b) {
    const bcref *x = (const bcref *)a, *y = (const bcref *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return 0;
}

static uint64_t encode_bc(const char *s, uint32_t len) {
    uint64_t r = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint64_t c = s[i] == 'A' ? 0 : s[i] == 'C' ? 1 : s[i] == 'G' ? 2 : 3;
        r = (r << 2) | c;
    }
    return r;
}

typedef struct {
    uint8_t lib;
    oracle_umicount uc;
} molrec;

static int cmp_molrec(const void *a, const void *b) {
    /* UmiCount derive(Ord): library_idx, feature_idx, umi, read_count, utype, probe_idx */
    const molrec *x = (const molrec *)a, *y = (const molrec *)b;
    if (x->lib != y->lib) return x->lib < y->lib ? -1 : 1;
    if (x->uc.feature_idx != y->uc.feature_idx) return x->uc.feature_idx < y->uc.feature_idx ? -1 : 1;
    if (x->uc.umi != y->uc.umi) return x->uc.umi < y->uc.umi ? -1 : 1;
    if (x->uc.read_count != y->uc.read_count) return x->uc.read_count < y->uc.read_count ? -1 : 1;
    if (x->uc.utype != y->uc.utype) return x->uc.utype < y->uc.utype ? -1 : 1;
    return 0;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

typedef struct {
    uint64_t n_mol;
    molrec *mol;
    uint64_t n_fc;
    uint32_t *fc_feature;
    uint32_t *fc_count;
} group_result;

/* one barcode: aligner.rs:283-334 + align_and_count.rs:279-336 */
static void process_barcode(const oracle_reads *reads, const bcref *refs, uint64_t m,
                            uint32_t multiplexing_lib_mask, oracle_dupinfo *dup_out,
                            group_result *gr) {
    const uint32_t UL = reads->umi_len;
    memset(gr, 0, sizeof(*gr));
    gr->mol = (molrec *)malloc(sizeof(molrec) * (m ? m : 1));
    char *umi = (char *)malloc((size_t)m * UL + 1);
    uint8_t *valid = (uint8_t *)malloc(m + 1);
    uint32_t *feat = (uint32_t *)malloc(sizeof(uint32_t) * (m + 1));
    uint8_t *ut = (uint8_t *)malloc(m + 1);
    uint64_t *qn = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1));
    uint64_t *src = (uint64_t *)malloc(sizeof(uint64_t) * (m + 1));
    oracle_dupinfo *di = (oracle_dupinfo *)malloc(sizeof(oracle_dupinfo) * (m + 1));
    oracle_umicount *uc = (oracle_umicount *)malloc(sizeof(oracle_umicount) * (m + 1));

    uint32_t libs_present = 0;
    for (uint64_t j = 0; j < m; j++) libs_present |= 1u << (reads->lib ? reads->lib[refs[j].idx] : 0);
    for (int lib = 0; lib < MAX_LIB; lib++) {
        if (!((libs_present >> lib) & 1u)) continue;
        uint64_t k = 0;
        for (uint64_t j = 0; j < m; j++) {
            uint64_t i = refs[j].idx;
            int l = reads->lib ? reads->lib[i] : 0;
            if (l != lib) continue;
            memcpy(umi + k * UL, reads->umi + i * UL, UL);
            uint32_t ul = 0; /* a UMI shorter than umi_len is padded with NUL bytes (variable-length UMIs) */
            while (ul < UL && reads->umi[i * UL + ul] != 0) ul++;
            valid[k] = (uint8_t)(ul > 0 && oracle_umi_is_valid(reads->umi + i * UL, reads->umi_qual + i * UL, ul));
            feat[k] = reads->feature[i];
            ut[k] = reads->utype ? reads->utype[i] : 0;
            qn[k] = i; /* read headers are unique; their rank == read index */
            src[k] = i;
            k++;
        }
        if (!k) continue;
        int corr_enabled = !((multiplexing_lib_mask >> lib) & 1u);
        uint64_t nu = oracle_mark_dups_group(umi, UL, valid, feat, ut, qn, k, corr_enabled, 1, di, uc);
        for (uint64_t j = 0; j < nu; j++) {
            gr->mol[gr->n_mol].lib = (uint8_t)lib;
            gr->mol[gr->n_mol].uc = uc[j];
            gr->n_mol++;
        }
        if (dup_out)
            for (uint64_t j = 0; j < k; j++) dup_out[src[j]] = di[j];
    }
    /* align_and_count.rs:314 */
    qsort(gr->mol, gr->n_mol, sizeof(molrec), cmp_molrec);
    /* types.rs:180-188 : histogram of feature_idx over all UmiCounts of the barcode */
    uint32_t *f = (uint32_t *)malloc(sizeof(uint32_t) * (gr->n_mol + 1));
    for (uint64_t j = 0; j < gr->n_mol; j++) f[j] = gr->mol[j].uc.feature_idx;
    qsort(f, gr->n_mol, sizeof(uint32_t), cmp_u32);
    gr->fc_feature = (uint32_t *)malloc(sizeof(uint32_t) * (gr->n_mol + 1));
    gr->fc_count = (uint32_t *)malloc(sizeof(uint32_t) * (gr->n_mol + 1));
    for (uint64_t j = 0; j < gr->n_mol;) {
        uint64_t e = j;
        while (e < gr->n_mol && f[e] == f[j]) e++;
        gr->fc_feature[gr->n_fc] = f[j];
        gr->fc_count[gr->n_fc] = (uint32_t)(e - j);
        gr->n_fc++;
        j = e;
    }
    free(f);
    free(umi);
    free(valid);
    free(feat);
    free(ut);
    free(qn);
    free(src);
    free(di);
    free(uc);
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

oracle_matrix *oracle_count_stage(const oracle_reads *reads, const oracle_bc_result *bc,
                                  oracle_hist *const *valid_hist, oracle_hist *const *corrected_hist,
                                  int n_lib, uint32_t multiplexing_lib_mask, int n_threads,
                                  oracle_dupinfo *dup_out) {
    const uint64_t n = reads->n;
    const uint32_t L = reads->cb_len;
    n_threads = clamp_threads(n_threads);
    if (dup_out) memset(dup_out, 0, sizeof(oracle_dupinfo) * n);
    double t0 = now_s();

    /* BarcodeIndex: sorted, dedup'd union of the barcodes in corrected_barcode_counts
     * (= raw valid counts merged with corrected counts, barcode_correction.rs:401-407) */
    uint64_t cap = 0;
    for (int l = 0; l < n_lib; l++) {
        if (valid_hist[l]) cap += oracle_hist_size(valid_hist[l]);
        if (corrected_hist[l]) cap += oracle_hist_size(corrected_hist[l]);
    }
    uint64_t *cols = (uint64_t *)malloc(sizeof(uint64_t) * (cap + 1));
    uint64_t nc = 0;
    for (int l = 0; l < n_lib; l++)
        for (int w = 0; w < 2; w++) {
            const oracle_hist *h = w ? corrected_hist[l] : valid_hist[l];
            if (!h) continue;
            uint64_t m = oracle_hist_size(h);
            char *seqs = (char *)malloc(m * L + 1);
            int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (m + 1));
            oracle_hist_dump_sorted(h, L, seqs, cnt);
            for (uint64_t k = 0; k < m; k++) cols[nc++] = encode_bc(seqs + k * L, L);
            free(seqs);
            free(cnt);
        }
    qsort(cols, nc, sizeof(uint64_t), cmp_u64);
    uint64_t V = 0;
    for (uint64_t k = 0; k < nc; k++)
        if (k == 0 || cols[k] != cols[k - 1]) cols[V++] = cols[k];

    /* shardio barcode order: only valid barcodes reach counting (align_and_count.rs:312).  The reference reads its
     * records already sorted by barcode; here the order is made by a bucket partition on the leading bases (per-thread
     * counts, one scatter) and a qsort per bucket on the worker threads -- the total order (barcode, read index) makes
     * the result independent of the thread count. */
    bcref *refs = (bcref *)malloc(sizeof(bcref) * (n + 1));
    uint64_t nv = 0;
    {
        const uint32_t key_bits = 2 * L, bbits = key_bits < 12 ? key_bits : 12, nbk = 1u << bbits;
        const int T = n_threads;
        uint64_t *cnt = (uint64_t *)calloc((size_t)T * nbk + 1, sizeof(uint64_t));
        uint64_t *bstart = (uint64_t *)calloc((size_t)nbk + 1, sizeof(uint64_t));
#pragma omp parallel num_threads(T)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            const uint64_t lo = n * (uint64_t)t / (uint64_t)T, hi = n * (uint64_t)(t + 1) / (uint64_t)T;
            uint64_t *c = cnt + (size_t)t * nbk;
            for (uint64_t i = lo; i < hi; i++)
                if (bc->bc_state[i]) c[encode_bc(bc->corrected_cb + i * L, L) >> (key_bits - bbits)]++;
        }
        for (uint32_t b = 0; b < nbk; b++) { /* bucket-major, thread-minor exclusive prefix */
            bstart[b] = nv;
            for (int t = 0; t < T; t++) {
                const uint64_t c = cnt[(size_t)t * nbk + b];
                cnt[(size_t)t * nbk + b] = nv;
                nv += c;
            }
        }
        bstart[nbk] = nv;
#pragma omp parallel num_threads(T)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            const uint64_t lo = n * (uint64_t)t / (uint64_t)T, hi = n * (uint64_t)(t + 1) / (uint64_t)T;
            uint64_t *c = cnt + (size_t)t * nbk;
            for (uint64_t i = lo; i < hi; i++)
                if (bc->bc_state[i]) {
                    const uint64_t key = encode_bc(bc->corrected_cb + i * L, L);
                    const uint64_t o = c[key >> (key_bits - bbits)]++;
                    refs[o].key = key;
                    refs[o].idx = i;
                }
        }
#pragma omp parallel for schedule(dynamic, 8) num_threads(T)
        for (int64_t b = 0; b < (int64_t)nbk; b++)
            qsort(refs + bstart[b], bstart[b + 1] - bstart[b], sizeof(bcref), cmp_bcref);
        free(cnt);
        free(bstart);
    }

    /* group boundaries */
    uint64_t *gstart = (uint64_t *)malloc(sizeof(uint64_t) * (nv + 2));
    uint64_t ng = 0;
    for (uint64_t j = 0; j < nv; j++)
        if (j == 0 || refs[j].key != refs[j - 1].key) gstart[ng++] = j;
    gstart[ng] = nv;

    group_result *res = (group_result *)calloc(ng + 1, sizeof(group_result));
    g_timing[4] = now_s() - t0;
    t0 = now_s();
    /* par_proc.rs:106-164: groups fanned out to worker threads */
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
    for (int64_t g = 0; g < (int64_t)ng; g++)
        process_barcode(reads, refs + gstart[g], gstart[g + 1] - gstart[g], multiplexing_lib_mask, dup_out, &res[g]);

    g_timing[5] = now_s() - t0;
    t0 = now_s();
    oracle_matrix *M = (oracle_matrix *)calloc(1, sizeof(oracle_matrix));
    M->n_barcodes = V;
    M->cb_len = L;
    M->barcodes = (char *)malloc(V * L + 1);
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t c = 0; c < V; c++)
        for (uint32_t p = 0; p < L; p++) M->barcodes[c * L + p] = acgt[(cols[c] >> (2 * (L - 1 - p))) & 3];
    uint64_t nnz = 0, nmol = 0;
    for (uint64_t g = 0; g < ng; g++) {
        nnz += res[g].n_fc;
        nmol += res[g].n_mol;
    }
    M->nnz = nnz;
    M->indices = (int32_t *)malloc(sizeof(int32_t) * (nnz + 1));
    M->data = (int32_t *)malloc(sizeof(int32_t) * (nnz + 1));
    M->indptr = (int64_t *)calloc(V + 1, sizeof(int64_t));
    M->n_umi_counts = nmol;
    M->mol_bc_col = (uint32_t *)malloc(sizeof(uint32_t) * (nmol + 1));
    M->mol_lib = (uint8_t *)malloc(nmol + 1);
    M->mol = (oracle_umicount *)malloc(sizeof(oracle_umicount) * (nmol + 1));

    /* count_matrix.rs:407-435: groups arrive in (barcode, feature) order; barcode_counts[col] = n */
    int64_t *percol = (int64_t *)calloc(V + 1, sizeof(int64_t));
    uint64_t o = 0, om = 0;
    for (uint64_t g = 0; g < ng; g++) {
        uint64_t key = refs[gstart[g]].key;
        /* barcode_index.get_index(): every counted barcode is in the index by construction */
        uint64_t lo = 0, hi = V;
        while (lo < hi) {
            uint64_t mid = (lo + hi) / 2;
            if (cols[mid] < key) lo = mid + 1; else hi = mid;
        }
        uint64_t col = lo;
        if (col < V && cols[col] == key) percol[col] = (int64_t)res[g].n_fc;
        for (uint64_t j = 0; j < res[g].n_fc; j++) {
            M->indices[o] = (int32_t)res[g].fc_feature[j];
            M->data[o] = (int32_t)res[g].fc_count[j];
            o++;
        }
        for (uint64_t j = 0; j < res[g].n_mol; j++) {
            M->mol_bc_col[om] = (uint32_t)col;
            M->mol_lib[om] = res[g].mol[j].lib;
            M->mol[om] = res[g].mol[j].uc;
            om++;
        }
        free(res[g].mol);
        free(res[g].fc_feature);
        free(res[g].fc_count);
    }
    int64_t total = 0;
    M->indptr[0] = 0;
    for (uint64_t c = 0; c < V; c++) {
        total += percol[c];
        M->indptr[c + 1] = total;
    }
    free(percol);
    free(res);
    free(gstart);
    free(refs);
    free(cols);
    g_timing[6] = now_s() - t0;
    return M;
}

void oracle_matrix_free(oracle_matrix *m) {
    if (!m) return;
    free(m->barcodes);
    free(m->indptr);
    free(m->indices);
    free(m->data);
    free(m->mol_bc_col);
    free(m->mol_lib);
    free(m->mol);
    free(m);
}

int64_t oracle_write_mtx(const oracle_matrix *m, uint32_t n_features, const char *metadata_line,
                         const char *path) {
    /* write_matrix_market.rs:96-118 (uncompressed text; the reference gzips the same bytes) */
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int64_t w = 0;
    w += fprintf(f, "%%%%MatrixMarket matrix coordinate integer general\n");
    w += fprintf(f, "%s\n", metadata_line);
    w += fprintf(f, "%u %llu %llu\n", n_features, (unsigned long long)m->n_barcodes, (unsigned long long)m->nnz);
    for (uint64_t c = 0; c < m->n_barcodes; c++)
        for (int64_t k = m->indptr[c]; k < m->indptr[c + 1]; k++)
            w += fprintf(f, "%d %llu %d\n", 1 + m->indices[k], (unsigned long long)(1 + c), m->data[k]);
    fclose(f);
    return w;
}
