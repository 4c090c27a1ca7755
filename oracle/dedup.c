/*
 * dedup.c -- oracle restatement of UMI validity, UMI correction, low-support filtering and
 * duplicate marking for one (barcode, library type) group.
 * TEST INFRASTRUCTURE ONLY (see cr_oracle.h).
 *
 * Follows:
 *   umi/src/info.rs:6,20-37,60-75      UMI_MIN_QV, UmiInfo::new (has_n, is_homopolymer, low_min_qual)
 *   tx_annotation/src/mark_dups.rs:19-59    correct_umis
 *   tx_annotation/src/mark_dups.rs:87-108   determine_low_support_umigenes
 *   tx_annotation/src/mark_dups.rs:128-155  DupBuilder::observe
 *   tx_annotation/src/mark_dups.rs:202-277  BarcodeDupMarker::new
 *   tx_annotation/src/mark_dups.rs:280-363  BarcodeDupMarker::process
 *   umi/src/lib.rs UmiType ordering: Txomic < NonTxomic
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bytemap.h"
#include "cr_oracle.h"

#define UMI_MIN_QV 10
#define ILLUMINA_QUAL_OFFSET 33

int oracle_umi_is_valid(const char *seq, const uint8_t *qual, uint32_t len) {
    int has_n = 0;
    for (uint32_t i = 0; i < len; i++)
        if (seq[i] == 'N') has_n = 1;
    int is_homopolymer = 1; /* info.rs:60-67 (an empty / 1-base UMI counts as homopolymer) */
    for (uint32_t i = 1; i < len; i++)
        if (seq[i - 1] != seq[i]) {
            is_homopolymer = 0;
            break;
        }
    int low_min_qual = 0; /* info.rs:69-75: u8 subtraction, wraps in a release build */
    for (uint32_t i = 0; i < len; i++) {
        uint8_t d = (uint8_t)(qual[i] - ILLUMINA_QUAL_OFFSET);
        if (d < UMI_MIN_QV) {
            low_min_qual = 1;
            break;
        }
    }
    return !(has_n || is_homopolymer || low_min_qual);
}

uint32_t oracle_encode_2bit_u32(const char *seq, uint32_t len) {
    /* fastq_set SSeq::encode_2bit_u32: first base most significant, A0 C1 G2 T3 */
    uint32_t r = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint32_t c;
        switch (seq[i]) {
            case 'A': c = 0; break;
            case 'C': c = 1; break;
            case 'G': c = 2; break;
            case 'T': c = 3; break;
            default: c = 0; break; /* never reached for valid UMIs */
        }
        r = (r << 2) | c;
    }
    return r;
}

/* one (UmiSeq, Gene) entry of the reference's hash maps */
typedef struct {
    char umi[16];
    uint32_t gene;
    uint64_t count;      /* umigene_counts value (mutated by BarcodeDupMarker::new) */
    uint64_t raw_count;  /* value before any move */
    uint8_t min_utype;   /* umigene_min_key */
    uint64_t min_qname;
    int64_t corr;        /* umi_corrections: index of the target entry, -1 = none */
    uint8_t low;         /* low_support_umigenes membership */
} keyrec;

typedef struct {
    bytemap map; /* key = umi bytes + gene (4 bytes) ; v0 = index into recs */
    keyrec *recs;
    uint64_t n, cap;
    uint32_t umi_len;
} dupstate;

static void make_key(uint8_t *k, const char *umi, uint32_t umi_len, uint32_t gene) {
    memcpy(k, umi, umi_len);
    memcpy(k + umi_len, &gene, 4);
}

static int64_t ds_find(const dupstate *ds, const char *umi, uint32_t gene) {
    uint8_t k[BYTEMAP_KEY_MAX];
    make_key(k, umi, ds->umi_len, gene);
    bytemap_slot *s = bytemap_find(&ds->map, k, ds->umi_len + 4);
    return s ? s->v0 : -1;
}

static keyrec *ds_entry(dupstate *ds, const char *umi, uint32_t gene, int *fresh) {
    uint8_t k[BYTEMAP_KEY_MAX];
    make_key(k, umi, ds->umi_len, gene);
    bytemap_slot *s = bytemap_entry(&ds->map, k, ds->umi_len + 4, fresh);
    if (*fresh) {
        if (ds->n == ds->cap) {
            ds->cap = ds->cap ? ds->cap * 2 : 64;
            ds->recs = (keyrec *)realloc(ds->recs, ds->cap * sizeof(keyrec));
        }
        keyrec *r = &ds->recs[ds->n];
        memset(r, 0, sizeof(*r));
        memcpy(r->umi, umi, ds->umi_len);
        r->gene = gene;
        r->corr = -1;
        s->v0 = (int64_t)ds->n;
        ds->n++;
    }
    return &ds->recs[s->v0];
}

/* UMIs of several lengths in one group (UmiExtractor::extract_umi gives short reads shorter UMIs,
 * cr_types/src/rna_read.rs:103-138): the callers pad the shorter ones with NUL bytes up to ds->umi_len.  A UmiSeq is its
 * bytes AND its length, so padded strings compare and hash exactly like the reference's keys; only the code that walks
 * over the bases needs the real length. */
static uint32_t umi_real_len(const char *umi, uint32_t max_len) {
    uint32_t l = 0;
    while (l < max_len && umi[l] != 0) l++;
    return l;
}

/* mark_dups.rs:19-59 */
static void correct_umis(dupstate *ds) {
    static const char nucs[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t i = 0; i < ds->n; i++) {
        keyrec *r = &ds->recs[i];
        const uint32_t L = ds->umi_len;
        char test_umi[16];
        memcpy(test_umi, r->umi, L);
        uint64_t best_dest_count = r->count;
        char best_dest_umi[16];
        memcpy(best_dest_umi, r->umi, L);
        int64_t best_idx = (int64_t)i;
        for (uint32_t pos = 0; pos < umi_real_len(r->umi, L); pos++) { /* `for pos in 0..umi.len()` */
            for (int c = 0; c < 4; c++) {
                if (nucs[c] == r->umi[pos]) continue;
                test_umi[pos] = nucs[c];
                int64_t j = ds_find(ds, test_umi, r->gene);
                uint64_t test_count = j >= 0 ? ds->recs[j].count : 0;
                if (test_count > best_dest_count ||
                    (test_count == best_dest_count && memcmp(test_umi, best_dest_umi, L) > 0)) {
                    memcpy(best_dest_umi, test_umi, L);
                    best_dest_count = test_count;
                    best_idx = j; /* j >= 0 here: an absent UMI has count 0 < orig_count */
                }
            }
            test_umi[pos] = r->umi[pos];
        }
        if (memcmp(r->umi, best_dest_umi, L) != 0) r->corr = best_idx;
    }
}

static _Thread_local uint32_t g_cmp_len;
static _Thread_local const keyrec *g_cmp_recs;
static int cmp_by_umi(const void *a, const void *b) {
    uint64_t ia = *(const uint64_t *)a, ib = *(const uint64_t *)b;
    int c = memcmp(g_cmp_recs[ia].umi, g_cmp_recs[ib].umi, g_cmp_len);
    if (c) return c;
    if (g_cmp_recs[ia].gene != g_cmp_recs[ib].gene) return g_cmp_recs[ia].gene < g_cmp_recs[ib].gene ? -1 : 1;
    return 0;
}

/* mark_dups.rs:87-108 (operates on the CURRENT counts, zero-count keys included) */
static void determine_low_support(dupstate *ds) {
    uint64_t n = ds->n;
    if (!n) return;
    uint64_t *order = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) order[i] = i;
    g_cmp_len = ds->umi_len;
    g_cmp_recs = ds->recs;
    qsort(order, n, sizeof(uint64_t), cmp_by_umi);
    uint64_t s = 0;
    while (s < n) {
        uint64_t e = s + 1;
        while (e < n && memcmp(ds->recs[order[e]].umi, ds->recs[order[s]].umi, ds->umi_len) == 0) e++;
        uint64_t max_count = 0;
        for (uint64_t k = s; k < e; k++)
            if (ds->recs[order[k]].count > max_count) max_count = ds->recs[order[k]].count;
        uint64_t n_max = 0;
        for (uint64_t k = s; k < e; k++)
            if (ds->recs[order[k]].count == max_count) n_max++;
        int max_is_tied = n_max >= 2;
        for (uint64_t k = s; k < e; k++)
            if (max_is_tied || ds->recs[order[k]].count < max_count) ds->recs[order[k]].low = 1;
        s = e;
    }
    free(order);
}

static int selkey_less(uint8_t ut_a, uint64_t q_a, uint8_t ut_b, uint64_t q_b) {
    /* UmiSelectKey derive(Ord): (utype, qname) */
    if (ut_a != ut_b) return ut_a < ut_b;
    return q_a < q_b;
}

/* targeted-panel UMI filter (mark_dups.rs:311-320): state of the checker, set by the tests before a run.
 * on_target[f] != 0 for the features of the target set; threshold 0 = None (no filter). */
static const uint8_t *g_on_target = NULL;
static uint32_t g_n_target_features = 0;
static uint64_t g_target_min_reads = 0;
void oracle_set_target_filter(const uint8_t *on_target, uint32_t n_features, uint64_t min_read_count) {
    g_on_target = on_target;
    g_n_target_features = n_features;
    g_target_min_reads = on_target ? min_read_count : 0;
}

uint64_t oracle_mark_dups_group(const char *umi, uint32_t umi_len, const uint8_t *umi_valid,
                                const uint32_t *feature, const uint8_t *utype,
                                const uint64_t *qname, uint64_t n, int umi_correction_enabled,
                                int filter_umis, oracle_dupinfo *dup_out,
                                oracle_umicount *umi_counts_out) {
    dupstate ds;
    memset(&ds, 0, sizeof(ds));
    ds.umi_len = umi_len;
    bytemap_init(&ds.map, n / 2 + 16);

    /* DupBuilder::observe, mark_dups.rs:128-155 */
    for (uint64_t i = 0; i < n; i++) {
        if (!umi_valid[i] || feature[i] == ORACLE_NO_FEATURE) continue;
        int fresh;
        keyrec *r = ds_entry(&ds, umi + i * umi_len, feature[i], &fresh);
        r->count += 1;
        uint8_t ut = utype ? utype[i] : 0;
        if (fresh || selkey_less(ut, qname[i], r->min_utype, r->min_qname)) {
            r->min_utype = ut;
            r->min_qname = qname[i];
        }
    }
    for (uint64_t i = 0; i < ds.n; i++) ds.recs[i].raw_count = ds.recs[i].count;

    /* BarcodeDupMarker::new, mark_dups.rs:210-214 */
    if (umi_correction_enabled) correct_umis(&ds);

    /* mark_dups.rs:226-232: count one read of each corrected UMI first */
    for (uint64_t i = 0; i < ds.n; i++)
        if (ds.recs[i].corr >= 0) {
            ds.recs[i].count -= 1;
            ds.recs[ds.recs[i].corr].count += 1;
        }
    /* mark_dups.rs:234-239 */
    if (filter_umis) determine_low_support(&ds);
    /* mark_dups.rs:241-246 */
    for (uint64_t i = 0; i < ds.n; i++)
        if (ds.recs[i].corr >= 0) {
            ds.recs[i].count -= ds.recs[i].raw_count - 1;
            ds.recs[ds.recs[i].corr].count += ds.recs[i].raw_count - 1;
        }

    /* mark_dups.rs:248-268: representative-read bookkeeping */
    {
        int64_t *min_raw = (int64_t *)malloc(sizeof(int64_t) * (ds.n ? ds.n : 1));
        for (uint64_t i = 0; i < ds.n; i++) min_raw[i] = -1;
        for (uint64_t i = 0; i < ds.n; i++) {
            int64_t c = ds.recs[i].corr;
            if (c < 0) continue;
            if (memcmp(ds.recs[i].umi, ds.recs[c].umi, umi_len) < 0 || ds.recs[c].corr >= 0) {
                if (min_raw[c] < 0 || memcmp(ds.recs[i].umi, ds.recs[min_raw[c]].umi, umi_len) < 0)
                    min_raw[c] = (int64_t)i;
            }
        }
        /* min_umi_key_corrections is built from the ORIGINAL umigene_min_key, then applied */
        uint8_t *new_ut = (uint8_t *)malloc(ds.n ? ds.n : 1);
        uint64_t *new_q = (uint64_t *)malloc(sizeof(uint64_t) * (ds.n ? ds.n : 1));
        for (uint64_t i = 0; i < ds.n; i++)
            if (min_raw[i] >= 0) {
                new_ut[i] = ds.recs[min_raw[i]].min_utype;
                new_q[i] = ds.recs[min_raw[i]].min_qname;
            }
        for (uint64_t i = 0; i < ds.n; i++)
            if (min_raw[i] >= 0) {
                ds.recs[i].min_utype = new_ut[i];
                ds.recs[i].min_qname = new_q[i];
            }
        free(min_raw);
        free(new_ut);
        free(new_q);
    }

    /* BarcodeDupMarker::process, mark_dups.rs:280-363 */
    uint64_t n_out = 0;
    for (uint64_t i = 0; i < n; i++) {
        oracle_dupinfo di;
        memset(&di, 0, sizeof(di));
        if (umi_valid[i] && feature[i] != ORACLE_NO_FEATURE) {
            int64_t raw = ds_find(&ds, umi + i * umi_len, feature[i]);
            int64_t ck = ds.recs[raw].corr >= 0 ? ds.recs[raw].corr : raw;
            const keyrec *cr = &ds.recs[ck];
            di.has_dupinfo = 1;
            di.is_corrected = ds.recs[raw].corr >= 0;
            di.is_low_support = cr->low;
            int is_min_qname = qname[i] == cr->min_qname;
            di.read_count = (uint32_t)cr->count;
            di.processed_umi = oracle_encode_2bit_u32(cr->umi, umi_real_len(cr->umi, umi_len));
            /* mark_dups.rs:311-320 (None unless oracle_set_target_filter was called); subsample rate 1.0 (stages/stubs.rs:6-8) */
            di.is_filtered_target = g_target_min_reads && cr->gene < g_n_target_features && g_on_target[cr->gene] &&
                                    (uint64_t)cr->count < g_target_min_reads && !di.is_low_support;
            di.is_umi_count = !di.is_low_support && is_min_qname && !di.is_filtered_target;
            if (di.is_umi_count && umi_counts_out) {
                oracle_umicount *u = &umi_counts_out[n_out];
                memset(u, 0, sizeof(*u));
                u->feature_idx = cr->gene;
                u->umi = di.processed_umi;
                u->read_count = di.read_count;
                u->utype = utype ? utype[i] : 0;
            }
            if (di.is_umi_count) n_out++;
        }
        if (dup_out) dup_out[i] = di;
    }

    bytemap_free(&ds.map);
    free(ds.recs);
    return n_out;
}

void oracle_correct_umis(const char *umis, uint32_t umi_len, const uint32_t *genes,
                         const uint64_t *counts, uint64_t n, int64_t *corr_out) {
    dupstate ds;
    memset(&ds, 0, sizeof(ds));
    ds.umi_len = umi_len;
    bytemap_init(&ds.map, n + 16);
    for (uint64_t i = 0; i < n; i++) {
        int fresh;
        keyrec *r = ds_entry(&ds, umis + i * umi_len, genes[i], &fresh);
        r->count = counts[i];
    }
    correct_umis(&ds);
    for (uint64_t i = 0; i < n; i++) {
        int64_t me = ds_find(&ds, umis + i * umi_len, genes[i]);
        corr_out[i] = ds.recs[me].corr;
    }
    bytemap_free(&ds.map);
    free(ds.recs);
}
