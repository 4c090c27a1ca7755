/*
 * barcode.c -- oracle restatement of the reference's barcode whitelist + posterior corrector.
 * TEST INFRASTRUCTURE ONLY (see cr_oracle.h).
 *
 * Follows:
 *   barcode/src/whitelist.rs:453-546   Whitelist {Plain, Trans}, check_and_update, contains,
 *                                      match_to_whitelist
 *   barcode/src/corrector.rs:8-9,83,102-171   BC_MAX_QV, BASE_OPTS, Posterior, probability
 *   barcode/src/lib.rs:291-309,738-741 BarcodeSegmentState transitions; BarcodeSegment Ord is
 *                                      (state, content) so equal likelihoods tie-break on the
 *                                      (translated) sequence bytes
 *   metric/src/histogram.rs:26-32      SimpleHistogram
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bytemap.h"
#include "cr_oracle.h"

struct oracle_whitelist {
    bytemap map;   /* key = raw sequence; v0 = index into translated (or -1 for Plain) */
    uint32_t len;
    uint32_t n;
    char *translated; /* n x len or NULL */
};

oracle_whitelist *oracle_whitelist_new(const char *keys, uint32_t n, uint32_t len,
                                       const char *translated) {
    if (len == 0 || len > ORACLE_MAX_SEQ) return NULL;
    oracle_whitelist *wl = (oracle_whitelist *)calloc(1, sizeof(*wl));
    wl->len = len;
    wl->n = n;
    bytemap_init(&wl->map, n);
    if (translated) {
        wl->translated = (char *)malloc((size_t)n * len);
        memcpy(wl->translated, translated, (size_t)n * len);
    }
    for (uint32_t i = 0; i < n; i++) {
        int fresh;
        bytemap_slot *s = bytemap_entry(&wl->map, (const uint8_t *)keys + (size_t)i * len, len, &fresh);
        /* HashMap::collect keeps the LAST value for a duplicated key */
        s->v0 = translated ? (int64_t)i : -1;
    }
    return wl;
}

void oracle_whitelist_free(oracle_whitelist *wl) {
    if (!wl) return;
    bytemap_free(&wl->map);
    free(wl->translated);
    free(wl);
}

uint32_t oracle_whitelist_len(const oracle_whitelist *wl) { return wl->len; }

int oracle_whitelist_contains(const oracle_whitelist *wl, const char *seq, uint32_t len) {
    return bytemap_find(&wl->map, (const uint8_t *)seq, len) != NULL;
}

int oracle_whitelist_check_and_update(const oracle_whitelist *wl, const char *seq, uint32_t len,
                                      char *out_seq) {
    bytemap_slot *s = bytemap_find(&wl->map, (const uint8_t *)seq, len);
    if (!s) return 0;
    if (out_seq) {
        if (s->v0 >= 0)
            memcpy(out_seq, wl->translated + (size_t)s->v0 * wl->len, wl->len); /* Trans */
        else
            memcpy(out_seq, seq, len); /* Plain */
    }
    return 1;
}

int oracle_whitelist_match(const oracle_whitelist *wl, const char *seq, uint32_t len, char *out_seq) {
    /* whitelist.rs:531-545 */
    char tmp[ORACLE_MAX_SEQ];
    memcpy(tmp, seq, len);
    if (oracle_whitelist_contains(wl, tmp, len)) {
        memcpy(out_seq, tmp, len);
        return 1;
    }
    int pos_n = -1;
    for (uint32_t i = 0; i < len; i++)
        if (tmp[i] == 'N') {
            pos_n = (int)i;
            break;
        }
    if (pos_n < 0) return 0;
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    for (int b = 0; b < 4; b++) {
        tmp[pos_n] = acgt[b];
        if (oracle_whitelist_contains(wl, tmp, len)) {
            memcpy(out_seq, tmp, len);
            return 1;
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------- */

struct oracle_hist {
    bytemap map; /* v0 = count */
};

oracle_hist *oracle_hist_new(void) {
    oracle_hist *h = (oracle_hist *)calloc(1, sizeof(*h));
    bytemap_init(&h->map, 1024);
    return h;
}

void oracle_hist_free(oracle_hist *h) {
    if (!h) return;
    bytemap_free(&h->map);
    free(h);
}

void oracle_hist_observe_by(oracle_hist *h, const char *seq, uint32_t len, int64_t by) {
    int fresh;
    bytemap_slot *s = bytemap_entry(&h->map, (const uint8_t *)seq, len, &fresh);
    s->v0 += by;
}

int64_t oracle_hist_get(const oracle_hist *h, const char *seq, uint32_t len) {
    /* SimpleHistogram::get returns 0 for a missing key */
    bytemap_slot *s = bytemap_find(&h->map, (const uint8_t *)seq, len);
    return s ? s->v0 : 0;
}

uint64_t oracle_hist_size(const oracle_hist *h) { return h->map.size; }

static _Thread_local uint32_t g_sort_len;
static int cmp_seq_idx(const void *a, const void *b) {
    const bytemap_slot *const *pa = (const bytemap_slot *const *)a;
    const bytemap_slot *const *pb = (const bytemap_slot *const *)b;
    return memcmp((*pa)->key, (*pb)->key, g_sort_len);
}

void oracle_hist_dump_sorted(const oracle_hist *h, uint32_t len, char *seqs_out, int64_t *counts_out) {
    uint64_t n = h->map.size, j = 0;
    const bytemap_slot **ptrs = (const bytemap_slot **)malloc(sizeof(void *) * (n ? n : 1));
    for (uint64_t i = 0; i < h->map.cap; i++)
        if (h->map.slots[i].len) ptrs[j++] = &h->map.slots[i];
    g_sort_len = len;
    qsort(ptrs, n, sizeof(void *), cmp_seq_idx);
    for (uint64_t i = 0; i < n; i++) {
        memcpy(seqs_out + i * len, ptrs[i]->key, len);
        counts_out[i] = ptrs[i]->v0;
    }
    free(ptrs);
}

/* ---------------------------------------------------------------------------------------- */

#define BC_MAX_QV 66 /* corrector.rs:8 */
static const char BASE_OPTS[4] = {'A', 'C', 'G', 'T'}; /* corrector.rs:9 */

double oracle_probability(uint8_t qual) {
    /* corrector.rs:167-171 : 10f64.powf(-(q - 33.0) / 10.0) */
    double q = (double)qual;
    return pow(10.0, -(q - 33.0) / 10.0);
}

int oracle_posterior_correct(const oracle_whitelist *wl, const oracle_hist *bc_counts,
                             const char *seq, const uint8_t *qual, uint32_t len,
                             double max_expected_barcode_errors, double bc_confidence_threshold,
                             char *out_seq) {
    char a[ORACLE_MAX_SEQ];
    memcpy(a, seq, len);

    int have_best = 0;
    double best_like = 0.0;
    char best_seq[ORACLE_MAX_SEQ];
    double total_likelihood = 0.0;

    for (uint32_t pos = 0; pos < len; pos++) {
        /* corrector.rs:126 */
        uint8_t qv = qual ? (qual[pos] < BC_MAX_QV ? qual[pos] : BC_MAX_QV) : BC_MAX_QV;
        char existing = a[pos];
        for (int v = 0; v < 4; v++) {
            char val = BASE_OPTS[v];
            if (val == existing) continue;
            a[pos] = val;
            char trial[ORACLE_MAX_SEQ];
            if (oracle_whitelist_check_and_update(wl, a, len, trial)) {
                /* corrector.rs:135-146 */
                int64_t raw_count = bc_counts ? oracle_hist_get(bc_counts, trial, len) : 0;
                int64_t bc_count = 1 + raw_count;
                double prob_edit = oracle_probability(qv);
                double likelihood = prob_edit * (double)bc_count;
                if (!have_best) {
                    have_best = 1;
                    best_like = likelihood;
                    memcpy(best_seq, trial, len);
                } else {
                    /* old_best.max((likelihood, trial_bc)): Ord::max returns the second
                     * argument when equal; tuple order = (NotNan, BarcodeSegment{state, content}) */
                    int take_new;
                    if (likelihood > best_like)
                        take_new = 1;
                    else if (likelihood < best_like)
                        take_new = 0;
                    else
                        take_new = memcmp(trial, best_seq, len) >= 0;
                    if (take_new) {
                        best_like = likelihood;
                        memcpy(best_seq, trial, len);
                    }
                }
                total_likelihood += likelihood;
            }
        }
        a[pos] = existing;
    }

    /* corrector.rs:152 : NotNan::try_from(threshold).ok()? */
    if (isnan(bc_confidence_threshold)) return 0;

    /* corrector.rs:154 : sum of probability(q) over the UNCAPPED quals, in order */
    double expected_errors = 0.0;
    if (qual)
        for (uint32_t i = 0; i < len; i++) expected_errors += oracle_probability(qual[i]);

    if (have_best) {
        if (expected_errors < max_expected_barcode_errors &&
            best_like / total_likelihood >= bc_confidence_threshold) {
            memcpy(out_seq, best_seq, len);
            return 1;
        }
    }
    return 0;
}

/* ---- MAKE_SHARD read metrics (make_shard_metrics.rs:263-332) ------------------------------------------------ */
static int seq_has_n(const char *s, uint32_t len) {
    for (uint32_t i = 0; i < len; i++)
        if (s[i] == 'N') return 1;
    return 0;
}
static int seq_is_homopolymer(const char *s, uint32_t len) { /* every adjacent pair equal (umi/src/info.rs:57-65) */
    for (uint32_t i = 1; i < len; i++)
        if (s[i - 1] != s[i]) return 0;
    return 1;
}
static uint8_t min_qual(const uint8_t *q, uint32_t len) {
    uint8_t m = 255;
    for (uint32_t i = 0; i < len; i++)
        if (q[i] < m) m = q[i];
    return m;
}
void oracle_shard_metrics_scan(const char *cb, const uint8_t *cb_qual, uint32_t cb_len, const char *umi,
                               const uint8_t *umi_qual, uint32_t umi_len, const uint8_t *exact_hit, uint64_t n,
                               oracle_shard_metrics *out) {
    oracle_shard_metrics m;
    memset(&m, 0, sizeof(m));
    for (uint64_t r = 0; r < n; r++) {
        const char *b = cb + r * cb_len, *u = umi + r * umi_len;
        const uint8_t *bq = cb_qual + r * cb_len, *uq = umi_qual + r * umi_len;
        m.sequenced_reads++;
        for (uint32_t i = 0; i < cb_len; i++) { /* frac_n_bases / frac_q30_bases, :359-392 */
            m.bc_n_bases += b[i] == 'N';
            m.bc_bases++;
            if (bq[i] > 2 + 33) {
                m.bc_q30_den++;
                m.bc_q30_bases += bq[i] >= 30 + 33;
            }
        }
        for (uint32_t i = 0; i < umi_len; i++) {
            m.umi_n_bases += u[i] == 'N';
            m.umi_bases++;
            if (uq[i] > 2 + 33) {
                m.umi_q30_den++;
                m.umi_q30_bases += uq[i] >= 30 + 33;
            }
        }
        m.good_umi += (uint64_t)oracle_umi_is_valid(u, uq, umi_len);
        m.has_n_barcode += (uint64_t)seq_has_n(b, cb_len);
        m.has_n_umi += (uint64_t)seq_has_n(u, umi_len);
        m.homopolymer_barcode += (uint64_t)seq_is_homopolymer(b, cb_len);
        m.homopolymer_umi += (uint64_t)seq_is_homopolymer(u, umi_len);
        m.low_min_qual_barcode += (uint8_t)(min_qual(bq, cb_len) - 33) < 10; /* BARCODE_MIN_QUAL_THRESHOLD, :21,313-315 */
        m.low_min_qual_umi += (uint8_t)(min_qual(uq, umi_len) - 33) < 10;    /* UMI_MIN_QUAL_THRESHOLD, :22,316 */
        if (exact_hit) m.miss_whitelist_barcode += !exact_hit[r];
        if (umi_len >= 5) { /* UMI_POLYT_SUFFIX_LENGTH, :23,317-321 */
            int all_t = 1;
            for (uint32_t i = umi_len - 5; i < umi_len; i++) all_t &= u[i] == 'T';
            m.polyt_suffix_umi += (uint64_t)all_t;
        }
    }
    *out = m;
}
