/*
 * feature.c -- oracle restatement of feature-barcode matching/correction for ONE capture of a
 * tethered pattern.  TEST INFRASTRUCTURE ONLY (see cr_oracle.h).
 *
 * Follows:
 *   cr_types/src/reference/feature_extraction.rs:21-22   FEATURE_CONF_THRESHOLD 0.975, FEATURE_MAX_QV 33
 *   cr_types/src/reference/feature_extraction.rs:34-117  correct_feature_barcode
 *   cr_types/src/reference/feature_extraction.rs:443-470 find_closest
 *   cr_types/src/reference/feature_checker.rs:8-50       compute_feature_dist
 * With a single capture the reference's per-test_seq HashMap holds at most one entry per
 * whitelist sequence, so likelihood_sum is the plain sum in (position, A<C<G<T) insertion order.
 */
#include <string.h>

#include "cr_oracle.h"

#define FEATURE_CONF_THRESHOLD 0.975
#define FEATURE_MAX_QV 33
#define ILLUMINA_QUAL_OFFSET 33

extern double pow(double, double);

static int64_t get_feature(const char *feat_seqs, uint32_t n_feat, uint32_t len, const char *seq) {
    for (uint32_t i = 0; i < n_feat; i++)
        if (memcmp(feat_seqs + (size_t)i * len, seq, len) == 0) return (int64_t)i;
    return -1;
}

int64_t oracle_correct_feature_barcode(const char *feat_seqs, uint32_t n_feat, uint32_t len,
                                       const double *feat_dist, const char *seq,
                                       const uint8_t *qual) {
    static const char NUCLEOTIDES[4] = {'A', 'C', 'G', 'T'};
    double likelihood_sum = 0.0;
    double max_likelihood = -1.0;
    int64_t best = -1;

    /* fast path: exact hit is "100%" (feature_extraction.rs:77-81) */
    int64_t exact = get_feature(feat_seqs, n_feat, len, seq);
    if (exact >= 0) {
        double likelihood = feat_dist[exact];
        likelihood_sum += likelihood;
        if (likelihood > max_likelihood) {
            max_likelihood = likelihood;
            best = exact;
        }
    } else {
        char test_seq[64];
        memcpy(test_seq, seq, len);
        for (uint32_t i = 0; i < len; i++) {
            char orig_base = test_seq[i];
            for (int b = 0; b < 4; b++) {
                if (NUCLEOTIDES[b] == orig_base) continue;
                test_seq[i] = NUCLEOTIDES[b];
                int64_t f = get_feature(feat_seqs, n_feat, len, test_seq);
                if (f >= 0) {
                    /* feature_extraction.rs:41-47 */
                    uint8_t d = (uint8_t)(qual[i] - ILLUMINA_QUAL_OFFSET);
                    double qv = (double)(d < FEATURE_MAX_QV ? d : FEATURE_MAX_QV);
                    double p_edit = pow(10.0, -qv / 10.0);
                    double likelihood = feat_dist[f] * p_edit;
                    likelihood_sum += likelihood;
                    if (likelihood > max_likelihood) {
                        max_likelihood = likelihood;
                        best = f;
                    }
                }
            }
            test_seq[i] = orig_base;
        }
    }
    /* feature_extraction.rs:113 (NaN when nothing matched -> comparison false) */
    if ((max_likelihood / likelihood_sum) >= FEATURE_CONF_THRESHOLD) return best;
    return -1;
}

int64_t oracle_find_closest_feature(const char *feat_seqs, uint32_t n_feat, uint32_t len,
                                    const double *feat_dist, const char *seq, const uint8_t *qual) {
    int64_t exact = get_feature(feat_seqs, n_feat, len, seq);
    if (exact >= 0) return exact;
    if (feat_dist) return oracle_correct_feature_barcode(feat_seqs, n_feat, len, feat_dist, seq, qual);
    return -1;
}

void oracle_compute_feature_dist(const int64_t *counts, const uint32_t *feature_type,
                                 uint32_t n_feat, double *dist_out) {
    int all_zero = 1;
    for (uint32_t i = 0; i < n_feat; i++) {
        int64_t sum = 0;
        for (uint32_t j = 0; j < n_feat; j++)
            if (feature_type[j] == feature_type[i]) sum += counts[j];
        dist_out[i] = sum > 0 ? (double)counts[i] / (double)sum : 0.0;
        if (dist_out[i] != 0.0) all_zero = 0;
    }
    if (all_zero)
        for (uint32_t i = 0; i < n_feat; i++) dist_out[i] = 1.0 / (double)n_feat;
}
