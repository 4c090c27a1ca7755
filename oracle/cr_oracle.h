/*
 * cr_oracle.h -- CPU restatement ("oracle") of the Cell Ranger barcode-correct ->
 * UMI-dedup -> feature-barcode-matrix path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so -- and there only as the checker / the timed CPU baseline, never as
 * the thing that produces the product's results.
 *
 * Every function follows a file in the reference (paths relative to
 * /root/reference/lib/rust) and cites it.  The reference is Rust and cannot be
 * compiled in this image (no rustc/cargo), so parity is pinned by the reference's
 * own inline known-answer tests, transcribed as fixtures under tests/golden/:
 *   barcode/src/corrector.rs:196-341, barcode/src/whitelist.rs:554-567,
 *   tx_annotation/src/mark_dups.rs:371-405,
 *   cr_types/src/reference/feature_extraction.rs:638-827,
 *   cr_types/src/barcode_index.rs:111-133, umi/src/info.rs, rna_read.rs:1581-1637.
 * Whole-pipeline golden files do not exist in the reference drop, so the
 * end-to-end matrix is "parity pinned by unit vectors only" (see DESIGN.md).
 *
 * Third-party arithmetic restated from published behaviour (sources absent):
 *   fastq_set 0.5.3 SSeq: byte-lexicographic Ord, encode_2bit_u32 = first base most
 *   significant, A0 C1 G2 T3 (cross-checked with lib/python/cellranger/utils.py:230-246);
 *   ordered-float 3.9.2 NotNan<f64>: total order on non-NaN doubles.
 */
#ifndef CR_ORACLE_H
#define CR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_SEQ 24 /* fastq_set SSeq capacity used by BcSegSeq (23) / UmiSeq (16) */
#define ORACLE_NO_FEATURE 0xFFFFFFFFu
#define ORACLE_MISS 0xFFFFFFFFu

/* ------------------------------------------------------------------------------------------
 * Whitelist  (barcode/src/whitelist.rs:453-546)
 *   Plain : set of sequences            -> translated == NULL
 *   Trans : map raw sequence -> translated sequence
 * ---------------------------------------------------------------------------------------- */
typedef struct oracle_whitelist oracle_whitelist;

/* keys / translated: n x len ASCII, not NUL-terminated. */
oracle_whitelist *oracle_whitelist_new(const char *keys, uint32_t n, uint32_t len,
                                       const char *translated /* nullable */);
void oracle_whitelist_free(oracle_whitelist *wl);
uint32_t oracle_whitelist_len(const oracle_whitelist *wl);
/* Whitelist::contains (whitelist.rs:518-524) */
int oracle_whitelist_contains(const oracle_whitelist *wl, const char *seq, uint32_t len);
/* Whitelist::check_and_update (whitelist.rs:494-517): returns 1 on hit and writes the
 * (possibly translated) content to out_seq (len bytes). */
int oracle_whitelist_check_and_update(const oracle_whitelist *wl, const char *seq, uint32_t len,
                                      char *out_seq);
/* Whitelist::match_to_whitelist (whitelist.rs:526-546): exact, else repair a single N. */
int oracle_whitelist_match(const oracle_whitelist *wl, const char *seq, uint32_t len, char *out_seq);

/* ------------------------------------------------------------------------------------------
 * SimpleHistogram<BcSegSeq>  (metric/src/histogram.rs:26-32) -- sequence -> i64 count
 * ---------------------------------------------------------------------------------------- */
typedef struct oracle_hist oracle_hist;
oracle_hist *oracle_hist_new(void);
void oracle_hist_free(oracle_hist *h);
void oracle_hist_observe_by(oracle_hist *h, const char *seq, uint32_t len, int64_t by);
int64_t oracle_hist_get(const oracle_hist *h, const char *seq, uint32_t len);
uint64_t oracle_hist_size(const oracle_hist *h);
/* dump (sequence, count) pairs sorted by sequence bytes; seqs_out is size() x len */
void oracle_hist_dump_sorted(const oracle_hist *h, uint32_t len, char *seqs_out, int64_t *counts_out);

/* ------------------------------------------------------------------------------------------
 * Posterior::correct_barcode  (barcode/src/corrector.rs:111-171)
 * qual may be NULL (=> every position BC_MAX_QV, expected_errors = 0).
 * Returns 1 and writes the corrected (translated) sequence, else 0.
 * Defaults (corrector.rs:102-108): max_expected_errors = DBL_MAX, threshold = 0.975.
 * ---------------------------------------------------------------------------------------- */
double oracle_probability(uint8_t qual); /* corrector.rs:167-171 */
int oracle_posterior_correct(const oracle_whitelist *wl, const oracle_hist *bc_counts,
                             const char *seq, const uint8_t *qual, uint32_t len,
                             double max_expected_barcode_errors, double bc_confidence_threshold,
                             char *out_seq);

/* ------------------------------------------------------------------------------------------
 * UMI helpers  (umi/src/info.rs:20-37 ; fastq_set encode_2bit_u32)
 * ---------------------------------------------------------------------------------------- */
int oracle_umi_is_valid(const char *seq, const uint8_t *qual, uint32_t len);
uint32_t oracle_encode_2bit_u32(const char *seq, uint32_t len);

/* ------------------------------------------------------------------------------------------
 * mark_dups for ONE (barcode, library type) group  (tx_annotation/src/mark_dups.rs)
 * Inputs are the reads that reached DupBuilder::observe / BarcodeDupMarker::process:
 *   umi      : n x umi_len ASCII
 *   umi_valid: UmiInfo.is_valid per read
 *   feature  : conf_mapped_feature or ORACLE_NO_FEATURE
 *   utype    : 0 = Txomic, 1 = NonTxomic  (umi/src/lib.rs UmiType)
 *   qname    : rank of the read header among the group's headers (must be unique)
 * Outputs (caller-allocated, n entries): see oracle_dupinfo.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t has_dupinfo;    /* process() returned Some */
    uint8_t is_corrected;
    uint8_t is_low_support;
    uint8_t is_umi_count;
    uint32_t processed_umi; /* 2-bit of corrected umi */
    uint32_t read_count;    /* umigene_counts[corrected_key] */
    uint8_t is_filtered_target; /* DupInfo::is_filtered_target_umi (mark_dups.rs:311-320) */
    uint8_t pad_[3];
} oracle_dupinfo;
/* targeted_umi_min_read_count + the target set of the feature reference for the following runs (NULL: None) */
void oracle_set_target_filter(const uint8_t *on_target, uint32_t n_features, uint64_t min_read_count);

typedef struct {
    uint32_t feature_idx;
    uint32_t umi;        /* 2-bit, first base most significant */
    uint32_t read_count;
    uint8_t utype;
} oracle_umicount;

/* umi_correction_enabled: aligner.rs:315-318 (disabled for Multiplexing Capture);
 * filter_umis: aligner.rs:270 (always true).  umi_counts_out must hold n entries;
 * returns the number written (unsorted, in read order). */
uint64_t oracle_mark_dups_group(const char *umi, uint32_t umi_len, const uint8_t *umi_valid,
                                const uint32_t *feature, const uint8_t *utype,
                                const uint64_t *qname, uint64_t n, int umi_correction_enabled,
                                int filter_umis, oracle_dupinfo *dup_out,
                                oracle_umicount *umi_counts_out);

/* correct_umis alone (mark_dups.rs:19-59) for the golden vectors: keys are (umi, gene) with
 * counts; corr_out[i] = index of the key it is corrected to, or -1. */
void oracle_correct_umis(const char *umis, uint32_t umi_len, const uint32_t *genes,
                         const uint64_t *counts, uint64_t n, int64_t *corr_out);

/* ------------------------------------------------------------------------------------------
 * Whole path on a batch of reads of one GEM well  (stage glue)
 *   pass A  make_shard: rna_read.rs:352-366, make_shard_metrics.rs:171-188
 *   pass B  barcode_correction.rs:76-99,328-345,372-407
 *   count   aligner.rs:283-334, align_and_count.rs:279-336, types.rs:180-188
 *   matrix  barcode_index.rs:20-53, count_matrix.rs:382-448
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t n;
    uint32_t cb_len, umi_len;
    const char *cb;          /* n x cb_len ASCII */
    const uint8_t *cb_qual;  /* n x cb_len */
    const char *umi;         /* n x umi_len ASCII   (may be NULL for barcode-only runs) */
    const uint8_t *umi_qual; /* n x umi_len */
    const uint32_t *feature; /* n */
    const uint8_t *lib;      /* n : library-type id (0..15) */
    const uint8_t *utype;    /* n, nullable (=> Txomic) */
} oracle_reads;

typedef struct {
    /* per read */
    char *corrected_cb;      /* n x cb_len : final barcode content (translated), undefined if !valid */
    uint8_t *bc_state;       /* n : 0 invalid, 1 ValidBeforeCorrection, 2 ValidAfterCorrection */
} oracle_bc_result;

/* Run pass A + pass B for all reads. wl[lib] gives the whitelist of each library type
 * (NULL entries for unused ids).  valid_hist[lib] / corrected_hist[lib] (caller-created, may be
 * pre-filled for multi-batch use) receive make_shard's valid_bc_counts and
 * barcode_correction's bc_counts_corrected.  If prior_override[lib] != NULL it is used as the
 * prior (bc_counts) instead of valid_hist[lib].  n_threads>1 splits the reads into contiguous
 * chunks the way Martian chunks do (barcode_correction.rs:246-263). */
int oracle_barcode_stage(const oracle_reads *reads, const oracle_whitelist *const *wl,
                         oracle_hist **valid_hist, oracle_hist **corrected_hist,
                         const oracle_hist *const *prior_override, double max_expected_errors,
                         double threshold, int n_threads, oracle_bc_result *out);

typedef struct {
    uint64_t n_barcodes;  /* columns V */
    uint32_t cb_len;
    char *barcodes;       /* V x cb_len, sorted (barcode_index.rs:40-53) */
    int64_t *indptr;      /* V+1 */
    uint64_t nnz;
    int32_t *indices;     /* nnz feature idx (stored as i64 on disk, count_matrix.rs:399) */
    int32_t *data;        /* nnz */
    uint64_t n_umi_counts;   /* molecule table (types.rs:152-160), sorted per barcode */
    uint32_t *mol_bc_col;    /* column index of each molecule */
    uint8_t *mol_lib;
    oracle_umicount *mol;
} oracle_matrix;

/* Count stage on reads whose barcodes are final (bc_state / corrected_cb from the barcode stage).
 * Columns = sorted unique keys of valid_hist[*] U corrected_hist[*].  dup_out nullable (n). */
oracle_matrix *oracle_count_stage(const oracle_reads *reads, const oracle_bc_result *bc,
                                  oracle_hist *const *valid_hist, oracle_hist *const *corrected_hist,
                                  int n_lib, uint32_t multiplexing_lib_mask, int n_threads,
                                  oracle_dupinfo *dup_out);
void oracle_matrix_free(oracle_matrix *m);

/* write_matrix_mtx body (write_matrix_market.rs:96-118) without the %metadata_json line's
 * version text (caller passes the full second line).  Returns bytes written or <0. */
/* wall-clock seconds of the spans of the last barcode / count stage call (pipeline.c: g_timing), 8 doubles */
void oracle_get_timing(double *out8);
int64_t oracle_write_mtx(const oracle_matrix *m, uint32_t n_features, const char *metadata_line,
                         const char *path);

/* ------------------------------------------------------------------------------------------
 * Feature-barcode correction  (cr_types/src/reference/feature_extraction.rs:34-117)
 *   feat_seqs : n_feat x len ASCII ; feat_dist : proportions (feature_checker.rs:8-50)
 * Single capture (tethered pattern).  Returns index of the feature or -1.
 * ---------------------------------------------------------------------------------------- */
int64_t oracle_correct_feature_barcode(const char *feat_seqs, uint32_t n_feat, uint32_t len,
                                       const double *feat_dist, const char *seq,
                                       const uint8_t *qual);
/* find_closest (feature_extraction.rs:443-470) for one capture: exact hit fast path, else
 * correction when feat_dist != NULL. */
int64_t oracle_find_closest_feature(const char *feat_seqs, uint32_t n_feat, uint32_t len,
                                    const double *feat_dist /* nullable */, const char *seq,
                                    const uint8_t *qual);
/* compute_feature_dist (feature_checker.rs:8-50): per feature type proportions. */
void oracle_compute_feature_dist(const int64_t *counts, const uint32_t *feature_type,
                                 uint32_t n_feat, double *dist_out);

/* ---- FeatureExtractor with every pattern form (cr_types/src/reference/feature_extraction.rs:176-470): tethered
 * patterns ("5P...(BC)...3P", '^' / '$', N wildcards) and bare "(BC)" patterns whose every window within one mismatch
 * of a feature is a capture; several captures go through correct_feature_barcode's replace-if-greater map.  One
 * extractor holds the definitions of ONE feature type (match_read skips the groups of other types, :376-378). */
typedef struct {
    const char *pattern;  /* FeatureDef::pattern */
    const char *sequence; /* FeatureDef::sequence */
    uint32_t index;       /* FeatureDef::index (into feat_dist) */
    uint32_t read;        /* FeatureDef::read: 0 = R1, 1 = R2 */
} oracle_feature_def;
#define ORACLE_MAX_FEATURE_IDS 16
typedef struct {
    int matched;   /* match_read returned Some */
    int corrected; /* FeatureData::corrected_barcode is Some */
    uint32_t n_ids;
    uint32_t ids[ORACLE_MAX_FEATURE_IDS]; /* FeatureData::ids (feature indices), in pattern creation order */
    uint32_t read, start, len;            /* FeatureData::barcode / qual as a span of the read */
    char corrected_barcode[64];
} oracle_feature_data;
typedef struct oracle_extractor oracle_extractor;
oracle_extractor *oracle_extractor_new(const oracle_feature_def *defs, uint32_t n_defs, const double *feat_dist /* nullable */,
                                       uint32_t n_dist, char *err, size_t err_cap);
void oracle_extractor_free(oracle_extractor *x);
uint32_t oracle_extractor_n_patterns(const oracle_extractor *x);
const char *oracle_extractor_regex(const oracle_extractor *x, uint32_t pattern);
int oracle_match_read(const oracle_extractor *x, const char *r1, const uint8_t *q1, uint32_t l1, const char *r2,
                      const uint8_t *q2, uint32_t l2, oracle_feature_data *out);
/* match_read over n rows -> feature index when ids.len() == 1, else ORACLE_NO_FEATURE (OpenMP over rows) */
void oracle_match_rows(const oracle_extractor *x, int which_read, const char *rows, const uint8_t *quals, const uint32_t *len,
                       uint32_t stride, uint64_t n, int n_threads, uint32_t *feature_out);
/* compile_pattern (:307-343) / compile_bare_patterns (:291-305): the regular expression as a string; -1 = invalid */
int oracle_compile_feature_pattern(const char *orig_pat, uint32_t length, char *out, size_t out_cap);
int oracle_compile_bare_patterns(const char *const *seqs, uint32_t n, char *out, size_t out_cap);

/* ---- MAKE_SHARD read metrics over the barcode / UMI parts of a read (cr_lib/src/make_shard_metrics.rs:263-332,
 * :355-392; constants :20-23; RnaRead::barcode_min_qual / umi_min_qual cr_types/src/rna_read.rs:738-749).
 * PercentMetric numerators / denominators as plain counts.  bc_state: the oracle_barcode_stage result
 * (0 = not on the whitelist after pass A ... see BcResult); miss_whitelist counts reads whose barcode is invalid
 * after the exact match, i.e. bc_state != 1.  The whole-read metrics (R1/R2/I1/I2 N and Q30 fractions, perfect
 * homopolymers) are restated in numpy next to their GPU test (tests/test_gpu_configs.py). */
typedef struct {
    uint64_t sequenced_reads;
    uint64_t bc_n_bases, bc_bases;          /* bc_N_bases */
    uint64_t umi_n_bases, umi_bases;        /* umi_N_bases */
    uint64_t bc_q30_bases, bc_q30_den;      /* bc_bases_with_q30: q >= 30+33 over q > 2+33 */
    uint64_t umi_q30_bases, umi_q30_den;    /* umi_bases_with_q30 */
    uint64_t good_umi;                      /* Umi::is_valid */
    uint64_t has_n_barcode, has_n_umi;
    uint64_t homopolymer_barcode, homopolymer_umi;
    uint64_t low_min_qual_barcode, low_min_qual_umi; /* min quality - 33 < 10 */
    uint64_t miss_whitelist_barcode;
    uint64_t polyt_suffix_umi; /* last UMI_POLYT_SUFFIX_LENGTH = 5 bases are 'T' (make_shard_metrics.rs:23,317-321; SSeq::has_polyt_suffix
                                  is in the un-vendored fastq_set crate: the reading taken, parity unpinned) */
} oracle_shard_metrics;
void oracle_shard_metrics_scan(const char *cb, const uint8_t *cb_qual, uint32_t cb_len, const char *umi,
                               const uint8_t *umi_qual, uint32_t umi_len, const uint8_t *exact_hit /* nullable */,
                               uint64_t n, oracle_shard_metrics *out);

#ifdef __cplusplus
}
#endif
#endif
