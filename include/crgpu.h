/*
 * crgpu.h -- C ABI of libcrgpu.so, the MI355X (gfx950) barcode-correct -> UMI-dedup -> count engine.
 *
 * Drop-in boundary for Cell Ranger's `count` hot path.  The reference has no FFI seam on this
 * path (everything is in-process Rust); each entry point below names the reference
 * interface it replaces (paths relative to /root/reference/lib/rust).  INTEGRATION.md shows the
 * Rust `extern "C"` block a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success and a negative CRGPU_E* code on failure;
 *     crgpu_last_error(ctx) returns a NUL-terminated message (valid until the next call on ctx).
 *     No C++ exception crosses the ABI.
 *   - pointers named *_dev / `d_*` are DEVICE pointers (hipMalloc'ed by the caller, by torch, or by
 *     crgpu_malloc); all others are host pointers.  Inputs are caller-owned and never retained
 *     past the call unless documented (crgpu_set_whitelist copies).
 *   - sequences are 2-bit packed, A=0 C=1 G=2 T=3, FIRST base most significant
 *     (== fastq_set `encode_2bit_u32`, used by mark_dups.rs:347), right-aligned in a uint32
 *     (barcodes and UMIs of <= 16 bases).  Numeric order == byte-lexicographic order of the
 *     sequences (barcode/src/lib.rs:119-124 ordering).
 *   - quality arrays are `len` bytes per read: bits 0..6 = the FASTQ quality character
 *     (ASCII, Phred+33), bit 7 = "this base was N" (the packed code of an N base is 0).
 *   - a barcode index (`idx`) is the RANK of the canonical (translated) barcode in the ascending
 *     order of the canonical whitelist; CRGPU_MISS (0xFFFFFFFF) = not on the whitelist.
 *     crgpu_get_canon_order maps rank -> position in the caller's canon list.
 *   - library types are small ids 0..CRGPU_MAX_LIB-1 chosen by the caller (one per
 *     cr_types LibraryType in the GEM well); all libraries of a context share ONE canonical
 *     barcode space, as in the reference (Trans whitelists map onto the GEX list).
 *   - batch sizes: the barcode stage (crgpu_match_and_count*, crgpu_correct*) takes up to 2^32 - 2 reads per call
 *     (exercised with 2.5 G reads in one call), the count stage (crgpu_build_keys_dev, crgpu_count_keys_dev,
 *     crgpu_count_records_dev, crgpu_partition_keys_dev) up to 2^31 - 1 records / keys per call (exercised with 1 G);
 *     larger inputs are CRGPU_ERANGE, never truncated.
 *   - one context per (process, device, rank).  Every entry point that takes a context locks it (a recursive mutex)
 *     and makes the context's device current for the duration of the call (restoring the caller's device afterwards),
 *     so a context may be shared by several host threads -- ALIGN_AND_COUNT's four workers per chunk
 *     (cr_lib/src/stages/align_and_count.rs:698-732) -- whose calls are then executed one at a time in arrival order on
 *     the context's one stream.  Objects a call returns (crgpu_counts, crgpu_matrix*) belong to the thread that
 *     asked for them until it frees them.
 *   - the library keeps by-products of one call for the next one (K1's miss records for K2, the sort's digit histograms
 *     counted while the keys were built).  They are only used when the caller has promised, with
 *     crgpu_set_option(ctx, CRGPU_OPT_BUFFERS_UNCHANGED_BETWEEN_CALLS, 1), that the buffers it hands from one call to the
 *     next are not written in between by anything but crgpu_* calls on this context (which drop the by-products
 *     themselves); the default is 0 and every call then works from the buffers alone.  crgpu_invalidate drops them
 *     explicitly (after a write the library cannot see: a torch copy, an RCCL receive issued by the host).
 */
#ifndef CRGPU_H
#define CRGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: crgpu_records grew d_umi_len and crgpu_matrix grew barcode_seq_hi (both appended in round 2 without a bump: a binding
 * built against version 2 passes structs that are 8 bytes short); crgpu_abi_layout, crgpu_count_host, crgpu_dupinfo added.
 * The version changes whenever a public struct changes size or field order; a host checks it (and crgpu_abi_layout) once
 * at start-up. */
#define CRGPU_ABI_VERSION 3
#define CRGPU_MAX_LIB 16
#define CRGPU_MISS 0xFFFFFFFFu
#define CRGPU_NO_FEATURE 0xFFFFFFFFu

/* error codes */
#define CRGPU_OK 0
#define CRGPU_EINVAL (-1)   /* bad argument */
#define CRGPU_ENODEV (-2)   /* no usable gfx950 device / HIP runtime failure at create */
#define CRGPU_EHIP (-3)     /* a HIP call failed */
#define CRGPU_ENOMEM (-4)   /* host or device allocation failed */
#define CRGPU_ESTATE (-5)   /* call sequence error (e.g. whitelist not set) */
#define CRGPU_ERANGE (-6)   /* value does not fit the engine's key layout */
#define CRGPU_ECOMM (-7)    /* a collective failed (RCCL error, a rank of an in-process group went away) */

/* per-read flag byte */
#define CRGPU_FLAG_LIB_MASK 0x0Fu   /* bits 0..3: library-type id */
#define CRGPU_FLAG_CB_HAS_N 0x10u   /* barcode contains at least one N */
#define CRGPU_FLAG_NONTXOMIC 0x20u  /* UmiType::NonTxomic (umi/src/lib.rs); clear = Txomic */

typedef struct crgpu_ctx crgpu_ctx;

/* ---- context ------------------------------------------------------------------------------ */
int crgpu_abi_version(void);
/* The layout of a public struct as THIS build of the library sees it, for a binding to check its own declaration against
 * (Rust: size_of / offset_of of the #[repr(C)] mirror; tests/test_abi_and_host.py does it for INTEGRATION.md's blocks, the
 * ctypes table and this header).  struct_name: "crgpu_records", "crgpu_matrix", ... (every typedef struct of this header);
 * out[0] = sizeof, out[1] = alignof, out[2] = number of fields, then (offsetof, sizeof) per field in declaration order.
 * Returns the number of words the description has (writes at most cap of them), CRGPU_EINVAL for an unknown name. */
int crgpu_abi_layout(const char *struct_name, uint32_t *out, uint32_t cap);
/* The signature of SURVEY.md 8(b).  n_ranks / rank: this context's place among the GPUs that share ONE GEM well (reads
 * sharded over the ranks); unique_id (CRGPU_UNIQUE_ID_BYTES bytes, the same on every rank): the rendezvous
 * token, from crgpu_get_unique_id (one process per GPU, RCCL over xGMI: rank 0 makes it and ships the bytes to the other
 * processes by whatever channel the host has -- a file, MPI, the Martian stage args) or from crgpu_local_group_id
 * (several contexts inside one process, one host thread each).  n_ranks == 1 with unique_id == NULL: a single GPU, no
 * communicator (the collective entry points below are then no-ops / plain copies).  With n_ranks > 1 the call blocks
 * until every rank has arrived (ncclCommInitRank). */
#define CRGPU_UNIQUE_ID_BYTES 128
int crgpu_get_unique_id(void *id_out);
int crgpu_local_group_id(uint32_t n_ranks, void *id_out);
int crgpu_create(crgpu_ctx **out, int device_id, int n_ranks, int rank, const void *unique_id);
void crgpu_destroy(crgpu_ctx *ctx);
int crgpu_comm_info(crgpu_ctx *ctx, uint32_t *n_ranks_out, uint32_t *rank_out);
/* options (see the conventions above) */
#define CRGPU_OPT_BUFFERS_UNCHANGED_BETWEEN_CALLS 0
/* 1: molecule keys carry the barcode's position in the BarcodeIndex (the matrix column: canonical barcodes with a
 * non-zero VALID or CORRECTED count in any library, ascending -- cr_types/src/barcode_index.rs:20-53) instead of its rank
 * on the whitelist.  ~2 * 10^5 columns need 18 bits where the 3M-february-2018 list needs 23 and a GelBeadAndProbe product
 * space (737 K x 16) 24: the keys get shorter (one radix pass fewer on the 3M list) and layouts that need more than 64 bits
 * with whitelist ranks fit (crgpu_set_key_layout accepts them; the first key-building call fails with CRGPU_ERANGE if the
 * columns that occur do not fit either).  The index is taken from the tables when the first key is built, so the host
 * promises the reference's stage order: every read of the well has been through pass A and pass B (and, on several GPUs,
 * both tables have been all-reduced) BEFORE the first crgpu_build_keys_dev / crgpu_count_* call, and keys of one count call
 * were all built after the last change of a table.  Every call that changes a table drops the index.  Set the option
 * before crgpu_set_key_layout.  Outputs are unchanged: triplets, molecules and summaries report whitelist ranks.  A record
 * whose barcode has no read in the tables makes the key-building call fail (CRGPU_ESTATE) instead of being dropped. */
#define CRGPU_OPT_DENSE_BARCODE_KEYS 1
int crgpu_set_option(crgpu_ctx *ctx, int option, int64_t value);
int crgpu_invalidate(crgpu_ctx *ctx);
/* counters since the context was made */
#define CRGPU_STAT_SORT_FALLBACKS 0  /* sorts whose look-back watchdog fired and that the classic passes finished */
#define CRGPU_STAT_K1_SPLIT_ROUNDS 3 /* table rounds of pass A whose histogram was split: table hits counted per slot in LDS, the other hits staged */
#define CRGPU_STAT_FEATURE_READS_REQUEUED 2 /* reads of crgpu_extract_features_dev redone with the wide correction map */
#define CRGPU_STAT_FEATURE_FAST_LAUNCHES 4 /* crgpu_extract_features_dev calls served by the one-tethered-pattern LDS kernel */
#define CRGPU_STAT_FEATURE_RESUMED_READS 8 /* captures corrected from the records of the distribution-less pass (rows not read again) */
#define CRGPU_STAT_COMM_BYTES_C1 5  /* bytes this rank contributed to the table all-reduces (C1) */
#define CRGPU_STAT_COMM_BYTES_C2 6  /* bytes of molecule keys this rank put into key exchanges (C2; its own share included) */
#define CRGPU_STAT_COMM_BYTES_C3 7  /* bytes of triplets this rank sent to the root of gathers (C3) */
#define CRGPU_STAT_DISTINCT_KEYS 9          /* distinct molecule keys of the last count call (on this rank) */
#define CRGPU_STAT_LOW_SUPPORT_CANDIDATES 10 /* of those, the keys the low-support filter had to group by (barcode, UMI) */
#define CRGPU_STAT_MISS_RECORD_SETS 11       /* pass-A calls whose miss records are still kept for their pass B (at most 4) */
#define CRGPU_STAT_SORT_REFINISHED 1 /* sorts redone on all key bits because a run of equal top bits was too long for the finishing pass */
int crgpu_get_stat(crgpu_ctx *ctx, int which, uint64_t *value_out);
/* ctx may be NULL: returns the message of the last failed crgpu_create on this thread. */
const char *crgpu_last_error(const crgpu_ctx *ctx);
int crgpu_synchronize(crgpu_ctx *ctx);
/* The HIP stream (hipStream_t) every kernel of this context is launched on. */
void *crgpu_stream(crgpu_ctx *ctx);
/* device-memory helpers for hosts without their own allocator (Rust host, tests).  Backed by the
 * context's caching pool (multi-GB hipMalloc/hipFree cost more than the kernels): crgpu_free returns
 * the block to the pool, crgpu_trim gives cached blocks back to the driver.  A freed block may be
 * reused by later work on the context's stream, so synchronise other streams before freeing. */
int crgpu_malloc(crgpu_ctx *ctx, void **d_out, uint64_t bytes);
int crgpu_free(crgpu_ctx *ctx, void *d_ptr);
int crgpu_trim(crgpu_ctx *ctx);
int crgpu_memcpy_h2d(crgpu_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);
int crgpu_memcpy_d2h(crgpu_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);
int crgpu_memset(crgpu_ctx *ctx, void *d_dst, int value, uint64_t bytes);

/* ---- timing ledger (HIP events on the context's stream; used by bench.py's roofline) --------
 * Every kernel family has a slot; times accumulate while enabled. */
#define CRGPU_T_PACK 0
#define CRGPU_T_MATCH 1       /* K1 exact match + histogram */
#define CRGPU_T_CORRECT 2     /* K2 posterior correction */
#define CRGPU_T_KEYS 3        /* key building / compaction */
#define CRGPU_T_SORT 4        /* radix sort: stable scatter kernel (one span per pass) */
#define CRGPU_T_DEDUP 5       /* run-length, UMI correction, low support, counting */
#define CRGPU_T_MATRIX 6      /* CSC assembly */
#define CRGPU_T_SYNTH 7       /* synthetic data generation */
#define CRGPU_T_SORT_HIST 8   /* radix sort: digit histogram kernel (one span per pass) */
#define CRGPU_T_SCAN 9        /* small scans of block histograms / block counts */
#define CRGPU_T_COMM 10       /* collectives (C1 all-reduce, C2 key exchange, C3 gather) incl. their waiting time */
#define CRGPU_T_FEATURE 11    /* feature-barcode extraction / matching (K3, K3x) and the exact-match feature counts */
#define CRGPU_T_NSLOTS 12
int crgpu_timing_enable(crgpu_ctx *ctx, int on);
int crgpu_timing_reset(crgpu_ctx *ctx);
/* ms_out / launches_out / units_out [CRGPU_T_NSLOTS] (any may be NULL); synchronises.  units = the
 * elements (reads or keys) the timed launches of a slot processed. */
int crgpu_timing_get(crgpu_ctx *ctx, double *ms_out, uint64_t *launches_out, uint64_t *units_out);

/* ---- whitelist ---------------------------------------------------------------------------------
 * Replaces Whitelist::construct / WhitelistSource::as_whitelist (barcode/src/whitelist.rs:313-330,
 * 468-472).  keys: n x len ASCII (ACGT only).  canon: the canonical barcode list of the GEM well
 * (n_canon x len ASCII).  translate_to[i] = position in canon of key i's translation
 * (Whitelist::Trans, whitelist.rs:263-269,301-311); NULL => Whitelist::Plain and keys must equal
 * canon as a set.  The first call on a context fixes the canonical list; later calls (other
 * libraries) must pass the same canon.  len <= 16. */
int crgpu_set_whitelist(crgpu_ctx *ctx, int lib, const char *keys, uint32_t n, uint32_t len,
                        const char *canon, uint32_t n_canon, const uint32_t *translate_to);
/* Same with 2-bit packed sequences (skips ASCII parsing for very large lists). */
int crgpu_set_whitelist_packed(crgpu_ctx *ctx, int lib, const uint32_t *keys, uint32_t n, uint32_t len,
                               const uint32_t *canon, uint32_t n_canon, const uint32_t *translate_to);
int crgpu_whitelist_info(crgpu_ctx *ctx, uint32_t *n_canon_out, uint32_t *len_out);
/* order_out[rank] = position in the caller's canon list; seqs_out[rank] = packed sequence.
 * Either may be NULL.  n_canon entries each. */
int crgpu_get_canon_order(crgpu_ctx *ctx, uint32_t *order_out, uint32_t *seqs_out);

/* ---- segmented barcodes: GelBeadAndProbe and friends -------------------------------------------------------------
 * A barcode construct of several segments (BarcodeConstruct, e.g. gel bead 16 + probe 8 bases) is corrected segment by
 * segment, each against its own whitelist with its own prior (correct_barcode_in_read, BarcodeExtraction::Independent:
 * cr_lib/src/stages/barcode_correction.rs:85-99; the priors are MAKE_SHARD's valid_bc_segment_counts,
 * make_shard_metrics.rs:176-190), and the read's barcode is valid when every segment is.  Here: ONE CONTEXT PER SEGMENT
 * runs the barcode stage on that segment's bases (crgpu_pack_rows_dev with the segment's offset, crgpu_match_and_count_dev,
 * crgpu_correct_dev: its VALID table is that segment's valid_bc_segment_counts), and a COUNTING CONTEXT whose canonical
 * space is the product of the segments' whitelists takes the combined ranks:
 *   crgpu_set_barcode_segments  seg_seqs[s] = seg_n[s] packed sequences of seg_len[s] (<= 16) bases, ascending -- the
 *                               canonical lists of the segment contexts (crgpu_get_canon_order's seqs_out).  The space
 *                               has prod(seg_n) ranks (< 2^31): rank = (r0 * n1 + r1) * n2 + ..., which is the order of the
 *                               concatenated sequences (barcode/src/lib.rs:119-124).  Replaces crgpu_set_whitelist on
 *                               this context; `lib` gets VALID / CORRECTED tables over the product space.
 *   crgpu_combine_segments_dev  d_seg_idx[s] (host array of device pointers) = the segment contexts' idx arrays.
 *       after_correction == 0   d_idx_inout[i] = combined rank when every segment matched, else CRGPU_MISS; the
 *                               matched reads are counted into VALID (MakeShardHistograms::valid_bc_counts, :172-174)
 *       after_correction != 0   reads whose d_idx_inout[i] is CRGPU_MISS and whose segments are all valid now get their
 *                               combined rank and are counted into CORRECTED (corrected_barcode_counts,
 *                               barcode_correction.rs:401-407)
 * The count stage, matrices, summaries and collectives then work on this context as on any other; a matrix reports a
 * barcode as two words (crgpu_matrix::barcode_seq = the first 16 bases, barcode_seq_hi = the rest). */
#define CRGPU_MAX_SEGMENTS 4
int crgpu_set_barcode_segments(crgpu_ctx *ctx, int lib, uint32_t n_segments, const uint32_t *seg_n, const uint32_t *seg_len,
                               const uint32_t *const *seg_seqs);
int crgpu_combine_segments_dev(crgpu_ctx *ctx, int lib, const uint32_t *const *d_seg_idx, uint32_t n_segments, uint64_t n,
                               int after_correction, uint32_t *d_idx_inout);

/* ---- packing (ASCII -> 2-bit + N-flagged quality) ----------------------------------------------
 * Device-side part of RnaProcessor::process_read's slicing (cr_types/src/rna_read.rs:103-138,
 * 352-366): seq/qual are n x len ASCII (device).  packed_out n x u32, qualn_out n x len bytes,
 * flags_inout (nullable) n bytes: CRGPU_FLAG_CB_HAS_N is OR-ed in when a base is N. */
int crgpu_pack_dev(crgpu_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n, uint32_t len,
                   uint32_t *d_packed_out, uint8_t *d_qualn_out, uint8_t *d_flags_inout);
/* The same from whole read rows as the FASTQ holds them (R1: row_stride bytes of sequence / quality per read): packs
 * bases [offset, offset + len) of every row, e.g. (0, 16) for the barcode and (16, 12) for the UMI of a 28-base R1 -- the
 * host uploads R1 once and slices on the device (RnaRead's ranges, cr_types/src/rna_read.rs:103-138).  Outputs as
 * crgpu_pack_dev (qualn_out is n x len, not row_stride). */
int crgpu_pack_rows_dev(crgpu_ctx *ctx, const uint8_t *d_seq_rows, const uint8_t *d_qual_rows, uint64_t n,
                        uint32_t row_stride, uint32_t offset, uint32_t len, uint32_t *d_packed_out,
                        uint8_t *d_qualn_out, uint8_t *d_flags_inout);

/* ---- MAKE_SHARD read metrics ----------------------------------------------------------
 * The per-read quality metrics MakeShardVisitor::visit_processed_read accumulates for the barcode and UMI parts of a
 * read (cr_lib/src/make_shard_metrics.rs:263-332; frac_n_bases / frac_q30_bases :355-392; thresholds :20-23), as one
 * fused scan over the packed arrays (crgpu_pack_dev layout).  PercentMetrics come back as numerator / denominator
 * counts; they add up over batches in the caller.  d_idx (nullable): pass A's output, for miss_whitelist_barcode.
 * The whole-read metrics (R1 / R2 / I1 / I2 N and Q30 fractions, the perfect-homopolymer flags) are below. */
typedef struct {
    uint64_t sequenced_reads;
    uint64_t bc_n_bases, bc_bases;          /* bc_N_bases */
    uint64_t umi_n_bases, umi_bases;        /* umi_N_bases */
    uint64_t bc_q30_bases, bc_q30_den;      /* bc_bases_with_q30: q >= 30+33 over q > 2+33 */
    uint64_t umi_q30_bases, umi_q30_den;    /* umi_bases_with_q30 */
    uint64_t good_umi;                      /* Umi::is_valid (umi/src/info.rs:20-37) */
    uint64_t has_n_barcode, has_n_umi;
    uint64_t homopolymer_barcode, homopolymer_umi;
    uint64_t low_min_qual_barcode, low_min_qual_umi; /* min quality - 33 < 10 */
    uint64_t miss_whitelist_barcode;
    uint64_t polyt_suffix_umi;              /* the last 5 bases of the UMI are T (UMI_POLYT_SUFFIX_LENGTH, :23,317-321) */
} crgpu_shard_metrics;
int crgpu_shard_metrics_dev(crgpu_ctx *ctx, const uint32_t *d_cb, const uint8_t *d_cb_qualn, uint32_t cb_len,
                            const uint32_t *d_umi, const uint8_t *d_umi_qualn, uint32_t umi_len, const uint32_t *d_idx,
                            uint64_t n, crgpu_shard_metrics *out);

/* Whole-read metrics of MakeShardVisitor::visit_processed_read over read rows as the FASTQ holds them (n rows of
 * row_stride bytes, sequence and quality; d_len (nullable) = bases of every row, else row_stride):
 *   crgpu_rows_metrics_dev        frac_n_bases / frac_q30_bases of one read of the pair (read_N_bases, read_bases_with_q30,
 *                                 read2_*, i1_*, i2_*: make_shard_metrics.rs:266-279,355-392), as counts;
 *   crgpu_homopolymer_metrics_dev {A,C,G,T}_perfect_homopolymer (:281-300): reads whose R1 OR R2 (d_r2_rows nullable) holds
 *                                 run_len (HOMOPOLYMER_LENGTH = 15) equal bases in a row; out4 = counts for A, C, G, T.
 * PatternCheck lives in an un-vendored crate: "the pattern occurs in the read" is the reading taken; parity unpinned. */
typedef struct {
    uint64_t n_bases, bases;      /* frac_n_bases */
    uint64_t q30_bases, q30_den;  /* frac_q30_bases: q >= 30+33 over q > 2+33 */
} crgpu_rows_metrics;
int crgpu_rows_metrics_dev(crgpu_ctx *ctx, const uint8_t *d_seq_rows, const uint8_t *d_qual_rows, const uint32_t *d_len,
                           uint64_t n, uint32_t row_stride, crgpu_rows_metrics *out);
int crgpu_homopolymer_metrics_dev(crgpu_ctx *ctx, const uint8_t *d_r1_rows, uint32_t r1_stride, const uint32_t *d_r1_len,
                                  const uint8_t *d_r2_rows, uint32_t r2_stride, const uint32_t *d_r2_len, uint64_t n,
                                  uint32_t run_len, uint64_t *out4);
/* FASTQ text -> read rows (the device side of the ingest; decompression stays with the host): d_text holds
 * whole 4-line records (LF or CRLF; the last line may lack its line end).  Record r's sequence and quality go to row r
 * of d_seq_rows / d_qual_rows (row_stride bytes each, zero-padded; longer reads are cut and reported through d_len),
 * d_len_out[r] (nullable) = its length in the file.  A record whose header does not start with '@', whose third line does
 * not start with '+' or whose sequence and quality differ in length makes the call fail (CRGPU_EINVAL), as does a line
 * count that is not a multiple of four.  max_records: room in the row buffers (CRGPU_ERANGE beyond). */
int crgpu_fastq_to_rows_dev(crgpu_ctx *ctx, const uint8_t *d_text, uint64_t n_bytes, uint32_t row_stride, uint64_t max_records,
                            uint8_t *d_seq_rows, uint8_t *d_qual_rows, uint32_t *d_len_out, uint64_t *n_records_out);

/* ---- pass A: exact match + valid-barcode histogram (K1) -----------------------------------------
 * Replaces Whitelist::check_and_update per read (whitelist.rs:494-517, called from
 * rna_read.rs:352-366) and MakeShardHistograms::observe (cr_lib/src/make_shard_metrics.rs:171-188).
 * d_flags carries the library id and CB_HAS_N per read (NULL => library 0, no N).
 * d_idx_out[i] = canonical rank or CRGPU_MISS.  valid counts of the read's library accumulate in
 * the context (crgpu_get_counts / crgpu_counts_dev). */
int crgpu_match_and_count_dev(crgpu_ctx *ctx, const uint32_t *d_cb, const uint8_t *d_flags, uint64_t n,
                              uint32_t *d_idx_out);

/* ---- pass B: posterior 1-mismatch correction (K2) -----------------------------------------------
 * Replaces BarcodeCorrector::correct_barcode / Posterior::correct_barcode
 * (barcode/src/corrector.rs:48-60,111-165) as driven by correct_barcode_in_read
 * (cr_lib/src/stages/barcode_correction.rs:76-99,328-345).  Only reads with
 * d_idx_inout[i] == CRGPU_MISS are touched.  The prior is the library's valid-barcode histogram
 * accumulated so far (or the one installed by crgpu_set_counts(CRGPU_COUNTS_PRIOR)); it must be
 * complete -- over all batches and all ranks -- before this call.  d_qualn NULL => no qualities
 * (corrector.rs:126 map_or).  d_corrected_out (nullable) gets 1 for ValidAfterCorrection.
 * Corrected counts accumulate in the context.
 * When the call follows crgpu_match_and_count_dev on the SAME d_cb / d_flags / d_idx buffers and n (their contents
 * unchanged in between), the misses are taken from compact records that pass A left in the context instead of being
 * found again by a scan of d_idx (the records of the last four pass-A calls are kept: the libraries of a well are looked
 * up one after the other before the first pass B); any other call sequence scans.  The results are identical either way.
 * d_flags, when given, must carry CRGPU_FLAG_CB_HAS_N for exactly the reads whose quality bytes have bit 7 set (the pack
 * kernels write both): as in pass A the flag byte says which reads have an N, and a read without one fetches its quality
 * line only when two or more candidates have to be weighed (a single candidate wins whatever the qualities are, unless
 * the expected-error veto of crgpu_set_posterior is on).  Whitelists of more than 2 M entries take the recorded misses in
 * barcode order: one search of the two pigeonhole bins per run of equal sequences. */
int crgpu_set_posterior(crgpu_ctx *ctx, double max_expected_barcode_errors, double bc_confidence_threshold);
int crgpu_correct_dev(crgpu_ctx *ctx, const uint32_t *d_cb, const uint8_t *d_qualn, const uint8_t *d_flags,
                      uint64_t n, uint32_t *d_idx_inout, uint8_t *d_corrected_out);

/* ---- histograms ---------------------------------------------------------------------------------
 * Per-library u32[n_canon] tables indexed by canonical rank.
 *   VALID     = make_shard's valid_bc_counts (.bcc) == bc_counts prior of the corrector
 *   CORRECTED = barcode_correction's bc_counts_corrected (reads fixed in pass B)
 *   PRIOR     = the table pass B reads; aliases VALID until crgpu_set_counts(PRIOR) is called */
#define CRGPU_COUNTS_VALID 0
#define CRGPU_COUNTS_CORRECTED 1
#define CRGPU_COUNTS_PRIOR 2
int crgpu_get_counts(crgpu_ctx *ctx, int lib, int which, uint32_t *counts_out);
int crgpu_set_counts(crgpu_ctx *ctx, int lib, int which, const uint32_t *counts);
int crgpu_reset_counts(crgpu_ctx *ctx);
/* device pointer of the table, for collectives issued by the host (RCCL all-reduce of the prior) */
int crgpu_counts_dev(crgpu_ctx *ctx, int lib, int which, uint32_t **d_out);

/* ---- BARCODE_CORRECTION's join outputs (cr_lib/src/stages/barcode_correction.rs:372-448) ---------------------------------
 * crgpu_barcode_correction_metrics: what the stage's summary is made of, per library, from the VALID + CORRECTED tables:
 *   valid_reads / corrected_reads   the numerators of good_bc and corrected_bc (barcode_correction_metrics.rs:17-38,66-87:
 *                                   corrected_bc = corrected / all reads, good_bc = (valid + corrected) / all reads; the
 *                                   caller knows "all reads" of the library, reads that stay invalid are in no table),
 *   barcodes_detected, effective_barcode_diversity   BarcodeDiversityMetrics over bc_counts_corrected (:418-435;
 *                                   inverse Simpson index, metric/src/histogram.rs:161-171).
 * crgpu_total_barcode_counts: the total_barcode_counts histogram restricted to whitelist barcodes -- per barcode the sum of
 *   every library's raw valid count that reaches min_reads_to_report_bc (the join, :380-390) and of the corrected reads
 *   of all libraries when THEY reach it (the chunk's histogram, :345,360; the CORRECTED tables of the context count as one
 *   chunk: call per chunk and add up to follow a chunked run exactly).  Sequences that stay invalid (and are reported by
 *   the reference when one of them occurs min_reads times inside a chunk) are not covered: they are in no table.
 *   Ascending ranks; rank_out / count_out may be NULL to get *n_out only. */
typedef struct {
    uint64_t valid_reads, corrected_reads, barcodes_detected;
    double effective_barcode_diversity;
} crgpu_bc_correction_metrics;
int crgpu_barcode_correction_metrics(crgpu_ctx *ctx, int lib, crgpu_bc_correction_metrics *out);
int crgpu_total_barcode_counts(crgpu_ctx *ctx, int64_t min_reads_to_report_bc, uint32_t *rank_out, uint64_t *count_out,
                               uint64_t cap, uint64_t *n_out);

/* ---- collectives between the ranks of one GEM well ---------------------------------------------------
 * All of them are collective calls: every rank of the communicator must make the same call in the same order.  They run
 * on the context's stream (RCCL) and return when the result is usable by the next crgpu call.
 *
 * C1  crgpu_allreduce_counts: element-wise sum over the ranks of one histogram table, in place -- the corrector's prior
 *     must be the GLOBAL valid-barcode histogram before pass B (the make_shard join, make_shard.rs:343-358, feeding
 *     barcode_correction.rs:295-325), and the matrix columns are the barcodes seen on ANY rank (barcode_correction.rs:
 *     401-407).  lib < 0: every library that has a whitelist.  which: CRGPU_COUNTS_VALID or CRGPU_COUNTS_CORRECTED.
 * C2  crgpu_exchange_keys_dev: all-to-all of the molecule keys by contiguous barcode-rank range so that every barcode's
 *     reads meet on one GPU (what the reference gets from barcode-sorted shards + make_chunks, align_and_count.rs:505-524).
 *     The ranges are read-balanced from the all-reduced VALID + CORRECTED tables (crgpu_balanced_bounds: identical on
 *     every rank).  *d_recv_out: library-owned buffer with this rank's keys (free it with crgpu_free), *n_recv_out keys,
 *     ordered by source rank; bounds_out (nullable, n_ranks + 1 entries): the ranges used.  d_keys is left unchanged.
 * C3  crgpu_gatherv_dev: concatenation in rank order of every rank's device array on `root` (the disjoint triplet /
 *     CSC blocks of the ranks).  *d_out (root only, else NULL): library-owned, crgpu_free; bytes_out (nullable,
 *     n_ranks entries, root only): bytes received from each rank.
 *     crgpu_gather_triplets_dev: the three triplet arrays of a crgpu_counts; rank order == barcode order because the
 *     ranges of C2 are contiguous, so root can hand the result straight to crgpu_assemble_matrix_dev. */
int crgpu_barrier(crgpu_ctx *ctx);
int crgpu_allreduce_counts(crgpu_ctx *ctx, int lib, int which);
int crgpu_exchange_keys_dev(crgpu_ctx *ctx, const uint64_t *d_keys, uint64_t n_keys, uint64_t **d_recv_out,
                            uint64_t *n_recv_out, uint32_t *bounds_out);
int crgpu_gatherv_dev(crgpu_ctx *ctx, const void *d_src, uint64_t bytes, int root, void **d_out, uint64_t *bytes_out);
/* max over the ranks of a host double (bench timing) */
int crgpu_allreduce_max_f64(crgpu_ctx *ctx, double *value_inout);
/* element-wise sum over the ranks of a small host array, in place (the per-feature exact-match counts of a read-sharded
 * Feature Barcoding library before crgpu_compute_feature_dist: the join of make_shard.rs:343-358 sums them over chunks) */
int crgpu_allreduce_sum_i64(crgpu_ctx *ctx, int64_t *values_inout, uint32_t n);

/* ---- host-buffer convenience: the signatures of SURVEY.md 8(b) ------------------------------------
 * seq/qual are n x len ASCII host arrays exactly as the Rust host holds them (RnaRead raw barcode
 * and quality); the library id applies to the whole batch.  These upload, pack, run K1 / K2 and
 * download; PCIe-bound, for drop-in use, not for the bench. */
int crgpu_match_and_count(crgpu_ctx *ctx, int lib, const uint8_t *seq, const uint8_t *qual, uint64_t n,
                          uint32_t *idx_out);
int crgpu_correct(crgpu_ctx *ctx, int lib, const uint8_t *seq, const uint8_t *qual, uint64_t n,
                  uint32_t *idx_inout, uint8_t *corrected_flag_out);

/* ---- count stage --------------------------------------------------------------------------------
 * Device-resident SoA records (one GEM well).  Replaces, per (barcode, library type):
 * UmiInfo::new (umi/src/info.rs:20-37), DupBuilder::observe/build and BarcodeDupMarker::new/process
 * (tx_annotation/src/mark_dups.rs:128-363, driven by aligner.rs:283-334), and per barcode
 * BcUmiInfo::feature_counts (cr_types/src/types.rs:180-188, align_and_count.rs:312-333). */
typedef struct {
    uint64_t n;
    uint32_t umi_len;          /* <= 16 */
    const uint32_t *d_bc_idx;  /* canonical rank after pass A/B, CRGPU_MISS = invalid barcode */
    const uint32_t *d_umi;     /* 2-bit packed */
    const uint8_t *d_umi_qualn;/* n x umi_len, bit7 = N */
    const uint32_t *d_feature; /* conf-mapped feature index or CRGPU_NO_FEATURE */
    const uint8_t *d_flags;    /* library id / NONTXOMIC; nullable (library 0, Txomic) */
    const uint8_t *d_umi_len;  /* nullable: bases of every read's UMI, umi_min_len .. umi_len (crgpu_set_umi_min_len); the packed
                                  UMI holds that many bases right-aligned, d_umi_qualn keeps its stride of umi_len bytes */
    const int32_t *d_probe_idx;/* nullable: probe index of the read's confidently mapped LHS probe, CRGPU_NO_PROBE = None
                                  (RTL / Flex reads: mark_dups.rs:332-342).  Only crgpu_count_records_dev, its sharded twin and
                                  crgpu_count_host look at it: the UmiCount of a molecule carries the probe of its
                                  representative read (crgpu_counts_probe_idx) */
} crgpu_records;
#define CRGPU_NO_PROBE (-1)  /* PROBE_IDX_SENTINEL_VALUE (cr_types/src/types.rs:29) */

/* 64-bit molecule keys: the exchange unit between GPUs and the input of the
 * dedup.  Build keeps only reads that reach DupBuilder::observe (valid barcode, valid UMI,
 * feature != NONE).  d_keys_out must hold n entries; *n_keys_out = number written.
 * multiplexing_lib_mask: bit l set => library l is Multiplexing Capture (UMI correction
 * disabled, aligner.rs:315-318). */
int crgpu_set_key_layout(crgpu_ctx *ctx, uint32_t n_features, uint32_t umi_len, uint32_t n_libs,
                         uint32_t multiplexing_lib_mask);
/* Per-read UMI lengths: UmiExtractor::extract_umi (cr_types/src/rna_read.rs:103-138) gives a read that ends early
 * max(min(read_len - offset, length), min_length) bases (3' v3: 12, down to 10).  UMIs of different lengths are different
 * UmiSeqs: they never correct onto each other and never share a low-support group.  After crgpu_set_key_layout, declare the
 * shortest length with crgpu_set_umi_min_len (the key gains ceil(log2(umi_len - min + 1)) bits) and pass
 * crgpu_records.d_umi_len.  crgpu_pack_rows_var_dev slices such UMIs out of read rows: length per the formula above,
 * d_len_out[i] = it (0 when the range does not fit the read: the reference's check_range fails and the read has no UMI). */
int crgpu_set_umi_min_len(crgpu_ctx *ctx, uint32_t umi_min_len);
int crgpu_pack_rows_var_dev(crgpu_ctx *ctx, const uint8_t *d_seq_rows, const uint8_t *d_qual_rows, const uint32_t *d_read_len,
                            uint64_t n, uint32_t row_stride, uint32_t offset, uint32_t length, uint32_t min_length,
                            uint32_t *d_packed_out, uint8_t *d_qualn_out, uint8_t *d_len_out);
int crgpu_build_keys_dev(crgpu_ctx *ctx, const crgpu_records *recs, uint64_t *d_keys_out,
                         uint64_t *n_keys_out);
/* Targeted Gene Expression: DupBuilder::build(.., targeted_umi_min_read_count) with the target set of the feature
 * reference (tx_annotation/src/mark_dups.rs:156-169,311-320; threshold from mro/rna/_slfe_matrix_computer.mro:122-140): a
 * molecule of an on-target feature whose read count stays below min_read_count (and that is not low support) yields no
 * UmiCount, and its reads carry CRGPU_DUP_FILTERED_TARGET.  on_target: n_features bytes (host, copied), non-zero = in the
 * target set.  NULL or min_read_count == 0: no filter (the default).  Applies to every count call that follows. */
int crgpu_set_target_filter(crgpu_ctx *ctx, const uint8_t *on_target, uint32_t n_features, uint64_t min_read_count);
/* owner rank of a key's barcode for the all-to-all: rank r owns the contiguous canonical-rank range
 * [r*w, (r+1)*w), w = ceil(n_canon / n_ranks) -- barcode-range chunks like shardio's make_chunks
 * (align_and_count.rs:505-524) -- or, when `bounds` (host, n_ranks+1 ascending ranks, bounds[0] = 0,
 * bounds[n_ranks] >= n_canon) is given, the range [bounds[r], bounds[r+1]).  Stable partition of d_keys
 * (n) into n_ranks contiguous groups in d_keys_out; counts_out[r] = keys owned by rank r (host). */
int crgpu_partition_keys_dev(crgpu_ctx *ctx, const uint64_t *d_keys, uint64_t n, uint32_t n_ranks,
                             const uint32_t *bounds, uint64_t *d_keys_out, uint64_t *counts_out);
/* Read-balanced ranges from the VALID + CORRECTED tables (all libraries), as make_chunks balances
 * barcode ranges by record count.  Call after those tables were all-reduced: every rank then derives
 * the same bounds.  bounds_out: n_ranks + 1 entries. */
int crgpu_balanced_bounds(crgpu_ctx *ctx, uint32_t n_ranks, uint32_t *bounds_out);

/* result of the dedup: (barcode rank, feature, umi_count) triplets sorted by (barcode, feature),
 * i.e. the FeatureBarcodeCount stream in BarcodeThenFeatureOrder (types.rs:121-137), plus the
 * molecule table (UmiCount, types.rs:152-160) sorted per barcode as align_and_count.rs:314 does. */
typedef struct crgpu_counts crgpu_counts;
int crgpu_count_keys_dev(crgpu_ctx *ctx, uint64_t *d_keys_inout, uint64_t n_keys, crgpu_counts **out);
/* Same dedup straight from the records, additionally filling the per-read DupInfo the unchanged Rust host
 * needs for BAM tags (UB, duplicate flag, xf) and per-barcode metrics (mark_dups.rs:61-72,280-363;
 * tx_annotation/src/read.rs:536-590).  Output arrays are device, n entries, any may be NULL:
 *   processed_umi  2-bit corrected UMI (DupInfo::processed_umi)
 *   read_count     umigene_counts[corrected key] (the UmiCount::read_count of the read's molecule)
 *   dupflags       CRGPU_DUP_* bits; 0 for reads without DupInfo (invalid barcode / UMI, no feature).
 * The record's position in the arrays is its qname rank (read headers must be unique, SURVEY 8a'). */
#define CRGPU_DUP_HAS 0x01u          /* process() returned Some */
#define CRGPU_DUP_CORRECTED 0x02u    /* DupInfo::is_corrected */
#define CRGPU_DUP_LOW_SUPPORT 0x04u  /* DupInfo::is_low_support_umi */
#define CRGPU_DUP_UMI_COUNT 0x08u    /* DupInfo::is_umi_count: the representative read of its molecule */
#define CRGPU_DUP_FILTERED_TARGET 0x10u /* DupInfo::is_filtered_target_umi (crgpu_set_target_filter) */
int crgpu_count_records_dev(crgpu_ctx *ctx, const crgpu_records *recs, crgpu_counts **out,
                            uint32_t *d_processed_umi_out, uint32_t *d_read_count_out, uint8_t *d_dupflags_out);
/* The same for ONE GEM well sharded over the ranks of the context's communicator (collective).  Every rank
 * passes its shard -- contiguous slices of the well's read stream in rank order, so that a read's qname rank is its
 * position in that stream -- and receives (a) *out: the counts of the barcode range it owns (as crgpu_exchange_keys_dev +
 * crgpu_count_keys_dev give them; gather the triplets with crgpu_gather_triplets_dev) and (b) the DupInfo arrays of ITS
 * OWN reads: the keys go to the owners of their barcodes with their order kept, the owners dedup with the position in
 * the receive buffer as qname rank, and the packed per-read records travel back along the same routes.  The VALID and
 * CORRECTED tables must have been all-reduced (the owner ranges are derived from them on every rank). */
int crgpu_count_records_sharded_dev(crgpu_ctx *ctx, const crgpu_records *recs, crgpu_counts **out,
                                    uint32_t *d_processed_umi_out, uint32_t *d_read_count_out, uint8_t *d_dupflags_out);
int crgpu_counts_info(crgpu_ctx *ctx, const crgpu_counts *c, uint64_t *n_triplets, uint64_t *n_molecules);
/* C3 (see "collectives"): root receives every rank's triplets concatenated in rank order; the three arrays are
 * library-owned (crgpu_free each); on the other ranks they come back NULL with *n_total_out = 0. */
int crgpu_gather_triplets_dev(crgpu_ctx *ctx, const crgpu_counts *c, int root, uint32_t **d_bc_out,
                              uint32_t **d_feature_out, uint32_t **d_count_out, uint64_t *n_total_out);
/* device views (valid until crgpu_counts_free): bc rank u32[nt], feature u32[nt], count u32[nt] */
int crgpu_counts_triplets_dev(crgpu_ctx *ctx, const crgpu_counts *c, uint32_t **d_bc, uint32_t **d_feature,
                              uint32_t **d_count);
int crgpu_counts_triplets(crgpu_ctx *ctx, const crgpu_counts *c, uint32_t *bc_out, uint32_t *feature_out,
                          uint32_t *count_out);
/* The molecule table as the datasets MoleculeInfoWriter::fill appends (cr_h5/src/molecule_info.rs:972-998), one
 * entry per UmiCount in the order ALIGN_AND_COUNT emits them (barcodes ascending, inside a barcode sorted as
 * align_and_count.rs:314): gem_group (constant), barcode_idx = position of the barcode in the BarcodeIndex of this
 * context (== matrix column), feature_idx, library_idx, umi (2-bit), count (reads), umi_type (UmiType::to_u32: 0 Txomic,
 * 1 NonTxomic).  n_molecules entries each (crgpu_counts_info), host pointers, any may be NULL.  The eighth dataset,
 * probe_idx, comes from crgpu_counts_probe_idx. */
int crgpu_counts_molecule_info(crgpu_ctx *ctx, const crgpu_counts *c, uint16_t gem_group, uint16_t *gem_group_out,
                               uint64_t *barcode_idx_out, uint32_t *feature_idx_out, uint16_t *library_idx_out,
                               uint32_t *umi_out, uint32_t *count_out, uint32_t *umi_type_out);
/* UmiCount::probe_idx (cr_types/src/types.rs:152-160; the probe_idx dataset MoleculeInfoWriter::fill appends,
 * cr_h5/src/molecule_info.rs:980-987): per molecule, in the order of crgpu_counts_molecule_info / crgpu_counts_molecules, the
 * d_probe_idx of the molecule's representative read (the read with DupInfo::is_umi_count), CRGPU_NO_PROBE where that read
 * has none.  CRGPU_ESTATE when the counts were made without crgpu_records.d_probe_idx (or by crgpu_count_keys_dev: keys
 * carry no read identity). */
int crgpu_counts_probe_idx(crgpu_ctx *ctx, const crgpu_counts *c, int32_t *probe_idx_out);
/* molecule table: bc rank, library, feature, 2-bit umi, read_count, utype (0 Txomic,1 NonTxomic) */
int crgpu_counts_molecules(crgpu_ctx *ctx, const crgpu_counts *c, uint32_t *bc_out, uint8_t *lib_out,
                           uint32_t *feature_out, uint32_t *umi_out, uint32_t *read_count_out,
                           uint8_t *utype_out);

/* ---- per-barcode summary (barcode_summary.csv) --------------------------------------------------------
 * BarcodeSummary (cr_lib/src/aligner.rs:33-68), one row per (library, valid barcode) that has a read, filled as
 * AlignAndCountVisitor::visit_read_annotation does (cr_lib/src/align_metrics.rs:704-719):
 *   reads                = reads of the barcode (the context's VALID + CORRECTED histograms of that library, i.e. they
 *                          must cover exactly the reads that were counted),
 *   umis                 = reads with DupInfo::is_umi_count()  (= molecules),
 *   candidate_dup_reads  = reads with a DupInfo that is not low-support (= reads of the molecules),
 *   umi_corrected_reads  = reads with DupInfo::is_corrected.
 * The last column needs a table that crgpu_count_records_dev always keeps and crgpu_count_keys_dev keeps only after
 * crgpu_enable_barcode_summary(ctx, 1) (one more pass over the distinct keys).
 * crgpu_counts_barcode_summary: rows for barcode ranks in [rank_lo, rank_hi) (a rank's owner range in a multi-GPU run;
 * 0, UINT32_MAX = all), ordered by (library, rank).  rows_out may be NULL to get *n_rows only; CRGPU_ERANGE (with
 * *n_rows set) when cap is too small.
 * crgpu_write_barcode_summary_csv: the CSV ALIGN_AND_COUNT's join writes (align_and_count.rs:806-817): header
 * library_type,barcode,reads,umis,candidate_dup_reads,umi_corrected_reads; barcode = "SEQ-gem_group"; libraries with the
 * same library_type_order are ONE library type (their rows are summed) and rows are sorted by (library_type_order,
 * barcode), the derived Ord of the struct. */
typedef struct crgpu_barcode_summary_row {
    uint32_t barcode_rank;
    uint32_t library;
    uint64_t reads, umis, candidate_dup_reads, umi_corrected_reads;
} crgpu_barcode_summary_row;
int crgpu_enable_barcode_summary(crgpu_ctx *ctx, int on);
int crgpu_counts_barcode_summary(crgpu_ctx *ctx, const crgpu_counts *c, uint32_t rank_lo, uint32_t rank_hi,
                                 crgpu_barcode_summary_row *rows_out, uint64_t cap, uint64_t *n_rows);
int crgpu_write_barcode_summary_csv(crgpu_ctx *ctx, const crgpu_barcode_summary_row *rows, uint64_t n_rows,
                                    uint16_t gem_group, const uint32_t *library_type_order,
                                    const char *const *library_type_name, uint32_t n_libs, const char *path);
void crgpu_counts_free(crgpu_ctx *ctx, crgpu_counts *c);

/* ---- matrix (K6) ----------------------------------------------------------------------------------
 * Replaces BarcodeIndex::new (cr_types/src/barcode_index.rs:20-53) and write_matrix_h5_helper's
 * CSC assembly (cr_h5/src/count_matrix.rs:382-448).  Columns = every canonical barcode with a
 * non-zero VALID or CORRECTED count in any library, ascending.  Library-owned; host arrays. */
typedef struct {
    uint64_t n_barcodes;     /* V */
    uint64_t nnz;
    uint32_t n_features;
    uint32_t cb_len;
    const uint32_t *barcode_rank;  /* V canonical ranks (ascending) */
    const uint32_t *barcode_seq;   /* V packed sequences */
    const int64_t *indptr;         /* V + 1 */
    const int32_t *indices;        /* nnz  (written as int64 on disk, count_matrix.rs:399) */
    const int32_t *data;           /* nnz */
    const uint16_t *gem_group;     /* V gem groups of a merged matrix (crgpu_concat_matrices), else NULL */
    const uint32_t *barcode_seq_hi; /* V: bases 17.. of barcodes longer than 16 bases (segmented constructs), else NULL */
} crgpu_matrix;
/* triplets may come from several ranks (concatenated in any order of disjoint barcodes; they are
 * re-sorted by barcode here).  Host arrays. */
int crgpu_assemble_matrix(crgpu_ctx *ctx, const uint32_t *bc, const uint32_t *feature, const uint32_t *count,
                          uint64_t n_triplets, uint32_t n_features, crgpu_matrix **out);
void crgpu_matrix_free(crgpu_ctx *ctx, crgpu_matrix *m);
/* aggr-style post-processing of host matrices, as the reference does with scipy:
 * crgpu_sum_matrices    CountMatrix.merge / merge_matrices (lib/python/cellranger/matrix.py:479-482,1319-1329): element-wise
 *                       sum of two matrices of the same shape (same features, same barcodes in the same order);
 * crgpu_select_barcodes CountMatrix.select_barcodes (matrix.py:860-875): the given columns in the given order. */
int crgpu_sum_matrices(crgpu_ctx *ctx, const crgpu_matrix *a, const crgpu_matrix *b, crgpu_matrix **out);
int crgpu_select_barcodes(crgpu_ctx *ctx, const crgpu_matrix *a, const uint64_t *cols, uint64_t n_cols, crgpu_matrix **out);
/* aggr's MERGE_MOLECULES on the barcode_idx column of a sample's molecule table:
 * MoleculeInfoWriter::trim_barcodes (cr_h5/src/molecule_info.rs:890-960) + the offset of the join
 * (cr_aggr/src/merge_molecules.rs:131-330).  Retained = the barcodes of pass_filter and, unless pass_only, every barcode a
 * molecule refers to, ascending; d_barcode_idx_inout (device, n_molecules) and pass_filter_idx_inout (host, n_pass; column 0
 * of barcode_info/pass_filter) are rewritten to barcode_idx_offset + position in the retained list; retained_out (host,
 * room for n_barcodes entries, nullable) receives the old indices kept, *n_retained_out their number (the next sample's
 * offset is barcode_idx_offset + that).  The H5 container, the gem-group / library maps and the metrics stay with the host. */
int crgpu_trim_molecule_barcodes_dev(crgpu_ctx *ctx, uint64_t *d_barcode_idx_inout, uint64_t n_molecules, uint64_t n_barcodes,
                                     uint64_t *pass_filter_idx_inout, uint64_t n_pass, int pass_only,
                                     uint64_t barcode_idx_offset, uint64_t *retained_out, uint64_t *n_retained_out);
/* Several GEM wells of one sample (BASELINE configs[4]: one well per GPU): the merged matrix is the column
 * concatenation in (gem_group, barcode) order -- Barcode orders by gem group first (barcode/src/lib.rs:119-124).
 * gem_groups[i] is the group of mats[i], strictly ascending; all matrices share n_features and cb_len. */
int crgpu_concat_matrices(crgpu_ctx *ctx, const crgpu_matrix *const *mats, const uint16_t *gem_groups, uint32_t n_mats,
                          crgpu_matrix **out);
/* write_matrix_mtx body (cr_lib/src/stages/write_matrix_market.rs:80-122), uncompressed text;
 * metadata_line is the full "%metadata_json: ..." line.  gem_group suffixes barcodes.tsv rows. */
int crgpu_write_mtx(crgpu_ctx *ctx, const crgpu_matrix *m, const char *metadata_line, const char *mtx_path,
                    const char *barcodes_tsv_path, uint16_t gem_group);

/* The same assembly on the device: triplets (device arrays, sorted by (barcode rank, feature), unique
 * pairs -- what crgpu_count_keys_dev emits; per-rank outputs concatenated in rank order stay sorted
 * because crgpu_partition_keys_dev gives each rank a contiguous barcode range).  Library-owned. */
typedef struct {
    uint64_t n_barcodes;            /* V */
    uint64_t nnz;
    const uint32_t *d_barcode_rank; /* V canonical ranks, ascending */
    const int64_t *d_indptr;        /* V + 1 */
    const int32_t *d_indices;       /* nnz */
    const int32_t *d_data;          /* nnz */
} crgpu_matrix_dev;
int crgpu_assemble_matrix_dev(crgpu_ctx *ctx, const uint32_t *d_bc, const uint32_t *d_feature, const uint32_t *d_count,
                              uint64_t n_triplets, crgpu_matrix_dev **out);
void crgpu_matrix_dev_free(crgpu_ctx *ctx, crgpu_matrix_dev *m);
/* aggr-style post-processing of DEVICE matrices (the host versions are crgpu_sum_matrices / crgpu_select_barcodes):
 * crgpu_sum_matrices_dev     element-wise sum of two CSCs over the same columns (same barcode ranks in the same order;
 *                            CountMatrix.merge, lib/python/cellranger/matrix.py:479-482,1319-1329): per column a merge of the
 *                            two index-sorted entry lists, counted, scanned and written;
 * crgpu_select_barcodes_dev  the columns at the positions `cols` (host array) in the given order (matrix.py:860-875). */
int crgpu_sum_matrices_dev(crgpu_ctx *ctx, const crgpu_matrix_dev *a, const crgpu_matrix_dev *b, crgpu_matrix_dev **out);
int crgpu_select_barcodes_dev(crgpu_ctx *ctx, const crgpu_matrix_dev *a, const uint64_t *cols, uint64_t n_cols,
                              crgpu_matrix_dev **out);
/* copy to caller-allocated host arrays (any may be NULL) */
int crgpu_matrix_dev_download(crgpu_ctx *ctx, const crgpu_matrix_dev *m, uint32_t *rank_out, int64_t *indptr_out,
                              int32_t *indices_out, int32_t *data_out);

/* one-call convenience (single GPU): build keys -> dedup -> matrix */
int crgpu_count(crgpu_ctx *ctx, const crgpu_records *recs, uint32_t n_features, crgpu_matrix **out);
/* The count entry of SURVEY.md 8(b) for a host that holds its records in HOST memory (the Rust stage code after STAR
 * annotation): `recs` is a crgpu_records whose pointers are HOST arrays (same meaning, d_flags / d_umi_len nullable).
 * Uploads, builds keys, dedups, assembles the matrix (library-owned, crgpu_matrix_free) and, when per_read is not NULL,
 * fills the caller-allocated array of n crgpu_dupinfo -- DupInfo of mark_dups.rs:61-72 per read, in record order (the
 * record's position is its qname rank): what the unchanged host turns into the UB tag, the duplicate flag and xf
 * (tx_annotation/src/read.rs:536-590).  counts_out (nullable): the crgpu_counts of the call (molecule table, summary
 * rows), else it is freed.  PCIe-bound like crgpu_match_and_count / crgpu_correct: for drop-in use, not for the bench. */
typedef struct {
    uint32_t processed_umi; /* DupInfo::processed_umi, 2-bit */
    uint32_t read_count;    /* umigene_counts[corrected key] */
    uint8_t flags;          /* CRGPU_DUP_* ; 0 = process() returned None */
    uint8_t reserved[3];
} crgpu_dupinfo;
int crgpu_count_host(crgpu_ctx *ctx, const crgpu_records *recs_host, uint32_t n_features, crgpu_matrix **out,
                     crgpu_dupinfo *per_read, crgpu_counts **counts_out);

/* ---- feature-barcode matching (K3) ---------------------------------------------------------------
 * Replaces FeatureExtractor::find_closest / correct_feature_barcode for one tethered pattern
 * (cr_types/src/reference/feature_extraction.rs:34-117,443-470): feat_seqs n_feat x len ASCII,
 * feat_index[n_feat] = global feature index, feat_dist[n_feat] = compute_feature_dist proportions
 * (NULL => exact matches only).  d_seq / d_qualn: packed captures (len <= 16).
 * d_feature_out[i] = feature index or CRGPU_NO_FEATURE. */
int crgpu_set_feature_pattern(crgpu_ctx *ctx, int pattern, const char *feat_seqs, uint32_t n_feat, uint32_t len,
                              const uint32_t *feat_index, const double *feat_dist);
int crgpu_match_features_dev(crgpu_ctx *ctx, int pattern, const uint32_t *d_seq, const uint8_t *d_qualn,
                             uint64_t n, uint32_t *d_feature_out);

/* ---- feature extraction over whole reads, every pattern form (K3x) --------------------------------------------------
 * Replaces FeatureExtractor::new and FeatureExtractor::match_read (cr_types/src/reference/feature_extraction.rs:176-262,
 * :358-441) together with find_closest (:443-470) and correct_feature_barcode over any number of captures (:34-117):
 * tethered patterns ("5P" / '^', "3P" / '$', N wildcards around "(BC)"; one capture, the leftmost) and bare "(BC)"
 * patterns (every window within one mismatch of a feature of the same read and length is a capture).
 * One extractor holds the definitions of ONE feature type -- match_read skips the groups of other types (:376-378) --
 * so the caller registers one per library type with feature barcodes and routes its reads by library type.
 *   crgpu_set_feature_extractor  defs[n_defs]: pattern, sequence (A/C/G/T, <= 32 bases; N is refused), FeatureDef::index,
 *                                read (0 = R1, 1 = R2).  feat_dist (nullable) = compute_feature_dist proportions indexed
 *                                by FeatureDef::index (n_dist entries); without it only single exact captures match.
 *                                CRGPU_EINVAL with the reference's message for an invalid pattern or sequence and for
 *                                two definitions with the same read, pattern and sequence (:152-163).
 *   crgpu_compile_feature_pattern  compile_pattern (:307-343): the regular expression the reference would build, as text
 *                                (CRGPU_EINVAL for a pattern it rejects); crgpu_feature_extractor_regex returns the
 *                                expression of one compiled pattern (tethered or bare, :291-305) and the pattern count.
 *   crgpu_extract_features_dev   n read pairs as rows (crgpu_fastq_to_rows_dev layout: stride bytes per read, d_len
 *                                nullable = every row is full); a read the extractor has no pattern for may be NULL.
 *       d_feature_out[i]  the feature index when FeatureData::ids holds exactly one id (the reads that are counted:
 *                         tx_annotation read.rs:983-987, make_shard_metrics.rs:342), else CRGPU_NO_FEATURE
 *       d_n_ids_out[i]    (nullable) ids.len()
 *       d_capture_out[i]  (nullable) FeatureData::barcode / qual as a span: bit 31 = corrected_barcode is Some, bit 30 = read,
 *                         bits 8..29 = start, bits 0..7 = length; CRGPU_NO_CAPTURE when match_read returns None. */
typedef struct crgpu_feature_def {
    const char *pattern;
    const char *sequence;
    uint32_t index;
    uint32_t read;
} crgpu_feature_def;
#define CRGPU_NO_CAPTURE 0xFFFFFFFFu
int crgpu_set_feature_extractor(crgpu_ctx *ctx, int extractor, const crgpu_feature_def *defs, uint32_t n_defs,
                                const double *feat_dist, uint32_t n_dist);
int crgpu_compile_feature_pattern(const char *pattern, uint32_t length, char *regex_out, uint64_t cap);
int crgpu_feature_extractor_regex(crgpu_ctx *ctx, int extractor, uint32_t pattern, char *regex_out, uint64_t cap,
                                  uint32_t *n_patterns_out);
int crgpu_extract_features_dev(crgpu_ctx *ctx, int extractor, const uint8_t *d_r1_seq, const uint8_t *d_r1_qual,
                               const uint32_t *d_r1_len, uint32_t r1_stride, const uint8_t *d_r2_seq,
                               const uint8_t *d_r2_qual, const uint32_t *d_r2_len, uint32_t r2_stride, uint64_t n,
                               uint32_t *d_feature_out, uint32_t *d_n_ids_out, uint32_t *d_capture_out);

/* The prior of the feature-barcode correction: MAKE_SHARD counts, per feature, the reads whose
 * match_read WITHOUT a distribution yields exactly one id (cr_lib/src/make_shard_metrics.rs:336-345), and
 * compute_feature_dist turns the counts into proportions within each feature type (cr_types/src/reference/
 * feature_checker.rs:8-50; all-zero counts: 1 / n each).
 *   crgpu_feature_counts_dev      counts_inout[f] += reads with d_feature[i] == f (f < n_features; CRGPU_NO_FEATURE and
 *                                 out-of-range values are skipped): run crgpu_extract_features_dev on an extractor set
 *                                 WITHOUT feat_dist, then this; host array, accumulates over batches.
 *   crgpu_compute_feature_dist    feature_type[f] = small id of the feature's type (NULL: one type); dist_out[n_features]. */
int crgpu_feature_counts_dev(crgpu_ctx *ctx, const uint32_t *d_feature, uint64_t n, uint32_t n_features, int64_t *counts_inout);
int crgpu_compute_feature_dist(const int64_t *counts, const uint32_t *feature_type, uint32_t n_features, double *dist_out);

/* ---- synthetic workloads (bench / tests) ---------------------------------------------
 * Counter-based integer generator: read i of a given seed is identical on the host and on the
 * device.  Tables are host arrays built by cellranger_amd.synth. */
typedef struct {
    uint64_t seed;
    uint32_t cb_len, umi_len;
    uint32_t n_wl;            const uint32_t *wl_packed;     /* whitelist the reads are drawn from */
    uint32_t n_cells;         const uint32_t *cell_wl_pos;   /* whitelist positions of the cells */
                              const uint64_t *cell_cdf;      /* n_cells cumulative weights, last = 2^63 */
    uint32_t n_ambient;       const uint32_t *ambient_wl_pos;
    uint32_t n_genes;         const uint64_t *gene_cdf;      /* n_genes cumulative weights, last = 2^63 */
    uint32_t ambient_per_2_16;      /* P(read is ambient) in 1/65536 */
    uint32_t cb_err_per_2_16;       /* per-base substitution rate in 1/65536 */
    uint32_t umi_err_per_2_16;
    uint32_t n_per_2_20;            /* per-base N rate in 1/1048576 */
    uint32_t no_feature_per_2_16;   /* P(feature == NONE) in 1/65536 */
    uint32_t reads_per_umi;         /* mean reads per molecule */
    uint64_t n_total;               /* reads in the whole job (sets molecule multiplicities) */
    uint32_t n_libs;                /* library ids are drawn uniformly from [0, n_libs) */
} crgpu_synth_params;

typedef struct {
    uint32_t *cb;        /* n */
    uint8_t *cb_qualn;   /* n x cb_len */
    uint32_t *umi;       /* n */
    uint8_t *umi_qualn;  /* n x umi_len */
    uint32_t *feature;   /* n */
    uint8_t *flags;      /* n */
} crgpu_synth_out;   /* any pointer may be NULL (field not generated) */

/* reads [first, first+n) into device buffers / host buffers */
int crgpu_synth_dev(crgpu_ctx *ctx, const crgpu_synth_params *p, uint64_t first, uint64_t n,
                    const crgpu_synth_out *d_out);
int crgpu_synth_host(const crgpu_synth_params *p, uint64_t first, uint64_t n, const crgpu_synth_out *h_out);

/* Read rows of a Feature Barcoding library (BASELINE configs[3]; bench / tests): row i = row_stride random bases with plain
 * qualities whose bases [offset, offset + L) hold feat_seq[feature[i]] (2-bit packed, host array of n_feat sequences of L
 * <= 32 bases; a read whose feature is >= n_feat, e.g. CRGPU_NO_FEATURE, keeps random bases there), with substitutions
 * (err_per_2_16) and Ns (n_per_2_20) as in crgpu_synth_params.  ASCII rows as the FASTQ holds them
 * (crgpu_extract_features_dev's input).  Host and device produce the same bytes. */
int crgpu_synth_rows_dev(crgpu_ctx *ctx, uint64_t seed, uint64_t first, uint64_t n, const uint32_t *d_feature,
                         const uint64_t *feat_seq, uint32_t n_feat, uint32_t L, uint32_t offset, uint32_t row_stride,
                         uint32_t err_per_2_16, uint32_t n_per_2_20, uint8_t *d_seq_rows, uint8_t *d_qual_rows);
int crgpu_synth_rows_host(uint64_t seed, uint64_t first, uint64_t n, const uint32_t *feature, const uint64_t *feat_seq,
                          uint32_t n_feat, uint32_t L, uint32_t offset, uint32_t row_stride, uint32_t err_per_2_16,
                          uint32_t n_per_2_20, uint8_t *seq_rows, uint8_t *qual_rows);

#ifdef __cplusplus
}
#endif
#endif
