// common.h -- internal declarations shared by the libcrgpu translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/crgpu.h"

#define CR_WAVE 64
// words of the context's 4 KB scalar page with a fixed meaning (the other users take theirs by offset: 0, 8, 16, 28, 128, 760)
#define CR_SCALAR_N_WITHOUT_FLAGS 40

// --------------------------------------------------------------------------------------------
// whitelist tables of one library type (device + host mirrors)
//
// Pigeonhole layout (DESIGN.md "K1/K2"): a barcode of `len` bases is split into a head of hA
// bases and a tail of hB = len - hA bases.
//   table A: raw keys sorted ascending (== by head, then tail).  offA[head] .. offA[head+1] is the
//            bin of all keys with that head; tailA[] holds the tails (u16).
//   table B: the same keys sorted by (tail, head).  offB[tail] bins, headB[] holds the heads.
// Any Hamming-1 neighbour of a read either shares its head (mutation in the tail -> bin A) or
// shares its tail (mutation in the head -> bin B).
// valA[pos] = canonical rank of the key at sorted position pos (NULL => rank == pos, Plain list).
// --------------------------------------------------------------------------------------------
struct WlTables {
    bool set = false;
    uint32_t n = 0;          // raw keys (deduplicated)
    uint32_t bitsA = 0;      // head bits (2*hA)
    uint32_t bitsB = 0;      // tail bits (2*hB)
    uint32_t *d_offA = nullptr;   // (1<<bitsA)+1
    uint16_t *d_tailA = nullptr;  // n
    uint32_t *d_valA = nullptr;   // n or nullptr
    uint32_t *d_offB = nullptr;   // (1<<bitsB)+1
    uint32_t *d_offE = nullptr;   // exact-lookup index, (1 << (key bits - shiftE)) + 1
    uint32_t shiftE = 0;
    uint16_t *d_headB = nullptr;  // n
    uint32_t *d_key_of_rank = nullptr;  // n_canon, only when d_valA != nullptr: a key of this list that has the rank (0xFFFFFFFF: none)
    uint32_t *d_valB = nullptr;   // n: canonical rank of every entry of table B (k_correct_sorted: no second lookup for a head mutation)
    uint32_t *d_valid = nullptr;      // n_canon valid counts
    uint32_t *d_corrected = nullptr;  // n_canon corrected counts
    uint32_t *d_prior_override = nullptr;  // n_canon or nullptr (=> prior aliases d_valid)
};

struct FeaturePattern {
    bool set = false;
    uint32_t n = 0, len = 0;
    bool has_dist = false;
    uint32_t *d_seq = nullptr;    // n packed, sorted ascending
    uint32_t *d_index = nullptr;  // n global feature index
    double *d_dist = nullptr;     // n proportions
};

// the definitions of one feature type compiled for k_extract_features (feature_extract.hip): one device allocation
struct FeatureExtractorSet {
    bool set = false;
    void *d_blob = nullptr;
    size_t off_pat = 0, off_key = 0, off_dist = 0, off_index = 0, off_ha_key = 0, off_ha_f = 0, off_hb_key = 0, off_hb_f = 0,
           off_chars = 0;
    uint32_t n_pat = 0, max_feat = 0;
    bool has_dist = false;
    bool uses_read[2] = {false, false};
    std::vector<std::string> regex;  // regex_str of every pattern as the reference would build it
    // the extractor's only pattern when that one is tethered (the usual feature reference): k_extract_tethered_lds applies
    bool one_tethered = false;
    uint32_t t_read = 0, t_anchor5 = 0, t_anchor3 = 0, t_pre_len = 0, t_suf_len = 0, t_L = 0, t_n_feat = 0;
    std::string sig;  // read, pattern, (sequence, index) pairs: two extractors with equal signatures cut the same captures
    bool t_pre_dots = false, t_suf_dots = false;  // prefix / suffix are wildcards only
    // up to four leading literal characters of the suffix (else of the prefix): a floating pattern is first looked for by them
    uint32_t t_needle = 0, t_needle_len = 0, t_needle_off = 0;
};

// Captures that the one-pattern extraction kernel found no exact feature for, kept by a call WITHOUT a feature distribution
// for the call WITH one that follows on the same rows (MAKE_SHARD's exact-match counts, then ALIGN_AND_COUNT's corrected
// matches): only used under CRGPU_OPT_BUFFERS_UNCHANGED_BETWEEN_CALLS, dropped by cr_invalidate (feature_extract.hip)
struct FxPendingSet {
    bool valid = false;
    const void *d_seq = nullptr, *d_qual = nullptr, *d_len = nullptr;
    uint64_t n = 0, n_recs = 0;
    uint32_t stride = 0;
    void *d_feature_out = nullptr, *d_n_ids_out = nullptr, *d_capture_out = nullptr;
    std::string sig;         // the definitions the captures were cut for
    void *d_recs = nullptr;  // pool block
};

struct TimedSpan {
    int slot;
    hipEvent_t start, stop;
    uint64_t units;
};

struct KeyLayout {
    bool set = false;
    uint32_t bits_bc = 0, bits_feat = 0, bits_lib = 0, bits_umi = 0;  // + 1 utype bit (LSB)
    uint32_t bits_ulen = 0;    // per-read UMI length: tag = umi_len - length, 0 bits when every UMI has umi_len bases
    uint32_t umi_min_len = 0;  // shortest UMI a read may carry (== umi_len unless crgpu_set_umi_min_len was called)
    uint32_t n_features = 0, umi_len = 0, n_libs = 0, mux_mask = 0;
    // shifts inside the primary key  [bc][feature][lib][umi length tag][umi][nontx]: the tag sits right above the UMI, so
    // that (key >> sh_lib()) separates UMIs of different lengths like different libraries (they are different UmiSeqs)
    uint32_t sh_umi() const { return 1; }
    uint32_t sh_lib() const { return 1 + bits_umi; }                 // (library, length tag) as one field
    uint32_t sh_libid() const { return 1 + bits_umi + bits_ulen; }   // the library id proper
    uint32_t sh_feat() const { return 1 + bits_umi + bits_ulen + bits_lib; }
    uint32_t sh_bc() const { return 1 + bits_umi + bits_ulen + bits_lib + bits_feat; }
    uint32_t total_bits() const { return 1 + bits_umi + bits_ulen + bits_lib + bits_feat + bits_bc; }
};

// Compact (index, barcode, flags) records of the reads K1's table lookup kernel found no whitelist entry for:
// crgpu_correct_dev on the same buffers reads them instead of scanning idx and gathering cb/flags again.
struct MissRecords {
    bool valid = false;
    const uint32_t *d_cb = nullptr;
    const uint8_t *d_flags = nullptr;
    const uint32_t *d_idx = nullptr;
    uint64_t n = 0, first = 0;  // reads [0, first) (the sampling batch) are not covered
    int ulib = 0;               // the library of the call that wrote the records
    uint32_t regions = 0, cap = 0;  // one region per wave of the lookup kernel, cap records each
    uint32_t *d_i = nullptr, *d_key = nullptr, *d_count = nullptr;  // d_count[regions] = overflow flag
    uint8_t *d_fl = nullptr;
};

// digit plan of the onesweep sort: which bits every pass sorts on (sort.hip)
#define OS_MAX_PASSES 8
#define RADIX_MAX 512  // passes of 9 bits are used when they save a whole pass (61-bit keys: 8+8+9+9+9+9+9)
struct SweepPlan {
    uint32_t n_passes;
    uint32_t shift[OS_MAX_PASSES], mask[OS_MAX_PASSES];
};
// digit histograms of all passes that k_build_keys counted while it wrote the keys (consumed by the next sort of
// exactly these keys, which then skips its own histogram read)
struct KeyHistograms {
    bool valid = false;
    const uint64_t *d_keys = nullptr;
    uint64_t n = 0;
    SweepPlan plan{};
    uint32_t *d_hist = nullptr;  // OS_MAX_PASSES x RADIX_MAX, pool block
};

// CRGPU_OPT_DENSE_BARCODE_KEYS: the barcode field of a molecule key holds the barcode's POSITION IN THE BARCODE INDEX (the
// matrix column: the canonical barcodes with a non-zero VALID or CORRECTED count in any library, ascending --
// cr_types/src/barcode_index.rs:20-53) instead of its rank on the whitelist: ~2 * 10^5 values (18 bits) where the
// 3M-february-2018 list needs 23 and a GelBeadAndProbe product space 24+.  Built from the tables when the first key is
// built, dropped by everything that changes a table (dedup.hip: cr_dense_ensure / cr_dense_drop).
struct DenseIndex {
    bool on = false, valid = false;
    uint32_t V = 0;               // columns
    uint32_t canon_bits = 0;      // bits_bc of the whitelist-rank layout (restored on drop)
    // rank -> column as a rank/select bitmap: per 64 ranks {presence bits (x = low, y = high word), columns before the word (z)}:
    // 16 bytes per 64 ranks = 1.7 MB for the 6.8 M-entry list, resident in every L2 (a u32 per rank was 27 MB and cost the key
    // builder +3.6 ms per 1 B reads in gathers)
    uint4 *d_fwd = nullptr;
    uint32_t *d_back = nullptr;   // V: column -> rank
    std::vector<uint32_t> h_back;
};

struct CrComm;  // comm.hip: RCCL communicator or in-process group of this context (NULL: single GPU)

struct crgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // second in-order stream + fork / join events: the count stage runs the search for low-support candidates beside the
    // UMI correction (dedup.hip, CrFork); created with the context, idle otherwise
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_aux = nullptr;
    std::string err;
    std::recursive_mutex mu;  // every entry point holds it (CR_ENTER): calls of several host threads are serialised
    int n_ranks = 1, rank = 0;
    CrComm *comm = nullptr;
    bool trust_buffers = false;  // CRGPU_OPT_BUFFERS_UNCHANGED_BETWEEN_CALLS

    // canonical barcode space
    bool canon_set = false;
    uint32_t n_canon = 0, cb_len = 0;
    std::vector<uint32_t> canon_sorted;  // packed, ascending
    std::vector<uint32_t> canon_order;   // rank -> caller position

    // segmented barcode construct (crgpu_set_barcode_segments): the canonical space is the product of the segments'
    // whitelists, rank = mixed radix of the segment ranks, first segment most significant; canon_sorted stays empty
    uint32_t n_segments = 0;
    uint32_t seg_n[CRGPU_MAX_SEGMENTS] = {0}, seg_len[CRGPU_MAX_SEGMENTS] = {0};
    std::vector<uint32_t> seg_seq[CRGPU_MAX_SEGMENTS];  // packed, ascending

    WlTables wl[CRGPU_MAX_LIB];
    FeaturePattern pat[CRGPU_MAX_LIB];
    FeatureExtractorSet fx[CRGPU_MAX_LIB];
    FxPendingSet fxp;
    uint32_t *d_canon_keys = nullptr;         // n_canon packed canonical barcodes, ascending (rank -> sequence)
    unsigned long long *d_hot_image = nullptr;  // K1's hot-barcode table (HOT_SLOTS entries) + 256 u32 of scratch
    // the records of the last CR_REC_SETS pass-A calls (the libraries of a well are looked up one after the other, all of them
    // before the first pass B): a set is dropped when pass B has used it, when a later pass A gets the same buffers, and with
    // everything else that describes the caller's buffers (cr_invalidate)
#define CR_REC_SETS 4
    MissRecords recs[CR_REC_SETS];
    uint32_t rec_next = 0;  // the set the next pass A overwrites when none is free
    KeyHistograms ghist;
    std::map<const void *, size_t> lds_attr_done;  // kernels whose dynamic-LDS limit was raised on this context's device (to how much)
    uint32_t n_xcc = 0;                    // XCDs that receive workgroups (probed by the first onesweep sort); 0 = unknown
    uint64_t last_distinct_keys = 0, last_low_support_candidates = 0;  // of the last count call
    uint64_t sort_refinished = 0;          // sorts whose finishing pass met a run too long for it and that were redone on all bits
    uint64_t k1_split_rounds = 0;          // table rounds of K1 whose histogram was split (table slots in LDS + staged cold hits)
    uint64_t feature_resumed_reads = 0;    // captures a pass with a distribution took from the records of the pass without one
    uint64_t feature_fast_launches = 0;    // crgpu_extract_features_dev calls that took k_extract_tethered_lds
    uint64_t feature_reads_requeued = 0;   // reads k_extract_features handed to the wide-map launch
    uint64_t comm_bytes[3] = {0, 0, 0};    // bytes this rank put into C1 (table all-reduce), C2 (key exchange), C3 (triplet gather)
    uint64_t sort_fallbacks = 0;           // sorts whose look-back watchdog fired and that were finished by the classic passes

    double max_expected_errors = 1.7976931348623157e308;  // corrector.rs:104 (f64::MAX)
    double confidence_threshold = 0.975;                   // corrector.rs:83
    double *d_ptab = nullptr;  // 128 entries: probability(q) for q = 0..127 (corrector.rs:167-171)

    KeyLayout layout;   // bits_bc follows the dense index while one is valid
    DenseIndex dense;
    // targeted-panel UMI filter (mark_dups.rs:311-320): on-target flag per feature + minimum read count (0 = None)
    uint8_t *d_on_target = nullptr;
    uint32_t n_target_features = 0;
    uint64_t target_min_reads = 0;

    // scratch
    uint32_t *d_scalars = nullptr;  // small device counters
    uint32_t *d_sort_hist = nullptr;  // RADIX x 2048 block histograms of the radix passes
    void *d_scratch = nullptr;      // growable workspace
    uint64_t scratch_bytes = 0;
    struct PoolBlock {
        void *p;
        uint64_t bytes;
        bool in_use;
    };
    std::vector<PoolBlock> pool;
    bool barcode_summary_on = false;  // crgpu_count_keys_dev keeps the corrected-read table (crgpu_enable_barcode_summary)
    uint64_t pool_budget = 64ull << 30;  // bytes the pool may hold before it starts re-using larger blocks (set at create)

    // timing ledger
    bool timing = false;
    double acc_ms[CRGPU_T_NSLOTS] = {0};
    uint64_t acc_launches[CRGPU_T_NSLOTS] = {0};
    uint64_t acc_units[CRGPU_T_NSLOTS] = {0};  // elements (reads / keys) the timed launches processed
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> event_pool;
};

int cr_fail(crgpu_ctx *ctx, int code, const char *fmt, ...);

// Scope of an entry point: locks the context and makes its device current (pool allocations, hipFuncSetAttribute and
// launches act on the calling thread's current device, which need not be the context's when a host thread serves
// several GPUs); the caller's device is restored on exit.
struct CrEnter {
    crgpu_ctx *c;
    int prev = -1;
    explicit CrEnter(crgpu_ctx *ctx) : c(ctx) {
        c->mu.lock();
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != c->device && hipSetDevice(c->device) == hipSuccess) prev = cur;
    }
    ~CrEnter() {
        if (prev >= 0) (void)hipSetDevice(prev);
        c->mu.unlock();
    }
    CrEnter(const CrEnter &) = delete;
    CrEnter &operator=(const CrEnter &) = delete;
};
#define CR_ENTER(ctx) CrEnter _cr_enter(ctx)
// forget every by-product kept for the next call (K1's miss records, the key histograms)
void cr_invalidate(crgpu_ctx *ctx);
// dedup.hip: (re)build the dense barcode index from the tables when the option is on (no-op otherwise) / forget it
int cr_dense_ensure(crgpu_ctx *ctx);
void cr_dense_drop(crgpu_ctx *ctx);
void cr_dense_free(crgpu_ctx *ctx);
void cr_feature_extractors_free(crgpu_ctx *ctx);  // feature_extract.hip
void cr_drop_feature_pending(crgpu_ctx *ctx);      // feature_extract.hip
// the sequence of a canonical rank as up to 32 bases: *lo = the first min(16, cb_len) bases packed, *hi = the rest (0 when
// cb_len <= 16); whitelist.hip
void cr_rank_to_seq(const crgpu_ctx *ctx, uint32_t rank, uint32_t *lo, uint32_t *hi);
void cr_comm_destroy(crgpu_ctx *ctx);
int cr_comm_init(crgpu_ctx *ctx, int n_ranks, int rank, const void *unique_id);
// comm.hip transports (host arrays of n_ranks entries; offsets / sizes in bytes)
int cr_comm_allgather_u64(crgpu_ctx *ctx, const uint64_t *mine, uint32_t k, uint64_t *all_out);
// failure-symmetric steps of a collective (comm.hip): all ranks return together, with the failing rank's error or CRGPU_ECOMM
int cr_comm_agree(crgpu_ctx *ctx, int local_rc, const char *where);
int cr_comm_exchange_counts(crgpu_ctx *ctx, int local_rc, const uint64_t *send_cnt, uint64_t *all, uint64_t max_recv, const char *where);
int cr_comm_test_failure(crgpu_ctx *ctx);
int cr_comm_alltoallv(crgpu_ctx *ctx, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes, void *d_recv,
                      const uint64_t *recv_off, const uint64_t *recv_bytes);
int cr_partition_by_payload(crgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, const uint32_t *d_vin, uint32_t *d_vout,
                            uint64_t n, uint32_t shift);
int cr_partition_by_owner_kv(crgpu_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, const uint32_t *d_vin, uint32_t *d_vout,
                             uint64_t n, uint32_t sh_bc, uint32_t n_ranks, const uint32_t *bounds, uint64_t *counts_out);
void cr_set_thread_error(const char *msg);

#define CR_HIP(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t _e = (call);                                                                \
        if (_e != hipSuccess)                                                                  \
            return cr_fail((ctx), CRGPU_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                                \
    } while (0)

#define CR_TRY(expr)              \
    do {                          \
        int _rc = (expr);         \
        if (_rc != CRGPU_OK) return _rc; \
    } while (0)

#define CR_REQUIRE(ctx, cond, code, ...)                     \
    do {                                                     \
        if (!(cond)) return cr_fail((ctx), (code), __VA_ARGS__); \
    } while (0)

// workspace that only grows; returned pointer valid until the next cr_scratch call
int cr_scratch(crgpu_ctx *ctx, uint64_t bytes, void **out);
void cr_invalidate_range(crgpu_ctx *ctx, const void *p, uint64_t bytes);
void cr_drop_miss_records(crgpu_ctx *ctx);                  // all sets
void cr_drop_miss_records(crgpu_ctx *ctx, MissRecords &r);  // one set
// sort.hip: the plan radix_sort would use for 64-bit keys on bits [lo_bit, hi_bit); false = onesweep does not apply
bool cr_sweep_plan(uint32_t lo_bit, uint32_t hi_bit, SweepPlan *plan, uint32_t *widths);
// low bits of a molecule key of total_bits that the radix passes leave to the finishing pass (0: none)
uint32_t cr_sort_low_bits(uint32_t total_bits, uint32_t umi_bits);
bool cr_sort_finish_experiment();  // CRGPU_SORT_FINISH=1: the 16-bit finishing pass of round 2 instead of k_order_runs
int cr_order_runs(crgpu_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, uint64_t n, uint32_t low_bits, bool *fell_back);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (context, kernel): the attribute is per device
static inline void cr_allow_lds(crgpu_ctx *ctx, const void *kernel, size_t bytes) {
    size_t &have = ctx->lds_attr_done[kernel];
    if (bytes > have) {  // a later launch of the same kernel may need more than the first one did
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        have = bytes;
    }
}  // barcode.hip: forget (and release) the K1 -> K2 miss records

// Caching device pool for the per-step temporaries and results of the count stage.  hipMalloc /
// hipFree of multi-GB buffers cost far more than the kernels; blocks are recycled instead.  Reuse is
// safe without synchronisation because every kernel of a context runs on its one in-order stream.
int cr_pool_alloc(crgpu_ctx *ctx, void **out, uint64_t bytes);
void cr_pool_free(crgpu_ctx *ctx, void *p);
void cr_pool_release_all(crgpu_ctx *ctx);  // hipFree every cached block (destroy / memory pressure)

hipEvent_t cr_take_event(crgpu_ctx *ctx);  // an event of the ledger's pool (ctx.hip)
// timing scope: records a HIP event pair around the launches of one family when enabled
struct CrTimer {
    crgpu_ctx *ctx;
    int slot;
    hipEvent_t start = nullptr, stop = nullptr;
    uint64_t units;
    CrTimer(crgpu_ctx *c, int s, uint64_t units = 0);
    ~CrTimer();
};

static inline uint32_t cr_ceil_log2(uint64_t n) {
    uint32_t b = 0;
    while ((1ull << b) < n) b++;
    return b;
}

static inline uint32_t cr_grid(uint64_t n, uint32_t block, uint32_t max_blocks = 256u * 8u) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (uint32_t)g;
}

// internal entry points shared between translation units
int cr_radix_sort_u64(crgpu_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint32_t *d_vals, uint32_t *d_vals_tmp,
                      uint64_t n, uint32_t lo_bit, uint32_t hi_bit, bool *result_in_tmp);
// sort on the top key bits only when that saves passes: *low_left low bits are left to cr_finish_emit (0: fully sorted)
int cr_radix_sort_u64_top(crgpu_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint32_t *d_vals, uint32_t *d_vals_tmp,
                          uint64_t n, uint32_t hi_bit, bool *result_in_tmp, uint32_t *low_left);
int cr_radix_sort_u64_full(crgpu_ctx *ctx, uint64_t *d_keys, uint64_t *d_tmp, uint32_t *d_vals, uint32_t *d_vals_tmp, uint64_t n,
                           uint32_t hi_bit, bool *result_in_tmp);
int cr_finish_emit(crgpu_ctx *ctx, uint64_t *d_keys, uint32_t *d_vals, uint64_t n, uint32_t low_bits, uint64_t *d_ukey,
                   uint32_t *d_upos, uint64_t *nd_out, bool *fell_back);
int cr_radix_sort_u32(crgpu_ctx *ctx, uint32_t *d_keys, uint32_t *d_tmp, uint32_t *d_vals, uint32_t *d_vals_tmp,
                      uint64_t n, uint32_t lo_bit, uint32_t hi_bit, bool *result_in_tmp);
