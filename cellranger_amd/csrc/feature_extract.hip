// feature_extract.hip -- K3x: FeatureExtractor::match_read over whole read rows, every pattern form.
//
// Replaces, for the definitions of ONE feature type (cr_types/src/reference/feature_extraction.rs):
//   FeatureExtractor::new / compile_pattern / compile_bare_patterns   :176-343   (host, crgpu_set_feature_extractor)
//   match_read                                                        :358-441   (k_extract_features)
//   find_closest                                                      :443-470
//   correct_feature_barcode with any number of captures               :34-117
// feature.hip keeps the fast path for a pre-cut capture of one tethered pattern; this file takes the reads as the FASTQ
// holds them (crgpu_fastq_to_rows_dev rows) and does the pattern search itself.
//
// The reference turns every pattern into a regular expression; the grammar is so small ('^'? literals-or-'.' one group
// literals-or-'.' '$'?, or a group of same-length alternatives with one '.' each) that the search is restated directly:
//   tethered  leftmost start s with prefix at s, suffix behind the L captured bases ('^': s = 0, '$': end of read);
//             one capture (:397-400);
//   bare      "(BC)": every window of L bases within one mismatch of a feature of the group, left to right (the
//             reference restarts its search one base behind the previous match, :394-396).  The window test is a
//             pigeonhole split: a window within one mismatch of a feature equals it on its first or its last half, so
//             two sorted half-key tables give the candidates and a popcount verifies them.
// Captures of a pattern stream through correct_feature_barcode's map (whitelist sequence -> best likelihood and the
// capture it came from, likelihood_sum updated by the difference on replacement) in the reference's order -- capture,
// position, A<C<G<T -- in f64 without contraction, so the sums are bit-identical.  The map is a 16-entry array per
// read; a read whose captures reach more distinct features is queued and redone by a second launch with map rows in
// global memory (one row of n_feat entries per queued read), still on the GPU.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>

#include "common.h"
#include "wl_view.h"  // U32x4: 16-byte loads from 4-byte aligned addresses

struct FxPat {
    uint32_t read, tethered, anchor5, anchor3, never;
    uint32_t pre_off, pre_len, suf_off, suf_len;  // into the character pool, '.' = any
    uint32_t L;
    uint32_t feat_off, n_feat;  // this pattern's slice of the feature arrays (keys ascending)
    uint32_t least_index;       // FeaturePattern::least_feature_index
};

struct FxView {
    const FxPat *pat;
    uint32_t n_pat;
    const char *chars;
    const uint64_t *key;     // packed feature sequences, first base in the most significant pair
    const uint32_t *index;   // FeatureDef::index
    const double *dist;      // feat_dist[FeatureDef::index]; nullptr without a distribution
    const uint32_t *ha_key;  // first-half keys ascending + the feature (position in the pattern's slice) of each
    const uint32_t *ha_f;
    const uint32_t *hb_key;  // last-half keys
    const uint32_t *hb_f;
};

struct FxRows {
    const uint8_t *seq, *qual;
    const uint32_t *len;
    uint32_t stride;
};

struct FxEntry {  // one entry of whitelist_likelihoods (:57)
    double like;
    uint32_t f, cap;
};

#define FX_LOCAL_ENTRIES 16u
#define FX_NO_CAPTURE 0xFFFFFFFFu

__device__ __forceinline__ uint32_t fx_code(uint8_t c) {  // 0..3, 4 = anything else
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

__device__ __forceinline__ int fx_find(const uint64_t *keys, uint32_t n, uint64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && keys[lo] == key) ? (int)lo : -1;
}

__device__ __forceinline__ uint32_t fx_lower(const uint32_t *keys, uint32_t n, uint32_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// is the window (key, nm: a set low bit of a pair = that base is not A/C/G/T) within one mismatch of a feature?
__device__ __forceinline__ bool fx_near(const FxView &v, const FxPat &P, uint64_t key, uint64_t nm) {
    if (__popcll(nm) > 1) return false;
    const uint32_t h = P.L >> 1;  // bases of the last half
    const uint64_t low_mask = h ? (~0ull >> (64u - 2u * h)) : 0ull;
    const uint64_t *fk = v.key + P.feat_off;
    if ((nm >> (2u * h)) == 0) {  // the first half holds no N: features that agree on it
        const uint32_t ka = (uint32_t)(key >> (2u * h));
        const uint32_t *hk = v.ha_key + P.feat_off;
        for (uint32_t j = fx_lower(hk, P.n_feat, ka); j < P.n_feat && hk[j] == ka; j++) {
            const uint64_t d = key ^ fk[v.ha_f[P.feat_off + j]];
            if (__popcll(((d | (d >> 1)) & 0x5555555555555555ull) | nm) <= 1) return true;
        }
    }
    if ((nm & low_mask) == 0) {
        const uint32_t kb = (uint32_t)(key & low_mask);
        const uint32_t *hk = v.hb_key + P.feat_off;
        for (uint32_t j = fx_lower(hk, P.n_feat, kb); j < P.n_feat && hk[j] == kb; j++) {
            const uint64_t d = key ^ fk[v.hb_f[P.feat_off + j]];
            if (__popcll(((d | (d >> 1)) & 0x5555555555555555ull) | nm) <= 1) return true;
        }
    }
    return false;
}

struct FxMap {
    FxEntry *e;
    uint32_t cap, n;
    double sum;  // likelihood_sum
    bool overflow;
    // insert_hit (:61-75)
    __device__ __forceinline__ void insert(double like, uint32_t f, uint32_t c) {
        uint32_t k = 0;
        while (k < n && e[k].f != f) k++;
        if (k < n) {
            if (like > e[k].like) {
                const double old = e[k].like;
                e[k].like = like;
                e[k].cap = c;
                sum += like - old;
            }
        } else if (n < cap) {
            e[n].like = like;
            e[n].f = f;
            e[n].cap = c;
            n++;
            sum += like;
        } else {
            overflow = true;
        }
    }
};

// one capture through correct_feature_barcode's loop body (:76-98); returns the exact feature or -1
__device__ __forceinline__ int fx_capture(const FxView &v, const FxPat &P, const double *__restrict__ pedit, const uint8_t *qual,
                                          uint32_t start, uint64_t key, uint64_t nm, FxMap &map) {
    const uint64_t *fk = v.key + P.feat_off;
    const int exact = nm ? -1 : fx_find(fk, P.n_feat, key);
    if (!v.dist) return exact;
    const double *dist = v.dist + P.feat_off;
    if (exact >= 0) {
        map.insert(dist[exact], (uint32_t)exact, start);
        return exact;
    }
    if (__popcll(nm) > 1) return -1;  // every candidate keeps another N
    for (uint32_t pos = 0; pos < P.L; pos++) {
        const uint32_t sh = 2u * (P.L - 1u - pos);
        const bool is_n = (nm >> sh) & 1ull;
        if (nm && !is_n) continue;
        const uint32_t orig = (uint32_t)(key >> sh) & 3u;
        for (uint32_t b = 0; b < 4; b++) {
            if (!is_n && b == orig) continue;
            const int f = fx_find(fk, P.n_feat, (key & ~(3ull << sh)) | ((uint64_t)b << sh));
            if (f < 0) continue;
            uint32_t qv = (uint8_t)(qual[start + pos] - 33u);  // u8 arithmetic as in :43
            qv = qv < 33u ? qv : 33u;
            map.insert(dist[f] * pedit[qv], (uint32_t)f, start);
        }
    }
    return -1;
}

// match_read for one read pair; returns false when the map overflowed (the caller queues the read)
__device__ bool fx_match_read(const FxView &v, const double *__restrict__ pedit, const FxRows &r1, const FxRows &r2, uint64_t i,
                              FxEntry *entries, uint32_t cap, uint32_t *feature_out, uint32_t *n_ids_out, uint32_t *capture_out) {
    uint32_t n_ids = 0, the_id = CRGPU_NO_FEATURE;
    bool have_wl = false, have_pm = false;
    uint32_t wl_len = 0, wl_idx = 0, wl_cap = 0, pm_len = 0, pm_idx = 0, pm_cap = 0;
    for (uint32_t p = 0; p < v.n_pat; p++) {
        const FxPat P = v.pat[p];
        const FxRows &R = P.read ? r2 : r1;
        if (!R.seq || P.never) continue;
        const uint8_t *s = R.seq + i * R.stride, *q = R.qual + i * R.stride;
        const uint32_t len = R.len ? min(R.len[i], R.stride) : R.stride;
        const uint32_t L = P.L;
        const uint64_t mask = ~0ull >> (64u - 2u * L);
        FxMap map{entries, cap, 0u, 0.0, false};
        uint32_t n_caps = 0, last_start = 0;
        int exact0 = -1;
        if (P.tethered) {
            const uint32_t need = P.pre_len + L + P.suf_len;
            if (len < need) continue;
            const char *pre = v.chars + P.pre_off, *suf = v.chars + P.suf_off;
            uint32_t s_lo = 0, s_hi = len - need;
            if (P.anchor3) s_lo = s_hi;
            if (P.anchor5) {
                if (s_lo > 0) continue;
                s_hi = 0;
            }
            uint32_t found = FX_NO_CAPTURE;
            for (uint32_t st = s_lo; st <= s_hi && found == FX_NO_CAPTURE; st++) {
                bool ok = true;
                for (uint32_t k = 0; k < P.pre_len && ok; k++) ok = pre[k] == '.' || (uint8_t)pre[k] == s[st + k];
                for (uint32_t k = 0; k < P.suf_len && ok; k++) ok = suf[k] == '.' || (uint8_t)suf[k] == s[st + P.pre_len + L + k];
                if (ok) found = st + P.pre_len;
            }
            if (found == FX_NO_CAPTURE) continue;
            uint64_t key = 0, nm = 0;
            for (uint32_t k = 0; k < L; k++) {
                const uint32_t c = fx_code(s[found + k]);
                key = (key << 2) | (c & 3u);
                nm = (nm << 2) | (c >> 2);
            }
            exact0 = fx_capture(v, P, pedit, q, found, key, nm, map);
            n_caps = 1;
            last_start = found;
        } else {
            if (len < L) continue;
            uint64_t key = 0, nm = 0;
            for (uint32_t k = 0; k < len; k++) {
                const uint32_t c = fx_code(s[k]);
                key = ((key << 2) | (c & 3u)) & mask;
                nm = ((nm << 2) | (c >> 2)) & mask;
                if (k + 1 < L) continue;
                // a position of an N is coded as A in key; fx_near and fx_capture look at nm first
                if (!fx_near(v, P, key & ~(nm * 3ull), nm)) continue;
                const uint32_t st = k + 1 - L;
                const int ex = fx_capture(v, P, pedit, q, st, key & ~(nm * 3ull), nm, map);
                if (n_caps == 0) exact0 = ex;
                n_caps++;
                last_start = st;
            }
            if (n_caps == 0) continue;
        }
        if (map.overflow) return false;
        // find_closest (:443-470)
        int hit = -1;
        uint32_t hit_start = last_start;
        if (n_caps == 1 && exact0 >= 0) {
            hit = exact0;
        } else if (v.dist) {
            double mx = -1.0;
            for (uint32_t k = 0; k < map.n; k++)
                if (map.e[k].like > mx) {
                    mx = map.e[k].like;
                    hit = (int)map.e[k].f;
                    hit_start = map.e[k].cap;
                }
            if (!((mx / map.sum) >= 0.975)) hit = -1;  // FEATURE_CONF_THRESHOLD; NaN and -inf fail
        }
        if (hit >= 0) {
            const uint32_t idx = v.index[P.feat_off + (uint32_t)hit];
            n_ids++;
            the_id = idx;
            if (!have_wl || L > wl_len || (L == wl_len && idx < wl_idx)) {  // max_by_key (len, Reverse(index)) (:415-420)
                have_wl = true;
                wl_len = L;
                wl_idx = idx;
                wl_cap = 0x80000000u | (P.read << 30) | (hit_start << 8) | L;
            }
        } else if (!have_pm || L > pm_len || (L == pm_len && P.least_index <= pm_idx)) {
            // pattern_matches (:404-408, :434-437): equal keys only come from one pattern, whose last capture wins
            have_pm = true;
            pm_len = L;
            pm_idx = P.least_index;
            pm_cap = (P.read << 30) | (last_start << 8) | L;
        }
    }
    feature_out[i] = (have_wl && n_ids == 1) ? the_id : CRGPU_NO_FEATURE;
    if (n_ids_out) n_ids_out[i] = have_wl ? n_ids : 0u;
    if (capture_out) capture_out[i] = have_wl ? wl_cap : have_pm ? pm_cap : FX_NO_CAPTURE;
    return true;
}

__global__ __launch_bounds__(256) void k_extract_features(const FxView v, const double *__restrict__ pedit, const FxRows r1, const FxRows r2,
                                                          uint64_t n, uint32_t *__restrict__ feature_out, uint32_t *__restrict__ n_ids_out,
                                                          uint32_t *__restrict__ capture_out, uint32_t *__restrict__ n_queued,
                                                          uint64_t *__restrict__ queue, uint32_t queue_cap) {
    FxEntry local[FX_LOCAL_ENTRIES];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!fx_match_read(v, pedit, r1, r2, i, local, FX_LOCAL_ENTRIES, feature_out, n_ids_out, capture_out)) {
            const uint32_t slot = atomicAdd(n_queued, 1u);
            if (slot < queue_cap) queue[slot] = i;
        }
    }
}

// the queued reads again, every read with a map row of row_entries entries in global memory
__global__ __launch_bounds__(256) void k_extract_features_queued(const FxView v, const double *__restrict__ pedit, const FxRows r1,
                                                                 const FxRows r2, const uint64_t *__restrict__ queue, uint32_t n_q,
                                                                 FxEntry *__restrict__ rows, uint32_t row_entries,
                                                                 uint32_t *__restrict__ feature_out, uint32_t *__restrict__ n_ids_out,
                                                                 uint32_t *__restrict__ capture_out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_q) return;
    (void)fx_match_read(v, pedit, r1, r2, queue[k], rows + (size_t)k * row_entries, row_entries, feature_out, n_ids_out, capture_out);
}

// ---- ONE tethered pattern, wave per 64 rows, feature table in LDS -------------------------------------------------------
// The common feature reference holds one pattern for all its features (Antibody Capture "5PNNNNNNNNNN(BC)", CRISPR
// "(BC)GTTTAAGAGCTAAGCTGGAA").  k_extract_features gives such a read to one thread: 15 - 100 byte loads 96 bytes apart from
// the next lane's, a binary search in L2 per lookup and the 3L-candidate posterior inline (2.2 / 1.3 G reads/s,
// profiles/r02_feature_extract_throughput.txt).  Here:
//   * a wave copies the window of its 64 rows that the pattern can touch (anchored: prefix + capture + suffix; floating: the
//     whole row) into LDS with dword loads that sweep the rows in address order, and every lane then works on its row out
//     of LDS (row pitch odd: no bank conflicts);
//   * the features live in an LDS open-addressing set of 64-bit keys (sequences of up to 32 bases), as in feature.hip;
//   * with a single capture correct_feature_barcode's map holds every candidate feature at most once (candidates of one
//     capture are distinct sequences), so the map degenerates to a running sum and a first maximum in the reference's
//     order -- no 16-entry map, no overflow queue; captures that need the posterior (no exact hit, at most one N) are
//     queued per workgroup and corrected with every lane busy; only they touch the quality rows (in global memory).
// Same results as fx_match_read for an extractor with one tethered pattern; tests/test_gpu_feature_extract.py runs both.
#define FXT_EMPTY 0xFFFFFFFFFFFFFFFFull
struct FxtParams {
    uint32_t read, anchor5, anchor3, pre_len, suf_len, L, n_feat;
    uint32_t pre_dots, suf_dots;  // the prefix / suffix hold wildcards only: nothing to compare
    uint32_t win_lo, win_dw;      // first byte (multiple of 4) and dwords of the row window staged into LDS
    uint32_t lanes_per_row;       // power of two >= win_dw, <= 64
    uint32_t prefetch;            // the next batch's row loads are issued before this batch is worked on (quad_lanes 1 or 2)
    uint32_t quad_lanes;          // 16-byte loads: power of two >= ceil(win_dw / 4) lanes share a row (0: rows shorter than 16 bytes)
    uint32_t pitch;               // LDS dwords per row (odd)
    uint32_t slot_mask;
    uint32_t halves;              // 1: the sorted half-key tables of the features are staged in LDS (posterior by pigeonhole)
    uint32_t needle, needle_mask, needle_off;  // floating patterns: (bytes at start + needle_off) & mask == needle, or mask == 0
};
struct FxtPending {
    unsigned long long key;
    uint32_t i_lo, i_hi, start, npos;  // npos: 1 + position of the capture's N, 0 = none
};
__device__ __forceinline__ uint32_t fxt_hash(unsigned long long key, uint32_t mask) {
    return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & mask;
}
__device__ __forceinline__ int fxt_find(const unsigned long long *tk, const uint32_t *tv, uint32_t mask, unsigned long long key) {
    uint32_t s = fxt_hash(key, mask);
    for (;;) {
        const unsigned long long e = tk[s];
        if (e == FXT_EMPTY) return -1;
        if (e == key) return (int)tv[s];
        s = (s + 1u) & mask;
    }
}
__device__ __forceinline__ uint32_t fxt_byte(const uint32_t *row, uint32_t b) { return (row[b >> 2] >> (8u * (b & 3u))) & 0xFFu; }

template <uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void k_extract_tethered_lds(const FxView v, const FxtParams P, const double *__restrict__ pedit,
                                                                  const FxRows R, uint64_t n, uint32_t *__restrict__ feature_out,
                                                                  uint32_t *__restrict__ n_ids_out, uint32_t *__restrict__ capture_out,
                                                                  FxtPending *__restrict__ recs, unsigned long long *__restrict__ rec_count,
                                                                  uint64_t rec_cap, uint64_t n_recs_in) {
    // recs / rec_count (nullable) -- the two-pass flow of a Feature Barcoding library (MAKE_SHARD's exact-match counts first, the
    // posterior with the distribution afterwards: make_shard_metrics.rs:336-345, aligner.rs:463-518):
    //   * an extractor WITHOUT a distribution appends the captures that found no exact feature (at most one N) to recs -- the
    //     only reads whose answer a distribution can change;
    //   * n_recs_in > 0 (an extractor WITH a distribution, the same definitions, rows and outputs): only those captures are
    //     corrected; the rows are not read again (half of the pass's HBM traffic is the rows).
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_key[];
    const uint32_t slots = P.slot_mask + 1u;
    // LDS: [hash keys u64 x slots][feature keys u64 x n_feat (halves)][hash values u32 x slots][half tables 4 x n_feat u32 (halves)]
    //      [row windows][pending captures]
    unsigned long long *s_fkey = s_key + slots;
    const uint32_t nfh = P.halves ? P.n_feat : 0u;
    uint32_t *s_val = reinterpret_cast<uint32_t *>(s_fkey + nfh);
    uint32_t *s_ha_key = s_val + slots, *s_ha_f = s_ha_key + nfh, *s_hb_key = s_ha_f + nfh, *s_hb_f = s_hb_key + nfh;
    uint32_t *s_rows_all = s_hb_f + nfh;
    uint32_t *s_rows = s_rows_all + (threadIdx.x >> 6) * (64u * P.pitch);
    const uint32_t words_before_pend = slots + 4u * nfh + (THREADS / 64u) * 64u * P.pitch;
    FxtPending *s_pend = reinterpret_cast<FxtPending *>(s_val + words_before_pend + (words_before_pend & 1u));  // 8-byte aligned
    __shared__ uint32_t s_npend;
    __shared__ uint8_t s_pat[256];  // prefix then suffix of the pattern ('.' = any); longer ones take the generic kernel
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t L = P.L;
    const unsigned long long *fk = reinterpret_cast<const unsigned long long *>(v.key);  // the one pattern's slice starts at 0
    for (uint32_t t = tid; t < P.pre_len + P.suf_len; t += THREADS)
        s_pat[t] = (uint8_t)(t < P.pre_len ? v.chars[v.pat[0].pre_off + t] : v.chars[v.pat[0].suf_off + (t - P.pre_len)]);
    for (uint32_t t = tid; t < slots; t += THREADS) s_key[t] = FXT_EMPTY;
    if (tid == 0) s_npend = 0;
    __syncthreads();
    for (uint32_t f = tid; f < P.n_feat; f += THREADS) {
        const unsigned long long k = fk[f];
        uint32_t t = fxt_hash(k, P.slot_mask);
        while (atomicCAS(&s_key[t], FXT_EMPTY, k) != FXT_EMPTY) t = (t + 1u) & P.slot_mask;
        s_val[t] = f;
        if (P.halves) {
            s_fkey[f] = k;
            s_ha_key[f] = v.ha_key[f];
            s_ha_f[f] = v.ha_f[f];
            s_hb_key[f] = v.hb_key[f];
            s_hb_f[f] = v.hb_f[f];
        }
    }
    __syncthreads();
    const uint32_t need = P.pre_len + L + P.suf_len;
    const uint32_t tag = P.read << 30;
    const uint32_t hb = L >> 1;  // bases of the last half (the split of the host's half-key tables)
    const unsigned long long low_mask = hb ? (~0ull >> (64u - 2u * hb)) : 0ull;

    // The 1-mismatch posterior of one capture (correct_feature_barcode, :34-117; one capture: every candidate feature enters
    // the map once, so the map is a running sum and a first maximum in (position, A<C<G<T) order).
    // halves: a feature one substitution away agrees with the capture on its whole first half or on its whole last half, so
    // two searches in the sorted half-key tables (LDS) list every candidate -- instead of 3 L hash probes -- and the few
    // candidates are then put into the reference's order.  More than FXT_MAXC candidates (dense feature families): the probes.
    constexpr uint32_t FXT_MAXC = 8;
    __shared__ unsigned long long s_rec_base;
    auto posterior = [&](const FxtPending &e) {
        {
            const uint64_t i = ((uint64_t)e.i_hi << 32) | e.i_lo;
            const uint8_t *q = R.qual + i * R.stride + e.start;
            double sum = 0.0, mx = -1.0;
            int best = -1;
            bool done = false;
            if (P.halves) {
                uint32_t cf[FXT_MAXC], ck[FXT_MAXC];  // candidate feature, order key = position * 4 + base
                uint32_t nc = 0;
                bool overflow = false;
                const uint32_t nsh = e.npos ? 2u * (L - e.npos) : 0u;                    // bit position of the N's pair
                const unsigned long long nm = e.npos ? (1ull << nsh) : 0ull;
                auto consider = [&](uint32_t f) {
                    const unsigned long long d = e.key ^ s_fkey[f];
                    const unsigned long long y = ((d | (d >> 1)) & 0x5555555555555555ull) | nm;
                    if (__popcll(y) != 1) return;
                    const uint32_t sh = (uint32_t)__ffsll((long long)y) - 1u;             // even: the pair that differs
                    const uint32_t pos = L - 1u - (sh >> 1);
                    const uint32_t b = (uint32_t)(s_fkey[f] >> sh) & 3u;
                    if (nc < FXT_MAXC) {
                        cf[nc] = f;
                        ck[nc] = pos * 4u + b;
                        nc++;
                    } else {
                        overflow = true;
                    }
                };
                auto search = [&](const uint32_t *hk, const uint32_t *hf, uint32_t k32) {
                    uint32_t lo = 0, hi = P.n_feat;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (hk[mid] < k32) lo = mid + 1; else hi = mid;
                    }
                    for (; lo < P.n_feat && hk[lo] == k32; lo++) consider(hf[lo]);
                };
                // the half that holds the N (if any) cannot be the one that agrees
                if ((nm >> (2u * hb)) == 0ull) search(s_ha_key, s_ha_f, (uint32_t)(e.key >> (2u * hb)));
                if ((nm & low_mask) == 0ull && hb) search(s_hb_key, s_hb_f, (uint32_t)(e.key & low_mask));
                if (!overflow) {
                    done = true;
                    for (uint32_t a = 1; a < nc; a++) {  // insertion sort by (position, base): the reference's order
                        const uint32_t kf = cf[a], kk = ck[a];
                        uint32_t b = a;
                        while (b > 0 && ck[b - 1] > kk) {
                            cf[b] = cf[b - 1];
                            ck[b] = ck[b - 1];
                            b--;
                        }
                        cf[b] = kf;
                        ck[b] = kk;
                    }
                    for (uint32_t a = 0; a < nc; a++) {
                        uint32_t qv = (uint8_t)(q[ck[a] >> 2] - 33u);  // u8 arithmetic as in feature_extraction.rs:43
                        qv = qv < 33u ? qv : 33u;
                        const double like = v.dist[cf[a]] * pedit[qv];
                        sum += like;
                        if (like > mx) {
                            mx = like;
                            best = (int)cf[a];
                        }
                    }
                }
            }
            if (!done) {
                sum = 0.0;
                mx = -1.0;
                best = -1;
                for (uint32_t pos = 0; pos < L; pos++) {
                    if (e.npos && e.npos != pos + 1u) continue;  // the other positions keep the N: no candidate there is a feature
                    const uint32_t sh = 2u * (L - 1u - pos);
                    const uint32_t orig = (uint32_t)(e.key >> sh) & 3u;
                    for (uint32_t b = 0; b < 4; b++) {
                        if (!e.npos && b == orig) continue;
                        const int f = fxt_find(s_key, s_val, P.slot_mask, (e.key & ~(3ull << sh)) | ((unsigned long long)b << sh));
                        if (f < 0) continue;
                        uint32_t qv = (uint8_t)(q[pos] - 33u);
                        qv = qv < 33u ? qv : 33u;
                        const double like = v.dist[f] * pedit[qv];
                        sum += like;
                        if (like > mx) {
                            mx = like;
                            best = f;
                        }
                    }
                }
            }
            const bool hit = best >= 0 && (mx / sum) >= 0.975;  // FEATURE_CONF_THRESHOLD (:21-22,113)
            feature_out[i] = hit ? v.index[best] : CRGPU_NO_FEATURE;
            if (n_ids_out) n_ids_out[i] = hit ? 1u : 0u;
            if (capture_out) capture_out[i] = (hit ? 0x80000000u : 0u) | tag | (e.start << 8) | L;
        }
    };
    // pending captures [first, first + count), count <= THREADS; called at workgroup-uniform points
    auto drain = [&](uint32_t first, uint32_t count) {
        if (v.dist) {
            if (tid < count) posterior(s_pend[first + tid]);
            return;
        }
        // no distribution: nothing matches, the captures are kept for the pass that has one
        if (tid == 0) s_rec_base = count ? atomicAdd(rec_count, (unsigned long long)count) : 0ull;
        __syncthreads();
        if (tid < count) {
            const FxtPending e = s_pend[first + tid];
            const uint64_t i = ((uint64_t)e.i_hi << 32) | e.i_lo;
            feature_out[i] = CRGPU_NO_FEATURE;
            if (n_ids_out) n_ids_out[i] = 0u;
            if (capture_out) capture_out[i] = tag | (e.start << 8) | L;
            if (s_rec_base + tid < rec_cap) recs[s_rec_base + tid] = e;
        }
        __syncthreads();
    };
    if (n_recs_in) {  // second pass over the kept captures only
        for (uint64_t j = (uint64_t)blockIdx.x * THREADS + tid; j < n_recs_in; j += (uint64_t)gridDim.x * THREADS) posterior(recs[j]);
        return;
    }

    // 16-byte staging (P.quad_lanes != 0): one or two loads per row.  With at most two, the NEXT batch's loads are issued before
    // this batch is worked on and land in registers meanwhile (the kernel waited for its row loads half of the time)
    const uint32_t q_rows_per = P.quad_lanes ? 64u / P.quad_lanes : 64u, q_col = P.quad_lanes ? lane & (P.quad_lanes - 1u) : 0u,
                   q_sub = P.quad_lanes ? lane / P.quad_lanes : 0u;
    const uint32_t q_stride_dw = R.stride >> 2, q_lo_dw = P.win_lo >> 2, q_quads = (P.win_dw + 3u) >> 2;
    const uint32_t q_qc = q_col < q_quads ? q_col : q_quads - 1u;
    const uint32_t q_d0 = q_lo_dw + 4u * q_qc + 4u <= q_stride_dw ? q_lo_dw + 4u * q_qc : q_stride_dw - 4u;  // a quad that would run past its row starts earlier
    const uint32_t *__restrict__ q_src = reinterpret_cast<const uint32_t *>(R.seq);
    const bool prefetch = P.prefetch != 0u;
    U32x4 pf[2];
    auto q_issue = [&](uint64_t wrow_of, U32x4 *out) {
#pragma unroll
        for (uint32_t it = 0; it < 2; it++) {
            if (it >= P.quad_lanes) break;
            const uint64_t row = wrow_of + it * q_rows_per + q_sub;
            const uint64_t rc = row < n ? row : n - 1;  // clamped: no load behind a branch
            out[it] = *reinterpret_cast<const U32x4 *>(q_src + rc * q_stride_dw + q_d0);
        }
    };
    auto q_commit = [&](const U32x4 *in) {
#pragma unroll
        for (uint32_t it = 0; it < 2; it++) {
            if (it >= P.quad_lanes) break;
            const uint32_t r = it * q_rows_per + q_sub;
            if (q_col < q_quads) {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t c = q_d0 + k - q_lo_dw;  // (unsigned: a dword in front of the window wraps to a large value)
                    if (c < P.win_dw) s_rows[r * P.pitch + c] = in[it].w[k];
                }
            }
        }
    };
    if (prefetch && (uint64_t)blockIdx.x * THREADS < n) q_issue((uint64_t)blockIdx.x * THREADS + (tid & ~63u), pf);
    for (uint64_t base = (uint64_t)blockIdx.x * THREADS; base < n; base += (uint64_t)gridDim.x * THREADS) {
        const uint64_t wrow = base + (tid & ~63u);  // the wave's first row
        // ---- stage the window of 64 rows --------------------------------------------------------------------------
        __builtin_amdgcn_wave_barrier();
        if (prefetch) {
            q_commit(pf);
            const uint64_t next = base + (uint64_t)gridDim.x * THREADS;
            if (next < n) q_issue(next + (tid & ~63u), pf);
        } else if (P.quad_lanes) {
            for (uint32_t it = 0; it < P.quad_lanes; it++) {
                const uint32_t r = it * q_rows_per + q_sub;
                const uint64_t row = wrow + r;
                const uint64_t rc = row < n ? row : n - 1;
                const U32x4 q4 = *reinterpret_cast<const U32x4 *>(q_src + rc * q_stride_dw + q_d0);
                if (q_col < q_quads) {
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t c = q_d0 + k - q_lo_dw;
                        if (c < P.win_dw) s_rows[r * P.pitch + c] = q4.w[k];
                    }
                }
            }
        } else {
            const uint32_t rows_per = 64u / P.lanes_per_row, col = lane & (P.lanes_per_row - 1u), sub = lane / P.lanes_per_row;
            const uint32_t *__restrict__ src = reinterpret_cast<const uint32_t *>(R.seq);
            const uint64_t stride_dw = R.stride >> 2;
#pragma unroll 4
            for (uint32_t it = 0; it < P.lanes_per_row; it++) {
                const uint32_t r = it * rows_per + sub;
                const uint64_t row = wrow + r;
                const uint64_t rc = row < n ? row : n - 1;  // clamped: no load behind a branch
                const uint32_t cc = col < P.win_dw ? col : P.win_dw - 1u;
                const uint32_t w = src[rc * stride_dw + (P.win_lo >> 2) + cc];
                if (col < P.win_dw) s_rows[r * P.pitch + col] = w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint64_t i = base + tid;
        if (i < n) {
            const uint32_t *row = s_rows + lane * P.pitch;
            const uint32_t len = R.len ? min(R.len[i], R.stride) : R.stride;
            uint32_t found = FX_NO_CAPTURE;
            if (len >= need) {
                uint32_t s_lo = 0, s_hi = len - need;
                bool possible = true;
                if (P.anchor3) s_lo = s_hi;
                if (P.anchor5) {
                    possible = s_lo == 0;
                    s_hi = 0;
                }
                // prefix and suffix at start st (literals only; '.' matches anything): the pattern's characters sit in LDS
                auto verify = [&](uint32_t st) -> bool {
                    bool ok = true;
                    if (!P.pre_dots)
                        for (uint32_t k = 0; k < P.pre_len && ok; k++)
                            ok = s_pat[k] == (uint8_t)'.' || (uint32_t)s_pat[k] == fxt_byte(row, st + k - P.win_lo);
                    if (!P.suf_dots)
                        for (uint32_t k = 0; k < P.suf_len && ok; k++)
                            ok = s_pat[P.pre_len + k] == (uint8_t)'.' ||
                                 (uint32_t)s_pat[P.pre_len + k] == fxt_byte(row, st + P.pre_len + L + k - P.win_lo);
                    return ok;
                };
                if (possible && P.needle_mask) {
                    // Floating pattern: its first literal characters (up to four) as a masked 32-bit compare, FOUR starts per row
                    // dword -- the window slides over the row's dwords, the four byte alignments of (this dword, next dword) are
                    // tested at once, and only the starts that pass (left to right) get the full comparison.
                    const uint32_t p_lo = s_lo + P.needle_off - P.win_lo, p_hi = s_hi + P.needle_off - P.win_lo;
                    uint32_t w_lo = row[p_lo >> 2];
                    for (uint32_t d = p_lo >> 2; d <= (p_hi >> 2) && found == FX_NO_CAPTURE; d++) {
                        const uint32_t w_hi = row[d + 1];  // may lie behind the window: the full comparison decides
                        uint32_t m = (((w_lo ^ P.needle) & P.needle_mask) == 0u ? 1u : 0u) |
                                     (((((w_lo >> 8) | (w_hi << 24)) ^ P.needle) & P.needle_mask) == 0u ? 2u : 0u) |
                                     (((((w_lo >> 16) | (w_hi << 16)) ^ P.needle) & P.needle_mask) == 0u ? 4u : 0u) |
                                     (((((w_lo >> 24) | (w_hi << 8)) ^ P.needle) & P.needle_mask) == 0u ? 8u : 0u);
                        w_lo = w_hi;
                        while (m && found == FX_NO_CAPTURE) {
                            const uint32_t a = (uint32_t)__ffs((int)m) - 1u;
                            m &= m - 1u;
                            const uint32_t pp = 4u * d + a;
                            if (pp < p_lo || pp > p_hi) continue;
                            const uint32_t st = pp + P.win_lo - P.needle_off;
                            if (verify(st)) found = st + P.pre_len;
                        }
                    }
                } else if (possible) {
                    for (uint32_t st = s_lo; st <= s_hi && found == FX_NO_CAPTURE; st++)
                        if (verify(st)) found = st + P.pre_len;
                }
            }
            if (found == FX_NO_CAPTURE) {
                feature_out[i] = CRGPU_NO_FEATURE;
                if (n_ids_out) n_ids_out[i] = 0u;
                if (capture_out) capture_out[i] = FX_NO_CAPTURE;
            } else {
                unsigned long long key = 0;
                uint32_t n_bad = 0, npos = 0;
                for (uint32_t k = 0; k < L; k++) {
                    const uint32_t c = fx_code((uint8_t)fxt_byte(row, found + k - P.win_lo));
                    key = (key << 2) | (c & 3u);
                    if (c >> 2) {
                        n_bad++;
                        npos = k + 1u;
                    }
                }
                const int exact = n_bad ? -1 : fxt_find(s_key, s_val, P.slot_mask, key);
                if (exact >= 0) {
                    feature_out[i] = v.index[exact];  // find_closest's fast path (:452-457)
                    if (n_ids_out) n_ids_out[i] = 1u;
                    if (capture_out) capture_out[i] = 0x80000000u | tag | (found << 8) | L;
                } else if ((v.dist || recs) && n_bad <= 1u) {
                    const uint32_t slot = atomicAdd(&s_npend, 1u);  // < 2 * THREADS: fewer than THREADS left over + THREADS new
                    s_pend[slot] = FxtPending{key, (uint32_t)i, (uint32_t)(i >> 32), found, npos};
                } else {
                    feature_out[i] = CRGPU_NO_FEATURE;
                    if (n_ids_out) n_ids_out[i] = 0u;
                    if (capture_out) capture_out[i] = tag | (found << 8) | L;
                }
            }
        }
        __syncthreads();
        const uint32_t np = s_npend;
        if (np >= THREADS) {  // uniform
            drain(np - THREADS, THREADS);  // the newest ones: the older ones stay at the front
            __syncthreads();
            if (tid == 0) s_npend = np - THREADS;
        }
        __syncthreads();
    }
    drain(0u, s_npend);  // fewer than THREADS left
}

// ---- host: the patterns ------------------------------------------------------------------------------------------------
namespace {

struct Parsed {
    std::string left, right;  // around "(BC)", markers replaced ('^' / '$'), N still N
};

bool is_base_or_n(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N'; }

// compile_pattern (:307-343) up to the point where the reference hands the string to the regex crate
bool parse_pattern(const std::string &orig, Parsed &out) {
    std::string p = orig;
    if (!p.empty() && p[0] == '5') {  // ^5[Pp]?[-_]?
        size_t i = 1;
        if (i < p.size() && (p[i] == 'P' || p[i] == 'p')) i++;
        if (i < p.size() && (p[i] == '-' || p[i] == '_')) i++;
        p = "^" + p.substr(i);
    }
    for (size_t s = 0; s < p.size(); s++) {  // [-_]?3[Pp]?$ , leftmost
        size_t j = s;
        if (p[j] == '-' || p[j] == '_') j++;
        if (j >= p.size() || p[j] != '3') continue;
        j++;
        if (j < p.size() && (p[j] == 'P' || p[j] == 'p')) j++;
        if (j != p.size()) continue;
        p = p.substr(0, s) + "$";
        break;
    }
    size_t count = 0;
    for (size_t at = orig.find("(BC)"); at != std::string::npos; at = orig.find("(BC)", at + 4)) count++;
    const size_t bc = p.find("(BC)");
    if (count != 1 || bc == std::string::npos) return false;
    out.left = p.substr(0, bc);
    out.right = p.substr(bc + 4);
    const std::string rest = out.left + out.right;  // must read ^?[ACGTN]*$?
    size_t k = 0;
    if (k < rest.size() && rest[k] == '^') k++;
    while (k < rest.size() && is_base_or_n(rest[k])) k++;
    if (k < rest.size() && rest[k] == '$') k++;
    return k == rest.size();
}

std::string dots(std::string s) {
    for (char &c : s)
        if (c == 'N') c = '.';
    return s;
}

std::string tethered_regex(const Parsed &p, uint32_t L) {
    return dots(p.left) + "(.{" + std::to_string(L) + "," + std::to_string(L) + "})" + dots(p.right);
}

struct HostPattern {
    int read = 0;
    bool tethered = true, anchor5 = false, anchor3 = false, never = false;
    std::string prefix, suffix, regex;
    uint32_t L = 0;
    std::vector<std::pair<std::string, uint32_t>> feats;  // (sequence, FeatureDef::index) in definition order
};

bool pack_seq(const std::string &s, uint64_t *out) {
    uint64_t k = 0;
    for (char c : s) {
        uint64_t b;
        switch (c) {
            case 'A': b = 0; break;
            case 'C': b = 1; break;
            case 'G': b = 2; break;
            case 'T': b = 3; break;
            default: return false;
        }
        k = (k << 2) | b;
    }
    *out = k;
    return true;
}

}  // namespace

extern "C" int crgpu_compile_feature_pattern(const char *pattern, uint32_t length, char *regex_out, uint64_t cap) {
    if (!pattern || !regex_out || cap == 0) return CRGPU_EINVAL;
    Parsed p;
    if (std::strcmp(pattern, "(BC)") == 0 || !parse_pattern(pattern, p)) return CRGPU_EINVAL;
    const std::string r = tethered_regex(p, length);
    if (r.size() + 1 > cap) return CRGPU_ERANGE;
    std::memcpy(regex_out, r.c_str(), r.size() + 1);
    return CRGPU_OK;
}

static void fx_release(FeatureExtractorSet &X) {
    (void)hipFree(X.d_blob);
    X = FeatureExtractorSet();
}

void cr_feature_extractors_free(crgpu_ctx *ctx) {
    for (int k = 0; k < CRGPU_MAX_LIB; k++) fx_release(ctx->fx[k]);
    cr_drop_feature_pending(ctx);
}
void cr_drop_feature_pending(crgpu_ctx *ctx) {
    if (ctx->fxp.d_recs) cr_pool_free(ctx, ctx->fxp.d_recs);  // stream-ordered: the pool reuses the block only for later work
    ctx->fxp = FxPendingSet();
}

extern "C" int crgpu_set_feature_extractor(crgpu_ctx *ctx, int extractor, const crgpu_feature_def *defs, uint32_t n_defs,
                                           const double *feat_dist, uint32_t n_dist) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, extractor >= 0 && extractor < CRGPU_MAX_LIB, CRGPU_EINVAL, "extractor id %d out of range", extractor);
    CR_REQUIRE(ctx, defs && n_defs > 0, CRGPU_EINVAL, "no feature definitions");
    std::vector<HostPattern> pats;
    std::map<std::string, size_t> by_key;  // (read, regex_str) -> pattern, as FeatureExtractor::insert keys them
    for (uint32_t d = 0; d < n_defs; d++) {
        const crgpu_feature_def &fd = defs[d];
        CR_REQUIRE(ctx, fd.pattern && fd.sequence && fd.read <= 1, CRGPU_EINVAL, "feature definition %u is incomplete", d);
        const std::string seq = fd.sequence, pat = fd.pattern;
        CR_REQUIRE(ctx, !seq.empty() && std::all_of(seq.begin(), seq.end(), is_base_or_n), CRGPU_EINVAL,
                   "Invalid sequence: '%s'. The only allowed characters are A, C, G, T, and N.", fd.sequence);
        CR_REQUIRE(ctx, seq.size() <= 32, CRGPU_ERANGE, "feature barcode of %zu bases unsupported (<= 32)", seq.size());
        CR_REQUIRE(ctx, !feat_dist || fd.index < n_dist, CRGPU_EINVAL, "feature index %u outside the distribution", fd.index);
        HostPattern hp;
        hp.read = (int)fd.read;
        hp.L = (uint32_t)seq.size();
        std::string key;
        if (pat == "(BC)") {
            hp.tethered = false;
            key = std::to_string(fd.read) + "|bare|" + std::to_string(seq.size());
        } else {
            Parsed p;
            CR_REQUIRE(ctx, parse_pattern(pat, p), CRGPU_EINVAL,
                       "Invalid pattern: '%s'. The pattern must optionally start with '5P', optionally end with '3P', contain "
                       "exactly one instance of the string '(BC)' and otherwise contain only the characters A, C, G, T, and N.",
                       fd.pattern);
            hp.regex = tethered_regex(p, hp.L);
            hp.anchor5 = !p.left.empty() && p.left[0] == '^';
            hp.anchor3 = !p.right.empty() && p.right.back() == '$';
            // "(BC)^..." and "...$(BC)" pass the reference's validation and compile, but can never match
            hp.never = p.right.find('^') != std::string::npos || p.left.find('$') != std::string::npos;
            hp.prefix = dots(p.left.substr(hp.anchor5 ? 1 : 0));
            hp.suffix = dots(p.right.substr(0, p.right.size() - (hp.anchor3 ? 1 : 0)));
            key = std::to_string(fd.read) + "|" + hp.regex;
        }
        auto it = by_key.find(key);
        if (it == by_key.end()) {
            it = by_key.emplace(key, pats.size()).first;
            pats.push_back(hp);
        }
        HostPattern &P = pats[it->second];
        for (const auto &f : P.feats)
            CR_REQUIRE(ctx, f.first != seq, CRGPU_EINVAL,
                       "Found two feature definitions with the same read, pattern ('%s') and barcode sequence ('%s')", fd.pattern,
                       fd.sequence);
        P.feats.emplace_back(seq, fd.index);
    }
    // device image
    std::vector<FxPat> hp(pats.size());
    std::string chars;
    std::vector<uint64_t> key;
    std::vector<uint32_t> index, ha_key, ha_f, hb_key, hb_f;
    std::vector<double> dist;
    std::vector<std::string> regexes;
    uint32_t max_feat = 0;
    for (size_t p = 0; p < pats.size(); p++) {
        HostPattern &P = pats[p];
        if (!P.tethered) {  // compile_bare_patterns (:291-305), for crgpu_feature_extractor_regex
            std::string r = "(";
            for (const auto &f : P.feats)
                for (uint32_t i = 0; i < P.L; i++) {
                    if (r.size() > 1) r += '|';
                    std::string a = f.first;
                    a[i] = '.';
                    r += a;
                }
            P.regex = r + ")";
        }
        regexes.push_back(P.regex);
        struct E {
            uint64_t k;
            uint32_t idx;
        };
        std::vector<E> es;
        for (const auto &f : P.feats) {
            uint64_t k;
            CR_REQUIRE(ctx, pack_seq(f.first, &k), CRGPU_EINVAL, "feature sequence '%s': N in feature sequences is unsupported",
                       f.first.c_str());
            es.push_back({k, f.second});
        }
        std::sort(es.begin(), es.end(), [](const E &a, const E &b) { return a.k < b.k; });
        FxPat &D = hp[p];
        D.read = (uint32_t)P.read;
        D.tethered = P.tethered;
        D.anchor5 = P.anchor5;
        D.anchor3 = P.anchor3;
        D.never = P.never;
        D.pre_off = (uint32_t)chars.size();
        D.pre_len = (uint32_t)P.prefix.size();
        chars += P.prefix;
        D.suf_off = (uint32_t)chars.size();
        D.suf_len = (uint32_t)P.suffix.size();
        chars += P.suffix;
        D.L = P.L;
        D.feat_off = (uint32_t)key.size();
        D.n_feat = (uint32_t)es.size();
        D.least_index = 0xFFFFFFFFu;
        max_feat = std::max(max_feat, D.n_feat);
        const uint32_t h = P.L / 2;
        std::vector<std::pair<uint32_t, uint32_t>> a, b;
        for (uint32_t f = 0; f < es.size(); f++) {
            key.push_back(es[f].k);
            index.push_back(es[f].idx);
            dist.push_back(feat_dist ? feat_dist[es[f].idx] : 0.0);
            D.least_index = std::min(D.least_index, es[f].idx);
            a.emplace_back((uint32_t)(es[f].k >> (2 * h)), f);
            b.emplace_back((uint32_t)(h ? es[f].k & (~0ull >> (64 - 2 * h)) : 0ull), f);
        }
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        for (const auto &x : a) {
            ha_key.push_back(x.first);
            ha_f.push_back(x.second);
        }
        for (const auto &x : b) {
            hb_key.push_back(x.first);
            hb_f.push_back(x.second);
        }
    }
    // one allocation: patterns | keys | dist | index | ha_key | ha_f | hb_key | hb_f | chars
    const size_t nf = key.size();
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    FeatureExtractorSet X;
    X.off_pat = 0;
    X.off_key = up8(hp.size() * sizeof(FxPat));
    X.off_dist = X.off_key + nf * 8;
    X.off_index = X.off_dist + nf * 8;
    X.off_ha_key = X.off_index + up8(nf * 4);
    X.off_ha_f = X.off_ha_key + up8(nf * 4);
    X.off_hb_key = X.off_ha_f + up8(nf * 4);
    X.off_hb_f = X.off_hb_key + up8(nf * 4);
    X.off_chars = X.off_hb_f + up8(nf * 4);
    const size_t total = X.off_chars + up8(chars.size() + 1);
    std::vector<uint8_t> blob(total, 0);
    std::memcpy(blob.data() + X.off_pat, hp.data(), hp.size() * sizeof(FxPat));
    std::memcpy(blob.data() + X.off_key, key.data(), nf * 8);
    std::memcpy(blob.data() + X.off_dist, dist.data(), nf * 8);
    std::memcpy(blob.data() + X.off_index, index.data(), nf * 4);
    std::memcpy(blob.data() + X.off_ha_key, ha_key.data(), nf * 4);
    std::memcpy(blob.data() + X.off_ha_f, ha_f.data(), nf * 4);
    std::memcpy(blob.data() + X.off_hb_key, hb_key.data(), nf * 4);
    std::memcpy(blob.data() + X.off_hb_f, hb_f.data(), nf * 4);
    std::memcpy(blob.data() + X.off_chars, chars.data(), chars.size());
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    fx_release(ctx->fx[extractor]);
    CR_HIP(ctx, hipMalloc(&X.d_blob, total));
    CR_HIP(ctx, hipMemcpy(X.d_blob, blob.data(), total, hipMemcpyHostToDevice));
    X.n_pat = (uint32_t)hp.size();
    X.max_feat = max_feat;
    X.has_dist = feat_dist != nullptr;
    X.uses_read[0] = X.uses_read[1] = false;
    for (const auto &P : pats) X.uses_read[P.read] = true;
    X.regex = regexes;
    if (pats.size() == 1 && pats[0].tethered && !pats[0].never) {
        const HostPattern &P0 = pats[0];
        X.one_tethered = true;
        {
            std::vector<std::pair<std::string, uint32_t>> fs(P0.feats);
            std::sort(fs.begin(), fs.end());
            X.sig = std::to_string(P0.read) + "|" + P0.regex;
            for (const auto &f : fs) X.sig += "|" + f.first + ":" + std::to_string(f.second);
        }
        X.t_read = (uint32_t)P0.read;
        X.t_anchor5 = P0.anchor5;
        X.t_anchor3 = P0.anchor3;
        X.t_pre_len = (uint32_t)P0.prefix.size();
        X.t_suf_len = (uint32_t)P0.suffix.size();
        X.t_L = P0.L;
        X.t_n_feat = (uint32_t)P0.feats.size();
        X.t_pre_dots = P0.prefix.find_first_not_of('.') == std::string::npos;
        X.t_suf_dots = P0.suffix.find_first_not_of('.') == std::string::npos;
        const std::string &src = (!P0.suffix.empty() && P0.suffix[0] != '.') ? P0.suffix : P0.prefix;
        if (!src.empty() && src[0] != '.') {
            uint32_t len = 0;
            while (len < 4 && len < src.size() && src[len] != '.') {
                X.t_needle |= (uint32_t)(uint8_t)src[len] << (8 * len);
                len++;
            }
            X.t_needle_len = len;
            X.t_needle_off = (&src == &P0.suffix) ? (uint32_t)P0.prefix.size() + P0.L : 0u;
        }
    }
    X.set = true;
    ctx->fx[extractor] = X;
    return CRGPU_OK;
}

extern "C" int crgpu_feature_extractor_regex(crgpu_ctx *ctx, int extractor, uint32_t pattern, char *regex_out, uint64_t cap,
                                             uint32_t *n_patterns_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, extractor >= 0 && extractor < CRGPU_MAX_LIB && ctx->fx[extractor].set, CRGPU_ESTATE,
               "feature extractor %d not set", extractor);
    const FeatureExtractorSet &X = ctx->fx[extractor];
    if (n_patterns_out) *n_patterns_out = X.n_pat;
    if (!regex_out) return CRGPU_OK;
    CR_REQUIRE(ctx, pattern < X.n_pat, CRGPU_EINVAL, "pattern %u out of range", pattern);
    CR_REQUIRE(ctx, X.regex[pattern].size() + 1 <= cap, CRGPU_ERANGE, "regex buffer too small");
    std::memcpy(regex_out, X.regex[pattern].c_str(), X.regex[pattern].size() + 1);
    return CRGPU_OK;
}

extern "C" int crgpu_extract_features_dev(crgpu_ctx *ctx, int extractor, const uint8_t *d_r1_seq, const uint8_t *d_r1_qual,
                                          const uint32_t *d_r1_len, uint32_t r1_stride, const uint8_t *d_r2_seq,
                                          const uint8_t *d_r2_qual, const uint32_t *d_r2_len, uint32_t r2_stride, uint64_t n,
                                          uint32_t *d_feature_out, uint32_t *d_n_ids_out, uint32_t *d_capture_out) {
    if (!ctx) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    CR_REQUIRE(ctx, extractor >= 0 && extractor < CRGPU_MAX_LIB && ctx->fx[extractor].set, CRGPU_ESTATE,
               "feature extractor %d not set", extractor);
    if (n == 0) return CRGPU_OK;
    const FeatureExtractorSet &X = ctx->fx[extractor];
    CR_REQUIRE(ctx, d_feature_out, CRGPU_EINVAL, "crgpu_extract_features_dev: NULL output");
    CR_REQUIRE(ctx, !X.uses_read[0] || (d_r1_seq && d_r1_qual && r1_stride), CRGPU_EINVAL,
               "the extractor holds R1 patterns but no R1 rows were given");
    CR_REQUIRE(ctx, !X.uses_read[1] || (d_r2_seq && d_r2_qual && r2_stride), CRGPU_EINVAL,
               "the extractor holds R2 patterns but no R2 rows were given");
    CR_REQUIRE(ctx, r1_stride < (1u << 22) && r2_stride < (1u << 22), CRGPU_ERANGE, "rows longer than 4 Mi bases");
    // the captures a distribution-less pass over these very rows kept (FxPendingSet): taken over before the by-products are dropped
    const FeatureExtractorSet &X0 = ctx->fx[extractor];
    FxPendingSet resume;
    {
        const FxPendingSet &Q = ctx->fxp;
        const bool r2 = X0.t_read != 0;
        if (ctx->trust_buffers && Q.valid && X0.one_tethered && X0.has_dist && Q.sig == X0.sig && Q.n == n &&
            Q.d_seq == (r2 ? d_r2_seq : d_r1_seq) && Q.d_qual == (r2 ? d_r2_qual : d_r1_qual) && Q.d_len == (r2 ? d_r2_len : d_r1_len) &&
            Q.stride == (r2 ? r2_stride : r1_stride) && Q.d_feature_out == d_feature_out && Q.d_n_ids_out == d_n_ids_out &&
            Q.d_capture_out == d_capture_out && !getenv("CRGPU_FXT_NO_RESUME")) {
            resume = ctx->fxp;
            ctx->fxp = FxPendingSet();  // the block now belongs to this call
        }
    }
    struct ResumeGuard {
        crgpu_ctx *c;
        void *p;
        ~ResumeGuard() { cr_pool_free(c, p); }
    } resume_guard{ctx, resume.d_recs};
    // caller buffers are written (4 bytes per read each): the by-products of earlier calls that describe them are not trusted
    // any more (pass A's miss records describe the barcodes: they stay)
    cr_invalidate_range(ctx, d_feature_out, n * sizeof(uint32_t));
    cr_invalidate_range(ctx, d_n_ids_out, n * sizeof(uint32_t));
    cr_invalidate_range(ctx, d_capture_out, n * sizeof(uint32_t));
    double pe[34];
    for (int q = 0; q < 34; q++) pe[q] = std::pow(10.0, -(double)q / 10.0);  // host libm as in :45
    double *d_pe = (double *)(ctx->d_scalars + 128);
    uint32_t *d_nq = ctx->d_scalars + 16;
    CR_HIP(ctx, hipMemcpyAsync(d_pe, pe, sizeof(pe), hipMemcpyHostToDevice, ctx->stream));
    CR_HIP(ctx, hipMemsetAsync(d_nq, 0, sizeof(uint32_t), ctx->stream));
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pe is a stack buffer
    const uint8_t *base = (const uint8_t *)X.d_blob;
    FxView v{(const FxPat *)(base + X.off_pat), X.n_pat, (const char *)(base + X.off_chars), (const uint64_t *)(base + X.off_key),
             (const uint32_t *)(base + X.off_index), X.has_dist ? (const double *)(base + X.off_dist) : nullptr,
             (const uint32_t *)(base + X.off_ha_key), (const uint32_t *)(base + X.off_ha_f), (const uint32_t *)(base + X.off_hb_key),
             (const uint32_t *)(base + X.off_hb_f)};
    FxRows r1{d_r1_seq, d_r1_qual, d_r1_len, r1_stride}, r2{d_r2_seq, d_r2_qual, d_r2_len, r2_stride};
    // ---- one tethered pattern: wave-per-rows kernel with the table in LDS (k_extract_tethered_lds) ----------------------
    if (X.one_tethered && X.t_n_feat <= 4096u && !getenv("CRGPU_FEATURES_GLOBAL")) {
        const FxRows &R = X.t_read ? r2 : r1;
        const uint32_t need = X.t_pre_len + X.t_L + X.t_suf_len;
        const bool aligned = R.stride % 4 == 0 && (uintptr_t)R.seq % 4 == 0;
        uint32_t win_lo = 0, win_hi = R.stride;           // floating pattern (or per-row lengths with '$'): the whole row
        if (X.t_anchor5) {
            win_hi = std::min(R.stride, (need + 3u) & ~3u);
            // a prefix of wildcards only ("^N{10}(BC)") is never looked at: the window starts at the capture
            if (X.t_pre_dots && !getenv("CRGPU_FXT_DWORD_LOADS")) win_lo = X.t_pre_len & ~3u;
        }
        else if (X.t_anchor3 && !R.len && need <= R.stride) win_lo = (R.stride - need) & ~3u;
        const uint32_t win_dw = (win_hi - win_lo) / 4u;
        if (aligned && win_dw >= 1 && win_dw <= 64u && R.stride >= 4 && X.t_pre_len + X.t_suf_len <= 256u) {
            FxtParams P{};
            P.read = X.t_read;
            P.anchor5 = X.t_anchor5;
            P.anchor3 = X.t_anchor3;
            P.pre_len = X.t_pre_len;
            P.suf_len = X.t_suf_len;
            P.L = X.t_L;
            P.n_feat = X.t_n_feat;
            P.pre_dots = X.t_pre_dots;
            P.suf_dots = X.t_suf_dots;
            P.win_lo = win_lo;
            P.win_dw = win_dw;
            P.lanes_per_row = 1;
            while (P.lanes_per_row < win_dw) P.lanes_per_row <<= 1;
            P.pitch = win_dw | 1u;
            P.quad_lanes = 0;
            if (R.stride >= 16 && !getenv("CRGPU_FXT_DWORD_LOADS")) {  // (A/B switch: one dword per lane and load, as before)
                P.quad_lanes = 1;
                while (P.quad_lanes < (win_dw + 3u) / 4u) P.quad_lanes <<= 1;
            }
            P.prefetch = (P.quad_lanes == 1u || P.quad_lanes == 2u) && !getenv("CRGPU_FXT_NO_PREFETCH");  // (A/B switch)
            uint32_t slots = 64;
            while (slots < 2u * X.t_n_feat) slots <<= 1;  // load factor <= 0.5
            P.slot_mask = slots - 1u;
            if (X.t_needle_len && !X.t_anchor5 && !X.t_anchor3) {  // anchored patterns have one start: nothing to skip
                P.needle = X.t_needle;
                P.needle_mask = X.t_needle_len >= 4 ? 0xFFFFFFFFu : ((1u << (8u * X.t_needle_len)) - 1u);
                P.needle_off = X.t_needle_off;
            }
            P.halves = X.has_dist && X.t_n_feat <= 1024u && X.t_L >= 2 && !getenv("CRGPU_FXT_PROBES");
            const size_t half_bytes = P.halves ? (size_t)X.t_n_feat * 24 : 0;
            // first pass of the two-pass flow (no distribution, by-products allowed): keep the captures without exact feature
            FxtPending *d_recs = (FxtPending *)resume.d_recs;
            unsigned long long *d_rec_count = (unsigned long long *)(ctx->d_scalars + 66);
            uint64_t rec_cap = 0, n_recs_in = resume.valid ? resume.n_recs : 0;
            void *new_recs = nullptr;
            const bool record = ctx->trust_buffers && !X.has_dist && !resume.valid && n >= 4096 && !getenv("CRGPU_FXT_NO_RESUME");
            if (record) {
                rec_cap = n / 2 + 1024;  // a third of the captures of a real library miss at most; more: the second pass reads the rows
                if (cr_pool_alloc(ctx, &new_recs, rec_cap * sizeof(FxtPending)) == CRGPU_OK) {
                    d_recs = (FxtPending *)new_recs;
                    CR_HIP(ctx, hipMemsetAsync(d_rec_count, 0, sizeof(unsigned long long), ctx->stream));
                } else {
                    rec_cap = 0;
                    (void)hipGetLastError();
                }
            }
            const uint64_t work = n_recs_in ? n_recs_in : n;
            CrTimer t(ctx, CRGPU_T_FEATURE, work);
            if (X.t_n_feat <= 1024u) {
                const size_t lds = (size_t)slots * 12 + half_bytes + (size_t)4 * 64 * P.pitch * 4 + 8 + 2 * 256 * sizeof(FxtPending);
                cr_allow_lds(ctx, (const void *)k_extract_tethered_lds<256>, lds);
                hipLaunchKernelGGL(k_extract_tethered_lds<256>, dim3(cr_grid(work, 256, 256u * 4u)), dim3(256), lds, ctx->stream, v, P,
                                   d_pe, R, n, d_feature_out, d_n_ids_out, d_capture_out, rec_cap || n_recs_in ? d_recs : nullptr, d_rec_count,
                                   rec_cap, n_recs_in);
            } else {
                const size_t lds = (size_t)slots * 12 + (size_t)16 * 64 * P.pitch * 4 + 8 + 2 * 1024 * sizeof(FxtPending);
                cr_allow_lds(ctx, (const void *)k_extract_tethered_lds<1024>, lds);
                hipLaunchKernelGGL(k_extract_tethered_lds<1024>, dim3(cr_grid(work, 1024, 256u)), dim3(1024), lds, ctx->stream, v, P, d_pe,
                                   R, n, d_feature_out, d_n_ids_out, d_capture_out, rec_cap || n_recs_in ? d_recs : nullptr, d_rec_count,
                                   rec_cap, n_recs_in);
            }
            CR_HIP(ctx, hipGetLastError());
            ctx->feature_fast_launches++;
            if (n_recs_in) ctx->feature_resumed_reads += n_recs_in;
            if (rec_cap) {
                unsigned long long cnt = 0;
                if (crgpu_memcpy_d2h(ctx, &cnt, d_rec_count, sizeof(cnt)) == CRGPU_OK && cnt <= rec_cap) {
                    FxPendingSet &Q = ctx->fxp;
                    Q.valid = true;
                    Q.d_seq = R.seq;
                    Q.d_qual = R.qual;
                    Q.d_len = R.len;
                    Q.n = n;
                    Q.n_recs = cnt;
                    Q.stride = R.stride;
                    Q.d_feature_out = d_feature_out;
                    Q.d_n_ids_out = d_n_ids_out;
                    Q.d_capture_out = d_capture_out;
                    Q.sig = X.sig;
                    Q.d_recs = new_recs;
                } else {
                    cr_pool_free(ctx, new_recs);  // more captures than room: the pass with the distribution reads the rows
                }
            }
            return CRGPU_OK;
        }
    }
    // the queue of reads whose map outgrew the local array: sized so that their global map rows stay below 256 MB;
    // more than that many are handled by further rounds over what is left
    const uint32_t row_entries = std::max(X.max_feat, 1u);
    const uint64_t by_rows = std::max<uint64_t>(1, (256ull << 20) / ((uint64_t)row_entries * sizeof(FxEntry)));
    const uint32_t queue_cap = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(n, by_rows), 1u << 20);
    void *queue_p = nullptr;
    CR_TRY(cr_pool_alloc(ctx, &queue_p, (uint64_t)queue_cap * sizeof(uint64_t)));
    struct Release {
        crgpu_ctx *c;
        void *p;
        ~Release() { cr_pool_free(c, p); }
    } rel_q{ctx, queue_p};
    uint64_t *d_queue = (uint64_t *)queue_p;
    CrTimer t(ctx, CRGPU_T_FEATURE, n);
    hipLaunchKernelGGL(k_extract_features, dim3(cr_grid(n, 256)), dim3(256), 0, ctx->stream, v, d_pe, r1, r2, n, d_feature_out,
                       d_n_ids_out, d_capture_out, d_nq, d_queue, queue_cap);
    CR_HIP(ctx, hipGetLastError());
    uint32_t nq = 0;
    CR_HIP(ctx, hipMemcpyAsync(&nq, d_nq, sizeof(nq), hipMemcpyDeviceToHost, ctx->stream));
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nq == 0) return CRGPU_OK;
    ctx->feature_reads_requeued += nq;
    CR_REQUIRE(ctx, nq <= queue_cap, CRGPU_ERANGE,
               "%u reads need the wide correction map but only %u fit in one call; split the batch", nq, queue_cap);
    void *rows_p = nullptr;
    CR_TRY(cr_pool_alloc(ctx, &rows_p, (uint64_t)nq * row_entries * sizeof(FxEntry)));
    Release rel_r{ctx, rows_p};
    hipLaunchKernelGGL(k_extract_features_queued, dim3((nq + 255u) / 256u), dim3(256), 0, ctx->stream, v, d_pe, r1, r2, d_queue, nq,
                       (FxEntry *)rows_p, row_entries, d_feature_out, d_n_ids_out, d_capture_out);
    CR_HIP(ctx, hipGetLastError());
    CR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the rows go back to the pool
    return CRGPU_OK;
}

// ---- the prior: MAKE_SHARD's exact-match feature counts and compute_feature_dist ------------------------------------------
// Few distinct features (hundreds of antibodies) carry all the reads, wherever their indices lie in a feature reference of
// tens of thousands of entries: a workgroup counts into an LDS hash table of (feature, count) pairs and flushes it once; a
// feature that finds no slot (more than FC_SLOTS * 3 / 4 distinct ones in one workgroup) is counted in global memory directly.
// (62.5 M global atomics on 200 addresses took 130 ms.)
#define FC_SLOTS 4096u
#define FC_EMPTY 0xFFFFFFFFu
__global__ __launch_bounds__(256) void k_feature_counts(const uint32_t *__restrict__ feature, uint64_t n, uint32_t n_features,
                                                        unsigned long long *__restrict__ counts) {
    __shared__ uint32_t s_f[FC_SLOTS], s_c[FC_SLOTS];
    __shared__ uint32_t s_used;
    for (uint32_t t = threadIdx.x; t < FC_SLOTS; t += 256) {
        s_f[t] = FC_EMPTY;
        s_c[t] = 0;
    }
    if (threadIdx.x == 0) s_used = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t f = feature[i];
        if (f >= n_features) continue;
        uint32_t t = (f * 0x9E3779B1u) >> 20;  // 12 bits
        bool placed = false;
        for (uint32_t probe = 0; probe < 16 && !placed; probe++, t = (t + 1u) & (FC_SLOTS - 1u)) {
            uint32_t cur = s_f[t];
            if (cur == FC_EMPTY && s_used < FC_SLOTS * 3u / 4u) {
                cur = atomicCAS(&s_f[t], FC_EMPTY, f);
                if (cur == FC_EMPTY) {
                    atomicAdd(&s_used, 1u);
                    cur = f;
                }
            }
            if (cur == f) {
                atomicAdd(&s_c[t], 1u);
                placed = true;
            }
        }
        if (!placed) atomicAdd(&counts[f], 1ull);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < FC_SLOTS; t += 256)
        if (s_f[t] != FC_EMPTY && s_c[t]) atomicAdd(&counts[s_f[t]], (unsigned long long)s_c[t]);
}

extern "C" int crgpu_feature_counts_dev(crgpu_ctx *ctx, const uint32_t *d_feature, uint64_t n, uint32_t n_features,
                                        int64_t *counts_inout) {
    if (!ctx || !counts_inout || !n_features) return CRGPU_EINVAL;
    CR_ENTER(ctx);
    if (n == 0) return CRGPU_OK;
    CR_REQUIRE(ctx, d_feature, CRGPU_EINVAL, "crgpu_feature_counts_dev: NULL features");
    void *d_c = nullptr;
    CR_TRY(cr_pool_alloc(ctx, &d_c, (uint64_t)n_features * 8));
    struct Rel {
        crgpu_ctx *c;
        void *p;
        ~Rel() { cr_pool_free(c, p); }
    } rel{ctx, d_c};
    CR_HIP(ctx, hipMemsetAsync(d_c, 0, (size_t)n_features * 8, ctx->stream));
    {
        CrTimer t(ctx, CRGPU_T_FEATURE, n);
        hipLaunchKernelGGL(k_feature_counts, dim3(cr_grid(n, 256 * 8, 256u * 4u)), dim3(256), 0, ctx->stream, d_feature, n, n_features,
                           (unsigned long long *)d_c);
        CR_HIP(ctx, hipGetLastError());
    }
    std::vector<unsigned long long> h(n_features);
    CR_TRY(crgpu_memcpy_d2h(ctx, h.data(), d_c, (uint64_t)n_features * 8));
    for (uint32_t f = 0; f < n_features; f++) counts_inout[f] += (int64_t)h[f];
    return CRGPU_OK;
}

// compute_feature_dist (cr_types/src/reference/feature_checker.rs:8-50)
extern "C" int crgpu_compute_feature_dist(const int64_t *counts, const uint32_t *feature_type, uint32_t n_features, double *dist_out) {
    if (!counts || !dist_out) return CRGPU_EINVAL;
    std::map<uint32_t, int64_t> sums;
    for (uint32_t i = 0; i < n_features; i++) sums[feature_type ? feature_type[i] : 0u] += counts[i];
    bool all_zero = true;
    for (uint32_t i = 0; i < n_features; i++) {
        const int64_t sum = sums[feature_type ? feature_type[i] : 0u];
        dist_out[i] = sum > 0 ? (double)counts[i] / (double)sum : 0.0;
        if (dist_out[i] != 0.0) all_zero = false;
    }
    if (all_zero)
        for (uint32_t i = 0; i < n_features; i++) dist_out[i] = 1.0 / (double)n_features;
    return CRGPU_OK;
}
