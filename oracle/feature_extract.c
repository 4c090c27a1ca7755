/*
 * feature_extract.c -- oracle restatement of FeatureExtractor (pattern compilation, match_read with tethered
 * and bare "(BC)" patterns, multi-capture correct_feature_barcode).  TEST INFRASTRUCTURE ONLY (see cr_oracle.h).
 *
 * Follows:
 *   cr_types/src/reference/feature_extraction.rs:176-262  FeatureExtractor::new (grouping of the definitions)
 *   cr_types/src/reference/feature_extraction.rs:291-305  compile_bare_patterns
 *   cr_types/src/reference/feature_extraction.rs:307-343  compile_pattern
 *   cr_types/src/reference/feature_extraction.rs:345-356  validate_sequence
 *   cr_types/src/reference/feature_extraction.rs:358-441  match_read
 *   cr_types/src/reference/feature_extraction.rs:443-470  find_closest
 *   cr_types/src/reference/feature_extraction.rs:34-117   correct_feature_barcode (any number of captures)
 *
 * The reference compiles every pattern to a regular expression of a small grammar ('^', '$', literals, '.', one
 * "(.{L,L})" group, or for bare patterns one "(a|b|...)" group of alternatives).  The oracle keeps the expression as a
 * STRING, exactly as the reference builds it, and interprets that string with the leftmost-first matcher below, so
 * that the product's structured (prefix / capture / suffix) representation is checked against something independent.
 * The reference's containers are HashMaps; wherever their iteration order could matter the result is shown to be order
 * free in the comments, and the oracle walks the patterns in creation order.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cr_oracle.h"

#define FEATURE_CONF_THRESHOLD 0.975
#define FEATURE_MAX_QV 33
#define ILLUMINA_QUAL_OFFSET 33

extern double pow(double, double);

typedef struct {
    char *regex;     /* regex_str */
    int read;        /* WhichRead: 0 R1, 1 R2 */
    int tethered;    /* PatternType */
    uint32_t len;    /* length of the captured barcode */
    uint32_t n_feat; /* FeaturePattern::features */
    char **seq;
    uint32_t *index;
} ox_pattern;

struct oracle_extractor {
    ox_pattern *pat;
    uint32_t n_pat;
    double *dist;
    uint32_t n_dist;
};

static char *dup_str(const char *s) {
    size_t n = strlen(s) + 1;
    char *r = (char *)malloc(n);
    memcpy(r, s, n);
    return r;
}

/* ---- the replacements compile_pattern does with its helper expressions ------------------------------------------ */
static int is_base_or_n(char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N'; }

/* compile_pattern (:307-343).  Returns 0 and the regex in out, or -1 (invalid pattern). */
int oracle_compile_feature_pattern(const char *orig_pat, uint32_t length, char *out, size_t out_cap) {
    char pat[512], tmp[512];
    size_t n = strlen(orig_pat);
    if (n + 8 >= sizeof(pat)) return -1;
    /* "^5[Pp]?[-_]?" -> "^" */
    size_t i = 0;
    if (orig_pat[0] == '5') {
        i = 1;
        if (orig_pat[i] == 'P' || orig_pat[i] == 'p') i++;
        if (orig_pat[i] == '-' || orig_pat[i] == '_') i++;
        snprintf(pat, sizeof(pat), "^%s", orig_pat + i);
    } else {
        snprintf(pat, sizeof(pat), "%s", orig_pat);
    }
    /* "[-_]?3[Pp]?$" -> "$": the leftmost start from which the expression reaches the end of the string */
    n = strlen(pat);
    for (size_t s = 0; s < n; s++) {
        size_t j = s;
        if (pat[j] == '-' || pat[j] == '_') j++;
        if (pat[j] != '3') continue;
        j++;
        if (pat[j] == 'P' || pat[j] == 'p') j++;
        if (j != n) continue;
        pat[s] = '$';
        pat[s + 1] = 0;
        break;
    }
    /* exactly one "(BC)" in the ORIGINAL pattern; the rest must read ^?[ACGTN]*$? */
    int count = 0;
    for (const char *p = orig_pat; (p = strstr(p, "(BC)")) != NULL; p += 4) count++;
    const char *bc = strstr(pat, "(BC)");
    if (bc) {
        size_t k = (size_t)(bc - pat);
        memcpy(tmp, pat, k);
        strcpy(tmp + k, bc + 4);
    } else {
        strcpy(tmp, pat);
    }
    const char *c = tmp;
    if (*c == '^') c++;
    while (is_base_or_n(*c)) c++;
    if (*c == '$') c++;
    if (*c != 0 || count != 1) return -1;
    /* N -> '.', (BC) -> (.{length,length}) */
    size_t o = 0;
    for (const char *p = pat; *p;) {
        char piece[48];
        if (p == bc) {
            snprintf(piece, sizeof(piece), "(.{%u,%u})", length, length);
            p += 4;
        } else {
            piece[0] = *p == 'N' ? '.' : *p;
            piece[1] = 0;
            p++;
        }
        size_t l = strlen(piece);
        if (o + l + 1 > out_cap) return -1;
        memcpy(out + o, piece, l);
        o += l;
    }
    out[o] = 0;
    return 0;
}

/* compile_bare_patterns (:291-305): "(" + every sequence with one position replaced by '.', joined by '|' + ")" */
static char *compile_bare(char **seqs, uint32_t n, uint32_t len) {
    size_t cap = (size_t)n * len * (len + 1) + 3;
    char *r = (char *)malloc(cap), *w = r;
    *w++ = '(';
    for (uint32_t f = 0; f < n; f++)
        for (uint32_t i = 0; i < len; i++) {
            if (w != r + 1) *w++ = '|';
            memcpy(w, seqs[f], len);
            w[i] = '.';
            w += len;
        }
    *w++ = ')';
    *w = 0;
    return r;
}

int oracle_compile_bare_patterns(const char *const *seqs, uint32_t n, char *out, size_t out_cap) {
    uint32_t len = (uint32_t)strlen(seqs[0]);
    char *r = compile_bare((char **)seqs, n, len);
    int ok = strlen(r) + 1 <= out_cap;
    if (ok) strcpy(out, r);
    free(r);
    return ok ? 0 : -1;
}

/* ---- a matcher for exactly the expressions above --------------------------------------------------------------------
 * match_here: does the expression re match text[pos..] (text_len bytes in all, starting at pos)?  On success the
 * capture group's span is stored.  Leftmost-first: alternatives are tried in the order written. */
static int match_seq(const char *re, const char *re_end, const char *text, uint32_t text_len, uint32_t pos,
                     uint32_t *cap_s, uint32_t *cap_e);

static int match_group(const char *re, const char *re_end, const char *text, uint32_t text_len, uint32_t pos,
                       uint32_t *cap_s, uint32_t *cap_e) {
    /* re points behind '(' */
    const char *close = memchr(re, ')', (size_t)(re_end - re));
    if (re[0] == '.' && re[1] == '{') { /* (.{L,L}) */
        uint32_t L = (uint32_t)strtoul(re + 2, NULL, 10);
        if (pos + L > text_len) return 0;
        uint32_t s = pos, e = pos + L;
        if (!match_seq(close + 1, re_end, text, text_len, e, cap_s, cap_e)) return 0;
        *cap_s = s;
        *cap_e = e;
        return 1;
    }
    for (const char *alt = re; alt < close;) {
        const char *bar = memchr(alt, '|', (size_t)(close - alt));
        const char *alt_end = bar ? bar : close;
        uint32_t L = (uint32_t)(alt_end - alt), k = 0;
        if (pos + L <= text_len) {
            while (k < L && (alt[k] == '.' || alt[k] == text[pos + k])) k++;
            if (k == L && match_seq(close + 1, re_end, text, text_len, pos + L, cap_s, cap_e)) {
                *cap_s = pos;
                *cap_e = pos + L;
                return 1;
            }
        }
        alt = alt_end + 1;
    }
    return 0;
}

static int match_seq(const char *re, const char *re_end, const char *text, uint32_t text_len, uint32_t pos,
                     uint32_t *cap_s, uint32_t *cap_e) {
    while (re < re_end) {
        if (*re == '(') return match_group(re + 1, re_end, text, text_len, pos, cap_s, cap_e);
        if (*re == '$') {
            if (pos != text_len) return 0;
            re++;
            continue;
        }
        if (pos >= text_len) return 0;
        if (*re != '.' && *re != text[pos]) return 0;
        re++;
        pos++;
    }
    return 1;
}

/* Regex::captures on text: leftmost match; returns 1 and the group's span (relative to text) */
static int regex_captures(const char *regex, const char *text, uint32_t text_len, uint32_t *cap_s, uint32_t *cap_e) {
    const char *re = regex, *re_end = regex + strlen(regex);
    if (*re == '^') return match_seq(re + 1, re_end, text, text_len, 0, cap_s, cap_e);
    for (uint32_t start = 0; start <= text_len; start++)
        if (match_seq(re, re_end, text, text_len, start, cap_s, cap_e)) return 1;
    return 0;
}

/* ---- FeatureExtractor::new ---------------------------------------------------------------------------------------- */
static void set_err(char *err, size_t cap, const char *msg) {
    if (err && cap) snprintf(err, cap, "%s", msg);
}

void oracle_extractor_free(oracle_extractor *x) {
    if (!x) return;
    for (uint32_t p = 0; p < x->n_pat; p++) {
        for (uint32_t f = 0; f < x->pat[p].n_feat; f++) free(x->pat[p].seq[f]);
        free(x->pat[p].seq);
        free(x->pat[p].index);
        free(x->pat[p].regex);
    }
    free(x->pat);
    free(x->dist);
    free(x);
}

/* FeatureExtractor::insert + FeaturePattern::insert (:264-289, :151-166) */
static int add_feature(oracle_extractor *x, int read, const char *regex, int tethered, const char *seq, uint32_t index,
                       char *err, size_t err_cap) {
    ox_pattern *P = NULL;
    for (uint32_t p = 0; p < x->n_pat; p++)
        if (x->pat[p].read == read && strcmp(x->pat[p].regex, regex) == 0) P = &x->pat[p];
    if (!P) {
        x->pat = (ox_pattern *)realloc(x->pat, (x->n_pat + 1) * sizeof(ox_pattern));
        P = &x->pat[x->n_pat++];
        memset(P, 0, sizeof(*P));
        P->regex = dup_str(regex);
        P->read = read;
        P->tethered = tethered;
        P->len = (uint32_t)strlen(seq);
    }
    for (uint32_t f = 0; f < P->n_feat; f++)
        if (strcmp(P->seq[f], seq) == 0) {
            set_err(err, err_cap, "Found two feature definitions with the same read, pattern and barcode sequence");
            return -1;
        }
    P->seq = (char **)realloc(P->seq, (P->n_feat + 1) * sizeof(char *));
    P->index = (uint32_t *)realloc(P->index, (P->n_feat + 1) * sizeof(uint32_t));
    P->seq[P->n_feat] = dup_str(seq);
    P->index[P->n_feat] = index;
    P->n_feat++;
    return 0;
}

oracle_extractor *oracle_extractor_new(const oracle_feature_def *defs, uint32_t n_defs, const double *feat_dist,
                                       uint32_t n_dist, char *err, size_t err_cap) {
    oracle_extractor *x = (oracle_extractor *)calloc(1, sizeof(*x));
    if (feat_dist) {
        x->dist = (double *)malloc(n_dist * sizeof(double));
        memcpy(x->dist, feat_dist, n_dist * sizeof(double));
        x->n_dist = n_dist;
    }
    char regex[1024];
    for (uint32_t d = 0; d < n_defs; d++) {
        const char *s = defs[d].sequence;
        size_t sl = strlen(s);
        int ok = sl > 0; /* validate_sequence: ^[ACGTN]+$ */
        for (size_t i = 0; i < sl; i++) ok = ok && is_base_or_n(s[i]);
        if (!ok) {
            set_err(err, err_cap, "Invalid sequence. The only allowed characters are A, C, G, T, and N.");
            oracle_extractor_free(x);
            return NULL;
        }
        if (strcmp(defs[d].pattern, "(BC)") == 0) continue; /* bundled below */
        if (oracle_compile_feature_pattern(defs[d].pattern, (uint32_t)sl, regex, sizeof(regex)) != 0) {
            set_err(err, err_cap, "Invalid pattern");
            oracle_extractor_free(x);
            return NULL;
        }
        if (add_feature(x, (int)defs[d].read, regex, 1, s, defs[d].index, err, err_cap) != 0) {
            oracle_extractor_free(x);
            return NULL;
        }
    }
    /* bare patterns: one expression per (read, sequence length) (:196-203, :222-236) */
    for (uint32_t d = 0; d < n_defs; d++) {
        if (strcmp(defs[d].pattern, "(BC)") != 0) continue;
        size_t sl = strlen(defs[d].sequence);
        int first = 1;
        for (uint32_t e = 0; e < d; e++)
            if (strcmp(defs[e].pattern, "(BC)") == 0 && defs[e].read == defs[d].read && strlen(defs[e].sequence) == sl)
                first = 0;
        if (!first) continue;
        uint32_t n = 0;
        char **seqs = (char **)malloc(n_defs * sizeof(char *));
        for (uint32_t e = d; e < n_defs; e++)
            if (strcmp(defs[e].pattern, "(BC)") == 0 && defs[e].read == defs[d].read && strlen(defs[e].sequence) == sl)
                seqs[n++] = (char *)defs[e].sequence;
        char *bare = compile_bare(seqs, n, (uint32_t)sl);
        int rc = 0;
        for (uint32_t e = d; e < n_defs && rc == 0; e++)
            if (strcmp(defs[e].pattern, "(BC)") == 0 && defs[e].read == defs[d].read && strlen(defs[e].sequence) == sl)
                rc = add_feature(x, (int)defs[e].read, bare, 0, defs[e].sequence, defs[e].index, err, err_cap);
        free(bare);
        free(seqs);
        if (rc != 0) {
            oracle_extractor_free(x);
            return NULL;
        }
    }
    return x;
}

uint32_t oracle_extractor_n_patterns(const oracle_extractor *x) { return x->n_pat; }
const char *oracle_extractor_regex(const oracle_extractor *x, uint32_t p) { return x->pat[p].regex; }

/* ---- correct_feature_barcode over any number of captures (:34-117) ------------------------------------------------- */
typedef struct {
    uint32_t start, end; /* the capture in the read */
} ox_capture;

typedef struct {
    int feat;          /* position in the pattern's feature list == the whitelist sequence (the HashMap key) */
    double likelihood; /* the entry's likelihood */
    uint32_t cap;      /* the capture the entry remembers */
} ox_entry;

static int get_feature(const ox_pattern *P, const char *seq) {
    for (uint32_t f = 0; f < P->n_feat; f++)
        if (memcmp(P->seq[f], seq, P->len) == 0) return (int)f;
    return -1;
}

static int correct_captures(const oracle_extractor *x, const ox_pattern *P, const char *seq, const uint8_t *qual,
                            const ox_capture *caps, uint32_t n_caps, uint32_t *cap_out) {
    static const char NUCLEOTIDES[4] = {'A', 'C', 'G', 'T'};
    ox_entry *map = (ox_entry *)malloc(P->n_feat * sizeof(ox_entry));
    uint32_t n_map = 0;
    double likelihood_sum = 0.0;
    char test_seq[256];
#define INSERT_HIT(LIKE, FEAT, CAP)                                                                  \
    do {                                                                                             \
        uint32_t e_ = 0;                                                                             \
        while (e_ < n_map && map[e_].feat != (FEAT)) e_++;                                           \
        if (e_ < n_map) {                                                                            \
            if ((LIKE) > map[e_].likelihood) { /* replace the old hit (:64-70) */                    \
                double old_ = map[e_].likelihood;                                                    \
                map[e_].likelihood = (LIKE);                                                         \
                map[e_].cap = (CAP);                                                                 \
                likelihood_sum += (LIKE) - old_;                                                     \
            }                                                                                        \
        } else {                                                                                     \
            map[n_map].feat = (FEAT);                                                                \
            map[n_map].likelihood = (LIKE);                                                          \
            map[n_map].cap = (CAP);                                                                  \
            n_map++;                                                                                 \
            likelihood_sum += (LIKE);                                                                \
        }                                                                                            \
    } while (0)
    for (uint32_t c = 0; c < n_caps; c++) {
        const char *s = seq + caps[c].start;
        const uint8_t *q = qual + caps[c].start;
        int f = get_feature(P, s);
        if (f >= 0) { /* "this edit is 100 %" (:77-81) */
            double like = x->dist[P->index[f]];
            INSERT_HIT(like, f, c);
            continue;
        }
        memcpy(test_seq, s, P->len);
        for (uint32_t i = 0; i < P->len; i++) {
            char orig = test_seq[i];
            for (int b = 0; b < 4; b++) {
                if (NUCLEOTIDES[b] == orig) continue;
                test_seq[i] = NUCLEOTIDES[b];
                f = get_feature(P, test_seq);
                if (f >= 0) {
                    uint8_t d = (uint8_t)(q[i] - ILLUMINA_QUAL_OFFSET);
                    double qv = (double)(d < FEATURE_MAX_QV ? d : FEATURE_MAX_QV);
                    double p_edit = pow(10.0, -qv / 10.0);
                    double like = x->dist[P->index[f]] * p_edit;
                    INSERT_HIT(like, f, c);
                }
            }
            test_seq[i] = orig;
        }
    }
#undef INSERT_HIT
    /* the maximum; the HashMap's order decides between equal likelihoods, but two equal maxima give a ratio of at most
     * one half, below the threshold, so the answer does not depend on it */
    double max_likelihood = -1.0;
    int best = -1;
    uint32_t best_cap = 0;
    for (uint32_t e = 0; e < n_map; e++)
        if (map[e].likelihood > max_likelihood) {
            max_likelihood = map[e].likelihood;
            best = map[e].feat;
            best_cap = map[e].cap;
        }
    free(map);
    if ((max_likelihood / likelihood_sum) >= FEATURE_CONF_THRESHOLD) {
        *cap_out = best_cap;
        return best;
    }
    return -1;
}

/* find_closest (:443-470): position in the pattern's feature list or -1; *cap_out = the capture it belongs to */
static int find_closest(const oracle_extractor *x, const ox_pattern *P, const char *seq, const uint8_t *qual,
                        const ox_capture *caps, uint32_t n_caps, uint32_t *cap_out) {
    if (n_caps == 0) return -1;
    if (n_caps == 1) {
        int f = get_feature(P, seq + caps[0].start);
        if (f >= 0) {
            *cap_out = 0;
            return f;
        }
    }
    if (x->dist) return correct_captures(x, P, seq, qual, caps, n_caps, cap_out);
    return -1;
}

/* ---- match_read (:358-441) ------------------------------------------------------------------------------------------ */
int oracle_match_read(const oracle_extractor *x, const char *r1, const uint8_t *q1, uint32_t l1, const char *r2,
                      const uint8_t *q2, uint32_t l2, oracle_feature_data *out) {
    memset(out, 0, sizeof(*out));
    /* whitelist_matches / pattern_matches, reduced on the fly: ids collects every whitelist match; the chosen barcode is
     * the maximum of (len, Reverse(feature index)) -- for pattern matches max_by_key keeps the LAST of equal keys, and
     * equal keys can only come from captures of one pattern (patterns differ in their least feature index) */
    int have_wl = 0, have_pm = 0;
    uint32_t wl_len = 0, wl_idx = 0, pm_len = 0, pm_idx = 0;
    oracle_feature_data pm = {0};
    ox_capture *caps = (ox_capture *)malloc(((size_t)(l1 > l2 ? l1 : l2) + 1) * sizeof(ox_capture));
    for (uint32_t p = 0; p < x->n_pat; p++) {
        const ox_pattern *P = &x->pat[p];
        const char *s = P->read == 0 ? r1 : r2;
        const uint8_t *q = P->read == 0 ? q1 : q2;
        const uint32_t len = P->read == 0 ? l1 : l2;
        if (!s) continue;
        uint32_t n_caps = 0, offset = 0, cs, ce;
        while (offset <= len && regex_captures(P->regex, s + offset, len - offset, &cs, &ce)) {
            caps[n_caps].start = cs + offset;
            caps[n_caps].end = ce + offset;
            n_caps++;
            offset += cs + 1; /* "just after this match" (:394-396) */
            if (P->tethered) break;
        }
        if (n_caps == 0) continue; /* not in regset.matches */
        uint32_t cap = 0;
        int f = find_closest(x, P, s, q, caps, n_caps, &cap);
        if (f >= 0) {
            uint32_t idx = P->index[f];
            if (out->n_ids < ORACLE_MAX_FEATURE_IDS) out->ids[out->n_ids] = idx;
            out->n_ids++;
            if (!have_wl || P->len > wl_len || (P->len == wl_len && idx < wl_idx)) {
                have_wl = 1;
                wl_len = P->len;
                wl_idx = idx;
                out->read = (uint32_t)P->read;
                out->start = caps[cap].start;
                out->len = P->len;
                memcpy(out->corrected_barcode, P->seq[f], P->len);
                out->corrected_barcode[P->len] = 0;
            }
        } else {
            uint32_t least = P->index[0];
            for (uint32_t k = 1; k < P->n_feat; k++)
                if (P->index[k] < least) least = P->index[k];
            for (uint32_t c = 0; c < n_caps; c++) {
                uint32_t blen = caps[c].end - caps[c].start;
                if (!have_pm || blen > pm_len || (blen == pm_len && least <= pm_idx)) {
                    have_pm = 1;
                    pm_len = blen;
                    pm_idx = least;
                    pm.read = (uint32_t)P->read;
                    pm.start = caps[c].start;
                    pm.len = blen;
                }
            }
        }
    }
    free(caps);
    if (have_wl) {
        out->matched = 1;
        out->corrected = 1;
        return 1;
    }
    if (have_pm) {
        out->matched = 1;
        out->corrected = 0;
        out->n_ids = 0;
        out->read = pm.read;
        out->start = pm.start;
        out->len = pm.len;
        return 1;
    }
    return 0;
}

/* match_read over rows (one read per row of `stride` bytes, every row full unless len is given): the feature index when
 * FeatureData::ids holds exactly one id, else ORACLE_NO_FEATURE -- what make_shard_metrics.rs:336-345 counts and what
 * tx_annotation counts as the read's feature (read.rs:983-987).  n_threads > 1: rows fanned out with OpenMP (the CPU
 * baseline of bench.py's cfg4; the reference does this per read inside its chunk processes). */
void oracle_match_rows(const oracle_extractor *x, int which_read, const char *rows, const uint8_t *quals, const uint32_t *len,
                       uint32_t stride, uint64_t n, int n_threads, uint32_t *feature_out) {
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(static, 4096)
    for (uint64_t i = 0; i < n; i++) {
        oracle_feature_data d;
        const uint32_t l = len ? (len[i] < stride ? len[i] : stride) : stride;
        const char *s = rows + i * stride;
        const uint8_t *q = quals + i * stride;
        int ok = which_read == 0 ? oracle_match_read(x, s, q, l, NULL, NULL, 0, &d) : oracle_match_read(x, NULL, NULL, 0, s, q, l, &d);
        feature_out[i] = (ok && d.n_ids == 1) ? d.ids[0] : ORACLE_NO_FEATURE;
    }
}
