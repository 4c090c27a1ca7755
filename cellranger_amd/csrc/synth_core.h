// synth_core.h -- counter-based, integer-only synthetic read generator.
// The same inline code is compiled for the host (crgpu_synth_host) and for gfx950
// (crgpu_synth_dev); read i of a seed is bit-identical on both.
#pragma once
#include <cstdint>

#include "../../include/crgpu.h"

#ifdef __HIPCC__
#define CR_HD __host__ __device__ __forceinline__
#else
#define CR_HD inline
#endif

CR_HD uint64_t cr_mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

CR_HD uint64_t cr_rnd(uint64_t seed, uint64_t i, uint64_t stream) {
    return cr_mix64(cr_mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull)) + i * 0x9E3779B97F4A7C15ull);
}

CR_HD uint64_t cr_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

// first index c with cdf[c] > u   (cdf ascending, cdf[n-1] == 2^63, u < 2^63)
CR_HD uint32_t cr_cdf_search(const uint64_t *cdf, uint32_t n, uint64_t u) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    return lo < n ? lo : n - 1;
}

// quality model: 4-level binned Illumina qualities {F=70, :=58, ,=44, #=35}; errors are biased low
CR_HD uint32_t cr_qual_ok(uint32_t r16) { return r16 < 55706u ? 70u : r16 < 62259u ? 58u : r16 < 64880u ? 44u : 35u; }
CR_HD uint32_t cr_qual_err(uint32_t r16) { return r16 < 6554u ? 70u : r16 < 19661u ? 58u : r16 < 39322u ? 44u : 35u; }

struct CrSynthRead {
    uint32_t cb, umi, feature;
    uint8_t flags;
    uint8_t cbq[16], umiq[16];
};

CR_HD void cr_mutate(uint64_t seed, uint64_t i, uint64_t stream0, uint32_t len, uint32_t err_per_2_16,
                     uint32_t n_per_2_20, uint32_t &seq, uint8_t *q, bool &any_n) {
    any_n = false;
    for (uint32_t j = 0; j < len; j++) {
        const uint64_t r = cr_rnd(seed, i, stream0 + j);
        const uint32_t sh = 2u * (len - 1u - j);
        const uint32_t r_err = (uint32_t)(r & 0xFFFFu);
        const uint32_t r_q = (uint32_t)((r >> 16) & 0xFFFFu);
        const uint32_t r_n = (uint32_t)((r >> 32) & 0xFFFFFu);
        const uint32_t r_sub = (uint32_t)(r >> 52) % 3u;
        if (r_n < n_per_2_20) {
            seq &= ~(3u << sh);  // N is stored as code 0
            q[j] = (uint8_t)(35u | 0x80u);
            any_n = true;
        } else if (r_err < err_per_2_16) {
            const uint32_t b = (seq >> sh) & 3u;
            seq = (seq & ~(3u << sh)) | (((b + 1u + r_sub) & 3u) << sh);
            q[j] = (uint8_t)cr_qual_err(r_q);
        } else {
            q[j] = (uint8_t)cr_qual_ok(r_q);
        }
    }
}

CR_HD void cr_synth_read(const crgpu_synth_params &p, uint64_t i, CrSynthRead &out) {
    const uint64_t r0 = cr_rnd(p.seed, i, 0);
    const bool ambient = p.n_ambient > 0 && (uint32_t)(r0 & 0xFFFFu) < p.ambient_per_2_16;
    uint32_t wl_pos;
    uint64_t w_cell;  // weight of the source (out of 2^63)
    uint64_t cell_id;
    if (ambient) {
        const uint32_t a = (uint32_t)(cr_rnd(p.seed, i, 1) % p.n_ambient);
        wl_pos = p.ambient_wl_pos[a];
        w_cell = 0;
        cell_id = (uint64_t)p.n_cells + a;
    } else {
        const uint64_t u = cr_rnd(p.seed, i, 1) >> 1;
        const uint32_t c = cr_cdf_search(p.cell_cdf, p.n_cells, u);
        wl_pos = p.cell_wl_pos[c];
        w_cell = p.cell_cdf[c] - (c ? p.cell_cdf[c - 1] : 0ull);
        cell_id = c;
    }
    out.cb = p.wl_packed[wl_pos];

    // gene + molecule
    uint32_t g = 0;
    uint64_t w_gene = 1ull << 63;
    if (p.n_genes) {
        const uint64_t u = cr_rnd(p.seed, i, 2) >> 1;
        g = cr_cdf_search(p.gene_cdf, p.n_genes, u);
        w_gene = p.gene_cdf[g] - (g ? p.gene_cdf[g - 1] : 0ull);
    }
    const bool no_feature = (uint32_t)(cr_rnd(p.seed, i, 3) & 0xFFFFu) < p.no_feature_per_2_16;
    out.feature = no_feature ? CRGPU_NO_FEATURE : g;
    // expected reads of (cell, gene) = n_total * (w_cell/2^63) * (w_gene/2^63)
    uint64_t n_mol = 1;
    if (!ambient) {
        const uint64_t prod = (w_cell >> 32) * (w_gene >> 32);          // <= 2^62 ; p_c*p_g ~ prod / 2^62
        const uint64_t hi = cr_mulhi64(p.n_total, prod);                 // (n_total*prod) >> 64
        const uint64_t lo = p.n_total * prod;
        const uint64_t expected = (hi << 2) | (lo >> 62);                // >> 62
        n_mol = expected / (p.reads_per_umi ? p.reads_per_umi : 1u);
        if (n_mol < 1) n_mol = 1;
    }
    const uint64_t r4 = cr_rnd(p.seed, i, 4);
    const uint64_t mol = ambient ? r4 : r4 % n_mol;
    const uint32_t umi_mask = p.umi_len >= 16 ? 0xFFFFFFFFu : ((1u << (2u * p.umi_len)) - 1u);
    out.umi = (uint32_t)cr_mix64(cr_mix64(p.seed ^ 0xA5A5A5A55A5A5A5Aull ^ (cell_id << 20) ^ g) + mol) & umi_mask;

    bool cb_n, umi_n;
    cr_mutate(p.seed, i, 16, p.cb_len, p.cb_err_per_2_16, p.n_per_2_20, out.cb, out.cbq, cb_n);
    cr_mutate(p.seed, i, 48, p.umi_len, p.umi_err_per_2_16, p.n_per_2_20, out.umi, out.umiq, umi_n);
    uint32_t lib = 0;
    if (p.n_libs > 1) lib = (uint32_t)(cr_rnd(p.seed, i, 5) % p.n_libs);
    out.flags = (uint8_t)(lib | (cb_n ? CRGPU_FLAG_CB_HAS_N : 0u));
}

// ---- read rows of a Feature Barcoding library (BASELINE configs[3]) ---------------------------------------------------------
// Row i: row_stride random bases with plain qualities; bases [offset, offset + L) hold the sequence of the read's true
// feature (feat_seq[feature], 2-bit packed, first base most significant) -- or stay random when the read has none --
// with per-base substitutions (quality from the error model) and Ns.  ASCII, as the FASTQ holds them.
CR_HD void cr_synth_row(uint64_t seed, uint64_t i, uint32_t feature, const uint64_t *feat_seq, uint32_t n_feat, uint32_t L,
                        uint32_t offset, uint32_t row_stride, uint32_t err_per_2_16, uint32_t n_per_2_20, uint8_t *seq,
                        uint8_t *qual) {
    const char acgt[4] = {'A', 'C', 'G', 'T'};
    const bool planted = feature < n_feat;
    const uint64_t fs = planted ? feat_seq[feature] : 0ull;
    for (uint32_t b = 0; b < row_stride; b++) {
        const uint64_t r = cr_rnd(seed, i, 1000u + b);
        uint32_t base = (uint32_t)(r >> 60) & 3u;
        uint32_t q = cr_qual_ok((uint32_t)((r >> 16) & 0xFFFFu));
        bool is_n = false;
        if (b >= offset && b < offset + L) {
            if (planted) base = (uint32_t)(fs >> (2u * (L - 1u - (b - offset)))) & 3u;
            const uint32_t r_err = (uint32_t)(r & 0xFFFFu), r_n = (uint32_t)((r >> 32) & 0xFFFFFu);
            if (r_n < n_per_2_20) {
                is_n = true;
                q = 35u;
            } else if (r_err < err_per_2_16) {
                base = (base + 1u + (uint32_t)(r >> 52) % 3u) & 3u;
                q = cr_qual_err((uint32_t)((r >> 16) & 0xFFFFu));
            }
        }
        seq[b] = is_n ? (uint8_t)'N' : (uint8_t)acgt[base];
        qual[b] = (uint8_t)q;
    }
}
