// crgpu.hpp -- header-only C++17 host layer over the C ABI (include/crgpu.h), mirroring the
// reference's Rust interface for this path (names, argument meaning, error behaviour), in batch form:
//
//   crgpu::Whitelist        barcode/src/whitelist.rs:453-546       Whitelist::{Plain, Trans}
//   crgpu::Posterior        barcode/src/corrector.rs:92-109        thresholds, Default
//   crgpu::BarcodeCorrector barcode/src/corrector.rs:14-71         new(whitelist, bc_counts, strategy), correct_barcode
//   crgpu::DupBuilder       tx_annotation/src/mark_dups.rs:118-169 observe(...), build(...)
//   crgpu::BarcodeDupMarker tx_annotation/src/mark_dups.rs:171-363 -> UmiCount stream + feature_counts()
//   crgpu::FeatureExtractor cr_types/src/reference/feature_extraction.rs:176-470  new(feature defs, feat_dist), match_read,
//                           compile_pattern
//   crgpu::BarcodeIndex     cr_types/src/barcode_index.rs:14-53
//   crgpu::CountMatrix      cr_h5/src/count_matrix.rs:382-448, cr_lib/src/stages/write_matrix_market.rs:80-122
//
// Errors are C++ exceptions carrying crgpu_last_error (the Rust returns anyhow::Result); nothing here
// computes on the CPU: every result comes from libcrgpu, and construction fails without a gfx950 device.
#pragma once

#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "crgpu.h"

namespace crgpu {

class Error : public std::runtime_error {
   public:
    Error(int code, const std::string &msg) : std::runtime_error("crgpu error " + std::to_string(code) + ": " + msg), code(code) {}
    int code;
};

class Context {
   public:
    // one GPU: Context(device); one rank of a multi-GPU well: Context(device, n_ranks, rank, id) with the id of
    // crgpu_get_unique_id (processes) or crgpu_local_group_id (threads of this process)
    explicit Context(int device_id = 0, int n_ranks = 1, int rank = 0, const void *unique_id = nullptr) {
        const int rc = crgpu_create(&h_, device_id, n_ranks, rank, unique_id);
        if (rc != CRGPU_OK) throw Error(rc, crgpu_last_error(nullptr));
    }
    ~Context() { crgpu_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    crgpu_ctx *get() const { return h_; }
    void check(int rc) const {
        if (rc != CRGPU_OK) throw Error(rc, crgpu_last_error(h_));
    }

   private:
    crgpu_ctx *h_ = nullptr;
};

/// SimpleHistogram<BcSegSeq> (metric/src/histogram.rs:26-32): sequence -> count, 0 when absent.
using SimpleHistogram = std::map<std::string, int64_t>;

/// Whitelist::{Plain, Trans}.  `translation` empty => Plain.
struct Whitelist {
    std::vector<std::string> sequences;    // raw sequences (file column 1)
    std::vector<std::string> translation;  // translated sequences (file column 2) or empty
    static Whitelist plain(std::vector<std::string> seqs) { return Whitelist{std::move(seqs), {}}; }
    static Whitelist trans(std::vector<std::string> raw, std::vector<std::string> translated) {
        return Whitelist{std::move(raw), std::move(translated)};
    }
};

/// Posterior (corrector.rs:92-109).
struct Posterior {
    double max_expected_barcode_errors = DBL_MAX;  // Default: f64::MAX
    double bc_confidence_threshold = 0.975;        // BARCODE_CONFIDENCE_THRESHOLD
};

/// BarcodeCorrector::new(whitelist, bc_counts, strategy) for one library type of a context.
/// `canon` is the canonical (matrix column) list of the GEM well; empty => the whitelist itself.
class BarcodeCorrector {
   public:
    BarcodeCorrector(Context &ctx, int lib, const Whitelist &wl, const SimpleHistogram &bc_counts, Posterior strategy = {},
                     const std::vector<std::string> &canon = {})
        : ctx_(ctx), lib_(lib) {
        if (wl.sequences.empty()) throw Error(CRGPU_EINVAL, "empty whitelist");
        len_ = (uint32_t)wl.sequences[0].size();
        const std::vector<std::string> &cn = !canon.empty() ? canon : (wl.translation.empty() ? wl.sequences : wl.translation);
        // canonical list: unique sequences in first-seen order
        std::map<std::string, uint32_t> canon_pos;
        std::string canon_flat;
        for (const auto &s : cn)
            if (canon_pos.emplace(s, (uint32_t)canon_pos.size()).second) canon_flat += s;
        std::string keys_flat;
        std::vector<uint32_t> translate_to;
        for (size_t i = 0; i < wl.sequences.size(); i++) {
            keys_flat += wl.sequences[i];
            if (!wl.translation.empty()) {
                auto it = canon_pos.find(wl.translation[i]);
                if (it == canon_pos.end()) throw Error(CRGPU_EINVAL, "translated sequence is not in the canonical list");
                translate_to.push_back(it->second);
            }
        }
        ctx_.check(crgpu_set_whitelist(ctx_.get(), lib, keys_flat.data(), (uint32_t)wl.sequences.size(), len_, canon_flat.data(),
                                       (uint32_t)canon_pos.size(), translate_to.empty() ? nullptr : translate_to.data()));
        // rank <-> sequence of the canonical space
        const uint32_t n = (uint32_t)canon_pos.size();
        std::vector<uint32_t> order(n);
        ctx_.check(crgpu_get_canon_order(ctx_.get(), order.data(), nullptr));
        rank_seq_.resize(n);
        std::vector<std::string> by_pos(n);
        for (const auto &kv : canon_pos) by_pos[kv.second] = kv.first;
        for (uint32_t r = 0; r < n; r++) rank_seq_[r] = by_pos[order[r]];
        for (uint32_t r = 0; r < n; r++) seq_rank_[rank_seq_[r]] = r;
        // prior = bc_counts keyed by the (translated) sequence
        std::vector<uint32_t> prior(n, 0);
        for (const auto &kv : bc_counts) {
            auto it = seq_rank_.find(kv.first);
            if (it != seq_rank_.end()) prior[it->second] = (uint32_t)kv.second;
        }
        ctx_.check(crgpu_set_counts(ctx_.get(), lib, CRGPU_COUNTS_PRIOR, prior.data()));
        ctx_.check(crgpu_set_posterior(ctx_.get(), strategy.max_expected_barcode_errors, strategy.bc_confidence_threshold));
    }

    /// Batch form of correct_barcode for Invalid segments: the corrected (translated) sequence, or
    /// nullopt when no correction is made.  quals may be empty (Option<BcSegQual> = None).
    std::vector<std::optional<std::string>> correct_barcodes(const std::vector<std::string> &seqs,
                                                             const std::vector<std::vector<uint8_t>> &quals) const {
        const size_t n = seqs.size();
        std::string flat;
        std::vector<uint8_t> q;
        for (size_t i = 0; i < n; i++) {
            if (seqs[i].size() != len_) throw Error(CRGPU_EINVAL, "barcode length differs from the whitelist's");
            flat += seqs[i];
            if (!quals.empty()) q.insert(q.end(), quals[i].begin(), quals[i].end());
        }
        std::vector<uint32_t> idx(n, CRGPU_MISS);
        std::vector<uint8_t> flag(n, 0);
        ctx_.check(crgpu_correct(ctx_.get(), lib_, reinterpret_cast<const uint8_t *>(flat.data()), quals.empty() ? nullptr : q.data(), n,
                                 idx.data(), flag.data()));
        std::vector<std::optional<std::string>> out(n);
        for (size_t i = 0; i < n; i++)
            if (flag[i]) out[i] = rank_seq_[idx[i]];
        return out;
    }
    std::optional<std::string> correct_barcode(const std::string &seq, const std::vector<uint8_t> &qual) const {
        return correct_barcodes({seq}, qual.empty() ? std::vector<std::vector<uint8_t>>{} : std::vector<std::vector<uint8_t>>{qual})[0];
    }
    /// Whitelist::check_and_update for a batch: translated sequence on a hit.
    std::vector<std::optional<std::string>> check_and_update(const std::vector<std::string> &seqs) const {
        std::string flat;
        for (const auto &s : seqs) flat += s;
        std::vector<uint32_t> idx(seqs.size(), CRGPU_MISS);
        ctx_.check(crgpu_match_and_count(ctx_.get(), lib_, reinterpret_cast<const uint8_t *>(flat.data()), nullptr, seqs.size(), idx.data()));
        std::vector<std::optional<std::string>> out(seqs.size());
        for (size_t i = 0; i < seqs.size(); i++)
            if (idx[i] != CRGPU_MISS) out[i] = rank_seq_[idx[i]];
        return out;
    }
    uint32_t rank_of(const std::string &seq) const { return seq_rank_.at(seq); }
    const std::string &sequence_of(uint32_t rank) const { return rank_seq_[rank]; }
    uint32_t barcode_length() const { return len_; }

   private:
    Context &ctx_;
    int lib_;
    uint32_t len_ = 0;
    std::vector<std::string> rank_seq_;
    std::map<std::string, uint32_t> seq_rank_;
};

/// UmiCount (cr_types/src/types.rs:152-160) without probe_idx.
struct UmiCount {
    uint32_t barcode_rank;
    uint16_t library_idx;
    uint32_t feature_idx, umi, read_count;
    uint8_t utype;  // 0 Txomic, 1 NonTxomic
};
/// FeatureBarcodeCount (types.rs:121-137), BarcodeThenFeatureOrder.
struct FeatureBarcodeCount {
    uint32_t barcode_rank, feature_idx, umi_count;
};

/// Result of DupBuilder::build: what BarcodeDupMarker::process + BcUmiInfo::feature_counts yield for all
/// barcodes of the batch.
/// DupInfo (mark_dups.rs:61-72): what BarcodeDupMarker::process returns for one read.
struct DupInfo {
    uint32_t processed_umi;  // 2-bit packed corrected UMI
    uint32_t read_count;     // reads of the read's molecule (UmiCount::read_count)
    bool is_corrected, is_low_support_umi, is_umi_count;
};

/// BarcodeSummary (cr_lib/src/aligner.rs:33-68) with the barcode as its canonical rank and the library id in place of
/// the LibraryType; `reads` comes from the context's histograms (the reads check_and_update / correct_barcodes saw).
struct BarcodeSummary {
    uint32_t barcode_rank, library;
    uint64_t reads, umis, candidate_dup_reads, umi_corrected_reads;
};

struct BarcodeDupMarker {
    std::vector<BarcodeSummary> barcode_summaries;    // ordered by (library, barcode), one per barcode with a read
    std::vector<UmiCount> umi_counts;                 // sorted per barcode as align_and_count.rs:314
    std::vector<FeatureBarcodeCount> feature_counts;  // sorted by (barcode, feature)
    /// process(read): one entry per observe() call, in call order (the order stands in for the qname
    /// rank: observe reads in read-header order).  nullopt = process() returned None.
    std::vector<std::optional<DupInfo>> dup_infos;
};

/// DupBuilder: observe() every annotated read of a batch (any number of barcodes / library types), then
/// build().  umi: ASCII; umi_qual: FASTQ quality bytes; feature: conf_mapped_feature or CRGPU_NO_FEATURE.
class DupBuilder {
   public:
    DupBuilder(Context &ctx, uint32_t n_features, uint32_t umi_len, uint32_t n_libs = 1, uint32_t multiplexing_lib_mask = 0)
        : ctx_(ctx), n_features_(n_features), umi_len_(umi_len) {
        ctx_.check(crgpu_set_key_layout(ctx_.get(), n_features, umi_len, n_libs, multiplexing_lib_mask));
    }
    void observe(uint32_t barcode_rank, int lib, const std::string &umi, const std::vector<uint8_t> &umi_qual, uint32_t feature,
                 bool is_txomic = true) {
        if (umi.size() != umi_len_ || umi_qual.size() != umi_len_) throw Error(CRGPU_EINVAL, "UMI length differs from the layout's");
        uint32_t packed = 0;
        for (uint32_t j = 0; j < umi_len_; j++) {
            const char c = umi[j];
            const uint32_t code = c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
            packed = (packed << 2) | (code & 3u);
            qualn_.push_back((uint8_t)((umi_qual[j] > 127 ? 127 : umi_qual[j]) | (code == 4u ? 0x80u : 0u)));
        }
        bc_.push_back(barcode_rank);
        umi_.push_back(packed);
        feature_.push_back(feature);
        flags_.push_back((uint8_t)((lib & 0x0F) | (is_txomic ? 0 : CRGPU_FLAG_NONTXOMIC)));
    }
    BarcodeDupMarker build() {
        const uint64_t n = bc_.size();
        struct Dev {
            const Context &c;
            void *p = nullptr;
            Dev(const Context &c, const void *h, uint64_t bytes) : c(c) {
                c.check(crgpu_malloc(c.get(), &p, bytes));
                c.check(crgpu_memcpy_h2d(c.get(), p, h, bytes));
            }
            ~Dev() { crgpu_free(c.get(), p); }
        };
        BarcodeDupMarker out;
        if (n == 0) return out;
        Dev d_bc(ctx_, bc_.data(), n * 4), d_umi(ctx_, umi_.data(), n * 4), d_q(ctx_, qualn_.data(), n * umi_len_),
            d_f(ctx_, feature_.data(), n * 4), d_fl(ctx_, flags_.data(), n);
        std::vector<uint32_t> pu(n), rc32(n);
        std::vector<uint8_t> df(n);
        Dev d_pu(ctx_, pu.data(), n * 4), d_rc(ctx_, rc32.data(), n * 4), d_df(ctx_, df.data(), n);
        crgpu_records recs{n, umi_len_, (const uint32_t *)d_bc.p, (const uint32_t *)d_umi.p, (const uint8_t *)d_q.p,
                           (const uint32_t *)d_f.p, (const uint8_t *)d_fl.p};
        crgpu_counts *c = nullptr;
        ctx_.check(crgpu_count_records_dev(ctx_.get(), &recs, &c, (uint32_t *)d_pu.p, (uint32_t *)d_rc.p, (uint8_t *)d_df.p));
        uint64_t nt = 0, nm = 0;
        ctx_.check(crgpu_counts_info(ctx_.get(), c, &nt, &nm));
        std::vector<uint32_t> tb(nt), tf(nt), tc(nt), mb(nm), mf(nm), mu(nm), mr(nm);
        std::vector<uint8_t> ml(nm), mt(nm);
        int rc = crgpu_counts_triplets(ctx_.get(), c, tb.data(), tf.data(), tc.data());
        if (rc == CRGPU_OK) rc = crgpu_counts_molecules(ctx_.get(), c, mb.data(), ml.data(), mf.data(), mu.data(), mr.data(), mt.data());
        std::vector<crgpu_barcode_summary_row> rows;
        if (rc == CRGPU_OK) {
            uint64_t n_rows = 0;
            rc = crgpu_counts_barcode_summary(ctx_.get(), c, 0, 0xFFFFFFFFu, nullptr, 0, &n_rows);
            rows.resize(n_rows);
            if (rc == CRGPU_OK && n_rows)
                rc = crgpu_counts_barcode_summary(ctx_.get(), c, 0, 0xFFFFFFFFu, rows.data(), n_rows, &n_rows);
        }
        crgpu_counts_free(ctx_.get(), c);
        ctx_.check(rc);
        ctx_.check(crgpu_memcpy_d2h(ctx_.get(), pu.data(), d_pu.p, n * 4));
        ctx_.check(crgpu_memcpy_d2h(ctx_.get(), rc32.data(), d_rc.p, n * 4));
        ctx_.check(crgpu_memcpy_d2h(ctx_.get(), df.data(), d_df.p, n));
        out.dup_infos.resize(n);
        for (uint64_t i = 0; i < n; i++)
            if (df[i] & CRGPU_DUP_HAS)
                out.dup_infos[i] = DupInfo{pu[i], rc32[i], (df[i] & CRGPU_DUP_CORRECTED) != 0, (df[i] & CRGPU_DUP_LOW_SUPPORT) != 0,
                                           (df[i] & CRGPU_DUP_UMI_COUNT) != 0};
        for (const auto &r : rows)
            out.barcode_summaries.push_back({r.barcode_rank, r.library, r.reads, r.umis, r.candidate_dup_reads, r.umi_corrected_reads});
        for (uint64_t i = 0; i < nt; i++) out.feature_counts.push_back({tb[i], tf[i], tc[i]});
        for (uint64_t i = 0; i < nm; i++) out.umi_counts.push_back({mb[i], ml[i], mf[i], mu[i], mr[i], mt[i]});
        return out;
    }

   private:
    Context &ctx_;
    uint32_t n_features_, umi_len_;
    std::vector<uint32_t> bc_, umi_, feature_;
    std::vector<uint8_t> qualn_, flags_;
};

/// FeatureDef (cr_types/src/reference/feature_reference.rs): the columns FeatureExtractor reads.
struct FeatureDef {
    uint32_t index;        // position in the feature reference (and in feat_dist)
    std::string pattern;   // "5PNNNNNNNNNN(BC)", "^(BC)", "(BC)GTTTAAG...", bare "(BC)"
    std::string sequence;  // A/C/G/T
    int read;              // WhichRead: 0 = R1, 1 = R2
};

/// FeatureData (feature_extraction.rs:163-173) of one read: ids as feature indices, barcode / qual as a span of the read.
struct FeatureData {
    std::vector<uint32_t> ids;        // one id when the read is counted; its value is ids[0] (more ids: only the count is kept)
    uint32_t n_ids = 0;
    int read = 0;
    uint32_t start = 0, len = 0;
    bool corrected = false;           // corrected_barcode is Some
};

/// FeatureExtractor of ONE feature type (match_read skips the definitions of other types, :376-378): new() compiles the
/// patterns (errors as the reference's: invalid pattern / sequence, duplicate definition), match_read runs on the GPU.
class FeatureExtractor {
   public:
    FeatureExtractor(Context &ctx, int slot, const std::vector<FeatureDef> &defs, const std::vector<double> *feat_dist = nullptr)
        : ctx_(ctx), slot_(slot) {
        std::vector<crgpu_feature_def> d(defs.size());
        for (size_t k = 0; k < defs.size(); k++)
            d[k] = crgpu_feature_def{defs[k].pattern.c_str(), defs[k].sequence.c_str(), defs[k].index, (uint32_t)defs[k].read};
        ctx_.check(crgpu_set_feature_extractor(ctx_.get(), slot_, d.data(), (uint32_t)d.size(), feat_dist ? feat_dist->data() : nullptr,
                                               feat_dist ? (uint32_t)feat_dist->size() : 0u));
    }
    /// compile_pattern (:307-343): the regular expression as text; throws for a pattern the reference rejects
    static std::string compile_pattern(const std::string &pattern, uint32_t length) {
        char buf[4096];
        const int rc = crgpu_compile_feature_pattern(pattern.c_str(), length, buf, sizeof(buf));
        if (rc != CRGPU_OK) throw Error(rc, "Invalid pattern: '" + pattern + "'");
        return buf;
    }
    /// regex_str of every compiled pattern (tethered ones as compile_pattern gives them, bare groups as compile_bare_patterns)
    std::vector<std::string> regexes() const {
        uint32_t n = 0;
        ctx_.check(crgpu_feature_extractor_regex(ctx_.get(), slot_, 0, nullptr, 0, &n));
        std::vector<std::string> out;
        std::vector<char> buf(1 << 20);
        for (uint32_t p = 0; p < n; p++) {
            ctx_.check(crgpu_feature_extractor_regex(ctx_.get(), slot_, p, buf.data(), buf.size(), nullptr));
            out.emplace_back(buf.data());
        }
        return out;
    }
    /// match_read (:358-441) for a batch of read pairs (either read may be absent: empty vectors); nullopt = None
    std::vector<std::optional<FeatureData>> match_reads(const std::vector<std::string> &r1_seq, const std::vector<std::string> &r1_qual,
                                                        const std::vector<std::string> &r2_seq,
                                                        const std::vector<std::string> &r2_qual) const {
        const size_t n = r1_seq.empty() ? r2_seq.size() : r1_seq.size();
        Rows a = upload(r1_seq, r1_qual), b = upload(r2_seq, r2_qual);
        DevBuf f(ctx_, n * 4), ni(ctx_, n * 4), cap(ctx_, n * 4);
        ctx_.check(crgpu_extract_features_dev(ctx_.get(), slot_, a.seq.u8(), a.qual.u8(), a.len.u32(), a.stride, b.seq.u8(), b.qual.u8(),
                                              b.len.u32(), b.stride, n, f.u32(), ni.u32(), cap.u32()));
        std::vector<uint32_t> hf(n), hn(n), hc(n);
        ctx_.check(crgpu_memcpy_d2h(ctx_.get(), hf.data(), f.p, n * 4));
        ctx_.check(crgpu_memcpy_d2h(ctx_.get(), hn.data(), ni.p, n * 4));
        ctx_.check(crgpu_memcpy_d2h(ctx_.get(), hc.data(), cap.p, n * 4));
        std::vector<std::optional<FeatureData>> out(n);
        for (size_t i = 0; i < n; i++) {
            if (hc[i] == CRGPU_NO_CAPTURE) continue;
            FeatureData d;
            d.corrected = (hc[i] >> 31) != 0;
            d.read = (int)((hc[i] >> 30) & 1u);
            d.start = (hc[i] >> 8) & 0x3FFFFFu;
            d.len = hc[i] & 0xFFu;
            d.n_ids = hn[i];
            if (hn[i] == 1) d.ids.push_back(hf[i]);
            out[i] = d;
        }
        return out;
    }

   private:
    struct DevBuf {
        Context &c;
        void *p = nullptr;
        DevBuf(Context &ctx, size_t bytes) : c(ctx) {
            if (bytes) c.check(crgpu_malloc(c.get(), &p, bytes));
        }
        ~DevBuf() {
            if (p) crgpu_free(c.get(), p);
        }
        DevBuf(const DevBuf &) = delete;
        DevBuf(DevBuf &&o) noexcept : c(o.c), p(o.p) { o.p = nullptr; }
        uint8_t *u8() const { return (uint8_t *)p; }
        uint32_t *u32() const { return (uint32_t *)p; }
    };
    struct Rows {
        DevBuf seq, qual, len;
        uint32_t stride;
    };
    Rows upload(const std::vector<std::string> &seq, const std::vector<std::string> &qual) const {
        uint32_t stride = 0;
        for (const auto &s : seq) stride = s.size() > stride ? (uint32_t)s.size() : stride;
        stride = (stride + 3u) & ~3u;
        const size_t n = seq.size();
        Rows r{DevBuf(ctx_, n * stride), DevBuf(ctx_, n * stride), DevBuf(ctx_, n * 4), stride};
        if (n == 0) return r;
        std::vector<uint8_t> s(n * stride, 0), q(n * stride, 0);
        std::vector<uint32_t> l(n);
        for (size_t i = 0; i < n; i++) {
            std::copy(seq[i].begin(), seq[i].end(), s.begin() + i * stride);
            std::copy(qual[i].begin(), qual[i].end(), q.begin() + i * stride);
            l[i] = (uint32_t)seq[i].size();
        }
        ctx_.check(crgpu_memcpy_h2d(ctx_.get(), r.seq.p, s.data(), s.size()));
        ctx_.check(crgpu_memcpy_h2d(ctx_.get(), r.qual.p, q.data(), q.size()));
        ctx_.check(crgpu_memcpy_h2d(ctx_.get(), r.len.p, l.data(), n * 4));
        return r;
    }
    Context &ctx_;
    int slot_;
};

/// CountMatrix: the arrays write_matrix_h5 stores (count_matrix.rs:382-448) + write_matrix_mtx.
class CountMatrix {
   public:
    CountMatrix(Context &ctx, const std::vector<FeatureBarcodeCount> &counts, uint32_t n_features) : ctx_(ctx) {
        std::vector<uint32_t> b, f, c;
        for (const auto &x : counts) {
            b.push_back(x.barcode_rank);
            f.push_back(x.feature_idx);
            c.push_back(x.umi_count);
        }
        ctx_.check(crgpu_assemble_matrix(ctx_.get(), b.data(), f.data(), c.data(), counts.size(), n_features, &m_));
    }
    ~CountMatrix() { crgpu_matrix_free(ctx_.get(), m_); }
    CountMatrix(const CountMatrix &) = delete;
    CountMatrix &operator=(const CountMatrix &) = delete;
    const crgpu_matrix &arrays() const { return *m_; }
    /// BarcodeIndex::sorted_barcodes as canonical ranks
    std::vector<uint32_t> barcode_ranks() const { return {m_->barcode_rank, m_->barcode_rank + m_->n_barcodes}; }
    void write_matrix_mtx(const std::string &mtx_path, const std::string &barcodes_tsv_path, const std::string &metadata_line,
                          uint16_t gem_group = 1) const {
        ctx_.check(crgpu_write_mtx(ctx_.get(), m_, metadata_line.c_str(), mtx_path.c_str(),
                                   barcodes_tsv_path.empty() ? nullptr : barcodes_tsv_path.c_str(), gem_group));
    }

   private:
    Context &ctx_;
    crgpu_matrix *m_ = nullptr;
};

}  // namespace crgpu
