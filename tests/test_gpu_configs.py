"""Down-scaled parity cases of BASELINE.json configs[3] (Feature Barcoding: reads matched against a feature
reference, then counted) and configs[4] (several GEM wells, one per GPU, merged matrix), through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cfg4_feature_barcoding_match_then_count_bit_exact():
    """Antibody-Capture style library: the feature of a read is not given by an aligner but found by matching a
    15-base capture against the feature reference (exact, else posterior-corrected with the feature distribution:
    feature_extraction.rs:34-117), then barcode correction, UMI dedup and the matrix as for any library."""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd._lib import NO_FEATURE

    n, n_feat, L = 100_000, 48, 15
    rng = np.random.default_rng(44)
    w = S.Workload(n_total=n, seed=44, n_wl=3000, n_cells=60, n_ambient=300, n_genes=n_feat, reads_per_umi=3)
    r = w.host_reads(0, n)
    true_feat = r["feature"].copy()
    feats = np.unique(rng.integers(0, 1 << 30, size=4 * n_feat, dtype=np.uint64))[:n_feat].astype(np.uint32)
    feat_ascii = E.unpack_seqs(rng.permutation(feats), L)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seq = np.where((true_feat != NO_FEATURE)[:, None], feat_ascii[np.minimum(true_feat, n_feat - 1)], acgt[rng.integers(0, 4, (n, L))])
    u = rng.random(n)
    for i in np.nonzero(u < 0.12)[0]:  # one substitution, sometimes an N
        seq[i, rng.integers(0, L)] = ord("N") if u[i] < 0.02 else acgt[rng.integers(0, 4)]
    qual = rng.choice(np.array([35, 44, 58, 70], np.uint8), size=(n, L))
    # feature distribution from the exact matches, as MAKE_SHARD collects it (make_shard_metrics.rs:338-352)
    lut = {bytes(f): i for i, f in enumerate(feat_ascii)}
    exact = np.array([lut.get(bytes(s), -1) for s in seq])
    dist = O.compute_feature_dist(np.bincount(exact[exact >= 0], minlength=n_feat), np.zeros(n_feat, np.uint32))

    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    c.set_feature_pattern(0, feat_ascii, np.arange(n_feat, dtype=np.uint32), dist)
    code = np.full(256, 0, np.uint32)
    code[acgt] = np.arange(4)
    pk = np.zeros(n, np.uint32)
    for j in range(L):
        pk = (pk << np.uint32(2)) | code[seq[:, j]]
    qn = (qual | np.where(seq == ord("N"), 0x80, 0)).astype(np.uint8)
    d_feature = c.empty(n, np.uint32)
    c.match_features(0, c.upload(pk), c.upload(qn), n, d_feature)
    got_feat = d_feature.to_host()
    exp_feat = np.array([O.find_closest_feature(feat_ascii, dist, bytes(seq[i]), bytes(qual[i])) for i in range(n)])
    exp_feat = np.where(exp_feat < 0, NO_FEATURE, exp_feat).astype(np.uint32)
    assert np.array_equal(got_feat, exp_feat)
    assert (exp_feat != NO_FEATURE).mean() > 0.8 and (exp_feat != np.where(exact < 0, NO_FEATURE, exact)).sum() > 1000

    # barcode stage + count stage with the matched features (device resident end to end)
    idx_a, idx_b, corr, dev = G.gpu_barcode_stage(c, r, n)
    c.set_key_layout(n_feat, w.umi_len, 1, 0)
    recs = c.records(n, w.umi_len, dev["idx"], c.upload(r["umi"]), c.upload(r["umi_qualn"]), d_feature, dev["flags"])
    m = c.count(recs, n_feat)
    r2 = dict(r)
    r2["feature"] = exp_feat
    res = O.run_pipeline(G.oracle_reads_from_packed(r2, w.cb_len, w.umi_len), [O.Whitelist(E.unpack_seqs(w.wl_packed, 16))],
                         n_threads=4)
    assert np.array_equal(m.barcodes_ascii(), res.barcodes)
    assert np.array_equal(m.indptr, res.indptr) and np.array_equal(m.indices, res.indices) and np.array_equal(m.data, res.data)
    assert m.nnz > 1000
    c.close()


def test_cfg5_wells_counted_separately_then_merged(tmp_path):
    """Two GEM wells of one sample, each counted on its own (same whitelist, different cells and molecules), merged
    through crgpu_concat_matrices: columns in (gem_group, barcode) order, barcodes.tsv rows carry the well suffix."""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S

    n = 120_000
    w = S.Workload(n_total=2 * n, seed=55, n_wl=20_000, n_cells=80, n_ambient=2000, n_genes=300)
    wl = O.Whitelist(E.unpack_seqs(w.wl_packed, 16))
    mats, oracle = [], []
    ctxs = []
    for well in range(2):
        c = G.fresh_ctx()
        ctxs.append(c)
        c.set_whitelist(0, w.wl_packed, length=16)
        c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
        r = w.host_reads(well * n, n)
        idx_a, idx_b, corr, dev = G.gpu_barcode_stage(c, r, n)
        recs = c.records(n, w.umi_len, dev["idx"], c.upload(r["umi"]), c.upload(r["umi_qualn"]), c.upload(r["feature"]), dev["flags"])
        mats.append(c.count(recs, w.n_genes))
        oracle.append(O.run_pipeline(G.oracle_reads_from_packed(r, 16, w.umi_len), [wl], n_threads=4))
    merged = ctxs[0].concat_matrices(mats, [1, 2])
    exp_barcodes = np.concatenate([o.barcodes for o in oracle])
    exp_indptr = np.concatenate([oracle[0].indptr, oracle[1].indptr[1:] + oracle[0].indptr[-1]])
    assert np.array_equal(merged.barcodes_ascii(), exp_barcodes)
    assert np.array_equal(merged.gem_group, np.repeat([1, 2], [len(o.barcodes) for o in oracle]).astype(np.uint16))
    assert np.array_equal(merged.indptr, exp_indptr)
    assert np.array_equal(merged.indices, np.concatenate([o.indices for o in oracle]))
    assert np.array_equal(merged.data, np.concatenate([o.data for o in oracle]))
    p_mtx, p_bc = tmp_path / "m.mtx", tmp_path / "barcodes.tsv"
    merged.write_mtx(p_mtx, p_bc)
    rows = open(p_bc).read().split()
    assert rows[0] == bytes(exp_barcodes[0]).decode() + "-1" and rows[-1] == bytes(exp_barcodes[-1]).decode() + "-2"
    assert len(rows) == len(exp_barcodes) and merged.nnz > 1000
    # not a merge of merges, and the groups must ascend
    from cellranger_amd._lib import CrgpuError
    with pytest.raises(CrgpuError):
        ctxs[0].concat_matrices(mats, [2, 1])
    for c in ctxs:
        c.close()


def test_shard_metrics_match_oracle():
    """The fused MAKE_SHARD metric scan over packed arrays equals the oracle's per-read loop, including
    N bases, low qualities below the Q30 denominator cut-off, homopolymers and whitelist misses."""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import synth as S
    from cellranger_amd._lib import MISS

    n = 300_000
    w = S.Workload(n_total=n, seed=66, n_wl=5000, n_cells=50, n_ambient=500, n_genes=100, n_rate=0.01, cb_err=0.02)
    r = w.host_reads(0, n)
    rng = np.random.default_rng(66)
    # spread the qualities over the thresholds (2, 10, 30) and plant homopolymers
    lowq = np.array([33 + 1, 33 + 2, 33 + 3, 33 + 9, 33 + 10, 33 + 29, 33 + 30, 33 + 40], np.uint8)
    for key, L in (("cb_qualn", 16), ("umi_qualn", 12)):
        keep_n = r[key] & 0x80
        r[key] = (lowq[rng.integers(0, len(lowq), (n, L))] | keep_n).astype(np.uint8)
    r["umi"][rng.random(n) < 0.01] = 0           # AAAAAAAAAAAA
    r["cb"][rng.random(n) < 0.005] = 0xFFFFFFFF   # TTTTTTTTTTTTTTTT
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    idx_a, idx_b, corr, dev = G.gpu_barcode_stage(c, r, n)
    d_idx_a = c.upload(idx_a)
    got = c.shard_metrics(dev["cb"], dev["cbq"], 16, c.upload(r["umi"]), c.upload(r["umi_qualn"]), 12, d_idx_a, n)
    cb, cbq = S.to_ascii(r["cb"], r["cb_qualn"], 16)
    umi, uq = S.to_ascii(r["umi"], r["umi_qualn"], 12)
    exp = O.shard_metrics(cb, cbq, umi, uq, exact_hit=(idx_a != MISS).astype(np.uint8))
    assert got == exp
    assert exp["bc_n_bases"] > 1000 and exp["homopolymer_umi"] > 1000 and exp["low_min_qual_barcode"] > n // 2
    assert 0 < exp["bc_q30_bases"] < exp["bc_q30_den"] < exp["bc_bases"]
    c.close()


def test_aggr_merge_and_barcode_selection_match_scipy():
    """aggr-style post-processing: CountMatrix.merge is scipy's `m += other.m` on equal-shape matrices
    and select_barcodes is scipy column indexing (lib/python/cellranger/matrix.py:479-482,860-875) -- scipy itself is
    the checker here.  Two libraries of the SAME cells (same barcode index) are counted separately and summed."""
    import scipy.sparse as sp

    import gpu_helpers as G
    from cellranger_amd import synth as S
    from cellranger_amd._lib import CrgpuError

    n = 150_000
    w = S.Workload(n_total=2 * n, seed=77, n_wl=20_000, n_cells=60, n_ambient=1500, n_genes=200)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    # pass A/B over both halves so that both matrices share one barcode index, then count each half on its own
    r = w.host_reads(0, 2 * n)
    idx_a, idx_b, corr, dev = G.gpu_barcode_stage(c, r, 2 * n)
    d_umi, d_uq, d_ft = c.upload(r["umi"]), c.upload(r["umi_qualn"]), c.upload(r["feature"])
    mats = []
    for h in range(2):
        recs = c.records(n, w.umi_len, dev["idx"].ptr + 4 * h * n, d_umi.ptr + 4 * h * n, d_uq.ptr + 12 * h * n,
                         d_ft.ptr + 4 * h * n, dev["flags"].ptr + h * n)
        mats.append(c.count(recs, w.n_genes))
    a, b = mats
    assert np.array_equal(a.barcode_rank, b.barcode_rank) and a.nnz > 1000 and b.nnz > 1000

    def to_sp(m):
        return sp.csc_matrix((m.data, m.indices, m.indptr), shape=(m.n_features, m.n_barcodes))

    tot = c.sum_matrices(a, b)
    exp = to_sp(a) + to_sp(b)
    exp.sort_indices()
    assert np.array_equal(tot.indptr, exp.indptr) and np.array_equal(tot.indices, exp.indices) and np.array_equal(tot.data, exp.data)
    assert np.array_equal(tot.barcode_rank, a.barcode_rank)
    rng = np.random.default_rng(7)
    cols = rng.permutation(a.n_barcodes)[: a.n_barcodes // 3]
    sel = c.select_barcodes(tot, cols)
    exp_sel = exp[:, cols]
    assert np.array_equal(sel.indptr, exp_sel.indptr) and np.array_equal(sel.indices, exp_sel.indices)
    assert np.array_equal(sel.data, exp_sel.data) and np.array_equal(sel.barcode_rank, a.barcode_rank[cols])
    with pytest.raises(CrgpuError):
        c.sum_matrices(tot, sel)  # shapes differ
    # the same two operations on DEVICE matrices (crgpu_sum_matrices_dev / crgpu_select_barcodes_dev)
    dmats = []
    for h in range(2):
        recs = c.records(n, w.umi_len, dev["idx"].ptr + 4 * h * n, d_umi.ptr + 4 * h * n, d_uq.ptr + 12 * h * n,
                         d_ft.ptr + 4 * h * n, dev["flags"].ptr + h * n)
        keys = c.empty(n, np.uint64)
        cnt = c.count_keys(keys, c.build_keys(recs, keys))
        tb, tf, tc = cnt.triplets_dev()
        dmats.append((c.assemble_matrix_dev(tb, tf, tc, cnt.n_triplets), cnt))
    dtot = c.sum_matrices_dev(dmats[0][0], dmats[1][0])
    rank, indptr, indices, data = dtot.download()
    assert np.array_equal(rank, a.barcode_rank) and np.array_equal(indptr, exp.indptr)
    assert np.array_equal(indices, exp.indices) and np.array_equal(data, exp.data)
    dsel = c.select_barcodes_dev(dtot, cols)
    rank, indptr, indices, data = dsel.download()
    assert np.array_equal(rank, a.barcode_rank[cols]) and np.array_equal(indptr, exp_sel.indptr)
    assert np.array_equal(indices, exp_sel.indices) and np.array_equal(data, exp_sel.data)
    empty = c.select_barcodes_dev(dtot, np.zeros(0, np.uint64))
    assert empty.n_barcodes == 0 and empty.nnz == 0
    with pytest.raises(CrgpuError):
        c.sum_matrices_dev(dtot, dsel)
    c.close()


def test_merge_molecules_barcode_trimming_matches_the_reference_rule():
    """aggr's MERGE_MOLECULES on the barcode_idx column: trim_barcodes (cr_h5/src/molecule_info.rs:890-960: retain the
    pass_filter barcodes and every barcode with a molecule, ascending, re-index) + the per-sample offset of the join
    (cr_aggr/src/merge_molecules.rs:131-330), restated in numpy for two samples."""
    import gpu_helpers as G
    from cellranger_amd._lib import CrgpuError

    rng = np.random.default_rng(3)
    c = G.fresh_ctx()
    offset = 0
    for n_bc, n_mol, n_pass in ((5000, 40_000, 300), (737_280, 2_000_000, 9000)):
        used_pool = rng.choice(n_bc, size=n_bc // 4, replace=False)
        idx = np.sort(rng.choice(used_pool, size=n_mol)).astype(np.uint64)      # molecule rows are sorted by barcode
        pf = np.sort(rng.choice(n_bc, size=n_pass, replace=False)).astype(np.uint64)
        for pass_only in (False,):
            keep = np.union1d(pf, idx) if not pass_only else pf
            newpos = np.full(n_bc, -1, np.int64)
            newpos[keep] = np.arange(len(keep))
            d_idx = c.upload(idx)
            retained, pf_new = c.trim_molecule_barcodes(d_idx, n_mol, n_bc, pf, pass_only=pass_only, offset=offset)
            assert np.array_equal(retained, keep.astype(np.uint64))
            assert np.array_equal(d_idx.to_host(), (offset + newpos[idx]).astype(np.uint64))
            assert np.array_equal(pf_new, (offset + newpos[pf]).astype(np.uint64))
        offset += len(keep)
    # pass_only with a molecule outside pass_filter is an error (the reference panics on it), never a silent remap
    d_idx = c.upload(np.array([1, 2, 7], np.uint64))
    with pytest.raises(CrgpuError):
        c.trim_molecule_barcodes(d_idx, 3, 10, np.array([1, 2], np.uint64), pass_only=True)
    r, pf = c.trim_molecule_barcodes(d_idx, 3, 10, np.array([1, 2, 7, 9], np.uint64), pass_only=True)
    assert list(r) == [1, 2, 7, 9] and list(d_idx.to_host()) == [0, 1, 2] and list(pf) == [0, 1, 2, 3]
    with pytest.raises(CrgpuError):
        c.trim_molecule_barcodes(c.upload(np.array([11], np.uint64)), 1, 10)
    c.close()


def _fastq_text(seqs, quals, crlf=False, final_newline=True):
    nl = b"\r\n" if crlf else b"\n"
    recs = [b"@read%d some text" % i + nl + s + nl + b"+" + nl + q for i, (s, q) in enumerate(zip(seqs, quals))]
    return nl.join(recs) + (nl if final_newline else b"")


def test_fastq_ingest_rows_and_whole_read_metrics():
    """The MAKE_SHARD side of the ingest on the device: FASTQ text -> read rows (crgpu_fastq_to_rows_dev), the
    whole-read N / Q30 fractions and the perfect-homopolymer flags of MakeShardVisitor::visit_processed_read
    (cr_lib/src/make_shard_metrics.rs:266-300,355-392), then the barcode / UMI slices packed from the rows and their shard
    metrics incl. polyt_suffix_umi -- each against a plain numpy / Python restatement of the cited lines.  (The reference's
    FASTQ reader, PatternCheck and has_polyt_suffix live in the un-vendored fastq_set crate: parity unpinned.)"""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd._lib import CrgpuError

    rng = np.random.default_rng(17)
    n = 30_000
    seqs, quals = [], []
    for i in range(n):
        L = int(rng.integers(28, 120))
        s = bytearray(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L, p=[0.24, 0.24, 0.24, 0.24, 0.04]).tobytes())
        if i % 7 == 0:   # plant homopolymers of 14..17 bases
            k, b = int(rng.integers(14, 18)), b"ACGT"[i % 4]
            at = int(rng.integers(0, max(1, L - k)))
            s[at:at + k] = bytes([b]) * min(k, L - at)
        if i % 11 == 0:  # UMIs (bases 16..28) ending in T's
            s[23:28] = b"TTTTT"
        seqs.append(bytes(s))
        quals.append(rng.choice(np.array([33, 35, 36, 44, 58, 63, 70], np.uint8), size=L).tobytes())
    stride = 128
    c = G.fresh_ctx()
    for crlf, final_nl in ((False, True), (True, False)):
        text = np.frombuffer(_fastq_text(seqs, quals, crlf, final_nl), np.uint8)
        d_text = c.upload(text)
        d_seq, d_qual, d_len = c.empty((n, stride), np.uint8), c.empty((n, stride), np.uint8), c.empty(n, np.uint32)
        assert c.fastq_to_rows(d_text, len(text), stride, n, d_seq, d_qual, d_len) == n
        rows, qrows, lens = d_seq.to_host(), d_qual.to_host(), d_len.to_host()
        for i in (0, 1, 2, n // 2, n - 2, n - 1):
            L = len(seqs[i])
            assert lens[i] == L and rows[i, :L].tobytes() == seqs[i] and qrows[i, :L].tobytes() == quals[i]
            assert not rows[i, L:].any() and not qrows[i, L:].any()
        assert np.array_equal(lens, np.array([len(s) for s in seqs], np.uint32))
    # whole-read metrics (frac_n_bases, frac_q30_bases)
    m = c.rows_metrics(d_seq, d_qual, n, stride, d_len)
    allq = np.frombuffer(b"".join(quals), np.uint8)
    alls = np.frombuffer(b"".join(seqs), np.uint8)
    assert m == dict(n_bases=int((alls == ord("N")).sum()), bases=len(alls), q30_bases=int((allq >= 63).sum()),
                     q30_den=int((allq > 35).sum()))
    # perfect homopolymers: 15 equal bases in a row, in R1 or (second call) in R1 or R2
    def has(s, b):
        return (bytes([b]) * 15) in s
    exp1 = {ch: sum(has(s, ord(ch)) for s in seqs) for ch in "ACGT"}
    assert c.homopolymer_metrics(d_seq, stride, n, d_r1_len=d_len) == exp1 and min(exp1.values()) > 100
    r2 = [seqs[(i * 7 + 3) % n] for i in range(n)]
    d_r2 = c.upload(np.stack([np.frombuffer(s.ljust(stride, b"\0"), np.uint8) for s in r2]))
    d_r2len = c.upload(np.array([len(s) for s in r2], np.uint32))
    exp2 = {ch: sum(has(a, ord(ch)) or has(b, ord(ch)) for a, b in zip(seqs, r2)) for ch in "ACGT"}
    assert c.homopolymer_metrics(d_seq, stride, n, d_r2_rows=d_r2, r2_stride=stride, d_r1_len=d_len, d_r2_len=d_r2len) == exp2
    # the barcode / UMI slices from the rows, their metrics with the 5-T suffix flag
    d_cb, d_cbq, d_fl = c.empty(n, np.uint32), c.empty((n, 16), np.uint8), c.zeros(n, np.uint8)
    d_umi, d_uq = c.empty(n, np.uint32), c.empty((n, 10), np.uint8)
    c.pack_rows(d_seq, d_qual, n, stride, 0, 16, d_cb, d_cbq, d_fl)
    c.pack_rows(d_seq, d_qual, n, stride, 16, 10, d_umi, d_uq, None)
    got = c.shard_metrics(d_cb, d_cbq, 16, d_umi, d_uq, 10, None, n)
    cb = np.stack([np.frombuffer(s[:16], np.uint8) for s in seqs])
    cbq = np.stack([np.frombuffer(q[:16], np.uint8) for q in quals])
    um = np.stack([np.frombuffer(s[16:26], np.uint8) for s in seqs])
    uq = np.stack([np.frombuffer(q[16:26], np.uint8) for q in quals])
    exp = O.shard_metrics(cb, cbq, um, uq)
    exp["miss_whitelist_barcode"] = 0
    assert got == exp
    # malformed input is refused
    bad = np.frombuffer(b"@r\nACGT\n+\nIII\n", np.uint8)
    with pytest.raises(CrgpuError):
        c.fastq_to_rows(c.upload(bad), len(bad), stride, 4, d_seq, d_qual, d_len)
    bad = np.frombuffer(b"@r\nACGT\n+\nIIII\n@x\nAC\n", np.uint8)
    with pytest.raises(CrgpuError):
        c.fastq_to_rows(c.upload(bad), len(bad), stride, 4, d_seq, d_qual, d_len)
    c.close()


def test_cfg4_one_flow_trans_whitelist_pattern_two_libraries():
    """BASELINE configs[3] as SURVEY 8(d) specifies it, as ONE flow on the device: an Antibody Capture library on a `Trans`
    whitelist (raw FB barcodes translate onto the GEX list, whitelist.rs:497-504) beside a Gene Expression library in the
    same GEM well; the FB reads' features come from whole R2 rows through the anchored pattern ^N{10}(BC)
    (crgpu_extract_features_dev), first without a distribution to collect MAKE_SHARD's exact-match counts
    (make_shard_metrics.rs:336-345 -> compute_feature_dist), then with it; both libraries are then counted together.
    Every stage is compared with the oracle: per-read barcode ranks, the prior, per-read features, per-read DupInfo, matrix."""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd._lib import NO_FEATURE

    n0, n1, n_genes, n_fb, L, off, stride = 40_000, 60_000, 300, 48, 15, 10, 40
    rng = np.random.default_rng(40404)   # not the workload's seed: the same PCG64 stream would redraw its whitelist
    kw = dict(n_wl=3000, n_cells=60, n_ambient=300, reads_per_umi=3)
    w = S.Workload(n_total=n0, seed=404, n_genes=n_genes, **kw)
    canon = w.wl_packed
    raw = np.setdiff1d(np.unique(rng.integers(0, 1 << 32, size=5000, dtype=np.uint64)).astype(np.uint32), canon)[:3000]
    raw = rng.permutation(raw)
    translate_to = rng.permutation(3000).astype(np.uint32)     # raw[i] pairs with canon[translate_to[i]]
    # the FB library's reads come from the same cells: same tables, raw barcodes at the paired positions
    w_fb = S.Workload(n_total=n1, seed=404, n_genes=n_fb, **kw)
    inv = np.empty(3000, np.uint32)
    inv[translate_to] = np.arange(3000, dtype=np.uint32)
    w_fb.wl_packed[:] = raw[inv]      # position p of the generator's list <-> canon[p]
    w_fb.c.seed = 405
    r0, r1 = w.host_reads(0, n0), w_fb.host_reads(0, n1)
    r1["flags"] |= 1
    feats = np.unique(rng.integers(0, 1 << 30, size=4 * n_fb, dtype=np.uint64))[:n_fb]
    feats = rng.permutation(feats)
    feat_ascii = [bytes(x).decode() for x in E.unpack_seqs(feats.astype(np.uint32), L)]
    rows_s, rows_q = E.synth_rows_host(777, 0, n1, r1["feature"], feats, L, off, stride, err=0.01, n_rate=0.002)
    defs = [("5PNNNNNNNNNN(BC)", feat_ascii[k], n_genes + k, 1) for k in range(n_fb)]
    n_feat_all = n_genes + n_fb
    types = np.array([0] * n_genes + [1] * n_fb, np.uint32)

    # ---- device flow --------------------------------------------------------------------------------------------------
    c = G.fresh_ctx()
    c.set_whitelist(0, canon, length=16)
    c.set_whitelist(1, raw, canon=canon, translate_to=translate_to, length=16)
    _, canon_sorted = c.canon_order()
    dev = []
    for r, n in ((r0, n0), (r1, n1)):       # one library per call: the one-library kernels
        d = dict(cb=c.upload(r["cb"]), cbq=c.upload(r["cb_qualn"]), flags=c.upload(r["flags"]), idx=c.empty(n, np.uint32), n=n)
        c.match_and_count(d["cb"], d["flags"], n, d["idx"])
        dev.append(d)
    d_rs, d_rq = c.upload(rows_s), c.upload(rows_q)
    d_f1 = c.empty(n1, np.uint32)
    d_n_ids, d_cap = c.empty(n1, np.uint32), c.empty(n1, np.uint32)
    c.set_feature_extractor(1, defs)                                   # MAKE_SHARD: no distribution yet
    c.extract_features(1, n1, d_f1, r2=(d_rs, d_rq, None, stride), d_n_ids_out=d_n_ids, d_capture_out=d_cap)
    counts = c.feature_counts(d_f1, n1, n_feat_all)
    dist = E.compute_feature_dist(counts, types)
    c.set_feature_extractor(1, defs, dist)
    for d in dev:
        c.correct(d["cb"], d["cbq"], d["flags"], d["n"], d["idx"])
    # the pass with the distribution only corrects the captures the first pass kept (same rows, definitions and outputs, the
    # context may keep by-products): the rows are not read again ...
    c.extract_features(1, n1, d_f1, r2=(d_rs, d_rq, None, stride), d_n_ids_out=d_n_ids, d_capture_out=d_cap)
    assert c.stat(4) == 2                                              # both passes took the one-pattern LDS kernel
    resumed = c.stat(8)
    assert 0 < resumed < n1 // 2, resumed                              # CRGPU_STAT_FEATURE_RESUMED_READS
    got_f1, got_ids, got_cap = d_f1.to_host(), d_n_ids.to_host(), d_cap.to_host()
    # ... and gives what a full pass over the rows gives (nothing kept: crgpu_invalidate)
    c.invalidate()
    c.extract_features(1, n1, d_f1, r2=(d_rs, d_rq, None, stride), d_n_ids_out=d_n_ids, d_capture_out=d_cap)
    assert c.stat(8) == resumed and c.stat(4) == 3
    assert np.array_equal(d_f1.to_host(), got_f1) and np.array_equal(d_n_ids.to_host(), got_ids) and np.array_equal(d_cap.to_host(), got_cap)
    idx_all = np.concatenate([dev[0]["idx"].to_host(), dev[1]["idx"].to_host()])
    n = n0 + n1
    feat_all = np.concatenate([r0["feature"], got_f1])
    flags_all = np.concatenate([r0["flags"], r1["flags"]])
    umi_all, uq_all = np.concatenate([r0["umi"], r1["umi"]]), np.concatenate([r0["umi_qualn"], r1["umi_qualn"]])
    c.set_key_layout(n_feat_all, 12, 2, 0)
    recs = c.records(n, 12, c.upload(idx_all), c.upload(umi_all), c.upload(uq_all), c.upload(feat_all), c.upload(flags_all))
    d_pu, d_rc, d_fl = c.empty(n, np.uint32), c.empty(n, np.uint32), c.empty(n, np.uint8)
    cnt = c.count_records(recs, d_pu, d_rc, d_fl)
    m = c.assemble_matrix(*cnt.triplets(), n_feat_all)

    # ---- oracle -------------------------------------------------------------------------------------------------------
    ox0 = O.FeatureExtractor(defs, None)
    ocounts = np.zeros(n_feat_all, np.int64)
    for i in range(n1):
        h = ox0.match_read(None, None, bytes(rows_s[i]), bytes(rows_q[i]))
        if h is not None and h["n_ids"] == 1:
            ocounts[h["ids"][0]] += 1
    assert np.array_equal(counts, ocounts) and counts[n_genes:].sum() > n1 // 2 and not counts[:n_genes].any()
    odist = O.compute_feature_dist(ocounts, types)
    assert np.array_equal(dist, odist)
    ox1 = O.FeatureExtractor(defs, odist)
    exp_f1 = np.full(n1, NO_FEATURE, np.uint32)
    for i in range(n1):
        h = ox1.match_read(None, None, bytes(rows_s[i]), bytes(rows_q[i]))
        if h is not None and h["n_ids"] == 1:
            exp_f1[i] = h["ids"][0]
    assert np.array_equal(got_f1, exp_f1)
    exact = (exp_f1 != NO_FEATURE).sum()
    assert exact > counts.sum() + 500, "the posterior must recover reads the exact pass missed"
    r_all = {k: np.concatenate([r0[k], r1[k]]) for k in r0}
    r_all["feature"] = np.concatenate([r0["feature"], exp_f1])
    owl0 = O.Whitelist(E.unpack_seqs(canon, 16))
    owl1 = O.Whitelist(E.unpack_seqs(raw, 16), translated=E.unpack_seqs(canon[translate_to], 16))
    res = O.run_pipeline(G.oracle_reads_from_packed(r_all, 16, 12), [owl0, owl1], n_lib=2, n_threads=4, want_dupinfo=True)
    _, exp_b = G.oracle_expected_idx(res, canon_sorted)
    assert np.array_equal(idx_all, exp_b)
    for lib in (0, 1):
        assert np.array_equal(c.get_counts(lib, 0), G.hist_as_rank_counts(res.valid_hist[lib], 16, canon_sorted))
    assert (res.bc_state[n0:] == 2).sum() > 500        # FB reads corrected through the Trans list
    assert np.array_equal(m.barcodes_ascii(), res.barcodes)
    assert np.array_equal(m.indptr, res.indptr) and np.array_equal(m.indices, res.indices) and np.array_equal(m.data, res.data)
    assert (m.indices >= n_genes).sum() > 500 and (m.indices < n_genes).sum() > 500    # both libraries reach the matrix
    od, fl = res.dupinfo, d_fl.to_host()
    has = od["has_dupinfo"] != 0
    assert np.array_equal((fl & 1) != 0, has)
    for bit, name in ((2, "is_corrected"), (4, "is_low_support"), (8, "is_umi_count")):
        assert np.array_equal((fl & bit) != 0, od[name] != 0), name
    assert np.array_equal(d_pu.to_host()[has], od["processed_umi"][has]) and np.array_equal(d_rc.to_host()[has], od["read_count"][has])
    c.close()
