"""GPU parity tests of the count stage (keys, radix sort, UMI correction, low support, triplets,
CSC) through the C ABI, bit-for-bit against the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_gpu(c, r, n, w_cb_len, umi_len, n_features, n_libs=1, mux_mask=0):
    import gpu_helpers as G

    idx_a, idx_b, corr, dev = G.gpu_barcode_stage(c, r, n)
    c.set_key_layout(n_features, umi_len, n_libs, mux_mask)
    d_umi, d_uq, d_ft = c.upload(r["umi"]), c.upload(r["umi_qualn"]), c.upload(r["feature"])
    recs = c.records(n, umi_len, dev["idx"], d_umi, d_uq, d_ft, dev["flags"])
    d_keys = c.empty(max(n, 1), np.uint64)
    nk = c.build_keys(recs, d_keys)
    counts = c.count_keys(d_keys, nk)
    if counts.n_molecules:   # the corrected-read table is kept only on request
        with pytest.raises(Exception, match="corrected-read table"):
            counts.barcode_summary()
    bc, ft, ct = counts.triplets()
    mol = counts.molecules()
    mol["info"] = counts.molecule_info(gem_group=1)
    m = c.assemble_matrix(bc, ft, ct, n_features)
    # the one-call convenience entry point must agree
    m2 = c.count(recs, n_features)
    for a in ("barcode_rank", "indptr", "indices", "data"):
        assert np.array_equal(getattr(m, a), getattr(m2, a))
    # per-read DupInfo entry point: same triplets, plus one record per read
    d_pu, d_rc, d_fl = c.empty(max(n, 1), np.uint32), c.empty(max(n, 1), np.uint32), c.empty(max(n, 1), np.uint8)
    counts3 = c.count_records(recs, d_pu, d_rc, d_fl)
    for a, b in zip(counts3.triplets(), (bc, ft, ct)):
        assert np.array_equal(a, b)
    dup = dict(processed_umi=d_pu.to_host(count=n), read_count=d_rc.to_host(count=n), flags=d_fl.to_host(count=n))
    # BarcodeSummary rows: from the DupInfo path, and from the plain path once asked for
    dup["summary"] = counts3.barcode_summary()
    c.enable_barcode_summary(True)
    nk = c.build_keys(recs, d_keys)
    counts4 = c.count_keys(d_keys, nk)
    c.enable_barcode_summary(False)
    assert np.array_equal(counts4.barcode_summary(), dup["summary"])
    W = len(c.canon_order()[1])
    parts = [counts4.barcode_summary(lo, hi) for lo, hi in ((0, W // 3), (W // 3, W // 3), (W // 3, W))]
    assert sum(len(p) for p in parts) == len(dup["summary"])
    whole = np.concatenate(parts)
    order = np.lexsort((whole["barcode_rank"], whole["library"]))
    assert np.array_equal(whole[order], dup["summary"])
    # the host-pointer entry of SURVEY 8(b) (crgpu_count_host): same matrix, DupInfo as an array of structs, and with probe
    # indices per read the UmiCount::probe_idx of every molecule
    probe = (((np.arange(n, dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(7)) % np.uint64(37)).astype(np.int32) - 1
    m3, hd, counts5 = c.count_host(n_features, idx_b, r["umi"], r["umi_qualn"].reshape(n, umi_len), r["feature"], r["flags"],
                                   probe_idx=probe, want_counts=True)
    for a in ("barcode_rank", "indptr", "indices", "data"):
        assert np.array_equal(getattr(m, a), getattr(m3, a)), a
    assert np.array_equal(hd["processed_umi"], dup["processed_umi"]) and np.array_equal(hd["read_count"], dup["read_count"])
    assert np.array_equal(hd["flags"], dup["flags"]) and not hd["reserved"].any()
    for k, v in counts5.molecules().items():
        assert np.array_equal(v, mol[k]), k
    dup["probe_in"], dup["mol_probe"] = probe, counts5.probe_idx()
    m4, none = c.count_host(n_features, idx_b, r["umi"], r["umi_qualn"].reshape(n, umi_len), r["feature"], r["flags"], want_dupinfo=False)
    assert none is None and np.array_equal(m4.data, m.data) and np.array_equal(m4.indptr, m.indptr)
    if counts.n_molecules:
        with pytest.raises(Exception, match="without crgpu_records.d_probe_idx"):
            counts3.probe_idx()
    return idx_b, (bc, ft, ct), mol, m, dup


def _compare_with_oracle(c, w, r, n, n_features, n_libs=1, mux_mask=0, whitelists=None):
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E

    _, canon_sorted = c.canon_order()
    idx_b, trip, mol, m, dup = _run_gpu(c, r, n, w.cb_len, w.umi_len, n_features, n_libs, mux_mask)
    if whitelists is None:
        whitelists = [O.Whitelist(E.unpack_seqs(w.wl_packed, w.cb_len))] * n_libs
    res = O.run_pipeline(G.oracle_reads_from_packed(r, w.cb_len, w.umi_len), whitelists, n_lib=n_libs,
                         multiplexing_lib_mask=mux_mask, n_threads=4, want_dupinfo=True)
    # per-read DupInfo (mark_dups.rs:61-72): every field of every read
    od = res.dupinfo
    has = od["has_dupinfo"] != 0
    assert np.array_equal((dup["flags"] & 1) != 0, has)
    assert np.array_equal((dup["flags"] & 2) != 0, od["is_corrected"] != 0)
    assert np.array_equal((dup["flags"] & 4) != 0, od["is_low_support"] != 0)
    assert np.array_equal((dup["flags"] & 8) != 0, od["is_umi_count"] != 0)
    assert np.array_equal((dup["flags"] & 16) != 0, od["is_filtered_target"] != 0)
    assert np.array_equal(dup["processed_umi"][has], od["processed_umi"][has])
    assert np.array_equal(dup["read_count"][has], od["read_count"][has])
    assert not dup["read_count"][~has].any() and not dup["processed_umi"][~has].any()
    assert int(((dup["flags"] & 8) != 0).sum()) == len(mol["bc"])      # one representative read per molecule
    # UmiCount::probe_idx (mark_dups.rs:332-342): the probe of the read process() marks is_umi_count.  The reads with that
    # flag and the molecules are the same set of (barcode, library, feature, corrected UMI): align both by sorting
    rep = np.flatnonzero(od["is_umi_count"] != 0)
    r_order = np.lexsort((od["processed_umi"][rep], r["feature"][rep], (r["flags"][rep] & 0x0F), idx_b[rep]))
    m_order = np.lexsort((mol["umi"], mol["feature"], mol["lib"], mol["bc"]))
    assert np.array_equal(idx_b[rep][r_order], mol["bc"][m_order]) and np.array_equal(od["processed_umi"][rep][r_order], mol["umi"][m_order])
    assert np.array_equal(dup["mol_probe"][m_order], dup["probe_in"][rep][r_order])
    # BarcodeSummary (aligner.rs:33-68) per (library, barcode)
    osum, gsum = O.barcode_summary(res, (r["flags"] & 0x0F)), dup["summary"]
    assert len(gsum) == len(osum["reads"])
    assert np.array_equal(gsum["library"], osum["library"])
    assert np.array_equal(E.unpack_seqs(canon_sorted[gsum["barcode_rank"]], w.cb_len), osum["barcode"]) if len(gsum) else True
    for f in ("reads", "umis", "candidate_dup_reads", "umi_corrected_reads"):
        assert np.array_equal(gsum[f], osum[f]), f
    # columns, indptr, indices, data: the arrays write_matrix_h5 stores
    assert np.array_equal(m.barcodes_ascii(), res.barcodes)
    assert np.array_equal(m.indptr, res.indptr)
    assert np.array_equal(m.indices, res.indices)
    assert np.array_equal(m.data, res.data)
    # molecule table (UmiCount stream, types.rs:152-160) in the reference's per-barcode order
    col_rank = G.ranks_of(canon_sorted, res.barcodes) if len(res.barcodes) else np.zeros(0, np.uint32)
    assert np.array_equal(mol["bc"], col_rank[res.mol_bc_col])
    assert np.array_equal(mol["lib"], res.mol_lib)
    assert np.array_equal(mol["feature"], res.mol["feature_idx"])
    assert np.array_equal(mol["umi"], res.mol["umi"])
    assert np.array_equal(mol["read_count"], res.mol["read_count"])
    assert np.array_equal(mol["utype"], res.mol["utype"])
    # molecule_info.h5 datasets (MoleculeInfoWriter::fill, cr_h5/src/molecule_info.rs:972-998)
    info = mol["info"]
    assert np.array_equal(info["barcode_idx"], res.mol_bc_col.astype(np.uint64))
    assert np.array_equal(info["library_idx"], res.mol_lib.astype(np.uint16))
    assert np.array_equal(info["feature_idx"], res.mol["feature_idx"]) and np.array_equal(info["umi"], res.mol["umi"])
    assert np.array_equal(info["count"], res.mol["read_count"])
    assert np.array_equal(info["umi_type"], res.mol["utype"].astype(np.uint32)) and (info["gem_group"] == 1).all()
    return res, m


def test_radix_sort_matches_numpy():
    """The ballot-multisplit LSD sort, exercised through partition + count on raw keys."""
    import gpu_helpers as G

    from cellranger_amd import synth as S

    c = G.fresh_ctx(dense=False)   # hand-made keys in the whitelist-rank layout
    w = S.Workload(n_total=1000, seed=1, n_wl=1000, n_cells=10, n_ambient=10)
    c.set_whitelist(0, w.wl_packed, length=16)
    c.set_key_layout(1000, 12, 1, 0)
    rng = np.random.default_rng(1)
    # keys: [bc 10 bits][feature 10][umi 24][1]
    n = 300_000
    bc = rng.integers(0, 1000, n).astype(np.uint64)
    ft = rng.integers(0, 1000, n).astype(np.uint64)
    umi = rng.integers(0, 1 << 24, n).astype(np.uint64)
    keys = (bc << np.uint64(35)) | (ft << np.uint64(25)) | (umi << np.uint64(1))
    d_in, d_out = c.upload(keys), c.empty(n, np.uint64)
    for n_ranks in (1, 2, 3, 8):
        cnt = c.partition_keys(d_in, n, n_ranks, d_out)
        out = d_out.to_host()
        width = (1000 + n_ranks - 1) // n_ranks     # contiguous barcode-rank ranges
        owner = (keys >> np.uint64(35)) // np.uint64(width)
        exp = np.concatenate([keys[owner == r] for r in range(n_ranks)])  # stable
        assert np.array_equal(out, exp)
        assert list(cnt) == [int((owner == r).sum()) for r in range(n_ranks)]
    # explicit (histogram-balanced style) ranges, including an empty one
    bounds = np.array([0, 10, 10, 700, 1000], np.uint32)
    cnt = c.partition_keys(d_in, n, 4, d_out, bounds=bounds)
    owner = np.searchsorted(bounds[1:].astype(np.int64), (keys >> np.uint64(35)).astype(np.int64), side="right")
    exp = np.concatenate([keys[owner == r] for r in range(4)])
    assert np.array_equal(d_out.to_host(), exp) and list(cnt) == [int((owner == r).sum()) for r in range(4)]
    assert cnt[1] == 0
    # balanced bounds come from the context's histograms: none yet -> all ranges collapse but stay valid
    b = c.balanced_bounds(4)
    assert b[0] == 0 and b[-1] == 1000 and (np.diff(b.astype(np.int64)) >= 0).all()
    c.close()


def test_cfg3_model_1m_records_matrix_bit_exact():
    """1 M-record down-scale of cfg3: CSC arrays and molecule table equal the oracle's."""
    import gpu_helpers as G
    from cellranger_amd import synth as S

    n = 1_000_000
    w = S.Workload(n_total=n, seed=S.SEED0 + 3, n_cells=300, n_ambient=20000)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    r = w.host_reads(0, n)
    res, m = _compare_with_oracle(c, w, r, n, w.n_genes)
    assert m.nnz > 100_000 and m.n_barcodes > 10_000
    # UMI correction and low-support filtering both happened
    assert res.mol["read_count"].max() > 4
    c.close()


def test_adversarial_dense_umis_chains_ties_multilib():
    """Tiny UMI space + few features + several library types: long Hamming chains, count ties,
    zero-count keys, UMIs shared by features (low support), a Multiplexing library with UMI
    correction disabled, NonTxomic reads, invalid UMIs."""
    import gpu_helpers as G
    from cellranger_amd import synth as S
    from cellranger_amd._lib import FLAG_NONTXOMIC

    for seed, umi_len, n_genes, n in [(11, 4, 3, 60_000), (12, 5, 7, 120_000), (13, 6, 40, 200_000), (14, 3, 2, 20_000)]:
        w = S.Workload(n_total=n, seed=seed, n_wl=2000, n_cells=40, n_ambient=500, n_genes=n_genes, umi_len=umi_len,
                       umi_err=0.05, cb_err=0.01, n_rate=0.002, no_feature_frac=0.1, reads_per_umi=2, n_libs=3)
        c = G.fresh_ctx()
        for lib in range(3):
            c.set_whitelist(lib, w.wl_packed, length=16)
        r = w.host_reads(0, n)
        rng = np.random.default_rng(seed)
        r["flags"] = (r["flags"] | np.where(rng.random(n) < 0.3, FLAG_NONTXOMIC, 0)).astype(np.uint8)
        res, m = _compare_with_oracle(c, w, r, n, n_genes, n_libs=3, mux_mask=0b100)
        assert m.nnz > 0
        c.close()


def test_large_segments_use_binary_search_path():
    """One barcode, one feature, thousands of distinct UMIs: segments that cross tile edges at every size class of
    k_correct_umis_edges (<= 1024 keys, <= 4096 keys staged whole, larger ones chunked)."""
    import gpu_helpers as G
    from cellranger_amd import synth as S

    for n, min_mol in ((150_000, 2000), (30_000, 800), (9_000, 200)):
        w = S.Workload(n_total=n, seed=21, n_wl=100, n_cells=3, n_ambient=0, n_genes=2, umi_len=8, umi_err=0.02,
                       cb_err=0.0, n_rate=0.0, no_feature_frac=0.0, reads_per_umi=3, sigma=0.1)
        c = G.fresh_ctx()
        c.set_whitelist(0, w.wl_packed, length=16)
        r = w.host_reads(0, n)
        res, m = _compare_with_oracle(c, w, r, n, 2)
        assert m.data.max() > min_mol
        c.close()


def test_empty_and_degenerate_inputs():
    import gpu_helpers as G
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd._lib import MISS, NO_FEATURE

    w = S.Workload(n_total=1000, seed=2, n_wl=500, n_cells=20, n_ambient=100, n_genes=5)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    c.set_key_layout(5, 12, 1, 0)
    # no reads at all
    recs = c.records(0, 12, None, None, None, None, None)
    m = c.count(recs, 5)
    assert m.n_barcodes == 0 and m.nnz == 0 and list(m.indptr) == [0]
    assert c.count_records(recs, None, None, None).n_triplets == 0
    # more than 2^31 - 1 records in one call: refused before anything is read or allocated
    dummy = c.empty(16, np.uint64)
    too_many = c.records(1 << 31, 12, dummy, dummy, dummy, dummy, dummy)
    for call in (lambda: c.build_keys(too_many, dummy), lambda: c.count_records(too_many), lambda: c.count_keys(dummy, 1 << 31)):
        with pytest.raises(E.CrgpuError) as ei:
            call()
        assert ei.value.code == -6 and "2^31-1" in str(ei.value)
    # reads whose barcodes are all invalid / features all NONE: columns exist only for seen barcodes
    n = 1000
    r = w.host_reads(0, n)
    r["feature"][:] = NO_FEATURE
    res, m = _compare_with_oracle(c, w, r, n, 5)
    assert m.nnz == 0 and m.n_barcodes > 0
    c.close()


def test_mtx_text_matches_oracle(tmp_path):
    """write_matrix_mtx text (write_matrix_market.rs:96-118) is byte-identical to the oracle's."""
    import ctypes as C

    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S

    n = 50_000
    w = S.Workload(n_total=n, seed=8, n_wl=3000, n_cells=50, n_ambient=400, n_genes=30)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    r = w.host_reads(0, n)
    _, _, _, m, _ = _run_gpu(c, r, n, 16, 12, 30)
    meta = '%metadata_json: {"software_version": "cellranger-amd", "format_version": 2}'
    p_gpu, p_bc = tmp_path / "gpu.mtx", tmp_path / "barcodes.tsv"
    m.write_mtx(p_gpu, p_bc, metadata_line=meta, gem_group=1)

    # oracle text
    reads = G.oracle_reads_from_packed(r, 16, 12)
    owl = O.Whitelist(E.unpack_seqs(w.wl_packed, 16))
    # re-run the oracle keeping the C matrix alive long enough to write it
    res = O.run_pipeline(reads, [owl])
    lines = ["%%MatrixMarket matrix coordinate integer general", meta, "%d %d %d" % (30, len(res.barcodes), len(res.data))]
    for col in range(len(res.barcodes)):
        for k in range(res.indptr[col], res.indptr[col + 1]):
            lines.append("%d %d %d" % (1 + res.indices[k], 1 + col, res.data[k]))
    assert p_gpu.read_text() == "\n".join(lines) + "\n"
    assert p_bc.read_text().splitlines() == [bytes(b).decode() + "-1" for b in res.barcodes]
    c.close()


def test_barcode_summary_csv(tmp_path):
    """barcode_summary.csv (align_and_count.rs:806-817): two libraries of ONE library type are summed per barcode,
    another type follows; rows ordered by (library type, barcode)."""
    import gpu_helpers as G

    from cellranger_amd import engine as E
    from cellranger_amd import synth as S

    c = G.fresh_ctx()
    n = 60_000
    w = S.Workload(n_total=n, seed=77, n_wl=3000, n_cells=40, n_ambient=100, n_genes=50, n_libs=3)
    for lib in range(3):
        c.set_whitelist(lib, w.wl_packed, length=16)
    r = w.host_reads(0, n)
    _, _, _, dev = G.gpu_barcode_stage(c, r, n)
    c.set_key_layout(w.n_genes, w.umi_len, 3, 0)
    recs = c.records(n, w.umi_len, dev["idx"], c.upload(r["umi"]), c.upload(r["umi_qualn"]), c.upload(r["feature"]), dev["flags"])
    rows = c.count_records(recs).barcode_summary()
    assert set(rows["library"]) == {0, 1, 2}
    path = str(tmp_path / "barcode_summary.csv")
    types = (("Gene Expression", 0), ("Antibody Capture", 2), ("Gene Expression", 0))
    c.write_barcode_summary_csv(rows, path, gem_group=3, library_types=types)
    lines = open(path).read().split("\n")
    assert lines[0] == "library_type,barcode,reads,umis,candidate_dup_reads,umi_corrected_reads" and lines[-1] == ""
    _, canon_sorted = c.canon_order()
    exp = {}
    for row in rows:
        t = types[row["library"]]
        seq = bytes(E.unpack_seqs(canon_sorted[row["barcode_rank"]:row["barcode_rank"] + 1], 16)[0]).decode() + "-3"
        acc = exp.setdefault((t[1], seq, t[0]), np.zeros(4, np.uint64))
        acc += np.array([row["reads"], row["umis"], row["candidate_dup_reads"], row["umi_corrected_reads"]], np.uint64)
    want = ["%s,%s,%d,%d,%d,%d" % (k[2], k[1], *v) for k, v in sorted(exp.items(), key=lambda kv: kv[0][:2])]
    assert lines[1:-1] == want
    assert int(rows["reads"].sum()) == int((dev["idx"].to_host() != E.MISS).sum())

def _needs_onesweep():
    """The tests of the onesweep machinery itself (watchdog fallback, finishing passes) assert on its counters: they have no
    meaning in a run of the suite that forces the classic three-kernel passes (CRGPU_SORT=classic)."""
    import os

    if os.environ.get("CRGPU_SORT") == "classic":
        pytest.skip("CRGPU_SORT=classic: the onesweep path under test is switched off")


@pytest.mark.parametrize("big_run,finish,wide", [(50_000, "0", False), (50_000, "2", False), (90_000, "2", False), (50_000, "3", False),
                                                 (90_000, "3", False), (50_000, "2", True), (90_000, "2", True)])
def test_clustered_umis_and_long_runs_of_near_identical_keys(big_run, finish, wide, monkeypatch):
    """UMIs that differ only in their last bases (three 8-base prefixes), UmiTypes mixed inside every UMI, and keys that
    agree in everything but the last 2.5 bases of the UMI in runs of 2..30, of 100 and of 50 000 reads: Hamming-1
    neighbourhoods are dense, counts tie, and one (barcode, feature) segment holds most of the reads.
    finish (CRGPU_SORT_FINISH): "0" = radix passes on every key bit; "2" (the default since round 3) = the sort leaves the
    lowest key bits that save a pass to k_find_descents + k_repair_runs (only the runs of equal top bits that are out of order
    are touched: up to 12 keys in registers, up to 64 by a wave, longer ones by a workgroup in LDS, beyond 4096 keys an in-place
    bucket permutation through memory up to 65 536); "3" = round 2's k_order_runs (runs inside
    a wave by an odd-even transposition in registers).  A run of 90 000 keys makes either hand the job back to a sort on all
    bits (CRGPU_STAT_SORT_REFINISHED).  wide: a 6.8 M-entry whitelist and 36 601 features make the key 64 bits wide; the
    passes then leave TEN low bits (six passes instead of seven): runs of 13 .. 64 keys are put in order by a wave (a lane per
    key, place = number of keys that go before it), the long run by the two-level bucket permutation."""
    top_bits_sort = finish != "0"
    if top_bits_sort:
        _needs_onesweep()
    import gpu_helpers as G

    monkeypatch.setenv("CRGPU_SORT_FINISH", finish)
    from cellranger_amd import synth as S
    from cellranger_amd._lib import FLAG_NONTXOMIC

    n = 260_000
    w = S.Workload(n_total=n, seed=31, n_wl=6_794_880 if wide else 2000, n_cells=40, n_ambient=200, n_genes=40, umi_len=12, umi_err=0.0,
                   cb_err=0.01, n_rate=0.001, no_feature_frac=0.05, reads_per_umi=1)
    c = G.fresh_ctx(dense=False)   # the run lengths this test plants belong to the 42-bit whitelist-rank layout
    c.set_whitelist(0, w.wl_packed, length=16)
    r = w.host_reads(0, n)
    rng = np.random.default_rng(31)
    prefix = rng.integers(0, 1 << 16, 3, dtype=np.uint32)        # three 8-base prefixes
    r["umi"] = ((prefix[rng.integers(0, 3, n)] << np.uint32(8)) | rng.integers(0, 256, n, dtype=np.uint32)).astype(np.uint32)
    # one run of 50 000 reads: same barcode, feature and first 9.5 bases; 20 runs of 100 reads
    assert not (r["flags"][0] & 0x10) and not (r["cb_qualn"][0] & 0x80).any()
    r["cb"][:big_run], r["feature"][:big_run] = r["cb"][0], 7
    r["cb_qualn"][:big_run] = r["cb_qualn"][0]
    r["umi"][:big_run] = (r["umi"][0] & ~np.uint32(31)) | rng.integers(0, 32, big_run, dtype=np.uint32)
    for g in range(20):
        s = big_run + 100 * g
        r["cb"][s:s + 100], r["feature"][s:s + 100] = r["cb"][s], g
        r["cb_qualn"][s:s + 100] = r["cb_qualn"][s]
        r["umi"][s:s + 100] = (r["umi"][s] & ~np.uint32(31)) | rng.integers(0, 32, 100, dtype=np.uint32)
    r["flags"][:big_run + 2_000] = r["flags"][0] & 0x0F     # the planted runs copy a barcode without N
    r["flags"] = (r["flags"] | np.where(rng.random(n) < 0.4, FLAG_NONTXOMIC, 0)).astype(np.uint8)
    res, m = _compare_with_oracle(c, w, r, n, 36_601 if wide else 40)
    assert m.nnz > 1000
    assert (c.stat(1) > 0) == (top_bits_sort and big_run > 65_536)   # CRGPU_STAT_SORT_REFINISHED
    c.close()


@pytest.mark.parametrize("status64", [False, True])
def test_random_key_layouts_bit_exact(status64, monkeypatch):
    """Random whitelist sizes, feature counts, UMI lengths and library counts: key widths from 20 to 64 bits, i.e. every
    digit plan of the sort (all 8-bit, mixed 8/9-bit, a narrow last digit), the classic and the onesweep path, keys with
    and without library bits -- each compared with the oracle read by read."""
    if status64:   # the 64-bit status words of sorts of >= 2^30 keys, forced for inputs of test size
        monkeypatch.setenv("CRGPU_SORT_STATUS64", "1")
    import gpu_helpers as G
    from cellranger_amd import synth as S
    from cellranger_amd._lib import FLAG_NONTXOMIC

    rng = np.random.default_rng(2024)
    seen_bits = set()
    for trial in range(14):
        umi_len = int(rng.choice([3, 4, 5, 6, 8, 10, 11, 12, 12, 12]))
        n_genes = int(rng.choice([1, 2, 3, 17, 100, 1000, 5000, 36601, 60000]))
        n_wl = int(rng.choice([64, 300, 5000, 100_000, 737_280]))
        n_libs = int(rng.choice([1, 1, 2, 3, 5]))
        bits = 1 + 2 * umi_len + int(np.ceil(np.log2(max(n_libs, 1)))) + int(np.ceil(np.log2(max(n_genes, 1)))) + int(np.ceil(np.log2(n_wl)))
        if bits > 64:
            continue
        seen_bits.add(bits)
        n = int(rng.integers(5_000, 120_000))
        n_cells = min(int(rng.integers(3, 200)), n_wl // 2)
        w = S.Workload(n_total=n, seed=1000 + trial, n_wl=n_wl, n_cells=n_cells, n_ambient=min(300, n_wl - n_cells),
                       n_genes=n_genes, umi_len=umi_len, umi_err=0.02, cb_err=0.01, n_rate=0.002, no_feature_frac=0.05,
                       reads_per_umi=int(rng.integers(1, 5)), n_libs=n_libs)
        c = G.fresh_ctx()
        for lib in range(n_libs):
            c.set_whitelist(lib, w.wl_packed, length=16)
        r = w.host_reads(0, n)
        r["flags"] = (r["flags"] | np.where(rng.random(n) < 0.2, FLAG_NONTXOMIC, 0)).astype(np.uint8)
        _compare_with_oracle(c, w, r, n, n_genes, n_libs=n_libs)
        c.close()
    assert len(seen_bits) >= 8 and max(seen_bits) >= 58 and min(seen_bits) <= 30


@pytest.mark.parametrize("bad_pass,finish,status64", [(0, "2", False), (2, "2", False), (3, "2", False), (5, "0", False),
                                                      (1, "2", True), (3, "0", True)])
def test_onesweep_watchdog_falls_back_to_the_classic_passes(bad_pass, finish, status64, monkeypatch):
    """The look-back chain of one onesweep pass is stalled on purpose (CRGPU_SORT_FORCE_ABORT: chunk 0 never publishes).
    The watchdog raises the abort word, that pass and the ones queued behind it write nothing, and the host finishes the
    sort from the failed pass on with the classic histogram / scan / scatter passes inside the same call: the call
    succeeds and its results are identical to an undisturbed run."""
    _needs_onesweep()
    import gpu_helpers as G
    from cellranger_amd import synth as S

    monkeypatch.setenv("CRGPU_SORT_FINISH", finish)
    if status64:   # the 64-bit look-back status words every sort of >= 2^30 keys uses (no test input is that large)
        monkeypatch.setenv("CRGPU_SORT_STATUS64", "1")
    n = 400_000
    w = S.Workload(n_total=n, seed=58, n_wl=60_000, n_cells=150, n_ambient=5000, n_genes=500)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    # 16 + 9 + 24 + 1 = 50 bits: six passes (8+8+8+8+9+9) on all bits (finish "0"), five 9-bit passes on the top 45 bits +
    # k_repair_runs on the low 5 (the default, "2"; with dense barcode keys -- CRGPU_TEST_DENSE=1, 47 bits -- four passes +
    # eleven low bits: the stalled pass of a "2" case is one of the first four)
    c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    r = w.host_reads(0, n)
    _, _, _, dev = G.gpu_barcode_stage(c, r, n)
    d_umi, d_uq, d_ft = c.upload(r["umi"]), c.upload(r["umi_qualn"]), c.upload(r["feature"])
    recs = c.records(n, w.umi_len, dev["idx"], d_umi, d_uq, d_ft, dev["flags"])

    def run():
        keys = c.empty(n, np.uint64)
        nk = c.build_keys(recs, keys)
        assert nk > 200_000   # more than one 16 K-key chunk, or nothing can stall
        cnt = c.count_keys(keys, nk)
        out = cnt.triplets() + tuple(cnt.molecules()[k] for k in ("bc", "feature", "umi", "read_count", "utype"))
        cnt.free()
        return out

    ref = run()
    assert c.stat(0) == 0
    monkeypatch.setenv("CRGPU_SORT_FORCE_ABORT", str(bad_pass))
    got = run()
    monkeypatch.delenv("CRGPU_SORT_FORCE_ABORT")
    assert c.stat(0) == 1, "the forced stall did not reach the fallback"
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    again = run()       # and the context keeps working with the fast path
    assert c.stat(0) == 1
    for a, b in zip(again, ref):
        assert np.array_equal(a, b)
    c.close()


def test_3m_whitelist_64_bit_keys_bit_exact():
    """3M-february-2018-sized list x 36 601 features x 12-base UMIs: the molecule key needs 23 + 16 + 24 + 1 = 64 bits
    (eight radix passes).  500 k records: every read's index and DupInfo, the matrix, the molecule table and the
    BarcodeSummary rows equal the oracle's."""
    import gpu_helpers as G
    from cellranger_amd import synth as S

    n = 500_000
    w = S.Workload(n_total=n, seed=S.SEED0 + 7, n_wl=6_794_880, n_cells=2000, n_ambient=50_000)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    r = w.host_reads(0, n)
    res, m = _compare_with_oracle(c, w, r, n, w.n_genes)
    assert m.nnz > 50_000
    c.close()


def test_finishing_pass_hands_long_runs_back_to_the_full_sort(monkeypatch):
    """(experimental path, CRGPU_SORT_FINISH=1) The radix passes sort the molecule keys on their top bits and k_finish_runs orders the runs of equal top bits.
    One (barcode, feature) whose UMIs all share their leading bases makes a run far longer than the finishing pass
    stages (FIN_RUN_MAX): the sort must notice, redo the buffer on all key bits, and give the same molecules as the
    plain seven / eight-pass sort (CRGPU_SORT_FINISH=0) -- checked through the oracle."""
    _needs_onesweep()
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E

    monkeypatch.setenv("CRGPU_SORT_FINISH", "1")
    rng = np.random.default_rng(5)
    n = 60_000
    wl = ["ACGTACGTACGTACGT", "TTTTACGTACGTACGA", "GGGGACGTACGTACCC"]
    c = G.fresh_ctx(dense=False)   # records without a barcode stage: no tables to take a BarcodeIndex from
    c.set_whitelist_ascii(0, wl)
    c.set_key_layout(40_000, 12, 1, 0)   # 2 + 16 + 24 + 1 = 43 bits: three passes + 16 low bits
    # every UMI = 4 fixed leading bases + 8 random ones: 65 536 possible keys share the top bits of one run
    umi = (np.uint32(0b00011011) << np.uint32(16)) | rng.integers(0, 1 << 16, n, dtype=np.uint32)
    idx = np.zeros(n, np.uint32)
    feature = np.full(n, 7, np.uint32)
    uq = np.full((n, 12), 70, np.uint8)
    d_idx, d_umi, d_uq, d_ft = c.upload(idx), c.upload(umi), c.upload(uq), c.upload(feature)
    recs = c.records(n, 12, d_idx, d_umi, d_uq, d_ft, None)
    keys = c.empty(n, np.uint64)
    nk = c.build_keys(recs, keys)
    assert nk == n
    cnt = c.count_keys(keys, nk)
    assert c.stat(1) == 1, "the long run did not reach the full-sort fallback"
    mol = cnt.molecules()
    # oracle: one barcode, one feature
    umi_ascii = E.unpack_seqs(umi, 12)
    _, uc = O.mark_dups_group(umi_ascii, np.ones(n, np.uint8), feature, utype=np.zeros(n, np.uint8),
                              qname=np.arange(n, dtype=np.uint64), umi_correction=True)
    order = np.argsort(uc["umi"], kind="stable")
    assert np.array_equal(mol["umi"], uc["umi"][order]) and np.array_equal(mol["read_count"], uc["read_count"][order])
    assert len(mol["umi"]) > 5_000
    c.close()


@pytest.mark.parametrize("levels", ["1", "2"])
@pytest.mark.parametrize("n", [300_000, 2_000_000])
def test_windowed_dupinfo_scatter_matches_the_oracle(n, levels, monkeypatch):
    """The experimental windowed scatter of the per-read records (CRGPU_DUPINFO_WINDOWED=1: k_per_read_sorted ->
    cr_partition_by_payload -> k_scatter_records; =2: two partition passes, windows of 2^(bits - 18) reads): every read's
    DupInfo equals the oracle's, as the direct path's does in the other tests."""
    import gpu_helpers as G
    from cellranger_amd import synth as S

    monkeypatch.setenv("CRGPU_DUPINFO_WINDOWED", levels)
    w = S.Workload(n_total=n, seed=S.SEED0 + 9, n_wl=100_000, n_cells=300, n_ambient=20000, n_genes=2000)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    r = w.host_reads(0, n)
    res, m = _compare_with_oracle(c, w, r, n, w.n_genes)
    assert m.nnz > 50_000
    c.close()


def test_targeted_panel_umi_filter_matches_the_oracle():
    """Targeted Gene Expression (mark_dups.rs:311-320): molecules of on-target features with fewer reads than
    targeted_umi_min_read_count yield no UmiCount and their reads carry is_filtered_target_umi.  Half of the features
    on target, threshold 3: DupInfo flags, matrix, molecule table and BarcodeSummary equal the oracle's; then the filter is
    switched off again and the plain results come back."""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import synth as S

    n = 400_000
    w = S.Workload(n_total=n, seed=S.SEED0 + 11, n_wl=50_000, n_cells=200, n_ambient=5000, n_genes=600, reads_per_umi=3)
    on_target = (np.arange(w.n_genes) % 2 == 0).astype(np.uint8)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    r = w.host_reads(0, n)
    try:
        c.set_target_filter(on_target, 3)
        O.set_target_filter(on_target, 3)
        res, m = _compare_with_oracle(c, w, r, n, w.n_genes)
        n_filtered = int((res.dupinfo["is_filtered_target"] != 0).sum())
        assert n_filtered > 10_000
        nnz_filtered = m.nnz
    finally:
        O.set_target_filter(None)
    c.set_target_filter(None)
    c.reset_counts()
    res2, m2 = _compare_with_oracle(c, w, r, n, w.n_genes)
    assert int((res2.dupinfo["is_filtered_target"] != 0).sum()) == 0 and m2.data.sum() > m.data.sum() and m2.nnz >= nnz_filtered
    c.close()


def test_per_read_umi_lengths_match_the_oracle():
    """Reads that end early carry shorter UMIs (UmiExtractor::extract_umi, cr_types/src/rna_read.rs:103-138; the
    reference's vectors :1581-1637 go through crgpu_pack_rows_var_dev first).  A UmiSeq is its bases AND its length: a
    10-base UMI never corrects onto a 12-base one and never shares a low-support group with it, even when the packed
    values coincide.  30 % of the reads get 10- or 11-base UMIs (short UMI space on purpose: many coincidences): every
    read's DupInfo, the matrix and the molecule table (in UmiCount order) equal the oracle's."""
    import json

    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S

    c = G.fresh_ctx()
    # 1. the reference's slicing vectors
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "misc_vectors.json")))["umi_extraction"]
    stride = 64
    seq = np.zeros((2, stride), np.uint8)
    qual = np.zeros((2, stride), np.uint8)
    for i, case in enumerate(g["cases"]):
        seq[i, :len(case["seq"])] = np.frombuffer(case["seq"].encode(), np.uint8)
        qual[i, :len(case["qual"])] = np.frombuffer(case["qual"].encode(), np.uint8)
    rl = np.array([len(x["seq"]) for x in g["cases"]] , np.uint32)
    d_pk, d_qn, d_len = c.empty(2, np.uint32), c.empty((2, 12), np.uint8), c.empty(2, np.uint8)
    c.pack_rows_var(c.upload(seq), c.upload(qual), c.upload(rl), 2, stride, g["offset"], g["length"], g["min_length"], d_pk, d_qn, d_len)
    lens, pk = d_len.to_host(), d_pk.to_host()
    for i, case in enumerate(g["cases"]):
        assert lens[i] == case["range_len"]
        assert E.unpack_seqs(pk[i:i + 1], int(lens[i]))[0].tobytes().decode() == case["umi"]
    # a read too short even for min_length has no UMI (check_range fails in the reference)
    c.pack_rows_var(c.upload(seq), c.upload(qual), c.upload(np.array([20, 25], np.uint32)), 2, stride, 16, 12, 10, d_pk, d_qn, d_len)
    assert list(d_len.to_host()) == [0, 0]

    # 2. parity of the count stage with mixed lengths
    n, UL = 300_000, 6
    w = S.Workload(n_total=n, seed=S.SEED0 + 13, n_wl=20_000, n_cells=120, n_ambient=3000, n_genes=200, umi_len=UL, umi_err=0.03,
                   reads_per_umi=3)
    r = w.host_reads(0, n)
    rng = np.random.default_rng(4)
    L = np.full(n, UL, np.uint8)
    L[rng.random(n) < 0.2] = UL - 2
    L[rng.random(n) < 0.1] = UL - 1
    umi = (r["umi"] >> (2 * (UL - L.astype(np.uint32)))).astype(np.uint32)      # keep the leading L bases
    uq = r["umi_qualn"].copy()
    uq[np.arange(UL)[None, :] >= L[:, None]] = 0
    c.set_whitelist(0, w.wl_packed, length=16)
    idx_a, idx_b, corr, dev = G.gpu_barcode_stage(c, r, n)
    c.set_key_layout(w.n_genes, UL, 1, 0)
    c.set_umi_min_len(UL - 2)
    d_umi, d_uq, d_ft, d_L = c.upload(umi), c.upload(uq), c.upload(r["feature"]), c.upload(L)
    recs = c.records(n, UL, dev["idx"], d_umi, d_uq, d_ft, dev["flags"], d_umi_len=d_L)
    d_pu, d_rc, d_fl = c.empty(n, np.uint32), c.empty(n, np.uint32), c.empty(n, np.uint8)
    counts = c.count_records(recs, d_pu, d_rc, d_fl)
    bc, ft, ct = counts.triplets()
    mol = counts.molecules()
    m = c.assemble_matrix(bc, ft, ct, w.n_genes)
    # oracle: ASCII UMIs padded with NUL bytes behind their real length
    reads = G.oracle_reads_from_packed(r, 16, UL)
    ascii_umi = np.zeros((n, UL), np.uint8)
    is_n = (uq & 0x80) != 0
    for k in range(UL):
        has = k < L
        shift = (2 * (L.astype(np.int64) - 1 - k)).clip(0)
        base = np.frombuffer(b"ACGT", np.uint8)[(umi.astype(np.int64) >> shift) & 3]
        ascii_umi[:, k] = np.where(has, np.where(is_n[:, k], ord("N"), base), 0)
    reads["umi"] = ascii_umi
    reads["umi_qual"] = (uq & 0x7F).astype(np.uint8)
    res = O.run_pipeline(reads, [O.Whitelist(E.unpack_seqs(w.wl_packed, 16))], n_threads=4, want_dupinfo=True)
    od = res.dupinfo
    fl, pu, rc = d_fl.to_host(), d_pu.to_host(), d_rc.to_host()
    has = od["has_dupinfo"] != 0
    assert np.array_equal((fl & 1) != 0, has)
    assert np.array_equal((fl & 2) != 0, od["is_corrected"] != 0) and np.array_equal((fl & 4) != 0, od["is_low_support"] != 0)
    assert np.array_equal((fl & 8) != 0, od["is_umi_count"] != 0)
    assert np.array_equal(pu[has], od["processed_umi"][has]) and np.array_equal(rc[has], od["read_count"][has])
    assert np.array_equal(m.barcodes_ascii(), res.barcodes) and np.array_equal(m.indptr, res.indptr)
    assert np.array_equal(m.indices, res.indices) and np.array_equal(m.data, res.data)
    _, canon_sorted = c.canon_order()
    col_rank = G.ranks_of(canon_sorted, res.barcodes)
    assert np.array_equal(mol["bc"], col_rank[res.mol_bc_col]) and np.array_equal(mol["feature"], res.mol["feature_idx"])
    assert np.array_equal(mol["umi"], res.mol["umi"]) and np.array_equal(mol["read_count"], res.mol["read_count"])
    assert int(((fl & 2) != 0).sum()) > 1000 and int(((fl & 4) != 0).sum()) > 10
    # the lengths really collide: some packed value occurs with two lengths under one (barcode, feature)
    c.close()
