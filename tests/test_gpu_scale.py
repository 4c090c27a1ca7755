"""Size-independent properties of the HIP path at sizes the oracle cannot check in seconds
(50 M reads here; bench.py runs 1 B): conservation laws, sortedness, determinism, batch-split
invariance, agreement between the two ways of driving the count stage."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 50_000_000


def _checksum(*arrays):
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).view(np.uint8), c)
    return c


@pytest.fixture(scope="module")
def big():
    import gpu_helpers as G
    from cellranger_amd import synth as S

    w = S.Workload(n_total=N, seed=S.SEED0 + 3)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    d = dict(n=N, umi_len=w.umi_len)
    d["cb"], d["cb_qualn"], d["flags"] = c.empty(N, np.uint32), c.empty((N, 16), np.uint8), c.empty(N, np.uint8)
    d["umi"], d["umi_qualn"], d["feature"] = c.empty(N, np.uint32), c.empty((N, 12), np.uint8), c.empty(N, np.uint32)
    d["idx"] = c.empty(N, np.uint32)
    c.synth(w, 0, N, cb=d["cb"].ptr, cb_qualn=d["cb_qualn"].ptr, umi=d["umi"].ptr, umi_qualn=d["umi_qualn"].ptr,
            feature=d["feature"].ptr, flags=d["flags"].ptr)
    yield c, w, d
    c.close()


def _run(c, d):
    from cellranger_amd._lib import COUNTS_CORRECTED, COUNTS_VALID

    c.reset_counts()
    c.match_and_count(d["cb"], d["flags"], d["n"], d["idx"])
    idx_a = d["idx"].to_host()
    c.correct(d["cb"], d["cb_qualn"], d["flags"], d["n"], d["idx"])
    idx_b = d["idx"].to_host()
    valid, corrected = c.get_counts(0, COUNTS_VALID), c.get_counts(0, COUNTS_CORRECTED)
    recs = c.records(d["n"], d["umi_len"], d["idx"], d["umi"], d["umi_qualn"], d["feature"], d["flags"])
    keys = c.empty(d["n"], np.uint64)
    nk = c.build_keys(recs, keys)
    counts = c.count_keys(keys, nk)
    bc, ft, ct = counts.triplets()
    mol = counts.molecules()
    m = c.assemble_matrix(bc, ft, ct, c.n_features)
    return dict(idx_a=idx_a, idx_b=idx_b, valid=valid, corrected=corrected, nk=nk, bc=bc, ft=ft, ct=ct, mol=mol, m=m, recs=recs)


def test_conservation_sortedness_and_determinism(big):
    from cellranger_amd._lib import MISS

    c, w, d = big
    r = _run(c, d)
    # pass A / pass B bookkeeping: histograms are exactly the per-read results
    hit_a = r["idx_a"] != MISS
    assert r["valid"].sum() == hit_a.sum()
    assert np.array_equal(np.bincount(r["idx_a"][hit_a], minlength=c.n_canon).astype(np.uint32), r["valid"])
    fixed = (r["idx_b"] != MISS) & ~hit_a
    assert r["corrected"].sum() == fixed.sum()
    assert np.array_equal(r["idx_b"][hit_a], r["idx_a"][hit_a])        # pass B never touches a valid read
    assert 0.90 < hit_a.mean() < 0.94 and 0.05 < fixed.mean() < 0.09   # the cfg2/cfg3 error model
    # every corrected barcode is a Hamming-1 neighbour of what was read
    _, canon_sorted = c.canon_order()
    cb = d["cb"].to_host()
    sel = np.nonzero(fixed)[0][:2_000_000]
    x = canon_sorted[r["idx_b"][sel]] ^ cb[sel]
    y = (x | (x >> np.uint32(1))) & np.uint32(0x55555555)
    n_diff = np.zeros(len(sel), np.int64)
    for k in range(16):
        n_diff += ((y >> np.uint32(2 * k)) & np.uint32(1)).astype(np.int64)
    has_n = (d["flags"].to_host()[sel] & 0x10) != 0
    assert ((n_diff == 1) | has_n).all() and (n_diff <= 1).all()
    # triplets: strictly sorted by (barcode, feature); counts positive; consistent with the molecule table
    key = r["bc"].astype(np.uint64) << np.uint64(32) | r["ft"].astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all() and (r["ct"] > 0).all()
    mol = r["mol"]
    assert r["ct"].sum() == len(mol["bc"])
    mkey = mol["bc"].astype(np.uint64) << np.uint64(32) | mol["feature"].astype(np.uint64)
    uk, cnt = np.unique(mkey, return_counts=True)
    assert np.array_equal(uk, key) and np.array_equal(cnt.astype(np.uint32), r["ct"])
    # reads are conserved: every key lands on exactly one (corrected) key; molecules exclude low support only
    assert mol["read_count"].sum() <= r["nk"] and mol["read_count"].sum() > 0.97 * r["nk"]
    assert (mol["read_count"] > 0).all()
    # CSC invariants (count_matrix.rs:382-448)
    m = r["m"]
    assert m.indptr[0] == 0 and m.indptr[-1] == m.nnz == len(r["ct"]) and (np.diff(m.indptr) >= 0).all()
    assert (np.diff(m.barcode_rank.astype(np.int64)) > 0).all()
    seen = (r["valid"] > 0) | (r["corrected"] > 0)
    assert np.array_equal(m.barcode_rank, np.nonzero(seen)[0].astype(np.uint32))    # BarcodeIndex
    assert int(m.data.sum()) == len(mol["bc"])
    # a second run gives bit-identical outputs (no order-dependent atomics leak into results)
    r2 = _run(c, d)
    assert _checksum(r["idx_b"], r["valid"], r["corrected"], r["bc"], r["ft"], r["ct"]) == \
        _checksum(r2["idx_b"], r2["valid"], r2["corrected"], r2["bc"], r2["ft"], r2["ct"])
    for f in ("bc", "lib", "feature", "umi", "read_count", "utype"):
        assert np.array_equal(r["mol"][f], r2["mol"][f]), f
    # the one-call entry point and the device CSC agree with the step-by-step path
    m1 = c.count(r2["recs"], c.n_features)
    keys2, nk2 = _rebuild_keys(c, r2["recs"], d["n"])
    counts2 = c.count_keys(keys2, nk2)  # keep alive: its device triplets feed the device CSC assembly
    md = c.assemble_matrix_dev(*counts2.triplets_dev(), counts2.n_triplets)
    rank, indptr, indices, data = md.download()
    for a, b in ((m1.indptr, m.indptr), (m1.indices, m.indices), (m1.data, m.data), (indptr, m.indptr), (indices, m.indices),
                 (data, m.data), (rank, m.barcode_rank)):
        assert np.array_equal(a, b)
    # per-read DupInfo and the per-barcode summary obey the conservation laws of mark_dups.rs / aligner.rs:54-67
    pu, rc, fl = c.empty(d["n"], np.uint32), c.empty(d["n"], np.uint32), c.empty(d["n"], np.uint8)
    counts3 = c.count_records(r2["recs"], pu, rc, fl)
    flags, reads_of = fl.to_host(), rc.to_host()
    has = (flags & 1) != 0
    assert int(has.sum()) == r["nk"]                                     # every key-forming read gets a DupInfo
    assert int(((flags & 8) != 0).sum()) == len(mol["bc"])               # one is_umi_count read per molecule
    assert not (flags[~has]).any() and not reads_of[~has].any()
    kept = has & ((flags & 4) == 0)
    assert int(kept.sum()) == int(mol["read_count"].sum())               # reads of kept molecules
    assert int(reads_of[(flags & 8) != 0].sum()) == int(mol["read_count"].sum())
    rows = counts3.barcode_summary()
    assert (rows["library"] == 0).all() and np.array_equal(rows["barcode_rank"], m.barcode_rank)
    assert np.array_equal(rows["reads"], (r["valid"].astype(np.uint64) + r["corrected"])[m.barcode_rank])
    assert int(rows["umis"].sum()) == len(mol["bc"]) and int(rows["candidate_dup_reads"].sum()) == int(kept.sum())
    assert int(rows["umi_corrected_reads"].sum()) == int(((flags & 2) != 0).sum())
    assert np.array_equal(rows["umis"], np.bincount(mol["bc"], minlength=c.n_canon)[m.barcode_rank].astype(np.uint64))
    idx_valid = r["idx_b"] != MISS
    assert np.array_equal(rows["umi_corrected_reads"],
                          np.bincount(r["idx_b"][idx_valid & ((flags & 2) != 0)], minlength=c.n_canon)[m.barcode_rank].astype(np.uint64))


def _rebuild_keys(c, recs, n):
    keys = c.empty(n, np.uint64)
    nk = c.build_keys(recs, keys)
    return keys, nk


def test_batch_split_invariance(big):
    """The ABI accumulates histograms over batches (the Rust host feeds FASTQ chunks): two half batches followed
    by pass B give the same indices and tables as one batch."""
    from cellranger_amd._lib import COUNTS_CORRECTED, COUNTS_VALID

    c, w, d = big
    n = 10_000_000
    h = n // 2
    c.reset_counts()
    c.match_and_count(d["cb"], d["flags"], n, d["idx"])
    c.correct(d["cb"], d["cb_qualn"], d["flags"], n, d["idx"])
    one = d["idx"].to_host(count=n)
    v1, c1 = c.get_counts(0, COUNTS_VALID), c.get_counts(0, COUNTS_CORRECTED)
    c.reset_counts()
    for off in (0, h):  # pass A over both halves first: the prior must be complete before pass B
        c.match_and_count(d["cb"].ptr + 4 * off, d["flags"].ptr + off, h, d["idx"].ptr + 4 * off)
    for off in (0, h):
        c.correct(d["cb"].ptr + 4 * off, d["cb_qualn"].ptr + 16 * off, d["flags"].ptr + off, h, d["idx"].ptr + 4 * off)
    assert np.array_equal(d["idx"].to_host(count=n), one)
    assert np.array_equal(c.get_counts(0, COUNTS_VALID), v1) and np.array_equal(c.get_counts(0, COUNTS_CORRECTED), c1)


def test_pack_metrics_and_feature_scans_at_scale(big):
    """The streaming kernels either side of the path (crgpu_pack_dev, crgpu_shard_metrics_dev, crgpu_match_features_dev)
    on grids far past one wave of workgroups: pack/unpack round trip, metrics against the oracle's per-read loop, and
    tiling invariance of the feature matcher (whole array == two halves == an oracle-checked sample)."""
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd._lib import MISS, NO_FEATURE

    c, w, d = big
    n = 20_000_000
    cb, cbq = d["cb"].to_host(count=n), d["cb_qualn"].to_host(count=n * 16).reshape(n, 16)
    umi, uq = d["umi"].to_host(count=n), d["umi_qualn"].to_host(count=n * 12).reshape(n, 12)
    cb_a, cbq_a = S.to_ascii(cb, cbq, 16)
    umi_a, uq_a = S.to_ascii(umi, uq, 12)
    # pack: ASCII + plain qualities -> the packed arrays the generator made (N bases pack as A with bit 7 set)
    d_pk, d_qn, d_fl = c.empty(n, np.uint32), c.empty((n, 16), np.uint8), c.zeros(n, np.uint8)
    c.pack(c.upload(cb_a), c.upload(cbq_a), n, 16, d_pk, d_qn, d_fl)
    is_n = (cbq & 0x80) != 0
    assert np.array_equal(d_qn.to_host().reshape(n, 16), cbq)
    got_pk = d_pk.to_host()
    clean = ~is_n.any(axis=1)
    assert np.array_equal(got_pk[clean], cb[clean]) and clean.mean() > 0.98
    assert np.array_equal((d_fl.to_host() & 0x10) != 0, ~clean)
    # metrics
    c.reset_counts()
    d_idx = c.empty(n, np.uint32)
    c.match_and_count(d["cb"], d["flags"], n, d_idx)
    idx_a = d_idx.to_host()
    got = c.shard_metrics(d["cb"], d["cb_qualn"], 16, d["umi"], d["umi_qualn"], 12, d_idx, n)
    exp = O.shard_metrics(cb_a, cbq_a, umi_a, uq_a, exact_hit=(idx_a != MISS).astype(np.uint8))
    assert got == exp and got["sequenced_reads"] == n
    # feature matcher over 20 M captures (the UMI column stands in for 12-base captures)
    rng = np.random.default_rng(9)
    feats = np.unique(umi[:4000])[:300]
    feat_ascii = E.unpack_seqs(feats, 12)
    dist = O.compute_feature_dist(rng.integers(1, 1000, len(feats)), np.zeros(len(feats), np.uint32))
    index = np.arange(len(feats), dtype=np.uint32)
    c.set_feature_pattern(0, feat_ascii, index, dist)
    out = c.empty(n, np.uint32)
    c.match_features(0, d["umi"], d["umi_qualn"], n, out)
    whole = out.to_host()
    h = n // 2 + 12_345
    out2 = c.empty(n, np.uint32)
    c.match_features(0, d["umi"], d["umi_qualn"], h, out2)
    first = out2.to_host(count=h)
    assert np.array_equal(first, whole[:h])
    hits = np.nonzero(whole != NO_FEATURE)[0]
    assert len(hits) > 100
    sample = np.concatenate([hits[:2000], hits[-2000:], rng.integers(0, n, 4000)])
    for i in sample:
        f = O.find_closest_feature(feat_ascii, dist, bytes(umi_a[i]), bytes(uq_a[i]))
        assert whole[i] == (NO_FEATURE if f < 0 else index[f]), i


@pytest.mark.parametrize("n,model", [
    (50_000_000, {}), (1_000_000_000, {}),
    # other key widths at a size where every pass of the sort runs thousands of chunks: 48, 43 and 53 bits
    (30_000_000, dict(n_wl=100_000, n_cells=3000, n_ambient=20_000, n_genes=1000, umi_len=10)),
    (30_000_000, dict(n_wl=5000, n_cells=300, n_ambient=2000, n_genes=17, umi_len=12)),
    (30_000_000, dict(umi_len=8)),
    # three libraries (BASELINE configs[3] has two library types): per-library histograms, K1 / K2 without the
    # one-library fast paths, library bits inside the keys (60 bits)
    (40_000_000, dict(n_libs=3, n_genes=5000)),
])
def test_device_side_properties_up_to_the_full_1b_workload(n, model):
    """BASELINE configs[2] at its full size (1 B records on one GPU): the laws of cellranger_amd/selfcheck.py, checked on
    the device, and a second pass gives the same checksum of the whole triplet table."""
    import gpu_helpers as G
    from cellranger_amd import selfcheck
    from cellranger_amd import synth as S

    w = S.Workload(n_total=n, seed=S.SEED0 + 3, **model)
    U = w.umi_len
    c = G.fresh_ctx()
    libs = tuple(range(w.n_libs))
    for lib in libs:
        c.set_whitelist(lib, w.wl_packed, length=16)
    c.set_key_layout(w.n_genes, w.umi_len, w.n_libs, 0)
    d = dict(n=n, umi_len=w.umi_len)
    d["cb"], d["cb_qualn"], d["flags"] = c.empty(n, np.uint32), c.empty((n, 16), np.uint8), c.empty(n, np.uint8)
    d["umi"], d["umi_qualn"], d["feature"] = c.empty(n, np.uint32), c.empty((n, U), np.uint8), c.empty(n, np.uint32)
    d["idx"], d["keys"] = c.empty(n, np.uint32), c.empty(n, np.uint64)
    chunk = 1 << 27
    for off in range(0, n, chunk):
        m = min(chunk, n - off)
        c.synth(w, off, m, cb=d["cb"].ptr + 4 * off, cb_qualn=d["cb_qualn"].ptr + 16 * off, umi=d["umi"].ptr + 4 * off,
                umi_qualn=d["umi_qualn"].ptr + U * off, feature=d["feature"].ptr + 4 * off, flags=d["flags"].ptr + off)
    a = selfcheck.full_size_properties(c, d, libs=libs)
    b = selfcheck.full_size_properties(c, d, libs=libs)
    assert a == b
    assert 0.90 * n < a["valid_reads"] < 0.94 * n and 0.05 * n < a["corrected_reads"] < 0.09 * n   # SURVEY 8d error model
    assert a["molecules"] > 0.2 * a["keys"] and a["matrix_nnz"] == a["triplets"]
    c.close()


def test_barcode_stage_on_more_than_2_31_reads():
    """2.5 G reads in ONE batch (crgpu_match_and_count_dev / crgpu_correct_dev accept up to 2^32 - 2): read ordinals pass
    2^31 and byte offsets into the quality rows pass 2^35, so any 32-bit index arithmetic in K1 / K2 shows up as a broken
    histogram or a correction that is not a Hamming-1 neighbour."""
    import gpu_helpers as G
    from cellranger_amd import selfcheck
    from cellranger_amd import synth as S

    n = 2_500_000_000
    w = S.Workload(n_total=n, seed=S.SEED0 + 2, n_genes=0)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    d = dict(n=n)
    d["cb"], d["cb_qualn"], d["flags"], d["idx"] = c.empty(n, np.uint32), c.empty((n, 16), np.uint8), c.empty(n, np.uint8), c.empty(n, np.uint32)
    chunk = 1 << 27
    for off in range(0, n, chunk):
        m = min(chunk, n - off)
        c.synth(w, off, m, cb=d["cb"].ptr + 4 * off, cb_qualn=d["cb_qualn"].ptr + 16 * off, flags=d["flags"].ptr + off)
    a = selfcheck.barcode_stage_properties(c, d)
    assert 0.90 * n < a["valid_reads"] < 0.94 * n and 0.05 * n < a["corrected_reads"] < 0.09 * n
    c.close()


def test_bench_code_path_at_20m_reads_read_by_read():
    """The code path the bench times -- K1's LDS table chosen by itself (>= 16 Mi reads per call), K1's miss records feeding
    K2, the staged CORRECTED histogram, the key histograms counted by k_build_keys, the onesweep sort over thousands of
    chunk tickets -- checked against the oracle READ BY READ at 20 M reads of the cfg3 model (not only through properties):
    every read's barcode index and DupInfo, both histograms, the matrix, the molecule table, the BarcodeSummary rows.
    (scripts/parity_large.py does the same at 100 M reads outside the suite.)"""
    import gpu_helpers as G
    import test_gpu_count as T
    from cellranger_amd import synth as S

    n = 20_000_000
    w = S.Workload(n_total=n, seed=S.SEED0 + 3)
    c = G.fresh_ctx()        # by-products trusted, as in bench.py
    c.set_whitelist(0, w.wl_packed, length=16)
    r = w.host_reads(0, n)
    res, m = T._compare_with_oracle(c, w, r, n, w.n_genes)
    assert m.nnz > 5_000_000 and len(res.mol) > 8_000_000
    c.close()
