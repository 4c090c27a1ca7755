"""GPU parity tests of the barcode stage (K1 exact match + histogram, K2 posterior correction),
called through the C ABI and compared bit-for-bit with the oracle and the reference's golden vectors."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_context_is_gfx950_and_loaded_in_tree():
    import gpu_helpers as G
    from cellranger_amd import _lib

    c = G.ctx()
    assert c.h
    assert os.path.dirname(_lib.LIB_PATH).endswith("cellranger_amd")


@pytest.mark.parametrize("block", load("corrector_vectors.json")["posterior"], ids=lambda b: b["name"][:40])
def test_posterior_golden_vectors_on_gpu(block):
    """corrector.rs:196-313 through crgpu_match_and_count / crgpu_correct (host-buffer ABI)."""
    import gpu_helpers as G
    from cellranger_amd import engine as E
    from cellranger_amd._lib import COUNTS_PRIOR, MISS

    c = G.fresh_ctx()
    wl = block["whitelist"]
    c.set_whitelist_ascii(0, wl)
    order, seqs = c.canon_order()
    sorted_wl = [bytes(r).decode() for r in E.unpack_seqs(seqs, 5)]
    assert sorted_wl == sorted(wl)
    prior = np.zeros(len(wl), np.uint32)
    for k, v in block["bc_counts"].items():
        prior[sorted_wl.index(k)] = v
    c.set_counts(0, COUNTS_PRIOR, prior)
    c.set_posterior(block["max_expected_barcode_errors"], block["bc_confidence_threshold"])
    seqs_in = [case["seq"] for case in block["cases"]]
    quals = np.array([case["qual"] for case in block["cases"]], dtype=np.uint8)
    # the reference test feeds these as Invalid segments: start from MISS for all of them
    idx = np.full(len(seqs_in), MISS, np.uint32)
    idx2, flag = c.correct_host(0, seqs_in, quals, idx)
    for case, i, f in zip(block["cases"], idx2, flag):
        if case["expect"] is None:
            assert i == MISS and f == 0, case
        else:
            assert i != MISS and sorted_wl[i] == case["expect"] and f == 1, case
    c.close()


def test_posterior_n_any_position_on_gpu():
    import gpu_helpers as G
    from cellranger_amd._lib import MISS

    g = load("corrector_vectors.json")["prop_n_in_barcode"]
    c = G.fresh_ctx()
    c.set_whitelist_ascii(0, g["whitelist"])
    bc = g["whitelist"][0]
    seqs, quals = [], []
    for n_pos in range(16):
        seqs.append(bc[:n_pos] + "N" + bc[n_pos + 1:])
        q = [g["qual_default"]] * 16
        q[n_pos] = g["qual_at_n"]
        quals.append(q)
    quals = np.array(quals, np.uint8)
    for params in [(g["max_expected_barcode_errors"], g["bc_confidence_threshold"]), None]:
        if params:
            c.set_posterior(*params)
        else:
            c.set_posterior(np.finfo(np.float64).max, 0.975)  # Posterior::default (corrector.rs:102-108)
        idx = c.match_and_count_host(0, seqs, quals)
        assert (idx == MISS).all()  # an N never matches exactly (whitelist.rs:494)
        idx2, flag = c.correct_host(0, seqs, quals, idx)
        assert (idx2 == 0).all() and (flag == 1).all()
    # two Ns can never be repaired
    idx2, flag = c.correct_host(0, ["NNGATTGACCCAAAGG"], np.full((1, 16), 53, np.uint8), np.array([MISS], np.uint32))
    assert idx2[0] == MISS and flag[0] == 0
    c.close()


def _compare_barcode_stage(w, n, first=0, threshold=0.975, max_err=None):
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd._lib import COUNTS_CORRECTED, COUNTS_VALID

    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=w.cb_len)
    max_err = np.finfo(np.float64).max if max_err is None else max_err
    c.set_posterior(max_err, threshold)
    _, canon_sorted = c.canon_order()
    r = w.host_reads(first, n)
    idx_a, idx_b, corr, _ = G.gpu_barcode_stage(c, r, n)

    owl = O.Whitelist(E.unpack_seqs(w.wl_packed, w.cb_len))
    res = O.run_pipeline(G.oracle_reads_from_packed(r, w.cb_len, w.umi_len), [owl], count=False,
                         max_expected_errors=max_err, threshold=threshold, n_threads=4)
    exp_a, exp_b = G.oracle_expected_idx(res, canon_sorted)
    assert np.array_equal(idx_a, exp_a)
    assert np.array_equal(idx_b, exp_b)
    assert np.array_equal(corr, (res.bc_state == 2).astype(np.uint8))
    assert np.array_equal(c.get_counts(0, COUNTS_VALID), G.hist_as_rank_counts(res.valid_hist[0], w.cb_len, canon_sorted))
    assert np.array_equal(c.get_counts(0, COUNTS_CORRECTED),
                          G.hist_as_rank_counts(res.corrected_hist[0], w.cb_len, canon_sorted))
    stats = dict(valid=int((res.bc_state == 1).sum()), corrected=int((res.bc_state == 2).sum()),
                 invalid=int((res.bc_state == 0).sum()))
    # BARCODE_CORRECTION's join outputs (barcode_correction.rs:372-448) from the same tables
    v = G.hist_as_rank_counts(res.valid_hist[0], w.cb_len, canon_sorted).astype(np.uint64)
    cc = G.hist_as_rank_counts(res.corrected_hist[0], w.cb_len, canon_sorted).astype(np.uint64)
    m = c.barcode_correction_metrics(0)
    assert m["valid_reads"] == stats["valid"] and m["corrected_reads"] == stats["corrected"]
    t = v + cc
    assert m["barcodes_detected"] == int((t > 0).sum())
    exp_div = float(int(t.sum())) ** 2 / float(sum(int(x) * int(x) for x in t[t > 0]))   # inverse Simpson index
    assert abs(m["effective_barcode_diversity"] - exp_div) <= 1e-12 * exp_div
    for thr in (1, 25, 1000):
        ranks, counts = c.total_barcode_counts(thr)
        exp = np.where(v >= thr, v, 0) + np.where(cc >= thr, cc, 0)
        assert np.array_equal(ranks, np.nonzero(exp)[0].astype(np.uint32)) and np.array_equal(counts, exp[exp > 0])
    stats["k1_split_rounds"] = c.stat(3)  # CRGPU_STAT_K1_SPLIT_ROUNDS
    c.close()
    return stats


def test_cfg2_model_1m_reads_bit_exact():
    """1 M-read down-scale of cfg2: idx per read and both histograms equal the oracle's."""
    from cellranger_amd import synth as S

    w = S.Workload(n_total=1_000_000, seed=S.SEED0 + 2)
    st = _compare_barcode_stage(w, 1_000_000)
    assert st["corrected"] > 20_000 and st["invalid"] > 1_000 and st["valid"] > 800_000


def test_hot_barcode_table_path_bit_exact(monkeypatch):
    """K1's LDS table of the most frequent barcodes (used from 16 M reads per call) is only a cache: forced on at
    1 M reads, per-read indices and both histograms still equal the oracle's."""
    from cellranger_amd import synth as S

    monkeypatch.setenv("CRGPU_HOT_MIN_READS", "1")
    w = S.Workload(n_total=1_000_000, seed=S.SEED0 + 5)
    st = _compare_barcode_stage(w, 1_000_000)
    assert st["corrected"] > 20_000 and st["valid"] > 800_000
    # few cells on a tiny whitelist: nearly every read is answered by the table; TTTT...T (all ones) is a valid key
    w = S.Workload(n_total=200_000, seed=31, n_wl=64, n_cells=20, n_ambient=30, cb_len=4, umi_len=6, cb_err=0.05)
    _compare_barcode_stage(w, 200_000)


def test_split_histogram_rounds_and_their_overflow_fallback(monkeypatch):
    """With the LDS table, pass A counts table hits per table slot in LDS and stages only the other hits (per-wave regions);
    a region that overflows makes the round fall back to device atomics.  Both ways the histograms equal the oracle's, as
    does the full staging of round 1 (CRGPU_K1_FULL_STAGING=1)."""
    from cellranger_amd import synth as S

    monkeypatch.setenv("CRGPU_HOT_MIN_READS", "1")
    monkeypatch.setenv("CRGPU_K1_SPLIT", "1")   # the default takes it only for whitelists of more than 31 x 32768 barcodes
    w = S.Workload(n_total=600_000, seed=S.SEED0 + 9)
    assert _compare_barcode_stage(w, 600_000)["k1_split_rounds"] >= 1
    monkeypatch.setenv("CRGPU_COLD_CAP", "5")
    assert _compare_barcode_stage(w, 600_000)["k1_split_rounds"] >= 1   # overflow: counted by the atomics fallback
    monkeypatch.delenv("CRGPU_COLD_CAP")
    monkeypatch.setenv("CRGPU_K1_SPLIT", "0")
    assert _compare_barcode_stage(w, 600_000)["k1_split_rounds"] == 0


@pytest.mark.parametrize("case", ["plain", "cold_regions_overflow", "3m_list", "tiny_list_all_T"])
def test_table_hits_counted_inside_the_lookup_kernel(case, monkeypatch):
    """CRGPU_K1_MODE=count: the LDS table holds 4-byte keys and one counter per slot; a hit the table answers bumps its
    counter in the lookup kernel and takes its rank from the table image, the other hits go to per-wave regions and the
    staged histogram (a full region counts its surplus by device atomics).  Indices and both histograms equal the oracle's;
    the all-T barcode (the value of an empty slot) is never cached and still counted."""
    from cellranger_amd import synth as S

    monkeypatch.setenv("CRGPU_HOT_MIN_READS", "1")
    monkeypatch.setenv("CRGPU_K1_MODE", "count")
    if case == "plain":
        assert _compare_barcode_stage(S.Workload(n_total=600_000, seed=S.SEED0 + 19), 600_000)["k1_split_rounds"] >= 1
    elif case == "cold_regions_overflow":
        monkeypatch.setenv("CRGPU_COLD_CAP", "5")
        assert _compare_barcode_stage(S.Workload(n_total=600_000, seed=S.SEED0 + 20), 600_000)["k1_split_rounds"] >= 1
    elif case == "3m_list":
        st = _compare_barcode_stage(S.Workload(n_total=1_000_000, seed=S.SEED0 + 21, n_wl=6_794_880), 1_000_000)
        assert st["k1_split_rounds"] >= 1 and st["corrected"] > 20_000
    else:
        # 4-base barcodes, the whole space listed: TTTT = 0xFF...; few cells, so nearly every read is answered by the table
        w = S.Workload(n_total=200_000, seed=33, n_wl=256, n_cells=40, n_ambient=100, cb_len=4, umi_len=6, cb_err=0.05)
        _compare_barcode_stage(w, 200_000)


def test_miss_record_overflow_falls_back_to_the_scan(monkeypatch):
    """Pass A leaves compact records of its misses for pass B; when a wave's region overflows, pass B scans idx as it
    does for any other call sequence -- same indices, flags and histograms."""
    from cellranger_amd import synth as S

    monkeypatch.setenv("CRGPU_HOT_MIN_READS", "1")
    monkeypatch.setenv("CRGPU_MISS_RECORD_CAP", "3")
    w = S.Workload(n_total=400_000, seed=S.SEED0 + 6)
    st = _compare_barcode_stage(w, 400_000)
    assert st["corrected"] > 8_000


def test_dense_small_whitelists_many_neighbours_and_ties():
    """Short barcodes with a dense whitelist: several Hamming-1 neighbours per read, equal priors,
    thresholds that accept ties' winners -- exercises the f64 accumulation order and tie-breaks."""
    from cellranger_amd import synth as S

    for cb_len, n_wl, thr in [(5, 300, 0.5), (6, 1500, 0.3), (8, 20000, 0.6), (11, 100000, 0.9), (16, 5000, 0.975)]:
        w = S.Workload(n_total=200_000, seed=77 + cb_len, n_wl=n_wl, n_cells=min(200, n_wl // 3),
                       n_ambient=n_wl // 2, cb_len=cb_len, cb_err=0.03, n_rate=0.004)
        st = _compare_barcode_stage(w, 200_000, threshold=thr)
        assert st["corrected"] > 0
        # the reference's unit-test parameters (expected-error veto active)
        _compare_barcode_stage(w, 50_000, first=200_000, threshold=0.95, max_err=1.0)


def test_translation_whitelist_on_gpu():
    """Whitelist::Trans (whitelist.rs:497-504): hits report the partner's rank and the prior is
    looked up by the translated sequence (corrector.rs:135-137)."""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd._lib import COUNTS_VALID

    w = S.Workload(n_total=300_000, seed=5, n_wl=50_000, n_cells=2000, n_ambient=20000)
    rng = np.random.default_rng(9)
    canon = w.wl_packed                       # GEX list = canonical space
    raw = np.unique(rng.integers(0, 1 << 32, size=60_000, dtype=np.uint64))[:50_000].astype(np.uint32)
    raw = rng.permutation(raw)
    translate_to = rng.permutation(50_000).astype(np.uint32)   # pairing permutation
    c = G.fresh_ctx()
    c.set_whitelist(0, canon, length=16)
    c.set_whitelist(1, raw, canon=canon, translate_to=translate_to, length=16)
    _, canon_sorted = c.canon_order()
    # reads: library 1 reads carry RAW barcodes; build them by swapping the generator's whitelist
    w_fb = S.Workload(n_total=300_000, seed=5, n_wl=50_000, n_cells=2000, n_ambient=20000)
    w_fb.wl_packed[:] = raw
    r0 = w.host_reads(0, 150_000)
    r1 = w_fb.host_reads(150_000, 150_000)
    r = {k: np.concatenate([r0[k], r1[k]]) for k in r0}
    r["flags"][150_000:] |= 1
    idx_a, idx_b, corr, _ = G.gpu_barcode_stage(c, r, 300_000)
    owl0 = O.Whitelist(E.unpack_seqs(canon, 16))
    owl1 = O.Whitelist(E.unpack_seqs(raw, 16), translated=E.unpack_seqs(canon[translate_to], 16))
    res = O.run_pipeline(G.oracle_reads_from_packed(r, 16, 12), [owl0, owl1], n_lib=2, count=False, n_threads=4)
    exp_a, exp_b = G.oracle_expected_idx(res, canon_sorted)
    assert np.array_equal(idx_a, exp_a) and np.array_equal(idx_b, exp_b)
    for lib in (0, 1):
        assert np.array_equal(c.get_counts(lib, COUNTS_VALID),
                              G.hist_as_rank_counts(res.valid_hist[lib], 16, canon_sorted))
    assert (res.bc_state[150_000:] == 2).sum() > 1000
    c.close()


def test_pack_and_host_abi_match_device_abi():
    """crgpu_pack_dev + host-buffer entry points give the same answers as the packed device path."""
    import gpu_helpers as G
    from cellranger_amd import synth as S
    from cellranger_amd._lib import FLAG_CB_HAS_N

    w = S.Workload(n_total=100_000, seed=3, n_wl=40_000, n_cells=1000, n_ambient=10000)
    n = 100_000
    r = w.host_reads(0, n)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    seq_ascii, qual = S.to_ascii(r["cb"], r["cb_qualn"], 16)
    # device pack
    d_seq, d_qual = c.upload(seq_ascii), c.upload(qual)
    d_pk, d_qn, d_fl = c.empty(n, np.uint32), c.empty((n, 16), np.uint8), c.zeros(n, np.uint8)
    c.pack(d_seq, d_qual, n, 16, d_pk, d_qn, d_fl)
    assert np.array_equal(d_pk.to_host(), r["cb"])
    assert np.array_equal(d_qn.to_host(), r["cb_qualn"])
    assert np.array_equal(d_fl.to_host() & FLAG_CB_HAS_N, r["flags"] & FLAG_CB_HAS_N)
    # the same from whole R1 rows (16 barcode bases + 12 UMI bases + 3 bytes of anything, an odd stride on purpose)
    umi_ascii, umi_qual = S.to_ascii(r["umi"], r["umi_qualn"], 12)
    pad = np.full((n, 3), ord("G"), np.uint8)
    rows_s, rows_q = np.hstack([seq_ascii, umi_ascii, pad]), np.hstack([qual, umi_qual, pad])
    d_rs, d_rq = c.upload(rows_s), c.upload(rows_q)
    d_pk2, d_qn2, d_fl2 = c.empty(n, np.uint32), c.empty((n, 16), np.uint8), c.zeros(n, np.uint8)
    c.pack_rows(d_rs, d_rq, n, 31, 0, 16, d_pk2, d_qn2, d_fl2)
    assert np.array_equal(d_pk2.to_host(), r["cb"]) and np.array_equal(d_qn2.to_host(), r["cb_qualn"])
    assert np.array_equal(d_fl2.to_host(), d_fl.to_host())
    d_um, d_uq = c.empty(n, np.uint32), c.empty((n, 12), np.uint8)
    c.pack_rows(d_rs, d_rq, n, 31, 16, 12, d_um, d_uq)
    assert np.array_equal(d_um.to_host(), r["umi"]) and np.array_equal(d_uq.to_host(), r["umi_qualn"])
    with pytest.raises(Exception, match="outside a row"):
        c.pack_rows(d_rs, d_rq, n, 31, 20, 12, d_um, d_uq)
    idx_a, idx_b, corr, _ = G.gpu_barcode_stage(c, r, n)
    c.reset_counts()
    h_a = c.match_and_count_host(0, seq_ascii, qual)
    h_b, h_f = c.correct_host(0, seq_ascii, qual, h_a)
    assert np.array_equal(h_a, idx_a) and np.array_equal(h_b, idx_b) and np.array_equal(h_f, corr)
    c.close()


def test_empty_and_error_paths():
    import gpu_helpers as G
    from cellranger_amd._lib import CrgpuError

    c = G.fresh_ctx()
    with pytest.raises(CrgpuError):
        c.match_and_count(0, None, 10, 0)          # no whitelist yet
    with pytest.raises(CrgpuError):
        c.set_whitelist_ascii(0, ["ACGTN"])         # non-ACGT whitelist entry
    with pytest.raises(CrgpuError):
        c.set_whitelist_ascii(0, ["A" * 17])        # > 16 bases
    c.set_whitelist_ascii(0, ["ACGT", "TTTT"])
    c.match_and_count(None, None, 0, None)          # empty batch is a no-op
    c.correct(None, None, None, 0, None)
    assert c.get_counts(0).sum() == 0
    c.close()


@pytest.mark.parametrize("hot", [False, True])
def test_one_library_per_call_in_a_three_library_well(monkeypatch, hot):
    """MAKE_SHARD / BARCODE_CORRECTION hand over one library per call.  With three whitelists set, a call whose reads all
    carry one library id takes the one-library kernels (with `hot`: the LDS table and the miss records) with that
    library's tables; pass A runs for every library before pass B, so B's first two calls scan idx and the last one
    reads A's records.  A mixed call afterwards takes the general kernels.  All equal to the oracle."""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd._lib import COUNTS_CORRECTED, COUNTS_VALID

    if hot:
        monkeypatch.setenv("CRGPU_HOT_MIN_READS", "1")
    n = 600_000
    w = S.Workload(n_total=n, seed=S.SEED0 + 9, n_wl=20_000, n_cells=300, n_ambient=2000, n_libs=3)
    r = w.host_reads(0, n)
    lib_of = r["flags"] & 0x0F
    order = np.argsort(lib_of, kind="stable")                  # the reads library by library
    r = {k: np.ascontiguousarray(v[order]) for k, v in r.items()}
    lib_of = lib_of[order]
    c = G.fresh_ctx()
    for lib in range(3):
        c.set_whitelist(lib, w.wl_packed, length=16)
    _, canon_sorted = c.canon_order()
    bounds = np.searchsorted(lib_of, np.arange(4))
    dev = []
    for lib in range(3):                                        # pass A, one call per library
        a, b = bounds[lib], bounds[lib + 1]
        d = dict(cb=c.upload(r["cb"][a:b]), cbq=c.upload(r["cb_qualn"][a:b]), fl=c.upload(r["flags"][a:b]),
                 idx=c.empty(b - a, np.uint32), corr=c.empty(b - a, np.uint8))
        c.match_and_count(d["cb"], d["fl"], b - a, d["idx"])
        dev.append(d)
    idx_a = np.concatenate([d["idx"].to_host() for d in dev])
    for lib in range(3):                                        # pass B
        d = dev[lib]
        c.correct(d["cb"], d["cbq"], d["fl"], bounds[lib + 1] - bounds[lib], d["idx"], d["corr"])
    idx_b = np.concatenate([d["idx"].to_host() for d in dev])
    corr = np.concatenate([d["corr"].to_host() for d in dev])

    owl = O.Whitelist(E.unpack_seqs(w.wl_packed, 16))
    res = O.run_pipeline(G.oracle_reads_from_packed(r, 16, w.umi_len), [owl] * 3, count=False, n_threads=4)
    exp_a, exp_b = G.oracle_expected_idx(res, canon_sorted)
    assert np.array_equal(idx_a, exp_a) and np.array_equal(idx_b, exp_b)
    assert np.array_equal(corr, (res.bc_state == 2).astype(np.uint8))
    for lib in range(3):
        assert np.array_equal(c.get_counts(lib, COUNTS_VALID), G.hist_as_rank_counts(res.valid_hist[lib], 16, canon_sorted))
        assert np.array_equal(c.get_counts(lib, COUNTS_CORRECTED),
                              G.hist_as_rank_counts(res.corrected_hist[lib], 16, canon_sorted))
    assert (res.bc_state == 2).sum() > 10_000 and min(np.diff(bounds)) > 100_000
    # the same reads in ONE mixed call: the general kernels, same answers
    c.reset_counts()
    ia, ib, cr, _ = G.gpu_barcode_stage(c, r, n)
    assert np.array_equal(ia, exp_a) and np.array_equal(ib, exp_b) and np.array_equal(cr, corr)
    c.close()


@pytest.mark.parametrize("mode", ["default_external_write", "trusted_write_through_context", "trusted_then_invalidate"])
def test_kept_by_products_never_describe_rewritten_buffers(mode, monkeypatch):
    """The two-pass flow of the reference (all MAKE_SHARD batches first so that the prior is complete, then all
    BARCODE_CORRECTION batches) re-uses fixed device buffers: pass A runs on batch A, the SAME buffers are refilled with
    batch B, pass B runs.  K1's miss records describe batch A and must not be used:
      default_external_write         the option is off: the records are never kept, whoever writes the buffers;
      trusted_write_through_context  the option is on and the refill goes through crgpu_memcpy_h2d, which drops them;
      trusted_then_invalidate        the option is on, the refill bypasses the context, the host calls crgpu_invalidate."""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd._lib import MISS

    monkeypatch.setenv("CRGPU_HOT_MIN_READS", "1")  # the table lookup (and with it the records) at a test-sized batch
    n = 400_000
    w = S.Workload(n_total=2 * n, seed=77, n_wl=50_000, n_cells=500, n_ambient=5000)
    ra, rb = w.host_reads(0, n), w.host_reads(n, n)
    c = G.fresh_ctx(trust=(mode != "default_external_write"))
    other = G.fresh_ctx(trust=False)  # a second context stands in for "somebody else writes the buffers" (a torch copy, RCCL)
    c.set_whitelist(0, w.wl_packed, length=16)
    _, canon_sorted = c.canon_order()
    d_cb, d_cbq, d_fl = c.upload(ra["cb"]), c.upload(ra["cb_qualn"]), c.upload(ra["flags"])
    d_idx = c.empty(n, np.uint32)
    c.match_and_count(d_cb, d_fl, n, d_idx)          # pass A on batch A (its counts stay: the prior covers A)
    c.synchronize()
    writer = c if mode == "trusted_write_through_context" else other
    for dst, src in ((d_cb, rb["cb"]), (d_cbq, rb["cb_qualn"]), (d_fl, rb["flags"]), (d_idx, np.full(n, MISS, np.uint32))):
        a = np.ascontiguousarray(src)
        writer._check(writer.L.crgpu_memcpy_h2d(writer.h, dst.ptr, a.ctypes.data, a.nbytes))
    other.synchronize()
    if mode == "trusted_then_invalidate":
        c.invalidate()
    c.correct(d_cb, d_cbq, d_fl, n, d_idx)           # pass B on batch B, every read a MISS to start from
    got = d_idx.to_host()
    # oracle: the prior is batch A's valid histogram; every read of B goes through the corrector
    owl = O.Whitelist(E.unpack_seqs(w.wl_packed, 16))
    res_a = O.run_pipeline(G.oracle_reads_from_packed(ra, 16, 12), [owl], count=False, n_threads=4)
    seq_b, qual_b = S.to_ascii(rb["cb"], rb["cb_qualn"], 16)
    exp = np.full(n, MISS, np.uint32)
    for i in range(0, n, 37):  # a sample of the reads, through the scalar oracle call
        fixed = O.posterior_correct(owl, res_a.valid_hist[0], bytes(seq_b[i]), qual_b[i])
        if fixed is not None:
            exp[i] = G.ranks_of(canon_sorted, np.frombuffer(fixed, np.uint8).reshape(1, 16))[0]
    sample = np.arange(0, n, 37)
    assert np.array_equal(got[sample], exp[sample])
    assert (got[sample] != MISS).sum() > 100
    c.close()
    other.close()


def test_n_barcode_packed_without_flags_is_refused_not_matched():
    """an N packs as code 0 ('A'); without a flags array nothing marks the read, and pass A would look the
    barcode up as if N were A.  The pack kernels remember that they met an N without flags and the flag-less
    crgpu_match_and_count_dev that follows fails with CRGPU_EINVAL."""
    import gpu_helpers as G
    from cellranger_amd._lib import CrgpuError

    c = G.fresh_ctx()
    wl = ["AAAAAAAAAAAAAAAA", "ACGTACGTACGTACGT"]
    c.set_whitelist_ascii(0, wl)
    seq = np.frombuffer(b"NAAAAAAAAAAAAAAA" + b"ACGTACGTACGTACGT", np.uint8).reshape(2, 16).copy()
    qual = np.full((2, 16), 70, np.uint8)
    d_seq, d_qual = c.upload(seq), c.upload(qual)
    d_pk, d_qn, d_idx = c.empty(2, np.uint32), c.empty((2, 16), np.uint8), c.empty(2, np.uint32)
    c.pack(d_seq, d_qual, 2, 16, d_pk, d_qn, None)
    with pytest.raises(CrgpuError) as ei:
        c.match_and_count(d_pk, None, 2, d_idx)
    assert ei.value.code == -1 and "flags" in str(ei.value)
    # with flags the N read is a miss and the other one a hit
    d_fl = c.zeros(2, np.uint8)
    c.pack(d_seq, d_qual, 2, 16, d_pk, d_qn, d_fl)
    c.match_and_count(d_pk, d_fl, 2, d_idx)
    assert list(d_idx.to_host()) == [0xFFFFFFFF, 1]
    c.close()


@pytest.mark.parametrize("hot", [False, True])
def test_3m_whitelist_1m_reads_bit_exact(hot, monkeypatch):
    """The SC3Pv3 list (3M-february-2018, cr_types/src/chemistry/chemistry_defs.json:72) has 6 794 880 entries: ~104 keys
    per pigeonhole bin instead of ~11, tables of ~55 MB instead of ~6, a 23-bit barcode rank.  1 M reads of the cfg2
    model drawn from a random list of that size: per-read index and both histograms equal the oracle's, with the plain
    kernels and with K1's LDS table + miss records forced on."""
    from cellranger_amd import synth as S

    if hot:
        monkeypatch.setenv("CRGPU_HOT_MIN_READS", "1")
    w = S.Workload(n_total=1_000_000, seed=S.SEED0 + 6, n_wl=6_794_880)
    st = _compare_barcode_stage(w, 1_000_000)
    # a denser list: more reads with an error land on ANOTHER whitelist entry or find two candidates
    assert st["corrected"] > 20_000 and st["invalid"] > 1_000 and st["valid"] > 800_000
    # 208 histogram buckets: with the table on, the rounds behind the sampling batch split their histogram by default
    assert (st["k1_split_rounds"] >= 1) == hot


@pytest.mark.parametrize("case", ["cfg2_1m", "dense_ties", "3m_list", "many_distinct", "record_overflow"])
def test_pass_b_in_barcode_order_bit_exact(case, monkeypatch):
    """Pass B over the misses SORTED by sequence (k_correct_sorted: one cooperative scan of the two pigeonhole bins per run
    of equal keys, the run's reads weigh the shared candidates with their own qualities; the default for lists of more
    than 2 M entries) gives the per-read indices, flags and both histograms of the oracle: on the cfg2 model, on short
    dense lists with several neighbours, ties and N's (the reads with an N go to the per-read kernel), on the 6.8 M-entry
    list, with almost every miss a different sequence (many runs per wave) and when the records overflow (the scan of idx
    takes over)."""
    from cellranger_amd import synth as S

    monkeypatch.setenv("CRGPU_HOT_MIN_READS", "1")   # miss records come from the LDS-table lookup
    monkeypatch.setenv("CRGPU_K2_SORTED", "1")
    if case == "cfg2_1m":
        st = _compare_barcode_stage(S.Workload(n_total=1_000_000, seed=S.SEED0 + 12), 1_000_000)
        assert st["corrected"] > 20_000
    elif case == "dense_ties":
        for cb_len, n_wl, thr in [(5, 300, 0.5), (6, 1500, 0.3), (8, 20000, 0.6), (11, 100000, 0.9)]:
            w = S.Workload(n_total=200_000, seed=177 + cb_len, n_wl=n_wl, n_cells=min(200, n_wl // 3),
                           n_ambient=n_wl // 2, cb_len=cb_len, cb_err=0.03, n_rate=0.004)
            assert _compare_barcode_stage(w, 200_000, threshold=thr)["corrected"] > 0
            _compare_barcode_stage(w, 50_000, first=200_000, threshold=0.95, max_err=1.0)
    elif case == "3m_list":
        monkeypatch.delenv("CRGPU_K2_SORTED")            # the default for a list of this size
        st = _compare_barcode_stage(S.Workload(n_total=1_000_000, seed=S.SEED0 + 16, n_wl=6_794_880), 1_000_000)
        assert st["corrected"] > 20_000
    elif case == "many_distinct":
        # as many cells as reads / 4: runs of equal misses are short, waves hold more than KS_MAX_RUNS different keys
        w = S.Workload(n_total=300_000, seed=S.SEED0 + 13, n_cells=60_000, n_ambient=100_000, cb_err=0.02)
        assert _compare_barcode_stage(w, 300_000)["corrected"] > 20_000
    else:
        monkeypatch.setenv("CRGPU_MISS_RECORD_CAP", "3")
        assert _compare_barcode_stage(S.Workload(n_total=400_000, seed=S.SEED0 + 14), 400_000)["corrected"] > 8_000


@pytest.mark.parametrize("variant", ["full_staging", "split_histogram", "pass_b_sorted", "partial_list"])
def test_table_of_frequent_barcodes_on_a_translated_list(variant, monkeypatch):
    """A Feature Barcoding library looks its reads up in a translated list (Whitelist::Trans, whitelist.rs:497-504): the hit
    reports the partner's rank.  In a one-library call pass A's LDS table of frequent barcodes is built from the list's own
    rank -> key array (the canonical keys are the PARTNERS' sequences), cold hits map position -> rank, the miss records
    feed pass B.  Per-read indices, flags and both histograms equal the oracle's -- also with the split histogram, with
    pass B in barcode order, and for a plain list that holds only part of the canonical space."""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd._lib import COUNTS_CORRECTED, COUNTS_VALID

    monkeypatch.setenv("CRGPU_HOT_MIN_READS", "1")
    if variant == "split_histogram":
        monkeypatch.setenv("CRGPU_K1_SPLIT", "1")
    if variant == "pass_b_sorted":
        monkeypatch.setenv("CRGPU_K2_SORTED", "1")
    n, n_wl = 600_000, 50_000   # (600 K: from there the staging area is large enough for the split histogram)
    w = S.Workload(n_total=n, seed=11, n_wl=n_wl, n_cells=2000, n_ambient=20000)
    canon = w.wl_packed.copy()
    rng = np.random.default_rng(19)
    c = G.fresh_ctx()
    c.set_whitelist(0, canon, length=16)
    if variant == "partial_list":
        # library 1: a plain list of every other canonical barcode (ranks != positions, some ranks without a key)
        raw = np.ascontiguousarray(np.sort(canon)[::2])
        c.set_whitelist(1, raw, canon=canon, length=16)
        owl1 = O.Whitelist(E.unpack_seqs(raw, 16))
    else:
        raw = np.unique(rng.integers(0, 1 << 32, size=60_000, dtype=np.uint64))[:n_wl].astype(np.uint32)
        raw = rng.permutation(raw)
        translate_to = rng.permutation(n_wl).astype(np.uint32)
        c.set_whitelist(1, raw, canon=canon, translate_to=translate_to, length=16)
        owl1 = O.Whitelist(E.unpack_seqs(raw, 16), translated=E.unpack_seqs(canon[translate_to], 16))
    _, canon_sorted = c.canon_order()
    w_fb = S.Workload(n_total=n, seed=11, n_wl=len(raw), n_cells=2000, n_ambient=min(20000, len(raw) - 2000))
    w_fb.wl_packed[:] = raw
    r = w_fb.host_reads(0, n)
    r["flags"] |= 1                                             # every read belongs to library 1: a one-library call
    idx_a, idx_b, corr, _ = G.gpu_barcode_stage(c, r, n)
    assert c.stat(3) >= 1 or variant != "split_histogram"       # CRGPU_STAT_K1_SPLIT_ROUNDS
    owl0 = O.Whitelist(E.unpack_seqs(canon, 16))
    res = O.run_pipeline(G.oracle_reads_from_packed(r, 16, 12), [owl0, owl1], n_lib=2, count=False, n_threads=4)
    exp_a, exp_b = G.oracle_expected_idx(res, canon_sorted)
    assert np.array_equal(idx_a, exp_a) and np.array_equal(idx_b, exp_b)
    assert np.array_equal(corr, (res.bc_state == 2).astype(np.uint8))
    assert np.array_equal(c.get_counts(1, COUNTS_VALID), G.hist_as_rank_counts(res.valid_hist[1], 16, canon_sorted))
    assert np.array_equal(c.get_counts(1, COUNTS_CORRECTED), G.hist_as_rank_counts(res.corrected_hist[1], 16, canon_sorted))
    assert (res.bc_state == 2).sum() > 5_000 and (res.bc_state == 1).sum() > 400_000
    c.close()


def test_miss_records_are_kept_per_call_and_dropped_by_the_range_that_is_written(monkeypatch):
    """Pass A of every library of a well runs before the first pass B: the records of up to four pass-A calls are kept, each
    until its pass B has used it.  A write through the context drops exactly the sets that describe the written range (the
    feature extraction between the passes writes feature indices, not barcodes)."""
    import gpu_helpers as G
    from cellranger_amd import synth as S

    monkeypatch.setenv("CRGPU_HOT_MIN_READS", "1")
    n = 300_000
    w = S.Workload(n_total=2 * n, seed=S.SEED0 + 23, n_wl=20_000, n_cells=300, n_ambient=2000)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    sets = lambda: c.stat(11)   # CRGPU_STAT_MISS_RECORD_SETS
    dev = []
    for k in range(2):
        r = w.host_reads(k * n, n)
        d = dict(cb=c.upload(r["cb"]), cbq=c.upload(r["cb_qualn"]), fl=c.upload(r["flags"]), idx=c.empty(n, np.uint32), r=r)
        dev.append(d)
    other = c.empty(n, np.uint32)
    for k, d in enumerate(dev):
        c.match_and_count(d["cb"], d["fl"], n, d["idx"])
        assert sets() == k + 1
    other.upload(np.zeros(n, np.uint32))                       # an unrelated buffer: both sets stay
    assert sets() == 2
    c.correct(dev[0]["cb"], dev[0]["cbq"], dev[0]["fl"], n, dev[0]["idx"])   # pass B consumes its set
    assert sets() == 1
    dev[1]["cb"].upload(dev[1]["r"]["cb"])                     # the barcodes of the second call rewritten: its set goes
    assert sets() == 0
    c.match_and_count(dev[1]["cb"], dev[1]["fl"], n, dev[1]["idx"])
    assert sets() == 1
    c.match_and_count(dev[1]["cb"], dev[1]["fl"], n, dev[1]["idx"])          # the same buffers again: replaced, not added
    assert sets() == 1
    c.close()
