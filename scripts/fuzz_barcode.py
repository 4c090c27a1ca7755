"""One-off randomized parity hunt for the barcode stage (not part of the suite): barcode lengths 4..16, dense and sparse
whitelists, a second library with a translation whitelist, confidence thresholds and expected-error limits, batch sizes
on tile edges, one call per library or mixed calls -- indices, corrected flags and all four histograms against the oracle.
usage (GPU box): python3 scripts/fuzz_barcode.py [n_trials] [seed]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_helpers as G  # noqa: E402
import oracle_lib as O  # noqa: E402
from cellranger_amd import engine as E  # noqa: E402
from cellranger_amd import synth as S  # noqa: E402
from cellranger_amd._lib import COUNTS_CORRECTED, COUNTS_VALID  # noqa: E402


def one(rng, t):
    cb_len = int(rng.choice([4, 5, 6, 8, 10, 11, 12, 14, 16]))
    space = 4 ** cb_len
    n_wl = int(min(rng.choice([20, 300, 5000, 100_000, 737_280]), space // 2))
    two_libs = rng.random() < 0.5
    n = int(rng.choice([1, 63, 65, 4095, 4097, 8192, 16385, 32769, 65537])) if rng.random() < 0.4 else int(rng.integers(2, 250_000))
    thr = float(rng.choice([0.3, 0.6, 0.9, 0.975]))
    max_err = float(rng.choice([np.finfo(np.float64).max, 0.05, 1.0]))
    per_call = two_libs and rng.random() < 0.5
    cfg = dict(cb_len=cb_len, n_wl=n_wl, two_libs=two_libs, n=n, thr=thr, max_err=max_err, per_call=per_call)
    kw = dict(n_total=n, seed=7000 + t, n_wl=n_wl, n_cells=max(1, min(int(rng.integers(1, 300)), n_wl // 2)),
              n_ambient=min(int(rng.integers(0, 2000)), n_wl // 2), cb_len=cb_len, cb_err=float(rng.choice([0.0, 0.02, 0.1])),
              n_rate=float(rng.choice([0.0, 0.01, 0.05])), n_genes=0)
    w = S.Workload(**kw)
    canon = w.wl_packed.copy()
    c = G.fresh_ctx()
    c.set_whitelist(0, canon, length=cb_len)
    c.set_posterior(max_err, thr)
    owls = [O.Whitelist(E.unpack_seqs(canon, cb_len))]
    r = w.host_reads(0, n)
    if two_libs:
        raw = rng.permutation(np.unique(rng.integers(0, space, size=3 * n_wl + 16, dtype=np.uint64)))[:n_wl].astype(np.uint32)
        if len(raw) < n_wl:
            return None
        translate_to = rng.permutation(n_wl).astype(np.uint32)
        c.set_whitelist(1, raw, canon=canon, translate_to=translate_to, length=cb_len)
        owls.append(O.Whitelist(E.unpack_seqs(raw, cb_len), translated=E.unpack_seqs(canon[translate_to], cb_len)))
        w1 = S.Workload(**kw)
        w1.wl_packed[:] = raw
        r1 = w1.host_reads(n, n)
        r1["flags"] |= 1
        if per_call:
            r = {k: np.concatenate([r[k], r1[k]]) for k in r}          # library 0 first, then library 1
        else:
            mix = rng.permutation(2 * n)
            r = {k: np.concatenate([r[k], r1[k]])[mix] for k in r}
        n = 2 * n
    _, canon_sorted = c.canon_order()
    if per_call:
        h = n // 2
        parts = [{k: np.ascontiguousarray(v[a:b]) for k, v in r.items()} for a, b in ((0, h), (h, n))]
        devs = []
        for p in parts:                                                # pass A for every library, then pass B
            d = dict(cb=c.upload(p["cb"]), cbq=c.upload(p["cb_qualn"]), fl=c.upload(p["flags"]), idx=c.empty(len(p["cb"]), np.uint32),
                     corr=c.empty(len(p["cb"]), np.uint8))
            c.match_and_count(d["cb"], d["fl"], len(p["cb"]), d["idx"])
            devs.append(d)
        idx_a = np.concatenate([d["idx"].to_host() for d in devs])
        for d in devs:
            c.correct(d["cb"], d["cbq"], d["fl"], d["idx"].size, d["idx"], d["corr"])
        idx_b = np.concatenate([d["idx"].to_host() for d in devs])
        corr = np.concatenate([d["corr"].to_host() for d in devs])
    else:
        idx_a, idx_b, corr, _ = G.gpu_barcode_stage(c, r, n)
    res = O.run_pipeline(G.oracle_reads_from_packed(r, cb_len, 12), owls, n_lib=len(owls), count=False, n_threads=4,
                         max_expected_errors=max_err, threshold=thr)
    exp_a, exp_b = G.oracle_expected_idx(res, canon_sorted)
    assert np.array_equal(idx_a, exp_a), "idx after pass A"
    assert np.array_equal(idx_b, exp_b), "idx after pass B"
    assert np.array_equal(corr, (res.bc_state == 2).astype(np.uint8)), "corrected flags"
    for lib in range(len(owls)):
        assert np.array_equal(c.get_counts(lib, COUNTS_VALID), G.hist_as_rank_counts(res.valid_hist[lib], cb_len, canon_sorted)), "valid"
        assert np.array_equal(c.get_counts(lib, COUNTS_CORRECTED),
                              G.hist_as_rank_counts(res.corrected_hist[lib], cb_len, canon_sorted)), "corrected"
    c.close()
    cfg["n_corrected"] = int((res.bc_state == 2).sum())
    return cfg


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
    bad = 0
    for t in range(trials):
        try:
            cfg = one(rng, t)
            print("ok  ", cfg, flush=True)
        except Exception:
            bad += 1
            print("FAIL trial", t, flush=True)
            traceback.print_exc()
    print("trials done, failures:", bad)
    sys.exit(1 if bad else 0)


main()
