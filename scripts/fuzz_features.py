"""One-off randomized parity hunt for feature-barcode matching (not part of the suite): capture lengths 6..16, 1 to 5000
features (LDS hash set with 256 / 1024 threads and the global-memory fallback), with and without a prior, captures that
are exact, one or two substitutions away, contain an N, or are random.  usage: python3 scripts/fuzz_features.py [n] [seed]"""
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
from cellranger_amd._lib import NO_FEATURE  # noqa: E402

ACGT = np.frombuffer(b"ACGT", np.uint8)


def one(rng, t):
    L = int(rng.choice([6, 8, 10, 12, 15, 16]))
    n_feat = int(min(rng.choice([1, 2, 7, 200, 1023, 1025, 4095, 4097, 5000]), 4 ** L // 4))
    n = int(rng.choice([1, 255, 257, 1025, 4097])) if rng.random() < 0.4 else int(rng.integers(1, 30_000))
    feats = rng.permutation(np.unique(rng.integers(0, 4 ** L, size=4 * n_feat + 8, dtype=np.uint64)))[:n_feat].astype(np.uint32)
    n_feat = len(feats)
    feat_ascii = E.unpack_seqs(feats, L)
    counts = rng.integers(0, 1000, n_feat)
    counts[: max(1, n_feat // 10)] = 0
    use_prior = rng.random() < 0.7
    dist = O.compute_feature_dist(counts, np.zeros(n_feat, np.uint32)) if use_prior else None
    index = rng.permutation(n_feat + 50)[:n_feat].astype(np.uint32)
    src = rng.integers(0, n_feat, n)
    seq = feat_ascii[src].copy()
    qual = rng.choice(np.array([34, 35, 44, 58, 70, 73], np.uint8), size=(n, L))
    u = rng.random(n)
    for i in np.nonzero(u < 0.5)[0]:
        if u[i] < 0.3:
            seq[i, rng.integers(0, L)] = rng.choice(ACGT)
        elif u[i] < 0.4:
            for p in rng.integers(0, L, 2):
                seq[i, p] = rng.choice(ACGT)
        elif u[i] < 0.45:
            seq[i, rng.integers(0, L)] = ord("N")
        else:
            seq[i] = rng.choice(ACGT, L)
    c = G.fresh_ctx()
    c.set_feature_pattern(0, feat_ascii, index, dist)
    is_n = seq == ord("N")
    code = np.zeros(256, np.uint32)
    for k, ch in enumerate(b"ACGT"):
        code[ch] = k
    pk = np.zeros(n, np.uint32)
    for j in range(L):
        pk = (pk << np.uint32(2)) | np.where(is_n[:, j], 0, code[seq[:, j]]).astype(np.uint32)
    qn = (qual | np.where(is_n, 0x80, 0)).astype(np.uint8)
    out = c.empty(max(n, 1), np.uint32)
    c.match_features(0, c.upload(pk), c.upload(qn), n, out)
    got = out.to_host(count=n)
    exp = np.zeros(n, np.uint32)
    for i in range(n):
        f = O.find_closest_feature(feat_ascii, dist, bytes(seq[i]), bytes(qual[i]))
        exp[i] = NO_FEATURE if f < 0 else index[f]
    assert np.array_equal(got, exp), "first difference at %d" % int(np.nonzero(got != exp)[0][0])
    c.close()
    return dict(L=L, n_feat=n_feat, n=n, prior=use_prior, matched=int((exp != NO_FEATURE).sum()))


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    bad = 0
    for t in range(trials):
        try:
            print("ok  ", one(rng, t), flush=True)
        except Exception:
            bad += 1
            print("FAIL trial", t, flush=True)
            traceback.print_exc()
    print("trials done, failures:", bad)
    sys.exit(1 if bad else 0)


main()
