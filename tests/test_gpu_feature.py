"""GPU parity tests of K3 (feature-barcode matching / correction) against the reference's golden
vectors (feature_extraction.rs:638-827) and the oracle on random captures."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pack_capture(seq, qual):
    """ASCII capture + quality string -> (packed u32, qualn bytes) like crgpu_pack_dev."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    k = 0
    qn = []
    for ch, q in zip(seq, qual):
        is_n = ch not in code
        k = (k << 2) | (0 if is_n else code[ch])
        qn.append((ord(q) if isinstance(q, str) else int(q)) | (0x80 if is_n else 0))
    return k, qn


def _gpu_match(c, pattern, seqs, quals):
    pk, qn = zip(*[_pack_capture(s, q) for s, q in zip(seqs, quals)])
    n = len(pk)
    d_seq = c.upload(np.array(pk, np.uint32))
    d_q = c.upload(np.array(qn, np.uint8))
    d_out = c.empty(n, np.uint32)
    c.match_features(pattern, d_seq, d_q, n, d_out)
    return d_out.to_host()


@pytest.mark.parametrize("name", ["correct_feature", "correct_bare_feature"])
def test_feature_golden_vectors_on_gpu(name):
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd._lib import NO_FEATURE

    with open(os.path.join(GOLD, "feature_vectors.json")) as f:
        g = json.load(f)[name]
    dist = O.compute_feature_dist(g["counts"], g["types"])
    c = G.fresh_ctx()
    for t in sorted(set(g["types"])):
        sel = [i for i, x in enumerate(g["types"]) if x == t]
        c.set_feature_pattern(t, [g["features"][i] for i in sel], sel, dist[sel])
    for case in g["cases"]:
        if case.get("multi_capture"):
            continue  # test_gpu_feature_extract.py
        got = _gpu_match(c, case["type"], [case["seq"]], [case["qual"]])[0]
        if case["expect"] is None:
            assert got == NO_FEATURE, case
        else:
            assert got != NO_FEATURE and g["features"][got] == case["expect"], case
    c.close()


@pytest.mark.parametrize("n_feat,n", [(200, 60_000), (1500, 20_000)])
def test_feature_matching_random_vs_oracle(n_feat, n):
    """(200 features: the 256-thread kernel; 1500: one 1024-thread workgroup per CU around a larger LDS table)"""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd._lib import NO_FEATURE

    rng = np.random.default_rng(4)
    L = 15
    feats = np.unique(rng.integers(0, 1 << 30, size=4 * n_feat, dtype=np.uint64))[:n_feat].astype(np.uint32)
    feats = rng.permutation(feats)
    feat_ascii = E.unpack_seqs(feats, L)
    counts = rng.integers(0, 1000, n_feat)
    counts[:5] = 0
    dist = O.compute_feature_dist(counts, np.zeros(n_feat, np.uint32))
    index = np.arange(100, 100 + n_feat, dtype=np.uint32)
    # captures: true feature, sometimes 1-2 substitutions, sometimes an N, sometimes random
    src = rng.integers(0, n_feat, n)
    seq = feat_ascii[src].copy()
    qual = rng.choice(np.array([35, 44, 58, 70], np.uint8), size=(n, L))
    for i in range(n):
        u = rng.random()
        if u < 0.3:
            p = rng.integers(0, L)
            seq[i, p] = rng.choice(np.frombuffer(b"ACGT", np.uint8))
        elif u < 0.4:
            for p in rng.integers(0, L, 2):
                seq[i, p] = rng.choice(np.frombuffer(b"ACGT", np.uint8))
        elif u < 0.45:
            seq[i, rng.integers(0, L)] = ord("N")
        elif u < 0.5:
            seq[i] = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    c = G.fresh_ctx()
    c.set_feature_pattern(0, feat_ascii, index, dist)
    c.set_feature_pattern(1, feat_ascii, index, None)   # exact matches only
    got = _gpu_match(c, 0, [bytes(s).decode() for s in seq], [list(q) for q in qual])
    got_exact = _gpu_match(c, 1, [bytes(s).decode() for s in seq], [list(q) for q in qual])
    exp = np.zeros(n, np.uint32)
    exp_exact = np.zeros(n, np.uint32)
    for i in range(n):
        f = O.find_closest_feature(feat_ascii, dist, bytes(seq[i]), bytes(qual[i]))
        exp[i] = NO_FEATURE if f < 0 else index[f]
        f = O.find_closest_feature(feat_ascii, None, bytes(seq[i]), bytes(qual[i]))
        exp_exact[i] = NO_FEATURE if f < 0 else index[f]
    assert np.array_equal(got, exp)
    assert np.array_equal(got_exact, exp_exact)
    assert (exp != NO_FEATURE).sum() > n // 2 and (exp != exp_exact).sum() > n // 60
    c.close()
