"""K3x parity: FeatureExtractor::match_read over whole reads (tethered and bare patterns, several captures) on the GPU
against the reference's vectors and against the oracle's string-interpreting restatement (oracle/feature_extract.c)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rows(seqs, quals, stride):
    n = len(seqs)
    s = np.zeros((n, stride), np.uint8)
    q = np.zeros((n, stride), np.uint8)
    ln = np.zeros(n, np.uint32)
    for i, (a, b) in enumerate(zip(seqs, quals)):
        a = a.encode() if isinstance(a, str) else bytes(a)
        b = b.encode() if isinstance(b, str) else bytes(b)
        s[i, :len(a)] = np.frombuffer(a, np.uint8)
        q[i, :len(b)] = np.frombuffer(b, np.uint8)
        ln[i] = len(a)
    return s, q, ln


def _gpu_extract(c, extractor, r1=None, r2=None, stride_pad=0, no_len=False):
    """r1 / r2: (seqs, quals) lists -> (feature, n_ids, capture) host arrays.  no_len: every row is full (d_len NULL)"""
    n = len((r1 or r2)[0])
    args = {}
    keep = []
    for name, r in (("r1", r1), ("r2", r2)):
        if r is None:
            continue
        stride = (max(4, max(len(x) for x in r[0])) + 3) // 4 * 4 + stride_pad   # dword rows unless stride_pad makes them odd
        s, q, ln = _rows(r[0], r[1], stride)
        ds, dq, dl = c.upload(s), c.upload(q), (None if no_len else c.upload(ln))
        keep += [ds, dq, dl]
        args[name] = (ds, dq, dl, stride)
    d_f, d_n, d_c = c.empty(n, np.uint32), c.empty(n, np.uint32), c.empty(n, np.uint32)
    c.extract_features(extractor, n, d_f, d_n_ids_out=d_n, d_capture_out=d_c, **args)
    return d_f.to_host(), d_n.to_host(), d_c.to_host()


def _decode(cap):
    from cellranger_amd._lib import NO_CAPTURE
    if cap == NO_CAPTURE:
        return None
    return dict(corrected=bool(cap >> 31), read=int(cap >> 30) & 1, start=int(cap >> 8) & 0x3FFFFF, len=int(cap & 0xFF))


@pytest.mark.parametrize("name", ["correct_feature", "correct_bare_feature"])
def test_extractor_golden_vectors_on_gpu(name):
    """the reference's own cases, run as its tests run them: through match_read on a read that is the sequence itself"""
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd._lib import NO_FEATURE

    with open(os.path.join(GOLD, "feature_vectors.json")) as f:
        g = json.load(f)[name]
    dist = O.compute_feature_dist(g["counts"], g["types"])
    c = G.fresh_ctx()
    for t in sorted(set(g["types"])):
        c.set_feature_extractor(t, [(g["pattern"], g["features"][i], i, 0) for i, x in enumerate(g["types"]) if x == t], dist)
    for case in g["cases"]:
        f, n_ids, cap = _gpu_extract(c, case["type"], r1=([case["seq"]], [case["qual"]]))
        if case["expect"] is None:
            assert f[0] == NO_FEATURE and n_ids[0] == 0, case
        else:
            assert f[0] != NO_FEATURE and g["features"][f[0]] == case["expect"] and n_ids[0] == 1, case
            assert _decode(cap[0])["corrected"], case
    c.close()


def test_compiled_patterns_match_reference_expressions():
    import gpu_helpers as G
    import oracle_lib as O

    defs = [("5PNN(BC)", "ACGT", 0, 1), ("(BC)GG3P", "TTTTT", 1, 1), ("(BC)", "CCCC", 2, 1), ("(BC)", "GGGG", 3, 1),
            ("(BC)", "GGGGG", 4, 1), ("5PNN(BC)", "TTTT", 5, 1), ("5PNN(BC)", "TTTT", 6, 0)]
    c = G.fresh_ctx()
    c.set_feature_extractor(0, defs)
    assert sorted(c.feature_extractor_regexes(0)) == sorted(O.FeatureExtractor(defs).regexes())
    from cellranger_amd import engine as E
    with pytest.raises(E.CrgpuError):
        c.set_feature_extractor(1, [("^(BC)", "ACGT", 0, 0), ("^(BC)", "ACGT", 1, 0)])
    with pytest.raises(E.CrgpuError):
        c.set_feature_extractor(1, [("(BC)Q", "ACGT", 0, 0)])
    with pytest.raises(E.CrgpuError):
        c.set_feature_extractor(1, [("^(BC)", "ACXT", 0, 0)])
    with pytest.raises(E.CrgpuError):  # R2 patterns but no R2 rows
        d = c.empty(1, np.uint32)
        c.extract_features(0, 1, d)
    c.close()


def _random_reads(rng, n, defs, lo, hi):
    """reads that hold planted features (exact, one or two substitutions, an N) behind random offsets, or noise"""
    seqs, quals = [], []
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for _ in range(n):
        ln = int(rng.integers(lo, hi + 1))
        s = acgt[rng.integers(0, 4, ln)].copy()
        for _ in range(int(rng.integers(0, 3))):
            pat, fs, _, _ = defs[int(rng.integers(0, len(defs)))]
            f = np.frombuffer(fs.encode(), np.uint8).copy()
            r = rng.random()
            if r < 0.35:
                k = int(rng.integers(0, len(f)))
                f[k] = acgt[rng.integers(0, 4)]
            elif r < 0.45:
                for k in rng.integers(0, len(f), 2):
                    f[k] = acgt[rng.integers(0, 4)]
            elif r < 0.55:
                f[int(rng.integers(0, len(f)))] = ord("N")
            # put it where the pattern would look for it, most of the time
            left = pat.split("(BC)")[0].lstrip("5Pp^-_")
            at = len(left) if (pat[0] in "5^" and rng.random() < 0.8) else int(rng.integers(0, max(1, ln - len(f) + 1)))
            if pat.endswith(("3P", "3p", "$")) and rng.random() < 0.8:
                right = pat.split("(BC)")[1].rstrip("3Pp$-_")
                at = max(0, ln - len(right) - len(f))
                if at + len(f) + len(right) <= ln:
                    s[at + len(f):at + len(f) + len(right)] = np.frombuffer(right.replace("N", "A").encode(), np.uint8)
            if at + len(f) <= ln:
                s[at:at + len(f)] = f
                if at >= len(left) and rng.random() < 0.8:
                    s[at - len(left):at] = np.frombuffer(left.replace("N", "C").encode(), np.uint8) if left else s[at:at]
                right = pat.split("(BC)")[1]
                if right and not pat.endswith(("3P", "3p", "$")) and at + len(f) + len(right) <= ln and rng.random() < 0.8:
                    s[at + len(f):at + len(f) + len(right)] = np.frombuffer(right.replace("N", "G").encode(), np.uint8)
        if rng.random() < 0.05:
            s[rng.integers(0, ln, 3)] = ord("N")
        q = rng.integers(33, 75, ln).astype(np.uint8)
        q[rng.random(ln) < 0.05] = 20  # below the offset: the reference's u8 subtraction wraps
        seqs.append(bytes(s))
        quals.append(bytes(q))
    return seqs, quals


def _compare_with_oracle(c, extractor, ox, r1, r2, n_feat_total, **kw):
    from cellranger_amd._lib import NO_FEATURE
    f, n_ids, cap = _gpu_extract(c, extractor, r1=r1, r2=r2, **kw)
    n = len(f)
    stats = dict(none=0, raw=0, one=0, multi=0)
    for i in range(n):
        a = (r1[0][i], r1[1][i]) if r1 else (None, None)
        b = (r2[0][i], r2[1][i]) if r2 else (None, None)
        want = ox.match_read(a[0], a[1], b[0], b[1])
        got = _decode(cap[i])
        if want is None:
            assert got is None and f[i] == NO_FEATURE and n_ids[i] == 0, (i, a, b, got)
            stats["none"] += 1
            continue
        assert got is not None, (i, a, b, want)
        assert (got["corrected"], got["read"], got["start"], got["len"]) == (want["corrected"], want["read"], want["start"], want["len"]), \
            (i, a, b, got, want)
        assert n_ids[i] == want["n_ids"], (i, a, b, n_ids[i], want)
        if want["n_ids"] == 1:
            assert f[i] == want["ids"][0], (i, a, b, f[i], want)
            stats["one"] += 1
        else:
            assert f[i] == NO_FEATURE
            stats["multi" if want["n_ids"] else "raw"] += 1
    return stats


@pytest.mark.parametrize("seed,with_dist", [(1, True), (2, True), (3, False)])
def test_extractor_random_vs_oracle(seed, with_dist):
    """mixed definitions on both reads: anchored, floating, N wildcards, 3' anchored, two bare groups of different length,
    20-base guides; reads with planted features, mutations, Ns, low qualities"""
    import gpu_helpers as G
    import oracle_lib as O

    rng = np.random.default_rng(seed)
    acgt = "ACGT"

    def rnd(L):
        return "".join(acgt[k] for k in rng.integers(0, 4, L))

    defs, idx = [], 0
    for pat, L, read, k in [("5PNNNN(BC)", 15, 1, 40), ("(BC)GTTTAAGAGC", 20, 1, 30), ("(BC)", 8, 1, 25), ("(BC)", 6, 1, 10),
                            ("^NN(BC)NNC", 10, 0, 12), ("(BC)AC3P", 7, 1, 9), ("ACGN(BC)", 12, 0, 9)]:
        seen = set()
        base = rnd(L)
        while len(seen) < k:
            # families of close sequences so that corrections compete
            s = list(base if rng.random() < 0.5 else rnd(L))
            for j in rng.integers(0, L, int(rng.integers(0, 3))):
                s[j] = acgt[rng.integers(0, 4)]
            seen.add("".join(s))
        for s in sorted(seen):
            defs.append((pat, s, idx, read))
            idx += 1
    order = rng.permutation(len(defs))
    defs = [defs[k] for k in order]
    counts = rng.integers(0, 1000, idx)
    counts[rng.random(idx) < 0.1] = 0
    dist = O.compute_feature_dist(counts, np.zeros(idx, np.uint32)) if with_dist else None
    c = G.fresh_ctx()
    c.set_feature_extractor(3, defs, dist)
    ox = O.FeatureExtractor(defs, dist)
    n = 6000
    r1 = _random_reads(rng, n, [d for d in defs if d[3] == 0], 12, 40)
    r2 = _random_reads(rng, n, [d for d in defs if d[3] == 1], 5, 110)
    stats = _compare_with_oracle(c, 3, ox, r1, r2, idx)
    assert stats["one"] > n // 10 and stats["raw"] > 0 and stats["none"] >= 0, stats
    if with_dist:
        assert stats["multi"] > 0, stats
    c.close()


def test_extractor_wide_map_goes_through_the_queue():
    """a bare group made of all 24 one-mismatch neighbours of a sequence that is NOT a feature itself: a window equal to that
    sequence reaches 24 distinct features, more than the 16-entry local map, so these reads are redone with map rows in
    global memory -- same answers"""
    import gpu_helpers as G
    import oracle_lib as O

    rng = np.random.default_rng(9)
    centre = "ACGTTGCA"
    feats = [centre[:i] + b + centre[i + 1:] for i in range(8) for b in "ACGT" if b != centre[i]]
    defs = [("(BC)", s, k, 1) for k, s in enumerate(feats)]
    counts = rng.integers(1, 50, len(feats))
    dist = O.compute_feature_dist(counts, np.zeros(len(feats), np.uint32))
    c = G.fresh_ctx()
    c.set_feature_extractor(0, defs, dist)
    ox = O.FeatureExtractor(defs, dist)
    r2 = _random_reads(rng, 3000, defs, 8, 60)
    seqs, quals = list(r2[0]), list(r2[1])
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for k in range(300):  # the centre itself, alone or next to an exact feature, under varied qualities
        ln = int(rng.integers(8, 50))
        s = acgt[rng.integers(0, 4, ln)].copy()
        at = int(rng.integers(0, ln - 7))
        s[at:at + 8] = np.frombuffer(centre.encode(), np.uint8)
        if k % 3 == 0 and at + 16 <= ln:
            s[at + 8:at + 16] = np.frombuffer(feats[int(rng.integers(0, 24))].encode(), np.uint8)
        seqs.append(bytes(s))
        quals.append(bytes(rng.integers(33, 75, ln).astype(np.uint8)))
    stats = _compare_with_oracle(c, 0, ox, None, (seqs, quals), len(feats))
    assert stats["one"] > 100, stats
    assert c.stat(2) >= 300  # CRGPU_STAT_FEATURE_READS_REQUEUED
    c.close()


@pytest.mark.parametrize("pat,L,read,fixed", [("5PNNNNNNNNNN(BC)", 15, 1, True), ("5PNNNNNNNNNN(BC)", 15, 1, False),
                                              ("(BC)GTTTAAGAGCTAAGCTGGAA", 20, 1, False), ("(BC)AC3P", 7, 1, True),
                                              ("(BC)AC3P", 7, 1, False), ("ACGN(BC)", 12, 0, False), ("^NN(BC)NNC$", 10, 0, False),
                                              ("^(BC)", 32, 0, True), ("TTN(BC)GNA", 9, 1, True)])
@pytest.mark.parametrize("with_dist", [True, False])
def test_single_tethered_pattern_lds_kernel_vs_oracle(pat, L, read, fixed, with_dist):
    """An extractor whose definitions share ONE tethered pattern (the usual feature reference) goes through the
    wave-per-rows kernel with the table in LDS: same answers as the oracle's regex interpreter and as the thread-per-read
    kernel (CRGPU_FEATURES_GLOBAL=1), on anchored / floating patterns, with and without per-row lengths, dense families of
    one-mismatch neighbours (more candidates than the generic kernel's 16-entry map), Ns and wrapping qualities."""
    import gpu_helpers as G
    import oracle_lib as O

    rng = np.random.default_rng(len(pat) * 131 + L + 7 * fixed + with_dist)
    acgt = "ACGT"

    def rnd(k):
        return "".join(acgt[j] for j in rng.integers(0, 4, k))

    centre = rnd(L)
    seen = set()
    for i in range(L):                      # every neighbour of a centre that is no feature itself
        for b in acgt:
            if b != centre[i]:
                seen.add(centre[:i] + b + centre[i + 1:])
    base = rnd(L)
    while len(seen) < 3 * L + 60:
        s_ = list(base if rng.random() < 0.5 else rnd(L))
        for j in rng.integers(0, L, int(rng.integers(0, 3))):
            s_[j] = acgt[rng.integers(0, 4)]
        if "".join(s_) != centre:
            seen.add("".join(s_))
    feats = sorted(seen)
    order = rng.permutation(len(feats))
    defs = [(pat, feats[k], int(j) + 3, read) for j, k in enumerate(order)]
    n_idx = len(feats) + 3
    counts = rng.integers(0, 1000, n_idx)
    counts[rng.random(n_idx) < 0.1] = 0
    dist = O.compute_feature_dist(counts, np.zeros(n_idx, np.uint32)) if with_dist else None
    c = G.fresh_ctx()
    c.set_feature_extractor(2, defs, dist)
    ox = O.FeatureExtractor(defs, dist)
    n = 5000
    left = pat.split("(BC)")[0].lstrip("5Pp^-_")
    right = pat.split("(BC)")[1].rstrip("3Pp$-_")
    need = len(left) + L + len(right)
    full = (need + 9 + 3) // 4 * 4            # rows without lengths must be full: one length, a multiple of 4
    lo, hi = (full, full) if fixed else (max(4, need - 3), need + 40)
    if pat.startswith("^") and pat.endswith("$"):
        lo, hi = need - 1, need + 1
    seqs, quals = _random_reads(rng, n, defs, lo, hi)
    seqs, quals = list(seqs), list(quals)
    an = np.frombuffer(b"ACGT", np.uint8)
    for k in range(400):    # the centre where the pattern looks, under varied qualities: up to 3 L competing candidates
        ln = int(rng.integers(lo, hi + 1))
        s_ = an[rng.integers(0, 4, ln)].copy()
        at = len(left) if pat[0] in "5^" else (ln - len(right) - L if pat.endswith(("3P", "$")) else int(rng.integers(len(left), max(len(left) + 1, ln - L - len(right) + 1))))
        if at >= len(left) and at + L + len(right) <= ln:
            s_[at - len(left):at] = np.frombuffer(left.replace("N", "C").encode(), np.uint8) if left else s_[at:at]
            s_[at:at + L] = np.frombuffer(centre.encode(), np.uint8)
            s_[at + L:at + L + len(right)] = np.frombuffer(right.replace("N", "G").encode(), np.uint8) if right else s_[at:at]
        seqs.append(bytes(s_))
        quals.append(bytes(rng.integers(33, 75, ln).astype(np.uint8)))
    reads = (seqs, quals)
    r1, r2 = (None, reads) if read else (reads, None)
    before = c.stat(4)
    stats = _compare_with_oracle(c, 2, ox, r1, r2, n_idx, no_len=fixed)
    assert c.stat(4) == before + 1, "the one-pattern LDS kernel did not run"
    assert stats["one"] > 500, stats
    fast = _gpu_extract(c, 2, r1=r1, r2=r2, no_len=fixed)
    os.environ["CRGPU_FEATURES_GLOBAL"] = "1"
    try:
        slow = _gpu_extract(c, 2, r1=r1, r2=r2, no_len=fixed)
    finally:
        del os.environ["CRGPU_FEATURES_GLOBAL"]
    assert c.stat(4) == before + 2
    for a, b in zip(fast, slow):
        assert np.array_equal(a, b)
    # an odd stride or base cannot take dword loads: the generic kernel serves those rows, same answers
    odd = _gpu_extract(c, 2, r1=r1, r2=r2, stride_pad=1)
    for a, b in zip(fast, odd):
        assert np.array_equal(a, b)
    c.close()
