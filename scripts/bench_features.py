"""Throughput of K3 (feature-barcode matching / correction, crgpu_match_features_dev) on synthetic captures.
usage (GPU box): python3 scripts/bench_features.py [n_reads] [n_features]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from cellranger_amd import engine as E  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    n_feat = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    L = 15
    rng = np.random.default_rng(5)
    feats = np.unique(rng.integers(0, 1 << 30, size=4 * n_feat, dtype=np.uint64))[:n_feat].astype(np.uint32)
    feat_ascii = E.unpack_seqs(feats, L)
    c = E.Context(0)
    dist = np.full(n_feat, 1.0 / n_feat)
    c.set_feature_pattern(0, feat_ascii, np.arange(n_feat, dtype=np.uint32), dist)
    m = min(n, 1 << 24)
    src = rng.integers(0, n_feat, m)
    pk = feats[src].copy()
    err = rng.random(m) < 0.1
    pos = rng.integers(0, L, m)
    pk[err] ^= (np.uint32(1) << (2 * pos[err]).astype(np.uint32))
    qn = rng.choice(np.array([35, 44, 58, 70], np.uint8), size=(m, L)).astype(np.uint8)
    reps = (n + m - 1) // m
    d_seq, d_q, d_out = c.empty(n, np.uint32), c.empty((n, L), np.uint8), c.empty(n, np.uint32)
    for r in range(reps):
        k = min(m, n - r * m)
        c._check(c.L.crgpu_memcpy_h2d(c.h, d_seq.ptr + 4 * r * m, pk.ctypes.data, 4 * k))
        c._check(c.L.crgpu_memcpy_h2d(c.h, d_q.ptr + L * r * m, qn.ctypes.data, L * k))
    c.match_features(0, d_seq, d_q, n, d_out)
    c.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        c.match_features(0, d_seq, d_q, n, d_out)
    c.synchronize()
    dt = (time.perf_counter() - t0) / 3
    out = d_out.to_host(count=m)
    print("n=%d features=%d: %.2f ms  %.1f G reads/s  (%.1f %% matched, %.0f GB/s of %d B/read)" %
          (n, n_feat, dt * 1e3, n / dt / 1e9, 100.0 * (out != 0xFFFFFFFF).mean(), n * (4 + L + 4) / dt / 1e9, 4 + L + 4))


main()
