"""Throughput of crgpu_extract_features_dev (K3x) on read rows resident in HBM: anchored, floating and bare patterns.
usage: python scripts/fx_bench.py [n_reads] [stride]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cellranger_amd import engine as E  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    rng = np.random.default_rng(0)
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def rnd(L):
        return bytes(acgt[rng.integers(0, 4, L)]).decode()

    cases = {
        "anchored 5PNNNNNNNNNN(BC) x200, L=15": [("5PNNNNNNNNNN(BC)", rnd(15), k, 1) for k in range(200)],
        "floating (BC)GTTTAAGAGCTAAGCTGGAA x100, L=20": [("(BC)GTTTAAGAGCTAAGCTGGAA", rnd(20), k, 1) for k in range(100)],
        "bare (BC) x200, L=15": [("(BC)", rnd(15), k, 1) for k in range(200)],
    }
    c = E.Context(0)
    c.enable_timing(True) if hasattr(c, "enable_timing") else None
    seq = acgt[rng.integers(0, 4, (n, stride), dtype=np.uint8)]
    qual = rng.integers(45, 74, (n, stride), dtype=np.uint8)
    for name, defs in cases.items():
        s = seq.copy()
        # plant a feature in 80 % of the reads where the pattern expects it, 10 % of them with one substitution
        L = len(defs[0][1])
        feats = np.stack([np.frombuffer(d[1].encode(), np.uint8) for d in defs])
        pick = feats[rng.integers(0, len(defs), n)]
        mut = rng.random(n) < 0.1
        pos = rng.integers(0, L, n)
        pick[mut, pos[mut]] = acgt[rng.integers(0, 4, int(mut.sum()))]
        planted = rng.random(n) < 0.8
        at = 10 if defs[0][0].startswith("5P") else min(30, max(0, stride - L - 20))
        s[planted, at:at + L] = pick[planted]
        if "GTTTAAG" in defs[0][0]:
            suf = np.frombuffer(b"GTTTAAGAGCTAAGCTGGAA", np.uint8)
            s[planted, at + L:at + L + len(suf)] = suf
        d_s, d_q = c.upload(s), c.upload(qual)
        d_f = c.empty(n, np.uint32)
        dist = np.full(len(defs), 1.0 / len(defs))
        c.set_feature_extractor(0, defs, dist)
        for _ in range(2):
            c.extract_features(0, n, d_f, r2=(d_s, d_q, None, stride))
        c.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            c.extract_features(0, n, d_f, r2=(d_s, d_q, None, stride))
        c.synchronize()
        dt = (time.perf_counter() - t0) / reps
        f = d_f.to_host()
        print(f"{name}: {dt * 1e3:.2f} ms for {n} reads = {n / dt / 1e9:.2f} G reads/s, {2 * n * stride / dt / 1e9:.0f} GB/s of rows, "
              f"{(f != 0xFFFFFFFF).mean() * 100:.1f} % assigned", flush=True)
        d_s.free(); d_q.free(); d_f.free()
    c.close()


if __name__ == "__main__":
    main()
