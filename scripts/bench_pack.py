"""Throughput of crgpu_pack_dev (ASCII barcode + qualities -> 2-bit + N-flagged qualities) and crgpu_shard_metrics_dev.
usage (GPU box): python3 scripts/bench_pack.py [n_reads]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cellranger_amd import engine as E  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
    L = 16
    rng = np.random.default_rng(3)
    m = 1 << 22
    seq = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, size=(m, L), p=[0.2499, 0.2499, 0.2499, 0.2499, 0.0004])]
    qual = rng.choice(np.array([35, 44, 58, 70], np.uint8), size=(m, L)).astype(np.uint8)
    c = E.Context(0)
    d_seq, d_qual = c.empty((n, L), np.uint8), c.empty((n, L), np.uint8)
    for r in range((n + m - 1) // m):
        k = min(m, n - r * m)
        c._check(c.L.crgpu_memcpy_h2d(c.h, d_seq.ptr + L * r * m, seq.ctypes.data, L * k))
        c._check(c.L.crgpu_memcpy_h2d(c.h, d_qual.ptr + L * r * m, qual.ctypes.data, L * k))
    d_pk, d_qn, d_fl = c.empty(n, np.uint32), c.empty((n, L), np.uint8), c.empty(n, np.uint8).zero()

    def timed(f, reps=3):
        f()
        c.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        c.synchronize()
        return (time.perf_counter() - t0) / reps

    dt = timed(lambda: c._check(c.L.crgpu_pack_dev(c.h, d_seq.ptr, d_qual.ptr, n, L, d_pk.ptr, d_qn.ptr, d_fl.ptr)))
    print("pack: n=%d  %.2f ms  %.1f G reads/s  %.0f GB/s of 53 B/read" % (n, dt * 1e3, n / dt / 1e9, n * 53 / dt / 1e9))
    dt = timed(lambda: c.shard_metrics(d_pk, d_qn, L, d_pk, d_qn, L, None, n))
    print("shard metrics: %.2f ms  %.1f G reads/s  %.0f GB/s of 40 B/read" % (dt * 1e3, n / dt / 1e9, n * 40 / dt / 1e9))


main()
