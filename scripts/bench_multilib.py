"""Barcode stage + count stage with one library against three libraries sharing a whitelist (the one-library fast paths
of K1 / K2 are off in the second case).  usage (GPU box): python3 scripts/bench_multilib.py [n_reads]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cellranger_amd import engine as E  # noqa: E402
from cellranger_amd import synth as S  # noqa: E402


def run(n, n_libs):
    w = S.Workload(n_total=n, seed=S.SEED0 + 3, n_libs=n_libs)
    c = E.Context(0)
    for lib in range(n_libs):
        c.set_whitelist(lib, w.wl_packed, length=16)
    c.set_key_layout(w.n_genes, w.umi_len, n_libs, 0)
    d = dict(cb=c.empty(n, np.uint32), cbq=c.empty((n, 16), np.uint8), fl=c.empty(n, np.uint8), umi=c.empty(n, np.uint32),
             uq=c.empty((n, 12), np.uint8), ft=c.empty(n, np.uint32), idx=c.empty(n, np.uint32), keys=c.empty(n, np.uint64))
    c.synth(w, 0, n, cb=d["cb"].ptr, cb_qualn=d["cbq"].ptr, umi=d["umi"].ptr, umi_qualn=d["uq"].ptr, feature=d["ft"].ptr,
            flags=d["fl"].ptr)
    recs = c.records(n, w.umi_len, d["idx"], d["umi"], d["uq"], d["ft"], d["fl"])

    def step():
        c.reset_counts()
        c.match_and_count(d["cb"], d["fl"], n, d["idx"])
        c.correct(d["cb"], d["cbq"], d["fl"], n, d["idx"])
        nk = c.build_keys(recs, d["keys"])
        c.count_keys(d["keys"], nk).free()

    for _ in range(2):
        step()
    c.synchronize()
    c.timing_reset()
    c.timing(True)
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    c.synchronize()
    dt = (time.perf_counter() - t0) / 3
    led = {k: round(v[0] / 3, 2) for k, v in c.timing_get().items() if v[1]}
    print("%d libraries: n=%d  %.2f ms/step  %.2f G reads/s  %s" % (n_libs, n, dt * 1e3, n / dt / 1e9, led))
    c.close()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
run(n, 1)
run(n, 3)
