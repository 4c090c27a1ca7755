"""Barcode stage + count stage with one library, with three libraries mixed in one call (general kernels), and with three
libraries handed over one per call as MAKE_SHARD does (one-library kernels chosen from the flag bytes).
usage (GPU box): python3 scripts/bench_multilib.py [n_reads]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cellranger_amd import engine as E  # noqa: E402
from cellranger_amd import synth as S  # noqa: E402


def run(n, n_libs, per_call=False):
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
    if per_call:
        # three equal slices, each relabelled to one library (the generator mixes the libraries)
        fl = d["fl"].to_host()
        third = n // 3
        for lib in range(3):
            fl[lib * third:(lib + 1) * third if lib < 2 else n] = (fl[lib * third:(lib + 1) * third if lib < 2 else n] & 0xF0) | lib
        d["fl"].upload(fl)
        cuts = [0, third, 2 * third, n]

    def step():
        c.reset_counts()
        if per_call:
            for a, b in zip(cuts[:-1], cuts[1:]):
                c.match_and_count(d["cb"].ptr + 4 * a, d["fl"].ptr + a, b - a, d["idx"].ptr + 4 * a)
            for a, b in zip(cuts[:-1], cuts[1:]):
                c.correct(d["cb"].ptr + 4 * a, d["cbq"].ptr + 16 * a, d["fl"].ptr + a, b - a, d["idx"].ptr + 4 * a)
        else:
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
    print("%d libraries%s: n=%d  %.2f ms/step  %.2f G reads/s  %s" % (n_libs, ", one per call" if per_call else "", n, dt * 1e3, n / dt / 1e9, led))
    c.close()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
run(n, 1)
run(n, 3)
run(n, 3, per_call=True)
