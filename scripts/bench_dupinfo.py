"""Time of crgpu_count_records_dev (dedup + per-read DupInfo) against build_keys + count_keys on the cfg3 model.
usage (GPU box): python3 scripts/bench_dupinfo.py [n_reads] [ordered]
ordered: the reads are first brought into barcode order (stable; invalid barcodes last), the order in which shardio hands
them to ALIGN_AND_COUNT in the reference (cr_lib/src/barcode_sort.rs:97-162, align_and_count.rs:505-524) -- the per-read
DupInfo records of a barcode then land next to each other instead of all over the output."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cellranger_amd import engine as E  # noqa: E402
from cellranger_amd import synth as S  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
    ordered = len(sys.argv) > 2 and sys.argv[2] == "ordered"
    if ordered:
        import torch  # before the crgpu context: torch must see the device first
        torch.cuda.init()
    w = S.Workload(n_total=n, seed=S.SEED0 + 3)
    c = E.Context(0)
    c.set_whitelist(0, w.wl_packed, length=16)
    c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    d = dict(cb=c.empty(n, np.uint32), cbq=c.empty((n, 16), np.uint8), fl=c.empty(n, np.uint8), umi=c.empty(n, np.uint32),
             uq=c.empty((n, 12), np.uint8), ft=c.empty(n, np.uint32), idx=c.empty(n, np.uint32))
    c.synth(w, 0, n, cb=d["cb"].ptr, cb_qualn=d["cbq"].ptr, umi=d["umi"].ptr, umi_qualn=d["uq"].ptr, feature=d["ft"].ptr,
            flags=d["fl"].ptr)
    c.match_and_count(d["cb"], d["fl"], n, d["idx"])
    c.correct(d["cb"], d["cbq"], d["fl"], n, d["idx"])
    if ordered:
        c.synchronize()
        t = {k: torch.as_tensor(v, device="cuda:0") for k, v in d.items()}
        t = {k: (v.view(torch.int32) if v.dtype == torch.uint32 else v) for k, v in t.items()}  # no uint32 indexing in torch
        order = torch.argsort(t["idx"].to(torch.int64) & 0xFFFFFFFF, stable=True)
        for k in ("cb", "fl", "umi", "ft", "idx", "cbq", "uq"):
            t[k].copy_(t[k][order])
        torch.cuda.synchronize()
        del order
        torch.cuda.empty_cache()
        print("reads in barcode order")
    recs = c.records(n, w.umi_len, d["idx"], d["umi"], d["uq"], d["ft"], d["fl"])
    keys = c.empty(n, np.uint64)
    pu, rc, df = c.empty(n, np.uint32), c.empty(n, np.uint32), c.empty(n, np.uint8)

    def plain():
        nk = c.build_keys(recs, keys)
        return c.count_keys(keys, nk)

    def dup():
        return c.count_records(recs, pu, rc, df)

    for name, f in (("build_keys + count_keys", plain), ("count_records (DupInfo)", dup)):
        for _ in range(2):  # the device pool settles in two rounds
            f().free()
        c.synchronize()
        dt = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            r = f()
            c.synchronize()
            dt = min(dt, time.perf_counter() - t0)
            nm = r.n_molecules
            r.free()
        print("%-28s n=%d  %.2f ms  %.2f G reads/s  (%d molecules)" % (name, n, dt * 1e3, n / dt / 1e9, nm))


main()
