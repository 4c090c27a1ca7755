"""Rehearsal of the N > 1 path at scale on ONE GPU: N ranks as host threads with their own contexts, joined by the in-process
group (barriers + device-to-device copies; the RCCL transport runs under the same code).  Not a scaling measurement -- the
ranks share one device -- but the exchange sizes, offsets and 32-bit limits of a real job are exercised, the rank-0 matrix
is checked against the single-context result on all reads, and the time spent in the collectives is reported.
usage (GPU box): python3 scripts/rehearse_ranks.py [ranks] [reads_per_rank]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from cellranger_amd import engine as E  # noqa: E402
from cellranger_amd import synth as S  # noqa: E402
from cellranger_amd.pipeline import CountPipeline, HipBackend  # noqa: E402


def make_shard(ctx, w, first, n):
    shard = dict(n=n, umi_len=w.umi_len)
    for name, shape, dt in (("cb", n, np.uint32), ("cb_qualn", (n, w.cb_len), np.uint8), ("flags", n, np.uint8), ("idx", n, np.uint32),
                            ("umi", n, np.uint32), ("umi_qualn", (n, w.umi_len), np.uint8), ("feature", n, np.uint32),
                            ("keys", n, np.uint64)):
        shard[name] = ctx.empty(shape, dt)
    chunk = 1 << 27
    for off in range(0, n, chunk):
        m = min(chunk, n - off)
        ctx.synth(w, first + off, m, cb=shard["cb"].ptr + off * 4, cb_qualn=shard["cb_qualn"].ptr + off * w.cb_len,
                  umi=shard["umi"].ptr + off * 4, umi_qualn=shard["umi_qualn"].ptr + off * w.umi_len,
                  feature=shard["feature"].ptr + off * 4, flags=shard["flags"].ptr + off)
    ctx.synchronize()
    return shard


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    n = world * per
    w = S.Workload(n_total=n, seed=S.SEED0 + 3)
    gid = E.local_group_id(world)
    out, errors = [None] * world, []

    def worker(rank):
        c = None
        try:
            c = E.Context(0, n_ranks=world, rank=rank, unique_id=gid)
            c.trust_unchanged_buffers(True)
            c.set_whitelist(0, w.wl_packed, length=w.cb_len)
            c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
            shard = make_shard(c, w, rank * per, per)
            be = HipBackend(c, 0)
            pipe = CountPipeline(be, libs=(0,))
            be.reset()
            m = pipe.run(shard)  # warm-up
            del m
            c.synchronize()
            c.barrier()
            c.timing_reset()
            c.timing(True)
            t0 = time.perf_counter()
            be.reset()
            m = pipe.run(shard)
            c.synchronize()
            c.barrier()
            dt = time.perf_counter() - t0
            c.timing(False)
            out[rank] = (c, m, dt, c.timing_get())
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            if c is not None:
                c.close()

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errors:
        print("FAILED", errors)
        sys.exit(1)
    c0, m0, dt, led = out[0]
    print("%d thread-ranks x %d reads on one GPU: step %.1f ms; rank 0 ledger (ms): %s" % (
        world, per, dt * 1e3, {k: round(v[0], 2) for k, v in led.items() if v[1]}))
    print("rank-0 matrix: %d columns, nnz %d" % (m0.n_barcodes, m0.nnz))
    cols, nnz, ip, ind, dat = m0.n_barcodes, m0.nnz, None, None, None
    rank_, indptr, indices, data = m0.download()
    for c, _, _, _ in out:
        c.close()
    # the same reads through ONE context
    c = E.Context(0)
    c.trust_unchanged_buffers(True)
    c.set_whitelist(0, w.wl_packed, length=w.cb_len)
    c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    shard = make_shard(c, w, 0, n)
    be = HipBackend(c, 0)
    m1 = CountPipeline(be, libs=(0,)).run(shard)
    r1, ip1, in1, d1 = m1.download()
    ok = (np.array_equal(rank_, r1) and np.array_equal(indptr, ip1) and np.array_equal(indices, in1) and np.array_equal(data, d1))
    print("equal to the single-context matrix on all %d reads: %s" % (n, ok))
    sys.exit(0 if ok else 1)


main()
