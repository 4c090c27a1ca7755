"""Experiment: two contexts (two HIP streams, two host threads) on ONE GPU, each running whole cfg3 steps on its own copy of
the per-step buffers: does the device overlap the latency-bound kernels of one step with the bandwidth-bound ones of the other?
usage (GPU box): python3 scripts/bench_lanes.py [reads] [lanes] [steps_per_lane]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402  (before the contexts: torch must see the device first)

torch.cuda.init()
from cellranger_amd import engine as E  # noqa: E402
from cellranger_amd import synth as S  # noqa: E402
from cellranger_amd.pipeline import CountPipeline, HipBackend  # noqa: E402


def make_lane(w, n):
    ctx = E.Context(0)
    ctx.trust_unchanged_buffers(True)
    ctx.set_whitelist(0, w.wl_packed, length=w.cb_len)
    shard = dict(n=n, umi_len=w.umi_len)
    shard["cb"] = ctx.empty(n, np.uint32)
    shard["cb_qualn"] = ctx.empty((n, w.cb_len), np.uint8)
    shard["flags"] = ctx.empty(n, np.uint8)
    shard["idx"] = ctx.empty(n, np.uint32)
    shard["umi"] = ctx.empty(n, np.uint32)
    shard["umi_qualn"] = ctx.empty((n, w.umi_len), np.uint8)
    shard["feature"] = ctx.empty(n, np.uint32)
    shard["keys"] = ctx.empty(n, np.uint64)
    ctx.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    chunk = 1 << 27
    for off in range(0, n, chunk):
        m = min(chunk, n - off)
        ctx.synth(w, off, m, cb=shard["cb"].ptr + off * 4, cb_qualn=shard["cb_qualn"].ptr + off * w.cb_len,
                  umi=shard["umi"].ptr + off * 4, umi_qualn=shard["umi_qualn"].ptr + off * w.umi_len,
                  feature=shard["feature"].ptr + off * 4, flags=shard["flags"].ptr + off)
    ctx.synchronize()
    be = HipBackend(ctx, 0)
    return ctx, shard, be, CountPipeline(be, libs=(0,))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    w = S.Workload(n_total=n, seed=S.SEED0 + 3)
    L = [make_lane(w, n) for _ in range(lanes)]

    def run(lane, k):
        ctx, shard, be, pipe = lane
        for _ in range(k):
            be.reset()
            r = pipe.run(shard)
            del r
        ctx.synchronize()

    for lane in L:
        run(lane, 1)  # warm-up, one lane at a time
    for name, group in (("one lane", L[:1]), ("%d lanes" % lanes, L)):
        th = [threading.Thread(target=run, args=(lane, steps)) for lane in group]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        total = steps * len(group)
        print("%-8s %d steps of %d reads in %.1f ms: %.2f ms per step, %.2f G reads/s" % (name, total, n, dt * 1e3, dt * 1e3 / total,
                                                                                         n * total / dt / 1e9), flush=True)


main()
