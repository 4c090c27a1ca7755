#!/usr/bin/env python3
"""bench.py -- throughput of the barcode-correct -> UMI-dedup -> count hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2] [--reads-per-gpu R]

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE).  A step is one pass of the hot path
over one batch of synthetic reads that are ALREADY resident in HBM (2-bit packed SoA, generated on
the device by the seeded integer generator of libcrgpu):

  cfg3 (default)  1 B post-alignment records per GPU: K1 exact match + histogram -> [C1 all-reduce] ->
                  K2 posterior correction -> molecule keys -> [C2 all-to-all] -> radix sort ->
                  UMI correction / low support / counting -> [C3 gather] -> CSC on the device
  cfg2            100 M reads per GPU, barcode correction only (K1 -> [C1] -> K2)

Weak scaling: every rank holds the same number of reads of ONE GEM well.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); set before any HIP call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# algorithmic bytes per unit of the whole step
STEP_BYTES = {"cfg2": 24, "cfg3": 60}
# per-kernel compulsory bytes per element of ONE launch (DESIGN.md "Kernels")
KERNEL_BYTES = {
    "match": ("k_match", 9),            # 4 B packed CB + 1 B flags in, 4 B index out, per read
    "correct": ("k_collect_miss+k_correct", 4),   # the miss scan reads every index once
    "keys": ("k_build_keys", 33),       # idx 4 + umi 4 + umi qual 12 + feature 4 + flags 1 in, key 8 out
    "sort_scatter": ("k_radix_scatter", 16),   # key 8 B in + 8 B out per pass
    "sort_hist": ("k_radix_hist", 8),   # key 8 B in per pass
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3"])
    ap.add_argument("--reads-per-gpu", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=24_000_000,
                    help="reads of the CPU-baseline sample (about 10-30 s of oracle work on the box's cores)")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed full-size property checks (cfg3, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(w, workload, sample):
    """The oracle (C restatement of the reference, hash-map based, the reference's own parallel shape:
    read chunks for correction, barcode groups fanned out to threads for dedup) timed on this box's
    host cores on a bounded sample of the same workload.  A reported baseline, not the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S

    cores = len(os.sched_getaffinity(0))
    r = w.host_reads(0, sample)
    cb, cbq = S.to_ascii(r["cb"], r["cb_qualn"], w.cb_len)
    reads = dict(cb=cb, cb_qual=cbq, lib=r["flags"] & 0x0F)
    if workload == "cfg3":
        umi, uq = S.to_ascii(r["umi"], r["umi_qualn"], w.umi_len)
        reads.update(umi=umi, umi_qual=uq, feature=r["feature"])
    wl = O.Whitelist(E.unpack_seqs(w.wl_packed, w.cb_len))
    t0 = time.perf_counter()
    O.run_pipeline(reads, [wl], n_threads=cores, count=(workload == "cfg3"))
    dt = time.perf_counter() - t0
    return {
        "value": sample / dt / 1e6,
        "unit": "M reads/s",
        "cores": cores,
        "kind": "port",
        "sample": "first %d reads of the same synthetic %s stream, oracle/ (C, OpenMP, %d threads), %.1f s"
                  % (sample, workload, cores, dt),
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd.pipeline import CountPipeline, HipBackend

    workload = args.workload
    n = args.reads_per_gpu or (1_000_000_000 if workload == "cfg3" else 100_000_000)
    n_total = n * world
    w = S.Workload(n_total=n_total, seed=S.SEED0 + (3 if workload == "cfg3" else 2))

    ctx = E.Context(local_rank)
    ctx.set_whitelist(0, w.wl_packed, length=w.cb_len)
    shard = dict(n=n, umi_len=w.umi_len)
    shard["cb"] = ctx.empty(n, np.uint32)
    shard["cb_qualn"] = ctx.empty((n, w.cb_len), np.uint8)
    shard["flags"] = ctx.empty(n, np.uint8)
    shard["idx"] = ctx.empty(n, np.uint32)
    if workload == "cfg3":
        shard["umi"] = ctx.empty(n, np.uint32)
        shard["umi_qualn"] = ctx.empty((n, w.umi_len), np.uint8)
        shard["feature"] = ctx.empty(n, np.uint32)
        shard["keys"] = ctx.empty(n, np.uint64)
        ctx.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    # this rank's slice of the job's read stream, generated straight into HBM
    chunk = 1 << 27
    for off in range(0, n, chunk):
        m = min(chunk, n - off)

        def sl(name, per=1, itemsize=1):
            a = shard.get(name)
            return None if a is None else a.ptr + off * per * itemsize

        ctx.synth(w, rank * n + off, m, cb=sl("cb", 1, 4), cb_qualn=sl("cb_qualn", w.cb_len), umi=sl("umi", 1, 4),
                  umi_qualn=sl("umi_qualn", w.umi_len), feature=sl("feature", 1, 4), flags=sl("flags"))
    ctx.synchronize()

    be = HipBackend(ctx, local_rank)
    pipe = CountPipeline(be, libs=(0,), dist=dist if world > 1 else None)

    def step():
        be.reset()
        if workload == "cfg2":
            pipe.correct_barcodes(shard)
            return None
        return pipe.run(shard)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    ctx.timing_reset()
    ctx.timing(True)
    sync()
    t0 = time.perf_counter()
    result = None
    step_marks = []
    for _ in range(args.steps):
        result = None      # the previous step's matrix goes back to the device pool first: steady state allocates nothing
        result = step()
        step_marks.append(time.perf_counter())   # host-side marks only (steps end in a blocking read of the totals)
    sync()
    dt = time.perf_counter() - t0
    ctx.timing(False)
    ledger = ctx.timing_get()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out_info = {}
    if workload == "cfg3" and rank == 0 and result is not None:
        out_info = {"matrix_columns": result.n_barcodes, "matrix_nnz": result.nnz}
    if workload == "cfg2" and rank == 0:
        idx = shard["idx"].to_host(count=min(n, 1 << 22))
        out_info = {"valid_frac_sample": float((idx != 0xFFFFFFFF).mean())}

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_total * args.steps / dt / 1e6
        # dominant kernel family by device time; per-launch algorithmic bytes / average launch duration
        fam = {k: v for k, v in ledger.items() if k in KERNEL_BYTES and v[1] > 0}
        roof = None
        if fam:
            name = max(fam, key=lambda k: fam[k][0])
            ms, launches, units = fam[name]
            kname, bpe = KERNEL_BYTES[name]
            # algorithmic bytes of the timed launches (bytes per element x elements the ledger counted)
            # over their summed HIP-event durations == per-launch bytes / average launch duration
            achieved = bpe * units / (ms * 1e-3) / 1e9
            # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
            # profiles/r01_m_*_pmc_fetch_write.json): measured bytes per element x elements per launch
            traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", "r01_m_cfg3_1B_pmc_fetch_write.json")) as f:
                    pmc = json.load(f)["derived"]
                if kname == "k_radix_scatter":
                    traffic = pmc["k_radix_scatter_hbm_bytes_per_element_weighted"] * units / launches
            except (OSError, KeyError, ValueError):
                traffic = None
            roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": ms / launches,
                    "launches_per_step": launches / args.steps, "bytes_per_element": bpe,
                    "elements_per_launch": units / launches,
                    # a plain 8 GiB device-to-device copy on this box (profiles/r01_m_copy_rates.txt): context, not the peak
                    "copy_rate_measured_GBps": 4835.0}
        step_gbs = STEP_BYTES[workload] * n / (ms_per_step * 1e-3) / 1e9
        line = {
            "metric": "M reads/sec barcode-correct+UMI-count" if workload == "cfg3" else "M reads/sec barcode-correct",
            "value": value,
            "unit": "M reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32/u64 integer keys; f64 posterior",
            "data": "synthetic",
            "config": {"workload": "%s: %d reads/GPU, 16 bp CB + 12 bp UMI, 737280-entry whitelist, 10k cells, "
                                   "36601 features%s" % (workload, n, "" if workload == "cfg3" else ", barcode correction only"),
                       "reads_per_gpu": n, "parallelism": "read-sharded x%d" % world},
            "roofline": roof,
            "step_roofline": {"basis_bytes_per_read": STEP_BYTES[workload], "achieved": step_gbs, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in ledger.items() if v[1]},
            "kernel_launches_per_step": {k: v[1] / args.steps for k, v in ledger.items() if v[1]},
            "output": out_info,
            "host_step_marks_ms": [round((m - t0) * 1e3, 2) for m in step_marks],
        }
        if world == 1 and workload == "cfg3" and not args.no_verify:
            # not timed: the laws of cellranger_amd/selfcheck.py on this very workload at its full size
            from cellranger_amd import selfcheck
            try:
                line["verify"] = dict(selfcheck.full_size_properties(ctx, shard, local_rank), ok=True)
            except AssertionError as e:  # report, never hide: the line still carries the timing
                import traceback
                line["verify"] = {"ok": False, "failed": traceback.format_exc(limit=2).strip().splitlines()[-3:], "msg": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, workload, min(args.cpu_sample, n))
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
