#!/usr/bin/env python3
"""bench.py -- throughput of the barcode-correct -> UMI-dedup -> count hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2] [--reads-per-gpu R] [--whitelist W] [--dupinfo]

One process per GPU.  Under torchrun (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) this process is one
rank; WITHOUT them `--gpus N` (N > 1) makes this process a launcher that starts N fresh rank processes itself (before
anything touches the GPU), relays rank 0's JSON line and fails if any rank fails.  `--gpus` must equal the world size.

A step is one pass of the hot path over one batch of synthetic reads that are ALREADY resident in HBM (2-bit packed SoA,
generated on the device by the seeded integer generator of libcrgpu):

  cfg3 (default)  1 B post-alignment records per GPU: K1 exact match + histogram -> [C1 all-reduce] ->
                  K2 posterior correction -> molecule keys -> [C2 key exchange] -> radix sort ->
                  UMI correction / low support / counting -> [C3 gather] -> CSC on the device
  cfg2            100 M reads per GPU, barcode correction only (K1 -> [C1] -> K2)

C1/C2/C3 are libcrgpu's own collectives (RCCL over xGMI, csrc/comm.hip).  Weak scaling: every rank holds the same number
of reads of ONE GEM well.  Prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); set before any HIP call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# algorithmic bytes per unit of the whole step
STEP_BYTES = {"cfg2": 24, "cfg3": 60}
# per-family compulsory bytes per element (DESIGN.md "Kernels"): the family with the largest device time is the line's
# `roofline`; (kernels of the family, bytes per unit, what a unit is)
KERNEL_BYTES = {
    "match": ("k_lookup_hot+k_stage_idx+k_hist_buckets", 9, "read: 4 B packed CB + 1 B flags in, 4 B index out"),
    "correct": ("k_correct_records", 4, "read: the index every read gets (the misses' 29 B are inside it)"),
    "keys": ("k_build_keys", 33, "record: idx 4 + umi 4 + umi qual 12 + feature 4 + flags 1 in, key 8 out"),
    "sort_scatter": ("k_radix_scatter", 16, "key per pass: 8 B in + 8 B out"),
    "sort_hist": ("k_finish_runs (+ k_global_hist / k_radix_hist when they run)", 8, "key: 8 B in (the keys that move are written back)"),
    "dedup": ("dedup family (run lengths, UMI correction, low support, molecules, triplets)", 8,
              "sorted key: 8 B in, once (outputs, a few per cent of it, not counted)"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3"])
    ap.add_argument("--reads-per-gpu", type=int, default=0)
    ap.add_argument("--whitelist", type=int, default=737280,
                    help="whitelist entries: 737280 (737K-august-2016) or 6794880 (3M-february-2018, the SC3Pv3 list)")
    ap.add_argument("--dupinfo", action="store_true",
                    help="cfg3: time crgpu_count_records_dev (per-read DupInfo for the BAM tags) instead of the keys-only count")
    ap.add_argument("--cpu-sample", type=int, default=64_000_000,
                    help="reads of the CPU-baseline sample (about 10-30 s of oracle work on the box's cores)")
    ap.add_argument("--cpu-1thread-sample", type=int, default=1_000_000)
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed full-size property checks (cfg3, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    return ap.parse_args()


def launch_ranks(n):
    """No torchrun: this process only starts the n rank processes (it has not imported torch or touched the GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile

    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # a rank that dies must not leave the others waiting at the rendezvous: the first failure ends them all
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.kill()       # exactly the processes started above
                    codes[r] = p.wait()
            break
        time.sleep(0.2)
    out0.seek(0)
    sys.stdout.write(out0.read().decode(errors="replace"))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        return 1
    return 0


def cpu_baseline(w, workload, sample, sample_1t):
    """The oracle (C restatement of the reference, hash-map based, the reference's own parallel shape: read chunks for
    correction, barcode groups fanned out to worker threads for dedup) timed on this box's host cores on a bounded sample
    of the same workload.  A reported baseline, not the target.  `value` excludes the step that only puts the reads into
    barcode order (the reference reads them from barcode-sorted shards, outside its hot path); the stages are listed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S

    # host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota when there is one (the
    # GPU boxes expose 256 hardware threads but give a one-GPU job the share of 16; 256 OpenMP threads on 16 cores' worth of
    # time only fight each other)
    affinity = len(os.sched_getaffinity(0))
    cores = affinity
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = max(1, min(affinity, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    if os.environ.get("CRGPU_CPU_THREADS"):
        cores = int(os.environ["CRGPU_CPU_THREADS"])

    def run(n_reads, threads):
        r = w.host_reads(0, n_reads)
        cb, cbq = S.to_ascii(r["cb"], r["cb_qualn"], w.cb_len)
        reads = dict(cb=cb, cb_qual=cbq, lib=r["flags"] & 0x0F)
        if workload == "cfg3":
            umi, uq = S.to_ascii(r["umi"], r["umi_qualn"], w.umi_len)
            reads.update(umi=umi, umi_qual=uq, feature=r["feature"])
        wl = O.Whitelist(E.unpack_seqs(w.wl_packed, w.cb_len))
        t0 = time.perf_counter()
        O.run_pipeline(reads, [wl], n_threads=threads, count=(workload == "cfg3"))
        wall = time.perf_counter() - t0
        st = O.last_timing()
        correct = st["pass_a"] + st["hist_join"] + st["pass_b"] + st["corrected_join"]
        dedup = (st["dedup"] + st["assembly"]) if workload == "cfg3" else 0.0
        return wall, correct, dedup, st

    wall, correct, dedup, st = run(sample, cores)
    out = {
        "value": sample / (correct + dedup) / 1e6,
        "unit": "M reads/s",
        "cores": cores,
        "threads_used": cores,
        "hardware_threads_visible": affinity,
        "kind": "port",
        "correct_M_reads_per_s": sample / correct / 1e6,
        "dedup_M_reads_per_s": (sample / dedup / 1e6) if dedup else None,
        "stages_s": {k: round(v, 3) for k, v in st.items()},
        "wall_s_incl_ordering_and_marshalling": round(wall, 2),
        "sample": "first %d reads of the same synthetic %s stream, oracle/ (C, OpenMP, %d threads): correction %.1f s, "
                  "dedup+matrix %.1f s; bringing the reads into barcode order (%.1f s, shardio's job in the reference) is "
                  "not in `value`" % (sample, workload, cores, correct, dedup, st["barcode_order"]),
    }
    if sample_1t:
        wall1, c1, d1, _ = run(sample_1t, 1)
        out["cpu_1thread"] = {"value": sample_1t / (c1 + d1) / 1e6, "unit": "M reads/s", "cores": 1,
                              "sample": "first %d reads, one thread: correction %.1f s, dedup+matrix %.1f s" % (sample_1t, c1, d1)}
    return out


def end_to_end(ctx, w, n_gpu, step_s, sample):
    """the rate including the hand-over of host buffers -- R1 rows (28 bases + 28 qualities per read) go up
    over PCIe from pinned memory and are sliced / 2-bit packed on the device (crgpu_pack_rows_dev); feature and flags
    (5 B/read) ride along.  PCIe-bound; reported beside `value`, never as `value`."""
    import numpy as np
    import torch

    row = w.cb_len + w.umi_len
    m = min(sample, n_gpu)
    h_seq = torch.empty((m, row), dtype=torch.uint8).pin_memory()
    h_qual = torch.empty((m, row), dtype=torch.uint8).pin_memory()
    h_seq.fill_(ord("A"))
    h_qual.fill_(70)
    d_seq = torch.empty((m, row), dtype=torch.uint8, device="cuda")
    d_qual = torch.empty((m, row), dtype=torch.uint8, device="cuda")
    d_cb, d_umi = ctx.empty(m, np.uint32), ctx.empty(m, np.uint32)
    d_cbq, d_uq = ctx.empty((m, w.cb_len), np.uint8), ctx.empty((m, w.umi_len), np.uint8)
    d_fl = ctx.zeros(m, np.uint8)

    def once():
        d_seq.copy_(h_seq, non_blocking=True)
        d_qual.copy_(h_qual, non_blocking=True)
        torch.cuda.synchronize()
        ctx.pack_rows(d_seq, d_qual, m, row, 0, w.cb_len, d_cb, d_cbq, d_fl)
        ctx.pack_rows(d_seq, d_qual, m, row, w.cb_len, w.umi_len, d_umi, d_uq, None)
        ctx.synchronize()

    once()
    t0 = time.perf_counter()
    once()
    dt = time.perf_counter() - t0
    ingest_s = dt * n_gpu / m           # the whole batch at the sample's rate
    return {"value": n_gpu / (ingest_s + step_s) / 1e6, "unit": "M reads/s",
            "ingest_M_reads_per_s": m / dt / 1e6, "h2d_plus_pack_GBps": 2 * row * m / dt / 1e9,
            "sample": "%d reads: 2 x %d B rows pinned host -> device + crgpu_pack_rows_dev (CB, UMI), scaled to the batch; "
                      "serial with the step (no overlap)" % (m, row)}


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None or "RANK" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus))
        world, rank, local_rank = 1, 0, 0
    else:
        world, rank = int(env_world), int(os.environ["RANK"])
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != args.gpus:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
            sys.exit(2)

    import numpy as np
    import torch

    if os.environ.get("CRGPU_BENCH_SHARE_DEVICE"):
        # rehearsal on a box with fewer GPUs than ranks: the ranks share the devices there are (RCCL must agree to it)
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd.pipeline import CountPipeline, HipBackend

    uid = None
    if world > 1:
        # the 128-byte RCCL id travels from rank 0 to the others over a CPU (gloo) group; everything on the data path
        # then runs on libcrgpu's own RCCL communicator
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one node: the id exchange stays on the loopback interface (the container's hostname need not resolve)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        box = [E.get_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]

    workload = args.workload
    n = args.reads_per_gpu or (1_000_000_000 if workload == "cfg3" else 100_000_000)
    n_total = n * world
    w = S.Workload(n_total=n_total, seed=S.SEED0 + (3 if workload == "cfg3" else 2), n_wl=args.whitelist)

    ctx = E.Context(local_rank, n_ranks=world, rank=rank, unique_id=uid)
    # the bench writes its buffers only through the context: K2 may use K1's miss records, the sort the key histograms
    ctx.trust_unchanged_buffers(True)
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
        if not args.dupinfo:
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
    pipe = CountPipeline(be, libs=(0,))
    dup_out = None
    if args.dupinfo and workload == "cfg3":
        dup_out = (ctx.empty(n, np.uint32), ctx.empty(n, np.uint32), ctx.empty(n, np.uint8))

    def step():
        be.reset()
        if workload == "cfg2":
            pipe.correct_barcodes(shard)
            return None
        # dup_out: the drop-in's path -- per-read DupInfo for the UB / duplicate-flag / xf tags (tx_annotation/src/read.rs:536-590)
        return pipe.run(shard, dupinfo=dup_out)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        ctx.barrier()

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
    dt = ctx.allreduce_max(dt)  # MAX over the ranks

    out_info = {}
    if workload == "cfg3" and rank == 0 and result is not None:
        out_info = {"matrix_columns": result.n_barcodes, "matrix_nnz": result.nnz}
    if workload == "cfg2" and rank == 0:
        idx = shard["idx"].to_host(count=min(n, 1 << 22))
        out_info = {"valid_frac_sample": float((idx != 0xFFFFFFFF).mean())}

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_total * args.steps / dt / 1e6
        # dominant kernel family by device time (HIP events on the context's stream, the library's own ledger);
        # algorithmic bytes of the timed launches / their summed durations == per-launch bytes / average launch duration
        # (a family that ran for less than 0.1 ms per step is bookkeeping -- e.g. the scan of the digit histograms that
        # k_build_keys already counted -- not a candidate)
        fam = {k: v for k, v in ledger.items() if k in KERNEL_BYTES and v[1] > 0 and v[2] > 0 and v[0] / args.steps >= 0.1}
        roof = None
        if fam:
            name = max(fam, key=lambda k: fam[k][0])
            ms, launches, units = fam[name]
            kname, bpe, unit_desc = KERNEL_BYTES[name]
            achieved = bpe * units / (ms * 1e-3) / 1e9
            # HBM bytes per launch are NOT measured in this run (PMC passes need rocprofv3): the figure of the committed
            # PMC profile of this round is attached only when it was taken on this workload and size
            traffic, traffic_src = None, None
            prof = os.path.join(ROOT, "profiles", "r02_cfg3_1B_pmc_fetch_write.json")
            if workload == "cfg3" and n == 1_000_000_000 and args.whitelist == 737280 and os.path.exists(prof):
                try:
                    with open(prof) as f:
                        per_el = json.load(f)["derived"].get(name + "_hbm_bytes_per_element")
                    if per_el:
                        traffic, traffic_src = per_el * units / launches, "replayed from profiles/" + os.path.basename(prof)
                except (OSError, KeyError, ValueError):
                    pass
            roof = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": ms / launches, "launches_per_step": launches / args.steps,
                    "family_ms_per_step": ms / args.steps, "bytes_per_element": bpe, "element": unit_desc,
                    "elements_per_step": units / args.steps}
        fam_fracs = {}
        for k, (ms, launches, units) in fam.items():
            fam_fracs[k] = round(KERNEL_BYTES[k][1] * units / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
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
            "config": {"workload": "%s: %d reads/GPU, 16 bp CB + 12 bp UMI, %d-entry whitelist, 10k cells, "
                                   "36601 features%s%s" % (workload, n, args.whitelist,
                                                           "" if workload == "cfg3" else ", barcode correction only",
                                                           ", per-read DupInfo" if dup_out is not None else ""),
                       "reads_per_gpu": n, "parallelism": "read-sharded x%d%s" % (
                           world, ", libcrgpu RCCL collectives (C1 all-reduce, C2 key exchange, C3 gather)" if world > 1 else "")},
            "roofline": roof,
            "family_roofline_frac": fam_fracs,
            "step_roofline": {"basis_bytes_per_read": STEP_BYTES[workload], "achieved": step_gbs, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in ledger.items() if v[1]},
            "kernel_launches_per_step": {k: v[1] / args.steps for k, v in ledger.items() if v[1]},
            "kernel_units_per_step": {k: v[2] / args.steps for k, v in ledger.items() if v[1] and v[2]},
            "output": out_info,
            "host_step_marks_ms": [round((m - t0) * 1e3, 2) for m in step_marks],
        }
        if world == 1 and workload == "cfg3" and not args.no_verify and dup_out is None:
            # not timed: the laws of cellranger_amd/selfcheck.py on this very workload at its full size
            from cellranger_amd import selfcheck
            try:
                line["verify"] = dict(selfcheck.full_size_properties(ctx, shard, local_rank), ok=True)
            except AssertionError as e:  # report, never hide: the line still carries the timing
                import traceback
                line["verify"] = {"ok": False, "failed": traceback.format_exc(limit=2).strip().splitlines()[-3:], "msg": str(e)}
        if world == 1 and workload == "cfg3" and not args.no_end_to_end:
            result = None
            be.reset()
            try:
                line["end_to_end"] = end_to_end(ctx, w, n, ms_per_step * 1e-3, 64_000_000)
            except Exception as e:  # noqa: BLE001 - an extra figure must never cost the line
                line["end_to_end"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w, workload, min(args.cpu_sample, n), min(args.cpu_1thread_sample, n))
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        ctx.barrier()
    ctx.close()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
