#!/usr/bin/env python3
"""bench.py -- throughput of the barcode-correct -> UMI-dedup -> count hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2|cfg4|cfg5] [--reads-per-gpu R] [--whitelist W] [--dupinfo]

One process per GPU.  Under torchrun (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) this process is one
rank; WITHOUT them `--gpus N` (N > 1) makes this process a launcher that starts N fresh rank processes itself (before
anything touches the GPU), relays rank 0's JSON line and fails if any rank fails.  `--gpus` must equal the world size.

A step is one pass of the hot path over one batch of synthetic reads that are ALREADY resident in HBM (2-bit packed SoA,
generated on the device by the seeded integer generator of libcrgpu):

  cfg3 (default)  1 B post-alignment records per GPU: K1 exact match + histogram -> [C1 all-reduce] ->
                  K2 posterior correction -> molecule keys -> [C2 key exchange] -> radix sort ->
                  UMI correction / low support / counting -> [C3 gather] -> CSC on the device
  cfg2            100 M reads per GPU, barcode correction only (K1 -> [C1] -> K2)
  cfg4            Feature Barcoding (BASELINE configs[3]): per GPU R Antibody Capture reads (default 500 M at N = 1, 62.5 M
                  otherwise) whose barcodes go through a Trans whitelist and whose features are extracted from whole R2 rows by
                  the anchored pattern ^N{10}(BC) against a 200-feature 15-mer reference (first without a distribution for
                  MAKE_SHARD's exact-match counts, then with it), plus R / 4 Gene Expression reads of the same GEM well;
                  both libraries are counted together
  cfg5            multi / aggr style (BASELINE configs[4]): ONE WHOLE GEM well of 500 M reads per GPU (gem group = rank + 1),
                  no C1 / C2; the per-well CSC blocks are gathered on rank 0 (C3) into the merged matrix

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
STEP_BYTES = {"cfg2": 24, "cfg3": 60, "cfg5": 60, "cfg4": 86}  # cfg4: per FB read; its GEX reads count 60
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
    "feature": ("k_extract_tethered_lds (+ k_feature_counts)", 30, "FB read per pass: 15 captured bases + 15 qualities (SURVEY 8d)"),
}
# the committed PMC passes of THIS round's code (scripts/pmc_families.sh), per workload at its default size
PMC_PROFILE = {"cfg3": "r03_cfg3_1B_pmc_fetch_write.json", "cfg4": "r03_cfg4_500M_pmc_fetch_write.json"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--fb-row-stride", type=int, default=92, help="cfg4: bytes per R2 row (90-base read + padding to a dword)")
    ap.add_argument("--fb-features", type=int, default=200)
    ap.add_argument("--dense-keys", type=int, default=-1,
                    help="CRGPU_OPT_DENSE_BARCODE_KEYS: 1 / 0; default 0 (measured on the 3M list: the rank -> column gather costs the "
                         "key builder 6 ms per 1 B reads, more than the radix pass it saves; the option is for layouts beyond 64 bits)")
    ap.add_argument("--no-default-options", action="store_true",
                    help="skip the short second timed loop with CRGPU_OPT_BUFFERS_UNCHANGED_BETWEEN_CALLS off")
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


def cpu_baseline_cfg4(cfg4, sample):
    """cfg4's CPU baseline: the oracle on a bounded sample of the Feature Barcoding stream on this box's host cores --
    barcode stage through the Trans whitelist, match_read over the R2 rows twice (exact-match counts, then with the
    distribution), dedup + matrix.  A reported baseline, not the target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S

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
    w = cfg4.w_fb
    r = w.host_reads(0, sample)
    rows_s, rows_q = E.synth_rows_host(cfg4.rows_seed, 0, sample, r["feature"], cfg4.feats, cfg4.L, cfg4.OFFSET, cfg4.stride)
    t0 = time.perf_counter()
    ox0 = O.FeatureExtractor(cfg4.defs, None)
    f0 = ox0.match_rows(rows_s, rows_q, read=1, n_threads=cores)
    counts = np.bincount(f0[f0 != 0xFFFFFFFF], minlength=cfg4.n_feat_all).astype(np.int64)
    dist = O.compute_feature_dist(counts, cfg4.types)
    ox1 = O.FeatureExtractor(cfg4.defs, dist)
    f1 = ox1.match_rows(rows_s, rows_q, read=1, n_threads=cores)
    t_feat = time.perf_counter() - t0
    cb, cbq = S.to_ascii(r["cb"], r["cb_qualn"], 16)
    umi, uq = S.to_ascii(r["umi"], r["umi_qualn"], 12)
    reads = dict(cb=cb, cb_qual=cbq, umi=umi, umi_qual=uq, feature=f1, lib=np.zeros(sample, np.uint8))
    # one library in the sample: its Trans whitelist (raw FB barcodes -> the canonical list)
    raw = w.wl_packed
    wl = O.Whitelist(E.unpack_seqs(raw, 16), translated=E.unpack_seqs(cfg4.w.wl_packed, 16))
    O.run_pipeline(reads, [wl], n_threads=cores, count=True)
    st = O.last_timing()
    correct = st["pass_a"] + st["hist_join"] + st["pass_b"] + st["corrected_join"]
    dedup = st["dedup"] + st["assembly"]
    return {"value": sample / (t_feat + correct + dedup) / 1e6, "unit": "M reads/s", "cores": cores, "kind": "port",
            "stages_s": {"feature_extraction_two_passes": round(t_feat, 3), "barcode_correction": round(correct, 3),
                         "dedup_matrix": round(dedup, 3), "barcode_order_not_counted": round(st["barcode_order"], 3)},
            "sample": "first %d Antibody Capture reads of the same stream (%d-byte R2 rows, Trans whitelist), oracle/ (C, OpenMP, "
                      "%d threads)" % (sample, cfg4.stride, cores)}


def end_to_end(ctx, w, n_gpu, step_s, k1_s, sample):
    """the rate including the hand-over of host buffers.  The sample's REAL R1 rows (28 bases + 28 qualities per
    read, ASCII as the FASTQ holds them) plus feature and flags (5 B/read) go up over PCIe from pinned memory in chunks on a
    copy stream; while chunk k + 1 travels, chunk k is sliced / 2-bit packed (crgpu_pack_rows_dev) and goes through pass A
    (K1) on the context's stream -- the part of the step that needs no global state -- so the ingest hides K1 and the rest
    of the step (K2 onwards needs the global prior) follows it.  PCIe-bound; reported beside `value`, never as `value`."""
    import numpy as np
    import torch

    from cellranger_amd import synth as S

    row = w.cb_len + w.umi_len
    m = min(sample, n_gpu)
    r = w.host_reads(0, m)
    cb, cbq = S.to_ascii(r["cb"], r["cb_qualn"], w.cb_len)
    umi, uq = S.to_ascii(r["umi"], r["umi_qualn"], w.umi_len)
    h_seq = torch.from_numpy(np.ascontiguousarray(np.hstack([cb, umi]))).pin_memory()
    h_qual = torch.from_numpy(np.ascontiguousarray(np.hstack([cbq, uq]))).pin_memory()
    h_ft = torch.from_numpy(r["feature"].view(np.int32)).pin_memory()
    h_fl = torch.from_numpy(r["flags"]).pin_memory()
    del r, cb, cbq, umi, uq
    chunk = min(m, 8_000_000)
    bufs = [dict(seq=torch.empty((chunk, row), dtype=torch.uint8, device="cuda"), qual=torch.empty((chunk, row), dtype=torch.uint8, device="cuda"),
                 ft=torch.empty(chunk, dtype=torch.int32, device="cuda"), fl=torch.empty(chunk, dtype=torch.uint8, device="cuda"),
                 up=torch.cuda.Event(), done=torch.cuda.Event()) for _ in range(2)]
    d_cb, d_umi = ctx.empty(m, np.uint32), ctx.empty(m, np.uint32)
    d_cbq, d_uq = ctx.empty((m, w.cb_len), np.uint8), ctx.empty((m, w.umi_len), np.uint8)
    d_idx = ctx.empty(m, np.uint32)
    copy_stream = torch.cuda.Stream()
    ext = torch.cuda.ExternalStream(ctx.stream_handle())

    def once():
        ctx.reset_counts()
        for k, off in enumerate(range(0, m, chunk)):
            b = bufs[k & 1]
            c = min(chunk, m - off)
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(b["done"])          # the buffer's previous chunk has been packed
                b["seq"][:c].copy_(h_seq[off:off + c], non_blocking=True)
                b["qual"][:c].copy_(h_qual[off:off + c], non_blocking=True)
                b["ft"][:c].copy_(h_ft[off:off + c], non_blocking=True)
                b["fl"][:c].copy_(h_fl[off:off + c], non_blocking=True)
                b["up"].record(copy_stream)
            ext.wait_event(b["up"])
            ctx.pack_rows(b["seq"], b["qual"], c, row, 0, w.cb_len, d_cb.ptr + 4 * off, d_cbq.ptr + w.cb_len * off, b["fl"])
            ctx.pack_rows(b["seq"], b["qual"], c, row, w.cb_len, w.umi_len, d_umi.ptr + 4 * off, d_uq.ptr + w.umi_len * off, None)
            ctx.match_and_count(d_cb.ptr + 4 * off, b["fl"], c, d_idx.ptr + 4 * off)
            b["done"].record(ext)
        ctx.synchronize()
        torch.cuda.synchronize()

    once()
    t0 = time.perf_counter()
    once()
    dt = time.perf_counter() - t0
    ingest_s = dt * n_gpu / m           # the whole batch at the sample's rate, K1 inside
    rest_s = max(step_s - k1_s, 0.0)
    bytes_up = (2 * row + 5) * m
    return {"value": n_gpu / (ingest_s + rest_s) / 1e6, "unit": "M reads/s",
            "ingest_M_reads_per_s": m / dt / 1e6, "h2d_GBps": bytes_up / dt / 1e9,
            "serial_value": n_gpu / (ingest_s + step_s) / 1e6,
            "sample": "%d reads of the synthetic stream as ASCII R1 rows (2 x %d B) + feature + flags from pinned host memory in "
                      "%d-read chunks, double-buffered: H2D of chunk k+1 on a copy stream beside crgpu_pack_rows_dev + pass A of "
                      "chunk k; scaled to the batch, then the rest of the step (%.1f ms of %.1f ms)" % (m, row, chunk, rest_s * 1e3, step_s * 1e3)}


class Cfg4:
    """BASELINE configs[3] on this rank: an Antibody Capture library (library 1, Trans whitelist, features from R2 rows) beside
    a Gene Expression library (library 0) of the same GEM well; see the module docstring."""

    L, OFFSET = 15, 10
    PATTERN = "5PNNNNNNNNNN(BC)"

    def __init__(self, ctx, args, n_fb, world, rank, E, S, np):
        self.ctx, self.np, self.E = ctx, np, E
        self.n_fb, self.n_gex = n_fb, n_fb // 4
        self.world, self.rank = world, rank
        self.stride = args.fb_row_stride
        self.n_genes, self.n_fbf = 36601, args.fb_features
        n_wl = args.whitelist
        seed = S.SEED0 + 4
        rng = np.random.Generator(np.random.PCG64(seed + 12345))   # not the workload's own stream
        self.w = S.Workload(n_total=self.n_gex * world, seed=seed, n_wl=n_wl)
        canon = self.w.wl_packed
        # Trans whitelist: a second list of the same size, raw[i] pairs with canon[translate_to[i]] (a pairing permutation)
        raw = np.unique(rng.integers(0, 1 << 32, size=int(n_wl * 1.3) + 64, dtype=np.uint64)).astype(np.uint32)
        raw = np.setdiff1d(raw, canon)
        raw = rng.permutation(raw)[:n_wl]
        assert len(raw) == n_wl
        translate_to = rng.permutation(n_wl).astype(np.uint32)
        inv = np.empty(n_wl, np.uint32)
        inv[translate_to] = np.arange(n_wl, dtype=np.uint32)
        self.w_fb = S.Workload(n_total=self.n_fb * world, seed=seed, n_wl=n_wl, n_genes=self.n_fbf)   # same cells
        self.w_fb.wl_packed[:] = raw[inv]
        self.w_fb.c.seed = seed + 1000
        ctx.set_whitelist(0, canon, length=16)
        ctx.set_whitelist(1, raw, canon=canon, translate_to=translate_to, length=16)
        feats = np.unique(rng.integers(0, 1 << 30, size=4 * self.n_fbf, dtype=np.uint64))
        feats = rng.permutation(feats)[:self.n_fbf]
        self.feats = feats
        fa = [bytes(x).decode() for x in E.unpack_seqs(feats.astype(np.uint32), self.L)]
        self.defs = [(self.PATTERN, fa[k], self.n_genes + k, 1) for k in range(self.n_fbf)]
        self.n_feat_all = self.n_genes + self.n_fbf
        self.types = np.array([0] * self.n_genes + [1] * self.n_fbf, np.uint32)
        ctx.set_key_layout(self.n_feat_all, 12, 2, 0)
        self.libs = []
        for lib, (w, n) in enumerate(((self.w, self.n_gex), (self.w_fb, self.n_fb))):
            sh = dict(n=n, umi_len=12, cb=ctx.empty(n, np.uint32), cb_qualn=ctx.empty((n, 16), np.uint8), flags=ctx.empty(n, np.uint8),
                      idx=ctx.empty(n, np.uint32), umi=ctx.empty(n, np.uint32), umi_qualn=ctx.empty((n, 12), np.uint8),
                      feature=ctx.empty(n, np.uint32))
            chunk = 1 << 27
            for off in range(0, n, chunk):
                m = min(chunk, n - off)
                ctx.synth(w, rank * n + off, m, cb=sh["cb"].ptr + 4 * off, cb_qualn=sh["cb_qualn"].ptr + 16 * off,
                          umi=sh["umi"].ptr + 4 * off, umi_qualn=sh["umi_qualn"].ptr + 12 * off, feature=sh["feature"].ptr + 4 * off,
                          flags=sh["flags"].ptr + off)
            if lib == 1:
                import torch
                torch.as_tensor(sh["flags"], device="cuda").bitwise_or_(1)     # library id 1
            self.libs.append(sh)
        fb = self.libs[1]
        self.rows_s, self.rows_q = ctx.empty((n_fb, self.stride), np.uint8), ctx.empty((n_fb, self.stride), np.uint8)
        chunk = 1 << 26
        for off in range(0, n_fb, chunk):   # rows from the TRUE features of the stream (then forgotten: the step re-derives them)
            m = min(chunk, n_fb - off)
            ctx.synth_rows(seed + 7, rank * n_fb + off, m, fb["feature"].ptr + 4 * off, feats, self.L, self.OFFSET, self.stride,
                           self.rows_s.ptr + self.stride * off, self.rows_q.ptr + self.stride * off)
        self.rows_seed = seed + 7
        self.keys = ctx.empty(self.n_gex + n_fb, np.uint64)
        ctx.set_feature_extractor(1, self.defs)     # MAKE_SHARD's extractor: no distribution
        ctx.synchronize()

    def step(self, be, collective):
        ctx, np = self.ctx, self.np
        from cellranger_amd._lib import COUNTS_CORRECTED, COUNTS_VALID
        be.reset()
        gex, fb = self.libs
        for sh in (gex, fb):                    # one library per call: the one-library kernels
            be.match_and_count(sh)
        if collective:
            be.allreduce_hist([0, 1], COUNTS_VALID)
        rows = (self.rows_s, self.rows_q, None, self.stride)
        ctx.extract_features(1, self.n_fb, fb["feature"], r2=rows)
        counts = ctx.feature_counts(fb["feature"], self.n_fb, self.n_feat_all)
        if collective:
            ctx.allreduce_sum(counts)
        dist = self.E.compute_feature_dist(counts, self.types)
        ctx.set_feature_extractor(2, self.defs, dist)
        for sh in (gex, fb):
            be.correct(sh)
        if collective:
            be.allreduce_hist([0, 1], COUNTS_CORRECTED)
        ctx.extract_features(2, self.n_fb, fb["feature"], r2=rows)
        n_keys = 0
        for sh in (gex, fb):
            recs = ctx.records(sh["n"], 12, sh["idx"], sh["umi"], sh["umi_qualn"], sh["feature"], sh["flags"])
            n_keys += ctx.build_keys(recs, self.keys.ptr + 8 * n_keys)
        keys = self.keys
        if collective:
            keys, n_keys = be.exchange_keys(keys, n_keys)
        cnt = be.count_keys(keys, n_keys)
        if not collective:
            b, f, c = be.triplet_arrays(cnt)
            return be.assemble(b, f, c, cnt.n_triplets)
        arrs, total = be.gather_triplets(cnt)
        return be.assemble(arrs[0], arrs[1], arrs[2], total) if self.rank == 0 else None


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
    default_n = {"cfg2": 100_000_000, "cfg3": 1_000_000_000, "cfg5": 500_000_000,
                 "cfg4": 500_000_000 if world == 1 else 62_500_000}[workload]
    n = args.reads_per_gpu or default_n

    ctx = E.Context(local_rank, n_ranks=world, rank=rank, unique_id=uid)
    # the bench writes its buffers only through the context: K2 may use K1's miss records, the sort the key histograms
    ctx.trust_unchanged_buffers(True)
    dense_keys = False if args.dense_keys < 0 else bool(args.dense_keys)
    if dense_keys:
        ctx.set_option(1, 1)   # before the key layout is set
    be = HipBackend(ctx, local_rank)
    dup_out = None
    cfg4 = None
    key_layout_args = None
    if workload == "cfg4":
        cfg4 = Cfg4(ctx, args, n, world, rank, E, S, np)
        key_layout_args = (cfg4.n_feat_all, 12, 2, 0)
        w = cfg4.w_fb
        n_reads_rank = cfg4.n_fb + cfg4.n_gex
        pipe = CountPipeline(be, libs=(0, 1))
    else:
        # cfg5: every rank holds a whole well of its own (different cells and molecules: the seed moves with the rank)
        n_total = n if workload == "cfg5" else n * world
        seed = S.SEED0 + {"cfg2": 2, "cfg3": 3, "cfg5": 5}[workload] + (100 * rank if workload == "cfg5" else 0)
        w = S.Workload(n_total=n_total, seed=seed, n_wl=args.whitelist)
        n_reads_rank = n
        ctx.set_whitelist(0, w.wl_packed, length=w.cb_len)
        shard = dict(n=n, umi_len=w.umi_len)
        shard["cb"] = ctx.empty(n, np.uint32)
        shard["cb_qualn"] = ctx.empty((n, w.cb_len), np.uint8)
        shard["flags"] = ctx.empty(n, np.uint8)
        shard["idx"] = ctx.empty(n, np.uint32)
        if workload != "cfg2":
            shard["umi"] = ctx.empty(n, np.uint32)
            shard["umi_qualn"] = ctx.empty((n, w.umi_len), np.uint8)
            shard["feature"] = ctx.empty(n, np.uint32)
            if not args.dupinfo:
                shard["keys"] = ctx.empty(n, np.uint64)
            key_layout_args = (w.n_genes, w.umi_len, 1, 0)
            ctx.set_key_layout(*key_layout_args)
        # this rank's slice of the job's read stream (cfg5: its own well), generated straight into HBM
        chunk = 1 << 27
        first0 = 0 if workload == "cfg5" else rank * n
        for off in range(0, n, chunk):
            m = min(chunk, n - off)

            def sl(name, per=1, itemsize=1):
                a = shard.get(name)
                return None if a is None else a.ptr + off * per * itemsize

            ctx.synth(w, first0 + off, m, cb=sl("cb", 1, 4), cb_qualn=sl("cb_qualn", w.cb_len), umi=sl("umi", 1, 4),
                      umi_qualn=sl("umi_qualn", w.umi_len), feature=sl("feature", 1, 4), flags=sl("flags"))
        ctx.synchronize()
        pipe = CountPipeline(be, libs=(0,))
        if args.dupinfo and workload == "cfg3":
            dup_out = (ctx.empty(n, np.uint32), ctx.empty(n, np.uint32), ctx.empty(n, np.uint8))
    n_total_reads = n_reads_rank * world

    def step():
        if workload == "cfg4":
            return cfg4.step(be, pipe.collective)
        be.reset()
        if workload == "cfg2":
            pipe.correct_barcodes(shard)
            return None
        if workload == "cfg5":
            return pipe.run_wells(shard)
        # dup_out: the drop-in's path -- per-read DupInfo for the UB / duplicate-flag / xf tags (tx_annotation/src/read.rs:536-590)
        return pipe.run(shard, dupinfo=dup_out)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        ctx.barrier()

    def timed(steps):
        sync()
        ctx.timing_reset()
        ctx.timing(True)
        sync()
        t0 = time.perf_counter()
        result, marks = None, []
        for _ in range(steps):
            result = None      # the previous step's matrix goes back to the device pool first: steady state allocates nothing
            result = step()
            marks.append(time.perf_counter())   # host-side marks only (steps end in a blocking read of the totals)
        sync()
        dt = time.perf_counter() - t0
        ctx.timing(False)
        return ctx.allreduce_max(dt), ctx.timing_get(), result, [round((m - t0) * 1e3, 2) for m in marks]

    for _ in range(args.warmup):
        step()
    dt, ledger, result, step_marks = timed(args.steps)

    out_info = {}
    if workload in ("cfg3", "cfg4") and rank == 0 and result is not None:
        out_info = {"matrix_columns": result.n_barcodes, "matrix_nnz": result.nnz}
    if workload == "cfg5" and rank == 0 and result is not None:
        out_info = {"matrix_columns": int(result["barcode_rank"].numel()), "matrix_nnz": int(result["data"].numel()),
                    "wells": world}
    if workload in ("cfg3", "cfg4", "cfg5") and rank == 0 and out_info:
        out_info["distinct_keys"] = ctx.stat(9)                  # CRGPU_STAT_DISTINCT_KEYS
        out_info["low_support_candidates"] = ctx.stat(10)        # CRGPU_STAT_LOW_SUPPORT_CANDIDATES
    if workload == "cfg2" and rank == 0:
        idx = shard["idx"].to_host(count=min(n, 1 << 22))
        out_info = {"valid_frac_sample": float((idx != 0xFFFFFFFF).mean())}
    result = None

    # the same loop without the promise about the buffers (the default of the library): K2 scans for its misses, the sort
    # counts its digits itself.  What a default-configured host gets; `value` is the line above.
    default_ms = None
    if not args.no_default_options:
        ctx.trust_unchanged_buffers(False)
        if dense_keys and key_layout_args:
            ctx.set_option(1, 0)
            ctx.set_key_layout(*key_layout_args)
        step()
        dt_def, _, result, _ = timed(max(1, min(2, args.steps)))
        result = None
        default_ms = dt_def / max(1, min(2, args.steps)) * 1e3
        ctx.trust_unchanged_buffers(True)
        if dense_keys and key_layout_args:
            ctx.set_option(1, 1)
            ctx.set_key_layout(*key_layout_args)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_total_reads * args.steps / dt / 1e6
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
            # PMC profile of this round is attached only when it was taken on this very path, workload and size (the keys-only
            # count of cfg3 at 1 B reads with the 737 K list: not --dupinfo, not another workload)
            traffic, traffic_src = None, None
            prof = os.path.join(ROOT, "profiles", PMC_PROFILE.get(workload, "none"))
            at_profiled_size = (workload == "cfg3" and dup_out is None and n == 1_000_000_000) or (workload == "cfg4" and n == 500_000_000)
            if at_profiled_size and world == 1 and args.whitelist == 737280 and os.path.exists(prof):
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
        if workload == "cfg4":
            step_bytes = STEP_BYTES["cfg4"] * cfg4.n_fb + STEP_BYTES["cfg3"] * cfg4.n_gex
        else:
            step_bytes = STEP_BYTES[workload] * n
        step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
        desc = {"cfg2": "%d reads/GPU, 16 bp CB, barcode correction only" % n,
                "cfg3": "%d reads/GPU, 16 bp CB + 12 bp UMI, 10k cells, 36601 features%s" % (n, ", per-read DupInfo" if dup_out is not None else ""),
                "cfg5": "one whole GEM well of %d reads per GPU (gem group = rank + 1), 16 bp CB + 12 bp UMI, 36601 features, merged matrix on rank 0" % n,
                "cfg4": None}[workload]
        if workload == "cfg4":
            desc = ("%d Antibody Capture reads/GPU (Trans whitelist, %d-feature 15-mer reference, pattern %s on %d-byte R2 rows, "
                    "two extraction passes) + %d Gene Expression reads/GPU of the same well" % (cfg4.n_fb, cfg4.n_fbf, Cfg4.PATTERN, cfg4.stride, cfg4.n_gex))
        par = {"cfg5": "one well per GPU x%d, C3 gather only" % world}.get(
            workload, "read-sharded x%d%s" % (world, ", libcrgpu RCCL collectives (C1 all-reduce, C2 key exchange, C3 gather)" if world > 1 else ""))
        line = {
            "metric": "M reads/sec barcode-correct" if workload == "cfg2" else "M reads/sec barcode-correct+UMI-count",
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
            "config": {"workload": "%s: %s, %d-entry whitelist" % (workload, desc, args.whitelist),
                       "reads_per_gpu": n_reads_rank, "parallelism": par,
                       # a promise the host makes (off by default): K2 reuses K1's miss records, the sort the key histograms
                       "options": {"CRGPU_OPT_BUFFERS_UNCHANGED_BETWEEN_CALLS": 1, "CRGPU_OPT_DENSE_BARCODE_KEYS": int(dense_keys)}},
            "default_options_ms_per_step": default_ms,
            "roofline": roof,
            "family_roofline_frac": fam_fracs,
            "step_roofline": {"basis_bytes_per_read": step_bytes / n_reads_rank, "achieved": step_gbs, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in ledger.items() if v[1]},
            "kernel_launches_per_step": {k: v[1] / args.steps for k, v in ledger.items() if v[1]},
            "kernel_units_per_step": {k: v[2] / args.steps for k, v in ledger.items() if v[1] and v[2]},
            "output": out_info,
            "host_step_marks_ms": step_marks,
        }
        if world > 1 and ledger.get("comm", (0, 0, 0))[1]:
            cm = ledger["comm"]
            total_steps = args.warmup + args.steps + (0 if args.no_default_options else 1 + max(1, min(2, args.steps)))
            line["comm"] = {"ms_per_step": cm[0] / args.steps, "collectives_per_step": cm[1] / args.steps,
                            "bytes_per_step_rank0": {"C1_table_allreduce": ctx.stat(5) / total_steps, "C2_key_exchange": ctx.stat(6) / total_steps,
                                                     "C3_triplet_gather": ctx.stat(7) / total_steps},
                            "note": "rank 0's C1 / C2 / C3 spans incl. waiting for the other ranks (HIP events around each collective); "
                                    "bytes = what rank 0 put into each collective per step (C2: its keys, own share included)"}
        if world == 1 and workload == "cfg3" and not args.no_verify and dup_out is None:
            # not timed: the laws of cellranger_amd/selfcheck.py on this very workload at its full size
            from cellranger_amd import selfcheck
            try:
                line["verify"] = dict(selfcheck.full_size_properties(ctx, shard, local_rank), ok=True)
            except AssertionError as e:  # report, never hide: the line still carries the timing
                import traceback
                line["verify"] = {"ok": False, "failed": traceback.format_exc(limit=2).strip().splitlines()[-3:], "msg": str(e)}
        if world == 1 and workload == "cfg3" and not args.no_end_to_end:
            be.reset()
            try:
                k1_ms = ledger.get("match", (0.0, 0, 0))[0] / args.steps
                line["end_to_end"] = end_to_end(ctx, w, n, ms_per_step * 1e-3, k1_ms * 1e-3, 32_000_000)
            except Exception as e:  # noqa: BLE001 - an extra figure must never cost the line
                line["end_to_end"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and workload != "cfg4":
            line["cpu_baseline"] = cpu_baseline(w, "cfg2" if workload == "cfg2" else "cfg3", min(args.cpu_sample, n),
                                                min(args.cpu_1thread_sample, n))
        elif world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_cfg4(cfg4, min(args.cpu_sample // 4, cfg4.n_fb))
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
