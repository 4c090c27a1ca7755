"""Size-independent properties of one pass of the hot path, checked on the device at any size (bench.py --verify runs
them on the full 1 B-record workload, tests/test_gpu_scale.py at 1 B and 50 M).

Not a CPU implementation of anything: the path runs through libcrgpu as always, and torch only reduces / compares the
arrays it left in HBM (conservation laws, sortedness, the BarcodeIndex rule, CSC invariants).  Every check cites the
reference rule it follows from.
"""
import numpy as np

from ._lib import COUNTS_CORRECTED, COUNTS_VALID
from .pipeline import _DevView


def _i32(ptr, n, dev):
    import torch

    if n == 0:
        return torch.empty(0, dtype=torch.int32, device=dev)
    return torch.as_tensor(_DevView(ptr, (n,), "<i4"), device=dev)


_SLICE = 1 << 30   # torch's masked indexing / nonzero are used on slices of at most 2^30 elements


def _bincount_where(values, mask, minlength):
    import torch

    acc = torch.zeros(minlength, dtype=torch.int64, device=values.device)
    for s in range(0, values.numel(), _SLICE):
        v, m = values[s:s + _SLICE], mask[s:s + _SLICE]
        acc += torch.bincount(v[m].long(), minlength=minlength)
    return acc


def barcode_stage_properties(ctx, shard, device_index=0, _keep=None, libs=(0,)):
    """K1 -> K2 on `shard` (device arrays cb, cb_qualn, flags, idx; n; the reads' libraries `libs`, all with a whitelist
    over the same canonical list): per-library histograms == per-read results,
    pass B touches only invalid reads, every correction is a Hamming-1 neighbour.  Returns {check name: value}."""
    import torch

    dev = "cuda:%d" % device_index
    n = shard["n"]
    W = ctx.n_canon
    out = {}

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize(device_index)

    ctx.reset_counts()
    ctx.match_and_count(shard["cb"], shard["flags"], n, shard["idx"])
    sync()
    idx = _i32(shard["idx"].ptr, n, dev)
    idx_a = idx.clone()
    hit_a = idx_a >= 0                                            # CRGPU_MISS is -1 as i32
    fl = torch.as_tensor(_DevView(shard["flags"].ptr, (n,), "|u1"), device=dev)
    lib_of = fl & 0x0F
    in_lib = [lib_of == l for l in libs] if len(libs) > 1 else [torch.ones_like(hit_a)]
    valid = [_i32(ctx.counts_dev(l, COUNTS_VALID), W, dev).clone() for l in libs]
    # MakeShardHistograms::observe (make_shard_metrics.rs:171-188): one count per read whose barcode is on the whitelist,
    # in the histogram of the read's library type
    assert sum(int(v.sum()) for v in valid) == int(hit_a.sum())
    for v, m in zip(valid, in_lib):
        assert torch.equal(_bincount_where(idx_a, hit_a & m, W), v.long())
    out["valid_reads"] = int(hit_a.sum())
    sync()
    ctx.correct(shard["cb"], shard["cb_qualn"], shard["flags"], n, shard["idx"])
    sync()
    corrected = [_i32(ctx.counts_dev(l, COUNTS_CORRECTED), W, dev).clone() for l in libs]
    hit_b = idx >= 0
    fixed = hit_b & ~hit_a
    # barcode_correction.rs:328-345: only invalid barcodes are looked at; each corrected read counts once
    assert bool(((idx == idx_a) | ~hit_a).all())
    assert sum(int(v.sum()) for v in corrected) == int(fixed.sum())
    for v, m in zip(corrected, in_lib):
        assert torch.equal(_bincount_where(idx, fixed & m, W), v.long())
    out["corrected_reads"] = int(fixed.sum())
    # corrector.rs:111-171: a corrected barcode is a whitelist entry at Hamming distance exactly 1 (<= 1 with an N)
    _, canon_sorted = ctx.canon_order()
    canon = torch.as_tensor(canon_sorted.astype(np.int64), device=dev)
    cb = _i32(shard["cb"].ptr, n, dev)
    sel = torch.nonzero(fixed[:_SLICE]).squeeze(1)[:25_000_000]
    if n > _SLICE:   # both ends of the batch
        sel = torch.cat([sel, torch.nonzero(fixed[n - _SLICE:]).squeeze(1)[-25_000_000:] + (n - _SLICE)])
    x = (canon[idx[sel].long()] ^ (cb[sel].long() & 0xFFFFFFFF))
    y = (x | (x >> 1)) & 0x55555555
    n_diff = torch.zeros_like(y)
    for k in range(16):
        n_diff += (y >> (2 * k)) & 1
    has_n = (fl[sel] & 0x10) != 0
    assert bool(((n_diff == 1) | has_n).all()) and bool((n_diff <= 1).all())
    if _keep is not None:
        _keep.update(valid=valid, corrected=corrected, n_valid_after=int(hit_b.sum()))
    return out


def full_size_properties(ctx, shard, device_index=0, libs=(0,)):
    """Runs K1 -> K2 -> keys -> dedup -> matrix on `shard` (device arrays cb, cb_qualn, flags, idx, umi, umi_qualn,
    feature; n; umi_len; libraries `libs` = 0..len-1, key layout set accordingly) and returns {check name: value}; raises AssertionError on the first violated law."""
    import torch

    dev = "cuda:%d" % device_index
    n = shard["n"]
    keep = {}
    out = barcode_stage_properties(ctx, shard, device_index, keep, libs)
    valid, corrected, n_valid_after = keep["valid"], keep["corrected"], keep["n_valid_after"]

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize(device_index)

    sync()

    recs = ctx.records(n, shard["umi_len"], shard["idx"], shard["umi"], shard["umi_qualn"], shard["feature"], shard["flags"])
    keys = shard["keys"] if shard.get("keys") is not None else ctx.empty(n, np.uint64)
    ctx.enable_barcode_summary(True)
    try:
        nk = ctx.build_keys(recs, keys)
        counts = ctx.count_keys(keys, nk)
    finally:
        ctx.enable_barcode_summary(False)
    sync()
    out["keys"], out["molecules"], out["triplets"] = int(nk), counts.n_molecules, counts.n_triplets
    assert 0 < nk <= n_valid_after
    d_bc, d_ft, d_ct = counts.triplets_dev()
    nt = counts.n_triplets
    bc, ft, ct = _i32(d_bc, nt, dev), _i32(d_ft, nt, dev), _i32(d_ct, nt, dev)
    # BcUmiInfo::feature_counts (types.rs:180-188): one entry per (barcode, feature), BarcodeThenFeatureOrder, counts > 0
    tkey = (bc.long() << 32) | ft.long()
    assert bool((tkey[1:] > tkey[:-1]).all()) and bool((ct > 0).all())
    assert int(ct.long().sum()) == counts.n_molecules
    # every triplet's barcode has reads in the histograms (BarcodeIndex, barcode_index.rs:20-53)
    seen = torch.zeros_like(valid[0], dtype=torch.bool)
    for v in valid + corrected:
        seen |= v != 0
    assert bool(seen[bc.long()].all())
    sync()
    md = ctx.assemble_matrix_dev(d_bc, d_ft, d_ct, nt)
    rank, indptr, indices, data = md.download()
    # CSC arrays (count_matrix.rs:382-448)
    assert indptr[0] == 0 and indptr[-1] == nt == len(data) and (np.diff(indptr) >= 0).all()
    assert np.array_equal(rank, torch.nonzero(seen).squeeze(1).cpu().numpy().astype(np.uint32))
    assert int(data.astype(np.int64).sum()) == counts.n_molecules
    assert np.array_equal(indices, ft.cpu().numpy()) and np.array_equal(data, ct.cpu().numpy())
    per_col = np.diff(indptr)
    assert np.array_equal(np.repeat(rank, per_col), bc.cpu().numpy().astype(np.uint32))
    out["matrix_columns"], out["matrix_nnz"] = len(rank), int(nt)
    # BarcodeSummary (aligner.rs:33-68): one row per barcode with reads; umis = molecules; candidates = their reads
    rows = counts.barcode_summary()
    order = np.lexsort((rows["barcode_rank"], rows["library"]))
    assert np.array_equal(order, np.arange(len(rows)))                  # ordered by (library, rank)
    for l, v, cr in zip(libs, valid, corrected):
        mine = rows[rows["library"] == l]
        reads = (v.long() + cr.long()).cpu().numpy()
        assert np.array_equal(mine["barcode_rank"], np.nonzero(reads)[0].astype(np.uint32))   # a row per barcode with a read
        assert np.array_equal(mine["reads"], reads[mine["barcode_rank"]].astype(np.uint64))
    assert set(np.unique(rows["library"])) <= set(libs)
    assert int(rows["umis"].sum()) == counts.n_molecules
    cs = np.concatenate([np.zeros(1, np.int64), np.cumsum(data, dtype=np.int64)])
    col_umis = np.zeros(ctx.n_canon, np.int64)
    np.add.at(col_umis, rows["barcode_rank"], rows["umis"].astype(np.int64))      # summed over the libraries
    assert np.array_equal(col_umis[rank], cs[indptr[1:]] - cs[indptr[:-1]])        # == column sums of the matrix
    cand = int(rows["candidate_dup_reads"].sum())
    # mark_dups.rs: reads are conserved by the UMI correction; only low-support molecules drop out
    assert cand <= nk and cand > 0.9 * nk
    assert int(rows["umi_corrected_reads"].sum()) < 0.1 * nk
    out["candidate_dup_reads"], out["umi_corrected_reads"] = cand, int(rows["umi_corrected_reads"].sum())
    out["checksum"] = int((tkey * 1315423911 + ct.long()).sum().item() & 0x7FFFFFFFFFFFFFFF)
    md.free()
    counts.free()
    return out
