"""N>1 path on CPUs: CountPipeline's call sequence (C1 all-reduce of the prior, C2 all-to-all of molecule
keys by barcode range, C3 gather of triplets) under gloo with world_size 2, 3 and 8 (the size of the scaling run), driven through the
oracle-backed stand-in backend (whose collectives restate libcrgpu's with torch.distributed); the assembled matrix must equal
the single-process oracle's.  The product's own collectives (comm.hip) need a GPU: tests/test_gpu_pipeline.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cellranger_amd import synth as S
        from cellranger_amd.pipeline import CountPipeline
        from oracle_backend import OracleBackend

        w = S.Workload(n_total=n, seed=seed, n_wl=3000, n_cells=40, n_ambient=300, n_genes=25, umi_len=6,
                       umi_err=0.03, cb_err=0.01, reads_per_umi=2, n_libs=2)
        per = n // world
        r = w.host_reads(rank * per, per)
        shard = dict(n=per, umi_len=w.umi_len, cb=r["cb"], cb_qualn=r["cb_qualn"], flags=r["flags"],
                     idx=np.zeros(per, np.uint32), umi=r["umi"], umi_qualn=r["umi_qualn"], feature=r["feature"])
        be = OracleBackend(w.wl_packed, w.cb_len, w.n_genes, w.umi_len, n_libs=2, mux_mask=0b10, dist=dist)
        pipe = CountPipeline(be, libs=(0, 1))
        for _ in range(2):  # a second step must reset cleanly
            be.reset()
            m = pipe.run(shard)
        if rank == 0:
            np.savez(out_path, rank=m.barcode_rank, indptr=m.indptr, indices=m.indices, data=m.data)
        else:
            assert m is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_pipeline_collectives_match_single_process_oracle(world, tmp_path):
    sys.path.insert(0, HERE)
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S

    n, seed = 6000 * world, 31
    out = str(tmp_path / "m.npz")
    mp.spawn(_worker, args=(world, _free_port(), n, seed, out), nprocs=world, join=True)
    got = np.load(out)

    w = S.Workload(n_total=n, seed=seed, n_wl=3000, n_cells=40, n_ambient=300, n_genes=25, umi_len=6,
                   umi_err=0.03, cb_err=0.01, reads_per_umi=2, n_libs=2)
    r = w.host_reads(0, n)
    cb, cbq = S.to_ascii(r["cb"], r["cb_qualn"], 16)
    umi, uq = S.to_ascii(r["umi"], r["umi_qualn"], 6)
    wl = O.Whitelist(E.unpack_seqs(w.wl_packed, 16))
    res = O.run_pipeline(dict(cb=cb, cb_qual=cbq, umi=umi, umi_qual=uq, feature=r["feature"], lib=r["flags"] & 0x0F),
                         [wl, wl], n_lib=2, multiplexing_lib_mask=0b10)
    canon_sorted = np.sort(w.wl_packed)
    assert np.array_equal(E.unpack_seqs(canon_sorted[got["rank"]], 16), res.barcodes)
    assert np.array_equal(got["indptr"], res.indptr)
    assert np.array_equal(got["indices"], res.indices)
    assert np.array_equal(got["data"], res.data)
    assert len(res.data) > 100


def _wells_worker(rank, world, port, n, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cellranger_amd import synth as S
        from cellranger_amd.pipeline import CountPipeline
        from oracle_backend import OracleBackend

        w = S.Workload(n_total=n, seed=70, n_wl=3000, n_cells=40, n_ambient=300, n_genes=25, umi_len=6)
        r = _well_reads(rank, n)
        shard = dict(n=n, umi_len=6, cb=r["cb"], cb_qualn=r["cb_qualn"], flags=r["flags"], idx=np.zeros(n, np.uint32),
                     umi=r["umi"], umi_qualn=r["umi_qualn"], feature=r["feature"])
        be = OracleBackend(w.wl_packed, 16, 25, 6, dist=dist)
        m = CountPipeline(be).run_wells(shard)
        if rank == 0:
            np.savez(out_path, **{k: v.numpy() for k, v in m.items()})
        else:
            assert m is None
    finally:
        dist.destroy_process_group()


def _well_reads(well, n):
    """every well draws its own cells from the SAME whitelist (seed 70 fixes the list, the well seeds the reads)"""
    from cellranger_amd import synth as S

    base = S.Workload(n_total=n, seed=70, n_wl=3000, n_cells=40, n_ambient=300, n_genes=25, umi_len=6,
                      umi_err=0.03, cb_err=0.01, reads_per_umi=2)
    return base.host_reads(well * n, n)  # disjoint stretches of one stream: different molecules per well


@pytest.mark.parametrize("world", [2])
def test_wells_merge_is_column_concatenation_in_gem_group_order(world, tmp_path):
    """BASELINE configs[4] down-scaled: one GEM well per rank, merged matrix gathered on rank 0 == the oracle run on
    each well separately, concatenated in (gem_group, barcode) order."""
    sys.path.insert(0, HERE)
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S

    n = 8000
    out = str(tmp_path / "wells.npz")
    mp.spawn(_wells_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    got = np.load(out)
    base = S.Workload(n_total=n, seed=70, n_wl=3000, n_cells=40, n_ambient=300, n_genes=25, umi_len=6)
    wl = O.Whitelist(E.unpack_seqs(base.wl_packed, 16))
    canon_sorted = np.sort(base.wl_packed)
    col0 = nz0 = 0
    for well in range(world):
        r = _well_reads(well, n)
        cb, cbq = S.to_ascii(r["cb"], r["cb_qualn"], 16)
        umi, uq = S.to_ascii(r["umi"], r["umi_qualn"], 6)
        res = O.run_pipeline(dict(cb=cb, cb_qual=cbq, umi=umi, umi_qual=uq, feature=r["feature"]), [wl])
        V, nnz = len(res.barcodes), len(res.data)
        sl = slice(col0, col0 + V)
        assert (got["gem_group"][sl] == well + 1).all()
        assert np.array_equal(E.unpack_seqs(canon_sorted[got["barcode_rank"][sl].view(np.uint32)], 16), res.barcodes)
        assert np.array_equal(got["indptr"][col0:col0 + V + 1] - nz0, res.indptr)
        assert np.array_equal(got["indices"][nz0:nz0 + nnz], res.indices) and np.array_equal(got["data"][nz0:nz0 + nnz], res.data)
        col0 += V
        nz0 += nnz
    assert col0 == len(got["barcode_rank"]) and nz0 == len(got["data"]) and nz0 > 100
