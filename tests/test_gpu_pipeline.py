"""GPU tests of the call sequence with the product backend (HipBackend -> libcrgpu): the plain single-GPU path, the
collective path (C1/C2/C3 = comm.hip) on a 1-rank RCCL communicator, and on 3 and 4 thread-ranks joined by the in-process
group (the same orchestration code the 8-GPU RCCL run uses; only the transport differs)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make_shard(c, w, r, n):
    shard = dict(n=n, umi_len=w.umi_len)
    for k in ("cb", "cb_qualn", "flags", "umi", "umi_qualn", "feature"):
        shard[k] = c.upload(r[k])
    shard["idx"] = c.empty(n, np.uint32)
    return shard


def _check_against_oracle(c, w, r, m):
    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E

    res = O.run_pipeline(G.oracle_reads_from_packed(r, w.cb_len, w.umi_len), [O.Whitelist(E.unpack_seqs(w.wl_packed, 16))],
                         n_threads=4)
    rank, indptr, indices, data = m.download()
    _, canon_sorted = c.canon_order()
    assert np.array_equal(E.unpack_seqs(canon_sorted[rank], 16), res.barcodes)
    assert np.array_equal(indptr, res.indptr)
    assert np.array_equal(indices, res.indices)
    assert np.array_equal(data, res.data)
    assert m.nnz > 1000


def test_pipeline_single_gpu_device_csc():
    import gpu_helpers as G
    from cellranger_amd import synth as S
    from cellranger_amd.pipeline import CountPipeline, HipBackend

    n = 300_000
    w = S.Workload(n_total=n, seed=41, n_wl=60_000, n_cells=150, n_ambient=5000, n_genes=500)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    r = w.host_reads(0, n)
    shard = _make_shard(c, w, r, n)
    be = HipBackend(c, 0)
    pipe = CountPipeline(be)
    for _ in range(3):  # repeated steps recycle pooled buffers and must give the same answer
        be.reset()
        m = pipe.run(shard)
    _check_against_oracle(c, w, r, m)
    c.close()


def test_pipeline_collective_path_on_one_rank_rccl_communicator():
    """C1/C2/C3 through libcrgpu's own RCCL communicator (crgpu_create with a unique id, one rank): ncclAllReduce,
    the count exchange, grouped ncclSend/ncclRecv to self and the gather run for real."""
    import gpu_helpers as G
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd.pipeline import CountPipeline, HipBackend

    n = 200_000
    w = S.Workload(n_total=n, seed=42, n_wl=60_000, n_cells=150, n_ambient=5000, n_genes=500)
    c = E.Context(0, n_ranks=1, rank=0, unique_id=E.get_unique_id())
    c.set_whitelist(0, w.wl_packed, length=16)
    c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    r = w.host_reads(0, n)
    shard = _make_shard(c, w, r, n)
    be = HipBackend(c, 0)
    pipe = CountPipeline(be, force_collectives=True)
    for _ in range(2):
        be.reset()
        m = pipe.run(shard)
    _check_against_oracle(c, w, r, m)
    # one-well-per-rank mode (BASELINE configs[4]): the CSC gather on the 1-rank communicator returns the same matrix
    be.reset()
    merged = pipe.run_wells(shard)
    rank, indptr, indices, data = m.download()
    assert np.array_equal(merged["barcode_rank"].cpu().numpy().view(np.uint32), rank)
    assert np.array_equal(merged["indptr"].cpu().numpy(), indptr)
    assert np.array_equal(merged["indices"].cpu().numpy(), indices) and np.array_equal(merged["data"].cpu().numpy(), data)
    assert (merged["gem_group"].cpu().numpy() == 1).all()
    assert c.allreduce_max(3.5) == 3.5
    c.barrier()
    c.close()


@pytest.mark.parametrize("world", [3, 4])
def test_thread_ranks_share_one_gpu_through_the_local_group(world):
    """The multi-rank path end to end inside libcrgpu: `world` ranks as host threads with their own contexts, streams and
    pools, joined by crgpu_local_group_id (the transport a one-process, thread-per-GPU host uses; here all ranks sit on
    the one GPU of the test box).  C1 all-reduce, C2 exchange over histogram-balanced barcode ranges (partition, count
    exchange, offsets, ordering), C3 gather are the code the RCCL transport runs under, too; the matrix on rank 0 equals
    the single-process oracle on all reads."""
    import threading

    import gpu_helpers as G
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd.pipeline import CountPipeline, HipBackend

    per = 120_000
    n = world * per
    w = S.Workload(n_total=n, seed=43, n_wl=60_000, n_cells=150, n_ambient=5000, n_genes=500)
    r_all = w.host_reads(0, n)
    gid = E.local_group_id(world)
    results, errors = [None] * world, []

    def worker(rank):
        c = None
        try:
            c = E.Context(0, n_ranks=world, rank=rank, unique_id=gid)
            c.set_whitelist(0, w.wl_packed, length=16)
            c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
            r = {k: v[rank * per:(rank + 1) * per] for k, v in r_all.items()}
            shard = _make_shard(c, w, r, per)
            be = HipBackend(c, 0)
            pipe = CountPipeline(be)
            for _ in range(2):
                be.reset()
                m = pipe.run(shard)
            assert c.allreduce_max(float(rank)) == float(world - 1)
            results[rank] = (c, m)
        except Exception as e:  # noqa: BLE001 - surfaced below; closing the context breaks the group, so the others fail instead of waiting
            errors.append(e)
            if c is not None:
                c.close()

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(results[k][1] is None for k in range(1, world))
    c0, m0 = results[0]
    _check_against_oracle(c0, w, r_all, m0)
    for c, _ in results:
        c.close()


def test_a_rank_that_leaves_breaks_the_local_group_instead_of_hanging_it():
    import threading

    from cellranger_amd import engine as E
    from cellranger_amd._lib import CrgpuError

    gid = E.local_group_id(2)
    out = {}

    def stay():
        c = E.Context(0, n_ranks=2, rank=0, unique_id=gid)
        try:
            c.barrier()          # both arrive
            c.barrier()          # the other rank has left
            out["second"] = "returned"
        except CrgpuError as e:
            out["second"] = e.code
        c.close()

    def leave():
        c = E.Context(0, n_ranks=2, rank=1, unique_id=gid)
        c.barrier()
        c.close()

    ts = [threading.Thread(target=stay), threading.Thread(target=leave)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert out["second"] == -7  # CRGPU_ECOMM


def test_four_host_threads_share_one_context():
    """ALIGN_AND_COUNT runs four worker threads per chunk (cr_lib/src/stages/align_and_count.rs:698-732).  One context
    may be shared by them: every entry point locks it, so the calls execute one at a time in arrival order.  Four
    threads hammer one context with whole count-stage calls on their own batches; every result equals the one the same
    batch gives alone."""
    import threading

    import gpu_helpers as G
    from cellranger_amd import synth as S

    n, T, rounds = 60_000, 4, 3
    w = S.Workload(n_total=n * T, seed=91, n_wl=20_000, n_cells=100, n_ambient=2000, n_genes=300)
    c = G.fresh_ctx()
    c.set_whitelist(0, w.wl_packed, length=16)
    c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    # the barcode stage once for all batches (its histograms are context state, not per thread)
    batches = []
    for t in range(T):
        r = w.host_reads(t * n, n)
        d = {k: c.upload(r[k]) for k in ("cb", "cb_qualn", "flags", "umi", "umi_qualn", "feature")}
        d["idx"] = c.empty(n, np.uint32)
        c.match_and_count(d["cb"], d["flags"], n, d["idx"])
        batches.append(d)
    for d in batches:
        c.correct(d["cb"], d["cb_qualn"], d["flags"], n, d["idx"])

    def count(d):
        recs = c.records(n, w.umi_len, d["idx"], d["umi"], d["umi_qualn"], d["feature"], d["flags"])
        keys = c.empty(n, np.uint64)
        nk = c.build_keys(recs, keys)
        cnt = c.count_keys(keys, nk)
        out = cnt.triplets() + (cnt.molecules()["read_count"],)
        cnt.free()
        return out

    expect = [count(d) for d in batches]
    errors = []

    def worker(t):
        try:
            for _ in range(rounds):
                got = count(batches[t])
                for a, b in zip(got, expect[t]):
                    assert np.array_equal(a, b)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=300)
    assert not errors, errors
    assert len(expect[0][0]) > 100
    c.close()


def test_sharded_well_returns_every_reads_dupinfo_to_its_rank():
    """crgpu_count_records_sharded_dev on three thread-ranks: the per-read DupInfo each rank gets back for ITS OWN reads
    (processed UMI, read count, corrected / low-support / umi-count flags: what the unchanged host needs for the UB /
    duplicate-flag / xf tags, tx_annotation/src/read.rs:536-590) equals the single-process oracle's on the whole well,
    read by read, and the gathered triplets give the oracle's matrix."""
    import threading

    import gpu_helpers as G
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd._lib import COUNTS_CORRECTED, COUNTS_VALID

    world, per = 3, 100_000
    n = world * per
    w = S.Workload(n_total=n, seed=47, n_wl=40_000, n_cells=120, n_ambient=3000, n_genes=300, umi_len=8, umi_err=0.02)
    r_all = w.host_reads(0, n)
    gid = E.local_group_id(world)
    outs, errors = [None] * world, []

    def worker(rank):
        c = None
        try:
            c = E.Context(0, n_ranks=world, rank=rank, unique_id=gid)
            c.trust_unchanged_buffers(True)
            c.set_whitelist(0, w.wl_packed, length=16)
            c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
            r = {k: v[rank * per:(rank + 1) * per] for k, v in r_all.items()}
            sh = _make_shard(c, w, r, per)
            c.match_and_count(sh["cb"], sh["flags"], per, sh["idx"])
            c.allreduce_counts(-1, COUNTS_VALID)
            c.correct(sh["cb"], sh["cb_qualn"], sh["flags"], per, sh["idx"])
            c.allreduce_counts(-1, COUNTS_CORRECTED)
            recs = c.records(per, w.umi_len, sh["idx"], sh["umi"], sh["umi_qualn"], sh["feature"], sh["flags"])
            d_pu, d_rc, d_fl = c.empty(per, np.uint32), c.empty(per, np.uint32), c.empty(per, np.uint8)
            counts = c.count_records_sharded(recs, d_pu, d_rc, d_fl)
            arrs, total = c.gather_triplets(counts)
            m = None
            if rank == 0:
                m = c.assemble_matrix_dev(arrs[0], arrs[1], arrs[2], total).download()
            outs[rank] = (c, d_pu.to_host(), d_rc.to_host(), d_fl.to_host(), m, c.canon_order()[1])
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            if c is not None:
                c.close()

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not errors, errors
    res = O.run_pipeline(G.oracle_reads_from_packed(r_all, w.cb_len, w.umi_len), [O.Whitelist(E.unpack_seqs(w.wl_packed, 16))],
                         n_threads=4, want_dupinfo=True)
    od = res.dupinfo
    pu = np.concatenate([o[1] for o in outs])
    rc = np.concatenate([o[2] for o in outs])
    fl = np.concatenate([o[3] for o in outs])
    has = od["has_dupinfo"] != 0
    assert np.array_equal((fl & 1) != 0, has)
    assert np.array_equal((fl & 2) != 0, od["is_corrected"] != 0)
    assert np.array_equal((fl & 4) != 0, od["is_low_support"] != 0)
    assert np.array_equal((fl & 8) != 0, od["is_umi_count"] != 0)
    assert np.array_equal(pu[has], od["processed_umi"][has]) and np.array_equal(rc[has], od["read_count"][has])
    assert not pu[~has].any() and not rc[~has].any()
    assert int(((fl & 2) != 0).sum()) > 100 and int(((fl & 8) != 0).sum()) > 10_000
    rank_, indptr, indices, data = outs[0][4]
    assert np.array_equal(E.unpack_seqs(outs[0][5][rank_], 16), res.barcodes)
    assert np.array_equal(indptr, res.indptr) and np.array_equal(indices, res.indices) and np.array_equal(data, res.data)
    for o in outs:
        o[0].close()


@pytest.mark.parametrize("dupinfo", [False, True])
def test_a_rank_that_fails_in_front_of_the_key_exchange_fails_everywhere(dupinfo, monkeypatch):
    """collectives must be failure-symmetric.  Rank 1's preparation of the key exchange is made to fail
    (CRGPU_TEST_FAIL_EXCHANGE_RANK): it still takes part in the count exchange, which carries its status, so rank 1 returns
    its own error (CRGPU_ENOMEM) and ranks 0 and 2 return CRGPU_ECOMM naming it -- nobody enters the data exchange and waits
    there for ever (RCCL has no timeout).  Afterwards the same contexts run the whole pipeline successfully: the failure left
    the group usable.  Both exchanges: crgpu_exchange_keys_dev and crgpu_count_records_sharded_dev."""
    import threading

    import gpu_helpers as G
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S
    from cellranger_amd._lib import CrgpuError
    from cellranger_amd.pipeline import CountPipeline, HipBackend

    world, per = 3, 60_000
    n = world * per
    w = S.Workload(n_total=n, seed=47, n_wl=30_000, n_cells=100, n_ambient=3000, n_genes=300)
    r_all = w.host_reads(0, n)
    gid = E.local_group_id(world)
    codes, results, errors = [None] * world, [None] * world, []
    gate = threading.Barrier(world)

    def worker(rank):
        c = None
        try:
            c = E.Context(0, n_ranks=world, rank=rank, unique_id=gid)
            c.set_whitelist(0, w.wl_packed, length=16)
            c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
            r = {k: v[rank * per:(rank + 1) * per] for k, v in r_all.items()}
            shard = _make_shard(c, w, r, per)
            be = HipBackend(c, 0)
            pipe = CountPipeline(be)
            dup = (c.empty(per, np.uint32), c.empty(per, np.uint32), c.empty(per, np.uint8)) if dupinfo else None
            gate.wait()
            if rank == 0:
                os.environ["CRGPU_TEST_FAIL_EXCHANGE_RANK"] = "1"
            gate.wait()
            try:
                be.reset()
                pipe.run(shard, dupinfo=dup)
                codes[rank] = 0
            except CrgpuError as e:
                codes[rank] = (e.code, str(e))
            gate.wait()
            if rank == 0:
                del os.environ["CRGPU_TEST_FAIL_EXCHANGE_RANK"]
            gate.wait()
            be.reset()
            m = pipe.run(shard, dupinfo=dup)     # the group still works
            results[rank] = (c, m)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            gate.abort()
            if c is not None:
                c.close()

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a rank is still waiting in a collective"
    assert not errors, errors
    assert codes[1][0] == -4 and "forced failure" in codes[1][1], codes          # its own error
    assert codes[0][0] == -7 and codes[2][0] == -7 and "rank 1 failed" in codes[0][1], codes   # CRGPU_ECOMM on the others
    c0, m0 = results[0]
    _check_against_oracle(c0, w, r_all, m0)
    for c, _ in results:
        c.close()
