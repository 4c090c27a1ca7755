"""Host orchestration of one GEM well across the GPUs of a node.

One process per GPU (or one host thread per GPU inside a process).  Reads are sharded; the path has three real exchange
steps, all of them entry points of libcrgpu (comm.hip: RCCL over xGMI, or the in-process group):

  C1  crgpu_allreduce_counts     all-reduce(sum) of the per-library valid-barcode histogram -> the corrector's GLOBAL
                                 prior (make_shard.rs:343-358 join -> barcode_correction.rs:295-325)
  C2  crgpu_exchange_keys_dev    all-to-all of 64-bit molecule keys by barcode range -> every barcode's reads on one GPU
                                 (the reference: barcode-sorted shards + make_chunks, align_and_count.rs:505-524)
  C3  crgpu_gather_triplets_dev  gather of the disjoint (barcode, feature, count) triplets -> rank 0 assembles the CSC

`CountPipeline` is only the order of the calls; a Rust host makes the same calls.  The calls go through a
backend object: `HipBackend` is the product (libcrgpu via the C ABI, no CPU fallback); the CPU tests drive the same
sequence with an oracle-backed stand-in whose collectives are torch.distributed/gloo (tests/oracle_backend.py).
"""
import numpy as np

from . import engine as E
from ._lib import COUNTS_CORRECTED, COUNTS_VALID


class _DevView:
    """Alias of a device allocation with a torch-friendly dtype."""

    def __init__(self, ptr, shape, typestr, owner=None):
        self.ptr, self.shape, self.typestr, self.owner = ptr, tuple(shape), typestr, owner

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.typestr, "data": (self.ptr, False), "version": 2}


class HipBackend:
    """Compute + collectives backend = libcrgpu on one MI355X (the context carries rank / n_ranks / communicator)."""

    def __init__(self, ctx, device_index=0):
        self.ctx = ctx
        self.device_index = device_index
        self.world, self.rank = ctx.n_ranks, ctx.rank
        self._keep = []

    # -- barcode stage --------------------------------------------------------------------------------
    def reset(self):
        self.ctx.reset_counts()
        self._keep = []

    def match_and_count(self, shard):
        self.ctx.match_and_count(shard["cb"], shard.get("flags"), shard["n"], shard["idx"])

    def correct(self, shard):
        self.ctx.correct(shard["cb"], shard["cb_qualn"], shard.get("flags"), shard["n"], shard["idx"], shard.get("corrected"))

    # -- collectives (crgpu.h "collectives") -------------------------------------------------------------
    def allreduce_hist(self, libs, which):
        for lib in libs:
            self.ctx.allreduce_counts(lib, which)

    def exchange_keys(self, keys, n_keys):
        recv, n_recv, _ = self.ctx.exchange_keys(keys, n_keys)
        self._keep.append(recv)
        return recv, n_recv

    def gather_triplets(self, counts):
        arrs, total = self.ctx.gather_triplets(counts, root=0)
        if arrs is not None:
            self._keep.extend(arrs)
        return arrs, total

    # -- count stage ------------------------------------------------------------------------------------
    def build_keys(self, shard):
        recs = self.ctx.records(shard["n"], shard["umi_len"], shard["idx"], shard["umi"], shard["umi_qualn"],
                                shard["feature"], shard.get("flags"))
        keys = shard.get("keys")
        if keys is None:
            keys = self.ctx.empty(max(shard["n"], 1), np.uint64)
            self._keep.append(keys)
        n_keys = self.ctx.build_keys(recs, keys)
        return keys, n_keys

    def count_keys(self, keys, n_keys):
        counts = self.ctx.count_keys(keys, n_keys)
        self._keep.append(counts)
        return counts

    def count_records(self, shard, dupinfo, sharded):
        """dedup straight from the records with per-read DupInfo (processed UMI, read count, flags) for THIS rank's reads;
        sharded: the well is spread over the ranks (crgpu_count_records_sharded_dev), else this rank holds all of it"""
        recs = self.ctx.records(shard["n"], shard["umi_len"], shard["idx"], shard["umi"], shard["umi_qualn"],
                                shard["feature"], shard.get("flags"))
        f = self.ctx.count_records_sharded if sharded else self.ctx.count_records
        counts = f(recs, *dupinfo)
        self._keep.append(counts)
        return counts

    def triplet_arrays(self, counts):
        return counts.triplets_dev()

    def assemble(self, d_bc, d_feature, d_count, n_triplets):
        m = self.ctx.assemble_matrix_dev(d_bc, d_feature, d_count, n_triplets)
        self._keep.append(m)
        return m

    def _tensor(self, ptr, n, typestr):
        import torch

        if n == 0:
            dt = {"<i4": torch.int32, "<i8": torch.int64}[typestr]
            return torch.empty(0, dtype=dt, device="cuda:%d" % self.device_index)
        return torch.as_tensor(_DevView(ptr, (n,), typestr), device="cuda:%d" % self.device_index)

    def gather_wells(self, m):
        """run_wells: the per-well CSC blocks of every rank on rank 0 (C3 on the four arrays), merged into one matrix:
        dict of tensors (barcode_rank, gem_group, indptr, indices, data) on rank 0, None elsewhere."""
        import torch

        v = m._mv.contents
        V, nnz = m.n_barcodes, m.nnz
        ranks, Vs = self.ctx.gatherv(v.d_barcode_rank, V * 4, np.uint32)
        ends, _ = self.ctx.gatherv(v.d_indptr + 8, V * 8, np.int64)  # column ends relative to the well's own block
        indices, NZs = self.ctx.gatherv(v.d_indices, nnz * 4, np.int32)
        data, _ = self.ctx.gatherv(v.d_data, nnz * 4, np.int32)
        if self.rank != 0:
            return None
        self._keep.extend([ranks, ends, indices, data])
        self.ctx.synchronize()
        tV, tNZ = sum(Vs), sum(NZs)
        dev = "cuda:%d" % self.device_index
        ends_t = self._tensor(ends.ptr, tV, "<i8")
        shift = torch.zeros(tV, dtype=torch.int64, device=dev)
        gg = torch.zeros(tV, dtype=torch.int32, device=dev)
        v0 = nz0 = 0
        for r in range(self.world):
            shift[v0:v0 + Vs[r]] = nz0
            gg[v0:v0 + Vs[r]] = r + 1
            v0 += Vs[r]
            nz0 += NZs[r]
        indptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), ends_t + shift])
        return dict(barcode_rank=self._tensor(ranks.ptr, tV, "<i4"), gem_group=gg, indptr=indptr,
                    indices=self._tensor(indices.ptr, tNZ, "<i4"), data=self._tensor(data.ptr, tNZ, "<i4"))


class CountPipeline:
    """barcode-correct -> UMI-dedup -> matrix for one shard of reads per rank: the order of the calls."""

    def __init__(self, backend, libs=(0,), force_collectives=False):
        self.be = backend
        self.libs = list(libs)
        self.world, self.rank = backend.world, backend.rank
        # exercise C1/C2/C3 even with one rank (tests: a 1-rank RCCL communicator on the single-GPU box)
        self.collective = self.world > 1 or force_collectives

    def correct_barcodes(self, shard):
        """cfg2: pass A + C1 + pass B.  shard['idx'] receives the corrected barcode ranks."""
        self.be.match_and_count(shard)
        if self.collective:
            self.be.allreduce_hist(self.libs, COUNTS_VALID)
        self.be.correct(shard)

    def run_wells(self, shard):
        """BASELINE configs[4]: every rank holds ONE WHOLE GEM well (gem group = rank + 1).  The wells are independent
        (no C1/C2: barcodes of different gem groups never meet), each rank runs the single-GPU path; the only exchange
        is the gather of the per-well CSC blocks.  Rank 0 returns the merged matrix as a dict of tensors
        (barcode_rank, gem_group, indptr, indices, data): column concatenation in (gem_group, barcode) order
        (barcode/src/lib.rs:119-124); other ranks return None."""
        self.be.match_and_count(shard)
        self.be.correct(shard)
        keys, n_keys = self.be.build_keys(shard)
        counts = self.be.count_keys(keys, n_keys)
        b, f, c = self.be.triplet_arrays(counts)
        m = self.be.assemble(b, f, c, counts.n_triplets)
        return self.be.gather_wells(m)

    def run(self, shard, dupinfo=None):
        """Full path.  Returns the device CSC on rank 0 (None elsewhere).  dupinfo: three device arrays of shard['n']
        entries (processed UMI u32, read count u32, flags u8) that receive the DupInfo of this rank's reads."""
        self.correct_barcodes(shard)
        if self.collective:
            self.be.allreduce_hist(self.libs, COUNTS_CORRECTED)   # matrix columns = barcodes seen on ANY rank
        if dupinfo is not None:
            counts = self.be.count_records(shard, dupinfo, sharded=self.collective)
        else:
            keys, n_keys = self.be.build_keys(shard)
            if self.collective:
                keys, n_keys = self.be.exchange_keys(keys, n_keys)
            counts = self.be.count_keys(keys, n_keys)
        if not self.collective:
            b, f, c = self.be.triplet_arrays(counts)
            return self.be.assemble(b, f, c, counts.n_triplets)
        arrs, total = self.be.gather_triplets(counts)
        if self.rank != 0:
            return None
        return self.be.assemble(arrs[0], arrs[1], arrs[2], total)
