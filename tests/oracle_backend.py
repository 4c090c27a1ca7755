"""Test-only stand-in for cellranger_amd.pipeline.HipBackend: the same backend protocol implemented
with numpy + the C oracle, its collectives (C1 all-reduce, C2 all-to-all by barcode range, C3 gather) with
torch.distributed/gloo, so that the multi-rank call sequence of CountPipeline can be exercised on CPUs and its
result compared with the single-process oracle.  The product's collectives are libcrgpu's (comm.hip).

TEST INFRASTRUCTURE: lives in tests/, never imported by the cellranger_amd package."""
import numpy as np
import torch

import oracle_lib as O
from cellranger_amd import engine as E
from cellranger_amd import synth as S
from cellranger_amd._lib import COUNTS_CORRECTED, COUNTS_VALID, FLAG_CB_HAS_N, FLAG_NONTXOMIC, MISS, NO_FEATURE


class _Counts:
    def __init__(self, bc, ft, ct):
        self.bc, self.ft, self.ct = bc, ft, ct
        self.n_triplets = len(bc)


class _Matrix:
    def __init__(self, rank, indptr, indices, data):
        self.barcode_rank, self.indptr, self.indices, self.data = rank, indptr, indices, data
        self.n_barcodes, self.nnz = len(rank), len(data)


class OracleBackend:
    def __init__(self, wl_packed, cb_len, n_features, umi_len, n_libs=1, mux_mask=0, dist=None):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.cb_len, self.umi_len, self.n_features, self.n_libs, self.mux_mask = cb_len, umi_len, n_features, n_libs, mux_mask
        self.canon_sorted = np.sort(np.asarray(wl_packed, dtype=np.uint32))
        self.canon_ascii = E.unpack_seqs(self.canon_sorted, cb_len)
        self.n_canon = len(self.canon_sorted)
        self.owl = O.Whitelist(self.canon_ascii)
        self.bits_bc = int(np.ceil(np.log2(max(self.n_canon, 1)))) if self.n_canon > 1 else 0
        while (1 << self.bits_bc) < self.n_canon:
            self.bits_bc += 1
        self.bits_feat = 0
        while (1 << self.bits_feat) < n_features:
            self.bits_feat += 1
        self.bits_lib = 0
        while (1 << self.bits_lib) < n_libs:
            self.bits_lib += 1
        self.bits_umi = 2 * umi_len
        self.sh_umi = 1
        self.sh_lib = 1 + self.bits_umi
        self.sh_feat = self.sh_lib + self.bits_lib
        self.sh_bc = self.sh_feat + self.bits_feat
        self.reset()

    # -- protocol ------------------------------------------------------------------------------------
    def reset(self):
        self.hist = {COUNTS_VALID: [np.zeros(self.n_canon, np.int32) for _ in range(self.n_libs)],
                     COUNTS_CORRECTED: [np.zeros(self.n_canon, np.int32) for _ in range(self.n_libs)]}

    # -- collectives: the semantics of crgpu_allreduce_counts / crgpu_exchange_keys_dev / crgpu_gather_triplets_dev ----
    def allreduce_hist(self, libs, which):
        for lib in libs:
            self.dist.all_reduce(torch.from_numpy(self.hist[which][lib]), op=self.dist.ReduceOp.SUM)

    def exchange_keys(self, keys, n_keys):
        part, send_counts = self.partition(keys, n_keys, self.world)
        send_t = torch.tensor(send_counts, dtype=torch.int64)
        recv_t = torch.empty(self.world, dtype=torch.int64)
        self.dist.all_to_all_single(recv_t, send_t)
        recv_counts = [int(x) for x in recv_t.tolist()]
        recv = np.zeros(sum(recv_counts), np.uint64)
        self.dist.all_to_all_single(torch.from_numpy(recv.view(np.int64)), torch.from_numpy(part[:n_keys].view(np.int64)),
                                    output_split_sizes=recv_counts, input_split_sizes=send_counts)
        return recv, len(recv)

    def _gatherv(self, arr):
        """rank 0: (concatenation in rank order, per-rank sizes); others: (None, None)"""
        src = torch.from_numpy(np.ascontiguousarray(arr))
        n = int(src.numel())
        sizes = torch.zeros(self.world, dtype=torch.int64)
        self.dist.all_gather_into_tensor(sizes, torch.tensor([n], dtype=torch.int64))
        sizes = [int(x) for x in sizes.tolist()]
        total = sum(sizes) if self.rank == 0 else 0
        dst = torch.empty(total, dtype=src.dtype)
        self.dist.all_to_all_single(dst, src, output_split_sizes=sizes if self.rank == 0 else [0] * self.world,
                                    input_split_sizes=[n] + [0] * (self.world - 1))
        return (dst.numpy(), sizes) if self.rank == 0 else (None, None)

    def gather_triplets(self, counts):
        outs = [self._gatherv(a.view(np.int32))[0] for a in (counts.bc, counts.ft, counts.ct)]
        if self.rank != 0:
            return None, 0
        return [o.view(np.uint32) for o in outs], len(outs[0])

    def gather_wells(self, m):
        ranks, Vs = self._gatherv(m.barcode_rank.view(np.int32))
        ends, _ = self._gatherv(m.indptr[1:].astype(np.int64))
        indices, NZs = self._gatherv(m.indices.astype(np.int32))
        data, _ = self._gatherv(m.data.astype(np.int32))
        if self.rank != 0:
            return None
        shift = np.repeat(np.concatenate([[0], np.cumsum(NZs)[:-1]]), Vs).astype(np.int64)
        gg = np.repeat(np.arange(1, self.world + 1), Vs).astype(np.int32)
        indptr = np.concatenate([[0], ends + shift]).astype(np.int64)
        return dict(barcode_rank=torch.from_numpy(ranks), gem_group=torch.from_numpy(gg), indptr=torch.from_numpy(indptr),
                    indices=torch.from_numpy(indices), data=torch.from_numpy(data))

    def match_and_count(self, shard):
        pk = shard["cb"]
        pos = np.searchsorted(self.canon_sorted, pk)
        pos = np.minimum(pos, self.n_canon - 1)
        hit = (self.canon_sorted[pos] == pk) & ((shard["flags"] & FLAG_CB_HAS_N) == 0)
        shard["idx"][:] = np.where(hit, pos, MISS).astype(np.uint32)
        lib = shard["flags"] & 0x0F
        for l in range(self.n_libs):
            np.add.at(self.hist[COUNTS_VALID][l], pos[hit & (lib == l)], 1)

    def correct(self, shard):
        seq, qual = S.to_ascii(shard["cb"], shard["cb_qualn"], self.cb_len)
        lib = shard["flags"] & 0x0F
        priors = []
        for l in range(self.n_libs):
            h = O.Hist()
            nz = np.nonzero(self.hist[COUNTS_VALID][l])[0]
            for r in nz:
                h.observe_by(bytes(self.canon_ascii[r]), int(self.hist[COUNTS_VALID][l][r]))
            priors.append(h)
        for i in np.nonzero(shard["idx"] == MISS)[0]:
            got = O.posterior_correct(self.owl, priors[lib[i]], bytes(seq[i]), qual[i])
            if got is not None:
                r = int(np.searchsorted(self.canon_sorted, E.pack_seqs([got])[0][0]))
                shard["idx"][i] = r
                self.hist[COUNTS_CORRECTED][lib[i]][r] += 1

    def build_keys(self, shard):
        n = shard["n"]
        seq, qual = S.to_ascii(shard["umi"], shard["umi_qualn"], self.umi_len)
        keep = np.zeros(n, bool)
        for i in range(n):
            keep[i] = (shard["idx"][i] != MISS and shard["feature"][i] != NO_FEATURE
                       and O.umi_is_valid(bytes(seq[i]), qual[i]))
        lib = (shard["flags"] & 0x0F).astype(np.uint64)
        nontx = ((shard["flags"] & FLAG_NONTXOMIC) != 0).astype(np.uint64)
        keys = ((shard["idx"].astype(np.uint64) << np.uint64(self.sh_bc)) | (shard["feature"].astype(np.uint64) << np.uint64(self.sh_feat))
                | (lib << np.uint64(self.sh_lib)) | (shard["umi"].astype(np.uint64) << np.uint64(self.sh_umi)) | nontx)
        keys = np.ascontiguousarray(keys[keep])
        return keys, len(keys)

    def balanced_bounds(self, n_ranks):
        tot = np.zeros(self.n_canon, np.int64)
        for which in (COUNTS_VALID, COUNTS_CORRECTED):
            for l in range(self.n_libs):
                tot += self.hist[which][l]
        cum = np.cumsum(tot)
        total = int(cum[-1]) if len(cum) else 0
        bounds = [0]
        for k in range(1, n_ranks):
            bounds.append(int(np.searchsorted(cum, total * k // n_ranks, side="right")))
        bounds.append(self.n_canon)
        return np.array(bounds, np.uint32)

    def partition(self, keys, n_keys, n_ranks):
        bounds = self.balanced_bounds(n_ranks)
        bc = (keys[:n_keys] >> np.uint64(self.sh_bc)).astype(np.int64)
        owner = np.searchsorted(bounds[1:].astype(np.int64), bc, side="right").astype(np.uint64)
        owner = np.minimum(owner, np.uint64(n_ranks - 1))
        order = np.argsort(owner, kind="stable")
        return np.ascontiguousarray(keys[:n_keys][order]), [int((owner == r).sum()) for r in range(n_ranks)]

    def count_keys(self, keys, n_keys):
        k = keys[:n_keys]
        bc = (k >> np.uint64(self.sh_bc)).astype(np.uint32)
        ft = ((k >> np.uint64(self.sh_feat)) & np.uint64((1 << self.bits_feat) - 1)).astype(np.uint32)
        lib = ((k >> np.uint64(self.sh_lib)) & np.uint64((1 << self.bits_lib) - 1)).astype(np.uint32)
        umi = ((k >> np.uint64(self.sh_umi)) & np.uint64((1 << self.bits_umi) - 1)).astype(np.uint32)
        nontx = (k & np.uint64(1)).astype(np.uint8)
        umi_ascii = E.unpack_seqs(umi, self.umi_len)
        out = []
        order = np.lexsort((lib, bc))
        i = 0
        while i < len(order):
            j = i
            while j < len(order) and bc[order[j]] == bc[order[i]] and lib[order[j]] == lib[order[i]]:
                j += 1
            sel = order[i:j]
            enabled = not ((self.mux_mask >> int(lib[sel[0]])) & 1)
            _, uc = O.mark_dups_group(umi_ascii[sel], np.ones(len(sel), np.uint8), ft[sel], utype=nontx[sel],
                                      qname=np.arange(len(sel), dtype=np.uint64), umi_correction=enabled)
            for f in uc["feature_idx"]:
                out.append((int(bc[sel[0]]), int(f)))
            i = j
        out.sort()
        tb, tf, tc = [], [], []
        for key in out:
            if tb and (tb[-1], tf[-1]) == key:
                tc[-1] += 1
            else:
                tb.append(key[0]); tf.append(key[1]); tc.append(1)
        return _Counts(np.array(tb, np.uint32), np.array(tf, np.uint32), np.array(tc, np.uint32))

    def triplet_arrays(self, counts):
        return counts.bc, counts.ft, counts.ct

    def assemble(self, bc, ft, ct, n_triplets):
        seen = np.zeros(self.n_canon, bool)
        for which in (COUNTS_VALID, COUNTS_CORRECTED):
            for l in range(self.n_libs):
                seen |= self.hist[which][l] != 0
        rank = np.nonzero(seen)[0].astype(np.uint32)
        bc, ft, ct = bc[:n_triplets], ft[:n_triplets], ct[:n_triplets]
        assert (np.diff(bc.astype(np.int64)) >= 0).all(), "triplets must arrive sorted by barcode"
        indptr = np.searchsorted(bc, np.append(rank, np.uint32(0xFFFFFFFF))).astype(np.int64)
        indptr[-1] = n_triplets
        return _Matrix(rank, indptr, ft.astype(np.int32), ct.astype(np.int32))
