"""Thin Python host over the C ABI (include/crgpu.h): device buffers + one method per entry point.

The product path: every method ends in a libcrgpu call; there is no CPU implementation behind
any of them (a missing library / GPU raises CrgpuError).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (COUNTS_CORRECTED, COUNTS_PRIOR, COUNTS_VALID, MISS, NO_FEATURE, CrgpuError, MatrixView,
                   Records, SynthOut, SynthParams, ptr)

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def pack_seqs(seqs, length=None):
    """list of str/bytes or (n, len) uint8 ASCII array -> (packed uint32[n], len).  ACGT only."""
    a = ascii_matrix(seqs, length)
    n, L = a.shape
    if L > 16:
        raise ValueError("sequences longer than 16 bases do not fit 32 bits")
    code = np.full(256, 255, dtype=np.uint8)
    code[_ACGT] = np.arange(4, dtype=np.uint8)
    c = code[a]
    if (c == 255).any():
        raise ValueError("non-ACGT base")
    out = np.zeros(n, dtype=np.uint32)
    for j in range(L):
        out = (out << np.uint32(2)) | c[:, j].astype(np.uint32)
    return out, L


def unpack_seqs(packed, length):
    packed = np.asarray(packed, dtype=np.uint32)
    out = np.zeros((len(packed), length), dtype=np.uint8)
    for j in range(length):
        out[:, j] = _ACGT[(packed >> np.uint32(2 * (length - 1 - j))) & np.uint32(3)]
    return out


def compute_feature_dist(counts, feature_types=None):
    """compute_feature_dist (feature_checker.rs:8-50) through the C ABI (host code)"""
    c = np.ascontiguousarray(counts, dtype=np.int64)
    t = None if feature_types is None else np.ascontiguousarray(feature_types, dtype=np.uint32)
    out = np.zeros(len(c), np.float64)
    rc = _lib.load().crgpu_compute_feature_dist(ptr(c), ptr(t), len(c), ptr(out))
    if rc != 0:
        raise _lib.CrgpuError(rc, "crgpu_compute_feature_dist")
    return out


def synth_rows_host(seed, first, n, feature, feat_seq, L, offset, row_stride, err=0.005, n_rate=0.0005):
    """host twin of Context.synth_rows: (seq rows, qual rows) uint8 (n, row_stride)"""
    fs = np.ascontiguousarray(feat_seq, dtype=np.uint64)
    ft = None if feature is None else np.ascontiguousarray(feature, dtype=np.uint32)
    s, q = np.zeros((n, row_stride), np.uint8), np.zeros((n, row_stride), np.uint8)
    rc = _lib.load().crgpu_synth_rows_host(seed, first, n, ptr(ft), ptr(fs), len(fs), L, offset, row_stride,
                                           int(round(err * 65536)), int(round(n_rate * (1 << 20))), ptr(s), ptr(q))
    if rc != 0:
        raise _lib.CrgpuError(rc, "crgpu_synth_rows_host")
    return s, q


def compile_feature_pattern(pattern, length):
    """compile_pattern (feature_extraction.rs:307-343): the regular expression as text, None for a rejected pattern"""
    buf = C.create_string_buffer(4096)
    rc = _lib.load().crgpu_compile_feature_pattern(pattern.encode(), length, buf, 4096)
    return buf.value.decode() if rc == 0 else None


def ascii_matrix(seqs, length=None):
    if isinstance(seqs, np.ndarray) and seqs.dtype == np.uint8 and seqs.ndim == 2:
        return np.ascontiguousarray(seqs)
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    if length is None:
        length = len(bs[0]) if bs else 0
    if any(len(b) != length for b in bs):
        raise ValueError("sequences must all have length %d" % length)
    return np.frombuffer(b"".join(bs), dtype=np.uint8).reshape(len(bs), length).copy()


class DeviceArray:
    """A device allocation owned through crgpu_malloc/crgpu_free (or adopted from a library call that returns a
    library-owned buffer to be released with crgpu_free: `adopt`)."""

    def __init__(self, ctx, shape, dtype, adopt=None):
        self.ctx = ctx
        self.shape = (shape,) if np.isscalar(shape) else tuple(shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        if adopt is not None:
            self.ptr = int(adopt)
            return
        p = C.c_void_p()
        ctx._check(ctx.L.crgpu_malloc(ctx.h, C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def to_host(self, count=None):
        n = self.size if count is None else int(count)
        out = np.empty(n, dtype=self.dtype)
        if n:
            self.ctx._check(self.ctx.L.crgpu_memcpy_d2h(self.ctx.h, ptr(out), self.ptr, n * self.dtype.itemsize))
        if count is None:
            out = out.reshape(self.shape)
        return out

    def upload(self, arr):
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.nbytes <= self.nbytes
        if a.nbytes:
            self.ctx._check(self.ctx.L.crgpu_memcpy_h2d(self.ctx.h, self.ptr, ptr(a), a.nbytes))
        return self

    def zero(self):
        self.ctx._check(self.ctx.L.crgpu_memset(self.ctx.h, self.ptr, 0, self.nbytes))
        return self

    def free(self):
        if self.ptr is not None and self.ctx.h:
            self.ctx.L.crgpu_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def __cuda_array_interface__(self):
        # lets torch.as_tensor(..., device="cuda") alias the buffer for collectives
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2}


def _p(x):
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return C.c_void_p(x.ptr)
    if hasattr(x, "data_ptr"):  # torch tensor
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


class Matrix:
    """Host view of crgpu_matrix (the arrays write_matrix_h5 stores; count_matrix.rs:382-448)."""

    def __init__(self, ctx, mv_ptr):
        self.ctx, self._mv = ctx, mv_ptr
        m = mv_ptr.contents
        self.n_barcodes, self.nnz = int(m.n_barcodes), int(m.nnz)
        self.n_features, self.cb_len = int(m.n_features), int(m.cb_len)

        def arr(p, dtype, n):
            if n == 0:
                return np.zeros(0, dtype=dtype)
            return np.frombuffer((C.c_char * (n * np.dtype(dtype).itemsize)).from_address(p), dtype=dtype).copy()

        self.barcode_rank = arr(m.barcode_rank, np.uint32, self.n_barcodes)
        self.barcode_seq = arr(m.barcode_seq, np.uint32, self.n_barcodes)
        self.indptr = arr(m.indptr, np.int64, self.n_barcodes + 1)
        self.indices = arr(m.indices, np.int32, self.nnz)
        self.data = arr(m.data, np.int32, self.nnz)
        self.gem_group = arr(m.gem_group, np.uint16, self.n_barcodes) if m.gem_group else None
        # barcodes longer than 16 bases (segmented constructs): barcode_seq = the first 16 bases, barcode_seq_hi = the rest
        self.barcode_seq_hi = arr(m.barcode_seq_hi, np.uint32, self.n_barcodes) if m.barcode_seq_hi else None

    def barcodes_ascii(self):
        if self.barcode_seq_hi is None:
            return unpack_seqs(self.barcode_seq, self.cb_len)
        return np.concatenate([unpack_seqs(self.barcode_seq, 16), unpack_seqs(self.barcode_seq_hi, self.cb_len - 16)], axis=1)

    def write_mtx(self, mtx_path, barcodes_path=None, metadata_line='%metadata_json: {"format_version": 2}',
                  gem_group=1):
        self.ctx._check(self.ctx.L.crgpu_write_mtx(self.ctx.h, self._mv, metadata_line.encode(),
                                                   None if mtx_path is None else str(mtx_path).encode(),
                                                   None if barcodes_path is None else str(barcodes_path).encode(),
                                                   gem_group))

    def free(self):
        if self._mv is not None and self.ctx.h:
            self.ctx.L.crgpu_matrix_free(self.ctx.h, self._mv)
        self._mv = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MatrixDev:
    """crgpu_matrix_dev: device-resident CSC (barcode_rank, indptr, indices, data)."""

    def __init__(self, ctx, mv_ptr):
        self.ctx, self._mv = ctx, mv_ptr
        self.n_barcodes, self.nnz = int(mv_ptr.contents.n_barcodes), int(mv_ptr.contents.nnz)

    def download(self):
        V, nnz = self.n_barcodes, self.nnz
        rank, indptr = np.zeros(V, np.uint32), np.zeros(V + 1, np.int64)
        indices, data = np.zeros(nnz, np.int32), np.zeros(nnz, np.int32)
        self.ctx._check(self.ctx.L.crgpu_matrix_dev_download(self.ctx.h, self._mv, ptr(rank), ptr(indptr), ptr(indices), ptr(data)))
        return rank, indptr, indices, data

    def free(self):
        if self._mv is not None and self.ctx.h:
            self.ctx.L.crgpu_matrix_dev_free(self.ctx.h, self._mv)
        self._mv = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Counts:
    """crgpu_counts: sorted (barcode, feature, count) triplets + the molecule table."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        nt, nm = C.c_uint64(), C.c_uint64()
        ctx._check(ctx.L.crgpu_counts_info(ctx.h, handle, C.byref(nt), C.byref(nm)))
        self.n_triplets, self.n_molecules = nt.value, nm.value

    def triplets(self):
        n = self.n_triplets
        bc, ft, ct = (np.zeros(n, np.uint32) for _ in range(3))
        if n:
            self.ctx._check(self.ctx.L.crgpu_counts_triplets(self.ctx.h, self.h, ptr(bc), ptr(ft), ptr(ct)))
        return bc, ft, ct

    def triplets_dev(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.ctx._check(self.ctx.L.crgpu_counts_triplets_dev(self.ctx.h, self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def molecules(self):
        n = self.n_molecules
        out = dict(bc=np.zeros(n, np.uint32), lib=np.zeros(n, np.uint8), feature=np.zeros(n, np.uint32),
                   umi=np.zeros(n, np.uint32), read_count=np.zeros(n, np.uint32), utype=np.zeros(n, np.uint8))
        if n:
            self.ctx._check(self.ctx.L.crgpu_counts_molecules(self.ctx.h, self.h, ptr(out["bc"]), ptr(out["lib"]),
                                                              ptr(out["feature"]), ptr(out["umi"]),
                                                              ptr(out["read_count"]), ptr(out["utype"])))
        return out

    def molecule_info(self, gem_group=1):
        """the datasets MoleculeInfoWriter::fill appends (cr_h5/src/molecule_info.rs:972-998)"""
        n = self.n_molecules
        out = dict(gem_group=np.zeros(n, np.uint16), barcode_idx=np.zeros(n, np.uint64), feature_idx=np.zeros(n, np.uint32),
                   library_idx=np.zeros(n, np.uint16), umi=np.zeros(n, np.uint32), count=np.zeros(n, np.uint32),
                   umi_type=np.zeros(n, np.uint32))
        if n:
            self.ctx._check(self.ctx.L.crgpu_counts_molecule_info(
                self.ctx.h, self.h, gem_group, ptr(out["gem_group"]), ptr(out["barcode_idx"]), ptr(out["feature_idx"]),
                ptr(out["library_idx"]), ptr(out["umi"]), ptr(out["count"]), ptr(out["umi_type"])))
        return out

    def probe_idx(self):
        """UmiCount::probe_idx per molecule, in the order of molecules() / molecule_info() (records with d_probe_idx)"""
        out = np.full(self.n_molecules, _lib.NO_PROBE, np.int32)
        if self.n_molecules:
            self.ctx._check(self.ctx.L.crgpu_counts_probe_idx(self.ctx.h, self.h, ptr(out)))
        return out

    def barcode_summary(self, rank_lo=0, rank_hi=0xFFFFFFFF):
        """BarcodeSummary rows (cr_lib/src/aligner.rs:33-68) of the barcode ranks in [rank_lo, rank_hi), ordered by
        (library, rank): a numpy record array of _lib.BARCODE_SUMMARY_DTYPE"""
        n = C.c_uint64()
        L, h = self.ctx.L, self.ctx.h
        self.ctx._check(L.crgpu_counts_barcode_summary(h, self.h, rank_lo, rank_hi, None, 0, C.byref(n)))
        rows = np.zeros(n.value, _lib.BARCODE_SUMMARY_DTYPE)
        if n.value:
            self.ctx._check(L.crgpu_counts_barcode_summary(h, self.h, rank_lo, rank_hi, ptr(rows), n.value, C.byref(n)))
        return rows

    def free(self):
        if self.h is not None and self.ctx.h:
            self.ctx.L.crgpu_counts_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def get_unique_id():
    """rendezvous token of an RCCL communicator (rank 0 makes it and ships the bytes to the other processes)"""
    L = _lib.load()
    buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    rc = L.crgpu_get_unique_id(buf)
    if rc != 0:
        raise CrgpuError(rc, (L.crgpu_last_error(None) or b"").decode())
    return buf.raw


def local_group_id(n_ranks):
    """rendezvous token for n_ranks contexts inside THIS process (one host thread each)"""
    L = _lib.load()
    buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    rc = L.crgpu_local_group_id(n_ranks, buf)
    if rc != 0:
        raise CrgpuError(rc, (L.crgpu_last_error(None) or b"").decode())
    return buf.raw


class Context:
    """crgpu_ctx: one per (process, device, rank).  n_ranks / rank / unique_id: see crgpu_create in crgpu.h."""

    def __init__(self, device=0, n_ranks=1, rank=0, unique_id=None):
        self.L = _lib.load()
        h = C.c_void_p()
        uid = None
        if unique_id is not None:
            assert len(unique_id) == _lib.UNIQUE_ID_BYTES
            uid = C.create_string_buffer(bytes(unique_id), _lib.UNIQUE_ID_BYTES)
        rc = self.L.crgpu_create(C.byref(h), device, n_ranks, rank, uid)
        if rc != 0:
            raise CrgpuError(rc, (self.L.crgpu_last_error(None) or b"").decode())
        self.h = h
        self.device = device
        self.n_ranks, self.rank = n_ranks, rank
        self.cb_len = None
        self.n_canon = None

    # ---- options / collectives (crgpu.h "collectives") ---------------------------------------------
    def set_option(self, option, value):
        self._check(self.L.crgpu_set_option(self.h, option, int(value)))

    def trust_unchanged_buffers(self, on=True):
        """promise that buffers handed from one call to the next are only written through this context"""
        self.set_option(_lib.OPT_BUFFERS_UNCHANGED_BETWEEN_CALLS, 1 if on else 0)

    def invalidate(self):
        self._check(self.L.crgpu_invalidate(self.h))

    def stat(self, which=0):
        v = C.c_uint64()
        self._check(self.L.crgpu_get_stat(self.h, which, C.byref(v)))
        return v.value

    def barrier(self):
        self._check(self.L.crgpu_barrier(self.h))

    def allreduce_counts(self, lib=-1, which=COUNTS_VALID):
        self._check(self.L.crgpu_allreduce_counts(self.h, lib, which))

    def allreduce_max(self, value):
        v = C.c_double(float(value))
        self._check(self.L.crgpu_allreduce_max_f64(self.h, C.byref(v)))
        return v.value

    def allreduce_sum(self, values):
        """element-wise sum over the ranks of a small int64 host array, in place"""
        assert values.dtype == np.int64 and values.flags["C_CONTIGUOUS"]
        self._check(self.L.crgpu_allreduce_sum_i64(self.h, ptr(values), len(values)))
        return values

    def exchange_keys(self, d_keys, n_keys):
        """C2: (DeviceArray of this rank's keys, n, bounds)"""
        p, n = C.c_void_p(), C.c_uint64()
        bounds = np.zeros(self.n_ranks + 1, np.uint32)
        self._check(self.L.crgpu_exchange_keys_dev(self.h, _p(d_keys), n_keys, C.byref(p), C.byref(n), ptr(bounds)))
        return DeviceArray(self, max(n.value, 1), np.uint64, adopt=p.value), n.value, bounds

    def gatherv(self, d_src, nbytes, dtype, root=0):
        """C3 for one array: (DeviceArray on root / None elsewhere, per-rank element counts on root)"""
        p = C.c_void_p()
        per = np.zeros(self.n_ranks, np.uint64)
        self._check(self.L.crgpu_gatherv_dev(self.h, _p(d_src), nbytes, root, C.byref(p), ptr(per)))
        if self.rank != root:
            return None, None
        item = np.dtype(dtype).itemsize
        counts = [int(x) // item for x in per]
        return DeviceArray(self, max(sum(counts), 1), dtype, adopt=p.value), counts

    def gather_triplets(self, counts, root=0):
        """C3: on root ((bc, feature, count) DeviceArrays, n_total), elsewhere (None, 0)"""
        a, b, c, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self.L.crgpu_gather_triplets_dev(self.h, counts.h, root, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        if self.rank != root:
            return None, 0
        return tuple(DeviceArray(self, max(n.value, 1), np.uint32, adopt=x.value) for x in (a, b, c)), n.value

    def close(self):
        if getattr(self, "h", None):
            self.L.crgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise CrgpuError(rc, (self.L.crgpu_last_error(self.h) or b"").decode())

    # ---- memory -------------------------------------------------------------------------------
    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype):
        return DeviceArray(self, shape, dtype).zero()

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        return DeviceArray(self, a.shape, a.dtype).upload(a)

    def stream_handle(self):
        """the hipStream_t of the context as an integer (torch.cuda.ExternalStream)"""
        return int(self.L.crgpu_stream(self.h) or 0)

    def synchronize(self):
        self._check(self.L.crgpu_synchronize(self.h))

    def trim(self):
        self._check(self.L.crgpu_trim(self.h))

    # ---- timing -------------------------------------------------------------------------------
    def timing(self, on=True):
        self._check(self.L.crgpu_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self._check(self.L.crgpu_timing_reset(self.h))

    def timing_get(self):
        ms = np.zeros(len(_lib.T_NAMES), np.float64)
        ln = np.zeros(len(_lib.T_NAMES), np.uint64)
        un = np.zeros(len(_lib.T_NAMES), np.uint64)
        self._check(self.L.crgpu_timing_get(self.h, ptr(ms), ptr(ln), ptr(un)))
        return {k: (float(ms[i]), int(ln[i]), int(un[i])) for i, k in enumerate(_lib.T_NAMES)}

    # ---- whitelist ------------------------------------------------------------------------------
    def set_whitelist(self, lib, keys, canon=None, translate_to=None, length=None):
        """keys/canon: list of str, ASCII matrix, or packed uint32 (then `length` is required)."""
        def packed(x):
            if isinstance(x, np.ndarray) and x.dtype == np.uint32:
                return np.ascontiguousarray(x), length
            return pack_seqs(x)
        pk, L1 = packed(keys)
        if canon is None:
            pc, L2 = pk, L1
        else:
            pc, L2 = packed(canon)
        assert L1 == L2 and L1 is not None
        tt = None if translate_to is None else np.ascontiguousarray(translate_to, dtype=np.uint32)
        self._check(self.L.crgpu_set_whitelist_packed(self.h, lib, ptr(pk), len(pk), L1, ptr(pc), len(pc), ptr(tt)))
        self.cb_len, self.n_canon = L1, len(pc)

    def set_whitelist_ascii(self, lib, keys, canon=None, translate_to=None):
        k = ascii_matrix(keys)
        c = k if canon is None else ascii_matrix(canon)
        tt = None if translate_to is None else np.ascontiguousarray(translate_to, dtype=np.uint32)
        self._check(self.L.crgpu_set_whitelist(self.h, lib, k.tobytes(), k.shape[0], k.shape[1], c.tobytes(),
                                               c.shape[0], ptr(tt)))
        self.cb_len, self.n_canon = k.shape[1], c.shape[0]

    def canon_order(self):
        order = np.zeros(self.n_canon, np.uint32)
        seqs = np.zeros(self.n_canon, np.uint32)
        self._check(self.L.crgpu_get_canon_order(self.h, ptr(order), ptr(seqs)))
        return order, seqs

    # ---- barcode stage ---------------------------------------------------------------------------
    def pack(self, d_seq, d_qual, n, length, d_packed, d_qualn, d_flags=None):
        self._check(self.L.crgpu_pack_dev(self.h, _p(d_seq), _p(d_qual), n, length, _p(d_packed), _p(d_qualn), _p(d_flags)))

    def pack_rows(self, d_seq_rows, d_qual_rows, n, row_stride, offset, length, d_packed, d_qualn, d_flags=None):
        """pack bases [offset, offset + length) of every row_stride-byte read row (barcode / UMI ranges of R1)"""
        self._check(self.L.crgpu_pack_rows_dev(self.h, _p(d_seq_rows), _p(d_qual_rows), n, row_stride, offset, length,
                                               _p(d_packed), _p(d_qualn), _p(d_flags)))

    def match_and_count(self, d_cb, d_flags, n, d_idx_out):
        self._check(self.L.crgpu_match_and_count_dev(self.h, _p(d_cb), _p(d_flags), n, _p(d_idx_out)))

    def set_posterior(self, max_expected_barcode_errors, bc_confidence_threshold):
        self._check(self.L.crgpu_set_posterior(self.h, float(max_expected_barcode_errors), float(bc_confidence_threshold)))

    def correct(self, d_cb, d_qualn, d_flags, n, d_idx_inout, d_corrected_out=None):
        self._check(self.L.crgpu_correct_dev(self.h, _p(d_cb), _p(d_qualn), _p(d_flags), n, _p(d_idx_inout), _p(d_corrected_out)))

    def get_counts(self, lib, which=COUNTS_VALID):
        out = np.zeros(self.n_canon, np.uint32)
        self._check(self.L.crgpu_get_counts(self.h, lib, which, ptr(out)))
        return out

    def set_counts(self, lib, which, counts):
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        assert len(c) == self.n_canon
        self._check(self.L.crgpu_set_counts(self.h, lib, which, ptr(c)))

    def reset_counts(self):
        self._check(self.L.crgpu_reset_counts(self.h))

    def counts_dev(self, lib, which):
        p = C.c_void_p()
        self._check(self.L.crgpu_counts_dev(self.h, lib, which, C.byref(p)))
        return p.value

    def barcode_correction_metrics(self, lib=0):
        """BARCODE_CORRECTION summary pieces of one library (barcode_correction.rs:409-448) as a dict"""
        m = _lib.BcCorrectionMetrics()
        self._check(self.L.crgpu_barcode_correction_metrics(self.h, lib, C.byref(m)))
        return {f: getattr(m, f) for f, _ in _lib.BcCorrectionMetrics._fields_}

    def total_barcode_counts(self, min_reads_to_report_bc=1000):
        """(ranks, counts) of total_barcode_counts restricted to whitelist barcodes (barcode_correction.rs:360,380-390)"""
        n = C.c_uint64()
        self._check(self.L.crgpu_total_barcode_counts(self.h, min_reads_to_report_bc, None, None, 0, C.byref(n)))
        ranks, counts = np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint64)
        if n.value:
            self._check(self.L.crgpu_total_barcode_counts(self.h, min_reads_to_report_bc, ptr(ranks), ptr(counts), n.value, C.byref(n)))
        return ranks, counts

    def match_and_count_host(self, lib, seq_ascii, qual):
        s = ascii_matrix(seq_ascii)
        q = None if qual is None else np.ascontiguousarray(qual, dtype=np.uint8)
        idx = np.zeros(s.shape[0], np.uint32)
        self._check(self.L.crgpu_match_and_count(self.h, lib, ptr(s), ptr(q), s.shape[0], ptr(idx)))
        return idx

    def correct_host(self, lib, seq_ascii, qual, idx):
        s = ascii_matrix(seq_ascii)
        q = None if qual is None else np.ascontiguousarray(qual, dtype=np.uint8)
        idx = np.ascontiguousarray(idx, dtype=np.uint32).copy()
        flag = np.zeros(s.shape[0], np.uint8)
        self._check(self.L.crgpu_correct(self.h, lib, ptr(s), ptr(q), s.shape[0], ptr(idx), ptr(flag)))
        return idx, flag

    # ---- count stage -------------------------------------------------------------------------------
    def set_key_layout(self, n_features, umi_len, n_libs=1, multiplexing_lib_mask=0):
        self._check(self.L.crgpu_set_key_layout(self.h, n_features, umi_len, n_libs, multiplexing_lib_mask))
        self.n_features, self.umi_len = n_features, umi_len

    def set_target_filter(self, on_target=None, min_read_count=0):
        """targeted-panel UMI filter (mark_dups.rs:311-320); None switches it off"""
        a = None if on_target is None else np.ascontiguousarray(on_target, dtype=np.uint8)
        self._check(self.L.crgpu_set_target_filter(self.h, ptr(a), 0 if a is None else len(a), int(min_read_count)))

    def records(self, n, umi_len, d_bc_idx, d_umi, d_umi_qualn, d_feature, d_flags=None, d_umi_len=None, d_probe_idx=None):
        r = Records()
        r.n, r.umi_len = n, umi_len
        r.d_bc_idx, r.d_umi, r.d_umi_qualn = _p(d_bc_idx), _p(d_umi), _p(d_umi_qualn)
        r.d_feature, r.d_flags, r.d_umi_len = _p(d_feature), _p(d_flags), _p(d_umi_len)
        r.d_probe_idx = _p(d_probe_idx)
        return r

    def count_host(self, n_features, bc_idx, umi, umi_qualn, feature, flags=None, umi_len_per_read=None, probe_idx=None,
                   want_dupinfo=True, want_counts=False):
        """crgpu_count_host: records in HOST arrays -> (Matrix, per-read crgpu_dupinfo records or None[, Counts])"""
        def h(a, dt):
            return None if a is None else np.ascontiguousarray(a, dtype=dt)
        bc_idx, umi, feature = h(bc_idx, np.uint32), h(umi, np.uint32), h(feature, np.uint32)
        umi_qualn, flags, ulen, probe = h(umi_qualn, np.uint8), h(flags, np.uint8), h(umi_len_per_read, np.uint8), h(probe_idx, np.int32)
        n = len(bc_idx)
        r = Records()
        r.n, r.umi_len = n, (umi_qualn.shape[1] if umi_qualn.ndim == 2 else self.umi_len)
        r.d_bc_idx, r.d_umi, r.d_umi_qualn, r.d_feature = ptr(bc_idx), ptr(umi), ptr(umi_qualn), ptr(feature)
        r.d_flags, r.d_umi_len, r.d_probe_idx = ptr(flags), ptr(ulen), ptr(probe)
        dup = np.zeros(n, _lib.DUPINFO_DTYPE) if want_dupinfo else None
        mv = C.POINTER(MatrixView)()
        ch = C.c_void_p()
        self._check(self.L.crgpu_count_host(self.h, C.byref(r), n_features, C.byref(mv), ptr(dup) if n else None,
                                            C.byref(ch) if want_counts else None))
        m = Matrix(self, mv)
        return (m, dup, Counts(self, ch)) if want_counts else (m, dup)

    def set_umi_min_len(self, umi_min_len):
        """per-read UMI lengths umi_min_len .. umi_len (after set_key_layout)"""
        self._check(self.L.crgpu_set_umi_min_len(self.h, umi_min_len))

    def pack_rows_var(self, d_seq_rows, d_qual_rows, d_read_len, n, row_stride, offset, length, min_length, d_packed, d_qualn, d_len):
        """UmiExtractor::extract_umi on read rows: per-read length max(min(read_len - offset, length), min_length)"""
        self._check(self.L.crgpu_pack_rows_var_dev(self.h, _p(d_seq_rows), _p(d_qual_rows), _p(d_read_len), n, row_stride, offset,
                                                   length, min_length, _p(d_packed), _p(d_qualn), _p(d_len)))

    def build_keys(self, recs, d_keys_out):
        n = C.c_uint64()
        self._check(self.L.crgpu_build_keys_dev(self.h, C.byref(recs), _p(d_keys_out), C.byref(n)))
        return n.value

    def partition_keys(self, d_keys, n, n_ranks, d_keys_out, bounds=None):
        counts = np.zeros(n_ranks, np.uint64)
        b = None if bounds is None else np.ascontiguousarray(bounds, dtype=np.uint32)
        assert b is None or len(b) == n_ranks + 1
        self._check(self.L.crgpu_partition_keys_dev(self.h, _p(d_keys), n, n_ranks, ptr(b), _p(d_keys_out), ptr(counts)))
        return counts

    def balanced_bounds(self, n_ranks):
        b = np.zeros(n_ranks + 1, np.uint32)
        self._check(self.L.crgpu_balanced_bounds(self.h, n_ranks, ptr(b)))
        return b

    def count_keys(self, d_keys, n_keys):
        h = C.c_void_p()
        self._check(self.L.crgpu_count_keys_dev(self.h, _p(d_keys), n_keys, C.byref(h)))
        return Counts(self, h)

    def count_records(self, recs, d_processed_umi=None, d_read_count=None, d_dupflags=None):
        """dedup from the records + per-read DupInfo (device output arrays, any may be None)"""
        h = C.c_void_p()
        self._check(self.L.crgpu_count_records_dev(self.h, C.byref(recs), C.byref(h), _p(d_processed_umi), _p(d_read_count),
                                                   _p(d_dupflags)))
        return Counts(self, h)

    def count_records_sharded(self, recs, d_processed_umi=None, d_read_count=None, d_dupflags=None):
        """collective: dedup of one well sharded over the ranks, DupInfo of this rank's own reads (crgpu.h)"""
        h = C.c_void_p()
        self._check(self.L.crgpu_count_records_sharded_dev(self.h, C.byref(recs), C.byref(h), _p(d_processed_umi), _p(d_read_count),
                                                           _p(d_dupflags)))
        return Counts(self, h)

    def enable_barcode_summary(self, on=True):
        """count_keys keeps what Counts.barcode_summary needs (count_records always does)"""
        self._check(self.L.crgpu_enable_barcode_summary(self.h, int(bool(on))))

    def write_barcode_summary_csv(self, rows, path, gem_group=1, library_types=(("Gene Expression", 0),)):
        """barcode_summary.csv of ALIGN_AND_COUNT (align_and_count.rs:806-817).  library_types[lib] = (display
        string of the LibraryType, its rank in the enum's order); libraries of equal rank are one type."""
        rows = np.ascontiguousarray(rows, dtype=_lib.BARCODE_SUMMARY_DTYPE)
        n = len(library_types)
        order = np.array([t[1] for t in library_types], np.uint32)
        names = (C.c_char_p * n)(*[t[0].encode() for t in library_types])
        self._check(self.L.crgpu_write_barcode_summary_csv(self.h, ptr(rows), len(rows), gem_group, ptr(order), names, n,
                                                           path.encode()))

    def shard_metrics(self, d_cb, d_cb_qualn, cb_len, d_umi, d_umi_qualn, umi_len, d_idx, n):
        """MAKE_SHARD's barcode / UMI read metrics as a dict of counts (make_shard_metrics.rs:263-332)"""
        m = _lib.ShardMetrics()
        self._check(self.L.crgpu_shard_metrics_dev(self.h, _p(d_cb), _p(d_cb_qualn), cb_len, _p(d_umi), _p(d_umi_qualn), umi_len,
                                                   _p(d_idx), n, C.byref(m)))
        return {f: int(getattr(m, f)) for f in _lib.SHARD_METRIC_FIELDS}

    def rows_metrics(self, d_seq_rows, d_qual_rows, n, row_stride, d_len=None):
        """frac_n_bases / frac_q30_bases of one read of the pair over FASTQ rows, as a dict of counts"""
        m = _lib.RowsMetrics()
        self._check(self.L.crgpu_rows_metrics_dev(self.h, _p(d_seq_rows), _p(d_qual_rows), _p(d_len), n, row_stride, C.byref(m)))
        return {f: int(getattr(m, f)) for f, _ in _lib.RowsMetrics._fields_}

    def homopolymer_metrics(self, d_r1_rows, r1_stride, n, d_r2_rows=None, r2_stride=0, d_r1_len=None, d_r2_len=None, run_len=15):
        """{A,C,G,T}_perfect_homopolymer: reads whose R1 or R2 holds run_len equal bases in a row"""
        out = np.zeros(4, np.uint64)
        self._check(self.L.crgpu_homopolymer_metrics_dev(self.h, _p(d_r1_rows), r1_stride, _p(d_r1_len), _p(d_r2_rows), r2_stride,
                                                         _p(d_r2_len), n, run_len, ptr(out)))
        return dict(zip("ACGT", (int(x) for x in out)))

    def fastq_to_rows(self, d_text, n_bytes, row_stride, max_records, d_seq_rows, d_qual_rows, d_len=None):
        """FASTQ text (device) -> rows; returns the number of records"""
        n = C.c_uint64()
        self._check(self.L.crgpu_fastq_to_rows_dev(self.h, _p(d_text), n_bytes, row_stride, max_records, _p(d_seq_rows), _p(d_qual_rows),
                                                   _p(d_len), C.byref(n)))
        return n.value

    def sum_matrices(self, a, b):
        """CountMatrix.merge: element-wise sum of two matrices of the same shape"""
        mv = C.POINTER(MatrixView)()
        self._check(self.L.crgpu_sum_matrices(self.h, a._mv, b._mv, C.byref(mv)))
        return Matrix(self, mv)

    def select_barcodes(self, m, cols):
        """CountMatrix.select_barcodes: the given columns in the given order"""
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        mv = C.POINTER(MatrixView)()
        self._check(self.L.crgpu_select_barcodes(self.h, m._mv, ptr(cols), len(cols), C.byref(mv)))
        return Matrix(self, mv)

    def sum_matrices_dev(self, a, b):
        """element-wise sum of two device CSCs over the same columns"""
        mv = C.POINTER(_lib.MatrixDevView)()
        self._check(self.L.crgpu_sum_matrices_dev(self.h, a._mv, b._mv, C.byref(mv)))
        return MatrixDev(self, mv)

    def select_barcodes_dev(self, m, cols):
        """the given columns of a device CSC in the given order"""
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        mv = C.POINTER(_lib.MatrixDevView)()
        self._check(self.L.crgpu_select_barcodes_dev(self.h, m._mv, ptr(cols), len(cols), C.byref(mv)))
        return MatrixDev(self, mv)

    def trim_molecule_barcodes(self, d_barcode_idx, n_molecules, n_barcodes, pass_filter_idx=None, pass_only=False, offset=0):
        """MERGE_MOLECULES on barcode_idx (crgpu.h): rewrites the device column in place; returns (retained old indices,
        rewritten pass_filter indices)"""
        pf = np.zeros(0, np.uint64) if pass_filter_idx is None else np.ascontiguousarray(pass_filter_idx, dtype=np.uint64).copy()
        retained = np.zeros(n_barcodes, np.uint64)
        n = C.c_uint64()
        self._check(self.L.crgpu_trim_molecule_barcodes_dev(self.h, _p(d_barcode_idx), n_molecules, n_barcodes, ptr(pf) if len(pf) else None,
                                                            len(pf), int(pass_only), offset, ptr(retained), C.byref(n)))
        return retained[: n.value], pf

    def concat_matrices(self, mats, gem_groups):
        """merged matrix of several GEM wells: column concatenation in (gem_group, barcode) order"""
        arr = (C.c_void_p * len(mats))(*[C.cast(m._mv, C.c_void_p) for m in mats])
        gg = np.ascontiguousarray(gem_groups, dtype=np.uint16)
        mv = C.POINTER(MatrixView)()
        self._check(self.L.crgpu_concat_matrices(self.h, arr, ptr(gg), len(mats), C.byref(mv)))
        return Matrix(self, mv)

    def assemble_matrix(self, bc, feature, count, n_features):
        bc = np.ascontiguousarray(bc, np.uint32)
        ft = np.ascontiguousarray(feature, np.uint32)
        ct = np.ascontiguousarray(count, np.uint32)
        mv = C.POINTER(MatrixView)()
        self._check(self.L.crgpu_assemble_matrix(self.h, ptr(bc), ptr(ft), ptr(ct), len(bc), n_features, C.byref(mv)))
        return Matrix(self, mv)

    def assemble_matrix_dev(self, d_bc, d_feature, d_count, n_triplets):
        mv = C.POINTER(_lib.MatrixDevView)()
        self._check(self.L.crgpu_assemble_matrix_dev(self.h, _p(d_bc), _p(d_feature), _p(d_count), n_triplets, C.byref(mv)))
        return MatrixDev(self, mv)

    def count(self, recs, n_features):
        mv = C.POINTER(MatrixView)()
        self._check(self.L.crgpu_count(self.h, C.byref(recs), n_features, C.byref(mv)))
        return Matrix(self, mv)

    # ---- feature barcodes ----------------------------------------------------------------------------
    def set_feature_pattern(self, pattern, feat_seqs, feat_index, feat_dist=None):
        s = ascii_matrix(feat_seqs)
        ix = np.ascontiguousarray(feat_index, dtype=np.uint32)
        d = None if feat_dist is None else np.ascontiguousarray(feat_dist, dtype=np.float64)
        self._check(self.L.crgpu_set_feature_pattern(self.h, pattern, s.tobytes(), s.shape[0], s.shape[1], ptr(ix), ptr(d)))

    def match_features(self, pattern, d_seq, d_qualn, n, d_feature_out):
        self._check(self.L.crgpu_match_features_dev(self.h, pattern, _p(d_seq), _p(d_qualn), n, _p(d_feature_out)))

    # ---- segmented barcode constructs (GelBeadAndProbe) ---------------------------------------------------
    def set_barcode_segments(self, lib, seg_seqs, seg_lens):
        """seg_seqs: per segment the packed sequences, ascending (a segment context's canonical list)"""
        k = len(seg_seqs)
        keep = [np.ascontiguousarray(a, dtype=np.uint32) for a in seg_seqs]
        n = (C.c_uint32 * k)(*[len(a) for a in keep])
        ln = (C.c_uint32 * k)(*seg_lens)
        pp = (C.c_void_p * k)(*[a.ctypes.data for a in keep])
        self._check(self.L.crgpu_set_barcode_segments(self.h, lib, k, n, ln, pp))
        self.cb_len, self.n_canon = int(sum(seg_lens)), int(np.prod([len(a) for a in keep], dtype=np.int64))

    def combine_segments(self, lib, d_seg_idx, n, d_idx_inout, after_correction=False):
        pp = (C.c_void_p * len(d_seg_idx))(*[_p(a) for a in d_seg_idx])
        self._check(self.L.crgpu_combine_segments_dev(self.h, lib, pp, len(d_seg_idx), n, int(after_correction), _p(d_idx_inout)))

    # whole reads, every pattern form (FeatureExtractor::match_read)
    def set_feature_extractor(self, extractor, defs, feat_dist=None):
        """defs: [(pattern, sequence, FeatureDef::index, read)] of ONE feature type, read 0 = R1, 1 = R2"""
        keep = [(p.encode(), s.encode()) for p, s, _, _ in defs]
        arr = (_lib.FeatureDef * len(defs))()
        for k, (_, _, index, read) in enumerate(defs):
            arr[k] = _lib.FeatureDef(keep[k][0], keep[k][1], index, read)
        d = None if feat_dist is None else np.ascontiguousarray(feat_dist, dtype=np.float64)
        self._check(self.L.crgpu_set_feature_extractor(self.h, extractor, arr, len(defs), ptr(d), 0 if d is None else len(d)))

    def feature_extractor_regexes(self, extractor):
        n = C.c_uint32(0)
        self._check(self.L.crgpu_feature_extractor_regex(self.h, extractor, 0, None, 0, C.byref(n)))
        out = []
        for p in range(n.value):
            buf = C.create_string_buffer(1 << 20)
            self._check(self.L.crgpu_feature_extractor_regex(self.h, extractor, p, buf, 1 << 20, None))
            out.append(buf.value.decode())
        return out

    def extract_features(self, extractor, n, d_feature_out, r1=None, r2=None, d_n_ids_out=None, d_capture_out=None):
        """r1 / r2: (d_seq_rows, d_qual_rows, d_len or None, stride) or None"""
        a = r1 or (None, None, None, 0)
        b = r2 or (None, None, None, 0)
        self._check(self.L.crgpu_extract_features_dev(self.h, extractor, _p(a[0]), _p(a[1]), _p(a[2]), a[3], _p(b[0]), _p(b[1]),
                                                      _p(b[2]), b[3], n, _p(d_feature_out), _p(d_n_ids_out), _p(d_capture_out)))

    def feature_counts(self, d_feature, n, n_features, counts=None):
        """MAKE_SHARD's feature_counts (make_shard_metrics.rs:336-345): counts[f] += reads whose feature is f"""
        counts = np.zeros(n_features, np.int64) if counts is None else counts
        self._check(self.L.crgpu_feature_counts_dev(self.h, _p(d_feature), n, n_features, ptr(counts)))
        return counts

    def synth_rows(self, seed, first, n, d_feature, feat_seq, L, offset, row_stride, d_seq_rows, d_qual_rows, err=0.005, n_rate=0.0005):
        """Feature Barcoding read rows on the device (crgpu_synth_rows_dev); feat_seq: packed u64 sequences"""
        fs = np.ascontiguousarray(feat_seq, dtype=np.uint64)
        self._check(self.L.crgpu_synth_rows_dev(self.h, seed, first, n, _p(d_feature), ptr(fs), len(fs), L, offset, row_stride,
                                                int(round(err * 65536)), int(round(n_rate * (1 << 20))), _p(d_seq_rows), _p(d_qual_rows)))

    # ---- synthetic data --------------------------------------------------------------------------------
    def synth(self, params, first, n, cb=None, cb_qualn=None, umi=None, umi_qualn=None, feature=None, flags=None):
        o = SynthOut()
        o.cb, o.cb_qualn, o.umi, o.umi_qualn = _p(cb), _p(cb_qualn), _p(umi), _p(umi_qualn)
        o.feature, o.flags = _p(feature), _p(flags)
        self._check(self.L.crgpu_synth_dev(self.h, C.byref(params.c), first, n, C.byref(o)))
