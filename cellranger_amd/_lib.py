"""ctypes binding of libcrgpu.so (include/crgpu.h).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is visible, loading or
`Context()` raises.  Nothing here imports the oracle.
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
# CRGPU_LIB_PATH: load another build of the same library (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("CRGPU_LIB_PATH") or os.path.join(PKG, "libcrgpu.so")

MISS = 0xFFFFFFFF
NO_FEATURE = 0xFFFFFFFF
MAX_LIB = 16
FLAG_LIB_MASK = 0x0F
FLAG_CB_HAS_N = 0x10
FLAG_NONTXOMIC = 0x20

COUNTS_VALID, COUNTS_CORRECTED, COUNTS_PRIOR = 0, 1, 2
T_NAMES = ["pack", "match", "correct", "keys", "sort_scatter", "dedup", "matrix", "synth", "sort_hist", "scan", "comm", "feature"]
UNIQUE_ID_BYTES = 128
OPT_BUFFERS_UNCHANGED_BETWEEN_CALLS = 0


class CrgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("crgpu error %d: %s" % (code, msg))
        self.code = code


class Records(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("umi_len", C.c_uint32),
        ("d_bc_idx", C.c_void_p),
        ("d_umi", C.c_void_p),
        ("d_umi_qualn", C.c_void_p),
        ("d_feature", C.c_void_p),
        ("d_flags", C.c_void_p),
        ("d_umi_len", C.c_void_p),
        ("d_probe_idx", C.c_void_p),
    ]


class MatrixView(C.Structure):
    _fields_ = [
        ("n_barcodes", C.c_uint64),
        ("nnz", C.c_uint64),
        ("n_features", C.c_uint32),
        ("cb_len", C.c_uint32),
        ("barcode_rank", C.c_void_p),
        ("barcode_seq", C.c_void_p),
        ("indptr", C.c_void_p),
        ("indices", C.c_void_p),
        ("data", C.c_void_p),
        ("gem_group", C.c_void_p),
        ("barcode_seq_hi", C.c_void_p),
    ]


class MatrixDevView(C.Structure):
    _fields_ = [
        ("n_barcodes", C.c_uint64),
        ("nnz", C.c_uint64),
        ("d_barcode_rank", C.c_void_p),
        ("d_indptr", C.c_void_p),
        ("d_indices", C.c_void_p),
        ("d_data", C.c_void_p),
    ]


SHARD_METRIC_FIELDS = ["sequenced_reads", "bc_n_bases", "bc_bases", "umi_n_bases", "umi_bases", "bc_q30_bases", "bc_q30_den",
                       "umi_q30_bases", "umi_q30_den", "good_umi", "has_n_barcode", "has_n_umi", "homopolymer_barcode",
                       "homopolymer_umi", "low_min_qual_barcode", "low_min_qual_umi", "miss_whitelist_barcode", "polyt_suffix_umi"]


class ShardMetrics(C.Structure):
    _fields_ = [(f, C.c_uint64) for f in SHARD_METRIC_FIELDS]


class RowsMetrics(C.Structure):
    _fields_ = [(f, C.c_uint64) for f in ("n_bases", "bases", "q30_bases", "q30_den")]


# crgpu_dupinfo (crgpu_count_host's per-read output) as a numpy record
DUPINFO_DTYPE = np.dtype([("processed_umi", np.uint32), ("read_count", np.uint32), ("flags", np.uint8), ("reserved", np.uint8, (3,))])
NO_PROBE = -1
ABI_VERSION = 3

# crgpu_barcode_summary_row as a numpy record
BARCODE_SUMMARY_DTYPE = np.dtype([("barcode_rank", np.uint32), ("library", np.uint32), ("reads", np.uint64), ("umis", np.uint64),
                                  ("candidate_dup_reads", np.uint64), ("umi_corrected_reads", np.uint64)])


class SynthParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("cb_len", C.c_uint32),
        ("umi_len", C.c_uint32),
        ("n_wl", C.c_uint32),
        ("wl_packed", C.c_void_p),
        ("n_cells", C.c_uint32),
        ("cell_wl_pos", C.c_void_p),
        ("cell_cdf", C.c_void_p),
        ("n_ambient", C.c_uint32),
        ("ambient_wl_pos", C.c_void_p),
        ("n_genes", C.c_uint32),
        ("gene_cdf", C.c_void_p),
        ("ambient_per_2_16", C.c_uint32),
        ("cb_err_per_2_16", C.c_uint32),
        ("umi_err_per_2_16", C.c_uint32),
        ("n_per_2_20", C.c_uint32),
        ("no_feature_per_2_16", C.c_uint32),
        ("reads_per_umi", C.c_uint32),
        ("n_total", C.c_uint64),
        ("n_libs", C.c_uint32),
    ]


class SynthOut(C.Structure):
    _fields_ = [
        ("cb", C.c_void_p),
        ("cb_qualn", C.c_void_p),
        ("umi", C.c_void_p),
        ("umi_qualn", C.c_void_p),
        ("feature", C.c_void_p),
        ("flags", C.c_void_p),
    ]


class BcCorrectionMetrics(C.Structure):
    _fields_ = [("valid_reads", C.c_uint64), ("corrected_reads", C.c_uint64), ("barcodes_detected", C.c_uint64),
                ("effective_barcode_diversity", C.c_double)]


# every symbol include/crgpu.h declares: (restype, argtypes)
_vp, _u8p, _u32, _u64, _i, _dbl = C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
SYMBOLS = {
    "crgpu_abi_version": (_i, []),
    "crgpu_abi_layout": (_i, [C.c_char_p, _vp, _u32]),
    "crgpu_count_host": (_i, [_vp, C.POINTER(Records), _u32, C.POINTER(C.POINTER(MatrixView)), _vp, C.POINTER(_vp)]),
    "crgpu_counts_probe_idx": (_i, [_vp, _vp, _vp]),
    "crgpu_get_unique_id": (_i, [_vp]),
    "crgpu_local_group_id": (_i, [_u32, _vp]),
    "crgpu_create": (_i, [C.POINTER(_vp), _i, _i, _i, _vp]),
    "crgpu_comm_info": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "crgpu_set_option": (_i, [_vp, _i, C.c_int64]),
    "crgpu_invalidate": (_i, [_vp]),
    "crgpu_get_stat": (_i, [_vp, _i, C.POINTER(_u64)]),
    "crgpu_barrier": (_i, [_vp]),
    "crgpu_allreduce_counts": (_i, [_vp, _i, _i]),
    "crgpu_exchange_keys_dev": (_i, [_vp, _vp, _u64, C.POINTER(_vp), C.POINTER(_u64), _vp]),
    "crgpu_gatherv_dev": (_i, [_vp, _vp, _u64, _i, C.POINTER(_vp), _vp]),
    "crgpu_gather_triplets_dev": (_i, [_vp, _vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_u64)]),
    "crgpu_allreduce_max_f64": (_i, [_vp, C.POINTER(_dbl)]),
    "crgpu_allreduce_sum_i64": (_i, [_vp, _vp, _u32]),
    "crgpu_destroy": (None, [_vp]),
    "crgpu_last_error": (C.c_char_p, [_vp]),
    "crgpu_synchronize": (_i, [_vp]),
    "crgpu_stream": (_vp, [_vp]),
    "crgpu_malloc": (_i, [_vp, C.POINTER(_vp), _u64]),
    "crgpu_free": (_i, [_vp, _vp]),
    "crgpu_trim": (_i, [_vp]),
    "crgpu_memcpy_h2d": (_i, [_vp, _vp, _vp, _u64]),
    "crgpu_memcpy_d2h": (_i, [_vp, _vp, _vp, _u64]),
    "crgpu_memset": (_i, [_vp, _vp, _i, _u64]),
    "crgpu_timing_enable": (_i, [_vp, _i]),
    "crgpu_timing_reset": (_i, [_vp]),
    "crgpu_timing_get": (_i, [_vp, _vp, _vp, _vp]),
    "crgpu_set_whitelist": (_i, [_vp, _i, C.c_char_p, _u32, _u32, C.c_char_p, _u32, _vp]),
    "crgpu_set_whitelist_packed": (_i, [_vp, _i, _vp, _u32, _u32, _vp, _u32, _vp]),
    "crgpu_whitelist_info": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "crgpu_get_canon_order": (_i, [_vp, _vp, _vp]),
    "crgpu_pack_dev": (_i, [_vp, _vp, _vp, _u64, _u32, _vp, _vp, _vp]),
    "crgpu_pack_rows_dev": (_i, [_vp, _vp, _vp, _u64, _u32, _u32, _u32, _vp, _vp, _vp]),
    "crgpu_shard_metrics_dev": (_i, [_vp, _vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, C.POINTER(ShardMetrics)]),
    "crgpu_rows_metrics_dev": (_i, [_vp, _vp, _vp, _vp, _u64, _u32, C.POINTER(RowsMetrics)]),
    "crgpu_homopolymer_metrics_dev": (_i, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _u32, _vp]),
    "crgpu_fastq_to_rows_dev": (_i, [_vp, _vp, _u64, _u32, _u64, _vp, _vp, _vp, C.POINTER(_u64)]),
    "crgpu_match_and_count_dev": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "crgpu_set_posterior": (_i, [_vp, _dbl, _dbl]),
    "crgpu_correct_dev": (_i, [_vp, _vp, _vp, _vp, _u64, _vp, _vp]),
    "crgpu_get_counts": (_i, [_vp, _i, _i, _vp]),
    "crgpu_set_counts": (_i, [_vp, _i, _i, _vp]),
    "crgpu_reset_counts": (_i, [_vp]),
    "crgpu_counts_dev": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "crgpu_barcode_correction_metrics": (_i, [_vp, _i, C.POINTER(BcCorrectionMetrics)]),
    "crgpu_total_barcode_counts": (_i, [_vp, C.c_int64, _vp, _vp, _u64, C.POINTER(_u64)]),
    "crgpu_match_and_count": (_i, [_vp, _i, _vp, _vp, _u64, _vp]),
    "crgpu_correct": (_i, [_vp, _i, _vp, _vp, _u64, _vp, _vp]),
    "crgpu_set_key_layout": (_i, [_vp, _u32, _u32, _u32, _u32]),
    "crgpu_set_umi_min_len": (_i, [_vp, _u32]),
    "crgpu_pack_rows_var_dev": (_i, [_vp, _vp, _vp, _vp, _u64, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "crgpu_set_target_filter": (_i, [_vp, _vp, _u32, _u64]),
    "crgpu_build_keys_dev": (_i, [_vp, C.POINTER(Records), _vp, C.POINTER(_u64)]),
    "crgpu_partition_keys_dev": (_i, [_vp, _vp, _u64, _u32, _vp, _vp, _vp]),
    "crgpu_balanced_bounds": (_i, [_vp, _u32, _vp]),
    "crgpu_count_keys_dev": (_i, [_vp, _vp, _u64, C.POINTER(_vp)]),
    "crgpu_count_records_dev": (_i, [_vp, C.POINTER(Records), C.POINTER(_vp), _vp, _vp, _vp]),
    "crgpu_count_records_sharded_dev": (_i, [_vp, C.POINTER(Records), C.POINTER(_vp), _vp, _vp, _vp]),
    "crgpu_counts_info": (_i, [_vp, _vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "crgpu_counts_triplets_dev": (_i, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "crgpu_counts_triplets": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "crgpu_counts_molecules": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "crgpu_counts_molecule_info": (_i, [_vp, _vp, C.c_uint16, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "crgpu_enable_barcode_summary": (_i, [_vp, _i]),
    "crgpu_counts_barcode_summary": (_i, [_vp, _vp, _u32, _u32, _vp, _u64, C.POINTER(_u64)]),
    "crgpu_write_barcode_summary_csv": (_i, [_vp, _vp, _u64, C.c_uint16, _vp, C.POINTER(C.c_char_p), _u32, C.c_char_p]),
    "crgpu_counts_free": (None, [_vp, _vp]),
    "crgpu_assemble_matrix": (_i, [_vp, _vp, _vp, _vp, _u64, _u32, C.POINTER(C.POINTER(MatrixView))]),
    "crgpu_matrix_free": (None, [_vp, C.POINTER(MatrixView)]),
    "crgpu_sum_matrices": (_i, [_vp, C.POINTER(MatrixView), C.POINTER(MatrixView), C.POINTER(C.POINTER(MatrixView))]),
    "crgpu_select_barcodes": (_i, [_vp, C.POINTER(MatrixView), _vp, _u64, C.POINTER(C.POINTER(MatrixView))]),
    "crgpu_trim_molecule_barcodes_dev": (_i, [_vp, _vp, _u64, _u64, _vp, _u64, _i, _u64, _vp, C.POINTER(_u64)]),
    "crgpu_concat_matrices": (_i, [_vp, _vp, _vp, _u32, C.POINTER(C.POINTER(MatrixView))]),
    "crgpu_write_mtx": (_i, [_vp, C.POINTER(MatrixView), C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint16]),
    "crgpu_assemble_matrix_dev": (_i, [_vp, _vp, _vp, _vp, _u64, C.POINTER(C.POINTER(MatrixDevView))]),
    "crgpu_sum_matrices_dev": (_i, [_vp, C.POINTER(MatrixDevView), C.POINTER(MatrixDevView), C.POINTER(C.POINTER(MatrixDevView))]),
    "crgpu_select_barcodes_dev": (_i, [_vp, C.POINTER(MatrixDevView), _vp, _u64, C.POINTER(C.POINTER(MatrixDevView))]),
    "crgpu_matrix_dev_free": (None, [_vp, C.POINTER(MatrixDevView)]),
    "crgpu_matrix_dev_download": (_i, [_vp, C.POINTER(MatrixDevView), _vp, _vp, _vp, _vp]),
    "crgpu_count": (_i, [_vp, C.POINTER(Records), _u32, C.POINTER(C.POINTER(MatrixView))]),
    "crgpu_set_feature_pattern": (_i, [_vp, _i, C.c_char_p, _u32, _u32, _vp, _vp]),
    "crgpu_match_features_dev": (_i, [_vp, _i, _vp, _vp, _u64, _vp]),
    "crgpu_set_barcode_segments": (_i, [_vp, _i, _u32, _vp, _vp, _vp]),
    "crgpu_combine_segments_dev": (_i, [_vp, _i, _vp, _u32, _u64, _i, _vp]),
    "crgpu_set_feature_extractor": (_i, [_vp, _i, _vp, _u32, _vp, _u32]),
    "crgpu_compile_feature_pattern": (_i, [C.c_char_p, _u32, C.c_char_p, _u64]),
    "crgpu_feature_extractor_regex": (_i, [_vp, _i, _u32, C.c_char_p, _u64, _vp]),
    "crgpu_extract_features_dev": (_i, [_vp, _i, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _u32, _u64, _vp, _vp, _vp]),
    "crgpu_feature_counts_dev": (_i, [_vp, _vp, _u64, _u32, _vp]),
    "crgpu_compute_feature_dist": (_i, [_vp, _vp, _u32, _vp]),
    "crgpu_synth_rows_dev": (_i, [_vp, _u64, _u64, _u64, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp]),
    "crgpu_synth_rows_host": (_i, [_u64, _u64, _u64, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp]),
    "crgpu_synth_dev": (_i, [_vp, C.POINTER(SynthParams), _u64, _u64, C.POINTER(SynthOut)]),
    "crgpu_synth_host": (_i, [C.POINTER(SynthParams), _u64, _u64, C.POINTER(SynthOut)]),
}

_lib = None


def load():
    """Load libcrgpu.so and bind every symbol of include/crgpu.h (raises if one is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CrgpuError(-2, "%s not found: build it with `python -m cellranger_amd.build` "
                             "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    missing = [name for name in SYMBOLS if not hasattr(L, name)]
    if missing:
        raise CrgpuError(-2, "libcrgpu.so does not export: %s" % ", ".join(missing))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def ptr(a):
    """numpy array / int / None -> c_void_p"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))


class FeatureDef(C.Structure):
    """crgpu_feature_def"""
    _fields_ = [("pattern", C.c_char_p), ("sequence", C.c_char_p), ("index", C.c_uint32), ("read", C.c_uint32)]


NO_CAPTURE = 0xFFFFFFFF
