"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol
include/crgpu.h declares, the host helpers behave, the synthetic generator is deterministic, and
the oracle is self-consistent (serial == chunk-parallel)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session", autouse=True)
def built_library():
    from cellranger_amd import build

    build.build()


def header_symbols():
    with open(os.path.join(ROOT, "include", "crgpu.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cellranger_amd import _lib

    declared = header_symbols()
    assert len(declared) > 30
    L = C.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    # and the Python binding table covers exactly the header
    assert sorted(_lib.SYMBOLS) == declared
    assert _lib.load().crgpu_abi_version() == _lib.ABI_VERSION == 3


# ---- the layout of every public struct: header == library == ctypes table == the Rust blocks of INTEGRATION.md -------------
_C_SIZES = {"uint64_t": 8, "int64_t": 8, "uint32_t": 4, "int32_t": 4, "uint16_t": 2, "uint8_t": 1, "double": 8, "int": 4}
_RUST_SIZES = {"u64": 8, "i64": 8, "u32": 4, "i32": 4, "u16": 2, "u8": 1, "f64": 8, "c_int": 4}


def _layout(fields):
    """C / repr(C) layout of [(name, size, align, count)] -> (sizeof, alignof, [(name, offset, size)])"""
    off, amax, out = 0, 1, []
    for name, size, align, count in fields:
        off = (off + align - 1) // align * align
        out.append((name, off, size * count))
        off += size * count
        amax = max(amax, align)
    return (off + amax - 1) // amax * amax, amax, out


def header_structs():
    with open(os.path.join(ROOT, "include", "crgpu.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\w*\s*\{(.*?)\}\s*(crgpu_\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"(?:const\s+)?(\w+)\s*((?:\*\s*(?:const\s*)?)*)(.*)$", decl)
            ctype, stars, names = mm.group(1), mm.group(2), mm.group(3)
            for nm in names.split(","):
                nm = nm.strip()
                ptr = "*" in stars or nm.startswith("*")
                nm = nm.lstrip("* ")
                arr = re.match(r"(\w+)\[(\d+)\]$", nm)
                count = int(arr.group(2)) if arr else 1
                nm = arr.group(1) if arr else nm
                size = 8 if ptr else _C_SIZES[ctype]
                fields.append((nm, size, size, count))
        structs[m.group(2)] = _layout(fields)
    return structs


def rust_structs():
    """every `#[repr(C)] pub struct Name { ... }  // crgpu_name` block of INTEGRATION.md"""
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        text = f.read()
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{\s*//\s*(crgpu_\w+)[^\n]*\n(.*?)\n\}", text, flags=re.S):
        body = re.sub(r"//[^\n]*", "", m.group(3))
        fields = []
        for nm, ty in re.findall(r"pub\s+(\w+)\s*:\s*([^,]+?)\s*(?:,|$)", body.replace("\n", " ")):
            ty = ty.strip()
            arr = re.match(r"\[(\w+);\s*(\d+)\]$", ty)
            if ty.startswith("*const") or ty.startswith("*mut"):
                fields.append((nm, 8, 8, 1))
            elif arr:
                fields.append((nm, _RUST_SIZES[arr.group(1)], _RUST_SIZES[arr.group(1)], int(arr.group(2))))
            else:
                fields.append((nm, _RUST_SIZES[ty], _RUST_SIZES[ty], 1))
        structs[m.group(2)] = (m.group(1), _layout(fields))
    return structs


def library_layout(name):
    from cellranger_amd import _lib
    L = _lib.load()
    w = np.zeros(256, np.uint32)
    n = L.crgpu_abi_layout(name.encode(), _lib.ptr(w), len(w))
    assert n >= 3, (name, n)
    nf = int(w[2])
    assert n == 3 + 2 * nf
    return int(w[0]), int(w[1]), [(int(w[3 + 2 * i]), int(w[4 + 2 * i])) for i in range(nf)]


def test_public_struct_layouts_agree_everywhere():
    """INTEGRATION.md's CrgpuRecords / CrgpuMatrix were 8 bytes short of the header.  Four views of every struct
    that crosses the ABI must agree on field count, order, offset and width: the header (parsed), the library
    (crgpu_abi_layout, what a binding checks at start-up), the ctypes table and the Rust #[repr(C)] blocks of INTEGRATION.md."""
    from cellranger_amd import _lib
    hdr = header_structs()
    assert {"crgpu_records", "crgpu_matrix", "crgpu_matrix_dev", "crgpu_dupinfo", "crgpu_barcode_summary_row",
            "crgpu_shard_metrics", "crgpu_rows_metrics", "crgpu_bc_correction_metrics", "crgpu_feature_def",
            "crgpu_synth_params", "crgpu_synth_out"} == set(hdr)
    for name, (size, align, fields) in hdr.items():
        lsize, lalign, lfields = library_layout(name)
        assert (size, align) == (lsize, lalign), name
        assert [(o, s) for _, o, s in fields] == lfields, name
    assert _lib.load().crgpu_abi_layout(b"crgpu_nonsense", None, 0) < 0
    # ctypes table
    ct = {"crgpu_records": _lib.Records, "crgpu_matrix": _lib.MatrixView, "crgpu_matrix_dev": _lib.MatrixDevView,
          "crgpu_shard_metrics": _lib.ShardMetrics, "crgpu_rows_metrics": _lib.RowsMetrics,
          "crgpu_bc_correction_metrics": _lib.BcCorrectionMetrics, "crgpu_feature_def": _lib.FeatureDef,
          "crgpu_synth_params": _lib.SynthParams, "crgpu_synth_out": _lib.SynthOut}
    for name, cls in ct.items():
        size, _, fields = hdr[name]
        assert C.sizeof(cls) == size, name
        assert [(f[0], getattr(cls, f[0]).offset, getattr(cls, f[0]).size) for f in cls._fields_] == fields, name
    for name, dt in (("crgpu_dupinfo", _lib.DUPINFO_DTYPE), ("crgpu_barcode_summary_row", _lib.BARCODE_SUMMARY_DTYPE)):
        size, _, fields = hdr[name]
        assert dt.itemsize == size, name
        assert [(n, dt.fields[n][1], dt.fields[n][0].itemsize) for n in dt.names] == fields, name
    # the Rust mirrors a maintainer would paste
    rust = rust_structs()
    assert {"crgpu_records", "crgpu_matrix", "crgpu_matrix_dev", "crgpu_dupinfo", "crgpu_barcode_summary_row",
            "crgpu_bc_correction_metrics", "crgpu_feature_def"} <= set(rust)
    for name, (rname, (size, align, fields)) in rust.items():
        assert name in hdr, (rname, name)
        assert (size, align, fields) == hdr[name], "INTEGRATION.md %s does not match %s of include/crgpu.h" % (rname, name)


def test_stale_rust_struct_is_caught():
    """the checker itself: the round-2 CrgpuRecords (7 fields, no d_umi_len) must not pass"""
    stale = _layout([("n", 8, 8, 1), ("umi_len", 4, 4, 1)] + [(f, 8, 8, 1) for f in
                    ("d_bc_idx", "d_umi", "d_umi_qualn", "d_feature", "d_flags")])
    assert stale != header_structs()["crgpu_records"]


def test_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    from cellranger_amd import engine as E
    from cellranger_amd._lib import CrgpuError

    with pytest.raises(CrgpuError) as ei:
        E.Context(0)
    assert "no CPU fallback" in str(ei.value)


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "cellranger_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert "oracle_lib" not in src and "liboracle" not in src and "cr_oracle.h" not in src, fn


def test_pack_unpack_roundtrip():
    from cellranger_amd import engine as E

    seqs = ["ACGTACGTACGTACGT", "TTTTTTTTTTTTTTTT", "AAAAAAAAAAAAAAAA", "GATTACAGATTACAGA"]
    pk, L = E.pack_seqs(seqs)
    assert L == 16 and pk[1] == 0xFFFFFFFF and pk[2] == 0
    assert [bytes(r).decode() for r in E.unpack_seqs(pk, 16)] == seqs
    # numeric order == byte-lexicographic order (barcode/src/lib.rs:119-124)
    assert [seqs[i] for i in np.argsort(pk)] == sorted(seqs)
    with pytest.raises(ValueError):
        E.pack_seqs(["ACGN"])


def test_synth_host_is_deterministic_and_has_the_cfg2_shape():
    from cellranger_amd import synth as S
    from cellranger_amd._lib import FLAG_CB_HAS_N, NO_FEATURE

    w = S.Workload(n_total=200_000, seed=S.SEED0 + 2, n_wl=50_000, n_cells=500, n_ambient=5000)
    a = w.host_reads(1000, 50_000)
    b = w.host_reads(1000, 50_000)
    for k in a:
        assert np.array_equal(a[k], b[k])
    # a window of a longer stream equals the same reads generated alone
    c = w.host_reads(1000 + 77, 1000)
    assert np.array_equal(c["cb"], a["cb"][77:1077]) and np.array_equal(c["umi_qualn"], a["umi_qualn"][77:1077])
    on_wl = np.isin(a["cb"], w.wl_packed) & ((a["flags"] & FLAG_CB_HAS_N) == 0)
    assert 0.88 < on_wl.mean() < 0.96          # ~7.7 % of reads miss the whitelist
    assert 0.05 < (a["feature"] == NO_FEATURE).mean() < 0.11
    n_flag = (a["flags"] & FLAG_CB_HAS_N) != 0
    assert np.array_equal(n_flag, ((a["cb_qualn"] & 0x80) != 0).any(axis=1))
    q = a["cb_qualn"] & 0x7F
    assert set(np.unique(q)) <= {35, 44, 58, 70}


def test_oracle_parallel_equals_serial():
    import oracle_lib as O
    from cellranger_amd import engine as E
    from cellranger_amd import synth as S

    w = S.Workload(n_total=60_000, seed=5, n_wl=5000, n_cells=60, n_ambient=800, n_genes=50, n_libs=2)
    r = w.host_reads(0, 60_000)
    cb, cbq = S.to_ascii(r["cb"], r["cb_qualn"], 16)
    umi, uq = S.to_ascii(r["umi"], r["umi_qualn"], 12)
    reads = dict(cb=cb, cb_qual=cbq, umi=umi, umi_qual=uq, feature=r["feature"], lib=r["flags"] & 0x0F)
    wl = O.Whitelist(E.unpack_seqs(w.wl_packed, 16))
    a = O.run_pipeline(reads, [wl, wl], n_lib=2, n_threads=1, want_dupinfo=True)
    b = O.run_pipeline(reads, [wl, wl], n_lib=2, n_threads=4, want_dupinfo=True)
    for f in ("corrected_cb", "bc_state", "barcodes", "indptr", "indices", "data", "mol", "dupinfo"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    # sanity of the restated semantics on this data set
    assert (a.bc_state == 2).sum() > 500 and a.dupinfo["is_corrected"].sum() > 100
    assert a.data.sum() == len(a.mol) and a.indptr[-1] == len(a.data)


def test_bench_launcher_starts_its_own_ranks_and_refuses_a_world_mismatch():
    """`bench.py --gpus N` without torchrun must start N rank processes itself and fail when a rank fails (here: no GPU),
    and must refuse a WORLD_SIZE that differs from --gpus instead of silently measuring one GPU."""
    import subprocess
    import sys

    import torch

    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0"), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and "--gpus 2 but WORLD_SIZE=4" in r.stderr
    if torch.cuda.device_count() > 0:
        return  # on a GPU box the launcher is exercised for real by tests/test_gpu_pipeline.py
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0", "--reads-per-gpu", "1000"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "ranks failed" in r.stderr  # the first rank to die ends the other one, too


def test_compile_feature_pattern_golden():
    """compile_pattern (feature_extraction.rs:307-343) is host code behind the C ABI: the reference's test_compile_pattern
    cases (:585-635) need no GPU"""
    import json
    from cellranger_amd import engine as E
    with open(os.path.join(os.path.dirname(__file__), "golden", "feature_vectors.json")) as f:
        g = json.load(f)
    for case in g["compile_pattern"]:
        assert E.compile_feature_pattern(case["pattern"], case["length"]) == case["regex"], case
