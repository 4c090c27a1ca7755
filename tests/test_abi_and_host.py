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
    assert _lib.load().crgpu_abi_version() == 2


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
