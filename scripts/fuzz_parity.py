"""One-off randomized parity hunt (not part of the suite): random layouts, sizes around tile boundaries, library mixes
and multiplexing masks, each compared with the oracle read by read through tests/test_gpu_count.py's checker.
usage (GPU box): python3 scripts/fuzz_parity.py [n_trials] [seed] [max_reads]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_helpers as G  # noqa: E402
import test_gpu_count as T  # noqa: E402
from cellranger_amd import synth as S  # noqa: E402
from cellranger_amd._lib import FLAG_NONTXOMIC  # noqa: E402


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 300_000
    edges = [1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 16383, 16384, 16385,
             32767, 32768, 32769, 65535, 65536, 65537, 131071, 131073]
    bad = 0
    for t in range(trials):
        umi_len = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]))
        n_genes = int(rng.choice([1, 2, 5, 33, 257, 4097, 36601]))
        n_wl = int(rng.choice([16, 100, 1000, 33_000, 737_280]))
        n_libs = int(rng.choice([1, 1, 2, 3, 4]))
        cb_len = 16 if n_wl > 256 else int(rng.choice([4, 8, 16]))
        bits = 1 + 2 * umi_len + int(np.ceil(np.log2(n_libs))) + int(np.ceil(np.log2(max(n_genes, 1)))) + int(np.ceil(np.log2(n_wl)))
        if bits > 64 or n_wl > 4 ** cb_len // 2:
            continue
        n = int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(1, max_n))
        n_cells = max(1, min(int(rng.integers(1, 400)), n_wl // 2))
        mux = int(rng.integers(0, 1 << n_libs)) if n_libs > 1 and rng.random() < 0.4 else 0
        cfg = dict(n=n, umi_len=umi_len, n_genes=n_genes, n_wl=n_wl, n_libs=n_libs, cb_len=cb_len, n_cells=n_cells, mux=mux, bits=bits)
        try:
            w = S.Workload(n_total=n, seed=5000 + t, n_wl=n_wl, n_cells=n_cells, n_ambient=min(int(rng.integers(0, 500)), n_wl - n_cells),
                           n_genes=n_genes, cb_len=cb_len, umi_len=umi_len, umi_err=float(rng.choice([0.0, 0.01, 0.08])),
                           cb_err=float(rng.choice([0.0, 0.01, 0.05])), n_rate=float(rng.choice([0.0, 0.002, 0.02])),
                           no_feature_frac=float(rng.choice([0.0, 0.1, 0.5])), reads_per_umi=int(rng.integers(1, 6)), n_libs=n_libs,
                           sigma=float(rng.choice([0.1, 1.0, 2.0])))
            c = G.fresh_ctx()
            for lib in range(n_libs):
                c.set_whitelist(lib, w.wl_packed, length=cb_len)
            r = w.host_reads(0, n)
            r["flags"] = (r["flags"] | np.where(rng.random(n) < 0.2, FLAG_NONTXOMIC, 0)).astype(np.uint8)
            T._compare_with_oracle(c, w, r, n, n_genes, n_libs=n_libs, mux_mask=mux)
            c.close()
            print("ok  ", cfg, flush=True)
        except Exception:
            bad += 1
            print("FAIL", cfg, flush=True)
            traceback.print_exc()
    print("trials done, failures:", bad)
    sys.exit(1 if bad else 0)


main()
