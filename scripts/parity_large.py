"""One-off: the cfg3 model at 12 M .. 100 M reads (thousands of chunk tickets in every pass of the sort)
compared with the oracle read by read -- the checker of tests/test_gpu_count.py at 12x .. 100x its usual size.
usage (GPU box): python3 scripts/parity_large.py [n_reads] [n_libs] [multiplexing_lib_mask] [whitelist entries]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_helpers as G  # noqa: E402
import test_gpu_count as T  # noqa: E402
from cellranger_amd import synth as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000
n_libs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mux = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
n_wl = int(sys.argv[4]) if len(sys.argv) > 4 else 737280
w = S.Workload(n_total=n, seed=S.SEED0 + 3, n_libs=n_libs, n_wl=n_wl)
c = G.fresh_ctx()
for lib in range(n_libs):
    c.set_whitelist(lib, w.wl_packed, length=16)
r = w.host_reads(0, n)
t0 = time.time()
res, m = T._compare_with_oracle(c, w, r, n, w.n_genes, n_libs=n_libs, mux_mask=mux)
print("parity ok: libs=%d mux=%d n=%d whitelist=%d columns=%d nnz=%d molecules=%d (%.0f s)" % (n_libs, mux, n, n_wl, m.n_barcodes, m.nnz, len(res.mol), time.time() - t0))
