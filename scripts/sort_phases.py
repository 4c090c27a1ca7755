"""Where a chunk of the onesweep scatter spends its time: phase attribution from a -DSORT_PHASE_TIMING build.
usage (GPU box): CRGPU_LIB_PATH=cellranger_amd/variants/libcrgpu_phases.so python3 scripts/sort_phases.py [n_reads]
Thread 0 of every workgroup adds the shader-clock ticks between the phase marks of k_radix_scatter to sixteen device
counters; printed as the share of each phase over all chunks of all passes of one molecule-key sort."""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from cellranger_amd import _lib  # noqa: E402
from cellranger_amd import engine as E  # noqa: E402
from cellranger_amd import synth as S  # noqa: E402

NAMES = ["ticket", "clear counters", "keys arrive", "rank own wave", "wait other waves", "digit scan", "LDS scatter",
         "look-back (thread 0)", "look-back (all digits)", "copy-out issue", "copy-out barrier"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
    w = S.Workload(n_total=n, seed=S.SEED0 + 3)
    c = E.Context(0)
    c.set_whitelist(0, w.wl_packed, length=16)
    c.set_key_layout(w.n_genes, w.umi_len, 1, 0)
    d = dict(cb=c.empty(n, np.uint32), cbq=c.empty((n, 16), np.uint8), fl=c.empty(n, np.uint8), umi=c.empty(n, np.uint32),
             uq=c.empty((n, 12), np.uint8), ft=c.empty(n, np.uint32), idx=c.empty(n, np.uint32))
    c.synth(w, 0, n, cb=d["cb"].ptr, cb_qualn=d["cbq"].ptr, umi=d["umi"].ptr, umi_qualn=d["uq"].ptr, feature=d["ft"].ptr,
            flags=d["fl"].ptr)
    c.match_and_count(d["cb"], d["fl"], n, d["idx"])
    c.correct(d["cb"], d["cbq"], d["fl"], n, d["idx"])
    recs = c.records(n, w.umi_len, d["idx"], d["umi"], d["uq"], d["ft"], d["fl"])
    keys = c.empty(n, np.uint64)
    lib = _lib.load()
    out = (ctypes.c_ulonglong * 16)()
    for rep in range(3):
        nk = c.build_keys(recs, keys)
        c.count_keys(keys, nk).free()
        c.synchronize()
        assert lib.crgpu_debug_sort_phases(out) == 0
    v = np.array(list(out), dtype=np.float64)
    chunks = max(v[13], 1.0)
    print("look-back of digit 0, per chunk: %.2f polls of unpublished statuses, %.2f predecessors walked (%d chunks)"
          % (v[11] / chunks, v[12] / chunks, int(v[13])))
    v = v[:11]
    tot = v.sum()
    print("n_keys %d, ticks per chunk and workgroup summed over all passes: %.3g" % (nk, tot))
    for i, name in enumerate(NAMES):
        print("%-26s %6.2f %%" % (name, 100.0 * v[i] / tot))


main()
