"""Shared helpers for the -m gpu parity tests: run the HIP path through the C ABI and the oracle on
the same inputs."""
import numpy as np

import oracle_lib as O
from cellranger_amd import engine as E
from cellranger_amd import synth as S
from cellranger_amd._lib import COUNTS_CORRECTED, COUNTS_VALID, FLAG_CB_HAS_N, MISS

_ctx = None


def ctx():
    """One context per test process (contexts are cheap but the GPU box allows few processes)."""
    global _ctx
    if _ctx is None:
        _ctx = fresh_ctx()
    return _ctx


def fresh_ctx(trust=True, dense=None):
    """trust: CRGPU_OPT_BUFFERS_UNCHANGED_BETWEEN_CALLS -- the tests write their buffers only through the context, so
    the by-product paths (K1's miss records for K2, the key histograms for the sort) are what most of them exercise;
    tests/test_gpu_barcode.py covers the default (off) and the invalidation rules.
    dense: CRGPU_OPT_DENSE_BARCODE_KEYS (keys carry BarcodeIndex columns instead of whitelist ranks); None = what the
    environment says (CRGPU_TEST_DENSE=1 runs the whole suite that way; tests that hand-craft keys pass dense=False)."""
    import os
    c = E.Context(0)
    if trust:
        c.trust_unchanged_buffers(True)
    if dense is None:
        dense = os.environ.get("CRGPU_TEST_DENSE") == "1"
    if dense:
        c.set_option(1, 1)
    return c


def oracle_reads_from_packed(r, cb_len, umi_len):
    cb, cbq = S.to_ascii(r["cb"], r["cb_qualn"], cb_len)
    d = dict(cb=cb, cb_qual=cbq)
    if "umi" in r and r["umi"] is not None:
        umi, uq = S.to_ascii(r["umi"], r["umi_qualn"], umi_len)
        d.update(umi=umi, umi_qual=uq, feature=r["feature"])
    d["lib"] = (r["flags"] & 0x0F).astype(np.uint8)
    d["utype"] = ((r["flags"] & 0x20) >> 5).astype(np.uint8)
    return d


def ranks_of(canon_sorted_packed, ascii_seqs):
    """ASCII barcodes (n, L) -> canonical ranks (they must be on the canonical list)."""
    pk, _ = E.pack_seqs(ascii_seqs)
    pos = np.searchsorted(canon_sorted_packed, pk)
    assert (canon_sorted_packed[np.minimum(pos, len(canon_sorted_packed) - 1)] == pk).all()
    return pos.astype(np.uint32)


def gpu_barcode_stage(c, r, n):
    """K1 + K2 on packed device inputs; returns idx (host), corrected flags, device buffers."""
    d_cb = c.upload(r["cb"])
    d_cbq = c.upload(r["cb_qualn"])
    d_flags = c.upload(r["flags"])
    d_idx = c.empty(n, np.uint32)
    d_corr = c.empty(n, np.uint8)
    c.match_and_count(d_cb, d_flags, n, d_idx)
    idx_a = d_idx.to_host()
    c.correct(d_cb, d_cbq, d_flags, n, d_idx, d_corr)
    return idx_a, d_idx.to_host(), d_corr.to_host(), dict(cb=d_cb, cbq=d_cbq, flags=d_flags, idx=d_idx)


def oracle_expected_idx(res, canon_sorted):
    """oracle PipelineResult -> expected idx after pass A and after pass B."""
    n = len(res.bc_state)
    exp_b = np.full(n, MISS, np.uint32)
    ok = res.bc_state > 0
    if ok.any():
        exp_b[ok] = ranks_of(canon_sorted, res.corrected_cb[ok])
    exp_a = exp_b.copy()
    exp_a[res.bc_state == 2] = MISS
    return exp_a, exp_b


def hist_as_rank_counts(hist, cb_len, canon_sorted):
    seqs, cnt = hist.dump_sorted(cb_len)
    out = np.zeros(len(canon_sorted), np.uint32)
    if len(cnt):
        out[ranks_of(canon_sorted, seqs)] = cnt.astype(np.uint32)
    return out
