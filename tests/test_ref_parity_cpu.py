"""The oracle against the REFERENCE's own read-level path (rows a7-a15 of SURVEY.md section 8).

oracle/_ref/libref_path.so is the reference's bloom.cc, readutils.cc, covariateutils.cc, recalibrateutils.cc, htsiter.cc
and minion.cc compiled in place (oracle/Makefile: ref_path, harness oracle/ref_path_probe.cc).  Those sources need
htslib, and the build recipe fires only where a REAL htslib is installed -- this repository holds no stand-in for
it.  Where the library is absent (this image, today) the tests skip and the read-level parity stays "unpinned"
(DESIGN.md section 3); where it is present they compare every intermediate of the four passes on the parity inputs.
"""
import ctypes
import os

import numpy as np
import pytest

import common
from oracle import pyoracle

SO = os.path.join(common.ROOT, "oracle", "_ref", "libref_path.so")
pytestmark = pytest.mark.skipif(not os.path.exists(SO), reason="oracle/_ref/libref_path.so not built: no real htslib on this machine")

# Every parity input of tests/common.py: the harness hands the reference the raw sequence text (soft-masked cases keep
# their case), the qualities as FASTQ characters (a value above 93 survives the char round trip modulo 256, as it would
# from a BAM), read-group labels in order of first appearance, reads of any length.
CASES = sorted(common.PARITY_CASES)


def _first_appearance(rg):
    """The reference numbers read groups in order of first appearance (readutils.cc:100-103)."""
    order, out = {}, np.empty_like(rg)
    for i, g in enumerate(rg.tolist()):
        out[i] = order.setdefault(g, len(order))
    return out


def run_reference(d, k=32, seed=777, alpha=None, n_rg=1):
    from kbbq_amd.engine import long_double_text, plan_parameters
    L = ctypes.CDLL(SO)
    u8p, u64p, i32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int32)
    L.rp_filter_bits.restype = ctypes.c_uint64
    L.rp_filter_bits.argtypes = [ctypes.c_uint64, ctypes.c_double]
    L.rp_run.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_double, ctypes.c_double,
                         ctypes.c_uint64, u8p, u8p, u64p, i32p, u8p, u64p, u64p, i32p, ctypes.POINTER(ctypes.c_double),
                         ctypes.c_char_p, ctypes.c_size_t, u8p, u8p, u8p, u64p, u64p]
    alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], alpha)
    fs, ft = float(np.longdouble("0.01")), float(np.longdouble("0.0005"))
    nb = len(d["seq"])
    seq = np.ascontiguousarray(d["seq"], dtype=np.uint8)
    qual = np.ascontiguousarray(d["qual"], dtype=np.uint8)
    off = np.ascontiguousarray(d["off"], dtype=np.uint64)
    rg = np.ascontiguousarray(d["rg"], dtype=np.int32)
    second = np.ascontiguousarray(d["second"], dtype=np.uint8)
    t0 = np.zeros(L.rp_filter_bits(approx, fs) // 64, dtype=np.uint64)
    t1 = np.zeros(L.rp_filter_bits(approx, ft) // 64, dtype=np.uint64)
    thr = np.zeros(k + 1, dtype=np.int32)
    ins0, ins1, fpr = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_double()
    ptxt = ctypes.create_string_buffer(64)
    infer, errs, recal = (np.zeros(nb, dtype=np.uint8) for _ in range(3))
    P = lambda a, t: a.ctypes.data_as(t)
    rc = L.rp_run(k, long_double_text(alpha_ld), seed & 0xFFFFFFFF, approx, fs, ft, len(off) - 1, P(seq, u8p), P(qual, u8p), P(off, u64p),
                  P(rg, i32p), P(second, u8p), ctypes.byref(ins0), ctypes.byref(ins1), P(thr, i32p), ctypes.byref(fpr), ptxt, 64,
                  P(infer, u8p), P(errs, u8p), P(recal, u8p), P(t0, u64p), P(t1, u64p))
    return dict(rc=rc, sampled_inserted=ins0.value, trusted_inserted=ins1.value, thresholds=thr, fpr=fpr.value, p_text=ptxt.value.decode(),
                infer_errors=infer.astype(bool), errors=errs.astype(bool), recal=recal, sampled_table=t0, trusted_table=t1)


@pytest.mark.parametrize("name", CASES)
def test_oracle_equals_the_reference_on(name):
    maker, dkw, rkw, _ = common.PARITY_CASES[name]
    d = maker(**dkw)
    d["rg"] = _first_appearance(np.asarray(d["rg"], dtype=np.int32))
    ref = run_reference(d, **rkw)
    ora = common.run_oracle(d, **rkw)
    assert ref["rc"] == 0
    for key in ("sampled_inserted", "trusted_inserted", "fpr", "p_text"):
        assert ref[key] == ora[key], key
    for key in ("thresholds", "sampled_table", "trusted_table", "infer_errors", "errors", "recal"):
        assert np.array_equal(np.asarray(ref[key]), np.asarray(ora[key])), key
