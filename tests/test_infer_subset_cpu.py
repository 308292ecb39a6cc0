"""The two-phase form of infer_read_errors (k_infer<SUB>, kbbq_amd/csrc/kernels.h) restated in numpy on the bench's own reads:
phase 1 decides every base it can from a subset of the k-mer lookups (lower bound > threshold: clean; lower bound + skipped
<= threshold: flagged), phase 2 makes the skipped lookups around the undecided bases.  With an exact set standing in for the
sampled filter and the oracle's thresholds this checks, without a GPU, that (a) the decisions equal those made from ALL
lookups, for any choice of skipped starts, and (b) how many lookups that saves on this workload -- the prediction the kernel
was built on (0.86 for every fourth start; `python tests/test_infer_subset_cpu.py [GENOME_LEN]` prints
the whole table)."""
import sys

import numpy as np

import common
from kbbq_amd.engine import plan_parameters
from oracle import pyoracle

K, L, COV = 32, 150, 30


def setup(G):
    d = common.make_dataset(seed=12345, genome_len=G, coverage=COV, n_per_million=100)
    alpha, cov, approx = plan_parameters(G, COV)
    n = len(d["off"]) - 1
    o = pyoracle.Oracle(K, alpha, 777, approx)
    o.sample(d["seq"], d["off"])
    thr = np.array(o.compute_thresholds()[0])
    seq, qual = d["seq"].reshape(n, L), d["qual"].reshape(n, L)
    code = np.full(256, 4, dtype=np.int64)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    b = code[seq]
    nk = L - K + 1
    fw = np.zeros((n, nk), dtype=np.uint64)
    rc = np.zeros((n, nk), dtype=np.uint64)
    valid = np.ones((n, nk), dtype=bool)
    for j in range(K):
        x = b[:, j:j + nk]
        valid &= x < 4
        xx = np.where(x < 4, x, 0).astype(np.uint64)
        fw = (fw << np.uint64(2)) | xx
        rc = rc | ((np.uint64(3) - xx) << np.uint64(2 * j))
    key = np.minimum(fw, rc)
    rng = np.random.RandomState(1)
    hint = (rng.rand(n, nk) < float(alpha)) & valid            # what this read sampled itself: the hint bits
    pres = valid & np.isin(key, np.unique(key[hint]))          # an exact set for the filter (its 0.5 % false positives aside)
    return dict(n=n, nk=nk, thr=thr, pres=pres, need=valid & ~hint, lowq=qual <= 2)


def windows_sum(a, n, nk):
    c = np.zeros((n, nk + 1), dtype=np.int64)
    c[:, 1:] = np.cumsum(a, axis=1)
    i = np.arange(L)
    lo, hi = np.maximum(0, i - K + 1), np.minimum(i, nk - 1)
    return c[:, hi + 1] - c[:, lo], hi - lo + 1


def two_phase(S, skipmask):
    """returns (lookups of phase 1, of phase 2, flags) for the starts `skipmask` leaves out of phase 1"""
    n, nk, thr, pres, need, lowq = S["n"], S["nk"], S["thr"], S["pres"], S["need"], S["lowq"]
    skipped = need & skipmask
    lb, possible = windows_sum(pres & ~skipped, n, nk)
    nsk, _ = windows_sum(skipped, n, nk)
    t = thr[possible][None, :]
    und = ~lowq & (lb <= t) & (lb + nsk > t)
    cu = np.zeros((n, L + 1), dtype=np.int64)
    cu[:, 1:] = np.cumsum(und, axis=1)
    s = np.arange(nk)
    p2 = skipped & ((cu[:, s + K] - cu[:, s]) > 0)
    fin, _ = windows_sum((pres & ~skipped) | (p2 & pres), n, nk)
    return int((need & ~skipmask).sum()), int(p2.sum()), (fin <= t) | lowq


def full(S):
    inn, possible = windows_sum(S["pres"], S["n"], S["nk"])
    return int(S["need"].sum()), (inn <= S["thr"][possible][None, :]) | S["lowq"]


def test_two_phase_decisions_are_exact_and_save_lookups():
    S = setup(60000)
    n_full, err = full(S)
    s = np.arange(S["nk"])[None, :]
    rng = np.random.RandomState(3)
    for name, m in (("every 4th, 8 from the ends", ((s & 3) == 3) & (s >= 8) & (s < S["nk"] - 8)),
                    ("every 8th, 14 from the ends", ((s & 7) == 7) & (s >= 14) & (s < S["nk"] - 14)),
                    ("every 2nd, everywhere", (s & 1) == 1),
                    ("a random third", rng.rand(S["n"], S["nk"]) < 0.33)):
        p1, p2, e2 = two_phase(S, np.broadcast_to(m, (S["n"], S["nk"])))
        assert np.array_equal(e2, err), name                     # exact, whatever is skipped
        assert p1 + p2 <= n_full
    p1, p2, _ = two_phase(S, np.broadcast_to(((s & 3) == 3) & (s >= 8) & (s < S["nk"] - 8), (S["n"], S["nk"])))
    assert 0.80 < (p1 + p2) / n_full < 0.90                      # the kernel measures 0.861 at full scale (profiles/r04_ab_infer_subset.json)


if __name__ == "__main__":
    S = setup(int(sys.argv[1]) if len(sys.argv) > 1 else 200000)
    n_full, err = full(S)
    print("thresholds", S["thr"].tolist())
    print("lookups per read now %.1f, flagged bases %.3f" % (n_full / S["n"], err.mean()))
    s = np.arange(S["nk"])[None, :]
    for period in (4, 3, 2):
        for E in (0, 8, 16, 24):
            m = ((s % period) == period - 1) & (s >= E) & (s < S["nk"] - E)
            p1, p2, e2 = two_phase(S, np.broadcast_to(m, (S["n"], S["nk"])))
            print("every %d, %2d from the ends: phase 1 %.1f + phase 2 %.1f per read = %.3f of now, exact %s" % (
                period, E, p1 / S["n"], p2 / S["n"], (p1 + p2) / n_full, np.array_equal(e2, err)))
