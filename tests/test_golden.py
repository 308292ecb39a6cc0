"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the
oracle): the oracle must keep reproducing them (CPU), and the HIP engine must reproduce them on the
GPU without the oracle in the loop."""
import os

import numpy as np
import pytest

import common
from oracle import pyoracle

CASES = ["mixed_k32", "small_k9"]


def load(name):
    z = np.load(os.path.join(common.GOLDEN, name + ".npz"))
    d = dict(seq=z["seq"], qual=z["qual"], off=z["off"], rg=z["rg"], second=z["second"],
             genome_len=int(z["genome_len"]), coverage=int(z["coverage"]))
    return z, d


def check(z, run):
    n = len(z["seq"])
    assert run["sampled_inserted"] == int(z["sampled_inserted"])
    assert run["trusted_inserted"] == int(z["trusted_inserted"])
    assert np.array_equal(run["thresholds"], z["thresholds"])
    assert run["p_text"].encode() == z["p_text"].tobytes()
    assert run["fpr"] == float(z["fpr"])
    assert pyoracle.fnv1a64(run["sampled_table"].tobytes()) == int(z["sampled_digest"])
    assert pyoracle.fnv1a64(run["trusted_table"].tobytes()) == int(z["trusted_digest"])
    assert np.array_equal(run["infer_errors"], np.unpackbits(z["infer_errors"], bitorder="little")[:n])
    assert np.array_equal(run["errors"], np.unpackbits(z["errors"], bitorder="little")[:n])
    assert np.array_equal(run["recal"], z["recal"])
    R, C = z["dq_cycle"].shape[0], z["dq_cycle"].shape[3]
    assert np.array_equal(run["dq"]["meanq"][:R], z["dq_meanq"])
    assert np.array_equal(run["dq"]["rg"][:R], z["dq_rg"])
    assert np.array_equal(run["dq"]["q"][:R], z["dq_q"])
    assert np.array_equal(run["dq"]["cycle"][:R, :, :, :C], z["dq_cycle"])
    assert np.array_equal(run["dq"]["dinuc"][:R], z["dq_dinuc"])
    assert np.array_equal(run["cov"]["rg"][:R], z["cov_rg"])
    assert np.array_equal(run["cov"]["q"][:R], z["cov_q"])
    assert pyoracle.fnv1a64(np.ascontiguousarray(run["cov"]["cycle"][:R, :, :, :C]).tobytes()) == int(z["cov_cycle_digest"])
    assert pyoracle.fnv1a64(np.ascontiguousarray(run["cov"]["dinuc"][:R]).tobytes()) == int(z["cov_dinuc_digest"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    z, d = load(name)
    check(z, common.run_oracle(d, k=int(z["k"]), n_rg=int(z["n_rg"])))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_engine_reproduces_golden(name):
    z, d = load(name)
    check(z, common.run_engine(d, k=int(z["k"]), n_rg=int(z["n_rg"]), uniform=False, n_batches=2))
