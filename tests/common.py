"""Shared helpers for the parity tests: seeded inputs, the oracle run and the engine run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from kbbq_amd import synth  # noqa: E402
from kbbq_amd.engine import Engine, plan_parameters  # noqa: E402
from kbbq_amd.reads import ReadBatch  # noqa: E402
from oracle import pyoracle  # noqa: E402  (tests are allowed to use the oracle)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def make_dataset(seed=12345, genome_len=20000, coverage=20, read_len=150, n_rg=1, paired=False, n_per_million=200,
                 ragged=False, short_reads=0, mid_reads=0, extra_errors=0):
    """Seeded synthetic reads; optionally trimmed to ragged lengths and with a few reads shorter than k."""
    n_reads = genome_len * coverage // read_len
    sp = synth.synth_params(seed, genome_len, n_reads, read_len, n_rg=n_rg, paired=paired, n_per_million=n_per_million)
    d = synth.generate(sp)
    rng = np.random.RandomState(seed & 0xFFFF)
    if extra_errors:
        # substitutions at quality 37, spaced closer than k in some reads: leaves reads without any trusted
        # k-mer (correct_one) and ties / unfixable stretches (bad prefix / suffix recursion)
        seq = d["seq"].reshape(n_reads, read_len)
        for r in rng.choice(n_reads, size=extra_errors, replace=False):
            step = rng.randint(8, 60)
            for p in range(rng.randint(0, step), read_len, step):
                seq[r, p] = ord("ACGT"[("ACGT".find(chr(seq[r, p])) + 1 + rng.randint(0, 3)) % 4]) if chr(seq[r, p]) in "ACGT" else seq[r, p]
    if ragged or short_reads or mid_reads:
        lens = np.full(n_reads, read_len, dtype=np.int64)
        if ragged:
            lens = rng.randint(read_len * 2 // 3, read_len + 1, size=n_reads)
        if short_reads:
            idx = rng.choice(n_reads, size=short_reads, replace=False)
            lens[idx] = rng.randint(1, 31, size=short_reads)
        if mid_reads:
            idx = rng.choice(n_reads, size=mid_reads, replace=False)
            lens[idx] = rng.randint(32, 64, size=mid_reads)
        keep = (np.arange(read_len)[None, :] < lens[:, None]).reshape(-1)
        d["seq"] = np.ascontiguousarray(d["seq"][keep])
        d["qual"] = np.ascontiguousarray(d["qual"][keep])
        off = np.zeros(n_reads + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        d["off"] = off
    d["genome_len"] = genome_len
    d["coverage"] = coverage
    return d


def run_oracle(d, k=32, seed=777, alpha=None, n_rg=1):
    alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], alpha)
    o = pyoracle.Oracle(k, alpha_ld, seed, approx)
    rg = np.ascontiguousarray(d["rg"], dtype=np.int32)
    second = np.ascontiguousarray(d["second"], dtype=np.uint8)
    out = o.run_all(d["seq"], d["qual"], d["off"], rg, second)
    out["sampled_table"] = o.filter_table(0).copy()
    out["trusted_table"] = o.filter_table(1).copy()
    out["filter_info"] = [o.filter_info(0), o.filter_info(1)]
    out["patterns"] = [o.filter_patterns(0).copy(), o.filter_patterns(1).copy()]
    return out


def run_engine(d, k=32, seed=777, alpha=None, n_rg=1, uniform=False, n_batches=1, max_read_len=None):
    alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], alpha)
    lens = np.diff(d["off"].astype(np.int64))
    if max_read_len is None:
        max_read_len = int(lens.max())
    e = Engine(k, alpha_ld, seed, approx, n_rg=n_rg, max_read_len=max_read_len)
    full = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=uniform)
    n = full.n_reads
    cuts = [n * i // n_batches for i in range(n_batches + 1)]
    batches = [full] if n_batches == 1 else [full.slice(cuts[i], cuts[i + 1]) for i in range(n_batches)]
    out = {}
    ordinal = 0
    for b in batches:
        e.subsample_kmers(b, ordinal)
        ordinal += b.n_kmer_positions(k)
    out["sampled_inserted"] = e.sample_finish()
    out["sampled_table"] = e.filter_table(0)
    thr, fpr, p_text, too_high = e.compute_thresholds()
    out.update(thresholds=thr, fpr=fpr, p_text=p_text, fpr_too_high=too_high)
    out["infer_errors"] = np.concatenate([e.find_trusted_kmers(b, want_errors=True) for b in batches])
    out["trusted_inserted"] = e.trusted_finish()
    out["trusted_table"] = e.filter_table(1)
    out["errors"] = np.concatenate([e.get_covariatedata(b, want_errors=True) for b in batches])
    out["cov"] = e.covariates()
    out["dq"] = e.get_dqs()
    out["recal"] = np.concatenate([e.recalibrate(b) for b in batches])
    out["filter_info"] = [e.filter_info(0), e.filter_info(1)]
    out["stats"] = e.stats()
    e.close()
    return out


def assert_same_run(eng, ora, C=None):
    """Bit-exact comparison of every intermediate of the four passes."""
    for w in (0, 1):
        for key in ("bits", "bits_unblocked", "nhash", "nsalt", "random_seed"):
            assert eng["filter_info"][w][key] == ora["filter_info"][w][key], (w, key)
        assert np.array_equal(eng["filter_info"][w]["salts"], ora["filter_info"][w]["salts"])
    assert eng["sampled_inserted"] == ora["sampled_inserted"]
    assert np.array_equal(eng["sampled_table"], ora["sampled_table"]), "sampled Bloom bit array differs"
    assert np.array_equal(eng["thresholds"], ora["thresholds"])
    assert eng["p_text"] == ora["p_text"]
    assert eng["fpr"] == ora["fpr"]
    assert np.array_equal(eng["infer_errors"], ora["infer_errors"]), "infer_read_errors flags differ"
    assert eng["trusted_inserted"] == ora["trusted_inserted"]
    assert np.array_equal(eng["trusted_table"], ora["trusted_table"]), "trusted Bloom bit array differs"
    bad = np.nonzero(eng["errors"] != ora["errors"])[0]
    assert len(bad) == 0, "get_errors flags differ at %d bases, first %s" % (len(bad), bad[:10])
    oc, ec = ora["cov"], eng["cov"]
    R, Co = oc["R"], oc["C"]
    assert ec["R"] >= R and ec["C"] >= Co
    assert np.array_equal(ec["rg"][:R], oc["rg"])
    assert np.array_equal(ec["q"][:R], oc["q"])
    assert np.array_equal(ec["cycle"][:R, :, :, :Co], oc["cycle"])
    assert ec["cycle"][:R, :, :, Co:].sum() == 0
    assert np.array_equal(ec["dinuc"][:R], oc["dinuc"])
    od, ed = ora["dq"], eng["dq"]
    assert np.array_equal(ed["meanq"][:R], od["meanq"])
    assert np.array_equal(ed["rg"][:R], od["rg"])
    assert np.array_equal(ed["q"][:R], od["q"])
    assert np.array_equal(ed["cycle"][:R, :, :, :Co], od["cycle"])
    assert np.array_equal(ed["dinuc"][:R], od["dinuc"])
    bad = np.nonzero(eng["recal"] != ora["recal"])[0]
    assert len(bad) == 0, "recalibrated qualities differ at %d bases" % len(bad)
