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

NQ = pyoracle.NQ      # quality rows of every covariate / delta-Q table (include/kbbq_engine.h: KBBQ_NQ)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def make_dataset(seed=12345, genome_len=20000, coverage=20, read_len=150, n_rg=1, paired=False, n_per_million=200,
                 ragged=False, short_reads=0, mid_reads=0, extra_errors=0, clusters=0, ragged_min=None):
    """Seeded synthetic reads; optionally trimmed to ragged lengths and with a few reads shorter than k."""
    n_reads = genome_len * coverage // read_len
    sp = synth.synth_params(seed, genome_len, n_reads, read_len, n_rg=n_rg, paired=paired, n_per_million=n_per_million)
    d = synth.generate(sp)
    rng = np.random.RandomState(seed & 0xFFFF)
    if extra_errors:
        # substitutions at quality 37, spaced closer than k in some reads: leaves reads without any trusted
        # k-mer (correct_one) and ties / unfixable stretches (bad prefix / suffix recursion)
        seq = d["seq"].reshape(n_reads, read_len)
        for r in rng.choice(n_reads, size=min(extra_errors, n_reads), replace=False):
            step = rng.randint(8, 60)
            for p in range(rng.randint(0, step), read_len, step):
                seq[r, p] = ord("ACGT"[("ACGT".find(chr(seq[r, p])) + 1 + rng.randint(0, 3)) % 4]) if chr(seq[r, p]) in "ACGT" else seq[r, p]
    if clusters:
        # 6-9 substitutions 2-4 bases apart: each is fixable in turn, so the over-correction window
        # (more than 4 fixes in 20 bases, readutils.cc:479-546) fires
        seq = d["seq"].reshape(n_reads, read_len)
        for r in rng.choice(n_reads, size=min(clusters, n_reads), replace=False):
            p = rng.randint(35, read_len - 60)
            for _ in range(rng.randint(6, 10)):
                if chr(seq[r, p]) in "ACGT":
                    seq[r, p] = ord("ACGT"[("ACGT".find(chr(seq[r, p])) + 1 + rng.randint(0, 3)) % 4])
                p += rng.randint(2, 5)
    if ragged or short_reads or mid_reads:
        lens = np.full(n_reads, read_len, dtype=np.int64)
        if ragged:
            lens = rng.randint(read_len * 2 // 3 if ragged_min is None else ragged_min, read_len + 1, size=n_reads)
        if short_reads:
            idx = rng.choice(n_reads, size=min(short_reads, n_reads), replace=False)
            lens[idx] = rng.randint(1, 31, size=len(idx))
        if mid_reads:
            idx = rng.choice(n_reads, size=min(mid_reads, n_reads), replace=False)
            lens[idx] = rng.randint(32, 64, size=len(idx))
        keep = (np.arange(read_len)[None, :] < lens[:, None]).reshape(-1)
        d["seq"] = np.ascontiguousarray(d["seq"][keep])
        d["qual"] = np.ascontiguousarray(d["qual"][keep])
        off = np.zeros(n_reads + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        d["off"] = off
    d["genome_len"] = genome_len
    d["coverage"] = coverage
    return d


def make_repeat_dataset(seed=77, genome_len=12000, coverage=30, read_len=100, n_variants=12):
    """A genome with a duplicated 400-base segment whose copies differ at a few single bases, and reads
    that carry a sequencing error exactly at such a base: two alternative bases are then equally good
    (both copies are in the filter) -- the tie of find_longest_fix that runs a full window
    (readutils.cc:293-322 without the early stop)."""
    rng = np.random.RandomState(seed)
    g = rng.randint(0, 4, size=genome_len)
    src, dst, seg = 1000, 7000, 400
    g[dst:dst + seg] = g[src:src + seg]
    var = np.sort(rng.choice(np.arange(60, seg - 60), size=n_variants, replace=False))
    var = var[np.concatenate(([True], np.diff(var) > 40))]
    g[dst + var] = (g[src + var] + 1 + rng.randint(0, 2, size=len(var))) % 4
    n_reads = genome_len * coverage // read_len
    starts = rng.randint(0, genome_len - read_len + 1, size=n_reads)
    # make sure both copies are well covered
    extra = np.concatenate([rng.randint(src - 50, src + seg - 50, size=200), rng.randint(dst - 50, dst + seg - 50, size=200)])
    starts = np.concatenate([starts, extra])
    n_reads = len(starts)
    codes = g[starts[:, None] + np.arange(read_len)[None, :]]
    qual = np.full((n_reads, read_len), 37, dtype=np.uint8)
    # ordinary errors
    err = rng.rand(n_reads, read_len) < 0.002
    codes = np.where(err, (codes + 1 + rng.randint(0, 3, size=codes.shape)) % 4, codes)
    # errors on the variant bases: a third base, neither copy's
    third = 0
    for r in range(n_reads):
        for base, v in ((src, var), (dst, var)):
            for x in v:
                p = base + x - starts[r]
                if 35 <= p < read_len - 35 and rng.rand() < 0.5:
                    a, b = g[src + x], g[dst + x]
                    c = [y for y in range(4) if y != a and y != b][rng.randint(0, 2)]
                    codes[r, p] = c
                    third += 1
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    off = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    return dict(seq=np.ascontiguousarray(seq.reshape(-1)), qual=np.ascontiguousarray(qual.reshape(-1)), off=off,
                rg=np.zeros(n_reads, dtype=np.int32), second=np.zeros(n_reads, dtype=np.uint8), genome_len=genome_len,
                coverage=coverage)


def make_sparse_rg_dataset(**kw):
    """Five declared read groups of which one never occurs and one occurs in a single read."""
    d = make_dataset(n_rg=5, **kw)
    rg = d["rg"].copy()
    rg[rg == 3] = 1
    rg[rg == 4] = 0
    rg[len(rg) // 2] = 4
    d["rg"] = rg
    return d


def make_wide_quality_dataset(seed=40, **kw):
    """Un-binned qualities: every value 2..41 occurs, in several read groups, so the tally's and the apply
    kernel's LDS tables (compacted over the qualities present) do not hold all groups at once."""
    d = make_dataset(seed=seed, **kw)
    rng = np.random.RandomState(seed)
    d["qual"] = np.ascontiguousarray(np.where(d["qual"] <= 2, d["qual"], rng.randint(3, 42, size=len(d["qual"]))).astype(np.uint8))
    return d


def make_high_quality_dataset(seed=94, share=0.4, **kw):
    """Qualities above KBBQ_MAXQ = 93 (a BAM can hold up to 255; 0xFF is what "missing" looks like): a share of the best
    bases gets 94, 95, 120, 200, 254 or 255.  The reference's tables grow with the largest quality seen
    (covariateutils.cc:65-76,102-116,147-164): those bases are tallied, modelled in rows of their own and only clamped on
    output (readutils.cc:592-594)."""
    d = make_dataset(seed=seed, **kw)
    rng = np.random.RandomState(seed)
    q = d["qual"].copy()
    best = np.nonzero(q == q.max())[0]
    at = rng.choice(best, size=int(len(best) * share), replace=False)
    q[at] = rng.choice([94, 95, 120, 200, 254, 255], size=len(at)).astype(np.uint8)
    d["qual"] = np.ascontiguousarray(q)
    return d


def make_softmasked_dataset(seed=707, frac=0.04, stretches=120, digits=30, **kw):
    """Soft-masked FASTQ text: scattered lower-case bases, whole lower-case stretches (masked repeats), and a few of
    the digits '0'..'3' that seq_nt16_table also folds to bases.  K-mers and covariates fold case, but the reference
    compares raw characters in find_longest_fix, adjust_right_anchor and correct_one (bloom.cc:142,218,249;
    readutils.cc:202): for such a base the candidate equal to it is tried too."""
    d = make_dataset(seed=seed, **kw)
    rng = np.random.RandomState(seed)
    seq = d["seq"].copy()
    off = d["off"].astype(np.int64)
    acgt = np.isin(seq, np.frombuffer(b"ACGT", dtype=np.uint8))
    low = (rng.rand(len(seq)) < frac) & acgt
    n = len(off) - 1
    for r in rng.choice(n, size=min(stretches, n), replace=False):
        a, b = off[r], off[r + 1]
        if b - a < 20:
            continue
        s = rng.randint(a, b - 10)
        low[s:min(b, s + rng.randint(10, 80))] = True
    low &= acgt
    seq[low] = seq[low] + 32                      # 'A' -> 'a'
    cand = np.nonzero(acgt & ~low)[0]
    idx = rng.choice(cand, size=min(digits, len(cand)), replace=False)
    seq[idx] = np.frombuffer(b"0123", dtype=np.uint8)[np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), seq[idx])]
    d["seq"] = np.ascontiguousarray(seq)
    return d


# The seeded inputs of the GPU parity suite: name -> (dataset builder, dataset kwargs, run kwargs, engine kwargs).
# tests/test_coverage_cpu.py proves on the CPU that together they reach every branch of get_errors.
PARITY_CASES = {
    "uniform_150": (make_dataset, dict(seed=12345, genome_len=30000, coverage=20), dict(), dict(uniform=True)),
    "ragged_2rg_paired": (make_dataset, dict(seed=99, genome_len=25000, coverage=24, n_rg=2, paired=True, n_per_million=3000,
                                             ragged=True, short_reads=40, mid_reads=300, extra_errors=200),
                          dict(n_rg=2), dict(uniform=False, n_batches=3)),
    "k21_low_alpha": (make_dataset, dict(seed=777, genome_len=20000, coverage=40, read_len=100), dict(k=21, alpha=0.05),
                      dict(uniform=True)),
    "reads_250": (make_dataset, dict(seed=31, genome_len=20000, coverage=20, read_len=250, n_per_million=1000), dict(),
                  dict(uniform=True)),
    "noisy": (make_dataset, dict(seed=4242, genome_len=30000, coverage=25, n_per_million=5000, mid_reads=800,
                                 extra_errors=600), dict(), dict(uniform=False)),
    "clusters": (make_dataset, dict(seed=606, genome_len=30000, coverage=25, clusters=400, extra_errors=100), dict(),
                 dict(uniform=True, n_batches=2)),
    "k9": (make_dataset, dict(seed=9, genome_len=3000, coverage=30, read_len=100, extra_errors=100), dict(k=9),
           dict(uniform=True)),
    "k12_clusters": (make_dataset, dict(seed=12, genome_len=6000, coverage=30, read_len=100, clusters=50), dict(k=12),
                     dict(uniform=True)),
    "repeat_ties": (make_repeat_dataset, dict(), dict(k=21), dict(uniform=True)),
    # BASELINE.json configs[0] (1 Mbp, 20x, k=32: the reference's own CPU-runnable case) and configs[4] (60x, k=21,
    # -c 60 -a 0.05) at oracle-friendly genome sizes
    "config0_1Mbp_20x": (make_dataset, dict(seed=2020, genome_len=1_000_000, coverage=20), dict(), dict(uniform=True, n_batches=2)),
    "config4_60x_k21": (make_dataset, dict(seed=6021, genome_len=40_000, coverage=60), dict(k=21, alpha=0.05), dict(uniform=True)),
    "sparse_read_groups": (make_sparse_rg_dataset, dict(seed=55, genome_len=20000, coverage=20, paired=True, extra_errors=60), dict(n_rg=5),
                           dict(uniform=True, n_batches=3)),
    "wide_qualities_6rg_250": (make_wide_quality_dataset, dict(genome_len=15000, coverage=24, read_len=250, n_rg=6, paired=True,
                                                              extra_errors=40), dict(n_rg=6), dict(uniform=True, n_batches=2)),
    # soft-masked text (lower-case bases and digits): raw-character comparisons of the reference, see make_softmasked_dataset
    "softmasked": (make_softmasked_dataset, dict(seed=606, frac=0.5, genome_len=30000, coverage=25, clusters=400, extra_errors=100), dict(),
                   dict(uniform=True, n_batches=2)),
    "softmasked_k9_ragged": (make_softmasked_dataset, dict(seed=9, frac=0.3, genome_len=3000, coverage=30, read_len=100, extra_errors=100,
                                                          ragged=True, n_rg=2, paired=True), dict(k=9, n_rg=2), dict(uniform=False, n_batches=3)),
    # reads longer than the 512 bases the staged kernels hold: windowed kernels, run-time-sized walk (long_reads.h)
    "long_ragged_600_3000": (make_dataset, dict(seed=3000, genome_len=60000, coverage=24, read_len=3000, ragged=True, ragged_min=600,
                                                n_per_million=1500, extra_errors=150, clusters=60, n_rg=2, paired=True),
                             dict(n_rg=2), dict(uniform=False, n_batches=2)),
    "long_uniform_1000_softmasked_k25": (make_softmasked_dataset, dict(seed=1000, frac=0.3, genome_len=40000, coverage=24, read_len=1000,
                                                                      extra_errors=80, clusters=40), dict(k=25), dict(uniform=True, n_batches=3)),
    # reads longer than the 4608 bases k_tally takes in windows of 192 cycles: one launch, cycle counters straight to the histograms
    "long_ragged_2000_9000": (make_dataset, dict(seed=9000, genome_len=90000, coverage=24, read_len=9000, ragged=True, ragged_min=2000,
                                                 n_per_million=1500, extra_errors=20, clusters=10, n_rg=2, paired=True),
                              dict(n_rg=2), dict(uniform=False, n_batches=2)),
    # ... and than the 65 535 bases that were the engine's limit until round 4 (positions in 16 bits): nanopore-class lengths
    "long_ragged_30000_70000": (make_dataset, dict(seed=70000, genome_len=300000, coverage=14, read_len=70000, ragged=True, ragged_min=30000,
                                                   n_per_million=2500, extra_errors=6, clusters=4),
                                dict(), dict(uniform=False, n_batches=2)),
    "reads_400": (make_dataset, dict(seed=400, genome_len=20000, coverage=20, read_len=400, n_per_million=500,
                                     extra_errors=60), dict(), dict(uniform=True)),
    # qualities above 93 (BAM-only values): 256 quality rows in every table, as the reference's growing tables
    "high_qualities_2rg": (make_high_quality_dataset, dict(genome_len=20000, coverage=24, n_rg=2, paired=True, extra_errors=60), dict(n_rg=2),
                           dict(uniform=False, n_batches=2)),
}


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


def run_engine(d, k=32, seed=777, alpha=None, n_rg=1, uniform=False, n_batches=1, max_read_len=None, tune=None):
    alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], alpha)
    lens = np.diff(d["off"].astype(np.int64))
    if max_read_len is None:
        max_read_len = int(lens.max())
    e = Engine(k, alpha_ld, seed, approx, n_rg=n_rg, max_read_len=max_read_len, tune=tune)
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
