"""Randomised differential run: the engine (host batches through every pass AND the command line's resident flow with
hint arrays and the two-stream pass 3) against the oracle, bit for bit, over random k, read lengths, read groups,
raggedness, N density, alpha, quality spread and batch splits.   python tests/fuzz_parity.py N_CASES SEED"""
import sys, os, time, traceback
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common
import ctypes
import torch
from kbbq_amd import _lib
from kbbq_amd.engine import Engine, plan_parameters
from kbbq_amd.reads import ReadBatch


def run_resident(d, k, alpha, n_rg, uniform, n_batches):
    """The command line's flow: batches uploaded once, hint arrays, asynchronous passes (pass 3 on two streams)."""
    alpha_ld, cov, approx = plan_parameters(d["genome_len"], d["coverage"], alpha)
    lens = np.diff(d["off"].astype(np.int64))
    e = Engine(k, alpha_ld, 777, approx, n_rg=n_rg, max_read_len=int(lens.max()))
    full = ReadBatch(d["seq"], d["qual"], d["off"], d["rg"], d["second"], uniform=uniform)
    n = full.n_reads
    cuts = [n * i // n_batches for i in range(n_batches + 1)]
    hb = [full] if n_batches == 1 else [full.slice(cuts[i], cuts[i + 1]) for i in range(n_batches)]
    hb = [b for b in hb if b.n_reads]
    devs, ordinal = [], 0
    for b in hb:
        db = e.upload(b)
        _lib.check(e.L.kbbq_reads_alloc_hints(ctypes.byref(db.c)))
        e.subsample_kmers(db, ordinal)
        ordinal += b.n_kmer_positions(k)
        devs.append(db)
    out = {}
    out["sampled_inserted"] = e.sample_finish()
    out["sampled_table"] = e.filter_table(0)
    e.compute_thresholds()
    for db in devs:
        e.find_trusted_kmers(db)
    out["trusted_inserted"] = e.trusted_finish()
    out["trusted_table"] = e.filter_table(1)
    for db in devs:
        e.get_covariatedata(db)
    out["cov"] = e.covariates()
    e.get_dqs()
    rec = []
    for b, db in zip(hb, devs):
        o = np.zeros(b.n_bases + 16, dtype=np.uint8)
        _lib.check(e.L.kbbq_recalibrate_batch_host(e.h, ctypes.byref(db.c), o.ctypes.data))
        rec.append(o[:b.n_bases])
    out["recal"] = np.concatenate(rec)
    for db in devs:
        _lib.check(e.L.kbbq_reads_free_hints(ctypes.byref(db.c)))
        db.free()
    e.close()
    return out


def same_resident(res, ora):
    assert res["sampled_inserted"] == ora["sampled_inserted"] and res["trusted_inserted"] == ora["trusted_inserted"]
    assert np.array_equal(res["sampled_table"], ora["sampled_table"]) and np.array_equal(res["trusted_table"], ora["trusted_table"])
    oc, ec = ora["cov"], res["cov"]
    R, Co = oc["R"], oc["C"]
    assert np.array_equal(ec["cycle"][:R, :, :, :Co], oc["cycle"]) and np.array_equal(ec["dinuc"][:R], oc["dinuc"])
    assert np.array_equal(res["recal"], ora["recal"])


def run_cases(n_cases, seed0, verbose=True):
    rng = np.random.RandomState(seed0)
    failures = []
    for case in range(n_cases):
        k = int(rng.choice([3, 5, 8, 11, 15, 16, 17, 21, 24, 27, 31, 32]))
        read_len = int(rng.choice([max(k + 3, 36), 50, 76, 100, 101, 150, 151, 193, 250, 300, 321, 400, 512, 513, 600, 777, 1100]))
        read_len = max(read_len, k + 2)
        n_rg = int(rng.choice([1, 1, 2, 3, 7]))
        paired = bool(rng.randint(0, 2))
        ragged = bool(rng.randint(0, 2))
        cov = int(rng.choice([8, 15, 25, 40]))
        glen = int(rng.choice([3000, 8000, 20000]))
        npm = int(rng.choice([0, 200, 3000, 20000]))
        alpha = None if rng.randint(0, 2) else float(rng.choice([0.02, 0.05, 0.15, 0.5, 0.9]))
        kw = dict(seed=int(rng.randint(1, 1 << 30)), genome_len=glen, coverage=cov, read_len=read_len, n_rg=n_rg, paired=paired,
                  n_per_million=npm, ragged=ragged, short_reads=int(rng.choice([0, 5])) if ragged else 0,
                  mid_reads=int(rng.choice([0, 20])) if ragged and read_len > 70 else 0,
                  extra_errors=int(rng.choice([0, 30])), clusters=int(rng.choice([0, 10])) if read_len >= 150 else 0)
        nb = int(rng.choice([1, 2, 5]))
        # slice-bucketed inserts forced onto these small filters in half of the cases, with a record capacity far below
        # a batch in some (every batch flushes, most records overflow into the direct path); soft-masked text in a quarter
        bucket = [None, "1"][rng.randint(0, 2)]
        records = [None, "2000", "60000"][rng.randint(0, 3)] if bucket else None
        soft = rng.randint(0, 4) == 0
        for name, val in (("KBBQ_BUCKET", bucket), ("KBBQ_BUCKET_RECORDS", records)):
            if val is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = val
        desc = "case %d k=%d %s alpha=%s batches=%d bucket=%s/%s soft=%d" % (case, k, kw, alpha, nb, bucket, records, soft)
        try:
            d = common.make_softmasked_dataset(frac=float(rng.choice([0.1, 0.5])), stretches=10, digits=5, **kw) if soft else common.make_dataset(**kw)
            if rng.randint(0, 3) == 0:
                d = dict(d)
                q = d["qual"].copy()
                r2 = np.random.RandomState(case)
                top = 45 if r2.randint(0, 3) else 256      # a third of these: qualities up to 255 (256 quality rows, round 3)
                q = np.where(q <= 2, q, r2.randint(3, top, size=len(q))).astype(np.uint8)
                d["qual"] = np.ascontiguousarray(q)
            t0 = time.time()
            ora = common.run_oracle(d, k=k, alpha=alpha, n_rg=n_rg)
            t1 = time.time()
            if ora.get("fpr_too_high"):
                if verbose: print("skip (fpr gate)", desc, flush=True)
                continue
            eng = common.run_engine(d, k=k, alpha=alpha, n_rg=n_rg, uniform=not ragged, n_batches=nb)
            common.assert_same_run(eng, ora)
            same_resident(run_resident(d, k, alpha, n_rg, not ragged, nb), ora)
            if verbose: print("ok   %.1fs/%.1fs walk=%d %s" % (t1 - t0, time.time() - t1, eng["stats"]["corrected_reads"], desc), flush=True)
        except Exception as ex:
            failures.append(desc)
            print("FAIL", desc, flush=True)
            traceback.print_exc()

    os.environ.pop("KBBQ_BUCKET", None)
    os.environ.pop("KBBQ_BUCKET_RECORDS", None)
    return failures


if __name__ == "__main__":
    f = run_cases(int(sys.argv[1]), int(sys.argv[2]))
    print("failures", len(f))
    sys.exit(1 if f else 0)
