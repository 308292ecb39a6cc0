#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference itself cannot run in this image (its path sources need htslib), so these vectors
come from the ORACLE (oracle/kbbq_oracle.cc), which is pinned by tests/test_oracle_kat.py.  They
freeze its behaviour: a later change to the oracle or to the engine that alters any output of the
four passes on these inputs fails tests/test_golden.py.

Usage: python tests/golden/make_golden.py [case ...]      (no argument: every case)

mixed_k32 and small_k9 date from round 1 (94 quality rows in their delta-Q arrays and digests) and have not been
rewritten since: tests/test_golden.py compares the rows they hold.  highq_k32 (round 3) has qualities up to 255.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import common  # noqa: E402
from oracle import pyoracle  # noqa: E402

CASES = {
    # name: (dataset kwargs, run kwargs)
    "mixed_k32": (dict(seed=2026, genome_len=6000, coverage=25, n_rg=2, paired=True, n_per_million=3000, ragged=True,
                       short_reads=10, mid_reads=80, extra_errors=80, clusters=60), dict(k=32, n_rg=2)),
    "small_k9": (dict(seed=909, genome_len=2500, coverage=30, read_len=100, extra_errors=60), dict(k=9)),
    # qualities above KBBQ_MAXQ = 93 (a BAM can hold them): the reference's tables grow with the largest quality seen
    "highq_k32": (dict(seed=4242, genome_len=5000, coverage=25, n_rg=2, paired=True, extra_errors=60, high_quals=2500), dict(k=32, n_rg=2)),
}


def build(name):
    dkw, rkw = CASES[name]
    dkw = dict(dkw)
    high = dkw.pop("high_quals", 0)
    d = common.make_dataset(**dkw)
    if high:      # sprinkle qualities 94..255 over the reads
        rng = np.random.RandomState(dkw["seed"])
        q = d["qual"].copy()
        at = rng.choice(len(q), size=high, replace=False)
        q[at] = rng.choice([94, 95, 100, 127, 128, 200, 254, 255], size=high).astype(np.uint8)
        d = dict(d, qual=np.ascontiguousarray(q))
    o = common.run_oracle(d, **rkw)
    out = dict(seq=d["seq"], qual=d["qual"], off=d["off"], rg=np.asarray(d["rg"], np.int32), second=d["second"],
               genome_len=d["genome_len"], coverage=d["coverage"], k=rkw.get("k", 32), n_rg=rkw.get("n_rg", 1),
               sampled_inserted=o["sampled_inserted"], trusted_inserted=o["trusted_inserted"], thresholds=o["thresholds"],
               p_text=np.frombuffer(o["p_text"].encode(), dtype=np.uint8), fpr=o["fpr"],
               sampled_digest=pyoracle.fnv1a64(o["sampled_table"].tobytes()),
               trusted_digest=pyoracle.fnv1a64(o["trusted_table"].tobytes()),
               infer_errors=np.packbits(o["infer_errors"], bitorder="little"),
               errors=np.packbits(o["errors"], bitorder="little"), recal=o["recal"],
               dq_meanq=o["dq"]["meanq"], dq_rg=o["dq"]["rg"], dq_q=o["dq"]["q"], dq_cycle=o["dq"]["cycle"].astype(np.int8),
               dq_dinuc=o["dq"]["dinuc"].astype(np.int8), cov_rg=o["cov"]["rg"], cov_q=o["cov"]["q"],
               cov_cycle_digest=pyoracle.fnv1a64(o["cov"]["cycle"].tobytes()),
               cov_dinuc_digest=pyoracle.fnv1a64(o["cov"]["dinuc"].tobytes()))
    return out


if __name__ == "__main__":
    for name in (sys.argv[1:] or CASES):
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **build(name))
        print("wrote", name, os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")
