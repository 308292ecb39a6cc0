"""The parity suite is only as good as its inputs: the oracle counts how often each branch of
get_errors (readutils.cc:238-570) runs, and the seeded cases of tests/common.py:PARITY_CASES must
together reach all of them.  (The same inputs are compared bit for bit against the HIP engine in
tests/test_parity_gpu.py.)"""
import ctypes

import common
from oracle import pyoracle


def test_parity_inputs_reach_every_branch_of_get_errors():
    L = pyoracle.lib()
    L.ko_counter_name.restype = ctypes.c_char_p
    L.ko_counter_value.restype = ctypes.c_uint64
    L.ko_counters_reset()
    for name, (build, dkw, rkw, ekw) in common.PARITY_CASES.items():
        out = common.run_oracle(build(**dkw), **rkw)
        assert not out["fpr_too_high"], name
        assert out["trusted_inserted"] > 0, name
    counts = {L.ko_counter_name(i).decode(): L.ko_counter_value(i) for i in range(L.ko_counter_count())}
    assert len(counts) == 25
    missing = [k for k, v in counts.items() if v == 0]
    assert not missing, "no parity input reaches: %s" % missing
    # the rare ones are hit more than once
    for key in ("tie_continue", "tie_stop", "adjust_moved", "early_patch_return", "unflag_backjump", "prefix_recursion"):
        assert counts[key] >= 5, (key, counts[key])
    # soft-masked text: the candidate equal to an off-case base is tried (and sets `multiple` in the anchor adjustment)
    for key in ("offcase_fix_candidate", "offcase_adjust_multiple"):
        assert counts[key] >= 100, (key, counts[key])


def test_the_case_of_a_base_changes_the_reference_answer():
    """The soft-masked parity inputs are only a test of the raw-character comparisons (bloom.cc:142,218,249;
    readutils.cc:202) if folding their case changes the oracle's result: it does."""
    import numpy as np
    build, dkw, rkw, _ = common.PARITY_CASES["softmasked"]
    d = build(**dkw)
    folded = d["seq"].copy()
    for a, b in zip(b"acgt0123", b"ACGTACGT"):
        folded[folded == a] = b
    soft = common.run_oracle(d, **rkw)
    plain = common.run_oracle(dict(d, seq=folded), **rkw)
    assert np.array_equal(soft["trusted_table"], plain["trusted_table"]) and np.array_equal(soft["infer_errors"], plain["infer_errors"])
    assert (soft["errors"] != plain["errors"]).sum() >= 5 and (soft["recal"] != plain["recal"]).any()
