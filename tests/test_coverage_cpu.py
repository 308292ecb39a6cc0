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
    assert len(counts) == 21
    missing = [k for k, v in counts.items() if v == 0]
    assert not missing, "no parity input reaches: %s" % missing
    # the rare ones are hit more than once
    for key in ("tie_continue", "tie_stop", "adjust_moved", "early_patch_return", "unflag_backjump", "prefix_recursion"):
        assert counts[key] >= 5, (key, counts[key])
