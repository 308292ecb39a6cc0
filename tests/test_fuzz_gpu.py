"""Seeded random shapes (k, read length, read groups, ragged / short reads, N density, alpha, quality spread, batch
splits) through both ways over the C ABI -- host batches on every pass, and uploaded-once batches with hint arrays
and the two-stream pass 3 -- against the oracle, bit for bit.  tests/fuzz_parity.py N SEED runs longer sweeps."""
import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_bit_exact():
    import fuzz_parity
    assert fuzz_parity.run_cases(40, 2026, verbose=False) == []
