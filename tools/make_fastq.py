#!/usr/bin/env python3
"""tools/make_fastq.py OUT.fq[.gz] GENOME_LEN COVERAGE -- a synthetic FASTQ from the bench generator's host twin
(kbbq_amd/synth.py), for timing the command line end to end.  Fixed-width names, 150-base reads."""
import gzip
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kbbq_amd import synth  # noqa: E402

out, G, cov = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L = 150
n_reads = G * cov // L
sp = synth.synth_params(12345, G, n_reads, L, n_rg=1, paired=False, n_per_million=100)
opener = (lambda p: gzip.open(p, "wb", compresslevel=4)) if out.endswith(".gz") else (lambda p: open(p, "wb"))
with opener(out) as fh:
    step = 200000
    for first in range(0, n_reads, step):
        n = min(step, n_reads - first)
        d = synth.generate(sp, first, n)
        names = np.char.zfill(np.arange(first, first + n).astype("S10"), 10)
        rec = np.empty((n, 1 + 1 + 10 + 3 + L + 3 + L + 1), dtype=np.uint8)      # @r0000000001/1\nSEQ\n+\nQUAL\n
        rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
        rec[:, 2:12] = np.frombuffer(names.tobytes(), dtype=np.uint8).reshape(n, 10)
        rec[:, 12] = ord("/"); rec[:, 13] = ord("1"); rec[:, 14] = 10
        rec[:, 15:15 + L] = d["seq"].reshape(n, L)
        rec[:, 15 + L] = 10; rec[:, 16 + L] = ord("+"); rec[:, 17 + L] = 10
        rec[:, 18 + L:18 + 2 * L] = d["qual"].reshape(n, L) + 33
        rec[:, 18 + 2 * L] = 10
        fh.write(rec.tobytes())
print("wrote %s: %d reads, %d bases" % (out, n_reads, n_reads * L))
