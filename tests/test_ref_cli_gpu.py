"""The command line against the REFERENCE's own command line (rows f1-f4 of SURVEY.md section 8).

oracle/_ref/kbbq_ref is the reference's kbbq.cc + bloom.cc, readutils.cc, covariateutils.cc, recalibrateutils.cc,
htsiter.cc and minion.cc compiled in place against the machine's REAL htslib (oracle/Makefile: ref_cli; the recipe fires
only where `#include <htslib/hts.h>` and -lhts work -- this repository holds no stand-in for the library).  Where it is
absent (this image, today) the tests skip and the htslib side of the path -- kseq record splitting, BAM aux handling, the
read-name rules in situ, BGZF framing -- stays "parity unpinned" (DESIGN.md section 3).  Where it is present:
the reference runs first (it draws its sampler seed from time and pid and prints it: kbbq.cc:268-271), kbbq_amd/kbbq runs
on the same file with KBBQ_SEED set to that number, and the two DECOMPRESSED outputs must be the same records in the same
order: FASTQ text line by line; BAM header and alignment blocks byte for byte, with --use-oq / --set-oq."""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

import bamutil
import common
from test_cli_gpu import CLI, bam_dataset, named_dataset, run_cli, write_fastq

REF = os.path.join(common.ROOT, "oracle", "_ref", "kbbq_ref")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/kbbq_ref not built: no real htslib on this machine")]


def run_reference(args):
    p = subprocess.run([REF] + [str(a) for a in args], capture_output=True, timeout=3600)
    err = p.stderr.decode(errors="replace")
    m = re.search(r"Seed: (\d+)", err)
    return p.returncode, p.stdout, err, (int(m.group(1)) if m else None)


def test_fastq_output_equals_the_reference(tmp_path):
    d, names, n_rg = named_dataset(seed=4001, genome_len=30000, coverage=24, n_per_million=3000, ragged=True, mid_reads=100, extra_errors=150)
    comments = ["c%d extra" % r if r % 5 == 0 else "" for r in range(len(names))]
    fq = tmp_path / "in.fq.gz"
    write_fastq(fq, d, names, comments)
    rc, want, err, seed = run_reference(["-g", d["genome_len"], fq])
    assert rc == 0 and seed is not None, err[-2000:]
    rc, got, err2 = run_cli(["-g", d["genome_len"], fq], {"KBBQ_SEED": str(seed)})
    assert rc == 0, err2
    assert gzip.decompress(got) == gzip.decompress(want)
    # the log lines that carry numbers: counts, rates, thresholds
    def numbers(e):
        return [ln.split("] ", 1)[-1] for ln in e.splitlines() if any(w in ln for w in ("Sampled ", "false positive rate", "log CDF", "coverage", "Sequence length"))]
    assert numbers(err2) == numbers(err)


@pytest.mark.parametrize("use_oq,set_oq", [(False, False), (False, True), (True, True)])
def test_bam_output_equals_the_reference(tmp_path, use_oq, set_oq):
    d, recs, path, n_rg = bam_dataset(tmp_path, use_oq=use_oq, rg_header=True, seed=4002, genome_len=30000, coverage=24, n_per_million=2000, ragged=True,
                                      extra_errors=80)
    args = (["--use-oq"] if use_oq else []) + (["--set-oq"] if set_oq else []) + [path]
    rc, want, err, seed = run_reference(args)
    assert rc == 0 and seed is not None, err[-2000:]
    for env in ({}, {"KBBQ_DEVICE_READER": "0"}):      # the device reader and the host parsers
        rc, got, err2 = run_cli(args, dict(env, KBBQ_SEED=str(seed)))
        assert rc == 0, err2
        a, b = bamutil.bgzf_decompress(got), bamutil.bgzf_decompress(want)
        ta, ra, xa = bamutil.parse(a)
        tb, rb, xb = bamutil.parse(b)
        assert ra == rb and len(xa) == len(xb)
        # (htslib may add an @PG line or not touch the text at all: compare the records and the rest of the header's lines)
        assert [ln for ln in ta.split("\n") if not ln.startswith("@PG")] == [ln for ln in tb.split("\n") if not ln.startswith("@PG")]
        for g, w in zip(xa, xb):
            assert g["raw"] == w["raw"], g["name"]
