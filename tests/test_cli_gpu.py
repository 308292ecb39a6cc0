"""The `kbbq` command line (kbbq_amd/csrc/kbbq_cli.cc) end to end on the GPU: FASTQ(.gz) in, recalibrated BGZF
FASTQ on stdout, against the oracle run on the same records (kbbq.cc:205-457 for FASTQ input)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import common
from test_cli_io_cpu import CLI, ref_name_rules

pytestmark = pytest.mark.gpu


def write_fastq(path, d, names, comments=None):
    off = d["off"].astype(np.int64)
    seq, qual = d["seq"].tobytes(), (d["qual"] + 33).astype(np.uint8).tobytes()
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as fh:
        for r, name in enumerate(names):
            head = name if not comments or not comments[r] else name + " " + comments[r]
            fh.write(b"@" + head.encode() + b"\n" + seq[off[r]:off[r + 1]] + b"\n+\n" + qual[off[r]:off[r + 1]] + b"\n")


def read_fastq_text(blob):
    lines = blob.decode().split("\n")
    assert lines[-1] == ""
    return [tuple(lines[i:i + 4]) for i in range(0, len(lines) - 1, 4)]


def named_dataset(**kw):
    d = common.make_dataset(**kw)
    n = len(d["off"]) - 1
    groups = ["", "laneA", "laneB"]
    names, rg_index, seen = [], [], {}
    for r in range(n):
        g = groups[(r * 7 // 3) % 3] if r > 4 else ""
        name = "read%d%s" % (r // 2, "/2" if r & 1 else "/1")
        if g:
            name += "_x_RG:Z:%s" % g
        rg, second, _ = ref_name_rules(name)
        rg_index.append(seen.setdefault(rg, len(seen)))
        assert second == bool(r & 1)
        names.append(name)
    d["rg"] = np.array(rg_index, dtype=np.int32)
    d["second"] = (np.arange(n) & 1).astype(np.uint8)
    return d, names, len(seen)


def run_cli(args, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    p = subprocess.run([CLI] + [str(a) for a in args], capture_output=True, env=env, timeout=600)
    return p.returncode, p.stdout, p.stderr.decode()


def test_cli_recalibrates_fastq_like_the_reference_pipeline(tmp_path):
    d, names, n_rg = named_dataset(seed=2024, genome_len=25000, coverage=24, n_per_million=3000, ragged=True, mid_reads=100,
                                   extra_errors=150)
    comments = ["c%d extra" % r if r % 5 == 0 else "" for r in range(len(names))]
    fq = tmp_path / "in.fq.gz"
    write_fastq(fq, d, names, comments)
    total = int(d["off"][-1])
    coverage = total // d["genome_len"]          # what the CLI estimates (kbbq.cc:229-250)
    rc, out, err = run_cli(["-g", d["genome_len"], fq], {"KBBQ_SEED": "777"})
    assert rc == 0, err
    for line in ("Estimating alpha.", "Estimating coverage.", "Total Sequence length: %d" % total, "Estimated coverage: %d" % coverage,
                 "Seed: 777", "Sampled ", "Approximate false positive rate:", "log CDF: [", "Finding trusted kmers", "Finding errors",
                 "Training model", "Recalibrating file"):
        assert line in err, line
    ora = common.run_oracle(dict(d, coverage=coverage), seed=777, n_rg=n_rg)
    assert " Sampled %d valid kmers." % ora["sampled_inserted"] in err
    recs = read_fastq_text(gzip.decompress(out))
    assert len(recs) == len(names)
    off = d["off"].astype(np.int64)
    seq = d["seq"].tobytes().decode()
    want_q = (ora["recal"] + 33).astype(np.uint8).tobytes().decode()
    for r, (h, s, plus, q) in enumerate(recs):
        assert h == "@" + names[r] and s == seq[off[r]:off[r + 1]] and plus == "+" + comments[r]   # htsiter.cc:75-86
        assert q == want_q[off[r]:off[r + 1]], "read %d" % r
    assert (ora["recal"] != d["qual"]).any()


def test_cli_keeps_the_case_of_a_soft_masked_fastq(tmp_path):
    """Lower-case bases: written back as they came, and recalibrated as the reference does (its raw-character
    comparisons included) -- resident and streaming modes alike."""
    kw = dict(seed=608, genome_len=30000, coverage=25, clusters=400, extra_errors=100)
    d, names, n_rg = named_dataset(**kw)
    folded = d["seq"]
    d["seq"] = common.make_softmasked_dataset(frac=0.5, **kw)["seq"]
    assert (d["seq"] >= ord("a")).sum() > 1000
    fq = tmp_path / "soft.fq.gz"
    write_fastq(fq, d, names)
    coverage = int(d["off"][-1]) // d["genome_len"]
    ora = common.run_oracle(dict(d, coverage=coverage), seed=99, n_rg=n_rg)
    plain = common.run_oracle(dict(d, seq=folded, coverage=coverage), seed=99, n_rg=n_rg)
    assert (ora["recal"] != plain["recal"]).any()          # the case of the bases does change the reference's answer
    want_q = (ora["recal"] + 33).astype(np.uint8).tobytes().decode()
    seq = d["seq"].tobytes().decode()
    off = d["off"].astype(np.int64)
    for env in ({"KBBQ_SEED": "99"}, {"KBBQ_SEED": "99", "KBBQ_RESIDENT": "0"}):
        rc, out, err = run_cli(["-g", d["genome_len"], fq], env)
        assert rc == 0, err
        recs = read_fastq_text(gzip.decompress(out))
        assert len(recs) == len(names)
        for r, (h, s_, plus, q) in enumerate(recs):
            assert s_ == seq[off[r]:off[r + 1]] and q == want_q[off[r]:off[r + 1]], "read %d" % r


def test_cli_takes_reads_longer_than_the_staged_kernels_hold(tmp_path):
    """600-2500-base reads (the reference has no length limit: covariateutils.cc:102-116, readutils.cc:238): windowed
    kernels and the run-time-sized walk, resident and streaming."""
    d, names, n_rg = named_dataset(seed=77, genome_len=50000, coverage=24, read_len=2500, ragged=True, ragged_min=600, n_per_million=1500,
                                   extra_errors=100, clusters=40)
    fq = tmp_path / "long.fq.gz"
    write_fastq(fq, d, names)
    coverage = int(d["off"][-1]) // d["genome_len"]
    ora = common.run_oracle(dict(d, coverage=coverage), seed=4, n_rg=n_rg)
    want = (ora["recal"] + 33).astype(np.uint8).tobytes().decode()
    for env in ({"KBBQ_SEED": "4"}, {"KBBQ_SEED": "4", "KBBQ_RESIDENT": "0"}):
        rc, out, err = run_cli(["-g", d["genome_len"], fq], env)
        assert rc == 0, err
        assert "".join(q for _, _, _, q in read_fastq_text(gzip.decompress(out))) == want


def test_cli_streaming_and_resident_modes_agree(tmp_path):
    """Default: the packed reads stay in GPU memory between the passes; KBBQ_RESIDENT=0: every pass decodes the
    file again like the reference.  Same bytes either way, also with a single compression thread."""
    d, names, n_rg = named_dataset(seed=11, genome_len=20000, coverage=20, ragged=True, short_reads=20)
    fq = tmp_path / "in.fq.gz"
    write_fastq(fq, d, names)
    rc, out_res, err_res = run_cli(["-g", d["genome_len"], fq], {"KBBQ_SEED": "5"})
    assert rc == 0 and "Reads are resident on the GPU" in err_res
    assert "their records in host memory" in err_res          # nothing is decoded twice
    rc, out_nocache, err_nocache = run_cli(["-g", d["genome_len"], fq], {"KBBQ_SEED": "5", "KBBQ_HOST_CACHE_MB": "0"})
    assert rc == 0 and "resident on the GPU" in err_nocache and "host memory" not in err_nocache     # second decode for the output
    rc, out_str, err_str = run_cli(["-g", d["genome_len"], "-t", 1, fq], {"KBBQ_SEED": "5", "KBBQ_RESIDENT": "0"})
    assert rc == 0 and "resident" not in err_str
    assert out_res == out_str == out_nocache
    ora = common.run_oracle(dict(d, coverage=int(d["off"][-1]) // d["genome_len"]), seed=5, n_rg=n_rg)
    got = "".join(q for _, _, _, q in read_fastq_text(gzip.decompress(out_res)))
    assert got == (ora["recal"] + 33).astype(np.uint8).tobytes().decode()


def test_cli_options_k_alpha_coverage_and_plain_input(tmp_path):
    d, names, n_rg = named_dataset(seed=5, genome_len=15000, coverage=40, read_len=100)
    fq = tmp_path / "in.fq"
    write_fastq(fq, d, names)
    rc, out, err = run_cli(["-k", 21, "-a", "0.05", "-c", 40, "--genomelen", d["genome_len"], "-t", 4, fq], {"KBBQ_SEED": "31337"})
    assert rc == 0, err
    assert "Estimating" not in err and "Sampling kmers at rate 0.05" in err
    ora = common.run_oracle(dict(d, coverage=40), k=21, seed=31337, alpha="0.05", n_rg=n_rg)
    recs = read_fastq_text(gzip.decompress(out))
    got = "".join(q for _, _, _, q in recs)
    assert got == (ora["recal"] + 33).astype(np.uint8).tobytes().decode()


def test_cli_fixed_mode_uses_the_corrected_file_as_truth(tmp_path):
    d, names, n_rg = named_dataset(seed=8, genome_len=12000, coverage=20, read_len=100)
    truth = dict(d)
    rng = np.random.RandomState(3)
    seq = d["seq"].copy()
    flip = rng.rand(len(seq)) < 0.01
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq[flip] = acgt[rng.randint(0, 4, size=int(flip.sum()))]
    truth["seq"] = seq
    fq, fixed = tmp_path / "in.fq.gz", tmp_path / "fixed.fq"
    write_fastq(fq, d, names)
    write_fastq(fixed, truth, names)
    rc, out, err = run_cli(["--fixed", fixed, fq])
    assert rc == 0, err
    assert "Using fixed file to find errors." in err
    errors = (d["seq"] != seq).astype(np.uint8)
    alpha_ld, cov, approx = common.plan_parameters(d["genome_len"], 20, None)
    o = common.pyoracle.Oracle(32, alpha_ld, 1, approx)
    o.tally(d["seq"], d["qual"], d["off"], d["rg"], d["second"], errors)       # consume_read, kbbq.cc:371-377
    o.train()
    want = o.recalibrate(d["seq"], d["qual"], d["off"], d["rg"], d["second"])
    got = "".join(q for _, _, _, q in read_fastq_text(gzip.decompress(out)))
    assert got == (want + 33).astype(np.uint8).tobytes().decode()


def test_cli_error_paths(tmp_path):
    d, names, _ = named_dataset(seed=5, genome_len=3000, coverage=10, read_len=100)
    fq = tmp_path / "in.fq"
    write_fastq(fq, d, names)
    rc, out, err = run_cli([fq])
    assert rc != 0 and "--genomelen must be specified" in err and out == b""
    rc, out, err = run_cli(["-g", 10 ** 9, fq])
    assert rc != 0 and "estimated coverage is 0" in err
    rc, out, err = run_cli(["-k", 33, "-g", 3000, fq])
    assert rc != 0 and "k must be <= 32" in err
    rc, out, err = run_cli(["-g", 3000, tmp_path / "missing.fq"])
    assert rc != 0 and "Error opening file" in err
    # a genome length far too small for the data: the sampled filter overflows its false-positive gate (kbbq.cc:311-316)
    rc, out, err = run_cli(["-g", 30, "-c", 10, fq], {"KBBQ_SEED": "1"})
    assert rc != 0 and "false positive rate is too high" in err


# ------------------------------------------------------------------ BAM (BASELINE.json configs[3]) ----
import bamutil  # noqa: E402


RG_HEADER = "@HD\tVN:1.6\tSO:unsorted\n@RG\tID:lane2\tSM:s\n@RG\tID:unused\n@RG\tSM:s\tID:lane:3\n@RG\tID:lane1\n@PG\tID:x\n"


def bam_dataset(tmp_path, use_oq=False, rg_header=False, **kw):
    """Unaligned BAM with RG:Z (and OQ:Z) tags, about half of the records reverse-flagged: those store the
    reverse complement of what was sequenced, qualities reversed (readutils.cc:36-39, htsiter.cc:27-31)."""
    d = common.make_dataset(**kw)
    n = len(d["off"]) - 1
    rng = np.random.RandomState(kw.get("seed", 0))
    off = d["off"].astype(np.int64)
    groups = ["lane1", "lane2", "lane:3"]
    seen, rg_index, recs = {}, [], []
    seq_text = d["seq"].tobytes().decode()
    for r in range(n):
        s, q = seq_text[off[r]:off[r + 1]], d["qual"][off[r]:off[r + 1]]
        flag = 1 | 4 | (128 if r & 1 else 64) | (16 if rng.rand() < 0.5 else 0)
        rg = groups[(r // 3) % 3]
        rg_index.append(seen.setdefault(rg, len(seen)))
        stored_s, stored_q = (bamutil.as_sequenced(s, q, flag) if flag & 16 else (s, q))     # the mapping is an involution on ACGTN
        tags = [("NM", "C", r % 200), ("RG", "Z", rg)]
        if use_oq:
            # the real qualities live in OQ; the record's own are something else entirely
            tags.append(("OQ", "Z", "".join(chr(33 + int(x)) for x in stored_q)))
            stored_q = np.full(len(s), 11, dtype=np.uint8)
        elif r % 4 == 0:
            tags.insert(1, ("OQ", "Z", "#" * len(s)))       # a stale OQ that --set-oq must overwrite in place
        if r % 5 == 0:
            tags.append(("XB", "BS", [r & 0xFFFF, 7]))
        recs.append(dict(name="read%d" % (r // 2), flag=flag, seq=stored_s, qual=stored_q, tags=tags))
    d["rg"] = np.array(rg_index, dtype=np.int32)
    d["second"] = (np.arange(n) & 1).astype(np.uint8)
    refs = [("chr1", d["genome_len"] - 1000), ("chr2", 1000)]
    stream = bamutil.header(RG_HEADER if rg_header else "@HD\tVN:1.6\tSO:unsorted\n", refs) + b"".join(
        bamutil.record(r["name"], r["flag"], r["seq"], r["qual"], r["tags"]) for r in recs)
    path = tmp_path / "in.bam"
    path.write_bytes(bamutil.bgzf_compress(stream, ragged_seed=5))
    return d, recs, path, len(seen)


@pytest.mark.parametrize("use_oq,set_oq", [(False, False), (False, True), (True, True)])
def test_cli_recalibrates_bam(tmp_path, use_oq, set_oq):
    d, recs, path, n_rg = bam_dataset(tmp_path, use_oq=use_oq, seed=77, genome_len=25000, coverage=24, n_per_million=3000, ragged=True,
                                      extra_errors=100)
    args = (["--use-oq"] if use_oq else []) + (["--set-oq"] if set_oq else []) + [path]
    rc, out, err = run_cli(args, {"KBBQ_SEED": "4242"})
    assert rc == 0, err
    total = int(d["off"][-1])
    coverage = total // d["genome_len"]
    for line in ("Estimating genome length", "Genome length is %d bp." % d["genome_len"], "Estimated coverage: %d" % coverage):   # kbbq.cc:196-216
        assert line in err, line
    ora = common.run_oracle(dict(d, coverage=coverage), seed=4242, n_rg=n_rg)
    text, refs, got = bamutil.parse(bamutil.bgzf_decompress(out))
    assert text == "@HD\tVN:1.6\tSO:unsorted\n" and refs == [("chr1", d["genome_len"] - 1000), ("chr2", 1000)]
    assert len(got) == len(recs)
    off = d["off"].astype(np.int64)
    changed = 0
    for r, (src, g) in enumerate(zip(recs, got)):
        want = ora["recal"][off[r]:off[r + 1]]
        if src["flag"] & 16:
            want = want[::-1]                                                  # htsiter.cc:27-31
        assert np.array_equal(g["qual"], want), "read %d" % r
        assert g["name"] == src["name"] and g["flag"] == src["flag"] and g["seq"] == src["seq"]
        tags = list(src["tags"])
        if set_oq:
            oq = "".join(chr(33 + int(x)) for x in src["qual"])                # the record's stored qualities (htsiter.cc:12-17)
            tags = [("OQ", "Z", oq) if t[0] == "OQ" else t for t in tags] if any(t[0] == "OQ" for t in tags) else tags + [("OQ", "Z", oq)]
        assert g["aux"] == bamutil.aux_bytes(tags), "read %d" % r
        changed += int((g["qual"] != src["qual"]).sum())
    assert changed > 0


def test_cli_bam_fixed_mode_and_errors(tmp_path):
    d, recs, path, n_rg = bam_dataset(tmp_path, seed=78, genome_len=12000, coverage=20, read_len=100)
    # the corrected file: same records, about 1 % of the stored bases changed
    rng = np.random.RandomState(9)
    fixed_recs, errors = [], []
    for r in recs:
        s = list(r["seq"])
        flip = rng.rand(len(s)) < 0.01
        for i in np.nonzero(flip)[0]:
            s[i] = "ACGT"[("ACGT".find(s[i]) + 1) % 4] if s[i] in "ACGT" else "A"
        fixed_recs.append(dict(r, seq="".join(s)))
        errors.append(flip[::-1] if r["flag"] & 16 else flip)                  # in sequencing orientation
    fixed = tmp_path / "fixed.bam"
    fixed.write_bytes(bamutil.bgzf_compress(bamutil.header("", []) + b"".join(
        bamutil.record(r["name"], r["flag"], r["seq"], r["qual"], r["tags"]) for r in fixed_recs)))
    rc, out, err = run_cli(["--fixed", fixed, path])
    assert rc == 0, err
    alpha_ld, cov, approx = common.plan_parameters(d["genome_len"], 20, None)
    o = common.pyoracle.Oracle(32, alpha_ld, 1, approx)
    e = np.concatenate(errors).astype(np.uint8)
    o.tally(d["seq"], d["qual"], d["off"], d["rg"], d["second"], e)
    o.train()
    want = o.recalibrate(d["seq"], d["qual"], d["off"], d["rg"], d["second"])
    _, _, got = bamutil.parse(bamutil.bgzf_decompress(out))
    off = d["off"].astype(np.int64)
    for r, g in enumerate(got):
        w = want[off[r]:off[r + 1]]
        assert np.array_equal(g["qual"], w[::-1] if recs[r]["flag"] & 16 else w)
    # no reference lengths in the header and no --genomelen
    bare = tmp_path / "bare.bam"
    bare.write_bytes(bamutil.bgzf_compress(bamutil.header("", []) + bamutil.record("r0", 4, "ACGT" * 10, [30] * 40, [("RG", "Z", "g")])))
    rc, out, err = run_cli([bare])
    assert rc != 0 and "Header does not contain genome information." in err
    # a record without RG
    bare.write_bytes(bamutil.bgzf_compress(bamutil.header("", [("c", 100)]) + bamutil.record("r0", 4, "ACGT" * 10, [30] * 40, [])))
    rc, out, err = run_cli([bare])
    assert rc != 0 and "Unable to read RG tag on read r0" in err and out == b""
    bare.write_bytes(b"CRAM" + b"\0" * 40)
    rc, out, err = run_cli([bare])
    assert rc != 0 and "CRAM" in err


def test_cli_empty_read_ends_sampling_and_coverage_but_not_the_other_passes(tmp_path):
    """next_str() == "" is the reference's end-of-file test in the coverage pass (kbbq.cc:234) and in the sampler
    (htsiter.cc:95), so an empty record stops those two loops; find_trusted_kmers, get_covariatedata and the output
    pass loop on next() >= 0 and see every record (recalibrateutils.cc:21,47,96)."""
    d, names, n_rg = named_dataset(seed=21, genome_len=20000, coverage=30, read_len=100)
    n = len(names)
    cut = n * 2 // 3
    off = d["off"].astype(np.int64)
    seq, qual = d["seq"].tobytes(), (d["qual"] + 33).astype(np.uint8).tobytes()
    fq = tmp_path / "in.fq"
    with open(fq, "wb") as fh:
        for r in range(n):
            if r == cut:
                fh.write(b"@empty/1\n\n+\n\n")
            fh.write(b"@" + names[r].encode() + b"\n" + seq[off[r]:off[r + 1]] + b"\n+\n" + qual[off[r]:off[r + 1]] + b"\n")
    rc, out, err = run_cli(["-g", d["genome_len"], fq], {"KBBQ_SEED": "9"})
    assert rc == 0, err
    assert "resident" not in err                       # the passes do not see the same reads: no shortcut
    total_before = int(off[cut])
    coverage = total_before // d["genome_len"]
    assert "Total Sequence length: %d" % total_before in err and "Estimated coverage: %d" % coverage in err
    # the same thing with the oracle: sample the reads before the empty one, everything else on all reads
    lens = np.diff(off)
    lens2 = np.concatenate([lens[:cut], [0], lens[cut:]])
    off2 = np.concatenate([[0], np.cumsum(lens2)]).astype(np.uint64)
    rg2 = np.concatenate([d["rg"][:cut], [d["rg"][cut - 1] * 0 + ref_rg_index_of_empty(names, cut)], d["rg"][cut:]]).astype(np.int32)
    sec2 = np.concatenate([d["second"][:cut], [0], d["second"][cut:]]).astype(np.uint8)
    alpha_ld, cov, approx = common.plan_parameters(d["genome_len"], coverage, None)
    o = common.pyoracle.Oracle(32, alpha_ld, 9, approx)
    o.sample(d["seq"][:total_before], d["off"][:cut + 1])
    assert " Sampled %d valid kmers." % o.filter_info(0)["inserted"] in err
    o.compute_thresholds()
    o.trusted(d["seq"], d["qual"], off2)
    o.errors(d["seq"], d["qual"], off2, rg2, sec2, tally=True)
    o.train()
    want = o.recalibrate(d["seq"], d["qual"], off2, rg2, sec2)
    recs = read_fastq_text(gzip.decompress(out))
    assert len(recs) == n + 1 and recs[cut] == ("@empty/1", "", "+", "")
    got = "".join(q for _, _, _, q in recs)
    assert got == (want + 33).astype(np.uint8).tobytes().decode()


def ref_rg_index_of_empty(names, cut):
    """Dense read-group index the empty read '@empty/1' gets: its group is "" (no RG field), which the first reads
    of named_dataset already introduced as index 0."""
    assert ref_name_rules(names[0])[0] == "" and cut > 0
    return 0


def test_cli_output_does_not_depend_on_the_batch_size(tmp_path):
    """KBBQ_BATCH_READS (a testing knob) cuts the input into many engine calls: sampler ordinals, resident batches with
    their hint arrays, the two-stream passes, the record cache and the writers all cross batch boundaries; the bytes of
    the decompressed output must stay those of the one-batch run -- FASTQ and BAM, resident and streaming."""
    d, names, n_rg = named_dataset(seed=4040, genome_len=20000, coverage=24, n_per_million=2000, ragged=True, mid_reads=60,
                                   short_reads=10, extra_errors=80)
    fq = tmp_path / "in.fq.gz"
    write_fastq(fq, d, names)
    rc, one, err = run_cli(["-g", d["genome_len"], fq], {"KBBQ_SEED": "99"})
    assert rc == 0, err
    one = gzip.decompress(one)
    for env in ({"KBBQ_BATCH_READS": "257"}, {"KBBQ_BATCH_READS": "1000", "KBBQ_RESIDENT": "0"},
                {"KBBQ_BATCH_READS": "64", "KBBQ_HOST_CACHE_MB": "0"}):
        rc, out, err = run_cli(["-g", d["genome_len"], fq], dict(env, KBBQ_SEED="99"))
        assert rc == 0, err
        assert gzip.decompress(out) == one, env
    db, recs, path, n_rg_b = bam_dataset(tmp_path, seed=78, genome_len=20000, coverage=24, n_per_million=2000, ragged=True, extra_errors=60)
    rc, one_b, err = run_cli(["--set-oq", path], {"KBBQ_SEED": "7"})
    assert rc == 0, err
    one_b = bamutil.bgzf_decompress(one_b)
    for env in ({"KBBQ_BATCH_READS": "300"}, {"KBBQ_BATCH_READS": "777", "KBBQ_RESIDENT": "0"}):
        rc, out, err = run_cli(["--set-oq", path], dict(env, KBBQ_SEED="7"))
        assert rc == 0, err
        assert bamutil.bgzf_decompress(out) == one_b, env


def test_cli_output_into_a_file_is_written_in_ranges(tmp_path):
    """Standard output that is a regular file gets its blocks by pwrite from several threads (kbbq_cli.cc:
    DeviceBgzfWriter::put); the file must hold the bytes a pipe receives, behind whatever the file held before, and a file
    opened for appending or KBBQ_WRITE_THREADS=1 takes the sequential writes."""
    d, names, n_rg = named_dataset(seed=515, genome_len=30000, coverage=30, n_per_million=2000, ragged=True, extra_errors=80)
    fq = tmp_path / "in.fq.gz"
    write_fastq(fq, d, names)
    rc, piped, err = run_cli(["-g", d["genome_len"], fq], {"KBBQ_SEED": "5"})
    assert rc == 0, err
    assert len(piped) > 4 * 65536          # enough for four ranges
    env = dict(os.environ, KBBQ_SEED="5")
    for mode, extra, prefix in (("wb", {}, b""), ("wb", {}, b"what was there before"), ("ab", {}, b"appended to"), ("wb", {"KBBQ_WRITE_THREADS": "1"}, b"")):
        out = tmp_path / "out.fq.gz"
        with open(out, "wb") as f:
            f.write(prefix)
        with open(out, mode) as f:
            if mode == "wb":
                f.write(prefix)
                f.flush()
            p = subprocess.run([CLI, "-g", str(d["genome_len"]), str(fq)], stdout=f, stderr=subprocess.PIPE, env=dict(env, **extra), timeout=600)
        assert p.returncode == 0, p.stderr.decode()
        assert out.read_bytes() == prefix + piped, (mode, extra, prefix)


def test_cli_block_parallel_and_serial_fastq_parse_agree(tmp_path):
    """The first scan parses strictly four-line FASTQ with a pool (fastq_io.h: FastqChunkParser) and starts over with the
    serial reader when the file turns out not to be of that shape: same bytes out either way, for a plain file, a
    gzip-ed one, small batches, and a file whose LAST record is multi-line (the restart happens at the very end)."""
    d, names, n_rg = named_dataset(seed=515, genome_len=20000, coverage=24, n_per_million=2000, ragged=True, mid_reads=60, extra_errors=80)
    comments = ["c%d" % r if r % 4 == 0 else "" for r in range(len(names))]
    fq, gz = tmp_path / "in.fq", tmp_path / "in.fq.gz"
    write_fastq(fq, d, names, comments)
    write_fastq(gz, d, names, comments)
    rc, want, err = run_cli(["-g", d["genome_len"], fq], {"KBBQ_SEED": "12", "KBBQ_SERIAL_PARSE": "1"})
    assert rc == 0, err
    want = gzip.decompress(want)
    for path, env in ((fq, {}), (gz, {}), (gz, {"KBBQ_BATCH_READS": "333"}), (fq, {"KBBQ_HOST_CACHE_MB": "0"})):
        rc, out, err = run_cli(["-g", d["genome_len"], path], dict(env, KBBQ_SEED="12"))
        assert rc == 0, err
        assert gzip.decompress(out) == want, (path, env)
    # the same records, the last one written over two sequence and two quality lines
    text = fq.read_bytes().decode().split("\n")
    assert text[-1] == "" and len(text[-4]) > 3
    s, q = text[-4], text[-2]
    multi = "\n".join(text[:-4] + [s[:2], s[2:], "+", q[:2], q[2:], ""])
    ml = tmp_path / "multi.fq"
    ml.write_text(multi)
    rc, out, err = run_cli(["-g", d["genome_len"], ml], {"KBBQ_SEED": "12"})
    assert rc == 0, err
    assert gzip.decompress(out) == want


@pytest.mark.parametrize("use_oq", [False, True])
def test_cli_block_parallel_and_serial_bam_parse_agree(tmp_path, use_oq):
    """BAM records go through the same pool (bam_io.h: BamChunkParser) in the first scan; the bytes written must be those
    of the serial reader (KBBQ_SERIAL_PARSE=1), with small batches and without the record cache as well."""
    db, recs, path, n_rg = bam_dataset(tmp_path, use_oq=use_oq, seed=616, genome_len=20000, coverage=24, n_per_million=2000, ragged=True, extra_errors=60)
    args = (["--use-oq"] if use_oq else []) + ["--set-oq", path]
    rc, want, err = run_cli(args, {"KBBQ_SEED": "21", "KBBQ_SERIAL_PARSE": "1"})
    assert rc == 0, err
    want = bamutil.bgzf_decompress(want)
    for env in ({}, {"KBBQ_BATCH_READS": "211"}, {"KBBQ_HOST_CACHE_MB": "0"}):
        rc, out, err = run_cli(args, dict(env, KBBQ_SEED="21"))
        assert rc == 0, err
        assert bamutil.bgzf_decompress(out) == want, env


@pytest.mark.parametrize("use_oq,set_oq", [(False, False), (False, True), (True, True), (True, False)])
def test_cli_bam_on_the_device_equals_the_host_parsers(tmp_path, use_oq, set_oq):
    """A BAM whose header names its read groups is read on the GPU (kbbq_bam_reader: inflate, record chain, field decode) and
    pass 4 rewrites the records there; the decompressed output must be the host path's (bam_io.cc: BamChunkParser + the host
    rewrite, KBBQ_DEVICE_READER=0) byte for byte -- with the chunks kept in HBM, with the file read a second time
    (KBBQ_KEEP_TEXT=0), and with chunks of 64 KB of file that cut BGZF blocks and records everywhere."""
    db, recs, path, n_rg = bam_dataset(tmp_path, use_oq=use_oq, rg_header=True, seed=919, genome_len=30000, coverage=24, n_per_million=2000, ragged=True,
                                       extra_errors=60)
    args = (["--use-oq"] if use_oq else []) + (["--set-oq"] if set_oq else []) + [path]
    rc, want, err = run_cli(args, {"KBBQ_SEED": "33", "KBBQ_DEVICE_READER": "0", "KBBQ_TIMING": "1"})
    assert rc == 0, err
    assert "reader on the GPU" not in err
    want = bamutil.bgzf_decompress(want)
    text, refs, got = bamutil.parse(want)
    assert text == RG_HEADER and len(got) == len(recs)
    for env in ({}, {"KBBQ_KEEP_TEXT": "0"}, {"KBBQ_READER_PIECE_KB": "64"}, {"KBBQ_READER_PIECE_KB": "64", "KBBQ_KEEP_TEXT": "0"}):
        rc, out, err = run_cli(args, dict(env, KBBQ_SEED="33", KBBQ_TIMING="1"))
        assert rc == 0, err
        assert "BAM reader on the GPU" in err, err[-2000:]
        assert ("one scan" in err) == ("KBBQ_KEEP_TEXT" not in env)
        assert bamutil.bgzf_decompress(out) == want, env
    # and the oracle on the same reads: the qualities are the reference pipeline's
    total = int(db["off"][-1])
    ora = common.run_oracle(dict(db, coverage=total // db["genome_len"]), seed=33, n_rg=n_rg)
    off = db["off"].astype(np.int64)
    for r, (src, g) in enumerate(zip(recs, got)):
        w = ora["recal"][off[r]:off[r + 1]]
        assert np.array_equal(g["qual"], w[::-1] if src["flag"] & 16 else w), "read %d" % r


def test_cli_bam_shapes_the_device_path_hands_back(tmp_path):
    """A read group the header does not name, and a record without RG: the device reader flags the chunk, the command line
    starts over with the host parsers and behaves as before (output / the reference's message)."""
    db, recs, path, n_rg = bam_dataset(tmp_path, rg_header=True, seed=920, genome_len=12000, coverage=20, read_len=100)
    recs2 = [dict(r, tags=[("RG", "Z", "lane9") if t[0] == "RG" else t for t in r["tags"]]) if i % 50 == 7 else r for i, r in enumerate(recs)]
    refs = [("chr1", db["genome_len"] - 1000), ("chr2", 1000)]
    p2 = tmp_path / "unnamed.bam"
    p2.write_bytes(bamutil.bgzf_compress(bamutil.header(RG_HEADER, refs) + b"".join(
        bamutil.record(r["name"], r["flag"], r["seq"], r["qual"], r["tags"]) for r in recs2), ragged_seed=3))
    rc, out, err = run_cli([p2], {"KBBQ_SEED": "5", "KBBQ_TIMING": "1"})
    assert rc == 0, err
    assert "reader on the GPU" not in err
    rc, want, err = run_cli([p2], {"KBBQ_SEED": "5", "KBBQ_DEVICE_READER": "0"})
    assert rc == 0 and bamutil.bgzf_decompress(out) == bamutil.bgzf_decompress(want)
    recs3 = [dict(r, tags=[t for t in r["tags"] if t[0] != "RG"]) if i == 31 else r for i, r in enumerate(recs)]
    p3 = tmp_path / "norg.bam"
    p3.write_bytes(bamutil.bgzf_compress(bamutil.header(RG_HEADER, refs) + b"".join(
        bamutil.record(r["name"], r["flag"], r["seq"], r["qual"], r["tags"]) for r in recs3)))
    rc, out, err = run_cli([p3], {"KBBQ_SEED": "5"})
    assert rc != 0 and "Unable to read RG tag on read " + recs3[31]["name"] in err


@pytest.mark.parametrize("fmt", ["fastq_rg_names", "bam"])
def test_cli_on_several_devices_writes_the_same_bytes(tmp_path, fmt):
    """KBBQ_DEVICES: passes 1-3 and the model sharded over several engines of one process (contiguous shards, global k-mer
    ordinals, the library's exchange steps between the passes: include/kbbq_exchange.h), output by the first device.  The
    list names the box's one GPU two and three times -- engines that meet through device-to-device copies, since RCCL
    refuses two ranks on one device -- and a list of one runs as before.  Every variant must write the one-device bytes and
    log the same counts."""
    if fmt == "bam":
        d, recs, path, n_rg = bam_dataset(tmp_path, rg_header=True, seed=515, genome_len=40000, coverage=24, n_per_million=2000, ragged=True, extra_errors=80)
        args = ["--set-oq", path]
    else:
        d, names, n_rg = named_dataset(seed=516, genome_len=40000, coverage=24, n_per_million=2000, ragged=True, extra_errors=80)
        path = tmp_path / "in.fq.gz"
        write_fastq(path, d, names, ["" for _ in names])
        args = ["-g", d["genome_len"], path]
    env = {"KBBQ_SEED": "99", "KBBQ_BATCH_READS": "700", "KBBQ_READER_PIECE_KB": "64", "KBBQ_QUAL_DIGEST": "1"}
    rc, want, err1 = run_cli(args, env)
    assert rc == 0, err1
    assert "Passes 1-3 on" not in err1

    def counts(err):
        return [ln.split("] ", 1)[-1] for ln in err.splitlines() if "Sampled " in ln or "trusted_inserted" in ln or "false positive rate" in ln]
    for devs in ("0,0", "0,0,0", "0"):
        rc, out, err = run_cli(args, dict(env, KBBQ_DEVICES=devs))
        assert rc == 0, err
        n = len(devs.split(","))
        assert ("Passes 1-3 on %d devices (in-process copies)" % n in err) == (n > 1), err[-1500:]
        assert out == want, devs
        assert counts(err) == counts(err1), (counts(err), counts(err1))
