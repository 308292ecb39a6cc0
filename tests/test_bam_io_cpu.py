"""The BAM codec behind the `kbbq` command line (kbbq_amd/csrc/bam_io.*), without a GPU, through the binary's
--io-test helpers.  Expected values come from tests/bamutil.py, an independent Python writer/reader, and from
the rules the reference applies to a record (readutils.hh:30-42, readutils.cc:13-61, htsiter.cc:11-34)."""
import os
import subprocess

import numpy as np
import pytest

import bamutil
import common
from test_cli_io_cpu import CLI


def some_records(seed=1, n=60):
    rng = np.random.RandomState(seed)
    recs = []
    for r in range(n):
        l = int(rng.choice([0, 1, 2, 31, 32, 33, 100, 151])) if r % 7 == 3 else int(rng.randint(20, 160))
        alphabet = "ACGT" if r % 5 else "ACGTNMR="
        seq = "".join(rng.choice(list(alphabet), l))
        qual = rng.randint(2, 42, l)
        flag = (16 if rng.rand() < 0.5 else 0) | (128 if r & 1 else 64) | 1 | 4
        rg = ["grpA", "grpB", "lane:3"][int(rng.randint(0, 3))] if r else "grpB"
        tags = [("NM", "C", 3), ("XA", "A", "q"), ("XS", "s", -77), ("XI", "I", 4000000000), ("XF", "f", 1.5), ("XH", "H", "1AE3"),
                ("XB", "BS", [1, 2, 65535]), ("XC", "Bc", []), ("XZ", "Z", "RG:Z:decoy")][: int(rng.randint(0, 10))]
        tags.append(("RG", "Z", rg))
        if r % 3 == 0:
            tags.append(("OQ", "Z", "".join(chr(33 + int(q)) for q in rng.randint(2, 42, l))))
        if r % 4 == 0:
            tags.append(("XT", "i", -5))
        recs.append(dict(name="read%d" % r, flag=flag, seq=seq, qual=qual, tags=tags))
    return recs


def write_bam(path, recs, refs=(("chr1", 1000), ("chrUn_x", 234567)), text="@HD\tVN:1.6\tSO:unsorted\n@RG\tID:grpA\n", ragged=None):
    stream = bamutil.header(text, refs) + b"".join(bamutil.record(r["name"], r["flag"], r["seq"], r["qual"], r["tags"]) for r in recs)
    path.write_bytes(bamutil.bgzf_compress(stream, ragged_seed=ragged))
    return stream


THREADS = {"KBBQ_IO_THREADS": "1"}      # set per test: 1 = zlib's gzread on the caller, > 1 = pool of inflaters over the BGZF blocks
MODE = {"what": "bam"}                  # set per test: "bam" = the serial reader, "bam-fast" = the block-parallel parser (BamChunkParser)


def io_bam(path, *more):
    p = subprocess.run([CLI, "--io-test", MODE["what"], str(path)] + list(more), capture_output=True, text=True, env=dict(os.environ, **THREADS))
    lines = p.stdout.rstrip("\n").split("\n")
    return lines[0], [ln.split("\t") for ln in lines[1:-1]], int(lines[-1].split()[1]), p.stderr


@pytest.fixture(params=[(1, "bam"), (4, "bam"), (4, "bam-fast")], autouse=True, ids=["serial-1", "serial-4", "pool-4"])
def io_threads(request):
    THREADS["KBBQ_IO_THREADS"] = str(request.param[0])
    MODE["what"] = request.param[1]
    yield request.param[0]
    THREADS["KBBQ_IO_THREADS"] = "1"
    MODE["what"] = "bam"


@pytest.mark.parametrize("ragged", [None, 7])
def test_reader_hands_the_passes_what_the_reference_would(tmp_path, ragged):
    recs = some_records()
    p = tmp_path / "a.bam"
    write_bam(p, recs, ragged=ragged)
    head, rows, rc, _ = io_bam(p)
    assert head == "#text %d genome %d refs 2" % (len("@HD\tVN:1.6\tSO:unsorted\n@RG\tID:grpA\n"), 1000 + 234567)   # kbbq.cc:204-207
    assert rc == -1 and len(rows) == len(recs)
    seen = {}
    for r, row in zip(recs, rows):
        seq, qual = bamutil.as_sequenced(r["seq"], r["qual"], r["flag"])
        rg = [t for t in r["tags"] if t[0] == "RG"][0][2]
        assert row[0] == r["name"] and int(row[1]) == r["flag"]
        assert row[2] == rg and int(row[3]) == seen.setdefault(rg, len(seen))      # dense index by first appearance
        assert int(row[4]) == (1 if r["flag"] & 128 else 0)
        assert row[5] == seq
        assert row[6] == "".join(chr(33 + int(q)) for q in qual)


def test_use_oq_takes_qualities_from_the_tag(tmp_path):
    recs = [r for r in some_records(seed=2) if any(t[0] == "OQ" for t in r["tags"])]
    p = tmp_path / "a.bam"
    write_bam(p, recs)
    _, rows, rc, _ = io_bam(p, "use-oq")
    assert rc == -1
    for r, row in zip(recs, rows):
        oq = [t for t in r["tags"] if t[0] == "OQ"][0][2]
        assert row[6] == (oq[::-1] if r["flag"] & 16 else oq)                     # readutils.cc:16-39
    # a record without OQ: the reference's message, then its exception
    recs[3]["tags"] = [t for t in recs[3]["tags"] if t[0] != "OQ"]
    write_bam(p, recs)
    _, rows, rc, err = io_bam(p, "use-oq")
    assert rc == -100 and len(rows) == 3
    assert "--use-oq was specified but unable to read OQ tag on read " + recs[3]["name"] in err and "OQ not found." in err


def test_missing_or_corrupt_read_group(tmp_path):
    recs = some_records(seed=3, n=6)
    recs[2]["tags"] = [t for t in recs[2]["tags"] if t[0] != "RG"]
    p = tmp_path / "a.bam"
    write_bam(p, recs)
    _, rows, rc, err = io_bam(p)
    assert rc == -100 and len(rows) == 2
    assert "Unable to read RG tag on read read2" in err and "Every read in the BAM must have an RG tag" in err
    # RG of a non-string type, and an aux area that stops in the middle of a value
    recs = some_records(seed=3, n=6)
    recs[1]["tags"] = [("RG", "i", 5)]
    write_bam(p, recs)
    _, rows, rc, err = io_bam(p)
    assert rc == -100 and len(rows) == 1 and "Tag data is corrupt" in err
    recs = some_records(seed=3, n=6)
    stream = bamutil.header("", []) + bamutil.record("r0", 4, "ACGT", [30] * 4, [("RG", "Z", "g")])[:-1] + b"x"   # no terminating NUL
    p.write_bytes(bamutil.bgzf_compress(stream))
    _, rows, rc, err = io_bam(p)
    assert rc == -100 and "Tag data is corrupt" in err


def test_truncated_and_foreign_files(tmp_path):
    recs = some_records(seed=4, n=10)
    p = tmp_path / "a.bam"
    stream = write_bam(p, recs)
    p.write_bytes(bamutil.bgzf_compress(stream[:-9]))
    _, rows, rc, _ = io_bam(p)
    assert rc == -2 and len(rows) == 9                 # sam_read1 reports a truncated record as an error
    p.write_bytes(bamutil.bgzf_compress(b"BAM\2" + stream[4:]))
    assert subprocess.run([CLI, "--io-test", MODE["what"], str(p)], capture_output=True, env=dict(os.environ, **THREADS)).returncode == 2
    # a block whose checksum does not match its data, and a file cut in the middle of a block
    good = bamutil.bgzf_compress(stream, block=4000)
    bad = bytearray(good)
    bad[len(good) // 2] ^= 0x55
    for blob in (bytes(bad), good[:len(good) // 2]):
        p.write_bytes(blob)
        run = subprocess.run([CLI, "--io-test", MODE["what"], str(p)], capture_output=True, text=True, env=dict(os.environ, **THREADS))
        lines = run.stdout.rstrip("\n").split("\n")
        # never a clean end of file: either the header already fails (zlib reads ahead) or the records stop with an error
        assert run.returncode == 2 or (lines[-1].startswith("#end") and int(lines[-1].split()[1]) < -1 and len(lines) - 2 < 10)


def test_writer_round_trip_and_set_oq(tmp_path):
    recs = some_records(seed=5)
    p = tmp_path / "a.bam"
    stream = write_bam(p, recs, ragged=11)
    out = subprocess.run([CLI, "--io-test", "bamcopy", str(p)], capture_output=True, check=True, env=dict(os.environ, **THREADS)).stdout
    assert out[-28:] == bamutil.BGZF_EOF
    assert bamutil.bgzf_decompress(out) == stream      # header, references and every record byte for byte
    out = subprocess.run([CLI, "--io-test", "bamcopy", str(p), "set-oq"], capture_output=True, check=True).stdout
    text, refs, got = bamutil.parse(bamutil.bgzf_decompress(out))
    assert refs == [("chr1", 1000), ("chrUn_x", 234567)] and len(got) == len(recs)
    for r, g in zip(recs, got):
        oq = "".join(chr(33 + int(q)) for q in r["qual"])          # the record's qualities as stored (htsiter.cc:12-17)
        tags = list(r["tags"])
        if any(t[0] == "OQ" for t in tags):
            tags = [("OQ", "Z", oq) if t[0] == "OQ" else t for t in tags]    # replaced where it stands (htslib >= 1.10)
        else:
            tags.append(("OQ", "Z", oq))                                     # appended
        assert g["aux"] == bamutil.aux_bytes(tags), r["name"]
        assert g["seq"] == r["seq"] and list(g["qual"]) == list(r["qual"]) and g["flag"] == r["flag"]
    # an OQ tag that is not a string cannot be updated
    recs[0]["tags"] = [("RG", "Z", "g"), ("OQ", "i", 1)]
    write_bam(p, recs)
    assert subprocess.run([CLI, "--io-test", "bamcopy", str(p), "set-oq"], capture_output=True).returncode == 3


def test_pool_parser_equals_the_serial_reader_across_chunks(tmp_path):
    """200 000 records (57 MB of alignment blocks: two 32 MB chunks, ~30 pieces): the block-parallel parser prints what
    the serial reader prints, record by record, with plain and with reversed records."""
    if MODE["what"] != "bam-fast":
        pytest.skip("compares the two modes itself")
    import sys
    p = tmp_path / "big.bam"
    subprocess.run([sys.executable, os.path.join(common.ROOT, "tools", "make_bam.py"), str(p), "1000000", "30"], check=True, capture_output=True)
    env = dict(os.environ, KBBQ_IO_THREADS="4")
    serial = subprocess.run([CLI, "--io-test", "bam", str(p)], capture_output=True, check=True, env=env).stdout
    for threads in ("1", "7"):
        pool = subprocess.run([CLI, "--io-test", "bam-fast", str(p), "", threads], capture_output=True, check=True, env=env).stdout
        assert pool == serial
    assert serial.count(b"\n") == 200002
