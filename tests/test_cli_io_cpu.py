"""Host I/O of the `kbbq` command line (kbbq_amd/csrc/fastq_io.*), exercised without a GPU through the
binary's hidden --io-test helpers: the FASTQ reader (what kseq_read hands to FastqFile, htsiter.cc:49-59),
the read-name rules of CReadData's FASTQ constructor (readutils.cc:64-104) and the BGZF writer behind
FastqFile::write (htsiter.cc:67-86)."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

import common

CLI = os.environ.get("KBBQ_CLI", os.path.join(common.ROOT, "kbbq_amd", "kbbq"))   # KBBQ_CLI: e.g. a sanitizer build


def ref_name_rules(fullname):
    """readutils.cc:74-97 restated in Python (rg = "", second = 2, namedelimiter = "_")."""
    cur = fullname.find("_")
    first = fullname if cur < 0 else fullname[:cur]
    rg = ""
    while rg == "" and cur >= 0:
        fullname = fullname[cur + 1:]
        cur = fullname.find("_")
        if fullname[:3] == "RG:":
            last_colon = fullname.rfind(":", 0, (cur + 1) if cur >= 0 else len(fullname))
            start = last_colon + 1
            rg = fullname[start:] if cur < 0 else fullname[start:start + cur]   # std::string::substr(pos, COUNT = cur)
    if len(first) < 2:
        return None
    tail = first[-2:]
    second = tail == "/2"
    if second or tail == "/1":
        first = first[:-2]
    return rg, second, first


def parse(path, threads=1):
    out = subprocess.run([CLI, "--io-test", "parse", str(path)], capture_output=True, text=True, check=True,
                         env=dict(os.environ, KBBQ_IO_THREADS=str(threads))).stdout
    lines = out.rstrip("\n").split("\n")
    assert lines[-1].startswith("#end")
    return [ln.split("\t") for ln in lines[:-1]], int(lines[-1].split()[1])


def test_cli_binary_is_built():
    assert os.path.exists(CLI), "make -C kbbq_amd/csrc builds kbbq_amd/kbbq"


def test_reader_plain_gzip_multiline_and_comments(tmp_path):
    text = ("@r1 first comment\nACGT\nNNAC\n+\nIIII\n!!#I\n"      # multi-line sequence and quality
            "@r2/2\tcomment after a tab\nAC\n+r2/2 repeated\nI+\n"   # '+' inside the quality; text after '+'
            "@r3\nA\n+\n@\n"                                        # quality line starting with '@'
            "@empty\n\n+\n\n"
            "@last\nACGTACGT\n+\nIIIIIIII")                         # no trailing newline
    p = tmp_path / "a.fq"
    p.write_text(text)
    g = tmp_path / "a.fq.gz"
    with gzip.open(g, "wt") as fh:
        fh.write(text)
    for path, threads in ((p, 1), (g, 1), (p, 3), (g, 3)):      # 3: inflate-ahead thread in front of the parser
        recs, rc = parse(path, threads)
        assert rc == -1
        assert [(r[0], r[1], r[6], r[7]) for r in recs] == [
            ("r1", "first comment", "ACGTNNAC", "IIII!!#I"), ("r2/2", "comment after a tab", "AC", "I+"),
            ("r3", "", "A", "@"), ("empty", "", "", ""), ("last", "", "ACGTACGT", "IIIIIIII")]
        assert [r[4] for r in recs] == ["0", "1", "0", "0", "0"]          # second-in-pair from the /2 suffix


def test_reader_on_bgzf_input_with_the_inflate_pool(tmp_path):
    import bamutil
    rng = np.random.RandomState(5)
    text = "".join("@r%d_RG:Z:g%d extra\n%s\n+\n%s\n" % (i, i % 3, "".join(rng.choice(list("ACGTN"), 151)), "".join(chr(c) for c in rng.randint(35, 74, 151)))
                   for i in range(4000))
    plain, bg = tmp_path / "a.fq", tmp_path / "a.fq.bgz"
    plain.write_text(text)
    bg.write_bytes(bamutil.bgzf_compress(text.encode(), ragged_seed=3))
    want = parse(plain)
    for threads in ("1", "6"):
        out = subprocess.run([CLI, "--io-test", "parse", str(bg)], capture_output=True, text=True, check=True,
                             env=dict(os.environ, KBBQ_IO_THREADS=threads)).stdout
        lines = out.rstrip("\n").split("\n")
        assert ([ln.split("\t") for ln in lines[:-1]], int(lines[-1].split()[1])) == want


def test_reader_reports_truncated_quality(tmp_path):
    p = tmp_path / "t.fq"
    p.write_text("@a\nACGT\n+\nIIII\n@b\nACGT\n+\nII\n")
    recs, rc = parse(p)
    assert len(recs) == 1 and rc == -2      # kseq_read's -2; the reference's pass loops simply end there


def test_read_name_rules_follow_the_reference(tmp_path):
    names = ["r1", "r1/1", "r1/2", "ab_RG:Z:grpA", "ab_RG:Z:grpA_rest", "ab_x_RG:g1_y_z", "ab/2_RG:lane1", "r9_foo_bar",
             "xy_RG:", "xy_RG:a:b:c_d", "q1_RG:Z:grpA_x/1", "zz_RG:Z:looooooooooong_tail_more", "a1_RG:one_RG:two", "ab__RG:x"]
    p = tmp_path / "n.fq"
    p.write_text("".join("@%s\nACGT\n+\nIIII\n" % n for n in names))
    recs, rc = parse(p)
    seen = {}
    for n, r in zip(names, recs):
        want = ref_name_rules(n)
        assert want is not None
        assert (r[2], r[4] == "1", r[5]) == want, n
        idx = seen.setdefault(want[0], len(seen))      # dense index in order of first appearance (readutils.cc:100-103)
        assert int(r[3]) == idx
    # names shorter than two characters before the first '_' make the reference throw (readutils.cc:90)
    p.write_text("@a\nACGT\n+\nIIII\n")
    recs, rc = parse(p)
    assert recs[0][2] == "!"
    assert ref_name_rules("a") is None


def test_bgzf_writer_round_trip_and_block_structure(tmp_path):
    rng = np.random.RandomState(0)
    fastq_like = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), 150).tolist()), bytes(rng.randint(35, 74, 150).tolist()))
                          for i in range(3000))
    for payload in (b"", b"x", fastq_like, bytes(rng.randint(0, 256, 300000).tolist())):   # incompressible data too
        out = subprocess.run([CLI, "--io-test", "bgzf"], input=payload, capture_output=True, check=True).stdout
        assert gzip.decompress(out) == payload
        # walk the blocks: gzip member with the 'BC' extra field, BSIZE consistent, <= 64 KiB, EOF block last
        pos, sizes = 0, []
        while pos < len(out):
            assert out[pos:pos + 4] == b"\x1f\x8b\x08\x04" and out[pos + 12:pos + 16] == b"BC\x02\x00"
            bsize = struct.unpack("<H", out[pos + 16:pos + 18])[0] + 1
            assert bsize <= 65536
            isize = struct.unpack("<I", out[pos + bsize - 4:pos + bsize])[0]
            assert isize <= 65536
            sizes.append(isize)
            pos += bsize
        assert pos == len(out) and sizes[-1] == 0 and out[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
        assert sum(sizes) == len(payload)
        # the pool only changes who deflates a block, not the stream
        threaded = subprocess.run([CLI, "--io-test", "bgzf", "5"], input=payload, capture_output=True, check=True).stdout
        assert threaded == out


def test_bgzf_level_knob_changes_size_not_content(tmp_path):
    rng = np.random.RandomState(1)
    payload = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), 150).tolist()), bytes(rng.randint(35, 74, 150).tolist()))
                       for i in range(3000))
    sizes = {}
    for level in ("", "1", "9", "bogus"):
        out = subprocess.run([CLI, "--io-test", "bgzf", "3"], input=payload, capture_output=True, check=True,
                             env=dict(os.environ, KBBQ_BGZF_LEVEL=level)).stdout
        assert gzip.decompress(out) == payload
        sizes[level] = len(out)
    assert sizes["1"] > sizes[""] >= sizes["9"] and sizes["bogus"] == sizes[""]     # default = zlib's default level, like the reference
