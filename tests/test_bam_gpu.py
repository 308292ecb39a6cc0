"""The BAM path on the device (include/kbbq_bgzf.h: kbbq_bam_reader; kbbq_amd/csrc/bam_device.h): the record chain found in
the inflated stream, the records decoded into the engine's read layout, and -- pass 4 -- rewritten around new qualities.

The definition stays the host codec (kbbq_amd/csrc/bam_io.cc): the device batch is compared word for word with what
`kbbq --io-test bam` (BamReader + decode_bam_read, the C++ the command line falls back to) hands to the passes for the
same file, and with tests/bamutil.py's independent reading of the reference's rules (readutils.hh:30-42,
readutils.cc:13-61, htsiter.cc:11-45).  Chunk cuts fall inside BGZF blocks and inside records; 0xFF qualities, IUPAC
codes, every aux type, headers and records longer than a segment, and streams that must go back to the host parser."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

import bamutil
import common  # noqa: F401
from kbbq_amd import _lib, bgzf
from kbbq_amd.reads import ReadBatch
from test_bgzf_gpu import download_batch
from test_bgzf_cpu import bgzf_blocks

pytestmark = pytest.mark.gpu

CLI = os.path.join(common.ROOT, "kbbq_amd", "kbbq")
RG_IDS = ["grpA", "grpB", "lane:3", "never_used", "g"]          # the header's @RG lines, in this order
HEADER_TEXT = "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@RG\tID:%s\tSM:x\n" % i for i in RG_IDS)


def aux_raw(tags):
    """like bamutil.aux_bytes, but Z values may be bytes (an OQ made of 0xFF qualities is not text)"""
    out = b""
    for tag, typ, val in tags:
        if typ in "ZH" and isinstance(val, (bytes, bytearray)):
            out += tag.encode() + typ.encode() + bytes(val) + b"\0"
        else:
            out += bamutil.aux_bytes([(tag, typ, val)])
    return out


def make_records(n, seed, lens=None, oq_every=3, big_quals=True):
    rng = np.random.RandomState(seed)
    recs = []
    for r in range(n):
        if lens is not None:
            l = int(lens[r % len(lens)])
        else:
            l = int(rng.choice([1, 2, 31, 32, 33, 100, 151])) if r % 7 == 3 else int(rng.randint(20, 200))
        alphabet = "ACGT" if r % 5 else "ACGTNMR=KY"
        seq = "".join(rng.choice(list(alphabet), l))
        qual = rng.randint(0, 94, l)
        if big_quals and r % 11 == 0:
            qual[rng.randint(0, l, max(1, l // 10))] = 255            # what "missing" looks like in a BAM
        flag = (16 if rng.rand() < 0.5 else 0) | (128 if r & 1 else 64) | 1 | 4
        rg = "lane:3" if r < 3 else RG_IDS[int(rng.choice([0, 1, 2, 4]))]      # first appearance is not the header's order
        tags = [("NM", "C", 3), ("XA", "A", "q"), ("XS", "s", -77), ("XI", "I", 4000000000), ("XF", "f", 1.5), ("XH", "H", "1AE3"),
                ("XB", "BS", [1, 2, 65535]), ("XC", "Bc", []), ("XZ", "Z", "RG:Z:decoy")][: int(rng.randint(0, 10))]
        if r % 2:
            tags.insert(0, ("RG", "Z", rg))
        else:
            tags.append(("RG", "Z", rg))
        if oq_every and r % oq_every == 0:
            oq = bytes(((rng.randint(0, 94, l) + 33) % 256).astype(np.uint8))
            tags.insert(int(rng.randint(0, len(tags) + 1)), ("OQ", "Z", oq))
        if r % 4 == 0:
            tags.append(("XT", "i", -5))
        rec = dict(name="read%d" % r if r % 9 else "r", flag=flag, seq=seq, qual=qual, tags=tags)
        if r % 3 == 1:      # an aligned record: reference, position, one to four CIGAR operations (op | len << 4), a mate, extra flag bits
            n_ops = int(rng.randint(1, 5))
            rec["aln"] = (int(rng.randint(0, 2)), int(rng.randint(0, 900)), int(rng.randint(0, 61)),
                          [(int(rng.randint(1, 200)) << 4) | int(rng.choice([0, 1, 2, 4])) for _ in range(n_ops)],
                          int(rng.randint(-1, 2)), int(rng.randint(-1, 900)), int(rng.randint(-500, 500)))
            rec["flag"] = (flag & ~4) | int(rng.choice([0, 0x100, 0x800, 0x400, 2]))
        recs.append(rec)
    return recs


def record_bytes(r):
    l = len(r["seq"])
    packed = bytearray((l + 1) // 2)
    for i, ch in enumerate(r["seq"]):
        packed[i >> 1] |= bamutil.CODES[ch] << (4 if i % 2 == 0 else 0)
    # aligned records carry a reference, a position, CIGAR operations and a mate (r["aln"] = (refID, pos, mapq, cigar words,
    # next refID, next pos, tlen)); the others are unaligned
    ref, pos, mapq, cigar, nref, npos, tlen = r.get("aln", (-1, -1, 0, [], -1, -1, 0))
    body = struct.pack("<iiBBHHHIiii", ref, pos, len(r["name"]) + 1, mapq, 4680, len(cigar), r["flag"], l, nref, npos, tlen)
    body += r["name"].encode() + b"\0" + b"".join(struct.pack("<I", c) for c in cigar) + bytes(packed)
    body += bytes(bytearray(int(q) for q in r["qual"])) + aux_raw(r["tags"])
    return struct.pack("<I", len(body)) + body


def bam_file(recs, text=HEADER_TEXT, refs=(("chr1", 1000), ("chrUn_x", 234567)), ragged=5):
    head = bamutil.header(text, refs)
    stream = head + b"".join(record_bytes(r) for r in recs)
    return bamutil.bgzf_compress(stream, ragged_seed=ragged), len(head), len(refs)


def feed(reader, comp, cuts):
    """the file in pieces; yields (info, data of the piece) for every piece"""
    pending = b""
    bounds = [0] + list(cuts) + [len(comp)]
    for a, b in zip(bounds[:-1], bounds[1:]):
        data = pending + comp[a:b]
        info = reader.chunk(data, b == len(comp))
        pending = data[info["consumed"]:]
        yield info
    assert not pending or info["flags"]


def host_rows(path, use_oq):
    """what bam_io.cc hands to the passes: rows of `kbbq --io-test bam` (bytes: qualities above 93 are not text)"""
    p = subprocess.run([CLI, "--io-test", "bam", str(path)] + (["use-oq"] if use_oq else []), capture_output=True, env=dict(os.environ, KBBQ_IO_THREADS="1"))
    assert p.returncode == 0, p.stderr
    lines = p.stdout.rstrip(b"\n").split(b"\n")
    assert lines[-1].startswith(b"#end")
    return [ln.split(b"\t") for ln in lines[1:-1]], int(lines[-1].split()[1])


def expected_batch(rows):
    seq = np.frombuffer(b"".join(r[5] for r in rows), dtype=np.uint8)
    qual = (np.frombuffer(b"".join(r[6] for r in rows), dtype=np.uint8).astype(np.int32) - 33).astype(np.uint8)
    off = np.concatenate([[0], np.cumsum([len(r[5]) for r in rows])]).astype(np.uint64)
    second = np.array([int(r[4]) for r in rows], dtype=np.uint8)
    rg = np.array([int(r[3]) for r in rows], dtype=np.uint16)
    return seq, qual, off, second, rg


def download_rg(d):
    import torch
    from kbbq_amd.engine import device_tensor
    return device_tensor(d.rg, int(d.n_reads) * 2, torch.uint8, 0).cpu().numpy().view(np.uint16).copy()


def check_batches(got, rows):
    at = 0
    for info, dev, rgs in got:
        n = info["n_records"]
        part = rows[at:at + n]
        at += n
        seq, qual, off, second, rg = expected_batch(part)
        hb = ReadBatch(seq, qual, off, rg, second, uniform=False)
        nbw, nmw = len(seq) // 32 + 1, len(seq) // 64 + 1
        assert info["n_bases"] == len(seq)
        assert np.array_equal(dev["bases"][:nbw], hb.bases[:nbw]), "2-bit bases differ"
        assert np.array_equal(dev["nmask"][:nmw], hb.nmask[:nmw]), "N mask differs"
        assert np.array_equal(dev["qual"], qual), "qualities differ"
        assert np.array_equal(dev["flags"], second), "second-in-pair flags differ"
        assert np.array_equal(rgs, rg), "read-group indices differ"
        assert dev["offcase"] is None
        if dev["offsets"] is None:
            assert len(set(len(r[5]) for r in part)) == 1 and dev["read_len"] == len(part[0][5])
        else:
            assert np.array_equal(dev["offsets"], off)
    assert at == len(rows)


@pytest.mark.parametrize("use_oq", [False, True])
@pytest.mark.parametrize("shape", ["ragged", "uniform_150", "long_header", "long_records"])
def test_device_batch_equals_the_host_decode(tmp_path, shape, use_oq):
    if shape == "ragged":
        recs = make_records(5000, seed=1, oq_every=1 if use_oq else 3)
    elif shape == "uniform_150":
        recs = make_records(6000, seed=2, lens=[150], oq_every=1 if use_oq else 0)
    elif shape == "long_header":
        recs = make_records(3000, seed=3, oq_every=1 if use_oq else 2)
    else:
        # records longer than a 32 KB segment: segments in which no record starts, landing points several segments on
        recs = make_records(120, seed=4, lens=[40000, 150, 60001, 33, 150, 150, 32768, 5], oq_every=1 if use_oq else 4, big_quals=False)
    text = HEADER_TEXT
    refs = (("chr1", 1000), ("chrUn_x", 234567))
    if shape == "long_header":
        refs = tuple(("contig_%05d_with_a_long_name" % i, 1000 + i) for i in range(4000))      # ~150 KB of header
        text = HEADER_TEXT + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    comp, head_len, n_ref = bam_file(recs, text=text, refs=refs)
    path = tmp_path / "a.bam"
    path.write_bytes(comp)
    rows, rc = host_rows(path, use_oq)
    assert rc == -1 and len(rows) == len(recs)
    # the host rows agree with the independent reading of the rules (a check of the checker)
    for r, row in list(zip(recs, rows))[:200]:
        seq, _ = bamutil.as_sequenced(r["seq"], r["qual"], r["flag"])
        assert row[5].decode() == seq
    reader = bgzf.BamReader(head_len, n_ref, RG_IDS, use_oq=use_oq)
    cuts = [len(comp) // 3 + 5, 2 * len(comp) // 3 + 11]
    got = []
    for info in feed(reader, comp, cuts):
        assert info["flags"] == 0, info
        if info["n_records"]:
            d = reader.batch()
            got.append((info, download_batch(d), download_rg(d)))
            _lib.check(_lib.lib().kbbq_reads_free(None, ctypes.byref(d)))
    check_batches(got, rows)
    # dense read-group indices were handed out in the order of first appearance
    order = [RG_IDS[i] for i in reader.read_groups()]
    seen = []
    for r in recs:
        g = [t for t in r["tags"] if t[0] == "RG"][0][2]
        if g not in seen:
            seen.append(g)
    assert order == seen
    reader.close()


def rewritten(recs, newq, set_oq):
    """BamFile::recalibrate + sam_write1 (htsiter.cc:11-45) of every record, by the rules"""
    out, at = [], 0
    for r in recs:
        l = len(r["seq"])
        q = newq[at:at + l]
        at += l
        tags = list(r["tags"])
        if set_oq:
            val = bytes(((np.asarray(r["qual"], dtype=np.int64) + 33) % 256).astype(np.uint8))
            idx = [i for i, t in enumerate(tags) if t[0] == "OQ"]
            if idx:
                tags[idx[0]] = ("OQ", "Z", val)
            else:
                tags.append(("OQ", "Z", val))
        out.append(record_bytes(dict(r, qual=q[::-1] if r["flag"] & 16 else q, tags=tags)))
    return b"".join(out)


@pytest.mark.parametrize("use_oq,set_oq", [(False, False), (False, True), (True, True), (True, False)])
def test_rewritten_records_equal_the_reference_rules(use_oq, set_oq):
    """Pass 4: new qualities in the quality field (reversed back for reverse-strand records), --set-oq stores the old ones as
    OQ:Z -- replaced in place where the tag exists, appended where it does not (bam_aux_update_str) -- and block_size
    follows; directly after the scan and again from the chunks kept compressed in HBM (kbbq_bam_reader_select)."""
    import torch
    recs = make_records(4000, seed=21, oq_every=1 if use_oq else 3)
    comp, head_len, n_ref = bam_file(recs)
    reader = bgzf.BamReader(head_len, n_ref, RG_IDS, use_oq=use_oq)
    writer = bgzf.BgzfWriter()
    reader.keep(True)
    rng = np.random.RandomState(5)
    newq = rng.randint(0, 94, sum(len(r["seq"]) for r in recs)).astype(np.uint8)
    cuts = [len(comp) // 4 + 3, len(comp) // 2 + 1, 3 * len(comp) // 4 + 7]
    payloads, counts, at_rec, at_base = [], [], 0, 0
    for info in feed(reader, comp, cuts):
        assert info["flags"] == 0, info
        n = info["n_records"]
        if not n:
            continue
        dq = torch.from_numpy(newq[at_base:at_base + info["n_bases"]].copy()).cuda()
        torch.cuda.synchronize()
        reader.write(writer, dq.data_ptr(), set_oq=set_oq)
        blob, n_payload = writer.collect()
        want = rewritten(recs[at_rec:at_rec + n], newq[at_base:at_base + info["n_bases"]], set_oq)
        got = b"".join(bgzf_blocks(blob))
        assert n_payload == len(want) and got == want
        counts.append((n, info["n_bases"]))
        at_rec += n
        at_base += info["n_bases"]
    assert at_rec == len(recs)
    # the same from the kept chunks: inflated and indexed again on the device, nothing read twice
    n_kept, kept_bytes = reader.kept()
    assert n_kept == len(counts) and 0 < kept_bytes < 2 * len(comp) + n_kept * 8192
    at_rec = at_base = 0
    for i, (n, nb) in enumerate(counts):
        info = reader.select(i)
        assert info["n_records"] == n and info["n_bases"] == nb and info["flags"] == 0
        dq = torch.from_numpy(newq[at_base:at_base + nb].copy()).cuda()
        torch.cuda.synchronize()
        reader.write(writer, dq.data_ptr(), set_oq=set_oq)
        blob, _ = writer.collect()
        assert b"".join(bgzf_blocks(blob)) == rewritten(recs[at_rec:at_rec + n], newq[at_base:at_base + nb], set_oq)
        at_rec += n
        at_base += nb
    reader.close()
    writer.close()


def test_hundreds_of_read_groups_go_through_the_hash_table(tmp_path):
    """More than eight @RG lines: the record kernel finds a record's group through a hash table over the ids instead of
    comparing with every one (merged cohorts carry hundreds).  Ids that are prefixes of each other, an id listed twice, one
    group that never occurs; dense indices by first appearance as always."""
    ids = ["s%d.lane%d" % (i // 4, i % 4) for i in range(300)] + ["s1", "s1.lane", "s1.lane1", "never"]
    rng = np.random.RandomState(77)
    used = [ids[int(i)] for i in rng.choice(len(ids) - 1, 40, replace=False)]
    recs = make_records(3000, seed=55, oq_every=0)
    for i, r in enumerate(recs):
        g = used[int(rng.randint(0, len(used)))]
        r["tags"] = [(t[0], t[1], g) if t[0] == "RG" else t for t in r["tags"]]
    text = "@HD\tVN:1.6\n" + "".join("@RG\tID:%s\n" % i for i in ids)
    comp, head_len, n_ref = bam_file(recs, text=text)
    path = tmp_path / "a.bam"
    path.write_bytes(comp)
    rows, rc = host_rows(path, False)
    assert rc == -1 and len(rows) == len(recs)
    reader = bgzf.BamReader(head_len, n_ref, ids)
    got = []
    for info in feed(reader, comp, [len(comp) // 2 + 3]):
        assert info["flags"] == 0, info
        d = reader.batch()
        got.append((info, download_batch(d), download_rg(d)))
        _lib.check(_lib.lib().kbbq_reads_free(None, ctypes.byref(d)))
    check_batches(got, rows)
    assert len(reader.read_groups()) == len(set(r[2] for r in rows))
    reader.close()


def test_a_flagged_chunk_gives_no_batch_and_no_output():
    """kbbq_bam_reader_batch / _write refuse a chunk whose flags say "host parser": no caller can get a wrong batch by
    ignoring them; and --set-oq is refused where bam_aux_update_str would fail."""
    import torch
    base = make_records(30, seed=19, oq_every=0)
    head = bamutil.header(HEADER_TEXT, [("chr1", 1000), ("c2", 5)])
    bad = [dict(r) for r in base]
    bad[4] = dict(bad[4], tags=[t for t in bad[4]["tags"] if t[0] != "RG"])
    reader = bgzf.BamReader(len(head), 2, RG_IDS)
    info = reader.chunk(bamutil.bgzf_compress(head + b"".join(record_bytes(r) for r in bad)), True)
    assert info["flags"] & 1
    with pytest.raises(_lib.KbbqError):
        reader.batch()
    reader.close()
    odd = [dict(r) for r in base]
    odd[4] = dict(odd[4], tags=odd[4]["tags"] + [("OQ", "A", "x")])
    reader = bgzf.BamReader(len(head), 2, RG_IDS)
    writer = bgzf.BgzfWriter()
    info = reader.chunk(bamutil.bgzf_compress(head + b"".join(record_bytes(r) for r in odd)), True)
    assert info["flags"] == 8
    dq = torch.zeros(info["n_bases"] + 16, dtype=torch.uint8, device="cuda")
    with pytest.raises(_lib.KbbqError):
        reader.write(writer, dq.data_ptr(), set_oq=True)
    reader.write(writer, dq.data_ptr(), set_oq=False)              # without --set-oq the tag is not touched
    blob, n_payload = writer.collect()
    newq = np.zeros(info["n_bases"], dtype=np.uint8)
    assert b"".join(bgzf_blocks(blob)) == rewritten(odd, newq, False)
    reader.close()
    writer.close()


def one_chunk_flags(recs=None, stream=None, use_oq=False, rg_ids=RG_IDS):
    if stream is None:
        head = bamutil.header(HEADER_TEXT, [("chr1", 1000)])
        stream = head + b"".join(record_bytes(r) if isinstance(r, dict) else r for r in recs)
    else:
        head = bamutil.header(HEADER_TEXT, [("chr1", 1000)])
    reader = bgzf.BamReader(len(head), 1, rg_ids, use_oq=use_oq)
    info = reader.chunk(bamutil.bgzf_compress(stream), True)
    reader.close()
    return info


def test_streams_that_go_back_to_the_host_parser():
    """Everything bam_io.cc reports, ends a stream on, or reads with a dictionary of its own raises a flag -- never a wrong
    batch: the command line then starts over with BamChunkParser, which prints the reference's messages."""
    base = make_records(40, seed=9, oq_every=1)

    def variant(i, **change):
        recs = [dict(r) for r in base]
        recs[i] = dict(recs[i], **change)
        return recs
    no_rg = [t for t in base[7]["tags"] if t[0] != "RG"]
    no_oq = [t for t in base[7]["tags"] if t[0] != "OQ"]
    assert one_chunk_flags(base)["flags"] == 0 and one_chunk_flags(base, use_oq=True)["flags"] == 0
    assert one_chunk_flags(variant(7, tags=no_rg))["flags"] & 1                                         # RG not found
    assert one_chunk_flags(variant(7, tags=no_rg + [("RG", "i", 5)]))["flags"] & 1                      # RG of another type
    assert one_chunk_flags(variant(7, tags=no_rg + [("RG", "Z", "grpZ")]))["flags"] & 1                 # no @RG line for it
    assert one_chunk_flags(variant(7, tags=no_rg + [("RG", "Z", "grp")]))["flags"] & 1                  # a prefix of an id is not the id
    assert one_chunk_flags(variant(7, tags=no_oq))["flags"] == 0
    assert one_chunk_flags(variant(7, tags=no_oq), use_oq=True)["flags"] & 1                            # --use-oq without the tag
    assert one_chunk_flags(variant(7, tags=no_oq + [("OQ", "Z", "II")]), use_oq=True)["flags"] & 1      # OQ of another length
    f = one_chunk_flags(variant(7, tags=no_oq + [("OQ", "A", "x")]))["flags"]
    assert f & 8 and not f & 1                                                                          # --set-oq could not update it; the passes can read the file
    # no RG among other, well-formed tags; a tag of an unknown type in front of RG (bam_aux_get: EINVAL) and behind it (never seen)
    assert one_chunk_flags(variant(7, tags=[("XX", "Z", "ok")] + no_rg))["flags"] & 1
    raw = record_bytes(dict(base[7], tags=[]))
    bad_aux = raw + b"XQ?\x01" + aux_raw([("RG", "Z", "grpA")])
    bad_aux = struct.pack("<I", len(bad_aux) - 4) + bad_aux[4:]
    assert one_chunk_flags([bad_aux])["flags"] & 1
    ok_aux = raw + aux_raw([("RG", "Z", "grpA")]) + b"XQ?\x01"
    ok_aux = struct.pack("<I", len(ok_aux) - 4) + ok_aux[4:]
    f = one_chunk_flags([ok_aux])["flags"]
    assert not f & 1 and f & 8                      # RG is found before the damage; an OQ could not be appended behind it
    # malformed blocks: a size below the fixed fields, a name length of zero, fields that overrun the block
    good = b"".join(record_bytes(r) for r in base[:5])
    head = bamutil.header(HEADER_TEXT, [("chr1", 1000)])
    assert one_chunk_flags(stream=head + good + struct.pack("<I", 20) + b"\0" * 20 + good)["flags"] & 1
    r5 = bytearray(record_bytes(base[5]))
    r5[12] = 0
    assert one_chunk_flags(stream=head + good + bytes(r5))["flags"] & 1
    r5 = bytearray(record_bytes(base[5]))
    r5[20:24] = struct.pack("<I", 1 << 20)          # l_seq far beyond the block
    assert one_chunk_flags(stream=head + good + bytes(r5))["flags"] & 1
    # the stream ends inside a record (sam_read1 < -1)
    f = one_chunk_flags(stream=head + good + record_bytes(base[5])[:-9])
    assert f["flags"] & 4 and f["n_records"] == 5
    # not BGZF at all
    reader = bgzf.BamReader(len(head), 1, RG_IDS)
    assert reader.chunk(head + good, True)["flags"] & 1
    reader.close()


def test_wrong_guesses_are_repaired_not_believed():
    """Aux payloads that look exactly like alignment records (a B array holding the bytes of real records) put false record
    starts at the front of segments: the chain check must overrule them."""
    base = make_records(300, seed=13, lens=[150], oq_every=0, big_quals=False)
    decoy = b"".join(record_bytes(r) for r in base[:40])            # ~11 KB of perfectly plausible records
    recs = []
    for i, r in enumerate(base):
        tags = list(r["tags"])
        if i % 3 == 0:
            tags.append(("XD", "BC", list(decoy)))
        recs.append(dict(r, tags=tags))
    comp, head_len, n_ref = bam_file(recs, ragged=None)
    reader = bgzf.BamReader(head_len, n_ref, RG_IDS)
    info = reader.chunk(comp, True)
    assert info["flags"] == 0 and info["n_records"] == len(recs)
    d = reader.batch()
    dev = download_batch(d)
    _lib.check(_lib.lib().kbbq_reads_free(None, ctypes.byref(d)))
    want = np.concatenate([bamutil.as_sequenced(r["seq"], r["qual"], r["flag"])[1] for r in recs]).astype(np.uint8)
    assert np.array_equal(dev["qual"], want)
    reader.close()
