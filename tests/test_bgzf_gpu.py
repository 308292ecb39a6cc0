"""The BGZF writer on the device (include/kbbq_bgzf.h, kbbq_amd/csrc/bgzf_device.*): whatever it returns must be a valid
BGZF stream that inflates to the payload -- block framing, CRC-32 and ISIZE checked per block with zlib -- for text,
binary, incompressible and degenerate payloads, for two submissions in flight, and for FASTQ records whose text the
device assembles around qualities that never leave HBM (FastqFile::write, htsiter.cc:75-86)."""
import ctypes
import zlib

import numpy as np
import pytest

import common  # noqa: F401
from kbbq_amd import _lib, bgzf
from test_bgzf_cpu import PAYLOADS, bgzf_blocks, fastq_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def writer():
    w = bgzf.BgzfWriter()
    yield w
    w.close()


@pytest.mark.parametrize("name", sorted(n for n in PAYLOADS if PAYLOADS[n]))
def test_device_encoder_writes_valid_bgzf(writer, name):
    data = PAYLOADS[name]
    comp = writer.compress(data)
    blocks = bgzf_blocks(comp)
    assert b"".join(blocks) == data
    assert all(len(b) == 0xff00 for b in blocks[:-1])


def test_empty_payload_is_refused(writer):
    with pytest.raises(_lib.KbbqError):
        writer.submit(b"")


def test_compression_ratio_on_fastq_text(writer):
    data = fastq_text(20000, seed=11)
    ours, ref = len(writer.compress(data)), len(zlib.compress(data, 6))
    assert ours <= 1.3 * ref, (ours, ref)


def test_two_submissions_in_flight_come_back_in_order(writer):
    a, b, c = fastq_text(4000, seed=1), PAYLOADS["random"], fastq_text(2500, seed=2, read_len=100)
    writer.submit(a)
    writer.submit(b)
    with pytest.raises(_lib.KbbqError):
        writer.submit(c)                      # a third one must wait for a collect
    got_a, n_a = writer.collect()
    writer.submit(c)
    got_b, n_b = writer.collect()
    got_c, n_c = writer.collect()
    assert (n_a, n_b, n_c) == (len(a), len(b), len(c))
    for got, want in ((got_a, a), (got_b, b), (got_c, c)):
        assert b"".join(bgzf_blocks(got)) == want
    with pytest.raises(_lib.KbbqError):
        writer.collect()


@pytest.mark.parametrize("uniform", [True, False])
def test_fastq_records_assembled_on_the_device(writer, uniform):
    """names, comments and sequence text from the host, the quality line from device memory (+33)."""
    import torch
    rng = np.random.RandomState(4 if uniform else 5)
    n = 5000
    lens_seq = np.full(n, 150) if uniform else rng.randint(1, 400, n)
    names = [("r%d/%d" % (i, 1 + i % 2)).encode() for i in range(n)]
    comments = [b"" if i % 3 else ("BX:Z:%d" % i).encode() for i in range(n)]
    seqs = ["".join(rng.choice(list("ACGTNacgt"), l)).encode() for l in lens_seq]
    quals = [rng.randint(0, 94, l).astype(np.uint8) for l in lens_seq]
    blob = b"".join(nm + cm + sq for nm, cm, sq in zip(names, comments, seqs))
    lens = np.array([[len(nm), len(cm), len(sq)] for nm, cm, sq in zip(names, comments, seqs)], dtype=np.uint32)
    q_all = torch.from_numpy(np.concatenate(quals)).cuda()
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens_seq)]).astype(np.int64)).cuda()
    torch.cuda.synchronize()
    writer.submit_fastq(blob, lens, q_all.data_ptr(), None if uniform else off.data_ptr(), 150 if uniform else 0)
    comp, n_text = writer.collect()
    want = b"".join(b"@" + nm + b"\n" + sq + b"\n+" + cm + b"\n" + bytes(q + 33) + b"\n" for nm, cm, sq, q in zip(names, comments, seqs, quals))
    assert n_text == len(want)
    assert b"".join(bgzf_blocks(comp)) == want


def test_large_payload_and_kernel_times(writer):
    data = fastq_text(60000, seed=21) * 8          # about 150 MB of text, 2 300 blocks
    comp = writer.compress(data)
    assert zlib.crc32(b"".join(bgzf_blocks(comp))) == zlib.crc32(data)
    ms = writer.kernel_ms()
    assert ms["deflate"] > 0
    print("deflate %.1f ms for %.1f MB -> %.1f MB" % (ms["deflate"], len(data) / 1e6, len(comp) / 1e6))


# ---- the input side: BGZF FASTQ read on the device -------------------------------------------------------------------------

def bgzip(data, level=6, block=0xff00):
    """BGZF as bgzip / htslib write it: zlib raw deflate per 0xff00 payload bytes + the EOF block."""
    import struct
    out = []
    for at in range(0, len(data), block):
        raw = data[at:at + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(raw) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body +
                   struct.pack("<II", zlib.crc32(raw), len(raw)))
    out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")
    return b"".join(out)


def make_records(n, seed, uniform=True, comments=True, lower=False):
    rng = np.random.RandomState(seed)
    recs = []
    alphabet = list("ACGTN") + (list("acgt") if lower else [])
    for i in range(n):
        l = 150 if uniform else int(rng.randint(1, 300))
        name = "r%d/%d" % (i, 1 + i % 2) if i % 5 else "lane3_tile%d_x%d" % (i, i * 7)
        comment = (" BX:Z:%d extra words" % i) if (comments and i % 3 == 0) else ("\tTAB" if comments and i % 7 == 0 else "")
        seq = "".join(rng.choice(alphabet, l))
        q = bytes(rng.randint(33, 33 + 60, l).astype(np.uint8)).decode()
        plus = "+" + (name if i % 11 == 0 else "")
        recs.append((name, comment, seq, q, plus))
    text = "".join("@%s%s\n%s\n%s\n%s\n" % (nm, cm, sq, pl, q) for nm, cm, sq, q, pl in recs).encode()
    return recs, text


def download_batch(d):
    import torch
    from kbbq_amd.engine import device_tensor

    def get(ptr, nbytes, dt):
        return device_tensor(ptr, nbytes, torch.uint8, 0).cpu().numpy().view(dt).copy() if ptr else None
    nb, nr = int(d.n_bases), int(d.n_reads)
    return dict(bases=get(d.bases, (nb // 32 + 1) * 8, np.uint64), nmask=get(d.nmask, (nb // 64 + 1) * 8, np.uint64), qual=get(d.qual, nb, np.uint8),
                offsets=get(d.offsets, (nr + 1) * 8, np.uint64), flags=get(d.flags, nr, np.uint8), offcase=get(d.offcase, (nb // 64 + 1) * 8, np.uint64),
                read_len=int(d.read_len))


@pytest.mark.parametrize("level", [1, 6, 9, 0])
def test_inflate_on_the_device_every_block_type(level):
    """zlib level 0 writes stored blocks, 1-9 dynamic ones, tiny payloads fixed ones: the reader's chunk() inflates them all
    (the text is checked through the records it finds)."""
    recs, text = make_records(3000, seed=level)
    r = bgzf.FastqReader()
    info = r.chunk(bgzip(text, level), True)
    assert info["flags"] == 0 and info["n_records"] == len(recs) and info["text_bytes"] == len(text)
    assert info["n_bases"] == sum(len(x[2]) for x in recs)
    r.close()


def bgzf_member(body, raw):
    import struct
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body +
            struct.pack("<II", zlib.crc32(raw), len(raw)))


EOF_BLOCK = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0"


def test_inflate_fixed_codes_and_several_deflate_blocks_in_one_member():
    """Shapes bgzip does not write but RFC 1951 allows and htslib reads: fixed Huffman codes (Z_FIXED), a member whose stream is
    several DEFLATE blocks of different kinds (full flushes between pieces compressed with different strategies: dynamic,
    stored through an incompressible piece, fixed), an empty member in the middle of the file."""
    recs, text = make_records(2500, seed=12)
    members = []
    for at in range(0, len(text), 30000):
        raw = text[at:at + 30000]
        kind = (at // 30000) % 3
        if kind == 0:
            co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
            body = co.compress(raw) + co.flush()
        elif kind == 1:
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = co.compress(raw[:9000]) + co.flush(zlib.Z_FULL_FLUSH)      # dynamic block + an empty stored block
            body += co.compress(raw[9000:20000]) + co.flush(zlib.Z_SYNC_FLUSH)
            body += co.compress(raw[20000:]) + co.flush()
        else:
            co = zlib.compressobj(1, zlib.DEFLATED, -15, 8, zlib.Z_HUFFMAN_ONLY)
            body = co.compress(raw) + co.flush()
        members.append(bgzf_member(body, raw))
        if kind == 1:
            members.append(bgzf_member(zlib.compressobj(6, zlib.DEFLATED, -15).flush(), b""))      # an empty member
    r = bgzf.FastqReader()
    info = r.chunk(b"".join(members) + EOF_BLOCK, True)
    assert info["flags"] == 0 and info["n_records"] == len(recs) and info["text_bytes"] == len(text)
    d = r.batch()
    got = download_batch(d)
    want_q = np.frombuffer("".join(x[3] for x in recs).encode(), dtype=np.uint8) - 33
    assert np.array_equal(got["qual"], want_q)
    _lib.check(_lib.lib().kbbq_reads_free(None, ctypes.byref(d)))
    r.close()


def test_inflate_arbitrary_streams_against_zlib():
    """tests/fuzz_inflate.py: random bytes, runs, long-range repeats and skewed alphabets through every strategy, level, window
    and flush zlib offers, as BGZF members of random sizes: the device's inflation equals the data byte for byte."""
    import fuzz_inflate
    assert fuzz_inflate.run(60, 71) > 0


def test_damaged_blocks_are_reported_not_followed():
    """What bgzf_read answers with an error: a flipped bit in the DEFLATE stream (bad code, bad distance, wrong size, or --
    when the stream still decodes to the right number of bytes -- the CRC-32), a wrong CRC in the trailer, a wrong ISIZE.
    The reader must return an error for every one of them and stay usable."""
    recs, text = make_records(1500, seed=3)
    good = bgzip(text, 6, block=20000)
    r = bgzf.FastqReader()
    rng = np.random.RandomState(9)
    n_err = 0
    for trial in range(40):
        bad = bytearray(good)
        if trial < 30:
            at = int(rng.randint(18, len(good) - 28 - 8))      # somewhere in the members (a header byte breaks the framing: flag or error)
            bad[at] ^= 1 << int(rng.randint(0, 8))
        elif trial < 35:
            # the CRC of the first member
            bsize = bad[16] | (bad[17] << 8)
            bad[bsize + 1 - 8] ^= 0x40
        else:
            bsize = bad[16] | (bad[17] << 8)
            bad[bsize + 1 - 4] ^= 0x01                          # ISIZE
        try:
            info = r.chunk(bytes(bad), True)
            # a flip inside a header's fixed fields makes the file "not BGZF" (flag) or cuts the chunk short
            assert info["flags"] != 0 or info["n_records"] != len(recs) or info["consumed"] != len(bad), trial
        except Exception as ex:
            n_err += 1
            assert "BGZF" in str(ex) or "CRC" in str(ex) or "block" in str(ex), ex
        r.rewind()
    assert n_err >= 30
    info = r.chunk(good, True)                                   # and the reader still works
    assert info["flags"] == 0 and info["n_records"] == len(recs)
    r.close()


@pytest.mark.parametrize("uniform,lower", [(True, False), (False, False), (False, True)])
def test_reader_batch_equals_the_host_packing(uniform, lower):
    """The device batch of a chunk (2-bit bases, N mask, qualities - 33, offsets, second-in-pair flags, off-case bits) equals
    what ReadBatch / kbbq_pack_bases_case make of the same records on the host; the file is fed in three chunks that cut
    BGZF blocks and records anywhere."""
    from kbbq_amd.reads import ReadBatch
    recs, text = make_records(4000, seed=7 + uniform, uniform=uniform, lower=lower)
    comp = bgzip(text, 6, block=20000)
    cuts = [0, len(comp) // 3 + 5, 2 * len(comp) // 3 + 11, len(comp)]
    r = bgzf.FastqReader()
    got, pending = [], b""
    for a, b in zip(cuts[:-1], cuts[1:]):
        data = pending + comp[a:b]
        info = r.chunk(data, b == len(comp))
        assert info["flags"] == 0
        pending = data[info["consumed"]:]
        if info["n_records"]:
            d = r.batch()
            got.append((info, download_batch(d)))
            _lib.check(_lib.lib().kbbq_reads_free(None, ctypes.byref(d)))
    assert not pending and sum(i["n_records"] for i, _ in got) == len(recs)
    at = 0
    for info, dev in got:
        n = info["n_records"]
        part = recs[at:at + n]
        at += n
        seq = np.frombuffer("".join(x[2] for x in part).encode(), dtype=np.uint8)
        qual = np.frombuffer("".join(x[3] for x in part).encode(), dtype=np.uint8) - 33
        off = np.concatenate([[0], np.cumsum([len(x[2]) for x in part])]).astype(np.uint64)
        second = np.array([1 if x[0].split("_")[0].endswith("/2") else 0 for x in part], dtype=np.uint8)
        hb = ReadBatch(seq, qual, off, np.zeros(n, np.uint16), second, uniform=False)
        nbw, nmw = len(seq) // 32 + 1, len(seq) // 64 + 1
        assert np.array_equal(dev["bases"][:nbw], hb.bases[:nbw]) and np.array_equal(dev["nmask"][:nmw], hb.nmask[:nmw])
        assert np.array_equal(dev["qual"], qual) and np.array_equal(dev["flags"], second)
        if dev["offsets"] is None:
            assert dev["read_len"] == 150 and all(len(x[2]) == 150 for x in part)
        else:
            assert np.array_equal(dev["offsets"], off)
        lower_bits = np.array([c in b"acgt0123" for c in seq.tobytes()], dtype=np.uint8)
        if lower_bits.any():
            assert np.array_equal(np.unpackbits(dev["offcase"].view(np.uint8), bitorder="little")[:len(seq)], lower_bits)
        else:
            assert dev["offcase"] is None
    r.close()


def test_reader_reports_the_shapes_it_does_not_take():
    r = bgzf.FastqReader()
    good = "@a1\nACGT\n+\nIIII\n"
    for bad, flag in (("@a1\nACGT\n+\nIII\n", 1),                 # qualities shorter than the sequence
                      ("@a1\nAC\nGT\n+\nIIII\n@b2\nAC\n+\nII\n@c\nA\n+\nI\n", 1),      # a two-line sequence shifts the lines
                      (">a1\nACGT\n>b2\nACGT\n", 1),                # FASTA
                      ("@a1\r\nACGT\r\n+\r\nIIII\r\n", 1),          # carriage returns
                      ("@x_RG:grp1_more\nACGT\n+\nIIII\n", 1),      # a read group in the name: the host path's dictionary
                      ("@a\nACGT\n+\nIIII\n", 2),                   # read name shorter than two characters (readutils.cc:90)
                      ("@a1\n\n+\n\n", 1),                          # an empty read
                      (good + "@b2\nACGT\n+\nIII", 4)):             # the file ends inside a record
        r.rewind()
        info = r.chunk(bgzip((good + bad).encode() if flag != 4 else bad.encode()), True)
        assert info["flags"] & flag, (bad, info)
    r.rewind()
    assert r.chunk(b"@a1\nACGT\n+\nIIII\n" * 10, True)["flags"] & 1      # not BGZF at all
    r.rewind()
    assert r.chunk(bgzip((good * 50).encode()), True)["flags"] == 0
    r.close()


def test_reader_and_writer_round_trip_with_new_qualities(writer):
    """Pass 4 of the device path: the chunk's records come back as "@name\\nseq\\n+comment\\nqual\\n" with the qualities
    taken from a device array -- the comment moves to the '+' line and whatever stood there is dropped (htsiter.cc:75-86)."""
    import torch
    recs, text = make_records(6000, seed=33, uniform=False)
    r = bgzf.FastqReader()
    info = r.chunk(bgzip(text, 6), True)
    assert info["flags"] == 0 and info["n_records"] == len(recs)
    rng = np.random.RandomState(1)
    newq = rng.randint(0, 94, info["n_bases"]).astype(np.uint8)
    dq = torch.from_numpy(newq).cuda()
    torch.cuda.synchronize()
    r.write(writer, dq.data_ptr())
    comp, n_text = writer.collect()
    want, at = [], 0
    for nm, cm, sq, q, pl in recs:
        l = len(sq)
        comment = cm[1:] if cm else ""
        want.append(b"@" + nm.encode() + b"\n" + sq.encode() + b"\n+" + comment.encode() + b"\n" + bytes(newq[at:at + l] + 33) + b"\n")
        at += l
    want = b"".join(want)
    assert n_text == len(want)
    assert b"".join(bgzf_blocks(comp)) == want
    r.close()


@pytest.mark.parametrize("form", ["whole", "names", "names_lower", "exotic"])
def test_kept_chunks_are_written_without_a_second_scan(writer, form):
    """kbbq_fastq_reader_keep: the chunks of the first scan stay on the device; pass 4 selects them one by one and gets the
    same text as a second inflation would give.  A file in three pieces, the cuts inside records.  whole: no batch was
    built, the text stays; names: the batch gives the sequence lines back (ACGTN), only names and comments stay;
    names_lower: the same with soft-masked bases; exotic: an IUPAC code in a read -- that chunk keeps its text."""
    import torch
    recs, text = make_records(9000, seed=41, uniform=False, lower=form == "names_lower")
    if form == "exotic":
        nm, cm, sq, q, pl = recs[100]
        recs[100] = (nm, cm, "R" + sq[1:], q, pl)
        text = "".join("@%s%s\n%s\n%s\n%s\n" % (nm, cm, sq, pl, q) for nm, cm, sq, q, pl in recs).encode()
    comp = bgzip(text, 6, block=20000)
    # cut at BGZF block boundaries (the reader takes whole blocks and reports what it consumed)
    r = bgzf.FastqReader()
    r.keep(True)
    pieces, at, counts, batches = [len(comp) // 3, 2 * len(comp) // 3, len(comp)], 0, [], []
    for end in pieces:
        info = r.chunk(comp[at:end], end == len(comp))
        assert info["flags"] == 0
        at += info["consumed"]
        counts.append((info["n_records"], info["n_bases"]))
        if form != "whole" and info["n_records"]:
            batches.append(r.batch())
    assert sum(c[0] for c in counts) == len(recs)
    r.rewind()
    n_kept, n_bytes = r.kept()
    assert n_kept == sum(1 for c in counts if c[0])
    if form in ("names", "names_lower"):
        assert n_bytes < len(text) // 2          # names, comments, lengths and two scans
    elif form == "whole":
        assert n_bytes > len(text)
    else:
        assert len(text) // 3 < n_bytes          # the first chunk whole, the others short
    r.rewind()
    assert r.kept()[0] == n_kept
    rng = np.random.RandomState(2)
    got, want, rec_at = [], [], 0
    k = 0
    for n_rec, n_bases in counts:
        if not n_rec:
            continue
        info = r.select(k)
        if form != "whole":
            if form != "exotic" or k > 0:
                with pytest.raises(Exception):       # a chunk kept without its text needs its batch
                    r.write(writer, 0x1000)
            r.attach(batches[k])
        k += 1
        assert info["n_records"] == n_rec and info["n_bases"] == n_bases
        newq = rng.randint(0, 94, n_bases).astype(np.uint8)
        dq = torch.from_numpy(newq).cuda()
        torch.cuda.synchronize()
        r.write(writer, dq.data_ptr())
        comp_out, n_text = writer.collect()
        got.append(b"".join(bgzf_blocks(comp_out)))
        qa = 0
        for nm, cm, sq, q, pl in recs[rec_at:rec_at + n_rec]:
            l = len(sq)
            want.append(b"@" + nm.encode() + b"\n" + sq.encode() + b"\n+" + (cm[1:] if cm else "").encode() + b"\n" + bytes(newq[qa:qa + l] + 33) + b"\n")
            qa += l
        rec_at += n_rec
    assert b"".join(got) == b"".join(want)
    for d in batches:
        _lib.check(_lib.lib().kbbq_reads_free(None, ctypes.byref(d)))
    # giving the kept chunks up: nothing left to select
    r.keep(False)
    assert r.kept() == (0, 0)
    with pytest.raises(Exception):
        r.select(0)
    r.close()


def test_reader_large_file_rates(writer):
    recs, text = make_records(20000, seed=5)
    text = text * 40                     # 250 MB of text
    comp = bgzip(text, 6)
    r = bgzf.FastqReader()
    info = r.chunk(comp, True)
    assert info["flags"] == 0 and info["n_records"] == 20000 * 40
    a, b = ctypes.c_double(), ctypes.c_double()
    _lib.check(_lib.lib().kbbq_fastq_reader_kernel_ms(r.h, ctypes.byref(a), ctypes.byref(b)))
    print("inflate %.1f ms, index %.1f ms for %.0f MB compressed -> %.0f MB" % (a.value, b.value, len(comp) / 1e6, len(text) / 1e6))
    r.close()
