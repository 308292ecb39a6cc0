"""A small BAM writer / reader for the tests (SAMv1 section 4), independent of the C++ codec under test:
records are packed with struct, BGZF blocks are made with zlib.  Unaligned records only."""
import struct
import zlib

import numpy as np

CODES = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
LETTERS = "=ACMGRSVTWYHKDBN"
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def bgzf_block(payload):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(payload) + co.flush()
    bsize = len(body) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + body +
            struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf_compress(data, block=0xff00, ragged_seed=None):
    """ragged_seed: cut the stream at random places, so records straddle block boundaries in odd ways."""
    out, pos = [], 0
    rng = np.random.RandomState(ragged_seed) if ragged_seed is not None else None
    while pos < len(data):
        n = block if rng is None else int(rng.randint(1, block))
        out.append(bgzf_block(data[pos:pos + n]))
        pos += n
    return b"".join(out) + BGZF_EOF


def bgzf_decompress(blob):
    out, pos = [], 0
    while pos < len(blob):
        assert blob[pos:pos + 4] == b"\x1f\x8b\x08\x04"
        bsize = struct.unpack("<H", blob[pos + 16:pos + 18])[0] + 1
        out.append(zlib.decompress(blob[pos + 18:pos + bsize - 8], -15))
        pos += bsize
    return b"".join(out)


def aux_bytes(tags):
    """tags: list of (tag, type, value); types A c C s S i I f Z H B(subtype, list)."""
    out = b""
    for tag, typ, val in tags:
        out += tag.encode() + typ[0].encode()
        if typ == "A":
            out += val.encode()
        elif typ in "cCsSiIf":
            out += struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[typ], val)
        elif typ in "ZH":
            out += val.encode() + b"\0"
        elif typ[0] == "B":
            sub = typ[1]
            out += sub.encode() + struct.pack("<I", len(val))
            out += b"".join(struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub], v) for v in val)
    return out


def record(name, flag, seq, qual, tags):
    """seq: text as stored in the record (reference orientation); qual: numeric, same orientation."""
    l = len(seq)
    packed = bytearray((l + 1) // 2)
    for i, ch in enumerate(seq):
        packed[i >> 1] |= CODES[ch] << (4 if i % 2 == 0 else 0)
    body = struct.pack("<iiBBHHHIiii", -1, -1, len(name) + 1, 0, 4680, 0, flag, l, -1, -1, 0)
    body += name.encode() + b"\0" + bytes(packed) + bytes(bytearray(int(q) for q in qual)) + aux_bytes(tags)
    return struct.pack("<I", len(body)) + body


def header(text, refs):
    out = b"BAM\1" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(refs))
    for name, length in refs:
        out += struct.pack("<I", len(name) + 1) + name.encode() + b"\0" + struct.pack("<I", length)
    return out


def parse(stream):
    """Decompressed BAM stream -> (text, refs, [dict(name, flag, seq, qual, aux(raw bytes), raw)])."""
    assert stream[:4] == b"BAM\1"
    l_text = struct.unpack_from("<I", stream, 4)[0]
    text = stream[8:8 + l_text].decode()
    pos = 8 + l_text
    n_ref = struct.unpack_from("<I", stream, pos)[0]
    pos += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<I", stream, pos)[0]
        name = stream[pos + 4:pos + 4 + ln - 1].decode()
        refs.append((name, struct.unpack_from("<I", stream, pos + 4 + ln)[0]))
        pos += 8 + ln
    recs = []
    while pos < len(stream):
        size = struct.unpack_from("<I", stream, pos)[0]
        b = stream[pos + 4:pos + 4 + size]
        l_name, n_cigar, flag, l_seq = b[8], struct.unpack_from("<H", b, 12)[0], struct.unpack_from("<H", b, 14)[0], struct.unpack_from("<I", b, 16)[0]
        at = 32 + l_name + 4 * n_cigar
        seq = "".join(LETTERS[(b[at + (i >> 1)] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        q_at = at + (l_seq + 1) // 2
        recs.append(dict(name=b[32:32 + l_name - 1].decode(), flag=flag, seq=seq, qual=np.frombuffer(b[q_at:q_at + l_seq], dtype=np.uint8).copy(),
                         aux=b[q_at + l_seq:], raw=b))
        pos += 4 + size
    return text, refs, recs


def as_sequenced(seq, qual, flag):
    """What CReadData's BAM constructor hands to the passes (readutils.hh:30-42, readutils.cc:36-39)."""
    if flag & 16:
        return "".join(COMP.get(c, "N") for c in reversed(seq)), np.asarray(qual)[::-1].copy()
    return seq, np.asarray(qual).copy()
