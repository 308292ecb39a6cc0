"""The scalar pieces of the device BGZF encoder (kbbq_amd/csrc/deflate_common.h: Huffman code lengths, canonical codes,
code-length header, token bits, CRC-32 chaining, BGZF framing) through their host-only twin: whatever it writes, zlib
must inflate to the input, block by block, with the right CRC and sizes -- no GPU needed."""
import gzip
import struct
import zlib

import numpy as np
import pytest

import common  # noqa: F401  (sys.path)
from kbbq_amd import bgzf


def bgzf_blocks(data):
    """Split a BGZF stream into its blocks, checking the framing of the SAM specification (section 4.1)."""
    out, at = [], 0
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04" and data[at + 10:at + 12] == b"\x06\x00" and data[at + 12:at + 14] == b"BC"
        assert struct.unpack_from("<H", data, at + 14)[0] == 2
        size = struct.unpack_from("<H", data, at + 16)[0] + 1
        body = data[at + 18:at + size - 8]
        crc, isize = struct.unpack_from("<II", data, at + size - 8)
        raw = zlib.decompress(body, -15)
        assert len(raw) == isize and zlib.crc32(raw) == crc
        out.append(raw)
        at += size
    assert at == len(data)
    return out


def fastq_text(n, seed=1, read_len=150):
    rng = np.random.RandomState(seed)
    recs = []
    for i in range(n):
        seq = "".join(rng.choice(list("ACGT"), read_len))
        q = "".join(rng.choice(list("#-7AF"), read_len, p=[.02, .05, .1, .2, .63]))
        recs.append("@read%d/1 extra\n%s\n+\n%s\n" % (i, seq, q))
    return "".join(recs).encode()


PAYLOADS = {
    "empty": b"",
    "one_byte": b"x",
    "tiny": b"abcabcabcabc",
    "zeros": bytes(70000),
    "one_block_exactly": bytes(range(256)) * 255,                       # 0xff00 bytes
    "one_over": bytes(range(256)) * 255 + b"!",
    "random": np.random.RandomState(5).randint(0, 256, 200000).astype(np.uint8).tobytes(),      # incompressible: stored blocks
    "fastq": fastq_text(1500),
    "runs": b"".join(bytes([c]) * n for c, n in zip(range(256), np.random.RandomState(2).randint(1, 900, 256))),
    "two_symbols": b"ab" * 40000,
    # every byte value, short matches, many symbols: a long code-length header (BAM-like binary content)
    "binary_records": b"".join(bytes(np.random.RandomState(100 + i).randint(0, 256, 40).astype(np.uint8)) + bytes([i & 255] * (i % 9)) + b"\x00\x01\x00\x00RG:Z:grp%d\x00" % (i % 4)
                               for i in range(4000)),
    "skewed": bytes(np.random.RandomState(9).choice(np.arange(40, dtype=np.uint8), 150000, p=np.array([2.0 ** -i for i in range(1, 40)] + [2.0 ** -39]))),
}


@pytest.mark.parametrize("name", sorted(PAYLOADS))
def test_host_twin_writes_valid_bgzf(name):
    data = PAYLOADS[name]
    comp = bgzf.host_compress(data)
    blocks = bgzf_blocks(comp)
    assert b"".join(blocks) == data
    assert all(len(b) == 0xff00 for b in blocks[:-1]) and (not blocks or 0 < len(blocks[-1]) <= 0xff00)
    if data:
        assert gzip.decompress(comp) == data
    assert len(comp) <= bgzf._lib.lib().kbbq_bgzf_bound(len(data))


def test_host_twin_compresses_fastq_within_a_third_of_zlib():
    data = fastq_text(3000, seed=3)
    ours, ref = len(bgzf.host_compress(data)), len(zlib.compress(data, 6))
    assert ours <= 1.3 * ref, (ours, ref)
