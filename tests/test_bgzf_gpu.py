"""The BGZF writer on the device (include/kbbq_bgzf.h, kbbq_amd/csrc/bgzf_device.*): whatever it returns must be a valid
BGZF stream that inflates to the payload -- block framing, CRC-32 and ISIZE checked per block with zlib -- for text,
binary, incompressible and degenerate payloads, for two submissions in flight, and for FASTQ records whose text the
device assembles around qualities that never leave HBM (FastqFile::write, htsiter.cc:75-86)."""
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
