"""Read batches in the engine's structure-of-arrays layout (include/kbbq_engine.h).

The host side of the boundary: what FastqFile::get / BamFile::get hand to the
pass loops one record at a time (htsiter.cc:9,57-59; readutils.cc:13-104)
becomes one packed batch.
"""
import ctypes

import numpy as np

from . import _lib


class ReadBatch:
    """A batch of reads on the host, packed for the engine.

    seq:    uint8 ASCII bases, all reads concatenated
    qual:   uint8 phred values (FASTQ character - 33), concatenated
    off:    uint64 read boundaries, n_reads + 1 entries (off[0] = 0)
    rg:     per-read dense read-group index (order of first appearance); None = 0
    second: per-read second-in-pair flag; None = 0
    uniform: pass the batch without offsets when every read has the same length
    """

    def __init__(self, seq, qual, off, rg=None, second=None, uniform=None):
        L = _lib.lib()
        self.seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self.off = np.ascontiguousarray(off, dtype=np.uint64)
        self.n_reads = len(self.off) - 1
        self.n_bases = int(self.off[-1])
        assert len(self.seq) == self.n_bases and len(qual) == self.n_bases
        # 16 spare bytes: the apply kernel loads qualities 16 at a time
        self.qual = np.zeros(self.n_bases + 16, dtype=np.uint8)
        self.qual[:self.n_bases] = qual
        self.bases = np.zeros(self.n_bases // 32 + 2, dtype=np.uint64)
        self.nmask = np.zeros(self.n_bases // 64 + 2, dtype=np.uint64)
        # the case bits travel only when some base is off-case (a soft-masked FASTQ): kbbq_reads.offcase stays NULL otherwise
        self.offcase = np.zeros(self.n_bases // 64 + 2, dtype=np.uint64)
        n_off = ctypes.c_uint64()
        _lib.check(L.kbbq_pack_bases_case(self.seq.ctypes.data_as(_lib.c_u8p), self.n_bases,
                                          self.bases.ctypes.data_as(_lib.c_u64p), self.nmask.ctypes.data_as(_lib.c_u64p),
                                          self.offcase.ctypes.data_as(_lib.c_u64p), ctypes.byref(n_off)))
        self.n_offcase = n_off.value
        lens = np.diff(self.off.astype(np.int64))
        self.max_len = int(lens.max()) if self.n_reads else 0
        if uniform is None:
            uniform = False
        if uniform:
            assert self.n_reads and (lens == lens[0]).all(), "uniform=True needs equal read lengths"
        self.uniform = bool(uniform)
        self.rg = None if rg is None else np.ascontiguousarray(rg, dtype=np.uint16)
        self.flags = None if second is None else np.ascontiguousarray(np.asarray(second) != 0, dtype=np.uint8)
        self.c = _lib.Reads()
        self.c.n_reads = self.n_reads
        self.c.n_bases = self.n_bases
        self.c.bases = self.bases.ctypes.data
        self.c.nmask = self.nmask.ctypes.data
        self.c.qual = self.qual.ctypes.data
        self.c.offsets = None if self.uniform else self.off.ctypes.data
        self.c.flags = None if self.flags is None else self.flags.ctypes.data
        self.c.rg = None if self.rg is None else self.rg.ctypes.data
        self.c.read_len = int(lens[0]) if self.uniform else 0
        self.c.on_device = 0
        self.c.offcase = self.offcase.ctypes.data if self.n_offcase else None

    def n_kmer_positions(self, k):
        lens = np.diff(self.off.astype(np.int64))
        return int(np.maximum(lens - k + 1, 0).sum())

    def slice(self, a, b):
        """Reads a..b as a new host batch."""
        s, e = int(self.off[a]), int(self.off[b])
        return ReadBatch(self.seq[s:e], self.qual[s:e], self.off[a:b + 1] - self.off[a],
                         None if self.rg is None else self.rg[a:b], None if self.flags is None else self.flags[a:b],
                         uniform=self.uniform)


def unpack_bits(words, n):
    """uint64 bit words (bit i%64 of word i/64) -> uint8 0/1 array of length n."""
    b = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")
    return b[:n].copy()


def pack_bits(bits):
    n = len(bits)
    out = np.zeros(n // 64 + 2, dtype=np.uint64)
    packed = np.packbits(np.asarray(bits, dtype=np.uint8), bitorder="little")
    out.view(np.uint8)[:len(packed)] = packed
    return out
