"""Host-side handle of the BGZF writer on the device (include/kbbq_bgzf.h): the output side of the reference's
FastqFile::write / BamFile::write (htsiter.cc:45,75-86) -- record assembly, DEFLATE and CRC-32 as HIP kernels."""
import ctypes

import numpy as np

from . import _lib


class BgzfWriter:
    def __init__(self, device=0):
        self.L = _lib.lib()
        self.h = _lib.c_vp()
        _lib.check(self.L.kbbq_bgzf_create(device, ctypes.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.L.kbbq_bgzf_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def submit(self, payload, after_stream=None):
        """payload: bytes / numpy uint8 array in host memory."""
        a = np.frombuffer(payload, dtype=np.uint8) if not isinstance(payload, np.ndarray) else np.ascontiguousarray(payload, np.uint8)
        _lib.check(self.L.kbbq_bgzf_submit(self.h, a.ctypes.data, a.size, 0, after_stream))

    def submit_device(self, ptr, n, after_stream=None):
        _lib.check(self.L.kbbq_bgzf_submit(self.h, ptr, n, 1, after_stream))

    def submit_fastq(self, blob, lens, d_qual, d_qual_offsets=None, uniform_len=0, after_stream=None):
        """blob: bytes (name|comment|seq per record), lens: uint32 array [n][3], d_qual: device pointer of the new qualities."""
        b = np.frombuffer(blob, dtype=np.uint8)
        ln = np.ascontiguousarray(lens, dtype=np.uint32).reshape(-1, 3)
        _lib.check(self.L.kbbq_bgzf_submit_fastq(self.h, b.ctypes.data, ln.ctypes.data, ln.shape[0], d_qual, d_qual_offsets,
                                                 uniform_len, after_stream))

    def collect(self):
        """The BGZF blocks of the oldest submission as bytes, and its payload size."""
        p, n, m = _lib.c_vp(), ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self.L.kbbq_bgzf_collect(self.h, ctypes.byref(p), ctypes.byref(n), ctypes.byref(m)))
        return ctypes.string_at(p.value, n.value), m.value

    def compress(self, payload):
        self.submit(payload)
        return self.collect()[0]

    def kernel_ms(self):
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _lib.check(self.L.kbbq_bgzf_kernel_ms(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return dict(format=a.value, deflate=b.value, gather=c.value)

    def eof_block(self):
        return ctypes.string_at(self.L.kbbq_bgzf_eof_block(), 28)


def host_compress(payload):
    """The host-only twin (no GPU): same Huffman / header / CRC code around a serial match finder."""
    L = _lib.lib()
    a = np.frombuffer(payload, dtype=np.uint8)
    cap = int(L.kbbq_bgzf_bound(a.size))
    out = np.zeros(cap, dtype=np.uint8)
    n = ctypes.c_uint64()
    _lib.check(L.kbbq_host_bgzf_compress(a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap, ctypes.byref(n)))
    return out[:n.value].tobytes()


class FastqReader:
    """A BGZF-compressed four-line FASTQ file read on the device (include/kbbq_bgzf.h: kbbq_fastq_reader)."""

    def __init__(self, device=0):
        self.L = _lib.lib()
        self.h = _lib.c_vp()
        _lib.check(self.L.kbbq_fastq_reader_create(device, ctypes.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.L.kbbq_fastq_reader_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def rewind(self):
        _lib.check(self.L.kbbq_fastq_reader_rewind(self.h))

    def chunk(self, data, last):
        """Feed bytes of the file; returns the kbbq_fastq_chunk as a dict."""
        a = np.frombuffer(data, dtype=np.uint8)
        info = _lib.FastqChunk()
        _lib.check(self.L.kbbq_fastq_reader_chunk(self.h, a.ctypes.data if a.size else None, a.size, 1 if last else 0, ctypes.byref(info)))
        return {k: getattr(info, k) for k, _ in _lib.FastqChunk._fields_}

    def batch(self):
        d = _lib.Reads()
        _lib.check(self.L.kbbq_fastq_reader_batch(self.h, ctypes.byref(d)))
        return d

    def write(self, writer, d_qual, after_stream=None):
        _lib.check(self.L.kbbq_fastq_reader_write(self.h, writer.h, d_qual, after_stream))

    def kernel_ms(self):
        a, b = ctypes.c_double(), ctypes.c_double()
        _lib.check(self.L.kbbq_fastq_reader_kernel_ms(self.h, ctypes.byref(a), ctypes.byref(b)))
        return dict(inflate=a.value, index=b.value)

    def keep(self, on=True):
        _lib.check(self.L.kbbq_fastq_reader_keep(self.h, 1 if on else 0))

    def kept(self):
        n, b = ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self.L.kbbq_fastq_reader_kept(self.h, ctypes.byref(n), ctypes.byref(b)))
        return n.value, b.value

    def select(self, i):
        info = _lib.FastqChunk()
        _lib.check(self.L.kbbq_fastq_reader_select(self.h, i, ctypes.byref(info)))
        return {k: getattr(info, k) for k, _ in _lib.FastqChunk._fields_}

    def attach(self, batch):
        _lib.check(self.L.kbbq_fastq_reader_attach(self.h, ctypes.byref(batch)))


class BamReader:
    """A BAM file read on the device (include/kbbq_bgzf.h: kbbq_bam_reader): inflate, record chain, field decode, and --
    pass 4 -- the records rewritten around the new qualities.  header_bytes / n_ref / rg_ids come from the caller's own
    parse of the BAM header."""

    def __init__(self, header_bytes, n_ref, rg_ids, use_oq=False, device=0):
        self.L = _lib.lib()
        self.h = _lib.c_vp()
        ids = (ctypes.c_char_p * max(1, len(rg_ids)))(*[i.encode() if isinstance(i, str) else i for i in rg_ids])
        _lib.check(self.L.kbbq_bam_reader_create(device, 1 if use_oq else 0, n_ref, header_bytes, ids, len(rg_ids), ctypes.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            self.L.kbbq_bam_reader_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def rewind(self):
        _lib.check(self.L.kbbq_bam_reader_rewind(self.h))

    def keep(self, on=True):
        _lib.check(self.L.kbbq_bam_reader_keep(self.h, 1 if on else 0))

    def kept(self):
        n, b = ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self.L.kbbq_bam_reader_kept(self.h, ctypes.byref(n), ctypes.byref(b)))
        return n.value, b.value

    def chunk(self, data, last):
        a = np.frombuffer(data, dtype=np.uint8)
        info = _lib.FastqChunk()
        _lib.check(self.L.kbbq_bam_reader_chunk(self.h, a.ctypes.data if a.size else None, a.size, 1 if last else 0, ctypes.byref(info)))
        return {k: getattr(info, k) for k, _ in _lib.FastqChunk._fields_}

    def select(self, i):
        info = _lib.FastqChunk()
        _lib.check(self.L.kbbq_bam_reader_select(self.h, i, ctypes.byref(info)))
        return {k: getattr(info, k) for k, _ in _lib.FastqChunk._fields_}

    def read_groups(self):
        """Table indices (into rg_ids) of the read groups met so far, in dense-index order."""
        n = ctypes.c_uint32()
        _lib.check(self.L.kbbq_bam_reader_read_groups(self.h, None, 0, ctypes.byref(n)))
        out = (ctypes.c_uint32 * max(1, n.value))()
        _lib.check(self.L.kbbq_bam_reader_read_groups(self.h, out, n.value, ctypes.byref(n)))
        return list(out[:n.value])

    def batch(self):
        d = _lib.Reads()
        _lib.check(self.L.kbbq_bam_reader_batch(self.h, ctypes.byref(d)))
        return d

    def write(self, writer, d_qual, set_oq=False, after_stream=None):
        _lib.check(self.L.kbbq_bam_reader_write(self.h, writer.h, d_qual, 1 if set_oq else 0, after_stream))

    def kernel_ms(self):
        a, b = ctypes.c_double(), ctypes.c_double()
        _lib.check(self.L.kbbq_bam_reader_kernel_ms(self.h, ctypes.byref(a), ctypes.byref(b)))
        return dict(inflate=a.value, index=b.value)
