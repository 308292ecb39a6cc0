"""Randomised differential runs of k_inflate (kbbq_amd/csrc/bgzf_inflate.h) against zlib: arbitrary bytes -- not only FASTQ text
-- through every kind of DEFLATE stream zlib can be made to write, as BGZF members of random sizes, inflated on the device by
kbbq_fastq_reader_inflate (what the host parsers of the command line and the BAM reader sit on) and compared byte for byte.

  data      random bytes (stored blocks, or codes of 8 bits and more), draws from a few symbols (short codes: three literals
            per table entry), geometric draws over all 256 values (codes longer than the 9-bit table: the bounds), runs and
            short periods (matches that run into themselves, distance 1 .. 300), long-range repeats (distances up to 32768:
            read back from HBM), and mixtures
  streams   levels 0-9, strategies default / filtered / Huffman-only / RLE / fixed, memLevel 1-9 (small ones cut a member into
            many DEFLATE blocks), window 9-15 bits, sync and full flushes inside a member (empty stored blocks), members of
            1 byte to 0xff00, empty members in the middle of the file

`python tests/fuzz_inflate.py [N_CASES] [SEED]` prints one line per case; tests/test_bgzf_gpu.py runs a short round."""
import ctypes
import struct
import sys
import zlib

import numpy as np


def make_data(rng, n):
    kind = int(rng.randint(0, 7))
    if kind == 0:
        return rng.randint(0, 256, n).astype(np.uint8).tobytes()
    if kind == 1:
        k = int(rng.randint(1, 40))
        return rng.choice(rng.randint(0, 256, k).astype(np.uint8), n).tobytes()
    if kind == 2:
        p = float(rng.uniform(0.02, 0.3))
        return (rng.geometric(p, n) - 1).clip(0, 255).astype(np.uint8).tobytes()
    if kind == 3:
        out = bytearray()
        while len(out) < n:
            period = int(rng.choice([1, 2, 3, 4, 7, 8, 31, 63, 64, 65, 100, 257, 258, 259, 300]))
            unit = rng.randint(0, 256, period).astype(np.uint8).tobytes()
            out += (unit * (int(rng.randint(1, 3000)) // period + 1))[:int(rng.randint(1, 3000))]
        return bytes(out[:n])
    if kind == 4:
        piece = rng.randint(0, 256, int(rng.randint(100, 20000))).astype(np.uint8).tobytes()
        out = bytearray()
        while len(out) < n:
            out += piece[:int(rng.randint(4, len(piece) + 1))]
            out += rng.randint(0, 256, int(rng.choice([0, 5, 1400, 1600, 12000, 31000, 32700, 32768 - len(piece) % 50]))).astype(np.uint8).tobytes()
        return bytes(out[:n])
    if kind == 5:
        words = [bytes(rng.randint(97, 123, int(rng.randint(2, 12))).astype(np.uint8)) for _ in range(int(rng.randint(5, 400)))]
        out = bytearray()
        while len(out) < n:
            out += words[int(rng.randint(0, len(words)))] + b" "
        return bytes(out[:n])
    a = make_data(rng, n // 2 + 1)
    b = make_data(rng, n // 2 + 1)
    return (a + b)[:n]


def member(raw, rng):
    level = int(rng.randint(0, 10))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    co = zlib.compressobj(level, zlib.DEFLATED, -int(rng.randint(9, 16)), int(rng.randint(1, 10)), strategy)
    body = b""
    at = 0
    while at < len(raw):
        step = len(raw) - at if rng.rand() < 0.6 else int(rng.randint(1, len(raw) - at + 1))
        body += co.compress(raw[at:at + step])
        at += step
        if at < len(raw):
            body += co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])))
    body += co.flush()
    if len(body) + 26 > 65536:
        return None
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body +
            struct.pack("<II", zlib.crc32(raw), len(raw)))


def make_file(rng):
    total = int(rng.choice([1, 70, 5000, 70000, 300000, 1500000]))
    data = make_data(rng, total)
    out, at = [], 0
    while at < len(data):
        size = int(rng.choice([1, 2, 100, 4096, 20000, 0xff00, int(rng.randint(1, 0xff01))]))
        raw = data[at:at + size]
        m = member(raw, rng)
        while m is None:      # (incompressible bytes at a level that expands them: a smaller member)
            raw = raw[:len(raw) // 2]
            m = member(raw, rng)
        out.append(m)
        at += len(raw)
        if rng.rand() < 0.05:
            out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")
    return data, b"".join(out)


def run(n_cases, seed, verbose=False):
    from kbbq_amd import _lib, bgzf
    L = _lib.lib()
    r = bgzf.FastqReader(0)
    rng = np.random.RandomState(seed)
    n_bytes = 0
    for case in range(n_cases):
        data, comp = make_file(rng)
        got = bytearray()
        src = np.frombuffer(comp, dtype=np.uint8)
        at = 0
        # the file in pieces that cut members anywhere; what a piece leaves over is fed again in front of the next
        while at < len(comp):
            step = len(comp) - at if rng.rand() < 0.5 else min(len(comp) - at, int(rng.randint(70000, 400000)))
            piece = np.ascontiguousarray(src[at:at + step])
            out = np.zeros(min(1 << 27, 64 * step + (1 << 16)), dtype=np.uint8)      # (what does not fit is left for the next call)
            consumed, produced = ctypes.c_uint64(0), ctypes.c_uint64(0)
            _lib.check(L.kbbq_fastq_reader_inflate(r.h, piece.ctypes.data, piece.size, out.ctypes.data, out.size,
                                                   ctypes.byref(consumed), ctypes.byref(produced)))
            assert consumed.value > 0, (case, at, step)      # (a piece is never shorter than a member)
            got += out[:produced.value].tobytes()
            at += consumed.value
        assert bytes(got) == data, "case %d (seed %d): %d bytes expected, %d inflated, first difference at %d" % (
            case, seed, len(data), len(got), next((i for i, (x, y) in enumerate(zip(got, data)) if x != y), min(len(got), len(data))))
        n_bytes += len(data)
        if verbose:
            print("case %d: %d bytes in %d of BGZF ok" % (case, len(data), len(comp)), flush=True)
    r.close()
    return n_bytes


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 71
    total = run(n, seed, verbose=True)
    print("%d cases, %d bytes, 0 mismatches" % (n, total))
