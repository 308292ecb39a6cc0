#!/usr/bin/env python3
"""tools/inflate_probe.py [MB] -- what k_inflate costs per kind of DEFLATE symbol (MI355X): the same FASTQ text as BGZF blocks
made by zlib with different windows and strategies, each inflated by the device reader, kernel time from its own events.

  huffman_only   literals alone (Z_HUFFMAN_ONLY)
  w9 .. w15      matches no further back than 512 B .. 32 KB (wbits): below 7.5 KB every match is an LDS-to-LDS copy,
                 above it the far ones are read back from HBM
  level1/level9  zlib's fast and best parsers at the full window
"""
import os
import struct
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kbbq_amd import bgzf  # noqa: E402


def fastq_text(n_reads, rng):
    genome = rng.integers(0, 4, 2_000_000, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for i in range(n_reads):
        p = int(rng.integers(0, genome.size - 150))
        s = acgt[genome[p:p + 150]].tobytes()
        q = (rng.choice(np.array([2, 12, 23, 27, 32, 37, 40], dtype=np.uint8), 150, p=[.02, .04, .08, .1, .16, .4, .2]) + 33).astype(np.uint8).tobytes()
        out.append(b"@read%d/1 sample\n%s\n+\n%s\n" % (i, s, q))
    return b"".join(out)


def bgzf_blocks(text, **kw):
    out = []
    n_sym = 0
    for at in range(0, len(text), 0xff00):
        piece = text[at:at + 0xff00]
        c = zlib.compressobj(kw.get("level", 6), zlib.DEFLATED, -kw.get("wbits", 15), 8, kw.get("strategy", zlib.Z_DEFAULT_STRATEGY))
        d = c.compress(piece) + c.flush()
        bsize = len(d) + 25
        out.append(struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, bsize) + d +
                   struct.pack("<II", zlib.crc32(piece), len(piece)))
    return b"".join(out)


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 16      # BGZF blocks stand alone: the same file several times over fills the chip
    rng = np.random.default_rng(5)
    if os.environ.get("INFLATE_DATA") == "bam":      # BAM records (tools/deflate_probe.py) instead of FASTQ text
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from deflate_probe import bam_bytes
        text = bam_bytes(mb * 1000000 // 450, rng)
    else:
        text = fastq_text(mb * 1000000 // 330, rng)
    print("text", len(text), flush=True)
    cases = [("huffman_only", dict(strategy=zlib.Z_HUFFMAN_ONLY)), ("w9", dict(wbits=9)), ("w12", dict(wbits=12)),
             ("w13", dict(wbits=13)), ("w15", dict(wbits=15)), ("level1", dict(level=1)), ("device", dict(device=True))]
    if len(sys.argv) > 3:      # only these cases (a counter run wants one kind of stream per process)
        cases = [c for c in cases if c[0] in sys.argv[3].split(",")]
    r = bgzf.FastqReader(0)
    for name, kw in cases:
        t0 = time.time()
        if kw.get("device"):      # the stream this library's own encoder writes (k_deflate)
            w = bgzf.BgzfWriter(0)
            comp = np.tile(np.frombuffer(w.compress(np.frombuffer(text, dtype=np.uint8)), dtype=np.uint8), reps)
            w.close()
        else:
            comp = np.tile(np.frombuffer(bgzf_blocks(text, **kw), dtype=np.uint8), reps)
        t_host = time.time() - t0
        best = 1e9
        for rep in range(3):
            r.rewind()
            before = r.kernel_ms()["inflate"]
            try:
                info = r.chunk(comp, True)
            except Exception as ex:      # (a build that inflates wrongly on purpose: the time is still the kernel's)
                info = dict(text_bytes=len(text) * reps)
                if rep == 0:
                    print("   (%s)" % str(ex)[:80])
            best = min(best, r.kernel_ms()["inflate"] - before)
        n_text = len(text) * reps
        assert info["text_bytes"] == n_text, (info, n_text)
        print("%-13s ratio %.2f  inflate %.2f ms  %.1f GB/s of text  %.1f GB/s of stream   (zlib on the host: %.1f s)"
              % (name, n_text / comp.size, best, n_text / best / 1e6, comp.size / best / 1e6, t_host), flush=True)


if __name__ == "__main__":
    main()
