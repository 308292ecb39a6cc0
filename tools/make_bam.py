#!/usr/bin/env python3
"""tools/make_bam.py OUT.bam GENOME_LEN COVERAGE -- the reads of tools/make_fastq.py wrapped as unaligned BAM with RG:Z
tags, about half of them reverse-flagged (BASELINE configs[3]), for timing the command line.  Vectorised record
assembly: fixed-width names, 150-base reads; BGZF blocks through zlib."""
import os
import struct
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kbbq_amd import synth  # noqa: E402

out, G, cov = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
L = 150
n_reads = G * cov // L
sp = synth.synth_params(12345, G, n_reads, L, n_rg=1, paired=False, n_per_million=100)
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf(fh, data):
    for pos in range(0, len(data), 0xff00):
        chunk = data[pos:pos + 0xff00]
        co = zlib.compressobj(4, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(body) + 25) + body +
                 struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


code = np.zeros(256, dtype=np.uint8)
for ch, c in zip(b"ACGTN", (1, 2, 4, 8, 15)):
    code[ch] = c
comp = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    comp[a] = b
with open(out, "wb") as fh:
    text = b"@HD\tVN:1.6\tSO:unsorted\n"
    hdr = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", 1) + struct.pack("<I", 5) + b"chr1\0" + struct.pack("<I", G)
    bgzf(fh, hdr)
    name_len, aux = 12, b"RGZlane1\0"
    body = 32 + name_len + L // 2 + L + len(aux)
    step = 200000
    for first in range(0, n_reads, step):
        n = min(step, n_reads - first)
        d = synth.generate(sp, first, n)
        seq, qual = d["seq"].reshape(n, L), d["qual"].reshape(n, L)
        rev = ((np.arange(first, first + n) * 2654435761) >> 7) & 1 == 1
        seq = np.where(rev[:, None], comp[seq[:, ::-1]], seq)          # stored = reverse complement of what was sequenced
        qual = np.where(rev[:, None], qual[:, ::-1], qual)
        rec = np.zeros((n, 4 + body), dtype=np.uint8)
        rec[:, 0:4] = np.frombuffer(struct.pack("<I", body), dtype=np.uint8)
        rec[:, 4:12] = 0xFF                                               # refID, pos = -1
        rec[:, 12] = name_len
        rec[:, 14:16] = np.frombuffer(struct.pack("<H", 4680), dtype=np.uint8)
        flag = (4 | np.where(rev, 16, 0)).astype(np.uint16)
        rec[:, 18] = flag & 0xFF
        rec[:, 19] = flag >> 8
        rec[:, 20:24] = np.frombuffer(struct.pack("<I", L), dtype=np.uint8)
        rec[:, 24:32] = 0xFF                                              # next refID, next pos = -1
        names = np.char.zfill(np.arange(first, first + n).astype("S10"), 10)
        rec[:, 36] = ord("r")
        rec[:, 37:47] = np.frombuffer(names.tobytes(), dtype=np.uint8).reshape(n, 10)
        c = code[seq]
        rec[:, 48:48 + L // 2] = (c[:, 0::2] << 4) | c[:, 1::2]
        rec[:, 48 + L // 2:48 + L // 2 + L] = qual
        rec[:, 48 + L // 2 + L:] = np.frombuffer(aux, dtype=np.uint8)
        bgzf(fh, rec.tobytes())
    fh.write(EOF_BLOCK)
print("wrote %s: %d reads, %d bases" % (out, n_reads, n_reads * L))
