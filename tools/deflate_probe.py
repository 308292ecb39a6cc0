#!/usr/bin/env python3
"""tools/deflate_probe.py [MB] -- where k_deflate's time goes: run with KBBQ_LIB=tools/build/libkbbq_prof.so (bgzf_device.hip
compiled with -DKBBQ_DFL_PROFILE: cycle counter read between the phases of every block, summed over the wavefronts)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbbq_amd import _lib, bgzf  # noqa: E402
from inflate_probe import fastq_text  # noqa: E402

NAMES = ["reset", "crc bytes", "sweep 1: hash table", "sweep 2: match lengths", "sweep 3: parse + counts", "codes + header (lane 0)",
         "block size + token bits", "crc join + framing"]


def bam_bytes(n_reads, rng):
    """records shaped like the ones kbbq --io-test synth-bam writes with `oq`: 150 bases as 4-bit codes, the quality field all 11,
    RG:Z and the qualities as OQ:Z"""
    out = []
    nib = np.array([1, 2, 4, 8], dtype=np.uint8)
    for i in range(n_reads):
        name = b"read%09d\0" % i
        seq = (nib[rng.integers(0, 4, 75)] << 4 | nib[rng.integers(0, 4, 75)]).astype(np.uint8).tobytes()
        oq = (rng.choice(np.array([2, 12, 23, 27, 32, 37, 40], dtype=np.uint8), 150, p=[.02, .04, .08, .1, .16, .4, .2]) + 33).astype(np.uint8).tobytes()
        body = (np.array([0, int(rng.integers(0, 1 << 28))], dtype="<i4").tobytes() + bytes([len(name), 60, 0x49, 0x12]) +
                np.array([1, 16 if i & 1 else 0], dtype="<u2").tobytes() + np.array([150, -1, -1, 0], dtype="<i4").tobytes() +
                name + np.array([150 << 4], dtype="<u4").tobytes() + seq + b"\x0b" * 150 + b"RGZgrp0\0" + b"OQZ" + oq + b"\0")
        out.append(np.array([len(body)], dtype="<u4").tobytes() + body)
    return b"".join(out)


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    if len(sys.argv) > 3 and sys.argv[3] == "bam":      # BAM records instead of FASTQ text
        text = np.tile(np.frombuffer(bam_bytes(mb * 1000000 // 450, np.random.default_rng(5)), dtype=np.uint8), reps)
    else:
        text = np.tile(np.frombuffer(fastq_text(mb * 1000000 // 330, np.random.default_rng(5)), dtype=np.uint8), reps)
    if os.environ.get("KBBQ_PROBE_ZLIB"):      # the same text through zlib level 6 in BGZF-sized blocks (what bgzip / htslib write)
        import zlib
        one = text[:text.size // reps].tobytes()
        z = sum(len(zlib.compress(one[i:i + 0xff00], 6)) - 6 + 26 for i in range(0, len(one), 0xff00))
        print("zlib level 6, per 0xff00-byte block: %.0f MB for %.0f MB" % (z * reps / 1e6, text.size / 1e6), flush=True)
    w = bgzf.BgzfWriter(0)
    L = _lib.lib()
    have = hasattr(L, "kbbq_bgzf_debug_profile")
    for rep in range(3):
        before = w.kernel_ms()["deflate"]
        comp = w.compress(text)
        ms = w.kernel_ms()["deflate"] - before
        print("deflate %.1f ms for %.0f MB -> %.0f MB: %.1f GB/s" % (ms, text.size / 1e6, len(comp) / 1e6, text.size / ms / 1e6), flush=True)
        if have:
            out = (ctypes.c_uint64 * 16)()
            L.kbbq_bgzf_debug_profile.restype = ctypes.c_int
            L.kbbq_bgzf_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            _lib.check(L.kbbq_bgzf_debug_profile(w.h, out))
            tot = sum(out[:8])
            n_blocks = (text.size + 0xff00 - 1) // 0xff00
            for i, nm in enumerate(NAMES):
                print("   %-28s %5.1f %%   %9.0f cycles per block" % (nm, 100.0 * out[i] / max(1, tot), out[i] / n_blocks))
            print("   total %.0f cycles per block;  inside sweep 3: the parse loop %.0f, counts + token stores %.0f" % (tot / n_blocks, out[8] / n_blocks, out[9] / n_blocks), flush=True)


if __name__ == "__main__":
    main()
