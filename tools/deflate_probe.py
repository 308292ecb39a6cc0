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


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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
