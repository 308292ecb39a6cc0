"""Seeded synthetic reads: the host twin of the k_synth kernel.

Counter-based, so any read can be generated independently on the host (numpy)
or on the device, bit for bit the same:
  genome base at p          = hash(seed, 0, p) >> 62                        (i.i.d. uniform ACGT)
  read r                    : start = (hash(seed,1,r) >> 1) % (G-L+1), strand = hash(seed,1,r) & 1
  base c of read r          : quality from a per-cycle discrete profile {2,12,22,32,37},
                              substitution with probability 10^(-q/10), N (q=2) at n_per_million
  read group / second flag  : hash(seed,6,r) % n_rg ; r & 1 when paired
"""
import ctypes

import numpy as np

from . import _lib

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_IDX = np.uint64(0xD1342543DE82EF95)


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _hash(seed, stream, idx):
    with np.errstate(over="ignore"):
        key = _mix64(np.uint64(seed) + np.uint64(stream) * _GOLD)
        return _mix64(key + np.asarray(idx, dtype=np.uint64) * _IDX)


def synth_params(seed, genome_len, n_reads, read_len=150, n_rg=1, paired=False, n_per_million=100):
    sp = _lib.SynthParams()
    sp.seed = seed
    sp.genome_len = genome_len
    sp.n_reads = n_reads
    sp.read_len = read_len
    sp.n_rg = n_rg
    sp.paired = 1 if paired else 0
    sp.n_per_million = n_per_million
    return sp


def tables(sp):
    qcum = np.zeros((sp.read_len, 4), dtype=np.uint32)
    errthr = np.zeros(94, dtype=np.uint32)
    _lib.check(_lib.lib().kbbq_synth_tables(ctypes.byref(sp), qcum.ctypes.data_as(_lib.c_u32p),
                                            errthr.ctypes.data_as(_lib.c_u32p)))
    return qcum, errthr


def generate(sp, first_read=0, n=None):
    """Reads [first_read, first_read+n) as host arrays: dict(seq ASCII, qual, off, rg, second)."""
    if n is None:
        n = int(sp.n_reads) - first_read
    L, G, seed = int(sp.read_len), int(sp.genome_len), int(sp.seed)
    qcum, errthr = tables(sp)
    r = np.arange(first_read, first_read + n, dtype=np.uint64)
    h1 = _hash(seed, 1, r)
    start = (h1 >> np.uint64(1)) % np.uint64(G - L + 1)
    strand = (h1 & np.uint64(1)).astype(bool)
    c = np.arange(L, dtype=np.uint64)
    gpos = np.where(strand[:, None], start[:, None] + np.uint64(L - 1) - c[None, :], start[:, None] + c[None, :])
    b = (_hash(seed, 0, gpos) >> np.uint64(62)).astype(np.int64)
    b = np.where(strand[:, None], 3 - b, b)
    idx = r[:, None] * np.uint64(L) + c[None, :]
    hb = _hash(seed, 3, idx)
    uq = (hb & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ue = (hb >> np.uint64(32)).astype(np.uint32)
    q = np.full((n, L), 37, dtype=np.int64)
    q = np.where(uq < qcum[None, :, 3], 32, q)
    q = np.where(uq < qcum[None, :, 2], 22, q)
    q = np.where(uq < qcum[None, :, 1], 12, q)
    q = np.where(uq < qcum[None, :, 0], 2, q)
    is_err = ue < errthr[q]
    sub = (_hash(seed, 4, idx) % np.uint64(3)).astype(np.int64)
    b = np.where(is_err, (b + 1 + sub) & 3, b)
    n_thr = (int(sp.n_per_million) << 20) // 1000000
    is_n = (_hash(seed, 5, idx) & np.uint64(0xFFFFF)) < np.uint64(n_thr)
    q = np.where(is_n, 2, q)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[b]
    seq = np.where(is_n, np.uint8(ord("N")), seq)
    rg = ((_hash(seed, 6, r) >> np.uint64(32)) % np.uint64(sp.n_rg)).astype(np.int32)
    second = (r & np.uint64(1)).astype(np.uint8) if sp.paired else np.zeros(n, dtype=np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    return dict(seq=np.ascontiguousarray(seq.reshape(-1), dtype=np.uint8),
                qual=np.ascontiguousarray(q.reshape(-1), dtype=np.uint8), off=off, rg=rg, second=second)
