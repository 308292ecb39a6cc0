"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under kbbq_amd/ does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

u8p = ctypes.POINTER(ctypes.c_uint8)
u32p = ctypes.POINTER(ctypes.c_uint32)
i32p = ctypes.POINTER(ctypes.c_int32)
u64p = ctypes.POINTER(ctypes.c_uint64)
c_u64 = ctypes.c_uint64
c_u32 = ctypes.c_uint32
c_dbl = ctypes.c_double
NQ = 256


_NATIVE = False


def build(force=False, native=False):
    """liboracle.so (portable -O2: what travels to the GPU box) or liboracle_native.so (-O2 -march=native, the
    reference's own flags, CMakeLists.txt:19 -- built on the machine that runs it, never shipped)."""
    name = "liboracle_native.so" if native else "liboracle.so"
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "kbbq_oracle.cc")
    if force or native or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-B" if (force or native) else "-s", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    return so


def use_native():
    """Make lib() load a -march=native build compiled on THIS host (bench.py's cpu_baseline leg).  Must be called
    before the first lib(); returns False (and keeps the portable build) when the compiler is not there."""
    global _NATIVE
    assert _LIB is None, "use_native() must come before the first lib()"
    try:
        build(native=True)
        _NATIVE = True
    except (OSError, subprocess.CalledProcessError):
        _NATIVE = False
    return _NATIVE


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build(native=_NATIVE))
        L.ko_new.restype = ctypes.c_void_p
        L.ko_new.argtypes = [ctypes.c_int, ctypes.c_char_p, c_u32, c_u64, c_dbl, c_dbl, c_u64]
        L.ko_free.argtypes = [ctypes.c_void_p]
        for name in ("ko_filter_bits", "ko_filter_bits_unblocked", "ko_filter_random_seed", "ko_filter_inserted"):
            getattr(L, name).restype = c_u64
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_int]
        for name in ("ko_filter_nhash", "ko_filter_nsalt"):
            getattr(L, name).restype = c_u32
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ko_filter_salts.restype = u32p
        L.ko_filter_salts.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ko_filter_table.restype = u64p
        L.ko_filter_table.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ko_filter_patterns.restype = u64p
        L.ko_filter_patterns.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ko_filter_insert_key.argtypes = [ctypes.c_void_p, ctypes.c_int, c_u64]
        L.ko_filter_contains_key.argtypes = [ctypes.c_void_p, ctypes.c_int, c_u64]
        L.ko_filter_block_of.restype = c_u64
        L.ko_filter_block_of.argtypes = [ctypes.c_void_p, ctypes.c_int, c_u64]
        L.ko_filter_pattern_of.restype = c_u64
        L.ko_filter_pattern_of.argtypes = [ctypes.c_void_p, ctypes.c_int, c_u64]
        L.ko_sample.argtypes = [ctypes.c_void_p, c_u64, u8p, u64p]
        L.ko_draws.restype = c_u64
        L.ko_draws.argtypes = [ctypes.c_void_p]
        L.ko_sampled_fpr.restype = c_dbl
        L.ko_sampled_fpr.argtypes = [ctypes.c_void_p]
        L.ko_compute_thresholds.argtypes = [ctypes.c_void_p, i32p, ctypes.c_char_p, ctypes.c_size_t]
        L.ko_set_thresholds.argtypes = [ctypes.c_void_p, i32p]
        L.ko_trusted.argtypes = [ctypes.c_void_p, c_u64, u8p, u8p, u64p, u8p]
        L.ko_errors.argtypes = [ctypes.c_void_p, c_u64, u8p, u8p, u64p, i32p, u8p, u8p, ctypes.c_int]
        L.ko_tally.argtypes = [ctypes.c_void_p, c_u64, u8p, u8p, u64p, i32p, u8p, u8p]
        for name in ("ko_cov_nrg", "ko_cov_ncycle"):
            getattr(L, name).restype = c_u64
            getattr(L, name).argtypes = [ctypes.c_void_p]
        for name in ("ko_cov_rg", "ko_cov_q", "ko_cov_cycle", "ko_cov_dinuc"):
            getattr(L, name).restype = u64p
            getattr(L, name).argtypes = [ctypes.c_void_p]
        L.ko_cov_set.argtypes = [ctypes.c_void_p, c_u64, c_u64, u64p, u64p, u64p, u64p]
        L.ko_train.argtypes = [ctypes.c_void_p]
        for name in ("ko_dq_meanq", "ko_dq_rg", "ko_dq_q", "ko_dq_cycle", "ko_dq_dinuc"):
            getattr(L, name).restype = i32p
            getattr(L, name).argtypes = [ctypes.c_void_p]
        L.ko_recalibrate.argtypes = [ctypes.c_void_p, c_u64, u8p, u8p, u64p, i32p, u8p, u8p]
        L.ko_optimal_parameters.argtypes = [c_u64, c_dbl, u32p, u64p]
        L.ko_hash_ap8.restype = c_u32
        L.ko_hash_ap8.argtypes = [c_u64, c_u32]
        L.ko_rng_outputs.argtypes = [c_u32, c_u64, u64p]
        L.ko_bernoulli_count.restype = c_u64
        L.ko_bernoulli_count.argtypes = [c_u32, c_dbl, c_u64]
        L.ko_bernoulli_one.argtypes = [c_u64, c_dbl]
        L.ko_kmer.argtypes = [ctypes.c_int, ctypes.c_char_p, u64p, u64p]
        L.ko_thresholds.argtypes = [ctypes.c_int, ctypes.c_char_p, i32p]
        L.ko_effective_fpp_text.argtypes = [c_u64, c_u64, c_u32, ctypes.c_char_p, ctypes.c_size_t]
        L.ko_phit_text.argtypes = [c_u64, c_u64, c_u32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        L.ko_normal_prior_text.argtypes = [c_u64, ctypes.c_char_p, ctypes.c_size_t]
        L.ko_log_binom_pmf_text.argtypes = [c_u64, c_u64, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        L.ko_p_to_q.argtypes = [ctypes.c_char_p]
        L.ko_base_code.restype = ctypes.c_uint8
        L.ko_base_code.argtypes = [ctypes.c_uint8]
        L.ko_map_q_minus_prior.argtypes = [c_u64, c_u64, ctypes.c_int]
        _LIB = L
    return _LIB


def ref_lib():
    """oracle/_ref/libref.so (stand-alone parts of the reference) or None."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libref.so")
        if not os.path.exists(so):
            if os.path.isdir("/root/reference/include/minionrng"):
                subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
            if not os.path.exists(so):
                return None
        R = ctypes.CDLL(so)
        R.ref_rng_outputs.argtypes = [c_u32, c_u64, u64p]
        R.ref_bernoulli_count.restype = c_u64
        R.ref_bernoulli_count.argtypes = [c_u32, c_dbl, c_u64]
        R.ref_bernoulli_bits.argtypes = [c_u32, c_dbl, c_u64, u8p]
        R.ref_optimal_parameters.argtypes = [c_u64, c_dbl, c_u64, u32p, u64p]
        R.ref_salts.argtypes = [c_u32, c_u64, u32p, u64p]
        R.ref_hash_ap.restype = c_u32
        R.ref_hash_ap.argtypes = [ctypes.c_char_p, c_u64, c_u32]
        R.ref_pattern_table.argtypes = [c_u32, c_u32, u64p]
        _REF = R
    return _REF


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


DEFAULT_BLOOM_SEED = 0xA5A5A5A55A5A5A5A


class Oracle:
    """The reference's four passes (kbbq.cc:258-457) on in-memory reads.

    Reads are given as: seq (uint8 ASCII, concatenated), qual (uint8 phred, no
    +33), off (uint64, n+1), rg (int32 per read), second (uint8 per read).
    """

    def __init__(self, k, alpha, seed, approx_kmers, fpr_sampled=float(np.longdouble("0.01")),
                 fpr_trusted=float(np.longdouble("0.0005")), bloom_seed=DEFAULT_BLOOM_SEED):
        self.L = lib()
        self.k = k
        # alpha may be a numpy longdouble: it crosses as decimal text with full precision
        alpha_text = np.format_float_scientific(np.longdouble(alpha), precision=25, unique=False).encode()
        self.h = ctypes.c_void_p(self.L.ko_new(k, alpha_text, seed, approx_kmers, fpr_sampled, fpr_trusted, bloom_seed))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ko_free(self.h)
            self.h = None

    # ---- filters
    def filter_info(self, which):
        L, h = self.L, self.h
        ns = L.ko_filter_nsalt(h, which)
        return dict(bits=L.ko_filter_bits(h, which), bits_unblocked=L.ko_filter_bits_unblocked(h, which),
                    nhash=L.ko_filter_nhash(h, which), nsalt=ns, random_seed=L.ko_filter_random_seed(h, which),
                    inserted=L.ko_filter_inserted(h, which),
                    salts=np.ctypeslib.as_array(L.ko_filter_salts(h, which), (ns,)).copy())

    def filter_table(self, which):
        n = self.L.ko_filter_bits(self.h, which) // 64
        return np.ctypeslib.as_array(self.L.ko_filter_table(self.h, which), (n,))

    def filter_patterns(self, which):
        return np.ctypeslib.as_array(self.L.ko_filter_patterns(self.h, which), (65536 * 8,))

    # ---- passes
    def sample(self, seq, off):
        self.L.ko_sample(self.h, len(off) - 1, _p(seq, u8p), _p(off, u64p))

    def compute_thresholds(self):
        thr = np.zeros(self.k + 1, dtype=np.int32)
        buf = ctypes.create_string_buffer(64)
        too_high = self.L.ko_compute_thresholds(self.h, _p(thr, i32p), buf, 64)
        return thr, buf.value.decode(), bool(too_high)

    def set_thresholds(self, thr):
        thr = np.ascontiguousarray(thr, dtype=np.int32)
        self.L.ko_set_thresholds(self.h, _p(thr, i32p))

    def sampled_fpr(self):
        return self.L.ko_sampled_fpr(self.h)

    def trusted(self, seq, qual, off, want_errors=False):
        err = np.zeros(len(seq), dtype=np.uint8) if want_errors else None
        self.L.ko_trusted(self.h, len(off) - 1, _p(seq, u8p), _p(qual, u8p), _p(off, u64p), _p(err, u8p))
        return err

    def errors(self, seq, qual, off, rg=None, second=None, tally=True):
        err = np.zeros(len(seq), dtype=np.uint8)
        self.L.ko_errors(self.h, len(off) - 1, _p(seq, u8p), _p(qual, u8p), _p(off, u64p), _p(rg, i32p),
                         _p(second, u8p), _p(err, u8p), 1 if tally else 0)
        return err

    def tally(self, seq, qual, off, rg, second, err):
        self.L.ko_tally(self.h, len(off) - 1, _p(seq, u8p), _p(qual, u8p), _p(off, u64p), _p(rg, i32p),
                        _p(second, u8p), _p(err, u8p))

    def covariates(self):
        L, h = self.L, self.h
        R, C = L.ko_cov_nrg(h), L.ko_cov_ncycle(h)
        if R == 0:
            return dict(R=0, C=0)
        return dict(R=R, C=C,
                    rg=np.ctypeslib.as_array(L.ko_cov_rg(h), (R, 2)).copy(),
                    q=np.ctypeslib.as_array(L.ko_cov_q(h), (R, NQ, 2)).copy(),
                    cycle=np.ctypeslib.as_array(L.ko_cov_cycle(h), (R, NQ, 2, C, 2)).copy(),
                    dinuc=np.ctypeslib.as_array(L.ko_cov_dinuc(h), (R, NQ, 16, 2)).copy())

    def set_covariates(self, cov):
        a = [np.ascontiguousarray(cov[k], dtype=np.uint64) for k in ("rg", "q", "cycle", "dinuc")]
        self.L.ko_cov_set(self.h, cov["R"], cov["C"], *[_p(x, u64p) for x in a])

    def train(self):
        L, h = self.L, self.h
        L.ko_train(h)
        R, C = L.ko_cov_nrg(h), L.ko_cov_ncycle(h)
        return dict(R=R, C=C,
                    meanq=np.ctypeslib.as_array(L.ko_dq_meanq(h), (R,)).copy(),
                    rg=np.ctypeslib.as_array(L.ko_dq_rg(h), (R,)).copy(),
                    q=np.ctypeslib.as_array(L.ko_dq_q(h), (R, NQ)).copy(),
                    cycle=np.ctypeslib.as_array(L.ko_dq_cycle(h), (R, NQ, 2, C)).copy(),
                    dinuc=np.ctypeslib.as_array(L.ko_dq_dinuc(h), (R, NQ, 16)).copy())

    def recalibrate(self, seq, qual, off, rg=None, second=None):
        out = np.zeros(len(seq), dtype=np.uint8)
        self.L.ko_recalibrate(self.h, len(off) - 1, _p(seq, u8p), _p(qual, u8p), _p(off, u64p), _p(rg, i32p),
                              _p(second, u8p), _p(out, u8p))
        return out

    def run_all(self, seq, qual, off, rg=None, second=None):
        """All four passes; returns a dict of every intermediate."""
        out = {}
        self.sample(seq, off)
        out["sampled_inserted"] = self.filter_info(0)["inserted"]
        thr, p_text, too_high = self.compute_thresholds()
        out.update(thresholds=thr, p_text=p_text, fpr_too_high=too_high, fpr=self.sampled_fpr())
        out["infer_errors"] = self.trusted(seq, qual, off, want_errors=True)
        out["trusted_inserted"] = self.filter_info(1)["inserted"]
        out["errors"] = self.errors(seq, qual, off, rg, second, tally=True)
        out["cov"] = self.covariates()
        out["dq"] = self.train()
        out["recal"] = self.recalibrate(seq, qual, off, rg, second)
        return out


def fnv1a64(buf):
    h = 0xCBF29CE484222325
    for b in bytes(buf):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h
