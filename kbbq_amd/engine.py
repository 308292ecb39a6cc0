"""Host-side mirror of the reference's pass interface over the C ABI.

The method names are the reference's own (recalibrateutils.hh:29-44,
covariateutils.hh:171): subsample_kmers, find_trusted_kmers,
get_covariatedata, get_dqs, recalibrate.  Each one submits a batch to the HIP
engine; nothing here computes on the CPU and there is no fallback path.
"""
import ctypes

import numpy as np

from . import _lib
from .reads import ReadBatch, unpack_bits

NQ = _lib.NQ


def long_double_text(x):
    """Decimal text of a numpy longdouble (the ABI passes the reference's long double alpha as text)."""
    return np.format_float_scientific(np.longdouble(x), precision=25, unique=False).encode()


def default_fprs():
    # kbbq.cc:155-156: long double literals narrowed to the double Bloom ctor argument
    return float(np.longdouble("0.01")), float(np.longdouble("0.0005"))


def plan_parameters(genomelen, coverage=0, alpha=None, seqlen=None):
    """kbbq.cc:227-264: coverage, alpha (long double) and approx_kmers from the CLI inputs.
    alpha may be given as text, which is parsed straight to long double like std::stold (kbbq.cc:122)."""
    if alpha is None or alpha == 0:
        if coverage == 0:
            coverage = int(seqlen // genomelen)
        alpha_ld = np.longdouble(7.0) / np.longdouble(coverage)
    else:
        alpha_ld = np.longdouble(alpha)
        if coverage == 0:
            coverage = int(np.longdouble(7.0) / alpha_ld)
    approx = int(np.longdouble(genomelen * coverage) * alpha_ld)
    return alpha_ld, coverage, approx


class DeviceReads:
    """A batch resident in HBM (uploaded host batch or device-generated synthetic reads)."""

    def __init__(self, engine, creads, max_len):
        self.engine = engine
        self.c = creads
        self.n_reads = int(creads.n_reads)
        self.n_bases = int(creads.n_bases)
        self.max_len = max_len

    def set_hints(self, sampled_ptr, trusted_ptr):
        """Attach the two caller-owned, zeroed hint bit arrays (n_bases/64+2 u64 words each)."""
        self.c.hint_sampled = sampled_ptr
        self.c.hint_trusted = trusted_ptr

    def free(self):
        if self.c is not None and self.engine.h:
            _lib.check(self.engine.L.kbbq_reads_free(self.engine.h, ctypes.byref(self.c)))
        self.c = None

    def view(self, first_read, n_reads):
        """A sub-range of a uniform device batch (no copy).  first_read must keep the 2-bit words aligned."""
        assert not self.c.offsets, "views need a uniform batch"
        rl = int(self.c.read_len)
        b0 = first_read * rl
        assert b0 % 64 == 0, "sub-batches must start on a 64-base boundary"
        v = _lib.Reads()
        v.n_reads = n_reads
        v.n_bases = n_reads * rl
        v.bases = self.c.bases + b0 // 4
        v.nmask = self.c.nmask + b0 // 8
        v.qual = self.c.qual + b0
        v.offsets = None
        v.flags = (self.c.flags + first_read) if self.c.flags else None
        v.rg = (self.c.rg + 2 * first_read) if self.c.rg else None
        v.read_len = rl
        v.on_device = 1
        v.hint_sampled = (self.c.hint_sampled + b0 // 8) if self.c.hint_sampled else None
        v.hint_trusted = (self.c.hint_trusted + b0 // 8) if self.c.hint_trusted else None
        v.offcase = (self.c.offcase + b0 // 8) if self.c.offcase else None
        d = DeviceReads(self.engine, v, self.max_len)
        d.free = lambda: None
        return d


class Engine:
    def __init__(self, k, alpha, seed, approx_kmers, n_rg=1, max_read_len=160, device=0, fpr_sampled=None,
                 fpr_trusted=None, bloom_seed=_lib.DEFAULT_BLOOM_SEED, profile=False, flags=0, tune=None):
        """flags: KBBQ_F_* switches OR-ed together; tune: {"bucket_records": n, "pass4_piece": n} (kbbq_engine_tune)."""
        self.L = _lib.lib()
        fs, ft = default_fprs()
        self.alpha_ld = np.longdouble(alpha)
        p = _lib.Params()
        p.k = k
        p.device = device
        p.alpha = float(self.alpha_ld)      # KmerSubsampler(file, k, alpha, seed): double parameter, htsiter.hh:143
        p.seed = seed & 0xFFFFFFFF
        p.n_rg = n_rg
        p.approx_kmers = approx_kmers
        p.fpr_sampled = fs if fpr_sampled is None else fpr_sampled
        p.fpr_trusted = ft if fpr_trusted is None else fpr_trusted
        p.bloom_seed = bloom_seed
        p.max_read_len = max_read_len
        p.flags = (_lib.KBBQ_F_PROFILE if profile else 0) | int(flags)
        self.params = p
        self.k = k
        self.n_rg = n_rg
        self.max_read_len = max_read_len
        h = _lib.c_vp()
        _lib.check(self.L.kbbq_engine_create(ctypes.byref(p), ctypes.byref(h)))
        self.h = h
        for name, value in (tune or {}).items():
            _lib.check(self.L.kbbq_engine_tune(self.h, name.encode(), int(value)))

    def close(self):
        if getattr(self, "h", None):
            self.L.kbbq_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @staticmethod
    def _c(batch):
        return ctypes.byref(batch.c)

    def reset(self):
        _lib.check(self.L.kbbq_engine_reset(self.h))

    def sync(self):
        _lib.check(self.L.kbbq_engine_sync(self.h))

    # ---- filters
    def filter_info(self, which):
        fi = _lib.FilterInfo()
        _lib.check(self.L.kbbq_filter_info_get(self.h, which, ctypes.byref(fi)))
        return dict(bits=fi.bits, bits_unblocked=fi.bits_unblocked, n_blocks=fi.n_blocks, table_bytes=fi.table_bytes,
                    random_seed=fi.random_seed,
                    inserted=fi.inserted, nhash=fi.n_hash, nsalt=fi.n_salt,
                    salts=np.array(fi.salt[:fi.n_salt], dtype=np.uint32))

    def filter_table(self, which):
        n = self.filter_info(which)["n_blocks"] * 8
        out = np.zeros(n, dtype=np.uint64)
        _lib.check(self.L.kbbq_filter_download(self.h, which, out.ctypes.data_as(_lib.c_u64p), n))
        return out

    def filter_patterns(self, which):
        out = np.zeros(65536 * 8, dtype=np.uint64)
        _lib.check(self.L.kbbq_filter_patterns_download(self.h, which, out.ctypes.data_as(_lib.c_u64p)))
        return out

    # ---- staging
    def upload(self, batch):
        dev = _lib.Reads()
        _lib.check(self.L.kbbq_reads_upload(self.h, ctypes.byref(batch.c), ctypes.byref(dev)))
        return DeviceReads(self, dev, batch.max_len)

    def synth_reads(self, sp, first_read, n):
        dev = _lib.Reads()
        _lib.check(self.L.kbbq_synth_reads(self.h, ctypes.byref(sp), first_read, n, ctypes.byref(dev)))
        return DeviceReads(self, dev, int(sp.read_len))

    def download(self, dreads):
        """Device batch -> dict of host arrays (tests)."""
        import torch
        out = {}
        nb, nr = dreads.n_bases, dreads.n_reads
        for name, ptr, n, dt in (("bases", dreads.c.bases, nb // 32 + 1, np.uint64),
                                 ("nmask", dreads.c.nmask, nb // 64 + 1, np.uint64), ("qual", dreads.c.qual, nb, np.uint8),
                                 ("flags", dreads.c.flags, nr, np.uint8), ("rg", dreads.c.rg, nr, np.uint16)):
            if not ptr:
                out[name] = None
                continue
            t = device_tensor(ptr, n * np.dtype(dt).itemsize, torch.uint8, dreads.engine.params.device)
            out[name] = t.cpu().numpy().view(dt).copy()
        return out

    # ---- pass 1
    def subsample_kmers(self, batch, first_kmer_ordinal=0):
        _lib.check(self.L.kbbq_sample_batch(self.h, self._c(batch), first_kmer_ordinal))

    def count_kmer_positions(self, batch):
        out = ctypes.c_uint64()
        _lib.check(self.L.kbbq_count_kmer_positions(self.h, self._c(batch), ctypes.byref(out)))
        return out.value

    def sample_finish(self):
        out = ctypes.c_uint64()
        _lib.check(self.L.kbbq_sample_finish(self.h, ctypes.byref(out)))
        return out.value

    # ---- between passes
    def compute_thresholds(self):
        thr = np.zeros(self.k + 1, dtype=np.int32)
        fpr = ctypes.c_double()
        buf = ctypes.create_string_buffer(64)
        rc = _lib.check(self.L.kbbq_compute_thresholds(self.h, long_double_text(self.alpha_ld),
                                                       thr.ctypes.data_as(_lib.c_i32p), ctypes.byref(fpr), buf, 64))
        return thr, fpr.value, buf.value.decode(), bool(rc)

    def set_thresholds(self, thr):
        thr = np.ascontiguousarray(thr, dtype=np.int32)
        _lib.check(self.L.kbbq_set_thresholds(self.h, thr.ctypes.data_as(_lib.c_i32p), len(thr)))

    # ---- pass 2
    def find_trusted_kmers(self, batch, want_errors=False):
        if want_errors:
            assert isinstance(batch, ReadBatch), "error masks are returned for host batches"
            words = np.zeros(batch.n_bases // 64 + 2, dtype=np.uint64)
            _lib.check(self.L.kbbq_trusted_batch(self.h, self._c(batch), words.ctypes.data))
            return unpack_bits(words, batch.n_bases)
        _lib.check(self.L.kbbq_trusted_batch(self.h, self._c(batch), None))
        return None

    def trusted_finish(self):
        out = ctypes.c_uint64()
        _lib.check(self.L.kbbq_trusted_finish(self.h, ctypes.byref(out)))
        return out.value

    # ---- pass 3
    def get_covariatedata(self, batch, want_errors=False):
        if want_errors:
            assert isinstance(batch, ReadBatch)
            words = np.zeros(batch.n_bases // 64 + 2, dtype=np.uint64)
            _lib.check(self.L.kbbq_errors_batch(self.h, self._c(batch), words.ctypes.data))
            return unpack_bits(words, batch.n_bases)
        _lib.check(self.L.kbbq_errors_batch(self.h, self._c(batch), None))
        return None

    def tally(self, batch, error_words):
        error_words = np.ascontiguousarray(error_words, dtype=np.uint64)
        _lib.check(self.L.kbbq_tally_batch(self.h, self._c(batch), error_words.ctypes.data))

    def covariates(self):
        R, C = self.n_rg, self.max_read_len
        out = dict(R=R, C=C, rg=np.zeros((R, 2), np.uint64), q=np.zeros((R, NQ, 2), np.uint64),
                   cycle=np.zeros((R, NQ, 2, C, 2), np.uint64), dinuc=np.zeros((R, NQ, 16, 2), np.uint64))
        c = _lib.Covariates()
        c.rg, c.q, c.cycle, c.dinuc = (out[k].ctypes.data for k in ("rg", "q", "cycle", "dinuc"))
        _lib.check(self.L.kbbq_covariates_get(self.h, ctypes.byref(c)))
        return out

    # ---- model
    def get_dqs(self):
        _lib.check(self.L.kbbq_train(self.h))
        return self.dq()

    def dq(self):
        R, C = self.n_rg, self.max_read_len
        out = dict(R=R, C=C, meanq=np.zeros(R, np.int32), rg=np.zeros(R, np.int32), q=np.zeros((R, NQ), np.int32),
                   cycle=np.zeros((R, NQ, 2, C), np.int32), dinuc=np.zeros((R, NQ, 16), np.int32))
        d = _lib.Dq()
        d.meanq, d.rgdq, d.qdq, d.cycledq, d.dinucdq = (out[k].ctypes.data for k in ("meanq", "rg", "q", "cycle", "dinuc"))
        _lib.check(self.L.kbbq_dq_get(self.h, ctypes.byref(d)))
        return out

    def set_dq(self, dq):
        a = {k: np.ascontiguousarray(dq[k], dtype=np.int32) for k in ("meanq", "rg", "q", "cycle", "dinuc")}
        d = _lib.Dq()
        d.n_rg, d.n_cycle = self.n_rg, self.max_read_len
        d.meanq, d.rgdq, d.qdq, d.cycledq, d.dinucdq = (a[k].ctypes.data for k in ("meanq", "rg", "q", "cycle", "dinuc"))
        _lib.check(self.L.kbbq_set_dq(self.h, ctypes.byref(d)))

    # ---- pass 4
    def recalibrate(self, batch, out_device_ptr=None):
        if isinstance(batch, ReadBatch):
            out = np.zeros(batch.n_bases + 16, dtype=np.uint8)
            _lib.check(self.L.kbbq_recalibrate_batch(self.h, self._c(batch), out.ctypes.data))
            return out[:batch.n_bases]
        _lib.check(self.L.kbbq_recalibrate_batch(self.h, self._c(batch), out_device_ptr))
        return None

    # ---- whole pipeline on one host batch (kbbq.cc:258-457)
    def run_all(self, batch):
        out = {}
        self.subsample_kmers(batch, 0)
        out["sampled_inserted"] = self.sample_finish()
        thr, fpr, p_text, too_high = self.compute_thresholds()
        out.update(thresholds=thr, fpr=fpr, p_text=p_text, fpr_too_high=too_high)
        out["infer_errors"] = self.find_trusted_kmers(batch, want_errors=True)
        out["trusted_inserted"] = self.trusted_finish()
        out["errors"] = self.get_covariatedata(batch, want_errors=True)
        out["cov"] = self.covariates()
        out["dq"] = self.get_dqs()
        out["recal"] = self.recalibrate(batch)
        return out

    # ---- measurement
    def profile(self):
        n = ctypes.c_int32()
        arr = (_lib.ProfileEntry * 32)()
        _lib.check(self.L.kbbq_profile_get(self.h, arr, 32, ctypes.byref(n)))
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(min(n.value, 32))}

    def profile_reset(self):
        _lib.check(self.L.kbbq_profile_reset(self.h))

    def stats(self):
        out = (ctypes.c_uint64 * 9)()
        _lib.check(self.L.kbbq_stats_get(self.h, out, 9))
        return dict(corrected_reads=out[0], correction_queries=out[1], reads=out[2], infer_lookups=out[3],
                    bucket_flushes=(out[4], out[5]), bucket_direct=out[6], bucket_capacity=out[7],
                    quality_above_93=bool(out[8]))

    def stream_ptr(self):
        return self.L.kbbq_engine_stream(self.h)


class _CudaArray:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_tensor(ptr, nbytes, dtype, device=0):
    """A torch view (no copy) of engine-owned device memory, for collectives and checks."""
    import torch
    t = torch.as_tensor(_CudaArray(ptr, nbytes), device="cuda:%d" % device)
    return t.view(dtype)


def rng_state_at(seed, ordinal):
    out = (ctypes.c_uint64 * 4)()
    _lib.check(_lib.lib().kbbq_rng_state_at(seed, ordinal, out))
    return [int(x) for x in out]
