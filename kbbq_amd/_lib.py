"""ctypes binding of libkbbq_engine.so (include/kbbq_engine.h).

The product path has no CPU fallback: if the HIP library is missing or does not
load, importing the engine fails loudly.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# KBBQ_LIB: load another build of the same library (A/B experiments with kernel variants)
LIB_PATH = os.environ.get("KBBQ_LIB") or os.path.join(_HERE, "libkbbq_engine.so")

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u16p = ctypes.POINTER(ctypes.c_uint16)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_u64 = ctypes.c_uint64
c_vp = ctypes.c_void_p

KBBQ_OK = 0
KBBQ_F_PROFILE = 1
# behavioural switches of an engine (include/kbbq_engine.h); none changes a result
KBBQ_F_NO_OVERLAP = 2
KBBQ_F_BUCKET_OFF = 4
KBBQ_F_BUCKET_ON = 8
KBBQ_F_LANE_WALK = 16
KBBQ_F_NO_FASTPATH = 32
KBBQ_F_PASS2_INORDER = 64
KBBQ_F_NO_PASS4_PIPELINE = 128
DEFAULT_BLOOM_SEED = 0xA5A5A5A55A5A5A5A
NQ = 256
MAX_READ_LEN = (1 << 23) - 1      # KBBQ_MAX_READ_LEN (include/kbbq_engine.h)


class Params(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int32), ("device", ctypes.c_int32), ("alpha", ctypes.c_double),
                ("seed", ctypes.c_uint32), ("n_rg", ctypes.c_int32), ("approx_kmers", c_u64),
                ("fpr_sampled", ctypes.c_double), ("fpr_trusted", ctypes.c_double), ("bloom_seed", c_u64),
                ("max_read_len", ctypes.c_int32), ("flags", ctypes.c_int32)]


class Reads(ctypes.Structure):
    _fields_ = [("n_reads", c_u64), ("n_bases", c_u64), ("bases", c_vp), ("nmask", c_vp), ("qual", c_vp),
                ("offsets", c_vp), ("flags", c_vp), ("rg", c_vp), ("read_len", ctypes.c_uint32),
                ("on_device", ctypes.c_int32), ("hint_sampled", c_vp), ("hint_trusted", c_vp), ("offcase", c_vp)]


class FilterInfo(ctypes.Structure):
    _fields_ = [("bits", c_u64), ("bits_unblocked", c_u64), ("n_blocks", c_u64), ("random_seed", c_u64),
                ("inserted", c_u64), ("n_hash", ctypes.c_uint32), ("n_salt", ctypes.c_uint32),
                ("salt", ctypes.c_uint32 * 128), ("table_bytes", c_u64)]


class Covariates(ctypes.Structure):
    _fields_ = [("n_rg", c_u64), ("n_cycle", c_u64), ("rg", c_vp), ("q", c_vp), ("cycle", c_vp), ("dinuc", c_vp)]


class Dq(ctypes.Structure):
    _fields_ = [("n_rg", c_u64), ("n_cycle", c_u64), ("meanq", c_vp), ("rgdq", c_vp), ("qdq", c_vp),
                ("cycledq", c_vp), ("dinucdq", c_vp)]


class SynthParams(ctypes.Structure):
    _fields_ = [("seed", c_u64), ("genome_len", c_u64), ("n_reads", c_u64), ("read_len", ctypes.c_uint32),
                ("n_rg", ctypes.c_uint32), ("paired", ctypes.c_uint32), ("n_per_million", ctypes.c_uint32)]


class FastqChunk(ctypes.Structure):
    _fields_ = [("consumed", c_u64), ("n_records", c_u64), ("n_bases", c_u64), ("longest", ctypes.c_uint32),
                ("shortest", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("n_blocks", ctypes.c_uint32), ("text_bytes", c_u64)]


class ProfileEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("launches", c_u64), ("total_ms", ctypes.c_double)]


# every symbol include/kbbq_engine.h declares: (restype, argtypes)
SYMBOLS = {
    "kbbq_engine_create": (ctypes.c_int, [ctypes.POINTER(Params), ctypes.POINTER(c_vp)]),
    "kbbq_engine_destroy": (None, [c_vp]),
    "kbbq_engine_reset": (ctypes.c_int, [c_vp]),
    "kbbq_engine_sync": (ctypes.c_int, [c_vp]),
    "kbbq_engine_stream": (c_vp, [c_vp]),
    "kbbq_engine_tune": (ctypes.c_int, [c_vp, ctypes.c_char_p, c_u64]),
    "kbbq_last_error": (ctypes.c_char_p, []),
    "kbbq_filter_info_get": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.POINTER(FilterInfo)]),
    "kbbq_filter_device_table": (c_vp, [c_vp, ctypes.c_int]),
    "kbbq_filter_device_counter": (c_vp, [c_vp, ctypes.c_int]),
    "kbbq_filter_download": (ctypes.c_int, [c_vp, ctypes.c_int, c_u64p, c_u64]),
    "kbbq_filter_patterns_download": (ctypes.c_int, [c_vp, ctypes.c_int, c_u64p]),
    "kbbq_filter_or_from": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, c_u64, c_u64]),
    "kbbq_device_or": (ctypes.c_int, [c_vp, c_vp, c_vp, c_u64]),
    "kbbq_device_or_pieces": (ctypes.c_int, [c_vp, c_vp, c_vp, c_u64, ctypes.c_int32, ctypes.c_int32]),
    "kbbq_filter_set_inserted": (ctypes.c_int, [c_vp, ctypes.c_int, c_u64]),
    "kbbq_pack_bases": (ctypes.c_int, [c_u8p, c_u64, c_u64p, c_u64p]),
    "kbbq_pack_bases_case": (ctypes.c_int, [c_u8p, c_u64, c_u64p, c_u64p, c_u64p, c_u64p]),
    "kbbq_reads_upload": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), ctypes.POINTER(Reads)]),
    "kbbq_reads_free": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads)]),
    "kbbq_sample_batch": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_u64]),
    "kbbq_count_kmer_positions": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_u64p]),
    "kbbq_sample_finish": (ctypes.c_int, [c_vp, c_u64p]),
    "kbbq_compute_thresholds": (ctypes.c_int, [c_vp, ctypes.c_char_p, c_i32p, ctypes.POINTER(ctypes.c_double),
                                               ctypes.c_char_p, ctypes.c_size_t]),
    "kbbq_set_thresholds": (ctypes.c_int, [c_vp, c_i32p, ctypes.c_int32]),
    "kbbq_trusted_batch": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_vp]),
    "kbbq_trusted_finish": (ctypes.c_int, [c_vp, c_u64p]),
    "kbbq_errors_batch": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_vp]),
    "kbbq_tally_batch": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_vp]),
    "kbbq_covariates_get": (ctypes.c_int, [c_vp, ctypes.POINTER(Covariates)]),
    "kbbq_covariates_device": (c_vp, [c_vp, c_u64p]),
    "kbbq_train": (ctypes.c_int, [c_vp]),
    "kbbq_dq_get": (ctypes.c_int, [c_vp, ctypes.POINTER(Dq)]),
    "kbbq_set_dq": (ctypes.c_int, [c_vp, ctypes.POINTER(Dq)]),
    "kbbq_recalibrate_batch": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_vp]),
    "kbbq_recalibrate_batch_host": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_vp]),
    "kbbq_sample_batch_submit": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_u64, c_u64p]),
    "kbbq_trusted_batch_submit": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_u64p]),
    "kbbq_errors_batch_submit": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_u64p]),
    "kbbq_recalibrate_batch_submit": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), c_vp, c_u64p]),
    "kbbq_batch_wait": (ctypes.c_int, [c_vp, c_u64]),
    "kbbq_reads_alloc_hints": (ctypes.c_int, [ctypes.POINTER(Reads)]),
    "kbbq_reads_free_hints": (ctypes.c_int, [ctypes.POINTER(Reads)]),
    "kbbq_device_memory": (ctypes.c_int, [ctypes.c_int32, c_u64p, c_u64p]),
    "kbbq_host_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(c_vp)]),
    "kbbq_host_free": (ctypes.c_int, [c_vp]),
    "kbbq_device_alloc": (ctypes.c_int, [c_vp, ctypes.c_size_t, ctypes.POINTER(c_vp)]),
    "kbbq_device_free": (ctypes.c_int, [c_vp, c_vp]),
    "kbbq_measure_host_link": (ctypes.c_int, [ctypes.c_int32, c_u64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "kbbq_measure_host_link_duplex": (ctypes.c_int, [ctypes.c_int32, c_u64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "kbbq_synth_tables": (ctypes.c_int, [ctypes.POINTER(SynthParams), c_u32p, c_u32p]),
    "kbbq_synth_reads": (ctypes.c_int, [c_vp, ctypes.POINTER(SynthParams), c_u64, c_u64, ctypes.POINTER(Reads)]),
    "kbbq_profile_get": (ctypes.c_int, [c_vp, ctypes.POINTER(ProfileEntry), ctypes.c_int32, c_i32p]),
    "kbbq_profile_reset": (ctypes.c_int, [c_vp]),
    "kbbq_stats_get": (ctypes.c_int, [c_vp, c_u64p, ctypes.c_int32]),
    "kbbq_host_filter_spec": (ctypes.c_int, [c_u64, ctypes.c_double, c_u64, ctypes.POINTER(FilterInfo), c_u64p]),
    "kbbq_host_block_index": (ctypes.c_uint32, [ctypes.c_uint32, c_u64]),
    "kbbq_host_blocks_squeeze": (ctypes.c_int, [c_u64p, c_u64, c_u64p]),
    "kbbq_host_blocks_expand": (ctypes.c_int, [c_u64p, c_u64, c_u64p]),
    "kbbq_host_thresholds": (ctypes.c_int, [ctypes.c_int32, c_u64, c_u64, ctypes.c_uint32, ctypes.c_char_p, c_i32p,
                                            ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ctypes.c_size_t]),
    "kbbq_host_train": (ctypes.c_int, [ctypes.POINTER(Covariates), ctypes.POINTER(Dq)]),
    "kbbq_host_bernoulli_threshold": (c_u64, [ctypes.c_double, c_i32p]),
    "kbbq_rng_state_at": (ctypes.c_int, [ctypes.c_uint32, c_u64, c_u64p]),
    # include/kbbq_bgzf.h: the BGZF writer on the device
    "kbbq_bgzf_create": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(c_vp)]),
    "kbbq_bgzf_destroy": (None, [c_vp]),
    "kbbq_bgzf_submit": (ctypes.c_int, [c_vp, c_vp, c_u64, ctypes.c_int32, c_vp]),
    "kbbq_bgzf_submit_fastq": (ctypes.c_int, [c_vp, c_vp, c_vp, c_u64, c_vp, c_vp, ctypes.c_uint32, c_vp]),
    "kbbq_bgzf_collect": (ctypes.c_int, [c_vp, ctypes.POINTER(c_vp), c_u64p, c_u64p]),
    "kbbq_bgzf_eof_block": (c_vp, []),
    "kbbq_bgzf_bound": (c_u64, [c_u64]),
    "kbbq_bgzf_kernel_ms": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "kbbq_host_bgzf_compress": (ctypes.c_int, [c_vp, c_u64, c_vp, c_u64, c_u64p]),
    "kbbq_fastq_reader_create": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(c_vp)]),
    "kbbq_fastq_reader_destroy": (None, [c_vp]),
    "kbbq_fastq_reader_rewind": (ctypes.c_int, [c_vp]),
    "kbbq_fastq_reader_chunk": (ctypes.c_int, [c_vp, c_vp, c_u64, ctypes.c_int32, ctypes.POINTER(FastqChunk)]),
    "kbbq_fastq_reader_batch": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads)]),
    "kbbq_fastq_reader_write": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "kbbq_fastq_reader_keep": (ctypes.c_int, [c_vp, ctypes.c_int32]),
    "kbbq_fastq_reader_kept": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "kbbq_fastq_reader_select": (ctypes.c_int, [c_vp, ctypes.c_uint64, ctypes.c_void_p]),
    "kbbq_reads_upload_text": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads), ctypes.c_void_p, ctypes.POINTER(Reads)]),
    "kbbq_fastq_reader_inflate": (ctypes.c_int, [c_vp, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                                 ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "kbbq_fastq_reader_attach": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads)]),
    "kbbq_fastq_reader_kernel_ms": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "kbbq_reads_clone": (ctypes.c_int, [ctypes.POINTER(Reads), ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(Reads)]),
    "kbbq_engine_dims": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "kbbq_group_rccl_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "kbbq_group_rccl_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(c_vp)]),
    "kbbq_group_from_nccl_comm": (ctypes.c_int, [c_vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(c_vp)]),
    "kbbq_group_local_create": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(c_vp)]),
    "kbbq_group_destroy": (None, [c_vp]),
    "kbbq_group_rank": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "kbbq_exchange_filter": (ctypes.c_int, [c_vp, ctypes.c_int, c_vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    "kbbq_exchange_histograms": (ctypes.c_int, [c_vp, c_vp]),
    "kbbq_exchange_dq": (ctypes.c_int, [c_vp, c_vp]),
    "kbbq_exchange_ms": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double)]),
    "kbbq_digest_add": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint64]),
    "kbbq_digest_get": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int32]),
    "kbbq_bgzf_submit_synth": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint64)]),
    "kbbq_fastq_reader_preload": (ctypes.c_int, [c_vp, c_vp, c_u64, c_u64]),
    "kbbq_bam_reader_preload": (ctypes.c_int, [c_vp, c_vp, c_u64, c_u64]),
    "kbbq_bam_reader_create": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_char_p),
                                              ctypes.c_uint32, ctypes.POINTER(c_vp)]),
    "kbbq_bam_reader_destroy": (None, [c_vp]),
    "kbbq_bam_reader_rewind": (ctypes.c_int, [c_vp]),
    "kbbq_bam_reader_keep": (ctypes.c_int, [c_vp, ctypes.c_int32]),
    "kbbq_bam_reader_kept": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "kbbq_bam_reader_select": (ctypes.c_int, [c_vp, ctypes.c_uint64, ctypes.c_void_p]),
    "kbbq_bam_reader_chunk": (ctypes.c_int, [c_vp, c_vp, c_u64, ctypes.c_int32, ctypes.POINTER(FastqChunk)]),
    "kbbq_bam_reader_read_groups": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]),
    "kbbq_bam_reader_batch": (ctypes.c_int, [c_vp, ctypes.POINTER(Reads)]),
    "kbbq_bam_reader_write": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int32, c_vp]),
    "kbbq_bam_reader_kernel_ms": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
}

_LIB = None


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", src], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "kbbq_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C kbbq_amd/csrc). There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64 with the
        # same SONAME; loading torch first makes the engine bind to that copy.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is plumbing, not required by the C ABI
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


class KbbqError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("kbbq engine error %d: %s" % (code, text))
        self.code = code


def check(rc):
    if rc < 0:
        raise KbbqError(rc, lib().kbbq_last_error().decode(errors="replace"))
    return rc


class PinnedArray:
    """A numpy array over page-locked host memory from kbbq_host_alloc (DMA-able at the host link's rate)."""

    def __init__(self, count, dtype):
        import numpy as np
        self.nbytes = max(1, int(count) * np.dtype(dtype).itemsize)
        self.ptr = c_vp()
        check(lib().kbbq_host_alloc(self.nbytes, ctypes.byref(self.ptr)))
        buf = (ctypes.c_uint8 * self.nbytes).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(count))

    def free(self):
        if self.ptr is not None and self.ptr.value:
            self.array = None
            lib().kbbq_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
