"""kbbq_amd -- MI355X-native engine for kbbq's k-mer BQSR hot path.

Layout: csrc/ (HIP kernels for gfx950 + the C ABI of include/kbbq_engine.h),
_lib.py (ctypes binding), engine.py (host mirror of the reference's pass
functions), reads.py (batch packing), synth.py (seeded synthetic reads),
dist.py (multi-GPU exchange steps over torch.distributed / RCCL).
"""
from ._lib import KbbqError, LIB_PATH, build  # noqa: F401
from .engine import Engine, plan_parameters  # noqa: F401
from .reads import ReadBatch  # noqa: F401
