"""mixedprecisionblockqr_amd -- MI355X-native mixed-precision block QR (HIP/CDNA4) behind a C ABI.

The product is csrc/ (hand-written gfx950 kernels + the C-ABI driver, built into libmpqr.so) and
include/mpqr.h.  This package is the thin Python host mirror of the reference's interface used by
the tests and the benchmark.
"""
from . import _lib
from .api import *  # noqa: F401,F403
from .api import Handle, MpqrError

PREC_FP16, PREC_FP32, PREC_FP8 = _lib.PREC_FP16, _lib.PREC_FP32, _lib.PREC_FP8
