"""ctypes binding of the CPU oracle (oracle/liboracle_qr.so, oracle/_ref/libref_cppmain.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def _ensure_built(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", _HERE, name], check=True, capture_output=True)
    return path


def _load(name):
    lib = C.CDLL(_ensure_built(name))
    i, f, p = C.c_int, C.c_float, C.c_void_p
    sig = {
        "orc_householder_qr": (None, [_f32p, i, i, i, i]),
        "orc_q_backward_accumulation": (None, [_f32p, _f32p, i, i]),
        "orc_wy_transform": (None, [_f32p, _f32p, i, i, i, i]),
        "orc_block_qr": (None, [_f32p, _f32p, i, i, i]),
        "orc_mixed_precision_block_qr": (None, [_f32p, _f32p, i, i, i]),
        "orc_strip_R_from_A": (None, [_f32p, _f32p, i, i]),
        "orc_qr_flops_per_second": (f, [f, i, i]),
        "orc_backward_error": (f, [_f32p, _f32p, _f32p, i, i]),
        "orc_q_error": (f, [_f32p, i]),
        "orc_lower_trapezoid_error": (f, [_f32p, i, i]),
        "orc_error_passes": (i, [f, i, i]),
        "orc_q_error_fro": (C.c_double, [_f32p, i]),
        "orc_backward_error_f64": (C.c_double, [_f32p, _f32p, _f32p, i, i]),
        "orc_compact_wy_T": (None, [_f32p, _f32p, i, i, i, i, i]),
        "orc_extract_V": (None, [_f32p, _f32p, i, i, i, i]),
        "orc_block_qr_compact": (None, [_f32p, _f32p, i, i, i, i]),
        "orc_qr_factorization_f64": (None, [_f64p, _f64p, i]),
        "orc_read_euroc_jacobian": (i, [C.c_char_p, C.POINTER(i), C.POINTER(i), C.POINTER(C.POINTER(f))]),
        "orc_write_euroc_jacobian": (i, [C.c_char_p, i, i, _f32p]),
        "orc_free": (None, [p]),
        "orc_generate_random_matrix": (None, [_f32p, i, i, C.c_uint64]),
        "orc_round_fp16": (f, [f]),
    }
    for n, (res, args) in sig.items():
        fn = getattr(lib, n)
        fn.restype, fn.argtypes = res, args
    return lib


_lib = None
_lib_omp = None
_ref = None


def lib(omp=False):
    global _lib, _lib_omp
    if omp:
        if _lib_omp is None:
            _lib_omp = _load("liboracle_qr_omp.so")
        return _lib_omp
    if _lib is None:
        _lib = _load("liboracle_qr.so")
    return _lib


def ref_lib():
    """The real reference C++/main.cpp (oracle/_ref).  None if it was never built
    (it can only be built where /root/reference exists; the built file travels)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libref_cppmain.so")
        if not os.path.exists(path):
            if os.path.exists("/root/reference/C++/main.cpp"):
                subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)
            if not os.path.exists(path):
                return None
        r = C.CDLL(path)
        r.ref_qr_factorization.restype, r.ref_qr_factorization.argtypes = None, [_f64p, _f64p, C.c_int]
        r.ref_householder.restype, r.ref_householder.argtypes = None, [_f64p, C.c_int, _f64p]
        r.ref_main.restype, r.ref_main.argtypes = C.c_int, []
        _ref = r
    return _ref


# ---------------------------------------------------------------- helpers
def generate(m, n, seed=1234):
    A = np.empty((m, n), np.float32)
    lib().orc_generate_random_matrix(A, m, n, seed)
    return A


def padded(A):
    """(m+1) x n working copy with a zero extra row (Cuda/qr.cu:1869-1875)."""
    m, n = A.shape
    out = np.zeros((m + 1, n), np.float32)
    out[:m] = A
    return out


def strip_R(Aout, m, n):
    R = np.empty((m, n), np.float32)
    lib().orc_strip_R_from_A(np.ascontiguousarray(Aout[:m]), R, m, n)
    return R


def householder_qr(A):
    """Unblocked reference path: h_householder_qr over all columns + backward
    accumulation (Cuda/qr.cu:1354-1361).  Returns (A_out (m+1 x n), Q, R)."""
    m, n = A.shape
    Ao = padded(A)
    lib().orc_householder_qr(Ao, m, n, 0, n)
    Q = np.empty((m, m), np.float32)
    lib().orc_q_backward_accumulation(Ao, Q, m, n)
    return Ao, Q, strip_R(Ao, m, n)


def linear_least_square(A, y):
    """python/linear_least_sqare.py:5-22 on top of the oracle's Householder QR: b = Q^T y (the reference's pinv(Q) y for
    an orthogonal Q), then its back-substitution loop x_i = (b_i - sum_{k>i} R_ik x_k) / R_ii, here in fp64 on the fp32
    factors.  The solution does not depend on the sign convention of R."""
    m, n = A.shape
    _, Q, R = householder_qr(A)
    b = Q.astype(np.float64).T @ np.asarray(y, np.float64)
    Rd = R[:n].astype(np.float64)
    x = np.zeros(n)
    for i in reversed(range(n)):
        x[i] = (b[i] - Rd[i, i + 1:] @ x[i + 1:]) / Rd[i, i]
    return x


def block_qr(A, r, variant="dense", omp=False):
    """variant: 'dense' (h_block_qr), 'mixed' (fp16 Q accumulation emulation),
    'compact32' / 'compact16' (compact-WY restatement, fp32 / fp16-operand)."""
    m, n = A.shape
    Ao = padded(A)
    Q = np.eye(m, dtype=np.float32)
    L = lib(omp)
    if variant == "dense":
        L.orc_block_qr(Ao, Q, m, n, r)
    elif variant == "mixed":
        L.orc_mixed_precision_block_qr(Ao, Q, m, n, r)
    elif variant == "compact32":
        L.orc_block_qr_compact(Ao, Q, m, n, r, 0)
    elif variant == "compact16":
        L.orc_block_qr_compact(Ao, Q, m, n, r, 1)
    else:
        raise ValueError(variant)
    return Ao, Q, strip_R(Ao, m, n)


def wy_transform(Ao, m, n, go, pw):
    dim = m - go
    Qp = np.empty((dim, dim), np.float32)
    lib().orc_wy_transform(Ao, Qp, m, n, go, pw)
    return Qp


def extract_V(Ao, m, n, go, pw):
    V = np.empty((m - go, pw), np.float32)
    lib().orc_extract_V(Ao, V, m, n, go, pw)
    return V


def compact_T(Ao, m, n, go, pw, round_v_fp16=False):
    T = np.empty((pw, pw), np.float32)
    lib().orc_compact_wy_T(Ao, T, m, n, go, pw, int(round_v_fp16))
    return T


def metrics(A, R, Q):
    m, n = A.shape
    L = lib()
    A = np.ascontiguousarray(A, np.float32)
    R = np.ascontiguousarray(R, np.float32)
    Q = np.ascontiguousarray(Q, np.float32)
    return {
        "backward_error": float(L.orc_backward_error(A, R, Q, m, n)),
        "q_error_max_signed": float(L.orc_q_error(Q, m)),
        "lower_trapezoid": float(L.orc_lower_trapezoid_error(R, m, n)),
        "q_error_fro": float(L.orc_q_error_fro(Q, m)),
        "backward_error_f64": float(L.orc_backward_error_f64(A, R, Q, m, n)),
    }


def qr_factorization_f64(A):
    """C++/main.cpp semantics on a square matrix (numpy row-major in/out)."""
    n = A.shape[0]
    Ac = np.ascontiguousarray(A.T.astype(np.float64))      # column-major buffer
    Qc = np.ascontiguousarray(np.eye(n, dtype=np.float64))
    lib().orc_qr_factorization_f64(Ac, Qc, n)
    return Qc.T.copy(), Ac.T.copy()                          # Q, R(=A out)


def ref_qr_factorization(A):
    """The REAL C++/main.cpp:qr_factorization (oracle/_ref)."""
    r = ref_lib()
    if r is None:
        return None
    n = A.shape[0]
    Ac = np.ascontiguousarray(A.T.astype(np.float64))
    Qc = np.ascontiguousarray(np.eye(n, dtype=np.float64))
    r.ref_qr_factorization(Ac, Qc, n)
    return Qc.T.copy(), Ac.T.copy()


def read_euroc_jacobian(path):
    rows, cols = C.c_int(), C.c_int()
    ptr = C.POINTER(C.c_float)()
    rc = lib().orc_read_euroc_jacobian(path.encode(), C.byref(rows), C.byref(cols), C.byref(ptr))
    if rc != 0:
        raise IOError(f"orc_read_euroc_jacobian rc={rc}")
    M = np.ctypeslib.as_array(ptr, shape=(rows.value, cols.value)).copy()
    lib().orc_free(ptr)
    return M


def round_fp16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def round_e4m3(x):
    """Round to OCP fp8 e4m3 (4 exponent bits, bias 7, 3 mantissa bits, max 448, subnormals down to 2^-9), round to
    nearest even, saturating -- the conversion v_cvt_pk_fp8_f32 performs on gfx950 (test-side emulation of the fp8
    operand path of BASELINE config 5)."""
    x = np.clip(np.asarray(x, np.float64), -448.0, 448.0)
    ax = np.abs(x)
    e = np.floor(np.log2(np.maximum(ax, 2.0 ** -20)))
    e = np.maximum(e, -6.0)                                  # subnormal range shares the exponent of 2^-6
    q = 2.0 ** (e - 3)                                       # spacing: 3 mantissa bits
    return (np.sign(x) * np.round(ax / q) * q).astype(np.float32)   # np.round: half to even

