"""Host-side mirror of the reference's block-QR interface on top of the C ABI (include/mpqr.h).

Function names, argument meaning and buffer conventions follow the reference's free functions
(Cuda/qr.cuh:68-137, C++/main.cpp:16) so tests read like the reference's own testers
(Cuda/qr.cu:1806-1908).  Differences: errors raise MpqrError instead of exit(); buffers are numpy
arrays.  Everything computes on the GPU through libmpqr.so -- there is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class MpqrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mpqr error {code}: {msg}")
        self.code = code


def _opts(precision=L.PREC_FP16, outer_block=0, form_q=True, lookahead=True):
    o = L.MpqrOpts()
    L.lib().mpqr_default_opts(C.byref(o))
    o.precision, o.outer_block, o.form_q, o.lookahead = int(precision), int(outer_block), int(bool(form_q)), int(bool(lookahead))
    return o


class Handle:
    """One handle per GPU (owns HIP streams and all device workspace)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = L.lib().mpqr_create(C.byref(self._h), int(device))
        if rc != L.OK:
            raise MpqrError(rc, L.lib().mpqr_last_error(None).decode())

    def close(self):
        if self._h:
            L.lib().mpqr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != L.OK:
            raise MpqrError(rc, L.lib().mpqr_last_error(self._h).decode())

    # ---- device-resident driver
    def plan(self, m, n, r, **kw):
        self.m, self.n, self.r = m, n, r
        o = _opts(**kw)
        self._chk(L.lib().mpqr_plan(self._h, m, n, r, C.byref(o)))

    def set_matrix(self, A):
        A = np.ascontiguousarray(A, np.float32)
        self._chk(L.lib().mpqr_set_matrix_host(self._h, A, A.shape[1]))

    def set_matrix_device(self, ptr, ld):
        self._chk(L.lib().mpqr_set_matrix_device(self._h, C.c_void_p(ptr), ld))

    def generate(self, seed=1234):
        self._chk(L.lib().mpqr_generate_matrix(self._h, seed))

    def factor(self):
        self._chk(L.lib().mpqr_factor(self._h))

    def sync(self):
        self._chk(L.lib().mpqr_sync(self._h))

    def timings(self):
        t = L.MpqrTimings()
        self._chk(L.lib().mpqr_get_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in L.MpqrTimings._fields_ if k != "reserved"}

    def bench_gemm(self, kernel, mode, M, N, K, iters=10):
        """Mean ms per launch of one large-shape GEMM kernel alone on the GPU (include/mpqr.h: mpqr_bench_gemm)."""
        ms = C.c_float()
        self._chk(L.lib().mpqr_bench_gemm(self._h, kernel, mode, M, N, K, iters, C.byref(ms)))
        return ms.value

    def bench_mfma_peak(self, shape=0):
        """(TFLOP/s, GHz) of a bare fp16 MFMA loop on random operands (include/mpqr.h: mpqr_bench_mfma_peak)."""
        tf, ghz = C.c_float(), C.c_float()
        self._chk(L.lib().mpqr_bench_mfma_peak(self._h, shape, C.byref(tf), C.byref(ghz)))
        return tf.value, ghz.value

    def update_records(self):
        """The recorded C -= V Y^T launches of the last factorisation in launch order (far updates, then Q formation):
        list of dicts {set, M, N, K, flops, bytes} (include/mpqr.h: mpqr_get_update_records)."""
        n = C.c_int()
        self._chk(L.lib().mpqr_get_update_records(self._h, 0, None, None, None, None, C.byref(n)))
        k = n.value
        fl, by = np.zeros(k, np.float64), np.zeros(k, np.float64)
        dm, iq = np.zeros(3 * k, np.int32), np.zeros(k, np.int32)
        self._chk(L.lib().mpqr_get_update_records(self._h, k, fl.ctypes.data_as(C.c_void_p), by.ctypes.data_as(C.c_void_p),
                                                  dm.ctypes.data_as(C.c_void_p), iq.ctypes.data_as(C.c_void_p), C.byref(n)))
        return [{"set": "q" if iq[i] else "far", "M": int(dm[3 * i]), "N": int(dm[3 * i + 1]), "K": int(dm[3 * i + 2]),
                 "flops": float(fl[i]), "bytes": float(by[i])} for i in range(k)]

    def bench_leaf_solve(self, w=128, iters=50):
        """Mean launch time (us) of the serial core of one Gram-Householder leaf of width w, timed alone."""
        us = C.c_float()
        self._chk(L.lib().mpqr_bench_leaf_solve(self._h, w, iters, C.byref(us)))
        return us.value

    def factor_out(self):
        out = np.empty((self.m + 1, self.n), np.float32)
        self._chk(L.lib().mpqr_get_factor_host(self._h, out))
        return out

    def q(self):
        Q = np.empty((self.m, self.m), np.float32)
        self._chk(L.lib().mpqr_get_q_host(self._h, Q))
        return Q

    def r_matrix(self):
        R = np.empty((self.m, self.n), np.float32)
        self._chk(L.lib().mpqr_get_r_host(self._h, R))
        return R

    def metrics(self):
        mt = L.MpqrMetrics()
        self._chk(L.lib().mpqr_metrics_device(self._h, C.byref(mt)))
        return {k: getattr(mt, k) for k, _ in L.MpqrMetrics._fields_}


_default = None


def default_handle():
    global _default
    if _default is None:
        _default = Handle(0)
    return _default


# ---------------------------------------------------------------- reference-named entry points
def dev_mixed_precision_block_qr(A, Q, m, n, r, handle=None, **kw):
    """Cuda/qr.cuh:133.  A: (m+1) x n float32 (in: matrix + zero row; out: R + shifted reflectors),
    Q: m x m float32 (out).  Both modified in place, like the reference."""
    return _block_qr(A, Q, m, n, r, L.PREC_FP16, handle, **kw)


def dev_block_qr_wy(A, Q, m, n, r, handle=None, **kw):
    """Cuda/qr.cuh:131 (fp32 twin)."""
    return _block_qr(A, Q, m, n, r, L.PREC_FP32, handle, **kw)


def _block_qr(A, Q, m, n, r, precision, handle, **kw):
    h = handle or default_handle()
    assert A.dtype == np.float32 and A.shape == (m + 1, n) and A.flags.c_contiguous
    o = _opts(precision=precision, **kw)
    qp = None
    if o.form_q:
        assert Q.dtype == np.float32 and Q.shape == (m, m) and Q.flags.c_contiguous
        qp = Q.ctypes.data_as(C.c_void_p)
    h._chk(L.lib().mpqr_block_qr_f32(h._h, A, qp, m, n, r, C.byref(o)))
    h.m, h.n, h.r = m, n, r                                   # the call planned the handle for this shape


def h_householder_qr(A, m, n, global_offset, panel_width, handle=None, precision=L.PREC_FP32):
    """Cuda/qr.cu:198 semantics, executed on the GPU.  A: (m+1) x n, in place.  The reference function is pure fp32,
    so is the default here (PREC_FP16 runs the in-panel updates of panels wider than one leaf on the fp16 MFMA path)."""
    h = handle or default_handle()
    h._chk(L.lib().mpqr_householder_qr_f32(h._h, A, m, n, global_offset, panel_width, int(precision)))


def wy_transform(A, m, n, global_offset, panel_width, dense=False, handle=None):
    """Cuda/qr.cu:337/:535.  Returns T (compact WY), and the dense Q_panel if dense=True."""
    h = handle or default_handle()
    T = np.empty((panel_width, panel_width), np.float32)
    Qp = np.empty((m - global_offset, m - global_offset), np.float32) if dense else None
    h._chk(L.lib().mpqr_wy_transform_f32(h._h, np.ascontiguousarray(A, np.float32), m, n, global_offset, panel_width,
                                         T.ctypes.data_as(C.c_void_p),
                                         Qp.ctypes.data_as(C.c_void_p) if dense else None))
    return (T, Qp) if dense else T


def h_q_backward_accumulation(A, m, n, handle=None, precision=L.PREC_FP32):
    """Cuda/qr.cu:296 (pure fp32 in the reference, the default here).  Returns Q (m x m) from the reflectors stored in A ((m+1) x n)."""
    h = handle or default_handle()
    Q = np.empty((m, m), np.float32)
    h._chk(L.lib().mpqr_q_backward_accumulation_f32(h._h, np.ascontiguousarray(A, np.float32), Q, m, n, int(precision)))
    return Q


def h_q_error(Q, handle=None):
    """Cuda/qr.cu:137-171: max signed entry of Q^T Q - I (and its Frobenius norm)."""
    h = handle or default_handle()
    Q = np.ascontiguousarray(Q, np.float32)
    mt = L.MpqrMetrics()
    h._chk(L.lib().mpqr_q_error_f32(h._h, Q, Q.shape[0], C.byref(mt)))
    return {"q_error_max_signed": mt.q_error_max_signed, "q_error_fro": mt.q_error_fro}


def apply_panel_to_trailing(A, m, n, global_offset, panel_width, precision=L.PREC_FP16, handle=None):
    """Trailing update A[l:, tau:] <- Q_panel^T A[l:, tau:] (Cuda/qr.cu:1098-1106), in place."""
    h = handle or default_handle()
    h._chk(L.lib().mpqr_apply_panel_to_trailing_f32(h._h, A, m, n, global_offset, panel_width, precision))


def gemm_test(A, B, Cm, kernel, mode=0, handle=None):
    """Test aid: Cm (-)= A B through one of the library's MFMA GEMM kernels (include/mpqr.h: mpqr_gemm_test_f32); the
    counterpart of the reference's test_template_tensorcore_mmult_tiled (Cuda/mmult.cuh:387-435).  Cm is modified in place."""
    h = handle or default_handle()
    M, K = A.shape; K2, N = B.shape
    assert K == K2 and Cm.shape == (M, N) and all(x.dtype == np.float32 and x.flags.c_contiguous for x in (A, B, Cm))
    h._chk(L.lib().mpqr_gemm_test_f32(h._h, A, B, Cm, M, N, K, kernel, mode))
    return Cm


def qr_metrics(A, R, Q, handle=None):
    """h_backward_error / h_q_error / h_lower_trapezoid_error (Cuda/qr.cu:115-196) on the GPU."""
    h = handle or default_handle()
    m, n = A.shape
    mt = L.MpqrMetrics()
    f = lambda x: np.ascontiguousarray(x, np.float32)
    h._chk(L.lib().mpqr_metrics_f32(h._h, f(A), f(R), f(Q), m, n, C.byref(mt)))
    return {k: getattr(mt, k) for k, _ in L.MpqrMetrics._fields_}


def h_strip_R_from_A(A, m, n):
    """Cuda/qr.cu:85-100 (pure indexing; no arithmetic)."""
    return np.triu(A[:m, :n]).astype(np.float32)


def error_passes(err, m, precision_bits):
    return bool(L.lib().mpqr_error_passes(float(err), m, precision_bits))


def qr_factorization(A, handle=None):
    """C++/main.cpp:16 on the GPU in fp64.  A: m x n (numpy, row-major view); returns (Q, R)."""
    h = handle or default_handle()
    m, n = A.shape
    Ac = np.ascontiguousarray(A.T.astype(np.float64))       # column-major storage
    Qc = np.zeros((m, m), np.float64)
    h._chk(L.lib().mpqr_qr_factorization_f64(h._h, Ac, Qc, m, n))
    return Qc.T.copy(), Ac.T.copy()


def apply_qt(B, handle=None):
    """B <- Q^T B (m x nrhs), implicitly from the reflectors of the last factorisation (G&VL Alg. 5.3.2 step 2)."""
    h = handle or default_handle()
    B = np.ascontiguousarray(B, np.float32)
    B2 = B.reshape(B.shape[0], -1).copy()
    h._chk(L.lib().mpqr_apply_qt_host(h._h, B2, B2.shape[1]))
    return B2.reshape(B.shape)


def solve_ls(B, handle=None):
    """argmin ||A X - B|| for the matrix last factored on `handle`: X = R^-1 (Q^T B)[0:n]."""
    h = handle or default_handle()
    B = np.ascontiguousarray(B, np.float32)
    B2 = B.reshape(B.shape[0], -1)
    X = np.empty((h.n, B2.shape[1]), np.float32)
    h._chk(L.lib().mpqr_solve_ls_host(h._h, np.ascontiguousarray(B2), B2.shape[1], X))
    return X[:, 0] if B.ndim == 1 else X


def linear_least_square(A, y, r=128, handle=None):
    """python/linear_least_sqare.py:5-22 / dev_QR_Solver (Cuda/QR/Solver/solver.cu:39-87): x = argmin ||A x - y||."""
    h = handle or default_handle()
    A = np.ascontiguousarray(A, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    m, n = A.shape
    x = np.empty(n, np.float32)
    h._chk(L.lib().mpqr_qr_solver_f32(h._h, A, y, x, m, n, r))
    h.m, h.n, h.r = m, n, r                                   # the call planned the handle for this shape
    return x


def dev_QR_Solver(A, b, x, m, n):
    """void dev_QR_Solver(float* A, float* b, float* x, int m, int n)  Cuda/QR/Solver/solver.cu:39 (default handle)."""
    rc = L.lib().mpqr_dev_qr_solver(np.ascontiguousarray(A, np.float32), np.ascontiguousarray(b, np.float32), x, m, n)
    if rc != L.OK:
        raise MpqrError(rc, "dev_QR_Solver failed")


def read_euroc_jacobian(path):
    """Cuda/qr.cu:696.  Returns a dense float32 matrix."""
    rows, cols = C.c_int(), C.c_int()
    ptr = C.POINTER(C.c_float)()
    rc = L.lib().mpqr_read_euroc_jacobian(str(path).encode(), C.byref(rows), C.byref(cols), C.byref(ptr))
    if rc != L.OK:
        raise MpqrError(rc, f"cannot read {path}")
    M = np.ctypeslib.as_array(ptr, shape=(rows.value, cols.value)).copy()
    L.lib().mpqr_free_host(ptr)
    return M


def write_euroc_jacobian(path, M):
    M = np.ascontiguousarray(M, np.float32)
    rc = L.lib().mpqr_write_euroc_jacobian(str(path).encode(), M.shape[0], M.shape[1], M)
    if rc != L.OK:
        raise MpqrError(rc, f"cannot write {path}")


def synthetic_jacobian(cams=40, points=580, views=2, seed=1234, outliers=8, rank_deficiency=0):
    """Stand-in for a EuRoC bundle-adjustment Jacobian (BASELINE config 3; the reference's real files are an
    absent LFS blob): one 2-row block per observation with a dense 2x6 camera block and a dense 2x3 point block,
    everything else zero.  Values ~ N(0, 1) with a few large entries (1e3 .. 1e5) that would overflow fp16 without the
    library's power-of-two scale.  Returns a dense row-major fp32 matrix (rows = 2 * observations >= columns)."""
    rng = np.random.default_rng(seed)
    n = 6 * cams + 3 * points
    obs = [(pt, int(c)) for pt in range(points) for c in rng.choice(cams, size=views, replace=False)]
    while 2 * len(obs) < n:                                   # keep the system at least square
        obs.append((int(rng.integers(points)), int(rng.integers(cams))))
    M = np.zeros((2 * len(obs), n), np.float32)
    for k, (pt, c) in enumerate(obs):
        M[2 * k:2 * k + 2, 6 * c:6 * c + 6] = rng.standard_normal((2, 6))
        M[2 * k:2 * k + 2, 6 * cams + 3 * pt:6 * cams + 3 * pt + 3] = rng.standard_normal((2, 3))
    nz = np.argwhere(M != 0)
    for i, j in nz[rng.choice(len(nz), size=min(outliers, len(nz)), replace=False)]:
        M[i, j] *= 10.0 ** rng.uniform(3, 5)
    i, j = np.unravel_index(np.argmax(np.abs(M)), M.shape)
    if abs(M[i, j]) <= 65504.0:                                # at least one entry beyond the fp16 range
        M[i, j] = np.sign(M[i, j]) * 1.3e5
    if rank_deficiency:
        # gauge freedom of a real bundle-adjustment Jacobian (7 for a free similarity transform) as exactly dependent
        # columns: camera column 12q+7 becomes a combination of camera columns 12q+2 and 12q+9, so rank = n - rank_deficiency.
        # All of them lie in the first 128-column leaf: a dependency INSIDE a tall leaf is what the Gram-Householder
        # leaf cannot resolve (cond^2 in its Gram matrix) and must hand to the column-by-column kernels; a dependency
        # across leaves is reduced by the trailing updates first and arrives as a small, harmless column.
        assert rank_deficiency <= 10 and 6 * cams >= 128
        for q in range(rank_deficiency):
            M[:, 12 * q + 7] = 0.5 * M[:, 12 * q + 2] - 0.25 * M[:, 12 * q + 9]
    return M


def h_write_results_to_log(height, width, time_ms, flops_per_second, backward_error, file_name="logFile", log_dir="log"):
    """Cuda/qr.cu:58-83."""
    rc = L.lib().mpqr_write_results_to_log(str(log_dir).encode(), file_name.encode(), height, width, time_ms,
                                           flops_per_second, backward_error)
    if rc != L.OK:
        raise MpqrError(rc, "cannot write log")


def h_qr_flops_per_second(time_ms, m, n):
    return float(L.lib().mpqr_qr_flops_per_second(time_ms, m, n))


def generate_matrix(m, n, seed=1234):
    A = np.empty((m, n), np.float32)
    L.lib().mpqr_generate_matrix_host(A, m, n, seed)
    return A


def flops(m, n, r):
    l = L.lib()
    return {"geqrf": l.mpqr_flops_geqrf(m, n), "form_q": l.mpqr_flops_form_q(m, n),
            "trailing": l.mpqr_flops_trailing(m, n, r), "panel": l.mpqr_flops_panel(m, n, r)}
