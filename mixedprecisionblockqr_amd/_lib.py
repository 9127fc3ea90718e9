"""ctypes loader for libmpqr.so (the C-ABI library declared in include/mpqr.h).

There is no Python or CPU fallback: if the shared library is missing it is built in-tree with
hipcc (csrc/Makefile); if that is impossible the import fails loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpqr.so")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_IO, ERR_STATE = range(7)
PREC_FP16, PREC_FP32, PREC_FP8 = 0, 1, 2


class MpqrOpts(C.Structure):
    _fields_ = [("precision", C.c_int), ("outer_block", C.c_int), ("form_q", C.c_int), ("lookahead", C.c_int),
                ("reserved", C.c_int * 12)]


class MpqrMetrics(C.Structure):
    _fields_ = [("backward_error", C.c_double), ("q_error_max_signed", C.c_double), ("lower_trapezoid", C.c_double),
                ("q_error_fro", C.c_double), ("a_norm", C.c_double)]


class MpqrTimings(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_factor", C.c_float), ("ms_form_q", C.c_float),
                ("ms_trailing", C.c_float), ("ms_panel", C.c_float), ("ms_far_tn", C.c_float),
                ("ms_far_nn", C.c_float), ("n_far_launches", C.c_int), ("flops_far_tn", C.c_double),
                ("flops_far_nn", C.c_double), ("ms_chain_wait", C.c_float), ("n_passes", C.c_int),
                ("n_robust_leaves", C.c_int), ("ms_q_tn", C.c_float), ("ms_q_nn", C.c_float), ("n_q_launches", C.c_int),
                ("tflop_q", C.c_float), ("ms_host_enqueue", C.c_float), ("gbytes_far_nn", C.c_double), ("gbytes_q_nn", C.c_double),
                ("n_gh_leaves", C.c_int), ("us_gh_solve", C.c_float), ("n_q_ident_rows", C.c_int), ("restart_block", C.c_int),
                ("n_fused_leaves", C.c_int), ("n_tpoll_retries", C.c_int), ("n_deflated_columns", C.c_int), ("tflop_q_tn", C.c_float)]


def build(force=False):
    """Compile libmpqr.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], check=True)
    return LIB_PATH


_f32 = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i, _l, _f, _d, _p, _s = C.c_int, C.c_long, C.c_float, C.c_double, C.c_void_p, C.c_char_p
_H = C.c_void_p

# name -> (restype, argtypes); mirrors include/mpqr.h one to one
SIGNATURES = {
    "mpqr_version": (_s, []),
    "mpqr_abi_sizes": (None, [C.POINTER(C.c_int)]),
    "mpqr_default_opts": (None, [C.POINTER(MpqrOpts)]),
    "mpqr_create": (_i, [C.POINTER(_H), _i]),
    "mpqr_destroy": (_i, [_H]),
    "mpqr_last_error": (_s, [_H]),
    "mpqr_block_qr_f32": (_i, [_H, _f32, _p, _i, _i, _i, C.POINTER(MpqrOpts)]),
    "mpqr_dev_mixed_precision_block_qr": (_i, [_f32, _f32, _i, _i, _i]),
    "mpqr_dev_block_qr_wy": (_i, [_f32, _f32, _i, _i, _i]),
    "mpqr_plan": (_i, [_H, _i, _i, _i, C.POINTER(MpqrOpts)]),
    "mpqr_set_matrix_host": (_i, [_H, _f32, _l]),
    "mpqr_set_matrix_device": (_i, [_H, _p, _l]),
    "mpqr_generate_matrix": (_i, [_H, C.c_uint64]),
    "mpqr_factor": (_i, [_H]),
    "mpqr_sync": (_i, [_H]),
    "mpqr_get_timings": (_i, [_H, C.POINTER(MpqrTimings)]),
    "mpqr_bench_leaf_solve": (_i, [_H, _i, _i, C.POINTER(C.c_float)]),
    "mpqr_gemm_test_f32": (_i, [_H, _f32, _f32, _f32, _i, _i, _i, _i, _i]),
    "mpqr_bench_gemm": (_i, [_H, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float)]),
    "mpqr_bench_mfma_peak": (_i, [_H, _i, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mpqr_get_update_records": (_i, [_H, _i, _p, _p, _p, _p, C.POINTER(C.c_int)]),
    "mpqr_get_factor_host": (_i, [_H, _f32]),
    "mpqr_get_q_host": (_i, [_H, _f32]),
    "mpqr_get_r_host": (_i, [_H, _f32]),
    "mpqr_metrics_device": (_i, [_H, C.POINTER(MpqrMetrics)]),
    "mpqr_householder_qr_f32": (_i, [_H, _f32, _i, _i, _i, _i, _i]),
    "mpqr_wy_transform_f32": (_i, [_H, _f32, _i, _i, _i, _i, _p, _p]),
    "mpqr_q_backward_accumulation_f32": (_i, [_H, _f32, _f32, _i, _i, _i]),
    "mpqr_apply_panel_to_trailing_f32": (_i, [_H, _f32, _i, _i, _i, _i, _i]),
    "mpqr_metrics_f32": (_i, [_H, _f32, _f32, _f32, _i, _i, C.POINTER(MpqrMetrics)]),
    "mpqr_q_error_f32": (_i, [_H, _f32, _i, C.POINTER(MpqrMetrics)]),
    "mpqr_error_passes": (_i, [_d, _i, _i]),
    "mpqr_qr_factorization_f64": (_i, [_H, _f64, _f64, _i, _i]),
    "mpqr_apply_qt_host": (_i, [_H, _f32, _i]),
    "mpqr_solve_ls_host": (_i, [_H, _f32, _i, _f32]),
    "mpqr_qr_solver_f32": (_i, [_H, _f32, _f32, _f32, _i, _i, _i]),
    "mpqr_dev_qr_solver": (_i, [_f32, _f32, _f32, _i, _i]),
    "mpqr_read_euroc_jacobian": (_i, [_s, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.POINTER(_f))]),
    "mpqr_write_euroc_jacobian": (_i, [_s, _i, _i, _f32]),
    "mpqr_free_host": (None, [_p]),
    "mpqr_write_results_to_log": (_i, [_s, _s, _i, _i, _f, _f, _f]),
    "mpqr_qr_flops_per_second": (_f, [_f, _i, _i]),
    "mpqr_flops_geqrf": (_d, [_i, _i]),
    "mpqr_flops_form_q": (_d, [_i, _i]),
    "mpqr_flops_trailing": (_d, [_i, _i, _i]),
    "mpqr_flops_panel": (_d, [_i, _i, _i]),
    "mpqr_generate_matrix_host": (None, [_f32, _i, _i, C.c_uint64]),
    "mpqr_part_owner": (_i, [_i, _i, _i]),
    "mpqr_part_local_cols": (_i, [_i, _i, _i, _i]),
    "mpqr_part_local_index": (_i, [_i, _i, _i]),
    "mpqr_part_global_index": (_i, [_i, _i, _i, _i]),
    "mpqr_dist_plan": (_i, [_H, _i, _i, _i, _i, _i, C.POINTER(MpqrOpts)]),
    "mpqr_dist_block": (_i, [_H]),
    "mpqr_dist_num_blocks": (_i, [_H]),
    "mpqr_dist_block_owner": (_i, [_H, _i]),
    "mpqr_dist_local_cols": (_i, [_H]),
    "mpqr_dist_local_q_cols": (_i, [_H]),
    "mpqr_dist_set_local_matrix_host": (_i, [_H, _p, _l]),
    "mpqr_dist_generate_matrix": (_i, [_H, C.c_uint64]),
    "mpqr_dist_local_absmax": (_i, [_H, C.POINTER(_f)]),
    "mpqr_dist_begin": (_i, [_H, _f]),
    "mpqr_dist_factor_block": (_i, [_H, _i]),
    "mpqr_dist_flags": (_i, [_H, C.POINTER(C.c_int)]),
    "mpqr_dist_set_robust": (_i, [_H, _i]),
    "mpqr_dist_block_bytes": (_l, [_H, _i]),
    "mpqr_dist_pack_block": (_i, [_H, _i, _p]),
    "mpqr_dist_unpack_block": (_i, [_H, _i, _p]),
    "mpqr_dist_pack_block_async": (_i, [_H, _i, _p]),
    "mpqr_dist_unpack_block_async": (_i, [_H, _i, _p]),
    "mpqr_dist_chain_stream": (_i, [_H, C.POINTER(C.c_void_p)]),
    "mpqr_dist_update": (_i, [_H, _i]),
    "mpqr_dist_update_part": (_i, [_H, _i, _i]),
    "mpqr_dist_form_q": (_i, [_H]),
    "mpqr_dist_get_local_factor_host": (_i, [_H, _p]),
    "mpqr_dist_get_local_q_host": (_i, [_H, _p]),
    "mpqr_dist_get_local_input_host": (_i, [_H, _p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            try:
                build()
            except Exception as e:  # no fallback: fail loudly
                raise ImportError(f"libmpqr.so is missing and could not be built with hipcc: {e}") from e
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here == header/library mismatch: loud
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib
