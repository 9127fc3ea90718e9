// mpqr_reference_api.hpp -- the reference's own C++ names and signatures, forwarding to the C ABI (mpqr.h).
//
// A maintainer of jaidonlybbert/MixedPrecisionBlockQR who wants the MI355X path includes this header instead of
// Cuda/qr.cuh and links libmpqr.so; Cuda/main.cu's run list then compiles unchanged (see apps/mpqr_main.cpp).
// Every function below cites the declaration it stands in for.  Differences, all deliberate:
//   * GPU errors do not exit() the process (Cuda/helper_cuda.h:582-595); they throw std::runtime_error here.
//   * dev_wy_transform returned a device pointer to the dense (m-l)^2 Q_panel; the compact-WY T is what the
//     build keeps, so the dense form is only available through h_wy_transform (host buffer, as in the reference).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "mpqr.h"

namespace mpqr_ref {
inline void check(int rc, const char* what) {
    if (rc != MPQR_OK) throw std::runtime_error(std::string(what) + " failed with mpqr status " + std::to_string(rc));
}
inline mpqr_handle_t& handle() {
    static mpqr_handle_t h = nullptr;
    if (!h) check(mpqr_create(&h, 0), "mpqr_create");
    return h;
}
}  // namespace mpqr_ref

// Cuda/qr.cuh:55-66
struct QRProblemSize { int m, n, r; };
struct MatrixInfo { std::string filePath; int m, n; };
// Cuda/qr.cuh:68
typedef void QR_FUNC(int, int, int, float*);

// Cuda/qr.cuh:133  void dev_mixed_precision_block_qr(float* A, float* Q, int m, int n, int r)
inline void dev_mixed_precision_block_qr(float* A, float* Q, int m, int n, int r) {
    mpqr_opts o; mpqr_default_opts(&o); o.precision = MPQR_PREC_FP16;
    mpqr_ref::check(mpqr_block_qr_f32(mpqr_ref::handle(), A, Q, m, n, r, &o), "dev_mixed_precision_block_qr");
}
// Cuda/QR/Solver/solver.cu:39  void dev_QR_Solver(float* A, float* b, float* x, int m, int n)   (a stub in the reference)
inline void dev_QR_Solver(float* A, float* b, float* x, int m, int n) {
    mpqr_ref::check(mpqr_qr_solver_f32(mpqr_ref::handle(), A, b, x, m, n, 128), "dev_QR_Solver");
}

// Cuda/qr.cuh:131  void dev_block_qr_wy(float* A, float* Q, int m, int n, int r)   (fp32 twin)
inline void dev_block_qr_wy(float* A, float* Q, int m, int n, int r) {
    mpqr_opts o; mpqr_default_opts(&o); o.precision = MPQR_PREC_FP32;
    mpqr_ref::check(mpqr_block_qr_f32(mpqr_ref::handle(), A, Q, m, n, r, &o), "dev_block_qr_wy");
}
// Cuda/qr.cuh:129  void dev_block_qr(float* A, float* Q, int m, int n, int r)     (same maths, host-side WY in the reference)
inline void dev_block_qr(float* A, float* Q, int m, int n, int r) { dev_block_qr_wy(A, Q, m, n, r); }

// Cuda/qr.cuh:85   void h_householder_qr(float* A, int m, int n, int global_offset, int panel_width)
inline void h_householder_qr(float* A, int m, int n, int global_offset, int panel_width) {
    // the reference function is pure fp32 (qr.cu:198-293): MPQR_PREC_FP32 keeps every in-panel product on the exact-f32 MFMA
    mpqr_ref::check(mpqr_householder_qr_f32(mpqr_ref::handle(), A, m, n, global_offset, panel_width, MPQR_PREC_FP32), "h_householder_qr");
}
// Cuda/qr.cuh:88   void h_q_backward_accumulation(float* h_A, float** h_Q, int m, int n)   (callee mallocs *h_Q)
inline void h_q_backward_accumulation(float* h_A, float** h_Q, int m, int n) {
    *h_Q = (float*)malloc((size_t)m * m * sizeof(float));
    mpqr_ref::check(mpqr_q_backward_accumulation_f32(mpqr_ref::handle(), h_A, *h_Q, m, n, MPQR_PREC_FP32), "h_q_backward_accumulation");
}
// Cuda/qr.cuh:90   void h_wy_transform(float* h_A, float** h_Q, int m, int n, int global_offset, int panel_width)
inline void h_wy_transform(float* h_A, float** h_Q, int m, int n, int global_offset, int panel_width) {
    const size_t d = (size_t)(m - global_offset);
    *h_Q = (float*)malloc(d * d * sizeof(float));
    mpqr_ref::check(mpqr_wy_transform_f32(mpqr_ref::handle(), h_A, m, n, global_offset, panel_width, nullptr, *h_Q), "h_wy_transform");
}
// Cuda/qr.cuh:137  void h_block_qr(float* A, float* Q, int m, int n, int r)   (CPU twin in the reference; GPU fp32 here)
inline void h_block_qr(float* A, float* Q, int m, int n, int r) { dev_block_qr_wy(A, Q, m, n, r); }

// Cuda/qr.cuh:74-82 metrics.  precision_bits only selects the printed pass/fail line, as in the reference.
inline void h_strip_R_from_A(float* A, float* R, int m, int n) {
    for (int row = 0; row < m; row++)
        for (int col = 0; col < n; col++) R[(size_t)row * n + col] = (row <= col) ? A[(size_t)row * n + col] : 0.f;
}
inline float h_qr_flops_per_second(float time_ms, int m, int n) { return mpqr_qr_flops_per_second(time_ms, m, n); }
inline float h_backward_error(float* A, float* R, float* Q, int m, int n, int precision_bits) {
    mpqr_metrics mt; mpqr_ref::check(mpqr_metrics_f32(mpqr_ref::handle(), A, R, Q, m, n, &mt), "h_backward_error");
    printf("||A - QR||/||A|| = %e Error Criteria: %s\n", mt.backward_error,
           mpqr_error_passes(mt.backward_error, m, precision_bits) ? "True" : "False");
    return (float)mt.backward_error;
}
inline float h_q_error(float* Q, int m, int precision_bits) {
    // max signed entry of Q^T Q - I (qr.cu:137-171)
    mpqr_metrics mt; mpqr_ref::check(mpqr_q_error_f32(mpqr_ref::handle(), Q, m, &mt), "h_q_error");
    printf("||QT @ Q - Im|| = %E Error Criteria: %s\n", mt.q_error_max_signed,
           mpqr_error_passes(mt.q_error_max_signed, m, precision_bits) ? "True" : "False: should be less than ");
    return (float)mt.q_error_max_signed;
}
inline float h_lower_trapezoid_error(float* R, int m, int n, int precision_bits) {
    double s = 0;
    for (int row = 0; row < m; row++) for (int col = 0; col < n && col < row; col++) s += (double)R[(size_t)row * n + col] * R[(size_t)row * n + col];
    const float e = (float)sqrt(s);
    printf("||L|| = %e Error Criteria: %s\n", e, mpqr_error_passes(e, m, precision_bits) ? "True" : "False");
    return e;
}
// Cuda/qr.cuh:72   h_write_results_to_log
inline void h_write_results_to_log(int height, int width, float time_ms, float flops_per_second, float backward_error,
                                   std::string file_name = "logFile") {
    mpqr_write_results_to_log("log", file_name.c_str(), height, width, time_ms, flops_per_second, backward_error);
}
// Cuda/qr.cuh:103  void read_euroc_jacobian(std::string filename, int* rows, int* cols, float** matrix)
inline void read_euroc_jacobian(std::string filename, int* rows, int* cols, float** matrix) {
    mpqr_ref::check(mpqr_read_euroc_jacobian(filename.c_str(), rows, cols, matrix), "read_euroc_jacobian");
}
