// mpqr_main.cpp -- C++ entry point with the run list of the reference's Cuda/main.cu:11-26, on the MI355X path.
//
//   test_qr_by_random_matrix(f)  : the 20 fixed (m, n, r) shapes of Cuda/qr.cu:1762-1783, U[0,1) input
//                                  (fixed seed instead of srand(time(0)), Cuda/mmult.cuh:43-44)
//   test_qr(f)                   : Euroc Jacobian files A_%09d.txt, r = 16 (Cuda/qr.cu:1794-1804), if a directory is given
// for f in { householder + backward accumulation, fp32 block QR, mixed-precision block QR }, printing the three
// error lines and appending the reference's CSV log rows (log/cpu_householder.txt, log/gpu_block.txt).
//
//   usage: mpqr_main [--jacobians DIR] [--m M --n N --r R] [--seed S]
// build: make -C apps    (g++, links ../mixedprecisionblockqr_amd/libmpqr.so)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/mpqr_reference_api.hpp"

static uint64_t g_seed = 1234;

static float* generate(int m, int n) {
    float* A = (float*)malloc((size_t)m * n * sizeof(float));
    mpqr_generate_matrix_host(A, m, n, g_seed++);
    return A;
}

template <typename F>
static void run_case(const char* title, const char* logname, int bits, int m, int n, int r, float* A_in, F qr) {
    printf("\nTesting %s...\nDimensions of A (m, n, r): (%d, %d, %d)\n", title, m, n, r);
    std::vector<float> Q((size_t)m * m, 0.f), R((size_t)m * n), A_out((size_t)(m + 1) * n, 0.f);
    for (int i = 0; i < m; i++) Q[(size_t)i * m + i] = 1.f;                       // h_identity_mtx (qr.cu:1873)
    memcpy(A_out.data(), A_in, (size_t)m * n * sizeof(float));                    // h_matrix_cpy  (qr.cu:1875)
    auto t0 = std::chrono::high_resolution_clock::now();
    qr(A_out.data(), Q.data(), m, n, r);
    auto t1 = std::chrono::high_resolution_clock::now();
    const float ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    const float flops = h_qr_flops_per_second(ms, m, n);
    h_strip_R_from_A(A_out.data(), R.data(), m, n);
    const float be = h_backward_error(A_in, R.data(), Q.data(), m, n, bits);
    h_q_error(Q.data(), m, bits);
    h_lower_trapezoid_error(R.data(), m, n, bits);
    h_write_results_to_log(m, n, ms, flops / 1e9f, be * 1e8f, logname);           // GFLOP/s and error x 1e8 (qr.cu:1896-1898)
    printf("%s finished in %.2f ms, averaged %.2f GFLOPs (incl. host<->device copies, reference formula)\n", title, ms, flops / 1e9f);
    free(A_in);                                                                    // testers own A_in (qr.cu:1907)
}

static void test_h_householder_qr(int m, int n, int r, float* A_in) {
    run_case("GPU householder QR + backward accumulation", "cpu_householder", 23, m, n, r, A_in,
             [](float* A, float* Q, int m_, int n_, int) {
                 h_householder_qr(A, m_, n_, 0, n_);
                 float* Qn = nullptr; h_q_backward_accumulation(A, &Qn, m_, n_);
                 memcpy(Q, Qn, (size_t)m_ * m_ * sizeof(float)); free(Qn);
             });
}
static void test_dev_block_qr(int m, int n, int r, float* A_in) {
    run_case("GPU block QR", "gpu_block", 23, m, n, r, A_in, [](float* A, float* Q, int m_, int n_, int r_) { dev_block_qr_wy(A, Q, m_, n_, r_); });
}
static void test_dev_mixed_precision_block_qr(int m, int n, int r, float* A_in) {
    run_case("GPU mixed-precision block QR", "gpu_block", 11, m, n, r, A_in,
             [](float* A, float* Q, int m_, int n_, int r_) { dev_mixed_precision_block_qr(A, Q, m_, n_, r_); });
}

static void test_qr_by_random_matrix(QR_FUNC f) {
    static const QRProblemSize dims[20] = {{6, 4, 2}, {6, 4, 1}, {6, 4, 3}, {12, 8, 4}, {12, 8, 5}, {12, 8, 6}, {12, 8, 2},
        {12, 8, 8}, {12, 8, 3}, {24, 16, 8}, {24, 16, 12}, {60, 40, 8}, {60, 40, 16}, {80, 80, 16}, {97, 90, 16},
        {100, 80, 16}, {128, 80, 16}, {129, 80, 16}, {240, 160, 16}, {600, 400, 16}};
    for (const auto& d : dims) f(d.m, d.n, d.r, generate(d.m, d.n));
}

static void test_qr(QR_FUNC f, const char* dir) {
    std::vector<MatrixInfo> list;
    for (int i = 100; i <= 22500; i += 100) {
        char name[512]; snprintf(name, sizeof name, "%s/A_%09d.txt", dir, i);
        FILE* fp = fopen(name, "r");
        if (!fp) continue;
        MatrixInfo mi; mi.filePath = name;
        if (fscanf(fp, "%d %d", &mi.m, &mi.n) == 2) list.push_back(mi);
        fclose(fp);
    }
    std::sort(list.begin(), list.end(), [](const MatrixInfo& a, const MatrixInfo& b) { return a.m < b.m; });
    int used = 0;
    for (size_t i = 0; i < list.size() && used < 30; i += 2, used++) {            // every 2nd, at most 30 (qr.cu:1752-1757)
        int m, n; float* A_in;
        read_euroc_jacobian(list[i].filePath, &m, &n, &A_in);
        if (m < n) { printf("skipping %s: m < n\n", list[i].filePath.c_str()); free(A_in); continue; }
        f(m, n, 16, A_in);
    }
    if (list.empty()) printf("no Jacobian files under %s (the reference's data blob is a git-LFS pointer)\n", dir);
}

int main(int argc, char** argv) {
    const char* jac = nullptr; int m = 0, n = 0, r = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--jacobians") && i + 1 < argc) jac = argv[++i];
        else if (!strcmp(argv[i], "--m") && i + 1 < argc) m = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--n") && i + 1 < argc) n = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--r") && i + 1 < argc) r = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) g_seed = strtoull(argv[++i], nullptr, 10);
        else { fprintf(stderr, "usage: %s [--jacobians DIR] [--m M --n N --r R] [--seed S]\n", argv[0]); return 2; }
    }
    try {
        if (m > 0 && n > 0 && r > 0) { test_dev_mixed_precision_block_qr(m, n, r, generate(m, n)); return 0; }
        test_qr_by_random_matrix(test_h_householder_qr);
        test_qr_by_random_matrix(test_dev_block_qr);
        test_qr_by_random_matrix(test_dev_mixed_precision_block_qr);
        if (jac) { test_qr(test_h_householder_qr, jac); test_qr(test_dev_block_qr, jac); test_qr(test_dev_mixed_precision_block_qr, jac); }
    } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
