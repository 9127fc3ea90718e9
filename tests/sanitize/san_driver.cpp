// tests/sanitize/san_driver.cpp -- CPU sanitizer run (AddressSanitizer + UndefinedBehaviorSanitizer) of the code that runs on the
// host in this repository: the library's host utilities (csrc/host_util.cpp: EuRoC Jacobian reader / writer, CSV log, input
// generator, column partition) and the CPU oracle (oracle/oracle_qr.c).  Built and run by tests/test_sanitizers.py; GPU
// sanitizers are not available on the pool (task statement), so the device code is covered by the parity tests instead.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
extern "C" {
#include "oracle_qr.h"
int mpqr_read_euroc_jacobian(const char* path, int* rows, int* cols, float** matrix);
int mpqr_write_euroc_jacobian(const char* path, int rows, int cols, const float* M);
void mpqr_free_host(void* p);
int mpqr_write_results_to_log(const char* dir, const char* file_name, int height, int width, float time_ms, float flops_per_second,
                              float backward_error);
void mpqr_generate_matrix_host(float* A, int m, int n, uint64_t seed);
int mpqr_part_owner(int col, int block, int world);
int mpqr_part_local_cols(int n, int block, int world, int rank);
int mpqr_part_local_index(int col, int block, int world);
int mpqr_part_global_index(int lcol, int block, int world, int rank);
}
#define REQUIRE(x) do { if (!(x)) { fprintf(stderr, "FAILED: %s (line %d)\n", #x, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    // ---- generator: library and oracle bit-identical; ragged shapes
    for (int m : {1, 7, 97, 130}) for (int n : {1, 5, 90}) {
        if (n > m) continue;
        std::vector<float> a((size_t)m * n), b((size_t)m * n);
        mpqr_generate_matrix_host(a.data(), m, n, 1234); orc_generate_random_matrix(b.data(), m, n, 1234);
        REQUIRE(memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0);
    }
    // ---- Jacobian text format: write, read back with both readers, malformed and missing files
    {
        const int m = 37, n = 11;
        std::vector<float> M((size_t)m * n);
        mpqr_generate_matrix_host(M.data(), m, n, 7);
        M[5] = 0.f; M[17] = -1.5e5f; M[40] = 3e-7f;
        const std::string p = dir + "/san_jac.txt";
        REQUIRE(mpqr_write_euroc_jacobian(p.c_str(), m, n, M.data()) == 0);
        int r = 0, c = 0; float* R1 = nullptr; float* R2 = nullptr;
        REQUIRE(mpqr_read_euroc_jacobian(p.c_str(), &r, &c, &R1) == 0 && r == m && c == n);
        int r2 = 0, c2 = 0;
        REQUIRE(orc_read_euroc_jacobian(p.c_str(), &r2, &c2, &R2) == 0 && r2 == m && c2 == n);
        for (int i = 0; i < m * n; i++) REQUIRE(R1[i] == R2[i] && fabsf(R1[i] - M[i]) <= 1e-5f * fabsf(M[i]) + 1e-30f);
        mpqr_free_host(R1); orc_free(R2);
        float* R3 = nullptr;
        REQUIRE(mpqr_read_euroc_jacobian((dir + "/does_not_exist.txt").c_str(), &r, &c, &R3) != 0);
        const std::string bad = dir + "/san_bad.txt";
        FILE* f = fopen(bad.c_str(), "w"); REQUIRE(f); fputs("3 2\n1 2\n3 x\n", f); fclose(f);      // short, malformed
        R3 = nullptr;
        const int rcb = mpqr_read_euroc_jacobian(bad.c_str(), &r, &c, &R3);
        if (rcb == 0) mpqr_free_host(R3);                                                       // (either answer, no invalid access)
    }
    // ---- CSV log
    REQUIRE(mpqr_write_results_to_log(dir.c_str(), "san_log", 128, 64, 1.5f, 2.5e9f, 3e-7f) == 0);
    REQUIRE(mpqr_write_results_to_log(dir.c_str(), "san_log", 256, 128, 2.5f, 3.5e9f, 4e-7f) == 0);
    // ---- column partition: a bijection for every world size / block
    for (int world : {1, 2, 3, 8}) for (int block : {32, 128}) {
        const int n = 1000;
        std::vector<int> seen(n, 0);
        int total = 0;
        for (int rk = 0; rk < world; rk++) {
            const int nl = mpqr_part_local_cols(n, block, world, rk);
            total += nl;
            for (int l = 0; l < nl; l++) {
                const int gcol = mpqr_part_global_index(l, block, world, rk);
                REQUIRE(gcol >= 0 && gcol < n && mpqr_part_owner(gcol, block, world) == rk && mpqr_part_local_index(gcol, block, world) == l);
                seen[gcol]++;
            }
        }
        REQUIRE(total == n);
        for (int g = 0; g < n; g++) REQUIRE(seen[g] == 1);
    }
    // ---- the oracle's factorisations on ragged shapes (the reference sweep's odd ones), every variant
    const int shapes[][3] = {{6, 4, 3}, {12, 8, 5}, {97, 90, 16}, {129, 80, 16}, {100, 100, 7}};
    for (auto& s : shapes) {
        const int m = s[0], n = s[1], r = s[2];
        std::vector<float> A0((size_t)(m + 1) * n, 0.f), A, Q((size_t)m * m), R((size_t)m * n);
        orc_generate_random_matrix(A0.data(), m, n, 99);
        for (int variant = 0; variant < 4; variant++) {
            A = A0; std::fill(Q.begin(), Q.end(), 0.f);
            for (int i = 0; i < m; i++) Q[(size_t)i * m + i] = 1.f;
            if (variant == 0) orc_block_qr(A.data(), Q.data(), m, n, r);
            else if (variant == 1) orc_mixed_precision_block_qr(A.data(), Q.data(), m, n, r);
            else orc_block_qr_compact(A.data(), Q.data(), m, n, r, variant - 2);
            orc_strip_R_from_A(A.data(), R.data(), m, n);
            const double be = orc_backward_error_f64(A0.data(), R.data(), Q.data(), m, n);
            REQUIRE(std::isfinite(be) && be < (variant == 0 ? 1e-5 : 5e-3));
            REQUIRE(orc_lower_trapezoid_error(R.data(), m, n) == 0.f);
        }
        std::vector<float> T((size_t)r * r, 0.f), V((size_t)m * r, 0.f), Ap = A0;
        orc_householder_qr(Ap.data(), m, n, 0, r);
        orc_compact_wy_T(Ap.data(), T.data(), m, n, 0, r, 1);
        orc_extract_V(Ap.data(), V.data(), m, n, 0, r);
    }
    {   // fp64 explicit-H path (C++/main.cpp semantics)
        const int n = 24;
        std::vector<double> A((size_t)n * n), Q((size_t)n * n, 0.0);
        for (int i = 0; i < n * n; i++) A[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
        orc_qr_factorization_f64(A.data(), Q.data(), n);
        for (double x : A) REQUIRE(std::isfinite(x));
    }
    printf("sanitizer run ok\n");
    return 0;
}
