// tools/test_solve3.hip -- gh_solve3 (blocked) against gh_solve_kernel (step by step) on the same Gram matrix / top block:
// element-wise differences of every output and launch times.  Development tool, not part of the product or the tests.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../mixedprecisionblockqr_amd/csrc tools/test_solve3.hip -L mixedprecisionblockqr_amd -lmpqr
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "mpqr_internal.h"

using namespace mpqr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float float4s __attribute__((ext_vector_type(4)));
// layout probe: D = A B with A[i][k] = i + 100 k, B[k][j] = (k == 0) -> D[i][j] = i  (j any); then with B[k][j] = j (k == 0 only)
__global__ void probe_f32_16x16x4(float* out) {
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    float4s acc = {0.f, 0.f, 0.f, 0.f};
    // A[i][k] from lane i + 16 k ; B[k][j] from lane j + 16 k
    const float a = (float)(li) + 100.f * (float)lk;           // A[i = li][k = lk]
    const float b = (lk == 0) ? 1.f + 1000.f * (float)li : 0.f; // B[k = lk][j = li] = 1 + 1000 j for k = 0
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);   // D[i][j] = A[i][0] B[0][j] = i (1 + 1000 j)
    for (int e = 0; e < 4; e++) out[lane * 4 + e] = acc[e];
}

static uint64_t sm64(uint64_t& s) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

struct Dev { float *A, *vdiag, *Cv; half_t *Vh, *Vt; int* flag; };

int main(int argc, char** argv) {
    const int tall = argc > 1 ? atoi(argv[1]) : 2048;       // rows below (and including) the top block
    const int w = argc > 2 ? atoi(argv[2]) : 128;
    const int off = argc > 3 ? atoi(argv[3]) : 0;
    const int iters = argc > 4 ? atoi(argv[4]) : 20;
    const int mode = argc > 5 ? atoi(argv[5]) : 0;          // 1: a nearly dependent column (flag expected)
    const int cb = 128, c0 = cb + off, c1 = c0 + w, mrows = c0 + tall, lda = 384;
    {   // ---- f32 16x16x4 layout probe
        float* d; CK(hipMalloc(&d, 256 * 4));
        hipLaunchKernelGGL(probe_f32_16x16x4, dim3(1), dim3(64), 0, 0, d);
        std::vector<float> h(256); CK(hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int lane = 0; lane < 64; lane++) for (int e = 0; e < 4; e++) {
            const int i = 4 * (lane >> 4) + e, j = lane & 15;           // assumed: D[i][j] in lane j + 16 (i / 4), element i % 4
            if (h[lane * 4 + e] != (float)i * (1.f + 1000.f * j)) bad++;
        }
        printf("probe v_mfma_f32_16x16x4_f32 layout (D[i][j]: lane j + 16 (i/4), element i%%4): %s\n", bad ? "MISMATCH" : "ok");
        if (bad) { for (int lane = 0; lane < 64; lane += 16) printf("  lane %d: %g %g %g %g\n", lane, h[lane*4], h[lane*4+1], h[lane*4+2], h[lane*4+3]); }
        CK(hipFree(d));
    }
    // ---- panel: rows c0.., window columns [cb, cb+128)
    std::vector<float> P((size_t)tall * 128);
    uint64_t seed = 1234;
    for (auto& x : P) x = (float)(sm64(seed) >> 40) * (1.0f / 16777216.0f);
    if (mode == 1) for (int r = 0; r < tall; r++) P[(size_t)r * 128 + off + 5] = P[(size_t)r * 128 + off + 3] * (1.f + 1e-6f * (r & 1));
    std::vector<double> G(128 * 128, 0.0);
    for (int r = 0; r < tall; r++) {
        const float* row = &P[(size_t)r * 128];
        for (int i = 0; i < 128; i++) { const double ai = row[i]; for (int j = i; j < 128; j++) G[i * 128 + j] += ai * (double)row[j]; }
    }
    for (int i = 0; i < 128; i++) for (int j = 0; j < i; j++) G[i * 128 + j] = G[j * 128 + i];
    std::vector<float> Ah((size_t)mrows * lda, 0.f);
    for (int r = 0; r < tall; r++) for (int c = 0; c < 128; c++) Ah[(size_t)(c0 + r) * lda + cb + c] = P[(size_t)r * 128 + c];
    double* dG; CK(hipMalloc(&dG, 128 * 128 * 8)); CK(hipMemcpy(dG, G.data(), 128 * 128 * 8, hipMemcpyHostToDevice));
    float* dA0; CK(hipMalloc(&dA0, Ah.size() * 4)); CK(hipMemcpy(dA0, Ah.data(), Ah.size() * 4, hipMemcpyHostToDevice));
    const long ldvh = lda, ldvt = mrows;
    Dev d[2];
    for (int v = 0; v < 2; v++) {
        CK(hipMalloc(&d[v].A, Ah.size() * 4)); CK(hipMalloc(&d[v].vdiag, lda * 4)); CK(hipMalloc(&d[v].Cv, 128 * 128 * 4));
        CK(hipMalloc(&d[v].Vh, (size_t)mrows * ldvh * 2)); CK(hipMalloc(&d[v].Vt, (size_t)lda * ldvt * 2)); CK(hipMalloc(&d[v].flag, 4));
        CK(hipMemset(d[v].Vh, 0, (size_t)mrows * ldvh * 2)); CK(hipMemset(d[v].Vt, 0, (size_t)lda * ldvt * 2));
        CK(hipMemset(d[v].vdiag, 0, lda * 4)); CK(hipMemset(d[v].flag, 0, 4)); CK(hipMemset(d[v].Cv, 0xff, 128 * 128 * 4));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> times[2];
    for (int v = 0; v < 2; v++) {
        LeafArgs a{};
        a.A = d[v].A; a.lda = lda; a.mrows = mrows; a.cb = cb; a.c0 = c0; a.c1 = c1;
        a.Vh = d[v].Vh; a.ldvh = ldvh; a.Vt = d[v].Vt; a.ldvt = ldvt; a.vdiag = d[v].vdiag; a.Wk = nullptr;
        for (int it = 0; it < iters; it++) {
            CK(hipMemcpyAsync(d[v].A, dA0, Ah.size() * 4, hipMemcpyDeviceToDevice, st));
            CK(hipMemsetAsync(d[v].flag, 0, 4, st));
            CK(hipEventRecord(e0, st));
            if (v == 0) launch_gh_solve(a, dG, d[v].Cv, d[v].flag, st); else launch_gh_solve3(a, dG, d[v].Cv, d[v].flag, st);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); times[v].push_back(ms * 1000.f);
        }
        CK(hipGetLastError());
    }
    for (int v = 0; v < 2; v++) {
        std::sort(times[v].begin(), times[v].end());
        printf("%s: min %.1f us  median %.1f us  (%d launches)\n", v == 0 ? "gh_solve " : "gh_solve3", times[v][0], times[v][times[v].size() / 2], iters);
    }
    // ---- compare
    std::vector<float> A[2], Cv[2], vd[2]; std::vector<half_t> Vh[2], Vt[2]; int fl[2];
    for (int v = 0; v < 2; v++) {
        A[v].resize(Ah.size()); Cv[v].resize(128 * 128); vd[v].resize(lda); Vh[v].resize((size_t)mrows * ldvh); Vt[v].resize((size_t)lda * ldvt);
        CK(hipMemcpy(A[v].data(), d[v].A, Ah.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(Cv[v].data(), d[v].Cv, 128 * 128 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(vd[v].data(), d[v].vdiag, lda * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(Vh[v].data(), d[v].Vh, Vh[v].size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(Vt[v].data(), d[v].Vt, Vt[v].size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&fl[v], d[v].flag, 4, hipMemcpyDeviceToHost));
    }
    auto cmp = [&](const char* name, auto get0, auto get1, int rows, int cols) {
        double md = 0, mx = 0; int nanc = 0, wi = -1, wj = -1;
        for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) {
            const double x = get0(i, j), y = get1(i, j);
            if (std::isnan(x) || std::isnan(y)) { nanc++; continue; }
            if (std::fabs(x - y) > md) { md = std::fabs(x - y); wi = i; wj = j; }
            mx = std::max(mx, std::fabs(x));
        }
        printf("  %-6s max|diff| %.3e  (max|ref| %.3e, rel %.3e, at %d,%d, nan %d)\n", name, md, mx, mx > 0 ? md / mx : 0.0, wi, wj, nanc);
    };
    printf("flags: gh_solve %d  gh_solve3 %d\n", fl[0], fl[1]);
    cmp("R", [&](int i, int j) { return j >= i ? A[0][(size_t)(c0 + i) * lda + c0 + j] : 0.f; }, [&](int i, int j) { return j >= i ? A[1][(size_t)(c0 + i) * lda + c0 + j] : 0.f; }, w, w);
    cmp("Vtop", [&](int i, int j) { return j < i ? A[0][(size_t)(c0 + i) * lda + c0 + j] : 0.f; }, [&](int i, int j) { return j < i ? A[1][(size_t)(c0 + i) * lda + c0 + j] : 0.f; }, w, w);
    cmp("Arest", [&](int i, int j) { return A[0][(size_t)(c0 - 1 + i) * lda + j]; }, [&](int i, int j) { return (i >= 1 && i <= w && j >= c0 && j < c1) ? A[0][(size_t)(c0 - 1 + i) * lda + j] : A[1][(size_t)(c0 - 1 + i) * lda + j]; }, w + 2, lda);
    cmp("Vh", [&](int i, int j) { return (float)Vh[0][(size_t)(c0 + i) * ldvh + cb + j]; }, [&](int i, int j) { return (float)Vh[1][(size_t)(c0 + i) * ldvh + cb + j]; }, w, 128);
    cmp("Vt", [&](int i, int j) { return (float)Vt[0][(size_t)(cb + i) * ldvt + c0 + j]; }, [&](int i, int j) { return (float)Vt[1][(size_t)(cb + i) * ldvt + c0 + j]; }, 128, w);
    cmp("vdiag", [&](int i, int j) { return vd[0][j]; }, [&](int i, int j) { return vd[1][j]; }, 1, lda);
    cmp("Cv", [&](int i, int j) { return Cv[0][i * 128 + j]; }, [&](int i, int j) { return Cv[1][i * 128 + j]; }, 128, 128);
    return 0;
}
