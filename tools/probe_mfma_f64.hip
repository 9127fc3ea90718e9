// Probe of v_mfma_f64_16x16x4_f64 operand / result layout on gfx950 (build: hipcc --offload-arch=gfx950 -O2 -o probe probe_mfma_f64.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    // hypothesis: A[i][k] from lane i + 16 k, B[k][j] from lane j + 16 k
    const int i = l & 15, k = l >> 4;
    const double a = 1.0 + i + 100.0 * k;          // A[i][k]
    const double b = 1.0 + 2.0 * i + 1000.0 * k;   // B[k][j] with j = l & 15
    double4v c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; v++) out[l * 4 + v] = c[v];
}
int main() {
    double* d; hipMalloc(&d, 256 * 8);
    probe<<<1, 64>>>(d);
    double h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // reference D[i][j] = sum_k A[i][k] B[k][j]
    double D[16][16];
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += (1.0 + i + 100.0 * k) * (1.0 + 2.0 * j + 1000.0 * k); D[i][j] = s; }
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
        const double x = h[l * 4 + v];
        if (x != D[4 * (l / 16) + v][l % 16]) okA = 0;      // layout A: i = 4 (lane/16) + v, j = lane % 16
        if (x != D[(l / 16) + 4 * v][l % 16]) okB = 0;      // layout B: i = lane/16 + 4 v
    }
    printf("layout i=4*(lane/16)+v: %d   layout i=lane/16+4*v: %d\n", okA, okB);
    printf("lane 0: %g %g %g %g  lane 16: %g %g %g %g\n", h[0], h[1], h[2], h[3], h[64], h[65], h[66], h[67]);
    printf("D[0][0]=%g D[1][0]=%g D[4][0]=%g D[0][1]=%g\n", D[0][0], D[1][0], D[4][0], D[0][1]);
    return 0;
}
