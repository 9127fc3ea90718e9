// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands, e8m0 block scales) on gfx950: operand byte -> (row, k)
// map, scale-operand semantics (build: hipcc --offload-arch=gfx950 -O2 -o probe_mfma_fp8.bin tools/probe_mfma_fp8.hip).
// Each lane supplies 32 operand bytes (8 VGPRs).  The host fills them with small integers (exact in e4m3), tries
// candidate maps and reports which one reproduces the GPU result exactly; then it checks that scale bytes 127 (2^0) leave
// the result unchanged and that scale byte 128 on A doubles it.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef int int8v __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void probe(const int* a, const int* b, float* out, int sa, int sb) {
    const int l = threadIdx.x;
    int8v av, bv;
    for (int q = 0; q < 8; q++) { av[q] = a[l * 8 + q]; bv[q] = b[l * 8 + q]; }
    floatx16 c;
    for (int e = 0; e < 16; e++) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0 /*A: fp8 e4m3*/, 0 /*B: fp8 e4m3*/, 0, sa, 0, sb);
    for (int e = 0; e < 16; e++) out[l * 16 + e] = c[e];
}
static uint8_t e4m3(int v) {            // small non-negative integers 0..8 and their negatives, exact in e4m3
    static const uint8_t tab[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
    return v >= 0 ? tab[v] : (uint8_t)(tab[-v] | 0x80);
}
int main() {
    uint8_t ab[64][32], bb[64][32]; int av[64][32], bv[64][32];
    srand(7);
    for (int l = 0; l < 64; l++) for (int q = 0; q < 32; q++) {
        av[l][q] = rand() % 9 - 4; bv[l][q] = rand() % 7 - 3;
        ab[l][q] = e4m3(av[l][q]); bb[l][q] = e4m3(bv[l][q]);
    }
    int *da, *db; float* dout;
    hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(da, ab, 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, bb, 64 * 32, hipMemcpyHostToDevice);
    float h[64][16], h2[64][16];
    probe<<<1, 64>>>(da, db, dout, 0x7F7F7F7F, 0x7F7F7F7F);
    hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
    // candidate maps: byte q of lane (r = l & 31, half = l >> 5) is element k of row r
    for (int cand = 0; cand < 3; cand++) {
        static float A[32][64], B[64][32];
        for (int l = 0; l < 64; l++) for (int q = 0; q < 32; q++) {
            const int r = l & 31, hf = l >> 5;
            int k = 0;
            if (cand == 0) k = 32 * hf + q;                          // contiguous 32 per half
            if (cand == 1) k = 16 * hf + (q & 15) + 32 * (q >> 4);   // 16-interleaved
            if (cand == 2) k = 2 * q + hf;                           // fully interleaved
            A[r][k] = (float)av[l][q]; B[k][r] = (float)bv[l][q];
        }
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int e = 0; e < 16; e++) {
            const int i = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), j = l & 31;
            float s = 0; for (int k = 0; k < 64; k++) s += A[i][k] * B[k][j];
            if (s != h[l][e]) bad++;
        }
        printf("candidate %d: %d mismatches of 1024\n", cand, bad);
    }
    probe<<<1, 64>>>(da, db, dout, 0x7F7F7F80, 0x7F7F7F7F);      // A scale byte 0 = 128 -> 2^1
    hipMemcpy(h2, dout, sizeof h2, hipMemcpyDeviceToHost);
    int dbl = 0; for (int l = 0; l < 64; l++) for (int e = 0; e < 16; e++) if (h2[l][e] == 2.f * h[l][e]) dbl++;
    printf("scale_a byte0 = 128: %d of 1024 results doubled\n", dbl);
    return 0;
}
