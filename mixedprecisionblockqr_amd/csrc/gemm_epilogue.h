// gemm_epilogue.h -- shared device epilogue of the MFMA GEMM kernels
#pragma once
#include <hip/hip_runtime.h>
#ifndef MPQR_EPI_DEPTH
#define MPQR_EPI_DEPTH 1
#endif
namespace mpqr {
// Read-modify-write epilogue  C[m][n] -= alpha * acc  for NI x NJ 32x32 MFMA sub-tiles per wave (M % 32 == 0, so a
// sub-tile is valid or invalid as a whole).  The 16 old values of a sub-tile are loaded back to back (column index
// clamped instead of branched on: one wait per sub-tile instead of one per element), the loads of the next sub-tile
// of the next MPQR_EPI_DEPTH sub-tiles are issued before the current one is stored (depth 2 / 3 measured slower: the extra
// 16 / 32 registers spill in the 256-register kernels: 478 / 330 vs 575 TFLOP/s), only the stores are predicated.
template <int NI, int NJ, typename ACC>
__device__ __forceinline__ void epilogue_sub_f32(const ACC (&acc)[NI][NJ], float* __restrict__ C, long ldc, int M, int N,
                                                 int col_lo, float alpha, int row_base, int col_base, int r, int h,
                                                 _Float16* __restrict__ Ct = nullptr, long ldct = 0, float cts = 1.f) {
    constexpr int D = MPQR_EPI_DEPTH;                      // sub-tiles whose loads are in flight ahead of the one being stored
    float oldv[D + 1][16];
    auto tile_ptr = [&](int t) -> float* {
        const int i = t / NJ, j = t % NJ;
        const int n = min(col_base + j * 32 + r, N - 1);
        const int m0 = min(row_base + i * 32, M - 32) + 4 * h;
        return C + (long)m0 * ldc + n;
    };
    auto load_tile = [&](int t, float (&dst)[16]) {
        const float* p = tile_ptr(t);
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int e = 0; e < 4; e++) dst[q * 4 + e] = p[(long)e * ldc];
            p += 8 * ldc;
        }
    };
#pragma unroll
    for (int t = 0; t < D && t < NI * NJ; t++) load_tile(t, oldv[t % (D + 1)]);
#pragma unroll
    for (int t = 0; t < NI * NJ; t++) {
        if (t + D < NI * NJ) load_tile(t + D, oldv[(t + D) % (D + 1)]);
        const int i = t / NJ, j = t % NJ;
        const int n = col_base + j * 32 + r;
        const bool ok = (n < N) && (n >= col_lo) && (row_base + i * 32 < M);
        float* p = tile_ptr(t);
        if (ok) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int e = 0; e < 4; e++) p[(long)e * ldc] = oldv[t % (D + 1)][q * 4 + e] - alpha * acc[i][j][q * 4 + e];
                p += 8 * ldc;
            }
            if (Ct) {                                      // transposed fp16 shadow: 4 consecutive rows = one 8-byte store
                typedef _Float16 half4e __attribute__((ext_vector_type(4)));
                const int m0 = min(row_base + i * 32, M - 32) + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    half4e hv;
#pragma unroll
                    for (int e = 0; e < 4; e++) hv[e] = (_Float16)(cts * (oldv[t % (D + 1)][q * 4 + e] - alpha * acc[i][j][q * 4 + e]));
                    *(half4e*)&Ct[(long)n * ldct + m0 + 8 * q] = hv;
                }
            }
        }
    }
}
}  // namespace mpqr
