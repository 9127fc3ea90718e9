// panel_dev.h -- device helpers shared by the panel translation units (kernels_panel.hip, kernels_solve.hip)
#pragma once
#include "mpqr_internal.h"
#include <type_traits>

namespace mpqr {

extern __shared__ __attribute__((aligned(16))) char gh_smem[];   // dynamic LDS of the kernels that need > 64 KiB

// optional in-kernel phase timing (make EXTRA=-DMPQR_KTRACE): thread 0 of block 0 stamps s_memtime at phase
// boundaries and prints the deltas for the first few launches of each kernel
#ifdef MPQR_KTRACE
static __device__ int g_ktrace_left[8] = {3, 3, 3, 3, 3, 3, 3, 3};
#define KT_DECL long kt_[16]; int kn_ = 0; const bool kon_ = (threadIdx.x == 0 && blockIdx.x == 0)
#define KT() do { if (kon_ && kn_ < 16) kt_[kn_++] = clock64(); } while (0)
#define KT_DUMP(id, name) do { if (kon_ && atomicSub(&g_ktrace_left[id], 1) > 0) { printf("ktrace %s:", name); \
    for (int q_ = 1; q_ < kn_; q_++) printf(" %ld", kt_[q_] - kt_[q_ - 1]); printf("\n"); } } while (0)
#else
#define KT_DECL
#define KT() do {} while (0)
#define KT_DUMP(id, name) do {} while (0)
#endif
constexpr double GH_RHO_MIN = 1e-8;
constexpr double GH_SKIP_ROW_MAX = 6e-4;   // in-line deflation (kernels_solve.hip): largest B[k][j]^2 of a skipped step's dead row, relative to max_j ||a_j||^2 of the leaf
constexpr int GW = 128;          // window width

__device__ __forceinline__ double bcast_lane_d(double v, int srclane) {
    const long b = __builtin_bit_cast(long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffL), srclane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
    return __builtin_bit_cast(double, ((long)hi << 32) | ((long)lo & 0xffffffffL));
}
__device__ __forceinline__ float bcast_lane_f(float v, int srclane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), srclane));
}

// 1/sqrt(x) in fp64 from an fp32 seed and two Newton steps (2^-23 -> 2^-46 -> 2^-92); x outside the fp32 range
// takes the library path
__device__ __forceinline__ double refine_rsqrt(double x, double y) {
    y = y * fma(-0.5 * x, y * y, 1.5);
    y = y * fma(-0.5 * x, y * y, 1.5);
    return y;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    if (!(x > 1e-30 && x < 1e30)) return 1.0 / sqrt(x);
    return refine_rsqrt(x, (double)rsqrtf((float)x));
}

// ---- in-wave broadcasts: DPP row_newbcast:n = lane n of the own 16-lane row (gfx90a+).  On gfx950 (tools/ubench_valu.hip)
// v_fmac_f32_dpp costs what a plain v_fmac_f32 does (4.6 cycles per wave-instruction: the broadcast is free) and
// v_mov_b64_dpp 4.6; v_readlane_b32 + s_nop + use costs 10-20 per value.  So every 16-lane row of a chain wave keeps its OWN
// copy of the block's 16 x 16 diagonal block (lane li = column or row k0 + li) and never leaves its row.
// Software hazard (not interlocked on gfx9): a DPP read of a VGPR needs 2 wait states after the VALU write -> NOP = 1 puts an
// s_nop 1 in front where the source may just have been written.
template <int I, int NOP>
__device__ __forceinline__ double dpp_bcast64(double x) {
    double r;
    if (NOP) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(I));
    else asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(I));
    return r;
}
template <int I, int NOP>
__device__ __forceinline__ float dpp_bcast32(float x) {
    float r;
    if (NOP) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(I));
    else asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(I));
    return r;
}
// acc += (lane I of a's row) * b
template <int I, int NOP>
__device__ __forceinline__ void fmac_dpp(float& acc, float a, float b) {
    if (NOP) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(I));
    else asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(I));
}
// compile-time loop I = LO .. 15
template <int LO, typename F>
__device__ __forceinline__ void static_for16(F&& f) {
    if constexpr (LO < 16) { f(std::integral_constant<int, LO>{}); static_for16<LO + 1>(f); }
}

template <int LO, int HI, typename F>
__device__ __forceinline__ void static_range(F&& f) {
    if constexpr (LO < HI) { f(std::integral_constant<int, LO>{}); static_range<LO + 1, HI>(f); }
}
typedef float floatx16p __attribute__((ext_vector_type(16)));
constexpr int TP = 128, TPS = 129;

// 32 x 32 tile of A (32 x K, LDS) * B (K x 32, LDS) on the exact-f32 MFMA; odd row strides: conflict-free
// k runs over [klo, khi), both multiples of 16 (callers skip the zero part of triangular factors)
__device__ __forceinline__ floatx16p lds_mm32(const float* A, int lda, const float* B, int ldb, int klo, int khi, int lane) {
    const int r = lane & 31, kk = lane >> 5;
    floatx16p acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    for (int k1 = klo; k1 < khi; k1 += 16) {             // 8 steps' LDS reads in flight together
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { av[u] = A[r * lda + k1 + 2 * u + kk]; bv[u] = B[(k1 + 2 * u + kk) * ldb + r]; }
#pragma unroll
        for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ void lds_store32(float* C, int ldc, const floatx16p& acc, float scale, int lane) {
    const int r = lane & 31, kk = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; e++) C[((e & 3) + 8 * (e >> 2) + 4 * kk) * ldc + r] = scale * acc[e];
}

// Ts = X^{-1} for the upper-triangular X = striu(Ss) + diag(1 / tdiag), 128 x 128 in LDS (stride TPS), Ts zero on
// entry.  nblk = active 32-blocks.  Diagonal blocks: row a of the inverse depends on row a only, lane a runs the
// column recurrence in registers.  Then two merge levels X_LR -> -T_L (X_LR T_R) on the MFMA; the intermediate
// product overwrites X_LR in Ss, so the lower triangle of Ts stays zero.  Needs >= 4 waves; ends with a barrier.
__device__ __forceinline__ void tri_inverse_128(float* Ss, const float* tdiag, float* Ts, int nblk, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
#ifdef MPQR_KTRACE
    long ti_[8]; int tn_ = 0;
    if (tid == 0 && blockIdx.x == 0) ti_[tn_++] = clock64();
#endif
    // diagonal 16 x 16 blocks: row a of the inverse depends on row a only -- lane a runs the column recurrence in registers
    // (eight blocks: two per wave, one per 16-lane row of the first 32 lanes of waves 0-3)
    if (wave < nblk && lane < 32) {                         // (four blocks per wave on all 64 lanes of waves 0 and 1 measured 6.4 k cycles against 2.0 k)
        const int base = 32 * wave + 16 * (lane >> 4), a = lane & 15;
        // lane l of the block's 16-lane row holds ROW l of the block (sc[i] = S[l][i]) and the diagonal entry of row l: 17 loads, all in
        // flight together.  Column i of the recurrence then takes S[q][i] from lane q through the DPP operand of the FMA (row_newbcast,
        // free on gfx950: tools/ubench_valu.hip) -- straight-line code.  (Written with S[q][i] read from LDS where it is used, the
        // compiler put each column's loads inside a divergent branch behind their own s_waitcnt: 74 LDS round trips, 5 - 8.5 k cycles of
        // the inverse's 20 - 23 k; in-kernel stamps, tools/ktrace_solve.sh.)
        float sc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) sc[i] = Ss[(base + a) * TPS + base + i];
        const float tda = tdiag[base + a];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float tr[16];
        static_for16<0>([&](auto II) {
            constexpr int i = decltype(II)::value;
            const float tii = dpp_bcast32<i, 0>(tda);
            float ps[4] = {0.f, 0.f, 0.f, 0.f};            // four partial sums: a quarter of the dependent chain
            static_range<0, i>([&](auto QQ) {
                constexpr int q = decltype(QQ)::value;
                fmac_dpp<q, 0>(ps[q & 3], sc[i], tr[q]);   // += S[q][i] (lane q's sc[i]) * tr[q]
            });
            const float sum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
            tr[i] = (a < i) ? -tii * sum : (a == i ? tii : 0.f);
        });
#ifdef MPQR_KTRACE
        if (tid == 0 && blockIdx.x == 0) ti_[tn_++] = clock64();
#endif
#pragma unroll
        for (int i = 0; i < 16; i++) Ts[(base + a) * TPS + base + i] = tr[i];
    }
#ifdef MPQR_KTRACE
    if (tid == 0 && blockIdx.x == 0) ti_[tn_++] = clock64();
#endif
    __syncthreads();
#ifdef MPQR_KTRACE
    if (tid == 0 && blockIdx.x == 0) ti_[tn_++] = clock64();
#endif
    // 16 -> 32 inside every 32-block (one wave each): with L / R the two halves, T_LR = -T_L (S_LR T_R).  On 32 x 32 tiles:
    // rows < 16 of  S_blk[:, 16:32] T_blk[16:32, :]  (k range 16..32) are S_LR T_R in the columns >= 16; it replaces S_LR,
    // then rows < 16 of  T_blk[:, 0:16] (that)[0:16, :]  (k range 0..16) are T_L (S_LR T_R).
    {
        const bool has = wave < nblk;
        const int base = 32 * wave, r = lane & 31, kk = lane >> 5;
        floatx16p acc;
        if (has) acc = lds_mm32(&Ss[base * TPS + base], TPS, &Ts[base * TPS + base], TPS, 16, 32, lane);
        __syncthreads();
        if (has && r >= 16) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = (e & 3) + 8 * (e >> 2) + 4 * kk;
                if (i < 16) Ss[(base + i) * TPS + base + r] = acc[e];
            }
        }
        __syncthreads();
        if (has) {
            acc = lds_mm32(&Ts[base * TPS + base], TPS, &Ss[base * TPS + base], TPS, 0, 16, lane);
            if (r >= 16) {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int i = (e & 3) + 8 * (e >> 2) + 4 * kk;
                    if (i < 16) Ts[(base + i) * TPS + base + r] = -acc[e];
                }
            }
        }
        __syncthreads();
    }
#ifdef MPQR_KTRACE
    if (tid == 0 && blockIdx.x == 0) ti_[tn_++] = clock64();
#endif
    for (int half = 32; half < 32 * nblk; half *= 2) {
        const int ts = half / 32, npair = TP / (2 * half);
        const bool has = wave < npair * ts * ts;
        const int p = wave / (ts * ts), t = wave % (ts * ts), bi = t / ts, bj = t % ts;
        const int L0 = p * 2 * half, R0 = L0 + half;
        floatx16p acc;
        if (has) acc = lds_mm32(&Ss[(L0 + 32 * bi) * TPS + R0], TPS, &Ts[R0 * TPS + R0 + 32 * bj], TPS, 0, 32 * (bj + 1), lane);   // T_R upper
        __syncthreads();
        if (has) lds_store32(&Ss[(L0 + 32 * bi) * TPS + R0 + 32 * bj], TPS, acc, 1.f, lane);
        __syncthreads();
        if (has) {
            acc = lds_mm32(&Ts[(L0 + 32 * bi) * TPS + L0], TPS, &Ss[L0 * TPS + R0 + 32 * bj], TPS, 32 * bi, half, lane);   // T_L upper
            lds_store32(&Ts[(L0 + 32 * bi) * TPS + R0 + 32 * bj], TPS, acc, -1.f, lane);
        }
        __syncthreads();
#ifdef MPQR_KTRACE
        if (tid == 0 && blockIdx.x == 0 && tn_ < 8) ti_[tn_++] = clock64();
#endif
    }
#ifdef MPQR_KTRACE
    if (tid == 0 && blockIdx.x == 0 && atomicSub(&g_ktrace_left[4], 1) > 0) {
        printf("ktrace tri_inverse rec|store|barrier|merge16|merge32|merge64:");
        for (int q = 1; q < tn_; q++) printf(" %ld", ti_[q] - ti_[q - 1]);
        printf("\n");
    }
#endif
}

}  // namespace mpqr
