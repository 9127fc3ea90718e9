// kernels_gemm2.hip -- large-shape variant of the fp16-operand / fp32-accumulate MFMA GEMM family.
//
// Same contract as gemm_f16_kernel (kernels_gemm.hip):  C[M x N] = A[M x K] * Bt[N][K]^T  with the three A
// staging modes and three epilogues, but built for the far trailing update and Q formation, where M, N are in
// the thousands:
//   * 256 x 256 x 64 tile, 512 threads = 8 waves as 2 (M) x 4 (N); each wave owns 128 x 64 = 4 x 2 MFMA
//     32x32x16 tiles (128 accumulator registers);
//   * LDS double buffer, 2 x (32 KiB A + 32 KiB B) = 128 KiB of the CU's 160 KiB, one barrier per K-tile;
//   * Bt (already fp16, k contiguous) goes HBM -> LDS directly with global_load_lds_dwordx4 (no VGPR staging);
//     the LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and
//     mirrored on the ds_read side:  physical 16-B chunk = logical chunk ^ ((row >> 1) & 7)  within 128-B rows;
//   * A is staged through registers (fp32 sources must be converted, A2 additionally transposed) into the same
//     swizzled image, issued before the MFMA block of the current tile and written after it;
//   * blockIdx -> tile map walks 4 x 8 tile groups inside each XCD's share of the grid, so the 32 workgroups
//     sharing an L2 reuse 4 A panels and 8 B panels instead of streaming them.
#include <type_traits>
#include "mpqr_internal.h"
#include "gemm_epilogue.h"

namespace mpqr {

typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace g2 {
typedef uint32_t U4 __attribute__((ext_vector_type(4)));   // first-class vector (register-resident staging arrays)

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef half_t half2v __attribute__((ext_vector_type(2)));
    half2v h = {(half_t)a, (half_t)b};
    return __builtin_bit_cast(uint32_t, h);
}
}  // namespace g2

extern __shared__ __attribute__((aligned(16))) char g2_smem[];

// WM x WN waves, each wave owns 128 x 64 of the (128 WM) x (64 WN) tile; BK = 32 or 64.
//   <2,4,64>: 256 x 256 x 64, 512 threads, 128 KiB LDS, one workgroup per CU
//   <2,2,32>: 256 x 128 x 32, 256 threads,  48 KiB LDS, two workgroups per CU (one's read-modify-write epilogue
//             overlaps the other's MFMA loop)
template <int AM, int EM, int WM, int WN, int BK>
__global__ __launch_bounds__(64 * WM * WN) void gemm2_f16_kernel(GemmArgs g, int tilesM, int tilesN) {
    using namespace g2;
    constexpr int BM = 128 * WM, BN = 64 * WN, NT = 64 * WM * WN;
    constexpr int ROWB = BK * 2;                       // bytes per LDS row
    constexpr int CPR = BK / 8;                        // 16-B chunks per row
    constexpr int RB = 256 / ROWB;                     // rows per 256-B bank row
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    // byte offset of logical chunk c of row r: XOR swizzle makes 16 consecutive rows hit 16 distinct 16-B bank slots
    auto swz = [](int r, int c) -> int { return r * ROWB + ((c ^ ((r / RB) & (CPR - 1))) << 4); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // ---- XCD-aware tile order (blocks b and b+8 share an XCD): contiguous run of the sequence per XCD, 4 x 8 groups
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    const int seq = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
    const int groupsN = (tilesN + 7) / 8;
    const int grp = seq / 32, within = seq % 32;
    const int tm = (grp / groupsN) * 4 + within / 8;
    const int tn = (grp % groupsN) * 8 + within % 8;
    if (tm >= tilesM || tn >= tilesN) return;
    const int bm = tm * BM, bn = tn * BN;

    // split-K over blockIdx.y (E_STORE_F32 only): slice z covers K-tiles [kt0, kt1) and writes slab z
    const int ktiles_all = g.K / BK;
    const int z = blockIdx.y;
    const int per = (ktiles_all + g.nsplit - 1) / g.nsplit;
    const int kt0 = z * per, kt1 = min(ktiles_all, kt0 + per);
    char* const As0 = g2_smem;
    char* const Bs0 = g2_smem + A_BYTES;

    // ---- A staging through registers
    constexpr int NA = BM * CPR / NT;                  // 16-B chunks per thread (A_H16 / A_F32)
    constexpr int NBLK = (BK / 4) * (BM / 4) / NT;     // 4 x 4 fp32 blocks per thread (A_F32T)
    typedef float F4 __attribute__((ext_vector_type(4)));   // first-class vectors: the loads land in their final registers
    U4 ra[NA];
    F4 raT[2][NBLK * 4];                                 // A_F32T: two register sets, loads run two K tiles ahead
    // A_F32T: per-thread row pointers advance by BK rows per K tile (no 64-bit multiplies in the loop)
    const float* pT[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; i++) {
        const int id = tid + NT * i, mg = id % (BM / 4), kg = id / (BM / 4);
        pT[i] = (const float*)g.A + (long)(kt0 * BK + kg * 4) * g.lda + bm + mg * 4;
    }
    const long ldaT = g.lda, stepT = (long)BK * g.lda;
    auto load_A = [&](int kt) {
        const int k = kt * BK;
        if (AM == A_H16) {
            const half_t* A = (const half_t*)g.A;
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const int c = tid + NT * i, row = c / CPR, kc = c % CPR;
                ra[i] = *(const U4*)(A + (long)(bm + row) * g.lda + k + kc * 8);
            }
        } else if (AM == A_F32T) {
            (void)k;                                      // handled by load_AT below

        } else {
            const float* A = (const float*)g.A;
            const float sc = g.in_scale;
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const int c = tid + NT * i, row = c / CPR, kc = c % CPR;
                const float* p = A + (long)(bm + row) * g.lda + k + kc * 8;
                float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
                for (int sl = 0; sl < g.nslab_in; sl++) {
                    const float4 a0 = *(const float4*)(p + (long)sl * g.slab_in_stride);
                    const float4 a1 = *(const float4*)(p + (long)sl * g.slab_in_stride + 4);
                    s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
                    s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
                }
                U4 v;
                v.x = pack2(s0.x * sc, s0.y * sc); v.y = pack2(s0.z * sc, s0.w * sc);
                v.z = pack2(s1.x * sc, s1.y * sc); v.w = pack2(s1.z * sc, s1.w * sc);
                ra[i] = v;
            }
        }
    };
    auto load_AT = [&](auto set) {                       // next K tile of the fp32 operand into register set `set`
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < NBLK; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) raT[S][i * 4 + j] = *(const F4*)(pT[i] + j * ldaT);
            pT[i] += stepT;                               // called once per K tile, in order
        }
    };
    auto store_AT = [&](auto set, int stage) {           // convert + transpose register set `set` into LDS stage `stage`
        constexpr int S = decltype(set)::value;
        char* As = As0 + stage * STAGE;
        const float sc = g.in_scale;
#pragma unroll
        for (int i = 0; i < NBLK; i++) {
            const int id = tid + NT * i, mg = id % (BM / 4), kg = id / (BM / 4);
            const F4 v0 = raT[S][i * 4 + 0], v1 = raT[S][i * 4 + 1], v2 = raT[S][i * 4 + 2], v3 = raT[S][i * 4 + 3];
            const int half_off = (kg & 1) * 8, chunk = kg >> 1;
            uint2 w;
            w.x = pack2(v0.x * sc, v1.x * sc); w.y = pack2(v2.x * sc, v3.x * sc); *(uint2*)(As + swz(mg * 4 + 0, chunk) + half_off) = w;
            w.x = pack2(v0.y * sc, v1.y * sc); w.y = pack2(v2.y * sc, v3.y * sc); *(uint2*)(As + swz(mg * 4 + 1, chunk) + half_off) = w;
            w.x = pack2(v0.z * sc, v1.z * sc); w.y = pack2(v2.z * sc, v3.z * sc); *(uint2*)(As + swz(mg * 4 + 2, chunk) + half_off) = w;
            w.x = pack2(v0.w * sc, v1.w * sc); w.y = pack2(v2.w * sc, v3.w * sc); *(uint2*)(As + swz(mg * 4 + 3, chunk) + half_off) = w;
        }
    };
    auto store_A = [&](int stage) {
        char* As = As0 + stage * STAGE;
        if (AM == A_F32T) {
        } else {
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const int c = tid + NT * i, row = c / CPR, kc = c % CPR;
                *(U4*)(As + swz(row, kc)) = ra[i];
            }
        }
    };
    // ---- B staging: HBM -> LDS directly.  One wave-instruction fills 1 KiB = RPI rows, lane-linear; lane l lands on
    // row rr = rb + l / CPR, physical chunk l % CPR, so it must FETCH the logical chunk that the swizzle maps there.
    constexpr int RPI = 1024 / ROWB;                   // rows per wave-instruction
    constexpr int NB = BN / RPI / (NT / 64);           // wave-instructions per wave
    auto stage_B = [&](int stage, int kt) {
        char* Bs = Bs0 + stage * STAGE;
        const int k = kt * BK;
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int rb = (i * (NT / 64) + wave) * RPI;
            const int rr = rb + lane / CPR;
            const int c = (lane % CPR) ^ ((rr / RB) & (CPR - 1));
            const half_t* src = g.Bt + (long)(bn + rr) * g.ldb + k + c * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Bs + rb * ROWB), 16, 0, 0);
        }
    };

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int wm = (wave / WN) * 128, wn = (wave % WN) * 64;

    auto mma_tile = [&](int cur) {
        const char* As = As0 + cur * STAGE;
        const char* Bs = Bs0 + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ks++) {
            half8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = *(const half8*)(As + swz(wm + i * 32 + r, ks * 2 + h));
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = *(const half8*)(Bs + swz(wn + j * 32 + r, ks * 2 + h));
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    if (AM == A_F32T) {
        // fp32 operand: HBM latency is longer than one K tile of MFMA work, so its loads run TWO tiles ahead (two
        // register sets); the fp16 operand's LDS-DMA runs one tile ahead.  vmcnt retires in issue order:
        // [A(kt+1)] [B(kt+1)] [A(kt+2)] -- the barrier needs B(kt+1) only.
        std::integral_constant<int, 0> S0; std::integral_constant<int, 1> S1;
        if (kt0 < kt1) {
            stage_B(kt0 & 1, kt0);
            load_AT(S0);
            if (kt0 + 1 < kt1) load_AT(S1);
            store_AT(S0, kt0 & 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        auto step = [&](auto setp, auto setq, int kt) {    // set p held tile kt (already in LDS), set q holds tile kt+1
            const int cur = kt & 1;
            const bool more = kt + 1 < kt1, more2 = kt + 2 < kt1;
            if (more) stage_B(cur ^ 1, kt + 1);
            if (more2) load_AT(setp);
            mma_tile(cur);
            if (more) store_AT(setq, cur ^ 1);
            if (more2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NBLK * 4) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        for (int kt = kt0; kt < kt1; kt += 2) {
            step(S0, S1, kt);
            if (kt + 1 < kt1) step(S1, S0, kt + 1);
        }
    } else {
        if (kt0 < kt1) {
            stage_B(kt0 & 1, kt0);
            load_A(kt0);
            store_A(kt0 & 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = kt0; kt < kt1; kt++) {
            const int cur = kt & 1;
            const bool more = kt + 1 < kt1;
            if (more) { stage_B(cur ^ 1, kt + 1); load_A(kt + 1); }
            mma_tile(cur);
            if (more) store_A(cur ^ 1);          // the other stage was last read one iteration ago (barrier below)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    const float alpha = g.alpha;
    if (EM == E_SUB_F32) {
        epilogue_sub_f32<4, 2>(acc, (float*)g.C, g.ldc, g.M, g.N, g.col_lo, alpha, bm + wm, bn + wn, r, h);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = bn + wn + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int m = bm + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < g.M && n < g.N) {
                    const float v = alpha * acc[i][j][e];
                    if (EM == E_STORE_F32) ((float*)g.C)[(long)z * g.slab_out_stride + (long)m * g.ldc + n] = v;
                    else ((half_t*)g.C)[(long)m * g.ldc + n] = (half_t)v;
                }
            }
        }
}

template <int AM, int EM, int WM, int WN, int BK>
static void launch2(const GemmArgs& g, hipStream_t s) {
    constexpr int BM = 128 * WM, BN = 64 * WN, NT = 64 * WM * WN;
    constexpr int LDS = 2 * (BM + BN) * BK * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm2_f16_kernel<AM, EM, WM, WN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    const int groups = ((tilesM + 3) / 4) * ((tilesN + 7) / 8);
    GemmArgs a = g;
    if (a.nslab_in < 1) a.nslab_in = 1;
    if (a.nsplit < 1 || EM != E_STORE_F32) a.nsplit = 1;
    hipLaunchKernelGGL((gemm2_f16_kernel<AM, EM, WM, WN, BK>), dim3(groups * 32, a.nsplit), dim3(NT), LDS, s, a, tilesM, tilesN);
}

// ------------------------------------------------------------------ all-DMA 4-stage ring (both operands fp16)
// C[M x N] (-)= A[M][K] * Bt[N][K]^T with A and Bt both fp16 and k contiguous: the far trailing update
// A2 -= V Y^T and the Q-formation twin, the dominant kernels of the factorisation.  256 x 256 x 32 tiles,
// 512 threads = 8 waves (2 x 4), a ring of 4 LDS stages (4 x 32 KiB); A and Bt both go HBM -> LDS with
// global_load_lds_dwordx4 (4 per wave and K-tile), three K-tiles in flight behind a COUNTED s_waitcnt vmcnt and
// a raw s_barrier (one per K-tile; __syncthreads() would drain the DMA queue).  Source-side XOR swizzle as above.
template <int EM>
__global__ __launch_bounds__(512) void gemm3_f16_kernel(GemmArgs g, int tilesM, int tilesN) {
    using namespace g2;
    constexpr int BM = 256, BN = 256, BK = 32, NS = 4;
    constexpr int ROWB = 64, CPR = 4, RB = 4;
    constexpr int A_BYTES = BM * ROWB, STAGE = 2 * A_BYTES;
    auto swz = [](int r, int c) -> int { return r * ROWB + ((c ^ ((r / RB) & (CPR - 1))) << 4); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    const int seq = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
    const int groupsN = (tilesN + 7) / 8;
    const int grp = seq / 32, within = seq % 32;
    const int tm = (grp / groupsN) * 4 + within / 8;
    const int tn = (grp % groupsN) * 8 + within % 8;
    if (tm >= tilesM || tn >= tilesN) return;
    const int bm = tm * BM, bn = tn * BN;
    const int ktiles = g.K / BK;
    const half_t* const A = (const half_t*)g.A;

    // lane l of wave-instruction (i, wave) lands on row rb + l/4, physical chunk l%4 of a 16-row slab
    auto issue = [&](int kt) {                              // kt past the end re-fetches the last tile into a free stage
        char* base = g2_smem + (kt & (NS - 1)) * STAGE;
        const int k = min(kt, ktiles - 1) * BK;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int rb = (i * 8 + wave) * 16;
            const int rr = rb + (lane >> 2);
            const int c = (lane & 3) ^ ((rr >> 2) & 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (long)(bm + rr) * g.lda + k + c * 8),
                                             (__attribute__((address_space(3))) void*)(base + rb * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.Bt + (long)(bn + rr) * g.ldb + k + c * 8),
                                             (__attribute__((address_space(3))) void*)(base + A_BYTES + rb * ROWB), 16, 0, 0);
        }
    };

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;

    for (int s = 0; s < 3; s++) issue(s);
    for (int kt = 0; kt < ktiles; kt++) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // exactly three tiles are always in flight: kt has landed
        __builtin_amdgcn_s_barrier();                        // tile kt landed everywhere; stage (kt-1)&3 is free
        const char* As = g2_smem + (kt & (NS - 1)) * STAGE;
        const char* Bs = As + A_BYTES;
        half8 a[2][4], b[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
#pragma unroll
            for (int i = 0; i < 4; i++) a[ks][i] = *(const half8*)(As + swz(wm + i * 32 + r, ks * 2 + h));
#pragma unroll
            for (int j = 0; j < 2; j++) b[ks][j] = *(const half8*)(Bs + swz(wn + j * 32 + r, ks * 2 + h));
        }
        issue(kt + 3);                                       // unconditional: keeps the loop body one basic block
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
        // issue order: the 12 fragment reads, then one LDS-DMA after every 4 MFMAs (its ~60-100 issue cycles hide in
        // the matrix pipe's shadow instead of stalling both waves of the SIMD right after the barrier)
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the three trailing (redundant) tiles
    __builtin_amdgcn_s_barrier();
    const float alpha = g.alpha;
    if (EM == E_SUB_F32) {
        // (an LDS-staged variant with 16-B accesses along full rows was measured: 527 vs 559 TFLOP/s -- not kept)
        epilogue_sub_f32<4, 2>(acc, (float*)g.C, g.ldc, g.M, g.N, g.col_lo, alpha, bm + wm, bn + wn, r, h);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = bn + wn + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int m = bm + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < g.M && n < g.N) ((float*)g.C)[(long)m * g.ldc + n] = alpha * acc[i][j][e];
            }
        }
}

// gemm4: the same all-DMA pipeline with a 256 x 128 x 32 tile, 256 threads (2 x 2 waves of 128 x 64) and a ring of
// THREE stages (72 KiB), so that TWO workgroups live on a CU.  The read-modify-write epilogue of C -= V Y^T is a
// memory phase (HBM round trips, MFMA idle) that takes about as long as the K loop at K = 1024; with one workgroup
// per CU the two cannot overlap, with two independent workgroups one tile's epilogue runs under the other's K loop.
template <int EM>
__global__ __launch_bounds__(256, 2) void gemm4_f16_kernel(GemmArgs g, int tilesM, int tilesN) {
    using namespace g2;
    constexpr int BM = 256, BN = 128, BK = 32, NS = 3;
    constexpr int ROWB = 64, CPR = 4, RB = 4;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    auto swz = [](int r, int c) -> int { return r * ROWB + ((c ^ ((r / RB) & (CPR - 1))) << 4); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8;
    const int seq = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
    // 64 concurrent tiles per XCD (two per CU): groups of 4 x 16 tiles = 1024 rows x 2048 columns share an L2
    const int groupsN = (tilesN + 15) / 16;
    const int grp = seq / 64, within = seq % 64;
    const int tm = (grp / groupsN) * 4 + within / 16;
    const int tn = (grp % groupsN) * 16 + within % 16;
    if (tm >= tilesM || tn >= tilesN) return;
    const int bm = tm * BM, bn = tn * BN;
    const int ktiles = g.K / BK;
    const half_t* const A = (const half_t*)g.A;

    // one wave-instruction fills 1 KiB = 16 rows; A: 16 of them (4 per wave), B: 8 (2 per wave)
    auto issue = [&](int kt) {                              // kt past the end re-fetches the last tile into a free stage
        char* base = g2_smem + (kt % NS) * STAGE;
        const int k = min(kt, ktiles - 1) * BK;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int rb = (i * 4 + wave) * 16;
            const int rr = rb + (lane >> 2);
            const int c = (lane & 3) ^ ((rr >> 2) & 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (long)(bm + rr) * g.lda + k + c * 8),
                                             (__attribute__((address_space(3))) void*)(base + rb * ROWB), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int rb = (i * 4 + wave) * 16;
            const int rr = rb + (lane >> 2);
            const int c = (lane & 3) ^ ((rr >> 2) & 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.Bt + (long)(bn + rr) * g.ldb + k + c * 8),
                                             (__attribute__((address_space(3))) void*)(base + A_BYTES + rb * ROWB), 16, 0, 0);
        }
    };

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const int wm = (wave >> 1) * 128, wn = (wave & 1) * 64;

    issue(0); issue(1);
    for (int kt = 0; kt < ktiles; kt++) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // exactly two tiles (6 DMAs each) are in flight: kt has landed
        __builtin_amdgcn_s_barrier();                        // tile kt landed everywhere; stage (kt-1)%3 is free
        const char* As = g2_smem + (kt % NS) * STAGE;
        const char* Bs = As + A_BYTES;
        half8 a[2][4], b[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
#pragma unroll
            for (int i = 0; i < 4; i++) a[ks][i] = *(const half8*)(As + swz(wm + i * 32 + r, ks * 2 + h));
#pragma unroll
            for (int j = 0; j < 2; j++) b[ks][j] = *(const half8*)(Bs + swz(wn + j * 32 + r, ks * 2 + h));
        }
        issue(kt + 2);                                       // unconditional: keeps the loop body one basic block
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
        // issue order: the 12 fragment reads, then one LDS-DMA after every 2-3 MFMAs (6 DMAs, 16 MFMAs)
#pragma unroll
        for (int q2 = 0; q2 < 4; q2++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int q2 = 0; q2 < 2; q2++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the two trailing (redundant) tiles
    __builtin_amdgcn_s_barrier();
    const float alpha = g.alpha;
    if (EM == E_SUB_F32) {
        epilogue_sub_f32<4, 2>(acc, (float*)g.C, g.ldc, g.M, g.N, g.col_lo, alpha, bm + wm, bn + wn, r, h);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = bn + wn + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int m = bm + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < g.M && n < g.N) ((float*)g.C)[(long)m * g.ldc + n] = alpha * acc[i][j][e];
            }
        }
}

template <int EM>
static void launch4(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = 3 * (256 + 128) * 64;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm4_f16_kernel<EM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int tilesM = (g.M + 255) / 256, tilesN = (g.N + 127) / 128;
    const int groups = ((tilesM + 3) / 4) * ((tilesN + 15) / 16);
    hipLaunchKernelGGL((gemm4_f16_kernel<EM>), dim3(groups * 64), dim3(256), LDS, s, g, tilesM, tilesN);
}

template <int EM>
static void launch3(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = 4 * 2 * 256 * 64;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm3_f16_kernel<EM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int tilesM = (g.M + 255) / 256, tilesN = (g.N + 255) / 256;
    const int groups = ((tilesM + 3) / 4) * ((tilesN + 7) / 8);
    hipLaunchKernelGGL((gemm3_f16_kernel<EM>), dim3(groups * 32), dim3(512), LDS, s, g, tilesM, tilesN);
}

// Large-shape path.  Preconditions (checked by the caller, driver.hip): no split-K, K % 64 == 0, and the
// operand buffers are readable up to the next multiple of 256 rows (tiles are loaded unmasked; rows past M / N
// only feed outputs that the epilogue masks).  config: 0 = 256x256x64 / 512 threads, 1 = 256x128x32 / 256 threads.
bool launch_gemm2_f16(AMode am, EMode em, const GemmArgs& g, hipStream_t s, int config) {
    if (am == A_H16 && config != 2 && config != 1) {          // config 2 keeps the register-staged kernel (A/B comparison)
        static const int use4 = []() { const char* e = getenv("MPQR_GEMM4"); return e ? atoi(e) : 0; }();
        if (em == E_SUB_F32 && use4) { launch4<E_SUB_F32>(g, s); return true; }
        if (em == E_SUB_F32) { launch3<E_SUB_F32>(g, s); return true; }
        if (em == E_STORE_F32) { launch3<E_STORE_F32>(g, s); return true; }
    }
#define MPQR_CASE2(A_, E_)                                                  \
    if (am == A_ && em == E_) {                                             \
        if (config == 1) launch2<A_, E_, 2, 2, 32>(g, s);                   \
        else launch2<A_, E_, 2, 4, 64>(g, s);                               \
        return true;                                                        \
    }
    MPQR_CASE2(A_F32T, E_STORE_F32)
    MPQR_CASE2(A_F32, E_STORE_H16)
    MPQR_CASE2(A_H16, E_SUB_F32)
    MPQR_CASE2(A_F32, E_STORE_F32)
#undef MPQR_CASE2
    return false;
}

}  // namespace mpqr
