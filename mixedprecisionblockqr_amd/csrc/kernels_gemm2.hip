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
#include <algorithm>
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
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int A_ALL = (AM == A_F32S ? 2 : 1) * A_BYTES;      // A_F32S: hi tile, then lo tile
    constexpr int STAGE = A_ALL + B_BYTES;
    // byte offset of logical chunk c of row r: XOR swizzle makes 16 consecutive rows hit 16 distinct 16-B bank slots
    auto swz = [](int r, int c) -> int { return r * ROWB + ((c ^ ((r / RB) & (CPR - 1))) << 4); };
    // A tile of the transposing mode (A_F32T, BK = 64): a thread stores four 8-byte pieces into rows FOUR apart, so with the swizzle
    // above the 16 lanes of a ds_write_b64 group fall on 4 bank slots (4-way conflicts: 43 % of the kernel's LDS cycles,
    // profiles/r02_c4_sq_stalls.txt).  This one -- slot = c ^ (r2 | r3 << 1 | (r1 ^ r4) << 2), r_i = bit i of the row -- keeps the
    // ds_read_b128 fragment reads conflict-free (16 distinct (row parity, slot) pairs in each of the instruction's lane groups) AND
    // gives 8 distinct slots to any 8 consecutive rows-4-apart; lanes 2j, 2j+1 then share a slot with different halves (store_AT).
    auto swzA = [&](int r, int c) -> int {
        if (AM == A_F32T && BK == 64) return r * ROWB + ((c ^ (((r >> 2) & 3) | ((((r >> 1) ^ (r >> 4)) & 1) << 2))) << 4);
        return swz(r, c);
    };

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
    int kt0 = z * per, kt1 = min(ktiles_all, kt0 + per);
    if (g.tri == 1) kt0 = max(kt0, bn / BK);                           // Bt[n][k] = 0 for k < n
    else if (g.tri == 2) kt1 = min(kt1, (bn + BN + BK - 1) / BK);      // Bt[n][k] = 0 for k > n
    char* const As0 = g2_smem;
    char* const Bs0 = g2_smem + A_ALL;

    // ---- A staging through registers
    constexpr int NA = BM * CPR / NT;                  // 16-B chunks per thread (A_H16 / A_F32)
    constexpr int NBLK = (BK / 4) * (BM / 4) / NT;     // 4 x 4 fp32 blocks per thread (A_F32T)
    typedef float F4 __attribute__((ext_vector_type(4)));   // first-class vectors: the loads land in their final registers
    U4 ra[NA];
    U4 ral[AM == A_F32S ? NA : 1];                      // A_F32S: the lo parts
    F4 raT[2][NBLK * 4];                                 // A_F32T: two register sets, loads run two K tiles ahead
    // A_F32T: per-thread row pointers advance by BK rows per K tile (no 64-bit multiplies in the loop)
    const float* pT[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; i++) {
        const int id = tid + NT * i, mg = (id >> 1) % (BM / 4), kg = ((id >> 1) / (BM / 4)) * 2 + (id & 1);   // lanes 2j, 2j+1: same rows, k halves
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
                if (AM == A_F32S) {                           // lo = x - fp16(x), rounded to fp16 itself
                    auto lo2 = [&](float a, float b) { return pack2(a * sc - (float)(half_t)(a * sc), b * sc - (float)(half_t)(b * sc)); };
                    U4 l;
                    l.x = lo2(s0.x, s0.y); l.y = lo2(s0.z, s0.w); l.z = lo2(s1.x, s1.y); l.w = lo2(s1.z, s1.w);
                    ral[i] = l;
                }
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
            const int id = tid + NT * i, mg = (id >> 1) % (BM / 4), kg = ((id >> 1) / (BM / 4)) * 2 + (id & 1);
            const F4 v0 = raT[S][i * 4 + 0], v1 = raT[S][i * 4 + 1], v2 = raT[S][i * 4 + 2], v3 = raT[S][i * 4 + 3];
            const int half_off = (kg & 1) * 8, chunk = kg >> 1;
            uint2 w;
            w.x = pack2(v0.x * sc, v1.x * sc); w.y = pack2(v2.x * sc, v3.x * sc); *(uint2*)(As + swzA(mg * 4 + 0, chunk) + half_off) = w;
            w.x = pack2(v0.y * sc, v1.y * sc); w.y = pack2(v2.y * sc, v3.y * sc); *(uint2*)(As + swzA(mg * 4 + 1, chunk) + half_off) = w;
            w.x = pack2(v0.z * sc, v1.z * sc); w.y = pack2(v2.z * sc, v3.z * sc); *(uint2*)(As + swzA(mg * 4 + 2, chunk) + half_off) = w;
            w.x = pack2(v0.w * sc, v1.w * sc); w.y = pack2(v2.w * sc, v3.w * sc); *(uint2*)(As + swzA(mg * 4 + 3, chunk) + half_off) = w;
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
                if (AM == A_F32S) *(U4*)(As + A_BYTES + swz(row, kc)) = ral[i];
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
            for (int i = 0; i < 4; i++) a[i] = *(const half8*)(As + swzA(wm + i * 32 + r, ks * 2 + h));
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = *(const half8*)(Bs + swz(wn + j * 32 + r, ks * 2 + h));
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            if (AM == A_F32S) {
                half8 al[4];
#pragma unroll
                for (int i = 0; i < 4; i++) al[i] = *(const half8*)(As + A_BYTES + swz(wm + i * 32 + r, ks * 2 + h));
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], b[j], acc[i][j], 0, 0, 0);
            }
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
        epilogue_sub_f32<4, 2>(acc, (float*)g.C, g.ldc, g.M, g.N, g.col_lo, alpha, bm + wm, bn + wn, r, h, g.Ct, g.ldct, g.ct_scale);
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
                    else {
                        const float vs = g.cscale ? v * g.cscale[(long)n * g.cscale_ld] : v;
                        const half_t hv = (half_t)vs;
                        ((half_t*)g.C)[(long)m * g.ldc + n] = hv;
                        if (g.C2) g.C2[(long)m * g.ldc + n] = (half_t)(vs - (float)hv);
                    }
                }
            }
        }
}

template <int AM, int EM, int WM, int WN, int BK>
static void launch2(const GemmArgs& g, hipStream_t s) {
    constexpr int BM = 128 * WM, BN = 64 * WN, NT = 64 * WM * WN;
    constexpr int LDS = 2 * ((AM == A_F32S ? 2 : 1) * BM + BN) * BK * 2;
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)gemm2_f16_kernel<AM, EM, WM, WN, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)));
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    const int groups = ((tilesM + 3) / 4) * ((tilesN + 7) / 8);
    GemmArgs a = g;
    if (a.nslab_in < 1) a.nslab_in = 1;
    if (a.nsplit < 1 || EM != E_STORE_F32) a.nsplit = 1;
    hipLaunchKernelGGL((gemm2_f16_kernel<AM, EM, WM, WN, BK>), dim3(groups * 32, a.nsplit), dim3(NT), LDS, s, a, tilesM, tilesN);
}

// ------------------------------------------------------------------ read-modify-write epilogue through LDS-DMA
// C[256 x 256 tile] -= alpha * acc for the 2 x 4 wave layout (wave tile 128 x 64 = 4 x 2 MFMA 32x32 tiles).
// At K = 1024 the register-staged epilogue (gemm_epilogue.h) costs as much as the K loop: a lane has at most 16-32
// four-byte loads of old C in flight, ~32 KiB per CU, so the tile's 256 KiB of old values arrive latency-bound.
// Here the old values come in by LDS-DMA (global_load_lds_dwordx4: no VGPR destination, nothing for a wave to wait
// on): the tile is cut into 8 chunks of 32 rows (16 from each wave row, so every wave has work in every chunk),
// 4 chunks = 128 KiB are in flight at once in the LDS the K loop has just released, and each wave then reads its
// 16 old values per chunk with conflict-free ds_read_b32 (lanes run along n), subtracts and stores.
// Rows >= M are clamped on the load side (never stored); columns past N read the row's continuation (inside the
// allocation: every matrix has 1024 floats of slack) and are never stored.
// Only for tiles that lie completely inside [0,M) x [col_lo,N): every lane stores, so the number of outstanding
// vector-memory operations at each wait is known at compile time.  Edge tiles take the register-staged epilogue.
// SH: also write the transposed fp16 shadow (GemmArgs::Ct).  The new values replace the accumulators; after the last chunk
// every wave transposes its 128 x 64 sub-tile through its own 16 KiB of the (now free) LDS ring -- 8-byte writes at
// [column][row], XOR-swizzled 16-byte chunks -- and stores whole 256-byte column segments (16 lanes x 16 bytes; the
// direct 8-byte stores of a first version cost the kernel 55 %).
template <bool SH, int NT, typename ACC>
__device__ __forceinline__ void epilogue_sub_f32_dma(ACC (&acc)[4][2], char* smem, float* __restrict__ C, long ldc,
                                                     float alpha, int bm, int bn, int wave, int lane,
                                                     half_t* __restrict__ Ct = nullptr, long ldct = 0, float cts = 1.f) {
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 2, wn = (wave & 3) * 64;
    // chunk c: tile rows 16 c .. 16 c + 15 of BOTH wave rows -> 32 rows x 1 KiB; wave w moves rows 4 w .. 4 w + 3 of it
    const float* src0 = C + (long)(bm + (wave >> 2) * 128 + (wave & 3) * 4) * ldc + bn + 4 * lane;
    auto issue = [&](int c) {
        char* base = smem + (c & 3) * 32768 + wave * 4096;
        const float* src = src0 + (long)(16 * c) * ldc;
#pragma unroll
        for (int q = 0; q < 4; q++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)q * ldc),
                                             (__attribute__((address_space(3))) void*)(base + q * 1024), 16, 0, NT ? 2 : 0);
    };
    issue(0); issue(1); issue(2); issue(3);
    float* pbase = C + (long)(bm + wr * 128 + 4 * h) * ldc + bn + wn + r;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        // Vector-memory operations retire in issue order.  Issue order of this wave:
        //   [0][1][2][3] st0 [4] st1 [5] st2 [6] st3 [7] st4 st5 st6 st7      ([k] = 4 DMAs of chunk k, st = 16 stores)
        // chunk c has landed when only what was issued after [c] is still outstanding:
        if (c == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");          // [1][2][3]
        else if (c == 1) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");     // [2][3] st0
        else if (c == 2) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");     // [3] st0 [4] st1
        else if (c <= 5) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");     // st [c+1] st [c+2] st
        else if (c == 6) asm volatile("s_waitcnt vmcnt(52)" ::: "memory");     // st3 [7] st4 st5
        else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");                 // st4 st5 st6
        __builtin_amdgcn_s_barrier();                  // chunk c landed for every wave; everybody is done reading chunk c-1
        if (c >= 1 && c + 3 < 8) issue(c + 3);         // into the buffer chunk c-1 has just released
        const int i = c >> 1, hf = c & 1;
        const char* base = smem + (c & 3) * 32768 + (16 * wr + 4 * h) * 1024 + (wn + r) * 4;
        float oldv[2][8];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int ee = 0; ee < 8; ee++)
                oldv[j][ee] = *(const float*)(base + ((ee & 3) + 8 * (ee >> 2)) * 1024 + j * 128);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            float* p = pbase + (long)(16 * c) * ldc + j * 32;
#pragma unroll
            for (int q = 0; q < 2; q++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v = oldv[j][q * 4 + e] - alpha * acc[i][j][8 * hf + q * 4 + e];
                    if (NT) __builtin_nontemporal_store(v, &p[(long)e * ldc]); else p[(long)e * ldc] = v;
                    if (SH) acc[i][j][8 * hf + q * 4 + e] = v;
                }
                p += 8 * ldc;
            }
        }
    }
    if (SH) {
        typedef half_t half4e __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_s_barrier();                  // every wave has read the last chunk: the ring is free
        char* tb = smem + wave * 16384;                // [64 columns][128 rows] fp16, 256 B per column
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    half4e hv;
#pragma unroll
                    for (int e = 0; e < 4; e++) hv[e] = (half_t)(cts * acc[i][j][4 * g4 + e]);
                    const int col = j * 32 + r, chunk = 4 * i + g4;          // rows 32 i + 8 g4 + 4 h + e
                    *(half4e*)(tb + col * 256 + ((chunk ^ (col & 15)) << 4) + 8 * h) = hv;
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        half_t* ct = Ct + (long)(bn + wn) * ldct + bm + wr * 128;
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const int col = it * 4 + (lane >> 4), ch = lane & 15;
            const uint4 v = *(const uint4*)(tb + col * 256 + ((ch ^ (col & 15)) << 4));
            if (NT) {
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
                u4v vv; vv[0] = v.x; vv[1] = v.y; vv[2] = v.z; vv[3] = v.w;
                __builtin_nontemporal_store(vv, (u4v*)(ct + (long)col * ldct + 8 * ch));
            } else *(uint4*)(ct + (long)col * ldct + 8 * ch) = v;
        }
    }
}

// ------------------------------------------------------------------ gemm6: ping-pong wave groups, 256 x 256 x 64
// Both operands fp16, k contiguous, staged by LDS-DMA; E_SUB_F32 / E_STORE_F32 / E_STORE_H16.  (Round 1's 4-stage ring kernel
// synchronised all eight waves once per 32-deep K step, and every wave then read its fragments before anybody could issue an
// MFMA: the matrix pipes idled while the LDS bursts, 44 % of the MFMA rate inside the K loop.  It is gone; git history.)  Here:
//   * K tile 64, two LDS buffers of 64 KiB (A 256 x 128 B, B 256 x 128 B, XOR-swizzled 16-B chunks as in gemm2);
//   * a wave's 128 x 64 output is cut into four 64 x 32 quadrants; one PHASE computes one quadrant over the K tile
//     (8 MFMA 32x32x16) and needs only the fragments of one 64-row A sub-tile / one 32-column B sub-tile:
//         p0: read A0 -> q00     p1: read B1 -> q01     p2: read A1 -> q11     p3: read B0 of the NEXT tile -> q10
//     (8 / 4 / 8 / 4 ds_read_b128 per phase);
//   * the two wave rows (waves 0-3 / 4-7; one wave of each lives on every SIMD) run HALF A PHASE APART: while one
//     group issues its 8 MFMAs (and, in their shadow, two LDS-DMAs) the other reads fragments, then they swap (two
//     s_barriers per phase, the second group enters the loop one barrier late);
//   * staging unit = "half tile" = the sub-tile rows of BOTH wave rows / all four wave columns (128 rows x 128 B =
//     16 KiB = 2 LDS-DMA per wave), one per phase, in the order A-S0, B-S0, B-S1, A-S1.  An LDS region is dead as
//     soon as its fragments are in registers, so half tile e = 4 tile + s is issued in phase e - 7: three to four half
//     tiles (48-64 KiB) are always in flight behind counted s_waitcnt vmcnt(8 / 6); the region a DMA overwrites was
//     last read at least one full phase earlier by either group, and the data a phase reads was waited for by every
//     wave before the barrier that precedes it.
// MF = 1 (the store epilogues' default, round 4): the same pipeline on v_mfma_f32_16x16x32_f16 -- a quadrant is 4 x 2 tiles of 16 x 16
// over two K steps of 32 (16 MFMAs of 16 cycles instead of 8 of 32); fragment registers and LDS reads are the same in number, lane
// l reads row l & 15, 16-byte chunk 4 ks + (l >> 4).  On random data the chip holds a higher clock on this shape
// (MI355X_MICROARCH.md, DVFS give-back (7); tools/ubench_mfma.hip).  The operands are passed SWAPPED, so a lane holds four
// consecutive columns of a row of C: 16-byte (fp32) / 8-byte (fp16) stores.
template <int EM, int DMA_EPI, int MF = 0, int DR = 0>
__global__ __launch_bounds__(512) void gemm6_f16_kernel(GemmArgs g, int tilesM, int tilesN) {
    using namespace g2;
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int ROWB = 128, A_BYTES = BM * ROWB, BUF = 2 * A_BYTES;
    auto swz = [](int r, int c) -> int { return r * ROWB + ((c ^ ((r >> 1) & 7)) << 4); };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    // triangular second operand (GemmArgs::tri): only the K tiles where Bt can be non-zero for this tile's columns
    const int kt_lo = g.tri == 1 ? bn / BK : 0;
    const int kt_hi = g.tri == 2 ? min(g.K / BK, (bn + BN + BK - 1) / BK) : g.K / BK;
    const int kth = kt_hi - kt_lo;                            // K tiles of one pass over Bt
    const int ktiles = g.A2 ? 2 * kth : kth;                  // second pass: the lo parts of A over the same Bt
    const half_t* const A = (const half_t*)g.A + (long)kt_lo * BK;
    const half_t* const A2 = g.A2 ? g.A2 + (long)kt_lo * BK : nullptr;
    const half_t* const Bt = g.Bt + (long)kt_lo * BK;
    const int wr = wave >> 2, wc = wave & 3;
    const int wm = wr * 128, wn = wc * 64;

    // half tile s of K tile tau -> buffer tau & 1.  s: 0 = A-S0, 1 = B-S0, 2 = B-S1, 3 = A-S1.
    // One wave-instruction = 8 rows x 128 B; lane l lands on row row0 + l/8, physical chunk l%8.
    auto issue = [&](int tau, int sidx) {
        const int tc = min(tau, ktiles - 1);                  // past the end: re-fetch the last tile into a dead region
        const bool second = tc >= kth;
        const int k = (second ? tc - kth : tc) * BK;
        const half_t* const Ab = second ? A2 : A;
        char* buf = g2_smem + (tau & 1) * BUF;
        const bool isA = (sidx == 0 || sidx == 3);
        const int sub = (sidx == 0 || sidx == 1) ? 0 : 1;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int gidx = i * 8 + wave;                    // 16 row groups of 8 per half tile
            int row0;
            if (isA) row0 = (gidx >> 3) * 128 + sub * 64 + 8 * (gidx & 7);
            else row0 = 64 * (gidx >> 2) + 32 * sub + 8 * (gidx & 3);
            const int rr = row0 + (lane >> 3);
            const int c = (lane & 7) ^ ((rr >> 1) & 7);
            if (isA)        // (non-temporal loads for the shadow operand of the far X = A2^T V: no effect, round 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ab + (long)(bm + rr) * g.lda + k + c * 8),
                                                 (__attribute__((address_space(3))) void*)(buf + row0 * ROWB), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bt + (long)(bn + rr) * g.ldb + k + c * 8),
                                                 (__attribute__((address_space(3))) void*)(buf + A_BYTES + row0 * ROWB), 16, 0, 0);
        }
    };

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    floatx4 acc16[MF ? 8 : 1][MF ? 4 : 1];                    // MF = 1: [16-row tile][16-column tile] of the wave's 128 x 64
    if (MF) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc16[MF ? i : 0][MF ? j : 0][e] = 0.f;
    }
    const int li = lane & 15, lk = lane >> 4;
    half8 af[4][2], b0a[4], b0b[4], b1[4];                    // A sub-tile fragments [k step][row tile]; B-S0 (two tiles), B-S1
    // (MF = 1: af[2 ks2 + (i4 >> 1)][i4 & 1] = row tile i4 of 16, K step ks2 of 32;  b[2 ks2 + j2] = column tile j2 of 16)
    auto read_A = [&](const char* As, int sub) {
        if (MF) {
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
                for (int i4 = 0; i4 < 4; i4++) af[2 * ks2 + (i4 >> 1)][i4 & 1] = *(const half8*)(As + swz(wm + sub * 64 + i4 * 16 + li, ks2 * 4 + lk));
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int i2 = 0; i2 < 2; i2++) af[ks][i2] = *(const half8*)(As + swz(wm + sub * 64 + i2 * 32 + r, ks * 2 + h));
    };
    auto read_B = [&](const char* Bs, int sub, half8 (&b)[4]) {
        if (MF) {
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
                for (int j2 = 0; j2 < 2; j2++) b[2 * ks2 + j2] = *(const half8*)(Bs + swz(wn + sub * 32 + j2 * 16 + li, ks2 * 4 + lk));
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) b[ks] = *(const half8*)(Bs + swz(wn + sub * 32 + r, ks * 2 + h));
    };
    // MFMA half phase: 8 MFMAs with the two LDS-DMAs of one half tile issued in their shadow (after the 3rd and 6th)
    auto mma = [&](int subA, int subB, const half8 (&b)[4], int tau, int sidx) {
        __builtin_amdgcn_s_setprio(1);
        if (MF) {
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ks2++)
#pragma unroll
                for (int i4 = 0; i4 < 4; i4++)
#pragma unroll
                    for (int j2 = 0; j2 < 2; j2++)
                        // operands SWAPPED (D^T = B-fragment x A-fragment): lane (li, lk) then holds C[m = 16 i + li][n = 16 j + 4 lk .. + 3],
                        // four CONSECUTIVE COLUMNS of a row-major C -> 16-byte loads and stores in the epilogue
                        acc16[MF ? subA * 4 + i4 : 0][MF ? subB * 2 + j2 : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                            b[2 * ks2 + j2], af[2 * ks2 + (i4 >> 1)][i4 & 1], acc16[MF ? subA * 4 + i4 : 0][MF ? subB * 2 + j2 : 0], 0, 0, 0);
            if (!DR) {
                issue(tau, sidx);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int i2 = 0; i2 < 2; i2++)
                acc[subA * 2 + i2][subB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][i2], b[ks], acc[subA * 2 + i2][subB], 0, 0, 0);
        if (!DR) {
            issue(tau, sidx);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // One K tile = four phases.  Phase f = 4 t + p reads (fragment half phase) and then computes (MFMA half phase):
    //   p0: A-S0(t)            -> q00 with B-S0(t) (read one phase ahead, in p3 of the previous tile)
    //   p1: B-S1(t)            -> q01
    //   p2: A-S1(t)            -> q11
    //   p3: B-S0(t+1) [next]   -> q10 with B-S0(t)
    // and issues half tile e = f + 7 inside its MFMA half phase.  Before the barrier that ends a fragment half phase
    // every wave waits until the half tile the NEXT phase reads has landed (retirement is in issue order):
    //   end of p0: needs e = 4t+2, issued <= 4t+6 -> 4 half tiles may be outstanding -> vmcnt(8); p1: the same;
    //   end of p2: needs 4t+5, issued <= 4t+8 -> vmcnt(6);  end of p3: 4t+4 landed with the previous wait.
    auto tile = [&](int t, half8 (&b0)[4], half8 (&b0n)[4]) {
        const char* As = g2_smem + (t & 1) * BUF;
        const char* Bs = As + A_BYTES;
        const char* Bn = g2_smem + ((t + 1) & 1) * BUF + A_BYTES;
        // DR: the half tile is issued by the wave in its FRAGMENT half phase (e = f + 6: what the MFMA half phase of phase f - 1 issued in
        // the other form, a quarter phase later): an LDS-DMA takes ~60 cycles to issue and the wave is in order, so between a wave's own
        // MFMAs it opens a bubble in the matrix pipe; the reading wave has the slack.  Same number of half tiles outstanding at every wait.
        read_A(As, 0);
        if (DR) issue(t + 1, 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mma(0, 0, b0, t + 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        read_B(Bs, 1, b1);
        if (DR) issue(t + 1, 3);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mma(0, 1, b1, t + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        read_A(As, 1);
        if (DR) issue(t + 2, 0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mma(1, 1, b1, t + 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        read_B(Bn, 0, b0n);
        if (DR) issue(t + 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mma(1, 0, b0, t + 2, 2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

#ifdef MPQR_KTRACE
    const bool kt_on = (tid == 0) && (bid % 397 == 5) && EM == E_SUB_F32 && gridDim.x > 2000;
    long kt0 = 0, kt1 = 0, kt2 = 0; unsigned long long rt0 = 0;
    if (kt_on) { kt0 = clock64(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    // prologue: half tiles 0..6 (tile 0 complete, A-S0 / B-S0 / B-S1 of tile 1); A-S0(0) and B-S0(0) landed
    issue(0, 0); issue(0, 1); issue(0, 2); issue(0, 3); issue(1, 0); issue(1, 1);
    if (!DR) { issue(1, 2); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_B(g2_smem + A_BYTES, 0, b0a);
    if (wr == 1) __builtin_amdgcn_s_barrier();               // second group: half a phase behind
    for (int t = 0; t < ktiles; t += 2) {
        tile(t, b0a, b0b);
        if (t + 1 < ktiles) tile(t + 1, b0b, b0a);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();               // first group waits for the second one's last half phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the trailing (redundant) half tiles
    __builtin_amdgcn_s_barrier();
#ifdef MPQR_KTRACE
    if (kt_on) kt1 = clock64();
#endif
    const float alpha = g.alpha;
    if (MF) {                                               // 16 x 16 accumulator tiles, transposed: row m = lane & 15, columns 4 (lane >> 4) + e
        typedef float F4 __attribute__((ext_vector_type(4)));
        typedef half_t H4 __attribute__((ext_vector_type(4)));
        const bool inner = bm + BM <= g.M && bn + BN <= g.N && (g.ldc & 3) == 0;     // interior tile: 16- / 8-byte stores
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int m = bm + wm + i * 16 + li, n0 = bn + wn + j * 16 + 4 * lk;
                const long o = (long)m * g.ldc + n0;
                F4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] = alpha * acc16[MF ? i : 0][MF ? j : 0][e];
                    if (EM == E_STORE_H16 && g.cscale) v[e] *= g.cscale[(long)min(n0 + e, g.N - 1) * g.cscale_ld];      // tau_n (unit-diagonal fp16 T)
                    if (EM == E_STORE_F32 && g.eye_minus) v[e] = (m == n0 + e ? 1.f : 0.f) - v[e];
                }
                if (EM == E_STORE_H16) {
                    H4 hv, lv;
#pragma unroll
                    for (int e = 0; e < 4; e++) { hv[e] = (half_t)v[e]; lv[e] = (half_t)(v[e] - (float)hv[e]); }
                    if (inner) { *(H4*)((half_t*)g.C + o) = hv; if (g.C2) *(H4*)(g.C2 + o) = lv; }
                    else if (m < g.M) {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (n0 + e < g.N) { ((half_t*)g.C)[o + e] = hv[e]; if (g.C2) g.C2[o + e] = lv[e]; }
                    }
                } else if (EM == E_STORE_F32) {
                    if (inner && g.nt_c) __builtin_nontemporal_store(v, (F4*)((float*)g.C + o));      // (a result nobody on the GPU reads again soon)
                    else if (inner) *(F4*)((float*)g.C + o) = v;
                    else if (m < g.M) {
#pragma unroll
                        for (int e = 0; e < 4; e++) if (n0 + e < g.N) ((float*)g.C)[o + e] = v[e];
                    }
                } else if (m < g.M) {                          // E_SUB_F32: test / bench path only (the library's updates use the 32 x 32 DMA epilogue)
#pragma unroll
                    for (int e = 0; e < 4; e++) if (n0 + e < g.N && n0 + e >= g.col_lo) ((float*)g.C)[o + e] -= v[e];
                }
            }
        return;
    }
    if (EM == E_SUB_F32) {
        const bool full = DMA_EPI && bm + BM <= g.M && bn + BN <= g.N && bn >= g.col_lo;      // uniform over the workgroup
        if (full) {
            if (g.nt_c) {
                if (g.Ct) epilogue_sub_f32_dma<true, 1>(acc, g2_smem, (float*)g.C, g.ldc, alpha, bm, bn, wave, lane, g.Ct, g.ldct, g.ct_scale);
                else epilogue_sub_f32_dma<false, 1>(acc, g2_smem, (float*)g.C, g.ldc, alpha, bm, bn, wave, lane);
            } else {
                if (g.Ct) epilogue_sub_f32_dma<true, 0>(acc, g2_smem, (float*)g.C, g.ldc, alpha, bm, bn, wave, lane, g.Ct, g.ldct, g.ct_scale);
                else epilogue_sub_f32_dma<false, 0>(acc, g2_smem, (float*)g.C, g.ldc, alpha, bm, bn, wave, lane);
            }
        }
        else epilogue_sub_f32<4, 2>(acc, (float*)g.C, g.ldc, g.M, g.N, g.col_lo, alpha, bm + wm, bn + wn, r, h, g.Ct, g.ldct, g.ct_scale);
#ifdef MPQR_KTRACE
        if (kt_on) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            kt2 = clock64();
            const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
            printf("ktrace gemm6 bid %d K %d full %d: loop %ld epi %ld cycles, total %.2f us\n", bid, g.K, (int)full, kt1 - kt0, kt2 - kt1,
                   (double)(rt1 - rt0) / 100.0);
        }
#endif
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
                    if (EM == E_STORE_H16) {
                        const float vs = g.cscale ? alpha * acc[i][j][e] * g.cscale[(long)n * g.cscale_ld] : alpha * acc[i][j][e];
                        const half_t hv = (half_t)vs;
                        ((half_t*)g.C)[(long)m * g.ldc + n] = hv;
                        if (g.C2) g.C2[(long)m * g.ldc + n] = (half_t)(vs - (float)hv);
                    }
                    else if (g.eye_minus) ((float*)g.C)[(long)m * g.ldc + n] = (m == n ? 1.f : 0.f) - alpha * acc[i][j][e];
                    else ((float*)g.C)[(long)m * g.ldc + n] = alpha * acc[i][j][e];
                }
            }
        }
}

template <int EM, int DMA_EPI, int MF = 0, int DR = 0>
static void launch6(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = 2 * 2 * 256 * 128;
    // Where a wave issues its LDS-DMAs (kernel comment, "DR"): in the fragment half phase for the 16x16x32 store forms (kernel alone, K = 8192
    // store 1277 -> 1313, X = Q2^T V 1079 -> 1142 TFLOP/s), between its MFMAs for the 32x32x16 read-modify-write form (825 vs 787 the
    // other way round; tools/bench_gemm.py, same box).  MPQR_G6_DR=0 / 1 forces one placement for every form (A/B hook).
    if (!DR && MF) {
        static const int dr = []() { const char* e = getenv("MPQR_G6_DR"); return e ? atoi(e) : 1; }();
        if (dr) { launch6<EM, DMA_EPI, MF, 1>(g, s); return; }
    } else if (!DR) {
        static const int dr = []() { const char* e = getenv("MPQR_G6_DR"); return e ? atoi(e) : 0; }();
        if (dr) { launch6<EM, DMA_EPI, MF, 1>(g, s); return; }
    }
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)gemm6_f16_kernel<EM, DMA_EPI, MF, DR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)));
    const int tilesM = (g.M + 255) / 256, tilesN = (g.N + 255) / 256;
    const int groups = ((tilesM + 3) / 4) * ((tilesN + 7) / 8);
    hipLaunchKernelGGL((gemm6_f16_kernel<EM, DMA_EPI, MF, DR>), dim3(groups * 32), dim3(512), LDS, s, g, tilesM, tilesN);
}

// Large-shape path.  Preconditions (checked by the caller, driver.hip): no split-K, K % 64 == 0, and the
// operand buffers are readable up to the next multiple of 256 rows (tiles are loaded unmasked; rows past M / N
// only feed outputs that the epilogue masks).  config: 0 = the library's choice, 2 = the register-staged kernel for fp16 A as well,
// 16 / 32 = the ping-pong kernel on 16x16x32 / 32x32x16 for every epilogue (test and bench entries).
bool launch_gemm2_f16(AMode am, EMode em, const GemmArgs& g, hipStream_t s, int config) {
    if (am == A_H16 && config == 16 && (g.K % 64) == 0 && !g.Ct) {   // mpqr_gemm_test_f32 / mpqr_bench_gemm kernel 16: the 16x16x32 form of every epilogue
        if (em == E_SUB_F32) { launch6<E_SUB_F32, 0, 1>(g, s); return true; }
        if (em == E_STORE_F32) { launch6<E_STORE_F32, 0, 1>(g, s); return true; }
        if (em == E_STORE_H16) { launch6<E_STORE_H16, 0, 1>(g, s); return true; }
        return false;
    }
    if (am == A_H16 && config != 2 && config != 1) {          // config 2 keeps the register-staged kernel (A/B comparison)
        if ((g.K % 64) == 0) {
            if (em == E_SUB_F32) { launch6<E_SUB_F32, 1>(g, s); return true; }
            // store epilogues (X = Q2^T V, W = V T, Q = I - W V^T, Y = X T): v_mfma_f32_16x16x32_f16 -- on random operands the chip holds
            // 1.87 GHz on this shape against 1.60 on 32x32x16 (tools/ubench_mfma.hip: 1900 vs 1550 TFLOP/s from registers); in this
            // kernel +5 % (tools/bench_gemm.py).  MPQR_MFMA16=0: the 32x32x16 form (A/B hook, run by the opt-in test)
            static const int mf16 = []() { const char* e = getenv("MPQR_MFMA16"); return e ? atoi(e) : 1; }();
            if (em == E_STORE_F32) { if (mf16 && config != 32) launch6<E_STORE_F32, 0, 1>(g, s); else launch6<E_STORE_F32, 0>(g, s); return true; }
            if (em == E_STORE_H16) { if (mf16 && config != 32) launch6<E_STORE_H16, 0, 1>(g, s); else launch6<E_STORE_H16, 0>(g, s); return true; }
        }
    }
#define MPQR_CASE2(A_, E_)                                                  \
    if (am == A_ && em == E_) {                                             \
        launch2<A_, E_, 2, 4, 64>(g, s);                                    \
        return true;                                                        \
    }
    MPQR_CASE2(A_F32T, E_STORE_F32)
    MPQR_CASE2(A_F32T, E_STORE_H16)
    MPQR_CASE2(A_F32, E_STORE_H16)
    MPQR_CASE2(A_H16, E_SUB_F32)
    MPQR_CASE2(A_F32, E_STORE_F32)
#undef MPQR_CASE2
    if (am == A_F32S && em == E_STORE_H16) { launch2<A_F32S, E_STORE_H16, 2, 4, 32>(g, s); return true; }
    return false;
}

}  // namespace mpqr
