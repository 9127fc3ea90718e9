// kernels_fp8.hip -- fp8 (OCP e4m3) operand path of the far trailing update (BASELINE config 5).
//
// The reference's dtype-templated WMMA GEMM (Cuda/mmult.cuh:252-300) has half / int8 instantiations; on CDNA4 the
// low-precision matrix path is the block-scaled MFMA  v_mfma_scale_f32_32x32x64_f8f6f4  (fp8 operands, fp32
// accumulate, twice the fp16 MFMA rate).  Here it carries the three GEMMs' two large ones,
//     X = (s A2)^T V      (tn, K = rows)        and        A2 -= (1/s) V Y^T      (nn, K = reflectors of the block),
// with operands quantised by plain power-of-two scales:  V * 2^8  (unit-norm reflectors: |v| <= 1),  s A2  (the
// factorisation's own scale s puts column norms in [2^7, 2^8), so |s a_ij| <= 2^8 < 448 = max e4m3),  Y * 2^-2.
// The instruction's per-32-element e8m0 scales are all 2^0 (0x7F); the powers of two leave through alpha.
// Precision: e4m3 keeps 4 significant bits (relative rounding error 2^-4 per operand element); the measured backward
// error of the factorisation is reported by bench.py / the tests, not assumed.
//
// gemm8 = gemm6's ping-pong pipeline (kernels_gemm2.hip) on 1-byte elements: K tile 128 (the same 128-B LDS rows, the
// same XOR swizzle and LDS-DMA geometry), a lane's MFMA fragment = 32 consecutive bytes of its row (two ds_read_b128;
// probed: the operand byte -> k map is irrelevant as long as A and B use the same one, tools/probe_mfma_fp8.hip).
#include <algorithm>
#include "mpqr_internal.h"
#include "gemm_epilogue.h"

namespace mpqr {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int int8v __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));

extern __shared__ __attribute__((aligned(16))) char g8_smem[];

__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    const float lim = 448.f;                                   // e4m3 max finite
    a = fminf(fmaxf(a, -lim), lim); b = fminf(fmaxf(b, -lim), lim);
    c = fminf(fmaxf(c, -lim), lim); d = fminf(fmaxf(d, -lim), lim);
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (uint32_t)v;
}

// dst[r][c] = fp8(scale * src[r][c]),  src fp16 [rows][lds], dst bytes [rows][ldd]; cols % 8 == 0
__global__ __launch_bounds__(256) void quant_h16_fp8_kernel(const half_t* __restrict__ src, long lds_, uint8_t* __restrict__ dst,
                                                            long ldd, int rows, int cols, float scale) {
    typedef half_t half8 __attribute__((ext_vector_type(8)));
    const long per_row = cols / 8;
    const long tot = (long)rows * per_row;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
        const long r = e / per_row, c8 = e % per_row;
        const half8 v = *(const half8*)(src + r * lds_ + c8 * 8);
        uint2 o;
        o.x = pack4_fp8(scale * (float)v[0], scale * (float)v[1], scale * (float)v[2], scale * (float)v[3]);
        o.y = pack4_fp8(scale * (float)v[4], scale * (float)v[5], scale * (float)v[6], scale * (float)v[7]);
        *(uint2*)(dst + r * ldd + c8 * 8) = o;
    }
}
void launch_quant_h16_fp8(const half_t* src, long lds_, uint8_t* dst, long ldd, int rows, int cols, float scale, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return;
    const long tot = (long)rows * (cols / 8);
    const int grid = (int)std::min<long>((tot + 255) / 256, 4096);
    hipLaunchKernelGGL(quant_h16_fp8_kernel, dim3(grid), dim3(256), 0, s, src, lds_, dst, ldd, rows, cols, scale);
}

// dst[c][r] = fp8(scale * src[r][c]):  src fp32 [rows][lds] -> dst bytes [cols][ldd] (transposed), 64 x 64 tiles.
// rows % 64 == 0 (row ranges start at 64-aligned rows of a 256-padded matrix); columns >= cols are written as zero.
__global__ __launch_bounds__(256) void quant_transpose_f32_fp8_kernel(const float* __restrict__ src, long lds_, uint8_t* __restrict__ dst,
                                                                      long ldd, int rows, int cols, float scale) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 16 float4 per row, 16 rows per pass
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int r = ty + 16 * p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = c0 + 4 * tx;
        const float* q = src + (long)(r0 + r) * lds_ + c;
        if (c + 3 < cols) v = *(const float4*)q;
        else { if (c < cols) v.x = q[0]; if (c + 1 < cols) v.y = q[1]; if (c + 2 < cols) v.z = q[2]; }
        tile[r][4 * tx] = v.x; tile[r][4 * tx + 1] = v.y; tile[r][4 * tx + 2] = v.z; tile[r][4 * tx + 3] = v.w;
    }
    __syncthreads();
    // each thread writes 16 consecutive rows (bytes) of one column: 4 threads per column
    const int c = threadIdx.x >> 2, rq = (threadIdx.x & 3) * 16;
    uint4 o;
    o.x = pack4_fp8(scale * tile[rq + 0][c], scale * tile[rq + 1][c], scale * tile[rq + 2][c], scale * tile[rq + 3][c]);
    o.y = pack4_fp8(scale * tile[rq + 4][c], scale * tile[rq + 5][c], scale * tile[rq + 6][c], scale * tile[rq + 7][c]);
    o.z = pack4_fp8(scale * tile[rq + 8][c], scale * tile[rq + 9][c], scale * tile[rq + 10][c], scale * tile[rq + 11][c]);
    o.w = pack4_fp8(scale * tile[rq + 12][c], scale * tile[rq + 13][c], scale * tile[rq + 14][c], scale * tile[rq + 15][c]);
    *(uint4*)(dst + (long)(c0 + c) * ldd + r0 + rq) = o;
}
void launch_quant_transpose_f32_fp8(const float* src, long lds_, uint8_t* dst, long ldd, int rows, int cols, float scale, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(quant_transpose_f32_fp8_kernel, dim3((cols + 63) / 64, rows / 64), dim3(256), 0, s, src, lds_, dst, ldd, rows, cols, scale);
}

// C[M x N] (-)= alpha * A[M][K] * Bt[N][K]^T, A and Bt fp8 e4m3 (1 byte per element, k contiguous), fp32 accumulate.
// E_STORE_F32: C(slab z) = alpha * acc, K split over gridDim.y slices; E_SUB_F32: C -= alpha * acc (columns >= col_lo).
// K % 128 == 0; operands readable up to the next multiple of 256 rows.
template <int EM, int DMA_EPI>
__global__ __launch_bounds__(512) void gemm8_fp8_kernel(GemmArgs g, int tilesM, int tilesN) {
    constexpr int BM = 256, BN = 256, BK = 128;                // BK in elements = bytes
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
    // split-K over blockIdx.y (E_STORE_F32 only)
    const int ktiles_all = g.K / BK;
    const int z = blockIdx.y;
    const int per = (ktiles_all + g.nsplit - 1) / g.nsplit;
    const int kt0 = z * per, ktiles = max(0, min(ktiles_all, kt0 + per) - kt0);
    const uint8_t* const A = (const uint8_t*)g.A + (long)kt0 * BK;
    const uint8_t* const Bt = (const uint8_t*)g.Bt + (long)kt0 * BK;
    const int wr = wave >> 2, wc = wave & 3;
    const int wm = wr * 128, wn = wc * 64;

    auto issue = [&](int tau, int sidx) {                     // half tiles: see gemm6 (kernels_gemm2.hip)
        const int k = min(tau, max(ktiles - 1, 0)) * BK;
        char* buf = g8_smem + (tau & 1) * BUF;
        const bool isA = (sidx == 0 || sidx == 3);
        const int sub = (sidx == 0 || sidx == 1) ? 0 : 1;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int gidx = i * 8 + wave;
            int row0;
            if (isA) row0 = (gidx >> 3) * 128 + sub * 64 + 8 * (gidx & 7);
            else row0 = 64 * (gidx >> 2) + 32 * sub + 8 * (gidx & 3);
            const int rr = row0 + (lane >> 3);
            const int c = (lane & 7) ^ ((rr >> 1) & 7);
            if (isA)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (long)(bm + rr) * g.lda + k + c * 16),
                                                 (__attribute__((address_space(3))) void*)(buf + row0 * ROWB), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bt + (long)(bn + rr) * g.ldb + k + c * 16),
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

    int8v af[2][2], b0a[2], b0b[2], b1[2];                    // [64-element k step][row tile]
    auto frag = [&](const char* base, int row, int ks) -> int8v {
        const int4v lo = *(const int4v*)(base + swz(row, ks * 4 + 2 * h));
        const int4v hi = *(const int4v*)(base + swz(row, ks * 4 + 2 * h + 1));
        int8v v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };
    auto read_A = [&](const char* As, int sub) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i2 = 0; i2 < 2; i2++) af[ks][i2] = frag(As, wm + sub * 64 + i2 * 32 + r, ks);
    };
    auto read_B = [&](const char* Bs, int sub, int8v (&b)[2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++) b[ks] = frag(Bs, wn + sub * 32 + r, ks);
    };
    auto mma = [&](int subA, int subB, const int8v (&b)[2], int tau, int sidx) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i2 = 0; i2 < 2; i2++)
                acc[subA * 2 + i2][subB] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                    af[ks][i2], b[ks], acc[subA * 2 + i2][subB], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        issue(tau, sidx);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto tile = [&](int t, int8v (&b0)[2], int8v (&b0n)[2]) {      // phases and waits exactly as in gemm6
        const char* As = g8_smem + (t & 1) * BUF;
        const char* Bs = As + A_BYTES;
        const char* Bn = g8_smem + ((t + 1) & 1) * BUF + A_BYTES;
        read_A(As, 0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mma(0, 0, b0, t + 1, 3);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        read_B(Bs, 1, b1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mma(0, 1, b1, t + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        read_A(As, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mma(1, 1, b1, t + 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        read_B(Bn, 0, b0n);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mma(1, 0, b0, t + 2, 2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    issue(0, 0); issue(0, 1); issue(0, 2); issue(0, 3); issue(1, 0); issue(1, 1); issue(1, 2);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_B(g8_smem + A_BYTES, 0, b0a);
    if (wr == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < ktiles; t += 2) {
        tile(t, b0a, b0b);
        if (t + 1 < ktiles) tile(t + 1, b0b, b0a);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
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
                if (m < g.M && n < g.N) ((float*)g.C)[(long)z * g.slab_out_stride + (long)m * g.ldc + n] = alpha * acc[i][j][e];
            }
        }
}

template <int EM>
static void launch8(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = 2 * 2 * 256 * 128;
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)gemm8_fp8_kernel<EM, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)));
    const int tilesM = (g.M + 255) / 256, tilesN = (g.N + 255) / 256;
    const int groups = ((tilesM + 3) / 4) * ((tilesN + 7) / 8);
    GemmArgs a = g;
    if (a.nsplit < 1 || EM != E_STORE_F32) a.nsplit = 1;
    hipLaunchKernelGGL((gemm8_fp8_kernel<EM, 0>), dim3(groups * 32, a.nsplit), dim3(512), LDS, s, a, tilesM, tilesN);
}

bool launch_gemm_fp8(EMode em, const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % 128) != 0) return false;
    if (em == E_SUB_F32) { launch8<E_SUB_F32>(g, s); return true; }
    if (em == E_STORE_F32) { launch8<E_STORE_F32>(g, s); return true; }
    return false;
}

}  // namespace mpqr
