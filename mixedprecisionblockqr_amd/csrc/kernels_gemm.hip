// kernels_gemm.hip -- fp16-operand / fp32-accumulate MFMA GEMM family for gfx950 (CDNA4).
//
// These kernels replace the reference's three GEMM-shaped steps
//   shared_mem_mmult_in_place_transpose_a   Cuda/mmult.cu:236-288  (fp32 trailing update, 32x32 smem tiles)
//   dev_tensorcore_mmult_tiled<half,half,float>  Cuda/mmult.cuh:252-300 (WMMA Q accumulation)
//   dev_cpy_and_cast_array / dev_cpy_strided_array  Cuda/mmult.cuh:104-200 (cast + copy-back passes)
// with one LDS-tiled kernel template built on v_mfma_f32_32x32x16_f16 (wave64).  The casts are
// fused into the staging pass and the copy-back into the epilogue (true in-place update).
//
// All products have the form  C[M x N] = A[M x K] * B[K x N]  with B given as Bt[N][K]
// (k contiguous), so both LDS images are [row][k] and every MFMA fragment is one ds_read_b128.
// Tile 128 x 128 x 64, 256 threads = 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles.
#include "mpqr_internal.h"
#include "gemm_epilogue.h"
#include "gemm_body.h"

namespace mpqr {

using namespace gemm128;

template <int AM, int EM>
__global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) half_t lds[(BM + BN) * LDSP];
    gemm_f16_body<AM, EM>(g, lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

template <int AM, int EM>
static void launch_one(const GemmArgs& g, hipStream_t s) {
    dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN, g.nsplit < 1 ? 1 : g.nsplit);
    GemmArgs a = g;
    if (a.nsplit < 1) a.nsplit = 1;
    if (a.nslab_in < 1) a.nslab_in = 1;
    hipLaunchKernelGGL((gemm_f16_kernel<AM, EM>), grid, dim3(256), 0, s, a);
}

bool launch_gemm_f16(AMode am, EMode em, const GemmArgs& g, hipStream_t s) {
    if (am == A_F32S) am = A_F32;                          // the 128-tile kernel has no split staging (small shapes only)
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return true;
    if (g.C2 || g.A2) return false;                        // hi + lo operands exist in the 256-wide kernels only: a caller bug
#define MPQR_CASE(A_, E_) if (am == A_ && em == E_) { launch_one<A_, E_>(g, s); return true; }
    MPQR_CASE(A_F32T, E_STORE_F32)
    MPQR_CASE(A_F32, E_STORE_H16)
    MPQR_CASE(A_H16, E_SUB_F32)
    MPQR_CASE(A_H16, E_STORE_F32)
    MPQR_CASE(A_H16, E_STORE_H16)
#undef MPQR_CASE
    return false;                                          // no kernel for this staging / epilogue pair
}

// ------------------------------------------------------------------ exact-f32 MFMA GEMM (T merges, metrics, fp32 mode)
// C[M x N] = alpha * op(A) * op(B) + beta * C on v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-exact
// fmaf chain (no reduced-precision step), at the f32 vector peak.  Tile TB x TB x 16, 256 threads = 4 waves
// (2 x 2), LDS images k-major (As[k][m], Bs[k][n]) so every fragment read is a conflict-free ds_read_b32.
constexpr int SK = 16;

template <int TB>
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(SgemmArgs g) {
    constexpr int LDT = TB + 4;
    constexpr int NT = TB / 64;                 // 32x32 MFMA tiles per wave per dimension
    constexpr int NL = TB * SK / 4 / 256;       // float4 loads per thread per operand
    __shared__ __attribute__((aligned(16))) float As[SK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[SK][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bm = blockIdx.y * TB, bn = blockIdx.x * TB;
    const int wm = (wave >> 1) * (TB / 2), wn = (wave & 1) * (TB / 2);

    int k_lo = 0, k_hi = g.K;
    if (g.upperA) k_lo = (bm / SK) * SK;                  // op(A)[i][k] = 0 for k < i
    if (g.upperB) k_hi = min(g.K, bn + TB);               // op(B)[k][j] = 0 for k > j
    float* Cz = g.C;
    if (g.ksplit > 1) {                                   // split K: this workgroup's range and its own output slab
        const int per = ((g.K + g.ksplit - 1) / g.ksplit + SK - 1) / SK * SK;
        k_lo = max(k_lo, (int)blockIdx.z * per); k_hi = min(k_hi, ((int)blockIdx.z + 1) * per);
        Cz += (long)blockIdx.z * g.slab_c;
    }

    const bool a_vec = ((g.lda & 3) == 0) && ((((uintptr_t)g.A) & 15) == 0) && ((g.slab_a & 3) == 0);
    const bool b_vec = ((g.ldb & 3) == 0) && ((((uintptr_t)g.B) & 15) == 0) && ((g.slab_b & 3) == 0);
    const int nsb = g.nslab_b > 1 ? g.nslab_b : 1;

    float4 ra[NL], rb[NL];
    // operand element (row index x in [0,TB), k) ; "kcontig": storage [x][k] (k contiguous) else [k][x]
    auto load_op = [&](const float* P, long ld, bool kcontig, int x0, int xmax, int k0, bool vec, int nslab, long slab,
                       float4* reg) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int id = tid + 256 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kcontig) {
                const int x = id >> 2, kq = (id & 3) * 4;
                const int gx = x0 + x, gk = k0 + kq;
                if (gx < xmax) {
                    const float* p = P + (long)gx * ld + gk;
                    for (int sl = 0; sl < nslab; sl++, p += slab) {
                        if (vec && gk + 3 < g.K) { const float4 t = *(const float4*)p; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                        else {
                            if (gk + 0 < g.K) v.x += p[0]; if (gk + 1 < g.K) v.y += p[1];
                            if (gk + 2 < g.K) v.z += p[2]; if (gk + 3 < g.K) v.w += p[3];
                        }
                    }
                }
            } else {
                const int kk = id / (TB / 4), xq = (id % (TB / 4)) * 4;
                const int gk = k0 + kk, gx = x0 + xq;
                if (gk < g.K) {
                    const float* p = P + (long)gk * ld + gx;
                    for (int sl = 0; sl < nslab; sl++, p += slab) {
                        if (vec && gx + 3 < xmax) { const float4 t = *(const float4*)p; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                        else {
                            if (gx + 0 < xmax) v.x += p[0]; if (gx + 1 < xmax) v.y += p[1];
                            if (gx + 2 < xmax) v.z += p[2]; if (gx + 3 < xmax) v.w += p[3];
                        }
                    }
                }
            }
            reg[i] = v;
        }
    };
    // Fast form of load_op for the common case -- one slab, 16-byte aligned rows, the whole TB x SK piece inside the operand: the loads
    // are unconditional, so a thread's loads of BOTH operands are in flight together.  (load_op's bounds / slab branches made the
    // compiler wait for every load before issuing the next: 50 of 51 loads of sgemm_mfma_kernel<64> serialised in the ISA, and these
    // GEMMs -- the column blocks of the block-level T, two per leaf beside the chain and on its critical path at every block boundary --
    // are short chains of K steps whose time is that latency.)
    auto fast_op = [&](const float* P, long ld, bool kcontig, int x0, int k0, float4* reg) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int id = tid + 256 * i;
            if (kcontig) reg[i] = *(const float4*)(P + (long)(x0 + (id >> 2)) * ld + k0 + (id & 3) * 4);
            else reg[i] = *(const float4*)(P + (long)(k0 + id / (TB / 4)) * ld + x0 + (id % (TB / 4)) * 4);
        }
    };
    auto store_op = [&](float (*Ls)[LDT], bool kcontig, const float4* reg) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int id = tid + 256 * i;
            if (kcontig) {
                const int x = id >> 2, kq = (id & 3) * 4;
                Ls[kq + 0][x] = reg[i].x; Ls[kq + 1][x] = reg[i].y; Ls[kq + 2][x] = reg[i].z; Ls[kq + 3][x] = reg[i].w;
            } else {
                const int kk = id / (TB / 4), xq = (id % (TB / 4)) * 4;
                *(float4*)&Ls[kk][xq] = reg[i];
            }
        }
    };

    floatx16 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const bool a_kc = !g.transA, b_kc = g.transB;
    // uniform over the workgroup: every K step of this workgroup reads full, aligned, single-slab pieces of both operands
    const bool k_full = (k_lo % SK) == 0 && k_hi <= g.K && ((k_hi - k_lo) % SK) == 0;
    const bool fast = k_full && a_vec && b_vec && g.nslab_a <= 1 && nsb <= 1 && bm + TB <= g.M && bn + TB <= g.N;
    if (k_lo < k_hi) {
        if (fast) { fast_op(g.A, g.lda, a_kc, bm, k_lo, ra); fast_op(g.B, g.ldb, b_kc, bn, k_lo, rb); }
        else {
            load_op(g.A, g.lda, a_kc, bm, g.M, k_lo, a_vec, g.nslab_a, g.slab_a, ra);
            load_op(g.B, g.ldb, b_kc, bn, g.N, k_lo, b_vec, nsb, g.slab_b, rb);
        }
        store_op(As, a_kc, ra); store_op(Bs, b_kc, rb);
        __syncthreads();
        for (int k0 = k_lo; k0 < k_hi; k0 += SK) {
            const bool more = k0 + SK < k_hi;
            if (more) {
                if (fast) { fast_op(g.A, g.lda, a_kc, bm, k0 + SK, ra); fast_op(g.B, g.ldb, b_kc, bn, k0 + SK, rb); }
                else {
                    load_op(g.A, g.lda, a_kc, bm, g.M, k0 + SK, a_vec, g.nslab_a, g.slab_a, ra);
                    load_op(g.B, g.ldb, b_kc, bn, g.N, k0 + SK, b_vec, nsb, g.slab_b, rb);
                }
            }
#pragma unroll
            for (int ks = 0; ks < SK / 2; ks++) {
                float a[NT], b[NT];
                const int kr = ks * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < NT; i++) a[i] = As[kr][wm + i * 32 + (lane & 31)];
#pragma unroll
                for (int j = 0; j < NT; j++) b[j] = Bs[kr][wn + j * 32 + (lane & 31)];
#pragma unroll
                for (int i = 0; i < NT; i++)
#pragma unroll
                    for (int j = 0; j < NT; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) { store_op(As, a_kc, ra); store_op(Bs, b_kc, rb); __syncthreads(); }
        }
    }
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int n = bn + wn + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int m = bm + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m < g.M && n < g.N) {
                    float* p = Cz + (long)m * g.ldc + n;
                    float v = g.alpha * acc[i][j][e];
                    if (g.beta != 0.f && g.ksplit <= 1) v += g.beta * (*p);
                    *p = v;
                }
            }
        }
}

void launch_sgemm(const SgemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0) return;
    SgemmArgs a = g;
    if (a.nslab_a < 1) a.nslab_a = 1;
    const long tiles128 = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
    if (a.ksplit < 1) a.ksplit = 1;
    if (tiles128 >= 256) {
        dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, a.ksplit);
        hipLaunchKernelGGL(sgemm_mfma_kernel<128>, grid, dim3(256), 0, s, a);
    } else {
        dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, a.ksplit);
        hipLaunchKernelGGL(sgemm_mfma_kernel<64>, grid, dim3(256), 0, s, a);
    }
}

}  // namespace mpqr
