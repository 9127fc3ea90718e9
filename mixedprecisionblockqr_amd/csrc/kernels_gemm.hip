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

namespace mpqr {

typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDSP = BK + 8;   // padded LDS row (halves): 144-B stride is conflict-free for ds_read_b128

typedef uint32_t U4 __attribute__((ext_vector_type(4)));   // a first-class vector: staging arrays of it stay in registers

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef half_t half2v __attribute__((ext_vector_type(2)));
    half2v h = {(half_t)a, (half_t)b};
    return __builtin_bit_cast(uint32_t, h);
}

template <int AM, int EM>
__global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) half_t lds[(BM + BN) * LDSP];
    half_t* As = lds;
    half_t* Bs = lds + BM * LDSP;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bm = blockIdx.x * BM, bn = blockIdx.y * BN;
    const int z = blockIdx.z;

    const int ktiles = g.K / BK;
    const int per = (ktiles + g.nsplit - 1) / g.nsplit;
    const int kt0 = z * per;
    const int kt1 = min(ktiles, kt0 + per);

    // ---- staging registers
    U4 ra16[4];        // A_H16 / A_F32 (after conversion)
    float4 raT[8];     // A_F32T
    U4 rb[4];

    auto load_tile = [&](int kt) {
        const int k = kt * BK;
        if (AM == A_H16) {
            const half_t* A = (const half_t*)g.A;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 8;
                int gm = bm + row;
                U4 v = {0, 0, 0, 0};
                if (gm < g.M) v = *(const U4*)(A + (long)gm * g.lda + k + kc);
                ra16[i] = v;
            }
        } else if (AM == A_F32T) {
            const float* A = (const float*)g.A;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                int id = tid + 256 * i, mg = id & 31, kg = id >> 5;
                int gm = bm + mg * 4;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gm < g.M) v = *(const float4*)(A + (long)(k + kg * 4 + j) * g.lda + gm);
                    raT[i * 4 + j] = v;
                }
            }
        } else {  // A_F32 with slab sum
            const float* A = (const float*)g.A;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 8;
                int gm = bm + row;
                float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
                if (gm < g.M) {
                    const float* p = A + (long)gm * g.lda + k + kc;
                    for (int sl = 0; sl < g.nslab_in; sl++) {
                        float4 a0 = *(const float4*)(p + (long)sl * g.slab_in_stride);
                        float4 a1 = *(const float4*)(p + (long)sl * g.slab_in_stride + 4);
                        s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
                        s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
                    }
                }
                const float sc = g.in_scale;
                U4 v;
                v.x = pack2(s0.x * sc, s0.y * sc); v.y = pack2(s0.z * sc, s0.w * sc);
                v.z = pack2(s1.x * sc, s1.y * sc); v.w = pack2(s1.z * sc, s1.w * sc);
                ra16[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 8;
            int gn = bn + row;
            U4 v = {0, 0, 0, 0};
            if (gn < g.N) v = *(const U4*)(g.Bt + (long)gn * g.ldb + k + kc);
            rb[i] = v;
        }
    };

    auto store_tile = [&]() {
        if (AM == A_F32T) {
            const float sc = g.in_scale;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                int id = tid + 256 * i, mg = id & 31, kg = id >> 5;
                const float4 v0 = raT[i * 4 + 0], v1 = raT[i * 4 + 1], v2 = raT[i * 4 + 2], v3 = raT[i * 4 + 3];
                uint2 w;
                half_t* base = As + (mg * 4) * LDSP + kg * 4;
                w.x = pack2(v0.x * sc, v1.x * sc); w.y = pack2(v2.x * sc, v3.x * sc); *(uint2*)(base) = w;
                w.x = pack2(v0.y * sc, v1.y * sc); w.y = pack2(v2.y * sc, v3.y * sc); *(uint2*)(base + LDSP) = w;
                w.x = pack2(v0.z * sc, v1.z * sc); w.y = pack2(v2.z * sc, v3.z * sc); *(uint2*)(base + 2 * LDSP) = w;
                w.x = pack2(v0.w * sc, v1.w * sc); w.y = pack2(v2.w * sc, v3.w * sc); *(uint2*)(base + 3 * LDSP) = w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 8;
                *(U4*)(As + row * LDSP + kc) = ra16[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int c = tid + 256 * i, row = c >> 3, kc = (c & 7) * 8;
            *(U4*)(Bs + row * LDSP + kc) = rb[i];
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    if (kt0 < kt1) {
        load_tile(kt0);
        store_tile();
        __syncthreads();
        for (int kt = kt0; kt < kt1; kt++) {
            const bool more = (kt + 1 < kt1);
            if (more) load_tile(kt + 1);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ks++) {
                half8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; i++) a[i] = *(const half8*)(As + (wm + i * 32 + r) * LDSP + ks * 16 + h * 8);
#pragma unroll
                for (int j = 0; j < 2; j++) b[j] = *(const half8*)(Bs + (wn + j * 32 + r) * LDSP + ks * 16 + h * 8);
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                store_tile();
                __syncthreads();
            }
        }
    }

    // ---- epilogue.  D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const float alpha = g.alpha;
    if (EM == E_SUB_F32) {
        epilogue_sub_f32<2, 2>(acc, (float*)g.C, g.ldc, g.M, g.N, g.col_lo, alpha, bm + wm, bn + wn, r, h, g.Ct, g.ldct, g.ct_scale);
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = bn + wn + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int m = bm + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < g.M && n < g.N) {
                    const float v = alpha * acc[i][j][e];
                    if (EM == E_STORE_F32) ((float*)g.C)[(long)z * g.slab_out_stride + (long)m * g.ldc + n] = v;
                    else ((half_t*)g.C)[(long)m * g.ldc + n] = (half_t)(g.cscale ? v * g.cscale[(long)n * g.cscale_ld] : v);
                }
            }
        }
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
