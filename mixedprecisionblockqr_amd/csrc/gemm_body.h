// gemm_body.h -- the 128 x 128 x 64 fp16 MFMA GEMM tile of kernels_gemm.hip as a device function
#pragma once
#include "mpqr_internal.h"
#include "gemm_epilogue.h"

namespace mpqr {
namespace gemm128 {
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

}  // namespace gemm128

// One 128 x 128 output tile (split-K slab z) by the calling 256-thread workgroup: tile (bx, by), LDS image `lds` ((BM + BN) * LDSP halves,
// 16-byte aligned).  A device function so that a heterogeneous launch (kernels_panel.hip: leaf_mid_kernel) can run GEMM tiles beside other work.
template <int AM, int EM>
__device__ __forceinline__ void gemm_f16_body(const GemmArgs& g, half_t* lds, int bx, int by, int z) {
    using namespace gemm128;
    half_t* As = lds;
    half_t* Bs = lds + BM * LDSP;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bm = bx * BM, bn = by * BN;

    const int ktiles = g.K / BK;
    const int per = (ktiles + g.nsplit - 1) / g.nsplit;
    const int kt0 = z * per;
    const int kt1 = min(ktiles, kt0 + per);

    // ---- staging registers: TWO sets for the fp16 / fp32-slab forms (round 5).  The K loop of this kernel is latency bound wherever it is
    // used (K tiles of 64 with 16 MFMAs each: part (a)'s X = A2^T V at 16384 rows ran at 170 TFLOP/s, its Y = X T' 53 us for 2 GFLOP):
    // the loads of tile kt + 2 are issued before tile kt is multiplied, so two load latencies overlap.  The A_F32T form keeps one set:
    // with two it needs 212 VGPRs, one workgroup per CU, and it is the T stream's X GEMM of every leaf.
    constexpr bool DEEP = (AM != A_F32T);
    U4 ra16_[2][4];        // A_H16 / A_F32 (after conversion)
    float4 raT_[2][8];     // A_F32T
    U4 rb_[2][4];

    auto load_tile = [&](int kt, U4 (&ra16)[4], float4 (&raT)[8], U4 (&rb)[4]) {
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

    auto store_tile = [&](const U4 (&ra16)[4], const float4 (&raT)[8], const U4 (&rb)[4]) {
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

    auto mma_tile = [&]() {
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
    };
    if (!DEEP) {
        if (kt0 < kt1) {
            load_tile(kt0, ra16_[0], raT_[0], rb_[0]);
            store_tile(ra16_[0], raT_[0], rb_[0]);
            __syncthreads();
            for (int kt = kt0; kt < kt1; kt++) {
                const bool more = (kt + 1 < kt1);
                if (more) load_tile(kt + 1, ra16_[0], raT_[0], rb_[0]);
                mma_tile();
                __syncthreads();
                if (more) {
                    store_tile(ra16_[0], raT_[0], rb_[0]);
                    __syncthreads();
                }
            }
        }
    } else if (kt0 < kt1) {
        load_tile(kt0, ra16_[0], raT_[0], rb_[0]);
        if (kt0 + 1 < kt1) load_tile(kt0 + 1, ra16_[1], raT_[1], rb_[1]);
        store_tile(ra16_[0], raT_[0], rb_[0]);
        __syncthreads();
        // two K tiles per trip so that the register sets are compile-time constants: set 0 holds the even tiles (from kt0), set 1 the odd ones
        for (int kt = kt0; kt < kt1; kt += 2) {
            if (kt + 2 < kt1) load_tile(kt + 2, ra16_[0], raT_[0], rb_[0]);       // (set 0 went to LDS before tile kt was multiplied)
            mma_tile();                                                            // tile kt
            __syncthreads();
            if (kt + 1 < kt1) {
                store_tile(ra16_[1], raT_[1], rb_[1]);
                __syncthreads();
                if (kt + 3 < kt1) load_tile(kt + 3, ra16_[1], raT_[1], rb_[1]);
                mma_tile();                                                        // tile kt + 1
                __syncthreads();
                if (kt + 2 < kt1) {
                    store_tile(ra16_[0], raT_[0], rb_[0]);
                    __syncthreads();
                }
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


}  // namespace mpqr
