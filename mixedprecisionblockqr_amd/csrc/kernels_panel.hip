// kernels_panel.hip -- fp32 Householder panel ("leaf") factorisation and compact-WY T construction.
//
// Replaces the reference's host-side panel  h_householder_qr  (Cuda/qr.cu:198-293, run on the CPU
// with a whole-matrix H2D/D2H per panel, qr.cu:1080-1082,1215) and its WY builder dev_wy_transform
// (qr.cu:428-600, dense (m-l)^2 Q_panel) by device kernels working on the resident matrix.
//
// Leaf = up to 32 adjacent columns inside one 32-aligned column window, all rows below the
// diagonal.  One launch per column, many workgroups per launch (256 rows each), no inter-workgroup
// waiting inside a launch: launch k applies reflector k to the leaf columns AND accumulates the
// partial dot products  a_{k+1}^T a_j  (j >= k+1) that reflector k+1 needs, so the next launch can
// derive ||u||, v^T a_j and ||v|| algebraically:
//     u = A[k:,k], s_j = u^T a_j, alpha = sgn(u_0)||u||, v = (u + alpha e_1)/||u + alpha e_1||,
//     ||u + alpha e_1||^2 = 2 (s_k + |u_0| ||u||),   v^T a_j = (s_j + alpha a_kj)/||.||
// (sign rule and zero-column skip exactly as qr.cu:229-244; R_kk = -alpha).
// Reductions: 8-lane row groups -> DPP/xor shuffles across the wave64 -> LDS across the 4 waves
// -> per-workgroup partial in HBM, summed by every workgroup of the next launch.
#include "mpqr_internal.h"
#include "panel_dev.h"
#include "gemm_body.h"

namespace mpqr {

constexpr int RPW = 256;   // rows per workgroup


__device__ __forceinline__ float pick(const float4& v, int c) {
    return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
}

// partial dots of column kn with leaf columns over rows >= row_lo handled by this workgroup
// (used to start a leaf: kn = c0, no reflector applied)
__global__ __launch_bounds__(256) void leaf_init_kernel(LeafArgs a, int kn, float* __restrict__ Pout) {
    __shared__ float sh_part[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = tid & 7, rl = tid >> 3;
    const int row0 = kn + blockIdx.x * RPW;
    const int kq = kn - a.cb;
    float pacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < RPW / 32; p++) {
        const int row = row0 + p * 32 + rl;
        const bool valid = row < a.mrows;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = *(const float4*)(a.A + (long)row * a.lda + a.cb + 4 * cg);
        const float comp = pick(v, kq & 3);
        const float aikn = __shfl(comp, (lane & ~7) | (kq >> 2));
        pacc[0] += aikn * v.x; pacc[1] += aikn * v.y; pacc[2] += aikn * v.z; pacc[3] += aikn * v.w;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float s = pacc[c];
        s += __shfl_xor(s, 8); s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        pacc[c] = s;
    }
    if (lane < 8) {
#pragma unroll
        for (int c = 0; c < 4; c++) sh_part[wave][lane * 4 + c] = pacc[c];
    }
    __syncthreads();
    if (tid < 32) Pout[blockIdx.x * 32 + tid] = sh_part[0][tid] + sh_part[1][tid] + sh_part[2][tid] + sh_part[3][tid];
}

__global__ __launch_bounds__(256) void leaf_step_kernel(LeafArgs a, int k, int nwg_in,
                                                        const float* __restrict__ Pin, float* __restrict__ Pout) {
    __shared__ float sh_part[8][32];
    __shared__ float sh_s[32], sh_rowk[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = tid & 7, rl = tid >> 3;
    const int kq = k - a.cb;
    // the workgroup's rows of the window first (clamped, masked in phase 1): eight loads in flight behind phase 0, instead of eight
    // memory latencies in a row inside phase 1 (the load sat in the bounds branch: load -> s_waitcnt vmcnt(0) -> use, eight times)
    float4 vin[RPW / 32];
#pragma unroll
    for (int p = 0; p < RPW / 32; p++)
        vin[p] = *(const float4*)(a.A + (long)min(k + (int)blockIdx.x * RPW + p * 32 + rl, a.mrows - 1) * a.lda + a.cb + 4 * cg);

    // ---- phase 0: every workgroup sums the previous launch's partials (tiny, L2-resident)
    {
        const int j = tid & 31, gq = tid >> 5;
        float s = 0.f;
        for (int w = gq; w < nwg_in; w += 8) s += Pin[w * 32 + j];
        sh_part[gq][j] = s;
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) s += sh_part[q][tid];
        sh_s[tid] = s;
        sh_rowk[tid] = a.A[(long)k * a.lda + a.cb + tid];
    }
    __syncthreads();
    const float sk = sh_s[kq], akk = sh_rowk[kq];
    float alpha = 0.f, inv = 0.f;
    if (sk != 0.f) {                       // exactly-zero column: skipped (qr.cu:242-244)
        const float nu = sqrtf(sk);
        alpha = (akk >= 0.f) ? nu : -nu;
        inv = 1.0f / sqrtf(2.0f * (sk + fabsf(akk) * nu));
    }
    float w[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int j = 4 * cg + c, gj = a.cb + j;
        w[c] = (gj > k && gj < a.c1) ? 2.0f * (sh_s[j] + alpha * sh_rowk[j]) * inv : 0.f;
    }
    __syncthreads();   // sh_part is reused below

    // ---- phase 1: apply reflector k to my rows, emit v, accumulate dots for reflector k+1
    const int kn = k + 1;
    const bool have_next = kn < a.c1;
    const int knq = have_next ? kn - a.cb : kq;
    const bool own_k = (cg == (kq >> 2));
    const int row0 = k + blockIdx.x * RPW;
    float pacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < RPW / 32; p++) {
        const int row = row0 + p * 32 + rl;
        const bool valid = row < a.mrows;
        float* ptr = a.A + (long)row * a.lda + a.cb + 4 * cg;
        float4 v = valid ? vin[p] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float comp = pick(v, kq & 3);
        const float aik = __shfl(comp, (lane & ~7) | (kq >> 2));
        const float vi = (aik + (row == k ? alpha : 0.f)) * inv;
        v.x -= vi * w[0]; v.y -= vi * w[1]; v.z -= vi * w[2]; v.w -= vi * w[3];
        if (own_k && sk != 0.f) {
            const float nv = (row == k) ? -alpha : vi;     // R_kk above, reflector below
            const int c = kq & 3;
            if (c == 0) v.x = nv; else if (c == 1) v.y = nv; else if (c == 2) v.z = nv; else v.w = nv;
        }
        if (valid) {
            *(float4*)ptr = v;
            if (own_k) {
                a.Vh[(long)row * a.ldvh + k] = (half_t)vi;
                a.Vt[(long)k * a.ldvt + row] = (half_t)vi;
                if (row == k) a.vdiag[k] = vi;
            }
        }
        const float comp2 = pick(v, knq & 3);
        const float aikn = __shfl(comp2, (lane & ~7) | (knq >> 2));
        if (have_next && valid && row > k) {
            pacc[0] += aikn * v.x; pacc[1] += aikn * v.y; pacc[2] += aikn * v.z; pacc[3] += aikn * v.w;
        }
    }
    if (!have_next) return;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float s = pacc[c];
        s += __shfl_xor(s, 8); s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        pacc[c] = s;
    }
    if (lane < 8) {
#pragma unroll
        for (int c = 0; c < 4; c++) sh_part[wave][lane * 4 + c] = pacc[c];
    }
    __syncthreads();
    if (tid < 32) Pout[blockIdx.x * 32 + tid] = sh_part[0][tid] + sh_part[1][tid] + sh_part[2][tid] + sh_part[3][tid];
}

// ------------------------------------------------------------------ whole leaf in one workgroup
// Panels of at most 128*RPT rows: the leaf (rows >= c0, the 32-column window) lives in the registers of
// one 1024-thread workgroup (128 row lanes x 8 column groups, RPT rows per thread).  Same arithmetic as
// leaf_step_kernel, but the per-column reduction stays on chip (wave shuffles + one LDS hop), so a leaf
// costs one launch instead of 33.
template <int RPT>
__global__ __launch_bounds__(1024) void leaf_wg_kernel(LeafArgs a) {
    __shared__ float sh_part[16][32];
    __shared__ float sh_s[2][32], sh_rowk[2][32], sh_vd[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = tid & 7, rl = tid >> 3;
    const int row0 = a.c0;
    if (tid < 32) sh_vd[tid] = 0.f;
    float4 v[RPT];
#pragma unroll
    for (int p = 0; p < RPT; p++) {
        const int row = row0 + p * 128 + rl;
        v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < a.mrows) v[p] = *(const float4*)(a.A + (long)row * a.lda + a.cb + 4 * cg);
    }
    float pacc[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const int kq = a.c0 - a.cb;
#pragma unroll
        for (int p = 0; p < RPT; p++) {
            const float aik = __shfl(pick(v[p], kq & 3), (lane & ~7) | (kq >> 2));
            pacc[0] += aik * v[p].x; pacc[1] += aik * v[p].y; pacc[2] += aik * v[p].z; pacc[3] += aik * v[p].w;
        }
    }
    for (int k = a.c0; k < a.c1; k++) {
        const int kq = k - a.cb, par = k & 1;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float s = pacc[c];
            s += __shfl_xor(s, 8); s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            pacc[c] = s;
        }
        if (lane < 8) {
#pragma unroll
            for (int c = 0; c < 4; c++) sh_part[wave][lane * 4 + c] = pacc[c];
        }
#pragma unroll
        for (int p = 0; p < RPT; p++)
            if (row0 + p * 128 + rl == k) *(float4*)&sh_rowk[par][4 * cg] = v[p];
        __syncthreads();
        if (tid < 32) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; q++) s += sh_part[q][tid];
            sh_s[par][tid] = s;
        }
        __syncthreads();
        const float sk = sh_s[par][kq], akk = sh_rowk[par][kq];
        float alpha = 0.f, inv = 0.f;
        if (sk != 0.f) {
            const float nu = sqrtf(sk);
            alpha = (akk >= 0.f) ? nu : -nu;
            inv = 1.0f / sqrtf(2.0f * (sk + fabsf(akk) * nu));
        }
        float w[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = 4 * cg + c, gj = a.cb + j;
            w[c] = (gj > k && gj < a.c1) ? 2.0f * (sh_s[par][j] + alpha * sh_rowk[par][j]) * inv : 0.f;
        }
        const int kn = k + 1;
        const bool have_next = kn < a.c1;
        const int knq = have_next ? kn - a.cb : kq;
        const bool own_k = (cg == (kq >> 2));
        pacc[0] = pacc[1] = pacc[2] = pacc[3] = 0.f;
        const int d = k - row0 - rl;                 // slot p holds row k iff p*128 == d
        const int src_k = (lane & ~7) | (kq >> 2), src_n = (lane & ~7) | (knq >> 2);
        const int ck = kq & 3, cn = knq & 3;
#pragma unroll
        for (int p = 0; p < RPT; p++) {
            const float aik = __shfl(pick(v[p], ck), src_k);
            const float vi = (p * 128 >= d) ? (aik + (p * 128 == d ? alpha : 0.f)) * inv : 0.f;
            v[p].x -= vi * w[0]; v[p].y -= vi * w[1]; v[p].z -= vi * w[2]; v[p].w -= vi * w[3];
            if (own_k && p * 128 >= d && sk != 0.f) {
                const float nv = (p * 128 == d) ? -alpha : vi;
                if (ck == 0) v[p].x = nv; else if (ck == 1) v[p].y = nv; else if (ck == 2) v[p].z = nv; else v[p].w = nv;
            }
            if (own_k && p * 128 == d) sh_vd[kq] = vi;
            const float aikn = __shfl(pick(v[p], cn), src_n);
            if (have_next && p * 128 > d) {
                pacc[0] += aikn * v[p].x; pacc[1] += aikn * v[p].y; pacc[2] += aikn * v[p].z; pacc[3] += aikn * v[p].w;
            }
            if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    if (tid < 32) {
        const int k = a.cb + tid;
        if (k >= a.c0 && k < a.c1 && k < a.mrows) {
            const float vd = sh_vd[tid];
            a.vdiag[k] = vd;
            a.Vh[(long)k * a.ldvh + k] = (half_t)vd;
            a.Vt[(long)k * a.ldvt + k] = (half_t)vd;
        }
    }
    // write back: A (R above the diagonal, reflectors below), fp16 copies of the reflectors
#pragma unroll
    for (int p = 0; p < RPT; p++) {
        const int row = row0 + p * 128 + rl;
        if (row < a.mrows) {
            *(float4*)(a.A + (long)row * a.lda + a.cb + 4 * cg) = v[p];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int gc = a.cb + 4 * cg + c;
                if (gc >= a.c0 && gc < a.c1 && row > gc) {
                    const half_t hv = (half_t)pick(v[p], c);
                    a.Vh[(long)row * a.ldvh + gc] = hv;
                    a.Vt[(long)gc * a.ldvt + row] = hv;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------ tall leaves: "Gram-Householder"
// Leaf = up to 128 adjacent columns [c0,c1) inside one 128-aligned window [cb, cb+128), w = c1-c0.
// After reflectors c0..k-1 every leaf column is (a combination of the ORIGINAL leaf columns) below row c1 plus
// explicit values in the top w rows:  B_low = A_low M,  B_top explicit.  All inner products over the tall part
// therefore follow from G = A_low^T A_low (w x w), and the Householder recursion (same u, alpha, v, w_j and sign
// rule as the kernels above; exactly-zero columns are handed to them through the flag) runs on w x w matrices:
//   gh_gram    : per-workgroup partial G over 128 rows; fp32 data, products and sums in fp64 (f64 MFMA, upper tiles)
//   gh_reduce  : G = sum of the partials (fixed order: deterministic), mirrored to the full symmetric matrix
//   gh_solve   : one workgroup; N = Gram of all not-yet-final rows (fp64) and B_top (fp32) in registers; a Cholesky
//                chain on N and a Householder chain on B_top on separate waves (see the kernel); emits R, V_top,
//                C (V_low = A_low C, one triangular inverse at the end) and checks rho_k = ||u_k||^2 / ||a_k||^2
//   gh_apply   : V_low = A_low C on the exact-f32 MFMA, written over A_low, plus fp16 copies V and V^T and the
//                partial Gram matrices of the fp16 V (for the leaf's T)
// No pass over the tall data is sequential in k.  V differs from Householder's by O(2^-24 / sqrt(rho)); a leaf
// with rho < GH_RHO_MIN or a column that cannot be reflected raises a flag and the driver redoes the work on the
// column-by-column kernels above.
constexpr int GH_TS = 132;       // LDS row stride (floats) of a staged 128-column tile
constexpr int GH_ROWS = 128;     // rows per workgroup in gh_gram

constexpr int GH_TD = 144;       // LDS row stride (doubles) of the staged tile: rows k, k+1 of one MFMA operand read land on disjoint bank halves
__global__ __launch_bounds__(256) void gh_gram_kernel(LeafArgs a, double* __restrict__ Gp, int iters) {
    double* tile = (double*)gh_smem;                      // [GH_ROWS][GH_TD] doubles: converted once while staging
    const int tid = threadIdx.x;
    KT_DECL; KT();
    const int hb = blockIdx.x & 1;                        // which half of G's rows this workgroup produces
    // G = T^T T on v_mfma_f64_16x16x4_f64 (A[i][k] from lane i + 16k, B[k][j] from lane j + 16k, D[i][j] in lane j + 16 (i%4),
    // element i/4 -- probed, tools/probe_mfma_f64.hip).  G is symmetric: only the 36 upper 16 x 16 tiles are computed,
    // tile t = hb + 2 wave + 8 s  (s < 5) by this wave; gh_reduce mirrors the sum.
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    int tti[5], ttj[5];
    typedef double double4g __attribute__((ext_vector_type(4)));
    double4g acc[5];
#pragma unroll
    for (int s = 0; s < 5; s++) {
        int t = hb + 2 * wave + 8 * s;                       // row-major index among the upper tiles of an 8 x 8 tile grid
        int ti = 0;
        if (t >= 36) { tti[s] = -1; ttj[s] = 0; }
        else {
            while (t >= 8 - ti) { t -= 8 - ti; ti++; }
            tti[s] = ti; ttj[s] = ti + t;
        }
        acc[s] = double4g{0, 0, 0, 0};
    }
    int ca[5], cb[5];
#pragma unroll
    for (int s = 0; s < 5; s++) { ca[s] = tti[s] >= 0 ? 16 * tti[s] : 0; cb[s] = tti[s] >= 0 ? 16 * ttj[s] : 0; }
    // `iters` blocks of GH_ROWS rows per workgroup pair (tall leaves: gh_gram_iters): the accumulators carry over, a 65536-row leaf
    // writes 128 partials (9 MB) instead of 512 (37 MB, more than the rows it reads)
    for (int it = 0; it < iters; it++) {
    const int row0 = a.c0 + ((int)(blockIdx.x >> 1) * iters + it) * GH_ROWS;   // ALL leaf rows, top block included: G = Gram of the remaining rows at k = c0
    if (row0 >= a.mrows) break;
    if (it) __syncthreads();                               // every wave is done with the previous tile
    // all 16 loads of a thread first (clamped row, masked afterwards), then the conversions and LDS stores: with the load inside the bounds
    // branch the compiler emitted load -> s_waitcnt vmcnt(0) -> store sixteen times, one memory latency each -- 14-24 k cycles of this
    // kernel's ~30 k (in-kernel stamps tools/ktrace_solve.sh; ISA: hipcc -S)
    float4 gv[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int id = tid + 256 * i, lr = id >> 5, c4 = id & 31, row = min(row0 + lr, a.mrows - 1);
        gv[i] = *(const float4*)(a.A + (long)row * a.lda + a.cb + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int id = tid + 256 * i, lr = id >> 5, c4 = id & 31, row = row0 + lr;
        const float4 v = (row < a.mrows) ? gv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        double* d = &tile[lr * GH_TD + 4 * c4];
        *(double2*)d = make_double2((double)v.x, (double)v.y);
        *(double2*)(d + 2) = make_double2((double)v.z, (double)v.w);
    }
    __syncthreads();
    KT();
    // two K steps per iteration, all 20 operand reads in flight before the 10 MFMAs (the loop was LDS-latency bound: 10 of
    // the kernel's 16 us); slots without a tile recompute tile 0 (never stored) so that the loop has no branches
    for (int k0 = 0; k0 < GH_ROWS; k0 += 8) {
        const double* tr0 = &tile[(k0 + lk) * GH_TD + li];
        const double* tr1 = tr0 + 4 * GH_TD;
        double a0[5], b0[5], a1[5], b1[5];
#pragma unroll
        for (int s = 0; s < 5; s++) { a0[s] = tr0[ca[s]]; b0[s] = tr0[cb[s]]; a1[s] = tr1[ca[s]]; b1[s] = tr1[cb[s]]; }
#pragma unroll
        for (int s = 0; s < 5; s++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b0[s], acc[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 5; s++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b1[s], acc[s], 0, 0, 0);
    }
    KT();
    }                                                      // row blocks
    double* out = Gp + (long)(blockIdx.x >> 1) * (GW * GW);
#pragma unroll
    for (int s = 0; s < 5; s++)
        if (tti[s] >= 0) {
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int i = 16 * tti[s] + lk + 4 * v, j = 16 * ttj[s] + li;
                out[i * GW + j] = acc[s][v];                 // upper tiles only: gh_reduce mirrors
            }
        }
    KT(); KT_DUMP(3, "gh_gram load|mfma|store");
}
__global__ __launch_bounds__(256) void gh_reduce_kernel(const double* __restrict__ Gp, int nwg, double* __restrict__ G) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int i = e >> 7, j = e & 127;
    const bool act = (i >> 4) <= (j >> 4);                // 16 x 16 tiles below the diagonal are not produced: mirrored below
    double s = 0;
    if (act) {                                            // 16 loads in flight per lane, summed in slab order
        int q = wave;
        for (; q + 60 < nwg; q += 64) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = Gp[(long)(q + 4 * u) * (GW * GW) + e];
#pragma unroll
            for (int u = 0; u < 16; u++) s += v[u];
        }
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = (q + 4 * u < nwg) ? Gp[(long)(q + 4 * u) * (GW * GW) + e] : 0.0;
#pragma unroll
        for (int u = 0; u < 16; u++) s += v[u];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && act) {
        const double g = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        G[e] = g;
        if ((i >> 4) != (j >> 4)) G[j * GW + i] = g;       // mirror: G is symmetric
    }
}


typedef half_t half8p __attribute__((ext_vector_type(8)));
// partial Gram of the fp16 reflectors held in Ts ([128 columns][72]: the 64 rows of this workgroup, k contiguous):
// upper 32 x 32 tiles of Ts Ts^T on v_mfma_f32_32x32x16_f16, written to Sp (fp32, 128 x 128, window coordinates)
// ga[0..2]: this wave's (up to three) 32 x 32 upper tiles of the partial Gram matrix; add = accumulate over 64 more rows in Ts
__device__ __forceinline__ void gh_partial_gram_add(const half_t* Ts, floatx16p (&ga)[3], int lane, int wave) {
    const int r = lane & 31, h = lane >> 5;
    int t = 0;
#pragma unroll
    for (int ti = 0; ti < 4; ti++)
#pragma unroll
        for (int tj = ti; tj < 4; tj++, t++) {
            if ((t & 3) != wave) continue;               // 10 upper tiles dealt to the 4 waves: tile t is this wave's number t / 4
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const half8p av = *(const half8p*)&Ts[(ti * 32 + r) * 72 + ks * 16 + 8 * h];
                const half8p bv = *(const half8p*)&Ts[(tj * 32 + r) * 72 + ks * 16 + 8 * h];
                ga[t >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, ga[t >> 2], 0, 0, 0);
            }
        }
}
__device__ __forceinline__ void gh_partial_gram_store(const floatx16p (&ga)[3], float* __restrict__ Sp, int lane, int wave) {
    const int r = lane & 31, h = lane >> 5;
    int t = 0;
#pragma unroll
    for (int ti = 0; ti < 4; ti++)
#pragma unroll
        for (int tj = ti; tj < 4; tj++, t++) {
            if ((t & 3) != wave) continue;
#pragma unroll
            for (int e = 0; e < 16; e++) Sp[(ti * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * GW + tj * 32 + r] = ga[t >> 2][e];
        }
}
__device__ __forceinline__ void gh_partial_gram(const half_t* Ts, float* __restrict__ Sp, int lane, int wave) {
    floatx16p ga[3];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int e = 0; e < 16; e++) ga[q][e] = 0.f;
    gh_partial_gram_add(Ts, ga, lane, wave);
    gh_partial_gram_store(ga, Sp, lane, wave);
}
__global__ __launch_bounds__(256) void gh_apply_kernel(LeafArgs a, const float* __restrict__ Cv, float* __restrict__ Sp, int nlow, int iters) {
    float* As = (float*)gh_smem;                         // [64][129] = 8256 floats
    float* Cs = (float*)gh_smem + 8256;                  // [128][GH_TS], 16-B aligned
    half_t* Ts = (half_t*)gh_smem;                       // [128][72] halves, reuses the As region after the MFMAs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    KT_DECL; KT();
    if ((int)blockIdx.x >= nlow) {                        // top-block rows: V^T tile straight from the fp16 copy
        const int trow0 = a.c0 + ((int)blockIdx.x - nlow) * 64;
        for (int e = tid; e < GW * 64; e += 256) {
            const int c = e & 127, lr = e >> 7, row = trow0 + lr, gc = a.cb + c;
            half_t v = (half_t)0.f;
            if (row < a.c1 && gc >= a.c0 && gc < a.c1 && row >= gc) v = a.Vh[(long)row * a.ldvh + gc];
            Ts[c * 72 + lr] = v;
        }
        __syncthreads();
        gh_partial_gram(Ts, Sp + (long)blockIdx.x * (GW * GW), lane, wave);
        return;
    }
    // C arrives in window coordinates, zero outside the leaf and below the diagonal
#pragma unroll
    for (int q = 0; q < GW * GW / 4 / 256; q++) {
        const int e4 = tid + 256 * q, wi = e4 >> 5, wk = (e4 & 31) * 4;
        *(float4*)&Cs[wi * GH_TS + wk] = *(const float4*)&Cv[wi * GW + wk];
    }
    // `iters` blocks of 64 rows per workgroup (tall leaves: gh_apply_iters): C is staged once, and the partial Gram matrix of the rounded
    // reflectors accumulates over the blocks -- a quarter of the 67 MB of partials a 65536-row leaf would write (and leaf_mid would read)
    floatx16p ga[3];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int e = 0; e < 16; e++) ga[q][e] = 0.f;
    for (int it = 0; it < iters; it++) {
    const int row0 = a.c1 + ((int)blockIdx.x * iters + it) * 64;
    if (row0 >= a.mrows) break;
    if (it) __syncthreads();                               // the previous block's V^T rows have been read from Ts (= As)
    float4 av[8];                                          // (loads first, as in gh_gram: eight serialised latencies otherwise)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int id = tid + 256 * i, lr = id >> 5, c4 = id & 31, row = min(row0 + lr, a.mrows - 1);
        av[i] = *(const float4*)(a.A + (long)row * a.lda + a.cb + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int id = tid + 256 * i, lr = id >> 5, c4 = id & 31, row = row0 + lr;
        const float4 v = (row < a.mrows) ? av[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = &As[lr * 129 + 4 * c4];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    KT();
    floatx16p acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; e++) { acc0[e] = 0.f; acc1[e] = 0.f; }
    const int n0 = 32 * wave, r = lane & 31, kk = lane >> 5;
    const int kend = min(GW, n0 + 32);                   // C is upper triangular: C[k][j] = 0 for k > j
    for (int k1 = 0; k1 < kend; k1 += 16) {              // kend is a multiple of 32; 8 steps' LDS reads in flight together
        float a0[8], a1[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = k1 + 2 * u + kk;
            a0[u] = As[r * 129 + k]; a1[u] = As[(32 + r) * 129 + k]; b[u] = Cs[k * GH_TS + n0 + r];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b[u], acc1, 0, 0, 0);
        }
    }
    __syncthreads();                                     // As is dead: reuse as the transpose buffer
    KT();
    const int gc = a.cb + n0 + r;
    const bool in_leaf = gc >= a.c0 && gc < a.c1;
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int lm = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            const float v = mt == 0 ? acc0[e] : acc1[e];
            const int row = row0 + lm;
            if (in_leaf && row < a.mrows) {
                a.A[(long)row * a.lda + gc] = v;
                a.Vh[(long)row * a.ldvh + gc] = (half_t)v;
            }
            Ts[(n0 + r) * 72 + lm] = (half_t)v;
        }
    }
    __syncthreads();
    KT();
    if (Sp) gh_partial_gram_add(Ts, ga, lane, wave);
    KT();
    // V^T rows: 128 columns x 64 rows of this workgroup, 16-B chunks along the row index
    for (int e = tid; e < GW * 8; e += 256) {
        const int c = e >> 3, ch = e & 7;
        const int gcol = a.cb + c;
        if (gcol >= a.c0 && gcol < a.c1) {
            half_t* dst = a.Vt + (long)gcol * a.ldvt + row0 + 8 * ch;
            if ((row0 & 7) == 0 && row0 + 8 * ch + 8 <= a.mrows) {
                *(uint4*)dst = *(const uint4*)&Ts[c * 72 + 8 * ch];
            } else {
                for (int q = 0; q < 8; q++)
                    if (row0 + 8 * ch + q < a.mrows) dst[q] = Ts[c * 72 + 8 * ch + q];
            }
        }
    }
    }                                                      // row blocks
    if (Sp) gh_partial_gram_store(ga, Sp + (long)blockIdx.x * (GW * GW), lane, wave);
    KT(); KT_DUMP(6, "gh_apply load|mfma|store+next|gram|vt");
}

// S = sum of nslab fp32 partials (128 x 128), fixed order; same scheme as gh_reduce_kernel
__global__ __launch_bounds__(256) void gh_reduce_f32_kernel(const float* __restrict__ Sp, int nslab, float* __restrict__ S) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const bool act = ((e >> 7) >> 5) <= ((e & 127) >> 5);   // the partials hold the 10 upper 32 x 32 tiles only; T reads j >= i
    float s = 0.f;
    if (act) {
        // 16 loads in flight per lane (the kernel is pure memory latency, and that latency grows several times while a far
        // update streams beside the chain), summed in a fixed order: four interleaved chains as before
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int q = wave;
        for (; q + 60 < nslab; q += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = Sp[(long)(q + 4 * u) * (GW * GW) + e];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
        }
        {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = (q + 4 * u < nslab) ? Sp[(long)(q + 4 * u) * (GW * GW) + e] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
        }
        s = (s0 + s1) + (s2 + s3);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0) S[e] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// blocks of 64 rows per gh_apply workgroup: tall leaves take 2 / 4, which still leaves >= 256 workgroups
static int gh_apply_iters(const LeafArgs& a) {
    const int rows = a.mrows - a.c1;
    return rows >= 49152 ? 4 : rows >= 24576 ? 2 : 1;
}
int gh_num_partials(const LeafArgs& a) {
    const int it = gh_apply_iters(a);
    return ((a.mrows - a.c1 + 63) / 64 + it - 1) / it + (a.c1 - a.c0 + 63) / 64;
}
void launch_gh_reduce_f32(const float* Sp, int nslab, float* S, hipStream_t s) {
    hipLaunchKernelGGL(gh_reduce_f32_kernel, dim3(256), dim3(256), 0, s, Sp, nslab, S);
}
static void gh_set_attrs() {
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)gh_gram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GH_ROWS * GH_TD * 8));
                         MPQR_IGNORE(hipFuncSetAttribute((const void*)gh_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (8256 + GW * GH_TS) * 4)));
}
// the four steps of a Gram-Householder leaf, separately launchable (the look-ahead schedule puts them on different streams)
void launch_gh_gram(const LeafArgs& a, double* Gp, double* G, hipStream_t s) {
    gh_set_attrs();
    const int rows = a.mrows - a.c0;
    const int it = rows >= 49152 ? 4 : rows >= 24576 ? 2 : 1;          // row blocks per workgroup pair: tall leaves still get >= 192 workgroups
    const int nwg = ((rows + GH_ROWS - 1) / GH_ROWS + it - 1) / it;
    hipLaunchKernelGGL(gh_gram_kernel, dim3(2 * nwg), dim3(256), GH_ROWS * GH_TD * 8, s, a, Gp, it);
    hipLaunchKernelGGL(gh_reduce_kernel, dim3(256), dim3(256), 0, s, Gp, nwg, G);
}
__global__ __launch_bounds__(64) void publish_word_kernel(int* flag, int word, int value) {
    if (threadIdx.x == 0) __hip_atomic_store(flag + word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
void launch_publish_word(int* flag, int word, int value, hipStream_t s) {
    hipLaunchKernelGGL(publish_word_kernel, dim3(1), dim3(64), 0, s, flag, word, value);
}
void launch_gh_solve(const LeafArgs& a, const double* G, float* Cv, int* flag, hipStream_t s, const SolveWait* ws) {
    gh_set_attrs();
    static const int dbg_skip = []() { const char* e = getenv("MPQR_DBG_NOSOLVE"); return e ? atoi(e) : 0; }();
    if (dbg_skip) return;                                  // timing experiment only (results are garbage): is the chain host-bound?
    // gh_solve3 (kernels_solve.hip): the recursion blocked by 16, chains in single waves.  (Round 2's step-by-step kernel, one
    // barrier per reflector, 104 us against 65, was kept behind MPQR_SOLVE3=0 for a round and is gone: git history.)
    launch_gh_solve3(a, G, Cv, flag, s, ws);
}
void launch_gh_apply(const LeafArgs& a, const float* Cv, float* Sp, hipStream_t s) {
    gh_set_attrs();
    const int it = gh_apply_iters(a);
    const int nlow = ((a.mrows - a.c1 + 63) / 64 + it - 1) / it;
    const int ntop = Sp ? (a.c1 - a.c0 + 63) / 64 : 0;      // extra workgroups: Gram contribution of the top block
    if (nlow + ntop == 0) return;
    hipLaunchKernelGGL(gh_apply_kernel, dim3(nlow + ntop), dim3(256), (8256 + GW * GH_TS) * 4, s, a, Cv, Sp, nlow, it);
}
void launch_leaf_gram_householder(const LeafArgs& a, double* Gp, double* G, float* Cv, int* flag, float* Sp, float* S,
                                  hipStream_t s) {
    launch_gh_gram(a, Gp, G, s);
    launch_gh_solve(a, G, Cv, flag, s);
    launch_gh_apply(a, Cv, Sp, s);
    if (Sp && S) launch_gh_reduce_f32(Sp, gh_num_partials(a), S, s);
}

// ------------------------------------------------------------------ the last leaf of a (nearly) square matrix: <= 128 rows left
// Columns [c0, c1) with at most 128 rows from row c0 down (m - c0 <= 128): no tall part, so no Gram-Householder (its Cholesky recursion
// needs >= w rows below the top block), and as 32-column leaf_wg leaves it cost four launches of ~52 us plus their in-leaf updates and
// three levels of T merges -- ~1 ms between the last tall leaf and the start of Q formation (kernel trace, 2048^2 and 16384^2 alike).
// Here: the rows x w block in the registers of one workgroup (thread = column j x group of 16 rows), the same Householder arithmetic as
// leaf_wg_kernel (u, alpha = sgn(u0) ||u||, v = (u + alpha e1) / ||u + alpha e1||, w_j = 2 v^T a_j), two barriers per column:
//   A  the owners of column k publish it (ucol), the owners of row k publish that (rowk)          | barrier
//   B  every thread: partial u^T a_j over its 16 rows -> part[g][j]                                | barrier
//   C  every thread: sums the partials of column j and of column k, forms alpha / inv / w_j itself, updates its 16 entries
// Row groups above row k have nothing left to do and only keep the barriers.  Outputs as a Gram-Householder leaf's: R and V in A, the
// fp16 copies V / V^T, vdiag, and S = V^T V of the fp16-ROUNDED reflectors (window coordinates, exact-f32 MFMA out of LDS) for t_panel.
__global__ __launch_bounds__(1024) void leaf_tail_kernel(LeafArgs a, float* __restrict__ Sout) {
    float* Vs = (float*)gh_smem;                               // [TP][TPS]: the rounded reflectors as fp32, row-major
    __shared__ __attribute__((aligned(16))) float ucol[TP];
    __shared__ float part[8][TP], rowk[2][TP], vdl[TP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = tid & 127;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 7);      // rows 16 g .. 16 g + 15: the same for a whole wave
    const int w = a.c1 - a.c0, rows = a.mrows - a.c0, off = a.c0 - a.cb;
    float c[16];
#pragma unroll
    for (int i = 0; i < 16; i++)                                 // unconditional loads (clamped), masked below: all 16 in flight together
        c[i] = a.A[(long)(a.c0 + min(16 * g + i, rows - 1)) * a.lda + a.c0 + min(j, w - 1)];
#pragma unroll
    for (int q = 0; q < TP * TP / 1024; q++) Sout[tid + 1024 * q] = 0.f;      // the window outside the leaf stays zero
    if (tid < TP) vdl[tid] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) c[i] = (16 * g + i < rows && j < w) ? c[i] : 0.f;
    for (int k = 0; k < w; k++) {
        const int par = k & 1, gk = k >> 4, ik = k & 15;
        if (g >= gk) {
            if (j == k) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float4 t;
                    t.x = (16 * g + 4 * q + 0 >= k) ? c[4 * q + 0] : 0.f; t.y = (16 * g + 4 * q + 1 >= k) ? c[4 * q + 1] : 0.f;
                    t.z = (16 * g + 4 * q + 2 >= k) ? c[4 * q + 2] : 0.f; t.w = (16 * g + 4 * q + 3 >= k) ? c[4 * q + 3] : 0.f;
                    *(float4*)&ucol[16 * g + 4 * q] = t;
                }
            }
            if (g == gk) {
                float rk = c[0];
#pragma unroll
                for (int i = 1; i < 16; i++) rk = (ik == i) ? c[i] : rk;
                rowk[par][j] = rk;
            }
        }
        __syncthreads();
        float u[16], p = 0.f;
        if (g >= gk) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 t = *(const float4*)&ucol[16 * g + 4 * q];
                u[4 * q] = t.x; u[4 * q + 1] = t.y; u[4 * q + 2] = t.z; u[4 * q + 3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) p += u[i] * c[i];
        }
        part[g][j] = p;
        __syncthreads();
        if (g >= gk) {
            float sj = 0.f, sk = 0.f;
#pragma unroll
            for (int q = 0; q < 8; q++) { sj += part[q][j]; sk += part[q][k]; }
            const float akk = rowk[par][k], rj = rowk[par][j];
            float alpha = 0.f, inv = 0.f;
            if (sk != 0.f) {
                const float nu = sqrtf(sk);
                alpha = (akk >= 0.f) ? nu : -nu;
                inv = 1.0f / sqrtf(2.0f * (sk + fabsf(akk) * nu));
            }
            if (g == gk) {                                       // u + alpha e1: row k is one of this group's
#pragma unroll
                for (int i = 0; i < 16; i++) u[i] += (ik == i) ? alpha : 0.f;
            }
            const float wj = (j > k && j < w) ? 2.0f * (sj + alpha * rj) * inv : 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float vi = u[i] * inv;                     // zero above row k (ucol is)
                c[i] -= vi * wj;
                if (j == k && sk != 0.f && 16 * g + i >= k) c[i] = (16 * g + i == k) ? -alpha : vi;
                if (j == k && 16 * g + i == k) vdl[k] = vi;
            }
        }
    }
    __syncthreads();
    // outputs: A (R on and above the diagonal, the reflectors below), fp16 V / V^T with the diagonal entries from vdl, Vs for S
    {
        const float vd = j < w ? vdl[j] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int r = 16 * g + i;
            half_t hv = (half_t)0.f;
            if (r < rows && j < w) {
                a.A[(long)(a.c0 + r) * a.lda + a.c0 + j] = c[i];
                if (r >= j) {
                    hv = (half_t)(r == j ? vd : c[i]);
                    a.Vh[(long)(a.c0 + r) * a.ldvh + a.c0 + j] = hv;
                    a.Vt[(long)(a.c0 + j) * a.ldvt + a.c0 + r] = hv;
                }
            }
            Vs[r * TPS + j] = (float)hv;
        }
        if (g == 0 && j < w) a.vdiag[a.c0 + j] = vd;
    }
    __syncthreads();
    // S[i][j] = sum_r V[r][i] V[r][j], upper 32 x 32 tiles (wave t: tile (t / 4, t % 4)); both operands are read along rows of Vs
    {
        const int ti = wave >> 2, tj = wave & 3;
        if (tj >= ti && 32 * ti < w && 32 * tj < w) {
            const int r = lane & 31, kk = lane >> 5;
            floatx16p acc;
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e] = 0.f;
            for (int k1 = 0; k1 < rows; k1 += 16) {              // (rows beyond `rows` are zero in Vs)
                float av[8], bv[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { av[q] = Vs[(k1 + 2 * q + kk) * TPS + 32 * ti + r]; bv[q] = Vs[(k1 + 2 * q + kk) * TPS + 32 * tj + r]; }
#pragma unroll
                for (int q = 0; q < 8; q++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * kk, jj = 32 * tj + r;
                if (i < w && jj < w) Sout[(long)(off + i) * TP + off + jj] = acc[e];
            }
        }
    }
}

void launch_leaf_tail(const LeafArgs& a, float* S, hipStream_t s) {
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)leaf_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TP * TPS * 4)));
    hipLaunchKernelGGL(leaf_tail_kernel, dim3(1), dim3(1024), TP * TPS * 4, s, a, S);
}

void launch_leaf_factor(const LeafArgs& a, hipStream_t s) {
    if (a.c1 <= a.c0) return;
    const int rows = a.mrows - a.c0;
    if (rows <= 2048) {
        if (rows <= 128) hipLaunchKernelGGL(leaf_wg_kernel<1>, dim3(1), dim3(1024), 0, s, a);
        else if (rows <= 256) hipLaunchKernelGGL(leaf_wg_kernel<2>, dim3(1), dim3(1024), 0, s, a);
        else if (rows <= 512) hipLaunchKernelGGL(leaf_wg_kernel<4>, dim3(1), dim3(1024), 0, s, a);
        else if (rows <= 1024) hipLaunchKernelGGL(leaf_wg_kernel<8>, dim3(1), dim3(1024), 0, s, a);
        else hipLaunchKernelGGL(leaf_wg_kernel<16>, dim3(1), dim3(1024), 0, s, a);
        return;
    }
    float* P0 = a.P;
    float* P1 = a.P + (long)a.maxwg * 32;
    int nwg = (a.mrows - a.c0 + RPW - 1) / RPW;
    if (nwg < 1) nwg = 1;
    hipLaunchKernelGGL(leaf_init_kernel, dim3(nwg), dim3(256), 0, s, a, a.c0, P0);
    int nwg_in = nwg;
    for (int k = a.c0; k < a.c1; k++) {
        int grid = (a.mrows - k + RPW - 1) / RPW;
        if (grid < 1) grid = 1;
        const bool even = ((k - a.c0) & 1) == 0;
        hipLaunchKernelGGL(leaf_step_kernel, dim3(grid), dim3(256), 0, s, a, k, nwg_in, even ? P0 : P1, even ? P1 : P0);
        nwg_in = grid;
    }
}

// ------------------------------------------------------------------ Y = fp16( (sum of X slabs) * T' ) for ONE leaf (128 reflectors)
// The in-block update of a leaf is  X = C2^T V (split-K slabs),  Y = X T',  C2 -= V Y^T.  This kernel replaces the slab sum
// (a launch that re-reads every slab) and the small GEMM behind it: workgroup b sums the slabs of 16 rows of X in slab
// order, rounds to fp16 as the GEMM's staging did, and multiplies by T' (fp16, Bt[n][k] = T'[k][n], upper triangular in
// (k, n): k <= n) on v_mfma_f32_16x16x16_f16; 4 waves x 2 column tiles.  Two launches and ~20 us less per leaf.
typedef half_t half4x __attribute__((ext_vector_type(4)));
typedef float float4x __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void leaf_xt_kernel(const float* __restrict__ X, int nslab, long slab_stride, int M1,
                                                      const half_t* __restrict__ Bt, long ldb, int tri,
                                                      half_t* __restrict__ Y, long ldy, const float* __restrict__ cscale, long cs_ld,
                                                      int* pub_flag, int pub_value) {
    // pub_flag: tell the T stream that everything BEFORE this kernel in the chain stream (the leaf's reflectors, its T) is complete and
    // visible -- it is, or this kernel would not have started.  wait_flag_kernel polls the word: an event record here would cost the
    // chain stream 3 - 4.5 us per leaf (tools/probe_graph_handoff.hip), this store costs it nothing.
    if (pub_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(pub_flag, pub_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ __attribute__((aligned(16))) half_t Xs[16][136];       // fp16(X)
    __shared__ __attribute__((aligned(16))) half_t Xl[16][136];       // X - fp16(X): the product keeps X to ~22 bits
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 16;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int e4 = tid + 256 * q, lr = e4 >> 5, c = (e4 & 31) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + lr < M1) {
            const float* p = X + (long)(row0 + lr) * 128 + c;
            int sl = 0;
            for (; sl + 8 <= nslab; sl += 8) {             // 8 loads in flight, summed in slab order
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = *(const float4*)(p + (long)(sl + u) * slab_stride);
#pragma unroll
                for (int u = 0; u < 8; u++) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
            }
            for (; sl < nslab; sl++) {
                const float4 v = *(const float4*)(p + (long)sl * slab_stride);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
        half4x hv; hv[0] = (half_t)a.x; hv[1] = (half_t)a.y; hv[2] = (half_t)a.z; hv[3] = (half_t)a.w;
        *(half4x*)&Xs[lr][c] = hv;
        half4x lv; lv[0] = (half_t)(a.x - (float)hv[0]); lv[1] = (half_t)(a.y - (float)hv[1]);
        lv[2] = (half_t)(a.z - (float)hv[2]); lv[3] = (half_t)(a.w - (float)hv[3]);
        *(half4x*)&Xl[lr][c] = lv;
    }
    __syncthreads();
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int n0 = 32 * wave + 16 * t;
        float4x acc = {0.f, 0.f, 0.f, 0.f};
        // tri == 2: Bt[n][k] = 0 for k > n (T^T as stored: rows end at the diagonal); tri == 1: zero for k < n
        const int klo = tri == 1 ? (n0 & ~15) : 0, khi = tri == 2 ? n0 + 16 : 128;
        for (int k0 = klo; k0 < khi; k0 += 16) {
            const half4x av = *(const half4x*)&Xs[li][k0 + 4 * lg];
            const half4x al = *(const half4x*)&Xl[li][k0 + 4 * lg];
            const half4x bv = *(const half4x*)&Bt[(long)(n0 + li) * ldb + k0 + 4 * lg];
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(av, bv, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bv, acc, 0, 0, 0);
        }
        const float sc = cscale ? cscale[(long)(n0 + li) * cs_ld] : 1.f;      // tau_n (the fp16 T has a unit diagonal)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int row = row0 + 4 * lg + e;
            if (row < M1) Y[(long)row * ldy + n0 + li] = (half_t)(sc * acc[e]);
        }
    }
}
void launch_leaf_xt(const float* X, int nslab, long slab_stride, int M1, const half_t* Bt, long ldb, int tri, half_t* Y, long ldy,
                    const float* cscale, long cscale_ld, hipStream_t s, int* pub_flag, int pub_value) {
    hipLaunchKernelGGL(leaf_xt_kernel, dim3((M1 + 15) / 16), dim3(256), 0, s, X, nslab, slab_stride, M1, Bt, ldb, tri, Y, ldy,
                       cscale, cscale_ld, pub_flag, pub_value);
}

// One wave waits until *flag >= value (published by a kernel of ANOTHER stream: leaf_xt_kernel), then the stream goes on: a cross-stream
// dependency that costs the publishing stream nothing.  The kernels behind this one start after it has ended, i.e. with the usual
// start-of-kernel acquire, so they see what the publisher's predecessors wrote.  The values of a handle only grow (a restarted pass
// publishes larger ones), so a stale word can only end the wait early for work whose inputs are older still.  Exit condition every
// wave reaches: the publisher is always enqueued BEFORE the waiter (host order); should it never run, the wait gives up after `ticks`
// (100 MHz; MPQR_TPOLL_TIMEOUT_MS, default 5 s: a slow chain -- a shared GPU, a serialising tool, other ranks' blocks in front of an unpack --
// is not a lost publish) and raises *timeout_word (mapped host memory) and flag[1]; later waiters see flag[1] and return at once.  mpqr_factor
// then repeats the factorisation with event hand-offs; the mpqr_dist_* steps report the error.
__global__ __launch_bounds__(64) void wait_flag_kernel(int* __restrict__ flag, int word, int value, int* __restrict__ timeout_word, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    if (__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;      // an earlier wait of this pass has timed out
    int* const marker = flag + 1;
    flag += word;                                            // (0: the chain is past the leaf's T; 2: past the leaf's reflectors)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) {
            __hip_atomic_store(timeout_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(marker, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}
void launch_wait_flag(int* flag, int value, int* timeout_word, unsigned long long ticks, hipStream_t s, int word) {
    hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, s, flag, word, value, timeout_word, ticks);
}

// ------------------------------------------------------------------ T of a leaf (up to 128 reflectors)
// T^{-1} = striu(V^T V) + diag(V^T V)/2  (compact WY with H_i = I - (2/v_i^T v_i) v_i v_i^T).  S is the Gram matrix
// of the fp16-ROUNDED reflectors, which keeps I - V T V^T orthogonal for the V the MFMA GEMMs actually use.
// One workgroup, everything in LDS: the four 32 x 32 diagonal blocks by the column recurrence
// (T_ii = 2/S_ii, T[:i,i] = -T_ii T[:i,:i] S[:i,i]; row a depends on row a only, so lane a runs it in registers),
// then two levels of  T_LR = -T_L (S_LR T_R) on the exact-f32 MFMA.  Indices >= w are padded with T = I, which decouples.
// Replaces the reference's dev_wy_transform loop (Cuda/qr.cu:535-600: r rounds of three kernels, dense (m-l)^2).
// NT threads (>= 256: tri_inverse_128 needs four waves).  SC1: S was written by OTHER workgroups of the same launch (leaf_mid_kernel): every
// load of it is an agent-scope relaxed atomic load (global_load_dword sc1: served past this CU's L1 and this XCD's L2 hit path as the
// hand-off table of MI355X_MICROARCH.md requires when the producer stored sc1).
// (two halves since round 5: the fused leaf computes Y = X T' out of LDS between the inverse and the stores of T)
template <int NT, bool SC1>
__device__ __forceinline__ void t_panel_load_invert(const float* __restrict__ S, int nslab, long slab_stride, int lds_, int a0, int c0, int c1,
                                                    float* Ss, float* tdiag) {
    float* Ts = Ss + TP * TPS;               // [TP][TPS]
    const int tid = threadIdx.x;
    const int w = c1 - c0, off = c0 - a0;
    const int nblk = (w + 31) / 32;          // active 32-blocks
    KT_DECL; KT();
    // the loads of a thread first (unconditional: clamped addresses, masked afterwards), THEN the LDS stores: with the loads inside the
    // branches the compiler kept load -> use in program order and the kernel's load phase took 16.8 k cycles for 64 KB (one miss
    // latency per load: S was just written by gh_reduce_f32 on other CUs) -- in-kernel stamps, tools/ktrace_solve.sh
    constexpr int NQ = TP * TP / NT;
    float sv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int e = tid + NT * q;
        const int i = min(e >> 7, w - 1), j = min(e & 127, w - 1);
        const float* p = &S[(long)(off + i) * lds_ + off + j];
        if (SC1) sv[q] = __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        else sv[q] = *p;
    }
    if (nslab > 1) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int e = tid + NT * q;
            const int i = min(e >> 7, w - 1), j = min(e & 127, w - 1);
            for (int sl = 1; sl < nslab; sl++) {
                const float* p = &S[(long)sl * slab_stride + (long)(off + i) * lds_ + off + j];
                sv[q] += SC1 ? __builtin_bit_cast(float, __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : *p;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int e = tid + NT * q;
        const int i = e >> 7, j = e & 127;
        float v = (i < w && j < w && j >= i) ? sv[q] : 0.f;
        if (i == j) { tdiag[i] = (i < w) ? (v > 0.f ? 2.0f / v : 0.f) : 1.f; v = 0.f; }
        Ss[i * TPS + j] = v;
        Ts[i * TPS + j] = 0.f;
    }
    __syncthreads();
    KT();
    tri_inverse_128(Ss, tdiag, Ts, nblk, tid);
    KT(); KT_DUMP(1, "t_panel load|inverse");
}
template <int NT>
__device__ __forceinline__ void t_panel_store(const float* Ts, int a0, int c0, int c1, float* __restrict__ T, half_t* __restrict__ Th,
                                              half_t* __restrict__ Tth, int ldt, int ld) {
    const int tid = threadIdx.x;
    const int w = c1 - c0, off = c0 - a0;
    // the ldt x ldt aligned range is written (zeros outside the leaf) into a matrix of leading dimension ld (>= ldt:
    // the node's own T, or its diagonal block inside the T of the enclosing top-level block)
    if (ldt == TP && off == 0 && (ld & 3) == 0) {
        // the usual case (a full 128-column leaf): four consecutive entries per thread, 16- and 8-byte stores
        typedef half_t half4t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = 0; q < TP * TP / 4 / NT; q++) {
            const int e4 = tid + NT * q, i = e4 >> 5, j = (e4 & 31) * 4;
            float4 v; half4t hv, ht;
            float t[4], u[4];
            const float tii = i < w ? Ts[i * TPS + i] : 0.f;
            const float rti = tii != 0.f ? 1.0f / tii : 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                t[c] = (i < w && j + c >= i && j + c < w) ? Ts[i * TPS + j + c] : 0.f;          // T[i][j+c]
                u[c] = (j + c < w && i >= j + c && i < w) ? Ts[(j + c) * TPS + i] : 0.f;        // T^T[i][j+c] = T[j+c][i]
                hv[c] = (half_t)(t[c] * rti); ht[c] = (half_t)(u[c] * rti);                    // fp16 copies: row i / tau_i
            }
            v.x = t[0]; v.y = t[1]; v.z = t[2]; v.w = t[3];
            *(float4*)&T[(long)i * ld + j] = v;
            *(half4t*)&Th[(long)i * ld + j] = hv;
            *(half4t*)&Tth[(long)i * ld + j] = ht;
        }
    } else {
    for (int e = tid; e < ldt * ldt; e += NT) {
        const int i = e / ldt, j = e % ldt;
        const int li = i - off, lj = j - off;
        float v = 0.f;
        if (li >= 0 && li < w && lj >= li && lj < w) v = Ts[li * TPS + lj];
        T[(long)i * ld + j] = v;
        const float tii = (li >= 0 && li < w) ? Ts[li * TPS + li] : 0.f;
        Th[(long)i * ld + j] = (half_t)(tii != 0.f ? v / tii : 0.f);       // fp16 copies carry T[n][k] / tau_n (unit diagonal)
    }
    for (int e = tid; e < ldt * ldt; e += NT) {          // T^T: consecutive lanes walk a column of T (odd LDS stride)
        const int j = e / ldt, i = e % ldt;
        const int li = i - off, lj = j - off;
        float v = 0.f;
        if (li >= 0 && li < w && lj >= li && lj < w) v = Ts[li * TPS + lj];
        const float tjj = (lj >= 0 && lj < w) ? Ts[lj * TPS + lj] : 0.f;
        Tth[(long)j * ld + i] = (half_t)(tjj != 0.f ? v / tjj : 0.f);
    }
    }
}
template <int NT, bool SC1>
__device__ __forceinline__ void t_panel_body(const float* __restrict__ S, int nslab, long slab_stride, int lds_, int a0, int c0, int c1,
                                             float* __restrict__ T, half_t* __restrict__ Th, half_t* __restrict__ Tth, int ldt, int ld,
                                             float* Ss, float* tdiag) {
    t_panel_load_invert<NT, SC1>(S, nslab, slab_stride, lds_, a0, c0, c1, Ss, tdiag);
    t_panel_store<NT>(Ss + TP * TPS, a0, c0, c1, T, Th, Tth, ldt, ld);
}
__global__ __launch_bounds__(1024) void t_panel_kernel(const float* __restrict__ S, int nslab, long slab_stride, int lds_,
                                                       int a0, int c0, int c1, float* __restrict__ T,
                                                       half_t* __restrict__ Th, half_t* __restrict__ Tth, int ldt, int ld) {
    __shared__ float tdiag[TP];
    t_panel_body<1024, false>(S, nslab, slab_stride, lds_, a0, c0, c1, T, Th, Tth, ldt, ld, (float*)gh_smem, tdiag);
}

// T_LR = -T_L (S T_R) for two children of at most 128 reflectors each, one workgroup, products on the exact-f32 MFMA
// out of LDS (S, T_R, then X = S T_R over S and T_L over T_R).  Replaces two small GEMM launches per merge.
__global__ __launch_bounds__(1024) void t_merge_kernel(const float* __restrict__ S, int ldl, int ldr,
                                                       const float* __restrict__ TL, const float* __restrict__ TR,
                                                       float* __restrict__ TLR) {
    float* Ss = (float*)gh_smem;             // [TP][TPS]: S, then X
    float* Tr = Ss + TP * TPS;               // [TP][TPS]: T_R, then T_L
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    KT_DECL; KT();
#pragma unroll
    for (int q = 0; q < TP * TP / 1024; q++) {
        const int e = tid + 1024 * q, i = e >> 7, j = e & 127;
        Ss[i * TPS + j] = (i < ldl && j < ldr) ? S[(long)i * ldr + j] : 0.f;
        Tr[i * TPS + j] = (i < ldr && j < ldr) ? TR[(long)i * ldr + j] : 0.f;
    }
    __syncthreads();
    KT();
    const int bi = wave >> 2, bj = wave & 3;
    const bool has = 32 * bi < ldl && 32 * bj < ldr;
    floatx16p acc;
    if (has) acc = lds_mm32(&Ss[32 * bi * TPS], TPS, &Tr[32 * bj], TPS, 0, min(ldr, 32 * (bj + 1)), lane);   // T_R upper
    KT();
    __syncthreads();
    if (has) lds_store32(&Ss[32 * bi * TPS + 32 * bj], TPS, acc, 1.f, lane);
#pragma unroll
    for (int q = 0; q < TP * TP / 1024; q++) {
        const int e = tid + 1024 * q, i = e >> 7, j = e & 127;
        Tr[i * TPS + j] = (i < ldl && j < ldl) ? TL[(long)i * ldl + j] : 0.f;
    }
    __syncthreads();
    KT();
    if (has) {
        acc = lds_mm32(&Tr[32 * bi * TPS], TPS, &Ss[32 * bj], TPS, 32 * bi, ldl, lane);                          // T_L upper
        const int r = lane & 31, kk = lane >> 5;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int i = 32 * bi + (e & 3) + 8 * (e >> 2) + 4 * kk, j = 32 * bj + r;
            if (i < ldl && j < ldr) TLR[(long)i * ldr + j] = -acc[e];
        }
    }
    KT(); KT_DUMP(0, "t_merge load|mm1|store+loadTL|mm2+out");
}

void launch_t_merge(const float* S, int ldl, int ldr, const float* TL, const float* TR, float* TLR, hipStream_t s) {
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)t_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TP * TPS * 4)));
    hipLaunchKernelGGL(t_merge_kernel, dim3(1), dim3(1024), 2 * TP * TPS * 4, s, S, ldl, ldr, TL, TR, TLR);
}

// Back substitution, one diagonal block: X[k0:k0+kb] = R[k0:k0+kb, k0:k0+kb]^-1 Y[k0:k0+kb]  (kb <= 128, R upper
// triangular fp32 in the working matrix, Y in place, nrhs columns).  The block is inverted with tri_inverse_128
// (a zero diagonal entry gives a zero row / column: x_i = 0) and applied as a small product out of LDS.
__global__ __launch_bounds__(1024) void trsm_diag_kernel(const float* __restrict__ R, long ldr, int k0, int kb,
                                                         float* __restrict__ Y, long ldy, int nrhs) {
    float* Ss = (float*)gh_smem;             // [TP][TPS]
    float* Ts = Ss + TP * TPS;               // [TP][TPS]
    __shared__ float tdiag[TP];
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < TP * TP / 1024; q++) {
        const int e = tid + 1024 * q, i = e >> 7, j = e & 127;
        float v = 0.f;
        if (i < kb && j < kb && j >= i) v = R[(long)(k0 + i) * ldr + k0 + j];
        if (i == j) { tdiag[i] = (i < kb) ? (v != 0.f ? 1.0f / v : 0.f) : 1.f; v = 0.f; }
        Ss[i * TPS + j] = v;
        Ts[i * TPS + j] = 0.f;
    }
    __syncthreads();
    tri_inverse_128(Ss, tdiag, Ts, (kb + 31) / 32, tid);
    // stage Y_k (kb x nrhs) in Ss (dead now), then X = Ts * Y
    for (int c0 = 0; c0 < nrhs; c0 += 128) {
        const int nc = min(128, nrhs - c0);
        for (int e = tid; e < kb * nc; e += 1024) { const int i = e / nc, c = e % nc; Ss[i * TPS + c] = Y[(long)(k0 + i) * ldy + c0 + c]; }
        __syncthreads();
        for (int e = tid; e < kb * nc; e += 1024) {
            const int i = e / nc, c = e % nc;
            float x = 0.f;
            for (int j = i; j < kb; j++) x = fmaf(Ts[i * TPS + j], Ss[j * TPS + c], x);
            Y[(long)(k0 + i) * ldy + c0 + c] = x;
        }
        __syncthreads();
    }
}
void launch_trsm_diag(const float* R, long ldr, int k0, int kb, float* Y, long ldy, int nrhs, hipStream_t s) {
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)trsm_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TP * TPS * 4)));
    hipLaunchKernelGGL(trsm_diag_kernel, dim3(1), dim3(1024), 2 * TP * TPS * 4, s, R, ldr, k0, kb, Y, ldy, nrhs);
}

void launch_t_leaf(const float* S, int nslab, long slab_stride, int lds_, int a0, int c0, int c1, float* T, half_t* Th,
                   half_t* Tth, int ldt, hipStream_t s, int ld) {
    if (ld <= 0) ld = ldt;
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)t_panel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TP * TPS * 4)));
    hipLaunchKernelGGL(t_panel_kernel, dim3(1), dim3(1024), 2 * TP * TPS * 4, s, S, nslab, slab_stride, lds_, a0, c0, c1, T, Th,
                       Tth, ldt, ld);
}

// ------------------------------------------------------------------ the middle of a Gram-Householder leaf in ONE launch
// Between gh_apply and leaf_xt a leaf needs two independent things: X = C2^T V_j (a split-K GEMM over all rows, ~16 us) and T_j (sum of
// gh_apply's partial Gram matrices, ~6 us, then the triangular inverse, ~16 us in one workgroup).  As launches on two streams they cost
// two cross-stream hand-offs on the critical path (chain -> side stream ~7 us, side stream -> chain ~10-14 us: kernel trace, and
// tools/probe_graph_handoff.hip); as launches on one stream they serialise.  Here they are ONE grid: the first MID_RB workgroups sum the
// partials, the others compute GEMM tiles (gemm_body.h); the reduce workgroups sum (four consecutive entries per lane, the same four interleaved chains per
// entry as gh_reduce_f32_kernel, so S is bit-identical to it for leaves of < 32768 rows; taller ones cut the partials into up to four ranges) and the one whose arrival is last runs the T kernel's body on 256 threads.
// Hand-off inside the launch (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the hand-off table): S is stored sc1, every
// storing wave waits vmcnt(0), one lane per workgroup adds to an agent-scope counter behind a workgroup barrier, the workgroup whose add
// returned MID_RB - 1 loads S with sc1 loads behind another barrier.  Nothing spins: no residency requirement.  One workgroup per CU (LDS).
constexpr int MID_RB = 64;
struct LeafMidArgs {
    GemmArgs g; int gx, nX;                                  // X tiles along M; GEMM workgroups = gx * nsplit
    const float* Sp; int nslab; float* S; int sh;            // partials -> S (128 x 128 window); T reads it from (sh, sh)
    int ngrp;                                                // the partials are cut into ngrp ranges (64 workgroups each -> S + grp * 16384), summed by the T body
    int* counter;                                            // zero between launches (the last arriver resets it)
    int a0, c0, c1; float* T; half_t* Th; half_t* Tth; int ldt, ld;
};
__global__ __launch_bounds__(256) void leaf_mid_kernel(LeafMidArgs m) {
    const int nrb = MID_RB * m.ngrp;
    if ((int)blockIdx.x >= nrb) {                            // (the reduce workgroups come first in the grid: theirs is the longer path)
        const int b = (int)blockIdx.x - nrb;
        gemm_f16_body<A_F32T, E_STORE_F32>(m.g, (half_t*)gh_smem, b % m.gx, 0, b / m.gx);
        return;
    }
    __shared__ float4 part[4][64];
    __shared__ float tdiag[TP];
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = (int)blockIdx.x / MID_RB;                // tall leaves: gh_apply leaves rows / 64 partials (1026 at 65536 rows, 67 MB) -- cut into
    const int q_lo = (int)((long)m.nslab * grp / m.ngrp), q_hi = (int)((long)m.nslab * (grp + 1) / m.ngrp);   // ranges of <= ~256
    const int e = (((int)blockIdx.x % MID_RB) * 64 + lane) * 4;   // four consecutive entries of one row of the window
    const bool act = ((e >> 7) >> 5) <= ((e & 127) >> 5);     // the partials hold the 10 upper 32 x 32 tiles only; T reads j >= i
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
        float4 s0 = s, s1 = s, s2 = s, s3 = s;
        auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
        int q = q_lo + wave;
        for (; q + 60 < q_hi; q += 64) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = *(const float4*)&m.Sp[(long)(q + 4 * u) * (GW * GW) + e];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { add(s0, v[u]); add(s1, v[u + 1]); add(s2, v[u + 2]); add(s3, v[u + 3]); }
        }
        {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = *(const float4*)&m.Sp[(long)min(q + 4 * u, q_hi - 1) * (GW * GW) + e];
#pragma unroll
            for (int u = 0; u < 16; u++) if (q + 4 * u >= q_hi) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 16; u += 4) { add(s0, v[u]); add(s1, v[u + 1]); add(s2, v[u + 2]); add(s3, v[u + 3]); }
        }
        s.x = (s0.x + s1.x) + (s2.x + s3.x); s.y = (s0.y + s1.y) + (s2.y + s3.y);
        s.z = (s0.z + s1.z) + (s2.z + s3.z); s.w = (s0.w + s1.w) + (s2.w + s3.w);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        const float4 p0 = part[0][lane], p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
        const float r4[4] = {(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w)};
#pragma unroll
        for (int c = 0; c < 4; c++)
            __hip_atomic_store((unsigned*)&m.S[(long)grp * (GW * GW) + e + c], __builtin_bit_cast(unsigned, r4[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have left before this workgroup signals
    }
    __syncthreads();
    if (tid == 0) is_last = __hip_atomic_fetch_add(m.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nrb - 1;
    __syncthreads();
    if (!is_last) return;
    if (tid == 0) __hip_atomic_store(m.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t_panel_body<256, true>(m.S + (long)m.sh * 128 + m.sh, m.ngrp, GW * GW, 128, m.a0, m.c0, m.c1, m.T, m.Th, m.Tth, m.ldt, m.ld, (float*)gh_smem, tdiag);
}
void launch_leaf_mid(const GemmArgs& g1, const float* Sp, int nslab, float* S, int sh, int* counter, int a0, int c0, int c1,
                     float* T, half_t* Th, half_t* Tth, int ldt, int ld, hipStream_t s) {
    constexpr int LDS = 2 * TP * TPS * 4;
    static_assert(LDS >= (gemm128::BM + gemm128::BN) * gemm128::LDSP * 2, "the GEMM tile's LDS image fits in the T body's");
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)leaf_mid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)));
    LeafMidArgs m{};
    m.g = g1;
    if (m.g.nsplit < 1) m.g.nsplit = 1;
    if (m.g.nslab_in < 1) m.g.nslab_in = 1;
    m.gx = (g1.M + gemm128::BM - 1) / gemm128::BM; m.nX = m.gx * m.g.nsplit;
    m.Sp = Sp; m.nslab = nslab; m.S = S; m.sh = sh; m.counter = counter;
    m.ngrp = std::max(1, std::min(LEAF_MID_MAX_GROUPS, nslab / 256));        // (S holds LEAF_MID_MAX_GROUPS windows)
    m.a0 = a0; m.c0 = c0; m.c1 = c1; m.T = T; m.Th = Th; m.Tth = Tth; m.ldt = ldt; m.ld = ld <= 0 ? ldt : ld;
    hipLaunchKernelGGL(leaf_mid_kernel, dim3(m.nX + MID_RB * m.ngrp), dim3(256), LDS, s, m);
}

// ------------------------------------------------------------------ fused leaf (round 5): three launches between two solves
// A steady leaf of the flat schedule used to cost the chain stream seven launches: gh_solve, gh_apply, leaf_mid (X for the rest of the block +
// Gram sum + T), leaf_xt (Y), the K = 128 update of the rest of the block, gh_gram, gh_reduce -- 150 us, 60 of them the solve (kernel trace,
// profiles/r04_c4_leaf_timeline.txt).  Every one of them re-reads what its predecessor held in LDS, and five of them work on the WHOLE rest of
// the block although the next solve needs only the next leaf's 128 columns.  Now the chain stream touches only those 128 columns (the rest of
// the block follows on the T stream, as the leaf-level look-ahead of round 4 did for tall leaves), in three launches that share row blocks:
//   leaf_a : gh_apply (V_low = A_low C, fp16 copies, partial Gram of the rounded reflectors) + the partial X = (s P)^T V of the NEXT panel P
//            over the same 64 rows (fp16 MFMA out of LDS: V is already there, P is staged like the GEMMs' A_F32T operand)
//   leaf_m : sums the partials of S and X (fixed order), and the last arriver (agent-scope counter, as leaf_mid_kernel) inverts T AND forms
//            Y = fp16(tau (X_hi + X_lo) T'^T) for the next panel out of LDS -- leaf_xt's arithmetic -- before it stores T
//   leaf_b : P -= (1/s) V Y^T on 64-row blocks (fp16 MFMA, operands straight from L2 in fragment layout) and, from the updated values still
//            in registers, the partial Gram matrix of the next leaf (fp64 MFMA, all 36 upper tiles: a block's rows are updated by ONE
//            workgroup) -- gh_gram's loop, so gh_reduce + gh_solve follow unchanged.  Its first thread publishes the chain's progress word
//            for the T stream (every earlier launch of the chain stream -- the leaf's reflectors and T -- is complete and visible).
typedef half_t half8f __attribute__((ext_vector_type(8)));
constexpr int FL_PT_OFF = (8256 + GW * GH_TS) * 4;                // byte offset of the staged next-panel tile in leaf_a's LDS
constexpr int FL_VS_OFF = FL_PT_OFF + GW * 72 * 2;              // ... of V (fp32, row-major [64][132]) on its way to 16-byte global stores
constexpr int FL_A_LDS = FL_VS_OFF + 64 * 132 * 4;
// X partial: this wave's 32 columns of the next panel (c = 32 wave + ...) against all 128 reflectors, over the 64 rows in Pt / Ts
__device__ __forceinline__ void fl_x_add(const half_t* Pt, const half_t* Ts, floatx16p (&xa)[4], int lane, int wave) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        const half8f av = *(const half8f*)&Pt[(32 * wave + r) * 72 + ks * 16 + 8 * h];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const half8f bv = *(const half8f*)&Ts[(32 * j + r) * 72 + ks * 16 + 8 * h];
            xa[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, xa[j], 0, 0, 0);
        }
    }
}
// stage 64 rows x 128 columns of the next panel (fp32, already in registers as 2 x 4 float4: thread block (mg, kg) = columns 4 mg .. +3 of
// rows 4 kg .. +3) as Pt[column][row] = fp16(scale * value): four consecutive rows = one 8-byte LDS store (the GEMMs' A_F32T staging)
__device__ __forceinline__ void fl_stage_pt(half_t* Pt, const float4 (&pv)[8], float sc, int tid) {
    using gemm128::pack2;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int id = tid + 256 * i, mg = id & 31, kg = id >> 5;
        const float4 v0 = pv[i * 4 + 0], v1 = pv[i * 4 + 1], v2 = pv[i * 4 + 2], v3 = pv[i * 4 + 3];
        uint2 w;
        half_t* base = Pt + (mg * 4) * 72 + kg * 4;
        w.x = pack2(v0.x * sc, v1.x * sc); w.y = pack2(v2.x * sc, v3.x * sc); *(uint2*)(base) = w;
        w.x = pack2(v0.y * sc, v1.y * sc); w.y = pack2(v2.y * sc, v3.y * sc); *(uint2*)(base + 72) = w;
        w.x = pack2(v0.z * sc, v1.z * sc); w.y = pack2(v2.z * sc, v3.z * sc); *(uint2*)(base + 2 * 72) = w;
        w.x = pack2(v0.w * sc, v1.w * sc); w.y = pack2(v2.w * sc, v3.w * sc); *(uint2*)(base + 3 * 72) = w;
    }
}
// workgroup b of n -> row block i with (ab0 + i) % 8 == b % 8 (a rotation inside every complete group of eight; the last, incomplete group keeps i = b)
__device__ __forceinline__ int fl_row_block(int b, int n, int ab0) {
    if ((b | 7) >= n) return b;
    return (b & ~7) + ((b - ab0) & 7);
}
__global__ __launch_bounds__(256) void leaf_a_kernel(LeafArgs a, const float* __restrict__ Cv, float* __restrict__ Sp, float* __restrict__ Xp,
                                                     int nlow, int iters, int nb, float in_scale) {
    float* As = (float*)gh_smem;                         // [64][129] = 8256 floats
    float* Cs = (float*)gh_smem + 8256;                  // [128][GH_TS]
    half_t* Ts = (half_t*)gh_smem;                       // [128][72] halves, reuses the As region after the MFMAs
    half_t* Pt = (half_t*)(gh_smem + FL_PT_OFF);         // [128][72] halves: the next panel's rows, transposed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    KT_DECL; KT();
    floatx16p xa[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int e = 0; e < 16; e++) xa[q][e] = 0.f;
    auto store_x = [&](int slot) {
        float* out = Xp + (long)slot * (GW * GW);
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) out[(32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h) * GW + 32 * j + r] = xa[j][e];
    };
    if ((int)blockIdx.x >= nlow) {                        // top-block rows: V^T tile straight from the fp16 copy gh_solve left
        const int trow0 = a.c0 + ((int)blockIdx.x - nlow) * 64;
        float4 pv[8];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int id = tid + 256 * i, mg = id & 31, kg = id >> 5;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = trow0 + kg * 4 + j;
                pv[i * 4 + j] = *(const float4*)(a.A + (long)min(row, a.mrows - 1) * a.lda + nb + 4 * mg);
                if (row >= a.c1 || row >= a.mrows) pv[i * 4 + j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int e = tid; e < GW * 64; e += 256) {
            const int c = e & 127, lr = e >> 7, row = trow0 + lr, gc = a.cb + c;
            half_t v = (half_t)0.f;
            if (row < a.c1 && gc >= a.c0 && gc < a.c1 && row >= gc) v = a.Vh[(long)row * a.ldvh + gc];
            Ts[c * 72 + lr] = v;
        }
        fl_stage_pt(Pt, pv, in_scale, tid);
        __syncthreads();
        gh_partial_gram(Ts, Sp + (long)blockIdx.x * (GW * GW), lane, wave);
        fl_x_add(Pt, Ts, xa, lane, wave);
        store_x((int)blockIdx.x);
        return;
    }
    // low rows.  All global loads of a row block first (C -- once --, the leaf's rows, the next panel's rows: one memory latency), the exact-f32
    // product V = A_low C with the triangular C's K blocks balanced over the waves, then V goes through LDS once more so that every global
    // store is a whole 16-byte (fp32, into A) or 8-byte (fp16 copy) piece of a row -- as 4- and 2-byte stores straight from the MFMA layout
    // (gh_apply_kernel) they were half of this kernel's time (in-kernel stamps, tools/ktrace_fl.sh).
    float* Vs = (float*)(gh_smem + FL_VS_OFF);            // [64][132] fp32: V of this row block, row-major
    floatx16p ga[3];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int e = 0; e < 16; e++) ga[q][e] = 0.f;
    float4 av[8], pv[8];
    auto issue_loads = [&](int row0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int id = tid + 256 * i, lr = id >> 5, c4 = id & 31, row = min(row0 + lr, a.mrows - 1);
            av[i] = *(const float4*)(a.A + (long)row * a.lda + a.cb + 4 * c4);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int id = tid + 256 * i, mg = id & 31, kg = id >> 5;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = row0 + kg * 4 + j;
                pv[i * 4 + j] = *(const float4*)(a.A + (long)min(row, a.mrows - 1) * a.lda + nb + 4 * mg);
            }
        }
    };
    // XCD-aware block -> rows mapping (speed only): workgroups b and b + 8 share an XCD and its L2 (observed round-robin placement,
    // MI355X_MICROARCH.md), so the 64-row block with ABSOLUTE index ab (row / 64 / iters) goes to a workgroup with b % 8 == ab % 8 -- in every
    // launch of leaf_a and leaf_b, for every leaf: the rows leaf_b wrote (the next panel) are read by the next leaf's leaf_a on the same XCD,
    // and leaf_b finds the reflector rows leaf_a has just written, and the panel rows it read, in its own L2.
    const int rb = fl_row_block((int)blockIdx.x, nlow, a.c1 / (64 * iters));
    {   // C arrives in window coordinates, zero outside the leaf and below the diagonal; its loads and the first row block's in flight together
        float4 cv[16];
#pragma unroll
        for (int q = 0; q < 16; q++) { const int e4 = tid + 256 * q; cv[q] = *(const float4*)&Cv[(e4 >> 5) * GW + (e4 & 31) * 4]; }
        issue_loads(a.c1 + rb * iters * 64);
#pragma unroll
        for (int q = 0; q < 16; q++) { const int e4 = tid + 256 * q; *(float4*)&Cs[(e4 >> 5) * GH_TS + (e4 & 31) * 4] = cv[q]; }
    }
    for (int it = 0; it < iters; it++) {
    const int row0 = a.c1 + (rb * iters + it) * 64;
    if (row0 >= a.mrows) break;
    if (it) { __syncthreads(); issue_loads(row0); }        // (the previous block's V^T rows and next-panel rows have been read)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int id = tid + 256 * i, lr = id >> 5, c4 = id & 31, row = row0 + lr;
        const float4 v = (row < a.mrows) ? av[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = &As[lr * 129 + 4 * c4];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    KT();
    floatx16p acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; e++) { acc0[e] = 0.f; acc1[e] = 0.f; }
    // C is upper triangular (C[k][j] = 0 for k > j): column tile ct needs k < 32 (ct + 1).  Wave w takes rows 0-31 of column tile w and
    // rows 32-63 of column tile 3 - w: five K blocks of 32 per wave (gh_apply_kernel: both row tiles of column tile w, up to eight)
    const int r = lane & 31, kk = lane >> 5;
    const int nA = 32 * wave, nB = 32 * (3 - wave);
    const int kendA = nA + 32, kendB = nB + 32;
    for (int k1 = 0; k1 < GW; k1 += 16) {
        if (k1 < kendA) {
            float a0[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int k = k1 + 2 * u + kk; a0[u] = As[r * 129 + k]; b[u] = Cs[k * GH_TS + nA + r]; }
#pragma unroll
            for (int u = 0; u < 8; u++) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b[u], acc0, 0, 0, 0);
        }
        if (k1 < kendB) {
            float a1[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int k = k1 + 2 * u + kk; a1[u] = As[(32 + r) * 129 + k]; b[u] = Cs[k * GH_TS + nB + r]; }
#pragma unroll
            for (int u = 0; u < 8; u++) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b[u], acc1, 0, 0, 0);
        }
    }
    __syncthreads();                                     // As is dead: reuse as the transpose buffer
    KT();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int id = tid + 256 * i, kg = id >> 5;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (row0 + kg * 4 + j >= a.mrows) pv[i * 4 + j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    fl_stage_pt(Pt, pv, in_scale, tid);                    // (the next panel's rows landed during the product)
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        const int n0 = mt == 0 ? nA : nB;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int lm = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            const float v = mt == 0 ? acc0[e] : acc1[e];
            Vs[lm * 132 + n0 + r] = v;
            Ts[(n0 + r) * 72 + lm] = (half_t)v;
        }
    }
    __syncthreads();
    KT();
    {
        typedef half_t half4a __attribute__((ext_vector_type(4)));
        const bool whole = a.c0 == a.cb && a.c1 == a.cb + GW;       // the leaf fills its window (the fused path's case)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int id = tid + 256 * i, lr = id >> 5, c4 = id & 31, row = row0 + lr;
            if (row >= a.mrows) continue;
            const float4 v = *(const float4*)&Vs[lr * 132 + 4 * c4];
            const int gc = a.cb + 4 * c4;
            if (whole) {
                *(float4*)(a.A + (long)row * a.lda + gc) = v;
                half4a hv; hv[0] = (half_t)v.x; hv[1] = (half_t)v.y; hv[2] = (half_t)v.z; hv[3] = (half_t)v.w;
                *(half4a*)(a.Vh + (long)row * a.ldvh + gc) = hv;
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (gc + c >= a.c0 && gc + c < a.c1) { a.A[(long)row * a.lda + gc + c] = vv[c]; a.Vh[(long)row * a.ldvh + gc + c] = (half_t)vv[c]; }
            }
        }
    }
    gh_partial_gram_add(Ts, ga, lane, wave);
    fl_x_add(Pt, Ts, xa, lane, wave);
    KT();
    // V^T rows: 128 columns x 64 rows of this workgroup, 16-B chunks along the row index
    for (int e = tid; e < GW * 8; e += 256) {
        const int c = e >> 3, ch = e & 7;
        const int gcol = a.cb + c;
        if (gcol >= a.c0 && gcol < a.c1) {
            half_t* dst = a.Vt + (long)gcol * a.ldvt + row0 + 8 * ch;
            if ((row0 & 7) == 0 && row0 + 8 * ch + 8 <= a.mrows) {
                *(uint4*)dst = *(const uint4*)&Ts[c * 72 + 8 * ch];
            } else {
                for (int q = 0; q < 8; q++)
                    if (row0 + 8 * ch + q < a.mrows) dst[q] = Ts[c * 72 + 8 * ch + q];
            }
        }
    }
    }                                                      // row blocks
    KT();
    gh_partial_gram_store(ga, Sp + (long)rb * (GW * GW), lane, wave);
    store_x(rb);
    KT(); KT_DUMP(2, "leaf_a loads+stage A,C|mfma|stage P,V|store V+gram+x|vt|store partials");
}

struct LeafM2Args {
    const float* Sp; const float* Xp; int nslab;          // gh_num_partials partial Gram matrices / partial X's (128 x 128 each)
    float* S; float* Xs; int ngrp;                        // sums: ngrp windows each (the partials are cut into ngrp ranges)
    int* counter;                                         // zero between launches (the last arriver resets it)
    int sh, a0, c0, c1; float* T; half_t* Th; half_t* Tth; int ldt, ld;
    half_t* Y;                                            // [128 columns of the next panel][128 reflectors]
    int* pub_flag; int pub_value;                         // published at the start: every earlier launch of the chain stream (the leaf's reflectors) is complete
};
typedef half_t half4m __attribute__((ext_vector_type(4)));
typedef float float4m __attribute__((ext_vector_type(4)));
// 1024 threads: the tail (one workgroup: loads of S and X, the inverse, Y, the stores of T) is latency- and issue-bound per thread; with 256
// threads it took 29 us of the kernel's 40 (in-kernel stamps, tools/ktrace_fl.sh), and the sixteen waves of a reducer take the partials in ONE
// batch of loads each.
__global__ __launch_bounds__(1024) void leaf_m_kernel(LeafM2Args m) {
    const int nred = MID_RB * m.ngrp;                        // reducers per matrix (S first, then X)
    __shared__ float4 part[16][64];
    __shared__ float tdiag[TP];
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef MPQR_KTRACE
    long kt_[16]; int kn_ = 0; const bool kon_ = (threadIdx.x == 0);
#endif
    KT();
    if (m.pub_flag && blockIdx.x == 0 && tid == 0) __hip_atomic_store(m.pub_flag, m.pub_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int kind = (int)blockIdx.x / nred, b = (int)blockIdx.x % nred;      // 0: S, 1: X
    const int grp = b / MID_RB;
    const int q_lo = (int)((long)m.nslab * grp / m.ngrp), q_hi = (int)((long)m.nslab * (grp + 1) / m.ngrp);
    const int e = ((b % MID_RB) * 64 + lane) * 4;            // four consecutive entries of one row of the window
    const bool act = kind ? true : (((e >> 7) >> 5) <= ((e & 127) >> 5));      // S: the partials hold the 10 upper 32 x 32 tiles only
    const float* src = kind ? m.Xp : m.Sp;
    float* dst = (kind ? m.Xs : m.S) + (long)grp * (GW * GW);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {                                               // wave w: partials q_lo + w, + 16, ...; 16 loads in flight, four interleaved chains
        float4 s0 = s, s1 = s, s2 = s, s3 = s;
        auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
        for (int q = q_lo + wave; q < q_hi; q += 256) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = *(const float4*)&src[(long)min(q + 16 * u, q_hi - 1) * (GW * GW) + e];
#pragma unroll
            for (int u = 0; u < 16; u++) if (q + 16 * u >= q_hi) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 16; u += 4) { add(s0, v[u]); add(s1, v[u + 1]); add(s2, v[u + 2]); add(s3, v[u + 3]); }
        }
        s.x = (s0.x + s1.x) + (s2.x + s3.x); s.y = (s0.y + s1.y) + (s2.y + s3.y);
        s.z = (s0.z + s1.z) + (s2.z + s3.z); s.w = (s0.w + s1.w) + (s2.w + s3.w);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        float4 t[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {                     // fixed order: ((p0 + p1) + (p2 + p3)) per group of four waves, then the same over the groups
            const float4 p0 = part[4 * g4][lane], p1 = part[4 * g4 + 1][lane], p2 = part[4 * g4 + 2][lane], p3 = part[4 * g4 + 3][lane];
            t[g4] = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
        }
        float4m r4;
        r4[0] = (t[0].x + t[1].x) + (t[2].x + t[3].x); r4[1] = (t[0].y + t[1].y) + (t[2].y + t[3].y);
        r4[2] = (t[0].z + t[1].z) + (t[2].z + t[3].z); r4[3] = (t[0].w + t[1].w) + (t[2].w + t[3].w);
        // one 16-byte store with sc1 (write-through: the hand-off table's producer side)
        const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, 0x7fffffff, 0x00020000);
        typedef unsigned u4s __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4s, r4), rs_d, e * 4, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have left before this workgroup signals
    }
    __syncthreads();
    if (tid == 0) is_last = __hip_atomic_fetch_add(m.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2 * nred - 1;
    __syncthreads();
    if (!is_last) return;
    KT();
    if (tid == 0) __hip_atomic_store(m.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // S and X were written (sc1, drained) by other workgroups of this launch: every load of them is a 16-byte buffer load with sc1 (aux bit 4;
    // MI355X_MICROARCH.md's hand-off table lists buffer_load_dwordx4 sc1 for the consumer side), all of a lane's loads in flight together.
    // X in MFMA fragment layout (v_mfma_f32_16x16x16_f16, A operand: lane (li, lg) = row li, k = 4 lg .. 4 lg + 3): wave w takes the 16 rows
    // of row tile w % 8 and the column tiles 4 (w / 8) .. + 3.  S: four consecutive entries of a row per load (the fused path takes full,
    // aligned leaves: window = leaf, sh = 0).
    const int li = lane & 15, lg = lane >> 4;
    const int rtile = wave & 7, nhalf = wave >> 3;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)m.Xs, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)m.S, 0, 0x7fffffff, 0x00020000);
    float4m xr[8], sv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) sv[q] = __builtin_bit_cast(float4m, __builtin_amdgcn_raw_buffer_load_b128(rs_s, (tid + 1024 * q) * 16, 0, 16));
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
        xr[ks] = __builtin_bit_cast(float4m, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ((16 * rtile + li) * GW + 16 * ks + 4 * lg) * 4, 0, 16));
    for (int g = 1; g < m.ngrp; g++) {
#pragma unroll
        for (int q = 0; q < 4; q++) sv[q] += __builtin_bit_cast(float4m, __builtin_amdgcn_raw_buffer_load_b128(rs_s, (g * GW * GW + (tid + 1024 * q) * 4) * 4, 0, 16));
#pragma unroll
        for (int ks = 0; ks < 8; ks++)
            xr[ks] += __builtin_bit_cast(float4m, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (g * GW * GW + (16 * rtile + li) * GW + 16 * ks + 4 * lg) * 4, 0, 16));
    }
    float* Ss = (float*)gh_smem;
    float* Ts = Ss + TP * TPS;
    KT();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int e4 = tid + 1024 * q, i = e4 >> 5, j = (e4 & 31) * 4;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float v = (j + c >= i) ? sv[q][c] : 0.f;
            if (i == j + c) { tdiag[i] = v > 0.f ? 2.0f / v : 0.f; v = 0.f; }
            Ss[i * TPS + j + c] = v;
            Ts[i * TPS + j + c] = 0.f;
        }
    }
    __syncthreads();
    tri_inverse_128(Ss, tdiag, Ts, 4, tid);
    KT();
    // Y[c][n] = fp16( tau_n * sum_k (X_hi + X_lo)[c][k] * fp16(T[k][n] / tau_n) ),  k <= n  -- leaf_xt_kernel's arithmetic with Bt = T^T's fp16
    // copy (t_panel_store: Tth[n][k] = fp16(T[k][n] * (1 / T[n][n]))), formed here from the fp32 T in LDS
    {
        half4m xh[8], xl[8];
#pragma unroll
        for (int ks = 0; ks < 8; ks++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const half_t hv = (half_t)xr[ks][c];
                xh[ks][c] = hv; xl[ks][c] = (half_t)(xr[ks][c] - (float)hv);
            }
#pragma unroll
        for (int cq = 0; cq < 4; cq++) {
            const int ct = 4 * nhalf + cq;
            const int n = 16 * ct + li;
            const float tnn = Ts[n * TPS + n];
            const float rti = tnn != 0.f ? 1.0f / tnn : 0.f;
            float4m acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ks++) {
                if (ks > ct) continue;                        // T^T rows end at the diagonal (tri == 2); wave-uniform
                half4m bv;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int k = 16 * ks + 4 * lg + c;
                    bv[c] = (half_t)((k <= n ? Ts[k * TPS + n] : 0.f) * rti);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x16f16(xh[ks], bv, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x16f16(xl[ks], bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int e2 = 0; e2 < 4; e2++) m.Y[(long)(16 * rtile + 4 * lg + e2) * GW + n] = (half_t)(tnn * acc[e2]);
        }
    }
    KT();
    t_panel_store<1024>(Ts, m.a0, m.c0, m.c1, m.T, m.Th, m.Tth, m.ldt, m.ld);
    KT(); KT_DUMP(7, "leaf_m reduce+arrive|loads issue|fill+invert|Y|store T");
}

// P[rows >= c0][nb .. nb + 128) -= alpha V Y^T  (V = the leaf's fp16 reflectors, Y from leaf_m), 64 rows per workgroup, and the partial Gram
// matrix (fp64, upper 16 x 16 tiles) of the updated rows from c1 on: the next leaf's gh_gram.  Workgroups 0 / 1 take the leaf's top rows
// [c0, c1) -- R rows, no Gram --, workgroup 2 + p the rows c1 + 64 (p iters + it).
constexpr int FL_B_ROWS = 64;
__global__ __launch_bounds__(256) void leaf_b_kernel(LeafArgs a, int nb, const half_t* __restrict__ Y, float alpha, double* __restrict__ Gp,
                                                     int iters, int do_gram, int* pub_flag, int pub_value) {
    if (pub_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(pub_flag, pub_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double* tile = (double*)gh_smem;                      // [64][GH_TD] doubles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, li = lane & 15, lk = lane >> 4;
#ifdef MPQR_KTRACE
    long kt_[16]; int kn_ = 0; const bool kon_ = (threadIdx.x == 0 && blockIdx.x == 2);
#endif
    KT();
    const bool top = blockIdx.x < 2;
    const int ngram = (int)gridDim.x - 2;
    // (XCD-aware: the block with absolute index ab goes to a workgroup with blockIdx % 8 == ab % 8 -- see leaf_a_kernel; workgroup g = 2 + u)
    const int u = (int)blockIdx.x - 2;
    const int p = top ? 0 : (((u | 7) >= ngram) ? u : (u & ~7) + ((u + 2 - a.c1 / (FL_B_ROWS * iters)) & 7));
    typedef double double4g __attribute__((ext_vector_type(4)));
    // Gram tiles of this wave: t = wave + 4 s (s < 9) among the 36 upper tiles of the 8 x 8 tile grid, row-major
    int ca[9], cb[9];
    double4g acc[9];
#pragma unroll
    for (int s = 0; s < 9; s++) {
        int t = wave + 4 * s, ti = 0;
        while (t >= 8 - ti) { t -= 8 - ti; ti++; }
        ca[s] = 16 * ti; cb[s] = 16 * (ti + t);
        acc[s] = double4g{0, 0, 0, 0};
    }
    const int rt = wave & 1, cp = wave >> 1;               // this wave's 32-row tile and pair of 32-column tiles of the 64 x 128 block
    for (int it = 0; it < (top ? 1 : iters); it++) {
        const int row0 = top ? a.c0 + (int)blockIdx.x * 64 : a.c1 + (p * iters + it) * FL_B_ROWS;
        const int rend = top ? a.c1 : a.mrows;
        if (row0 >= rend) break;
        if (it) __syncthreads();
        // operands in MFMA fragment layout straight from global memory (L2): V rows (k = reflector, 16-B = 8 reflectors), Y rows, old values
        half8f va[8], yb[2][8];
        float oldv[2][16];
        const int vrow = min(row0 + 32 * rt + r, a.mrows - 1);
#pragma unroll
        for (int ks = 0; ks < 8; ks++) va[ks] = *(const half8f*)&a.Vh[(long)vrow * a.ldvh + a.c0 + 16 * ks + 8 * h];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int ks = 0; ks < 8; ks++) yb[j][ks] = *(const half8f*)&Y[(long)(64 * cp + 32 * j + r) * GW + 16 * ks + 8 * h];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = min(row0 + 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * h, a.mrows - 1);
                oldv[j][e] = a.A[(long)row * a.lda + nb + 64 * cp + 32 * j + r];
            }
        floatx16p ua[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
#pragma unroll
            for (int e = 0; e < 16; e++) ua[j][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ks++) ua[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[ks], yb[j][ks], ua[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int lr = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * h, row = row0 + lr, col = 64 * cp + 32 * j + r;
                const float nv = oldv[j][e] - alpha * ua[j][e];
                tile[lr * GH_TD + col] = row < rend ? (double)nv : 0.0;      // (exact: the Gram matrix and the stores below see the fp32 value)
            }
        __syncthreads();
        KT();
        // the updated rows go out as whole 16-byte pieces (as 4-byte stores straight from the MFMA layout they cost this kernel ~3 us)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int id = tid + 256 * i, lr = id >> 5, c4 = id & 31, row = row0 + lr;
            const double2 d0 = *(const double2*)&tile[lr * GH_TD + 4 * c4], d1 = *(const double2*)&tile[lr * GH_TD + 4 * c4 + 2];
            if (row < rend) *(float4*)(a.A + (long)row * a.lda + nb + 4 * c4) = make_float4((float)d0.x, (float)d0.y, (float)d1.x, (float)d1.y);
        }
        if (top || !do_gram) continue;
        for (int k0 = 0; k0 < FL_B_ROWS; k0 += 8) {        // gh_gram_kernel's loop (two K steps per iteration), nine tiles per wave
            const double* tr0 = &tile[(k0 + lk) * GH_TD + li];
            const double* tr1 = tr0 + 4 * GH_TD;
            double a0[9], b0[9], a1[9], b1[9];
#pragma unroll
            for (int s = 0; s < 9; s++) { a0[s] = tr0[ca[s]]; b0[s] = tr0[cb[s]]; a1[s] = tr1[ca[s]]; b1[s] = tr1[cb[s]]; }
#pragma unroll
            for (int s = 0; s < 9; s++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], b0[s], acc[s], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 9; s++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s], b1[s], acc[s], 0, 0, 0);
        }
    }
    if (top || !do_gram) return;
    KT();
    double* out = Gp + (long)p * (GW * GW);
#pragma unroll
    for (int s = 0; s < 9; s++)
#pragma unroll
        for (int v = 0; v < 4; v++) out[(ca[s] + lk + 4 * v) * GW + cb[s] + li] = acc[s][v];   // upper tiles only: gh_reduce mirrors
    KT(); KT_DUMP(5, "leaf_b load+update+tile|gram mfma|store");
}

int fl_gram_partials(const LeafArgs& a) {                   // partial Gram matrices leaf_b leaves for the NEXT leaf (rows from a.c1 on)
    const int rows = a.mrows - a.c1;
    const int it = rows >= 49152 ? 4 : rows >= 24576 ? 2 : 1;
    return ((rows + FL_B_ROWS - 1) / FL_B_ROWS + it - 1) / it;
}
void launch_leaf_a(const LeafArgs& a, const float* Cv, float* Sp, float* Xp, int nb, float in_scale, hipStream_t s) {
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)leaf_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FL_A_LDS)));
    const int it = gh_apply_iters(a);
    const int nlow = ((a.mrows - a.c1 + 63) / 64 + it - 1) / it;
    const int ntop = (a.c1 - a.c0 + 63) / 64;
    hipLaunchKernelGGL(leaf_a_kernel, dim3(nlow + ntop), dim3(256), FL_A_LDS, s, a, Cv, Sp, Xp, nlow, it, nb, in_scale);
}
void launch_leaf_m(const float* Sp, const float* Xp, int nslab, float* S, float* Xs, int* counter, int sh, int a0, int c0, int c1,
                   float* T, half_t* Th, half_t* Tth, int ldt, int ld, half_t* Y, hipStream_t s, int* pub_flag, int pub_value) {
    constexpr int LDS = 2 * TP * TPS * 4;
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)leaf_m_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)));
    LeafM2Args m{};
    m.Sp = Sp; m.Xp = Xp; m.nslab = nslab; m.S = S; m.Xs = Xs; m.counter = counter;
    m.ngrp = std::max(1, std::min(LEAF_MID_MAX_GROUPS, nslab / 256));
    m.sh = sh; m.a0 = a0; m.c0 = c0; m.c1 = c1; m.T = T; m.Th = Th; m.Tth = Tth; m.ldt = ldt; m.ld = ld <= 0 ? ldt : ld; m.Y = Y;
    m.pub_flag = pub_flag; m.pub_value = pub_value;
    hipLaunchKernelGGL(leaf_m_kernel, dim3(2 * MID_RB * m.ngrp), dim3(1024), LDS, s, m);
}
void launch_leaf_b(const LeafArgs& a, int nb, const half_t* Y, float alpha, double* Gp, bool do_gram, int* pub_flag, int pub_value, hipStream_t s) {
    constexpr int LDS = FL_B_ROWS * GH_TD * 8;
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)leaf_b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)));
    const int rows = a.mrows - a.c1;
    const int it = rows >= 49152 ? 4 : rows >= 24576 ? 2 : 1;
    const int ng = rows > 0 ? ((rows + FL_B_ROWS - 1) / FL_B_ROWS + it - 1) / it : 0;
    hipLaunchKernelGGL(leaf_b_kernel, dim3(2 + ng), dim3(256), LDS, s, a, nb, Y, alpha, Gp, it, do_gram ? 1 : 0, pub_flag, pub_value);
}
// the reduction of leaf_b's partials (the second half of launch_gh_gram)
void launch_gh_gram_reduce(const double* Gp, int nwg, double* G, hipStream_t s) {
    hipLaunchKernelGGL(gh_reduce_kernel, dim3(256), dim3(256), 0, s, Gp, nwg, G);
}

// fp16 copies of one column block of a block-level T: Th[0:rows, c:c+w] = T[0:rows, c:c+w], Tth[c:c+w, 0:rows] = its transpose
// P != nullptr: the column block itself arrives as nz partial sums (slabs of `slab` floats, row stride ldp: a split-K product);
// they are summed here and written to T first
__global__ __launch_bounds__(256) void t_colblock_h16_kernel(float* __restrict__ T, half_t* __restrict__ Th,
                                                             half_t* __restrict__ Tth, int ld, int rows, int c, int w,
                                                             const float* __restrict__ P, int nz, long slab, int ldp) {
    __shared__ float tile[32][33];
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = i0 + ty + 8 * q, j = j0 + tx;
        float v = 0.f;
        if (i < rows && j < w) {
            if (P) {
                for (int z = 0; z < nz; z++) v += P[(long)z * slab + (long)i * ldp + j];
                T[(long)i * ld + c + j] = v;
            } else v = T[(long)i * ld + c + j];
            const float tii = T[(long)i * ld + i];                      // tau_i: the fp16 copies carry row n / tau_n
            Th[(long)i * ld + c + j] = (half_t)(tii != 0.f ? v / tii : 0.f);
        }
        tile[ty + 8 * q][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int j = j0 + ty + 8 * q, i = i0 + tx;
        if (i < rows && j < w) {
            const float tjj = T[(long)(c + j) * ld + c + j];
            Tth[(long)(c + j) * ld + i] = (half_t)(tjj != 0.f ? tile[tx][ty + 8 * q] / tjj : 0.f);
        }
    }
}
void launch_t_colblock_h16(float* T, half_t* Th, half_t* Tth, int ld, int rows, int c, int w, hipStream_t s,
                           const float* P, int nz, long slab, int ldp) {
    if (rows <= 0 || w <= 0) return;
    hipLaunchKernelGGL(t_colblock_h16_kernel, dim3((w + 31) / 32, (rows + 31) / 32), dim3(256), 0, s, T, Th, Tth, ld, rows, c, w,
                       P, nz, slab, ldp);
}

// parent T = [[T_L, T_LR], [0, T_R]] placed inside the parent's 64-aligned reflector range
__global__ __launch_bounds__(256) void t_assemble_kernel(float* __restrict__ T, half_t* __restrict__ Th,
                                                         half_t* __restrict__ Tth, int ldt, int A0,
                                                         const float* __restrict__ TL, int ldl, int aL0, int c0, int cm,
                                                         const float* __restrict__ TR, int ldr, int aR0, int c1,
                                                         const float* __restrict__ TLR, int ldlr, int nz, long zslab) {
    // TLR may arrive as nz partial sums (split-K product), zslab floats apart
    auto tlr = [&](long off) { float s = 0.f; for (int z = 0; z < nz; z++) s += TLR[(long)z * zslab + off]; return s; };
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)ldt * ldt) return;
    const int i = (int)(e / ldt), j = (int)(e % ldt);
    const int gi = A0 + i, gj = A0 + j;
    float v = 0.f;
    const bool iL = gi >= c0 && gi < cm, jL = gj >= c0 && gj < cm;
    const bool iR = gi >= cm && gi < c1, jR = gj >= cm && gj < c1;
    if (iL && jL) v = TL[(long)(gi - aL0) * ldl + (gj - aL0)];
    else if (iR && jR) v = TR[(long)(gi - aR0) * ldr + (gj - aR0)];
    else if (iL && jR) v = tlr((long)(gi - aL0) * ldlr + (gj - aR0));
    T[(long)i * ldt + j] = v;
    float dii = 0.f;                                      // tau_i = T[i][i]: both fp16 copies carry their row i divided by it
    if (gi >= c0 && gi < cm) dii = TL[(long)(gi - aL0) * ldl + (gi - aL0)];
    else if (gi >= cm && gi < c1) dii = TR[(long)(gi - aR0) * ldr + (gi - aR0)];
    const float rdi = dii != 0.f ? 1.0f / dii : 0.f;
    Th[(long)i * ldt + j] = (half_t)(v * rdi);
    {   // T^T entry (i, j) = T[j][i]: the same thread index walks T^T row-major, so this store is coalesced too
        const int gj2 = A0 + i, gi2 = A0 + j;             // T row gi2, column gj2
        float u = 0.f;
        const bool iL2 = gi2 >= c0 && gi2 < cm, jL2 = gj2 >= c0 && gj2 < cm;
        const bool iR2 = gi2 >= cm && gi2 < c1, jR2 = gj2 >= cm && gj2 < c1;
        if (iL2 && jL2) u = TL[(long)(gi2 - aL0) * ldl + (gj2 - aL0)];
        else if (iR2 && jR2) u = TR[(long)(gi2 - aR0) * ldr + (gj2 - aR0)];
        else if (iL2 && jR2) u = tlr((long)(gi2 - aL0) * ldlr + (gj2 - aR0));
        Tth[(long)i * ldt + j] = (half_t)(u * rdi);
    }
}

void launch_t_assemble(float* T, half_t* Th, half_t* Tth, int ldt, int A0, const float* TL, int ldl, int aL0, int c0,
                       int cm, const float* TR, int ldr, int aR0, int c1, const float* TLR, int ldlr, hipStream_t s, int nz,
                       long zslab) {
    const long tot = (long)ldt * ldt;
    hipLaunchKernelGGL(t_assemble_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, T, Th, Tth, ldt, A0,
                       TL, ldl, aL0, c0, cm, TR, ldr, aR0, c1, TLR, ldlr, nz < 1 ? 1 : nz, zslab);
}

}  // namespace mpqr
