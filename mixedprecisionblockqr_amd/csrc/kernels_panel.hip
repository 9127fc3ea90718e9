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
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) v = *(const float4*)ptr;
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
// After reflectors c0..k-1 every leaf column is (a combination of the ORIGINAL leaf columns) below row c1
// plus explicit values in the top w = c1-c0 rows:  B_low = A_low M,  B_top explicit.  All inner products over
// the tall part therefore follow from G = A_low^T A_low (w x w), so the Householder recursion (same u, alpha,
// v, w_j as above; same sign rule and zero-column skip) runs on w x w matrices:
//   launch 1  gh_gram   : per-workgroup partial G over 256 rows, fp32 data, products and sums in fp64
//   launch 2  gh_solve  : one workgroup, fp64: N = M^T G M kept by rank-2 updates, B_top, M; emits R, V_top,
//                         C (upper triangular, V_low = A_low C) and rho_k = ||u_k||^2 / ||a_k||^2
//   launch 3  gh_apply  : V_low = A_low C row-parallel in fp32, written over A_low, plus the fp16 copies
// No pass over the tall data is sequential in k.  Accuracy: V differs from Householder's by O(2^-24 / sqrt(rho));
// a leaf with rho < GH_RHO_MIN raises a flag and the driver redoes the factorisation on the column-by-column
// kernels above (which make no such assumption).
constexpr double GH_RHO_MIN = 1e-8;
constexpr int GH_TS = 36;   // LDS row stride (floats) of the staged 256 x 32 tile

__global__ __launch_bounds__(256) void gh_gram_kernel(LeafArgs a, double* __restrict__ Gp) {
    __shared__ __attribute__((aligned(16))) float tile[256 * GH_TS];
    const int tid = threadIdx.x;
    const int row0 = a.c1 + blockIdx.x * 256;
    {
        const int cg = tid & 7, rl = tid >> 3;
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int lr = p * 32 + rl, row = row0 + lr;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < a.mrows) v = *(const float4*)(a.A + (long)row * a.lda + a.cb + 4 * cg);
            *(float4*)&tile[lr * GH_TS + 4 * cg] = v;
        }
    }
    __syncthreads();
    const int i = tid >> 3, j0 = (tid & 7) * 4;
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll 8
    for (int r = 0; r < 256; r++) {
        const double ai = (double)tile[r * GH_TS + i];
        const float4 bj = *(const float4*)&tile[r * GH_TS + j0];
        acc0 += ai * (double)bj.x; acc1 += ai * (double)bj.y; acc2 += ai * (double)bj.z; acc3 += ai * (double)bj.w;
    }
    double* out = Gp + (long)blockIdx.x * 1024 + i * 32 + j0;
    out[0] = acc0; out[1] = acc1; out[2] = acc2; out[3] = acc3;
}

// One workgroup, 256 threads as a 16 x 16 grid; thread (ti,tj) keeps the 2 x 2 blocks
// {2ti,2ti+1} x {2tj,2tj+1} of N, B_top and M in registers (fp64).  Per reflector: the owners of row/column k
// publish them to LDS, every thread derives alpha, inv, w, v_top, c_k redundantly from those vectors, and
// updates its own entries -- two barriers per step.  s_j needs the top-row dot products sum_t B[t][k] B[t][j]:
// they are folded into N up front (N holds the Gram of ALL not-yet-final rows) and row k is removed again
// once it has become a row of R.
__global__ __launch_bounds__(256) void gh_solve_kernel(LeafArgs a, const double* __restrict__ Gp, int nwg,
                                                       float* __restrict__ Cv, int* __restrict__ flag) {
    __shared__ double rowN[32], rowB[32], colB[32], colM[32], col0[32];
    __shared__ double stage[32][33];
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const int w = a.c1 - a.c0, off = a.c0 - a.cb;
    double N[2][2], B[2][2], M[2][2];
    float C[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++) {
            const int i = 2 * ti + x, j = 2 * tj + y;
            double g = 0, b = 0;
            if (i < w && j < w) {
                for (int q = 0; q < nwg; q++) g += Gp[(long)q * 1024 + (off + i) * 32 + off + j];
                b = (double)a.A[(long)(a.c0 + i) * a.lda + a.c0 + j];
            }
            N[x][y] = g; B[x][y] = b; M[x][y] = (i == j) ? 1.0 : 0.0;
            stage[i][j] = b;
        }
    __syncthreads();
    // fold the top rows into N: N_ij += sum_t B[t][i] B[t][j]; col0_j = N_jj = ||a_j||^2 over all leaf rows
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++) {
            const int i = 2 * ti + x, j = 2 * tj + y;
            double s = 0;
            for (int t = 0; t < w; t++) s += stage[t][i] * stage[t][j];
            N[x][y] += s;
            if (i == j) col0[i] = N[x][y];
        }
    __syncthreads();
    for (int kr = 0; kr < w; kr++) {
        // publish row kr of N and B, column kr of B and M
        if (ti == (kr >> 1)) {
            const int x = kr & 1;
            rowN[2 * tj] = N[x][0]; rowN[2 * tj + 1] = N[x][1];
            rowB[2 * tj] = B[x][0]; rowB[2 * tj + 1] = B[x][1];
        }
        if (tj == (kr >> 1)) {
            const int y = kr & 1;
            colB[2 * ti] = B[0][y]; colB[2 * ti + 1] = B[1][y];
            colM[2 * ti] = M[0][y]; colM[2 * ti + 1] = M[1][y];
        }
        __syncthreads();
        const double sk = rowN[kr], u0 = rowB[kr];
        double alpha = 0, inv = 0;
        bool skip = true;
        if (sk > 0) {
            const double nu = sqrt(sk);
            alpha = (u0 >= 0) ? nu : -nu;
            inv = 1.0 / sqrt(2.0 * (sk + fabs(u0) * nu));
            skip = false;
        }
        if (tid == 0) {
            if (!skip && sk < GH_RHO_MIN * col0[kr]) atomicOr(flag, 1);
            if (skip && col0[kr] > 0) atomicOr(flag, 1);      // cancelled to <= 0 but not an exactly-zero column
            const int k = a.c0 + kr;
            const float vd = skip ? 0.f : (float)((u0 + alpha) * inv);
            a.vdiag[k] = vd;
            a.Vh[(long)k * a.ldvh + k] = (half_t)vd;
            a.Vt[(long)k * a.ldvt + k] = (half_t)vd;
        }
        if (!skip) {
            // With u = column kr over the remaining rows:  s_j = u^T b_j = N[kr][j],
            //   w_j = 2 v^T b_j = 2 (s_j + alpha B[kr][j]) inv,   v_top[t] = (B[t][kr] + [t==kr] alpha) inv,
            //   v_low = A_low (M[:,kr] inv).
            // H is orthogonal, so the Gram matrix of the reflected columns over the same rows is unchanged; the
            // only change to N is that row kr (now a row of R) leaves the set:  N'_ij = N_ij - R[kr][i] R[kr][j].
#pragma unroll
            for (int x = 0; x < 2; x++)
#pragma unroll
                for (int y = 0; y < 2; y++) {
                    const int i = 2 * ti + x, j = 2 * tj + y;
                    if (i < w && j < w) {
                        const double wi = (i > kr) ? 2.0 * (rowN[i] + alpha * rowB[i]) * inv : 0.0;
                        const double wj = (j > kr) ? 2.0 * (rowN[j] + alpha * rowB[j]) * inv : 0.0;
                        const double vti = (i >= kr) ? (colB[i] + (i == kr ? alpha : 0.0)) * inv : 0.0;
                        if (j > kr) {
                            // B' = B - v_top w^T on rows >= kr
                            if (i >= kr) B[x][y] -= vti * wj;
                            // M' = M - (M[:,kr] inv) w^T
                            if (i <= kr) M[x][y] -= colM[i] * inv * wj;
                            // remove the now-final row kr:  R[kr][j] = B[kr][j] - v_top[kr] w_j
                            if (i > kr) {
                                const double rki = rowB[i] - ((rowB[kr] + alpha) * inv) * wi;
                                const double rkj = rowB[j] - ((rowB[kr] + alpha) * inv) * wj;
                                N[x][y] -= rki * rkj;
                            }
                        } else if (j == kr) {
                            if (i > kr) B[x][y] = vti;            // reflector below the diagonal
                            else if (i == kr) B[x][y] = -alpha;    // R_kk
                            if (i <= kr) C[x][y] = (float)(colM[i] * inv);
                        }
                    }
                }
        }
        __syncthreads();
    }
    // outputs: top block of A (R on/above the diagonal, reflectors below), fp16 copies, C
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++) {
            const int i = 2 * ti + x, j = 2 * tj + y;
            Cv[i * 32 + j] = (i < w && j < w) ? C[x][y] : 0.f;
            if (i < w && j < w) {
                const float v = (float)B[x][y];
                const int row = a.c0 + i, col = a.c0 + j;
                a.A[(long)row * a.lda + col] = v;
                if (i > j) {
                    a.Vh[(long)row * a.ldvh + col] = (half_t)v;
                    a.Vt[(long)col * a.ldvt + row] = (half_t)v;
                }
            }
        }
}

__global__ __launch_bounds__(256) void gh_apply_kernel(LeafArgs a, const float* __restrict__ Cv) {
    __shared__ __attribute__((aligned(16))) float tile[256 * GH_TS];
    __shared__ __attribute__((aligned(16))) float Cs[32 * GH_TS];
    const int tid = threadIdx.x;
    const int cg = tid & 7, rl = tid >> 3;
    const int row0 = a.c1 + blockIdx.x * 256;
    const int off = a.c0 - a.cb, w = a.c1 - a.c0;
    // C placed at window coordinates: Cs[off+i][off+k] = Cv[i][k]
    for (int e = tid; e < 32 * GH_TS; e += 256) Cs[e] = 0.f;
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
        const int i = e >> 5, k = e & 31;
        if (i < w && k < w) Cs[(off + i) * GH_TS + off + k] = Cv[e];
    }
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int lr = p * 32 + rl, row = row0 + lr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < a.mrows) v = *(const float4*)(a.A + (long)row * a.lda + a.cb + 4 * cg);
        *(float4*)&tile[lr * GH_TS + 4 * cg] = v;
    }
    __syncthreads();
    float4 acc[8];
#pragma unroll
    for (int p = 0; p < 8; p++) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
        const float4 c = *(const float4*)&Cs[i * GH_TS + 4 * cg];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const float x = tile[(p * 32 + rl) * GH_TS + i];
            acc[p].x += x * c.x; acc[p].y += x * c.y; acc[p].z += x * c.z; acc[p].w += x * c.w;
        }
    }
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int lr = p * 32 + rl, row = row0 + lr;
        if (row < a.mrows) {
            float4 o = *(const float4*)&tile[lr * GH_TS + 4 * cg];      // columns outside the leaf keep their data
            const float vals[4] = {acc[p].x, acc[p].y, acc[p].z, acc[p].w};
            float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int gc = a.cb + 4 * cg + c;
                if (gc >= a.c0 && gc < a.c1) {
                    ov[c] = vals[c];
                    a.Vh[(long)row * a.ldvh + gc] = (half_t)vals[c];
                    a.Vt[(long)gc * a.ldvt + row] = (half_t)vals[c];
                }
            }
            *(float4*)(a.A + (long)row * a.lda + a.cb + 4 * cg) = make_float4(ov[0], ov[1], ov[2], ov[3]);
        }
    }
}

void launch_leaf_gram_householder(const LeafArgs& a, double* Gp, float* Cv, int* flag, hipStream_t s) {
    const int nwg = (a.mrows - a.c1 + 255) / 256;
    hipLaunchKernelGGL(gh_gram_kernel, dim3(nwg), dim3(256), 0, s, a, Gp);
    hipLaunchKernelGGL(gh_solve_kernel, dim3(1), dim3(256), 0, s, a, Gp, nwg, Cv, flag);
    hipLaunchKernelGGL(gh_apply_kernel, dim3(nwg), dim3(256), 0, s, a, Cv);
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

// ------------------------------------------------------------------ T of a leaf
// T^{-1} = striu(V^T V) + diag(V^T V)/2  (compact WY with H_i = I - (2/v_i^T v_i) v_i v_i^T), so
// T_ii = 2/S_ii and T[:i,i] = -T_ii T[:i,:i] S[:i,i].  S is the Gram matrix of the fp16-ROUNDED
// reflectors, which keeps I - V T V^T orthogonal for the V the MFMA GEMMs actually multiply with.
__global__ __launch_bounds__(256) void t_leaf_kernel(const float* __restrict__ S, int nslab, long slab_stride, int a0,
                                                     int c0, int c1, float* __restrict__ T, half_t* __restrict__ Th,
                                                     half_t* __restrict__ Tth, int ldt) {
    __shared__ float Ss[32][33];
    __shared__ float Ts[32][33];
    const int tid = threadIdx.x;
    const int w = c1 - c0, off = c0 - a0;
    for (int e = tid; e < 1024; e += 256) {       // one Gram entry per thread and pass, slabs summed in order
        const int i = e >> 5, j = e & 31;
        float v = 0.f;
        if (i < w && j < w && j >= i)
            for (int sl = 0; sl < nslab; sl++) v += S[(long)sl * slab_stride + (long)(off + i) * 64 + off + j];
        Ss[i][j] = v;
    }
    __syncthreads();
    // Row a of T depends on row a only: lane a runs the column recurrence in registers, S read as LDS broadcasts.
    if (tid < 32) {
        float tr[32];
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const float sii = Ss[i][i];
            const float tii = (i < w && sii > 0.f) ? 2.0f / sii : 0.f;
            float sum = 0.f;
#pragma unroll
            for (int b = 0; b < i; b++) sum += tr[b] * Ss[b][i];      // tr[b] == 0 for b < a
            tr[i] = (tid < i) ? -tii * sum : (tid == i ? tii : 0.f);
        }
#pragma unroll
        for (int i = 0; i < 32; i++) Ts[tid][i] = tr[i];
    }
    __syncthreads();
    for (int e = tid; e < ldt * ldt; e += 256) {
        const int i = e / ldt, j = e % ldt;
        const int li = i - off, lj = j - off;
        float v = 0.f;
        if (li >= 0 && li < w && lj >= 0 && lj < w) v = Ts[li][lj];
        T[(long)i * ldt + j] = v;
        Th[(long)i * ldt + j] = (half_t)v;
        Tth[(long)j * ldt + i] = (half_t)v;
    }
}

void launch_t_leaf(const float* S, int nslab, long slab_stride, int a0, int c0, int c1, float* T, half_t* Th,
                   half_t* Tth, int ldt, hipStream_t s) {
    hipLaunchKernelGGL(t_leaf_kernel, dim3(1), dim3(256), 0, s, S, nslab, slab_stride, a0, c0, c1, T, Th, Tth, ldt);
}

// parent T = [[T_L, T_LR], [0, T_R]] placed inside the parent's 64-aligned reflector range
__global__ __launch_bounds__(256) void t_assemble_kernel(float* __restrict__ T, half_t* __restrict__ Th,
                                                         half_t* __restrict__ Tth, int ldt, int A0,
                                                         const float* __restrict__ TL, int ldl, int aL0, int c0, int cm,
                                                         const float* __restrict__ TR, int ldr, int aR0, int c1,
                                                         const float* __restrict__ TLR, int ldlr) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)ldt * ldt) return;
    const int i = (int)(e / ldt), j = (int)(e % ldt);
    const int gi = A0 + i, gj = A0 + j;
    float v = 0.f;
    const bool iL = gi >= c0 && gi < cm, jL = gj >= c0 && gj < cm;
    const bool iR = gi >= cm && gi < c1, jR = gj >= cm && gj < c1;
    if (iL && jL) v = TL[(long)(gi - aL0) * ldl + (gj - aL0)];
    else if (iR && jR) v = TR[(long)(gi - aR0) * ldr + (gj - aR0)];
    else if (iL && jR) v = TLR[(long)(gi - aL0) * ldlr + (gj - aR0)];
    T[(long)i * ldt + j] = v;
    Th[(long)i * ldt + j] = (half_t)v;
    Tth[(long)j * ldt + i] = (half_t)v;
}

void launch_t_assemble(float* T, half_t* Th, half_t* Tth, int ldt, int A0, const float* TL, int ldl, int aL0, int c0,
                       int cm, const float* TR, int ldr, int aR0, int c1, const float* TLR, int ldlr, hipStream_t s) {
    const long tot = (long)ldt * ldt;
    hipLaunchKernelGGL(t_assemble_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, T, Th, Tth, ldt, A0,
                       TL, ldl, aL0, c0, cm, TR, ldr, aR0, c1, TLR, ldlr);
}

}  // namespace mpqr
