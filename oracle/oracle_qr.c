/*
 * oracle_qr.c -- CPU restatement of the reference block-QR hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle_qr.h).  Plain C99, single thread unless
 * built with -fopenmp (only the compact-WY baseline loops carry pragmas).
 */
#include "oracle_qr.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* fp16 emulation (IEEE binary16, round-to-nearest-even), used to restate the
 * reference's __float2half casts (Cuda/mmult.cuh:169-200) on the CPU.       */
static uint16_t f32_to_f16_bits(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007fffffu;
    int32_t  exp  = (int32_t)((x >> 23) & 0xff);
    if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u : 0));
    exp = exp - 127 + 15;
    if (exp >= 0x1f) return (uint16_t)(sign | 0x7c00u);            /* overflow -> inf */
    if (exp <= 0) {                                                  /* subnormal / zero */
        if (exp < -10) return (uint16_t)sign;
        mant |= 0x00800000u;
        int shift = 14 - exp;                                        /* 14..24 */
        uint32_t h = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)exp << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;          /* may carry into exp: ok */
    return (uint16_t)(sign | h);
}
static float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, mant = h & 0x3ffu, x;
    if (exp == 0) {
        if (mant == 0) x = sign;
        else { int e = -1; do { e++; mant <<= 1; } while (!(mant & 0x400u));
               x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3ffu) << 13); }
    } else if (exp == 0x1f) x = sign | 0x7f800000u | (mant << 13);
    else x = sign | ((exp + 127 - 15) << 23) | (mant << 13);
    float f; memcpy(&f, &x, 4); return f;
}
float orc_round_fp16(float x) { return f16_bits_to_f32(f32_to_f16_bits(x)); }

/* ------------------------------------------------------------------ */
/* Cuda/qr.cu:198-293.  Unblocked Householder restricted to the panel columns
 * [global_offset, min(global_offset+panel_width, n)).  fp32, sequential sums in
 * source order.  Zero column: skipped (qr.cu:242-244).  Last column of a square
 * matrix IS reflected (skip commented out at qr.cu:217-219).                   */
void orc_householder_qr(float* A, int m, int n, int global_offset, int panel_width) {
    int r = (panel_width + global_offset) > n ? n : panel_width + global_offset;
    for (int k = global_offset; k < r; k++) {
        int len = m - k;
        float* u = (float*)malloc((size_t)len * sizeof(float));
        for (int i = 0; i < len; i++) u[i] = A[(size_t)n * (i + k) + k];
        int sign = (u[0] >= 0) ? 1 : -1;
        float mag = 0;
        for (int i = 0; i < len; i++) mag += u[i] * u[i];
        if (mag == 0) { free(u); continue; }
        mag = sqrtf(mag);
        u[0] = sign * mag + u[0];
        mag = 0;
        for (int i = 0; i < len; i++) mag += u[i] * u[i];
        mag = sqrtf(mag);
        for (int i = 0; i < len; i++) u[i] /= mag;
        /* tmp = u^T A[k:m, k:r]  then  A[k:m,k:r] -= 2 u tmp */
        float* tmp = (float*)malloc((size_t)(r - k) * sizeof(float));
        for (int col = k; col < r; col++) {
            float ip = 0;
            for (int row = k; row < m; row++) ip += u[row - k] * A[(size_t)row * n + col];
            tmp[col - k] = ip;
        }
        for (int row = k; row < m; row++)
            for (int col = k; col < r; col++) {
                float t2 = u[row - k] * tmp[col - k];
                A[(size_t)row * n + col] = A[(size_t)row * n + col] - 2 * t2;
            }
        /* reflector stored one row below its natural place (qr.cu:283-285) */
        for (int row = k + 1; row < k + len + 1; row++) A[(size_t)row * n + k] = u[row - k - 1];
        free(tmp); free(u);
    }
}

/* Cuda/qr.cu:296-335 */
void orc_q_backward_accumulation(const float* A, float* Q, int m, int n) {
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Q[(size_t)i * m + j] = (i == j) ? 1.f : 0.f;
    float* tmp = (float*)malloc((size_t)m * sizeof(float));
    for (int j = n - 1; j >= 0; j--) {
        for (int col = j; col < m; col++) {
            float ip = 0.0f;
            for (int row = j; row < m; row++) ip += A[(size_t)(row + 1) * n + j] * Q[(size_t)row * m + col];
            tmp[col - j] = ip;
        }
        for (int row = j; row < m; row++)
            for (int col = j; col < m; col++)
                Q[(size_t)row * m + col] = Q[(size_t)row * m + col] - 2.0f * A[(size_t)(row + 1) * n + j] * tmp[col - j];
    }
    free(tmp);
}

/* Cuda/qr.cu:337-426.  W[:,0]=2 v0, Y[:,0]=v0; z_i = 2 (I - W Y^T)[:, i:] v_i[i:];
 * returns dense Q_panel = I - W Y^T (dim x dim, dim = m - global_offset).
 * The reference materialises I - W Y^T every column (O(dim^2 i)); here z is
 * formed as v - W (Y^T v) restricted to the same index range, which is the
 * same arithmetic expression per element up to summation order.              */
void orc_wy_transform(const float* A, float* Qp, int m, int n, int go, int pw) {
    int dim = m - go;
    float* W = (float*)calloc((size_t)dim * pw, sizeof(float));
    float* Y = (float*)calloc((size_t)dim * pw, sizeof(float));
    float* ytv = (float*)malloc((size_t)pw * sizeof(float));
    for (int i = 0; i < dim; i++) {
        Y[(size_t)i * pw] = A[(size_t)(i + go + 1) * n + go];
        W[(size_t)i * pw] = 2 * A[(size_t)(i + go + 1) * n + go];
    }
    for (int i = 1; i < pw; i++) {
        /* v_i lives at rows go+col+1, col in [i, dim) of column go+i (zeros above) */
        for (int idx = 0; idx < i; idx++) {
            float s = 0;
            for (int col = i; col < dim; col++) s += Y[(size_t)col * pw + idx] * A[(size_t)(go + col + 1) * n + go + i];
            ytv[idx] = s;
        }
        for (int row = 0; row < dim; row++) {
            float ip = 0;
            for (int idx = 0; idx < i; idx++) ip += W[(size_t)row * pw + idx] * ytv[idx];
            float vi = (row >= i) ? A[(size_t)(go + row + 1) * n + go + i] : 0.f;
            W[(size_t)row * pw + i] = 2 * (vi - ip);
        }
        for (int idx = 0; idx < dim; idx++)
            Y[(size_t)idx * pw + i] = (idx < i) ? 0.f : A[(size_t)(go + idx + 1) * n + go + i];
    }
    for (int row = 0; row < dim; row++)
        for (int col = 0; col < dim; col++) {
            float ip = 0;
            for (int idx = 0; idx < pw; idx++) ip += W[(size_t)row * pw + idx] * Y[(size_t)col * pw + idx];
            Qp[(size_t)row * dim + col] = (row == col) ? 1 - ip : -ip;
        }
    free(W); free(Y); free(ytv);
}

/* shared body of h_block_qr (qr.cu:1275-1326) and the mixed-precision driver
 * (qr.cu:1049-1226): fp16_q != 0 rounds both operands of the Q update to fp16
 * (qr.cu:1136-1163) and accumulates in fp32 (mmult.cuh:252-300).            */
static void block_qr_dense(float* A, float* Q, int m, int n, int r, int fp16_q) {
    int lambda = 0;
    while (lambda < n) {
        int tau = (lambda + r < n) ? (lambda + r) : n;
        int dim = m - lambda;
        orc_householder_qr(A, m, n, lambda, tau - lambda);
        float* Qp = (float*)malloc((size_t)dim * dim * sizeof(float));
        orc_wy_transform(A, Qp, m, n, lambda, tau - lambda);
        /* A[l:, tau:] = Qp^T A_old[l:, tau:]   (qr.cu:1295-1306 / mmult.cu:236-288) */
        int nk = n - tau;
        if (nk > 0) {
            float* Aold = (float*)malloc((size_t)dim * nk * sizeof(float));
            for (int i = 0; i < dim; i++)
                memcpy(Aold + (size_t)i * nk, A + (size_t)(i + lambda) * n + tau, (size_t)nk * sizeof(float));
            for (int row = 0; row < dim; row++)
                for (int col = 0; col < nk; col++) {
                    float ip = 0;
                    for (int k = 0; k < dim; k++) ip += Qp[(size_t)k * dim + row] * Aold[(size_t)k * nk + col];
                    A[(size_t)(row + lambda) * n + tau + col] = ip;
                }
            free(Aold);
        }
        /* Q[:, l:] = Q_old[:, l:] Qp   (qr.cu:1308-1320 / qr.cu:1109-1207) */
        float* Qold = (float*)malloc((size_t)m * dim * sizeof(float));
        for (int i = 0; i < m; i++)
            for (int j = 0; j < dim; j++) {
                float q = Q[(size_t)i * m + lambda + j];
                Qold[(size_t)i * dim + j] = fp16_q ? orc_round_fp16(q) : q;
            }
        if (fp16_q) for (size_t i = 0; i < (size_t)dim * dim; i++) Qp[i] = orc_round_fp16(Qp[i]);
        for (int row = 0; row < m; row++)
            for (int col = 0; col < dim; col++) {
                float ip = 0;
                for (int k = 0; k < dim; k++) ip += Qold[(size_t)row * dim + k] * Qp[(size_t)k * dim + col];
                Q[(size_t)row * m + lambda + col] = ip;
            }
        free(Qold); free(Qp);
        lambda = tau;
    }
}
void orc_block_qr(float* A, float* Q, int m, int n, int r) { block_qr_dense(A, Q, m, n, r, 0); }
void orc_mixed_precision_block_qr(float* A, float* Q, int m, int n, int r) { block_qr_dense(A, Q, m, n, r, 1); }

/* Cuda/qr.cu:85-100 */
void orc_strip_R_from_A(const float* A, float* R, int m, int n) {
    for (int row = 0; row < m; row++)
        for (int col = 0; col < n; col++)
            R[(size_t)row * n + col] = (row <= col) ? A[(size_t)row * n + col] : 0.f;
}

/* Cuda/qr.cu:102-113 -- fp32 arithmetic, as logged by the reference */
float orc_qr_flops_per_second(float time_ms, int m, int n) {
    float mf = (float)m, nf = (float)n;
    float flops = 4.0f * powf(mf, 2.f) * nf;
    flops -= mf * powf(nf, 2.f);
    flops += powf(nf, 3.f) / 3.0f;
    flops /= time_ms / 1000.0f;
    return flops;
}

/* Cuda/mmult.cu:41-55 (fp32 accumulate) */
static float matrix_norm_f32(const float* A, size_t count) {
    float s = 0;
    for (size_t i = 0; i < count; i++) s += A[i] * A[i];
    return sqrtf(s);
}

/* Cuda/qr.cu:115-135: ||A - Q R|| / ||A||, Q m x m, R m x n, h_mmult fp32 */
float orc_backward_error(const float* A, const float* R, const float* Q, int m, int n) {
    float* D = (float*)malloc((size_t)m * n * sizeof(float));
    for (int row = 0; row < m; row++)
        for (int col = 0; col < n; col++) {
            float ip = 0;
            for (int k = 0; k < m; k++) ip += Q[(size_t)row * m + k] * R[(size_t)k * n + col];
            D[(size_t)row * n + col] = A[(size_t)row * n + col] - ip;
        }
    float e = matrix_norm_f32(D, (size_t)m * n) / matrix_norm_f32(A, (size_t)m * n);
    free(D);
    return e;
}

/* Cuda/qr.cu:137-171: max over entries of the SIGNED value (Q^T Q - I)_ij */
float orc_q_error(const float* Q, int m) {
    float mx = 0;
    for (int row = 0; row < m; row++)
        for (int col = 0; col < m; col++) {
            float ip = 0;
            for (int k = 0; k < m; k++) ip += Q[(size_t)k * m + row] * Q[(size_t)k * m + col];
            float d = ip - ((row == col) ? 1.f : 0.f);
            if (d > mx) mx = d;
        }
    return mx;
}

/* Cuda/qr.cu:173-196 */
float orc_lower_trapezoid_error(const float* R, int m, int n) {
    float s = 0;
    for (int row = 0; row < m; row++)
        for (int col = 0; col < n; col++)
            if (col < row) s += R[(size_t)row * n + col] * R[(size_t)row * n + col];
    return sqrtf(s);
}

/* Cuda/qr.cu:120,127: pass iff err <= 2^-p * m */
int orc_error_passes(float err, int m, int precision_bits) {
    double lim = pow(2.0, -precision_bits);
    return ((double)err <= lim * m) ? 1 : 0;
}

double orc_q_error_fro(const float* Q, int m) {
    double s = 0;
    for (int row = 0; row < m; row++)
        for (int col = 0; col < m; col++) {
            double ip = 0;
            for (int k = 0; k < m; k++) ip += (double)Q[(size_t)k * m + row] * (double)Q[(size_t)k * m + col];
            double d = ip - ((row == col) ? 1.0 : 0.0);
            s += d * d;
        }
    return sqrt(s);
}

double orc_backward_error_f64(const float* A, const float* R, const float* Q, int m, int n) {
    double num = 0, den = 0;
    double* acc = (double*)malloc((size_t)n * sizeof(double));
    for (int row = 0; row < m; row++) {
        for (int col = 0; col < n; col++) acc[col] = 0;
        for (int k = 0; k < m; k++) {
            double q = Q[(size_t)row * m + k];
            if (q == 0) continue;
            const float* Rk = R + (size_t)k * n;
            for (int col = 0; col < n; col++) acc[col] += q * (double)Rk[col];
        }
        for (int col = 0; col < n; col++) {
            double a = A[(size_t)row * n + col], d = a - acc[col];
            num += d * d; den += a * a;
        }
    }
    free(acc);
    return sqrt(num) / sqrt(den);
}

/* ------------------------------------------------------------------ */
/* compact-WY restatement:
 *   Q_panel = I - V T V^T,  T_ii = 2/(v_i^T v_i),  T[:i,i] = -T_ii T[:i,:i] (V[:,:i]^T v_i)  */
void orc_extract_V(const float* A, float* V, int m, int n, int go, int pw) {
    int W = m - go;
    for (int i = 0; i < W; i++)
        for (int j = 0; j < pw; j++)
            V[(size_t)i * pw + j] = (i >= j) ? A[(size_t)(go + i + 1) * n + go + j] : 0.f;
}

static void compact_T_from_V(const float* V, float* T, int W, int pw) {
    double* S = (double*)calloc((size_t)pw * pw, sizeof(double));
    double* Td = (double*)calloc((size_t)pw * pw, sizeof(double));
    for (int i = 0; i < pw; i++)
        for (int j = i; j < pw; j++) {
            double s = 0;
            for (int k = j; k < W; k++) s += (double)V[(size_t)k * pw + i] * (double)V[(size_t)k * pw + j];
            S[(size_t)i * pw + j] = s;
        }
    for (int i = 0; i < pw; i++) {
        double tii = (S[(size_t)i * pw + i] > 0) ? 2.0 / S[(size_t)i * pw + i] : 0.0;
        Td[(size_t)i * pw + i] = tii;
        for (int a = 0; a < i; a++) {
            double s = 0;
            for (int b = a; b < i; b++) s += Td[(size_t)a * pw + b] * S[(size_t)b * pw + i];
            Td[(size_t)a * pw + i] = -tii * s;
        }
    }
    for (size_t i = 0; i < (size_t)pw * pw; i++) T[i] = (float)Td[i];
    free(S); free(Td);
}

void orc_compact_wy_T(const float* A, float* T, int m, int n, int go, int pw, int round_v_fp16) {
    int W = m - go;
    float* V = (float*)malloc((size_t)W * pw * sizeof(float));
    orc_extract_V(A, V, m, n, go, pw);
    if (round_v_fp16) for (size_t i = 0; i < (size_t)W * pw; i++) V[i] = orc_round_fp16(V[i]);
    compact_T_from_V(V, T, W, pw);
    free(V);
}

/* B[W x nc] (ld ldb)  <-  (I - V Tm V^T) B,  Tm = T^T if trans_t else T.
 * precision 1: V, B, and the intermediate are rounded to fp16 before each
 * product, products accumulate in fp32 (order: ascending k).               */
static void apply_compact(const float* V, const float* T, int W, int pw, float* B, int ldb, int nc,
                          int trans_t, int precision) {
    float* W1 = (float*)malloc((size_t)pw * nc * sizeof(float));
    float* W2 = (float*)malloc((size_t)pw * nc * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int c = 0; c < nc; c++) {
        for (int j = 0; j < pw; j++) {
            float s = 0;
            for (int k = j; k < W; k++) {
                float b = B[(size_t)k * ldb + c];
                if (precision) b = orc_round_fp16(b);
                s += V[(size_t)k * pw + j] * b;
            }
            W1[(size_t)j * nc + c] = s;
        }
        for (int j = 0; j < pw; j++) {
            float s = 0;
            if (trans_t) { for (int k = 0; k <= j; k++) s += T[(size_t)k * pw + j] * W1[(size_t)k * nc + c]; }
            else         { for (int k = j; k < pw; k++) s += T[(size_t)j * pw + k] * W1[(size_t)k * nc + c]; }
            W2[(size_t)j * nc + c] = precision ? orc_round_fp16(s) : s;
        }
    }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < W; k++) {
        int jmax = (k < pw - 1) ? k : pw - 1;
        for (int c = 0; c < nc; c++) {
            float s = 0;
            for (int j = 0; j <= jmax; j++) s += V[(size_t)k * pw + j] * W2[(size_t)j * nc + c];
            B[(size_t)k * ldb + c] -= s;
        }
    }
    free(W1); free(W2);
}

void orc_block_qr_compact(float* A, float* Q, int m, int n, int r, int precision) {
    int npan = (n + r - 1) / r;
    float** Vs = (float**)calloc((size_t)npan, sizeof(float*));
    float** Ts = (float**)calloc((size_t)npan, sizeof(float*));
    int p = 0;
    for (int lambda = 0; lambda < n; lambda += r, p++) {
        int tau = (lambda + r < n) ? lambda + r : n;
        int pw = tau - lambda, W = m - lambda;
        orc_householder_qr(A, m, n, lambda, pw);
        float* V = (float*)malloc((size_t)W * pw * sizeof(float));
        float* T = (float*)malloc((size_t)pw * pw * sizeof(float));
        orc_extract_V(A, V, m, n, lambda, pw);
        if (precision) for (size_t i = 0; i < (size_t)W * pw; i++) V[i] = orc_round_fp16(V[i]);
        compact_T_from_V(V, T, W, pw);
        Vs[p] = V; Ts[p] = T;
        if (tau < n) apply_compact(V, T, W, pw, A + (size_t)lambda * n + tau, n, n - tau, 1, precision);
    }
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Q[(size_t)i * m + j] = (i == j) ? 1.f : 0.f;
    for (p = npan - 1; p >= 0; p--) {
        int lambda = p * r;
        int tau = (lambda + r < n) ? lambda + r : n;
        int pw = tau - lambda, W = m - lambda;
        /* Q[l:, l:] <- (I - V T V^T) Q[l:, l:]  (columns < l of those rows are still zero) */
        apply_compact(Vs[p], Ts[p], W, pw, Q + (size_t)lambda * m + lambda, m, m - lambda, 0, precision);
        free(Vs[p]); free(Ts[p]);
    }
    free(Vs); free(Ts);
}

/* ------------------------------------------------------------------ */
/* C++/main.cpp:5-43.  Column-major doubles, square n x n.  w = (u - s*sigma*e1)/||.||
 * with s = -1 if u0 >= 0 else +1 and sigma TRUNCATED TO FLOAT (main.cpp:6);
 * H = I - 2 w w^T embedded at (i,i); Q = Q*H; A = H*A.  No zero-column guard.  */
void orc_qr_factorization_f64(double* A, double* Q, int n) {
    int m = n;
    double* w = (double*)malloc((size_t)m * sizeof(double));
    double* t = (double*)malloc((size_t)m * sizeof(double));
    for (int i = 0; i < n; i++) {
        int len = n - i;
        double s = 0;
        for (int k = 0; k < len; k++) { double x = A[(size_t)i * m + i + k]; s += x * x; }
        float sigma = (float)sqrt(s);
        int sign = (A[(size_t)i * m + i] >= 0) ? -1 : 1;
        for (int k = 0; k < len; k++) w[k] = A[(size_t)i * m + i + k];
        w[0] -= (double)sign * (double)sigma;
        double nw = 0;
        for (int k = 0; k < len; k++) nw += w[k] * w[k];
        nw = sqrt(nw);
        for (int k = 0; k < len; k++) w[k] /= nw;
        /* A = H A : rows i..m-1 of every column */
        for (int c = 0; c < n; c++) {
            double d = 0;
            for (int k = 0; k < len; k++) d += w[k] * A[(size_t)c * m + i + k];
            for (int k = 0; k < len; k++) t[k] = A[(size_t)c * m + i + k] - 2.0 * w[k] * d;
            for (int k = 0; k < len; k++) A[(size_t)c * m + i + k] = t[k];
        }
        /* Q = Q H : columns i..m-1 of every row */
        for (int r = 0; r < m; r++) {
            double d = 0;
            for (int k = 0; k < len; k++) d += Q[(size_t)(i + k) * m + r] * w[k];
            for (int k = 0; k < len; k++) Q[(size_t)(i + k) * m + r] -= 2.0 * d * w[k];
        }
    }
    free(w); free(t);
}

/* ------------------------------------------------------------------ */
/* Cuda/qr.cu:696-776.  Line 1: "<rows> <cols>"; then "<row> <col> <value>",
 * blanks-separated with optional leading blanks; later duplicates overwrite;
 * unspecified entries are zero.                                              */
int orc_read_euroc_jacobian(const char* path, int* rows, int* cols, float** matrix) {
    FILE* f = fopen(path, "r");
    if (!f) return 1;
    char line[512];
    if (!fgets(line, sizeof line, f)) { fclose(f); return 2; }
    if (sscanf(line, "%d %d", rows, cols) != 2 || *rows <= 0 || *cols <= 0) { fclose(f); return 3; }
    float* M = (float*)calloc((size_t)(*rows) * (size_t)(*cols), sizeof(float));
    if (!M) { fclose(f); return 4; }
    while (fgets(line, sizeof line, f)) {
        int r, c; char vs[128];
        if (sscanf(line, " %d %d %127s", &r, &c, vs) != 3) continue;
        if (r < 0 || r >= *rows || c < 0 || c >= *cols) { free(M); fclose(f); return 5; }
        M[(size_t)r * (*cols) + c] = strtof(vs, NULL);   /* std::stof at qr.cu:768 */
    }
    fclose(f);
    *matrix = M;
    return 0;
}

int orc_write_euroc_jacobian(const char* path, int rows, int cols, const float* M) {
    FILE* f = fopen(path, "w");
    if (!f) return 1;
    fprintf(f, "%d %d\n", rows, cols);
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++)
            if (M[(size_t)r * cols + c] != 0.f) fprintf(f, "%d %d %.9g\n", r, c, (double)M[(size_t)r * cols + c]);
    fclose(f);
    return 0;
}
void orc_free(void* p) { free(p); }

/* splitmix64 -> top 24 bits -> [0,1).  Element (i,j) depends only on (seed, i*n+j)
 * so shards of a matrix can be generated independently on each rank.          */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
void orc_generate_random_matrix(float* A, int m, int n, uint64_t seed) {
    uint64_t base = splitmix64(seed);
    for (size_t i = 0; i < (size_t)m * (size_t)n; i++)
        A[i] = (float)(splitmix64(base + i) >> 40) * (1.0f / 16777216.0f);
}
