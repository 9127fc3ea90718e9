// kernels_solve.hip -- gh_solve3: the w x w core of a Gram-Householder leaf (kernels_panel.hip), blocked by 16.
//
// Same inputs and outputs as gh_solve_kernel -- the Gram matrix G of the leaf's rows (fp64) and the leaf's top block
// B (fp32) in;  R, V_top, vdiag, the coefficient matrix C of V_low = A_low C and the flag out -- and the same two
// recursions (a Cholesky chain on N in fp64, the Householder chain on B_top in fp32 with the reference's sign rule,
// Cuda/qr.cu:229-257), but the serial part of a step no longer involves the workgroup:
//   * the Cholesky chain lives in ONE wave (wave 0): the block's 16 x 16 diagonal block and the 16-row panel right of it in
//     registers, 16 steps per block; a value another lane needs travels by DPP row_newbcast inside the wave's 16-lane rows
//     (every row keeps its own copy of the diagonal block) -- no LDS round trip, no barrier, no v_readlane inside a block;
//   * the Householder chain lives in TWO waves (wave 1: the row panel of B and the w vectors, wave 2: the column panel and
//     the v vectors), one block behind, same scheme (v_fmac_f32_dpp: the broadcast is free);
//   * eight update waves keep N (upper 16 x 16 tiles, fp64) and B (16 x 16 tiles, fp32) as MFMA accumulators and apply
//     a block's 16 steps as ONE rank-16 update (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, exact products);
//     the next block's panels first (handed to the chain waves through LDS), the rest while the chains run.
// Two workgroup barriers per block of 16 reflectors instead of one per reflector.
#include "mpqr_internal.h"
#include "panel_dev.h"
#include <type_traits>

namespace mpqr {

constexpr int S3_THREADS = 704;                         // wave 0: Cholesky chain, waves 1-2: Householder chain, waves 3-10: updates
constexpr int S3_BS = 144;                              // row stride (doubles) of a 16 x 128 fp64 panel: rows k, k+1 on disjoint bank halves
constexpr int S3_VS = 144;                              // row stride (floats) of the v_top panel
constexpr int S3_CS = 129;                              // row stride (floats) of the column panel of B
// dynamic LDS (byte offsets): Ws | buf | bpr | bpc | vpan | dummy.  After the loop Ts (TP x TPS floats) reuses buf..vpan.
constexpr unsigned S3_OFF_WS = 0;                                        // [TP][TPS] f32: row k = w^(k) right of the diagonal, column k = v_top^(k) below it
constexpr unsigned S3_OFF_BUF = TP * TPS * 4;                            // [2][16][S3_BS] f64: row panel of N for block b, then the block's Cholesky rows
constexpr unsigned S3_OFF_BPR = S3_OFF_BUF + 2 * 16 * S3_BS * 8;         // [16][128] f32: row panel of B for the next Householder block
constexpr unsigned S3_OFF_BPC = S3_OFF_BPR + 16 * 128 * 4;               // [16][S3_CS] f32: column panel of B (column-major)
constexpr unsigned S3_OFF_VPAN = S3_OFF_BPC + 16 * S3_CS * 4;            // [2][16][S3_VS] f32: v_top of a Householder block
constexpr unsigned S3_OFF_DMY = S3_OFF_VPAN + 2 * 16 * S3_VS * 4;        // [16][S3_VS] f32: where masked stores of the chain waves go
constexpr int S3_LDS_BYTES = S3_OFF_DMY + 16 * S3_VS * 4;
static_assert(S3_OFF_DMY - S3_OFF_BUF >= TP * TPS * 4, "Ts must fit behind Ws");
// LDS addresses of the chain waves are formed as (opaque VGPR base) + (compile-time offset): the offsets from the start of the
// dynamic LDS exceed the 16-bit immediate of the ds instructions, and the compiler otherwise keeps one address register per row
__device__ __forceinline__ unsigned s3_opaque(unsigned o) { asm volatile("" : "+v"(o)); return o; }
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the wave's global stores (s_waitcnt vmcnt(0)),
// ~1-2 us for the R rows the Householder wave has just sent out
__device__ __forceinline__ void s3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define S3F(o) (*(float*)(gh_smem + (o)))
#define S3D(o) (*(double*)(gh_smem + (o)))

#ifdef MPQR_KTRACE
__device__ long g_s3_trace[7][80];
__device__ int g_s3_trace_left = 1;
#define S3T(role, idx) do { if (lane == 0 && (idx) < 80) g_s3_trace[role][idx] = clock64(); } while (0)
#else
#define S3T(role, idx) do {} while (0)
#endif
typedef double double4s __attribute__((ext_vector_type(4)));
typedef float float4s __attribute__((ext_vector_type(4)));

// The chain blocks are straight-line code: no divergent branch inside the 16 steps (a branch splits the block and LLVM then
// sinks every deferred update next to its use).  Per-step scalars are collected in lane kk of a register and written after
// the block; masked LDS stores go to a dummy slot instead.
//
// Software pipelining.  One wave issues one vector instruction per ~4.6 cycles, dependent or not, but a DEPENDENT fp64 /
// transcendental chain runs at ~9 cycles per link: the ~25 scalar links of a step (pivot -> rsqrt -> Newton -> scale) cost
// ~250 cycles when they sit between the steps.  The next step's pivot is final as soon as THIS step has updated row k+1, which
// is the first row it updates; so the next step's scalar chain is issued one link at a time BETWEEN the updates of rows
// k+2, k+3, ... (s3_pin fences each link between two row updates; the row updates are asm volatile and keep their order).
template <class T> __device__ __forceinline__ void s3_pin(T& x) { asm volatile("" : "+v"(x)); }

// ---- Cholesky chain: 16 steps.  dg[r]: lane li holds N[k0 + r][k0 + li] (the diagonal block, one copy per 16-lane row);
// n[r][s]: N[k0 + r][k0 + 16 + lane + 64 s], the NS slots of columns right of the block.
// Step k: p = N[k][k];  N[i][j] -= N[k][i] N[k][j] / p for the block's later rows i (N[k][i] is broadcast BEFORE the row is
// scaled);  row k becomes c_k = N[k][:] / sqrt(p).
// row i of a step: t = N[k][k0 + I] broadcast in the 16-lane row, then row i of the diagonal block and of the NS panel slots.
// One asm statement: the broadcast value lives for NS + 2 instructions.
template <int NS, int I>
__device__ __forceinline__ void s3_chol_row(double& dgi, double& n0, double& n1, double src, double mdg, double m0, double m1) {
    double t;
    if constexpr (NS == 2)
        asm volatile("v_mov_b64_dpp %0, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %1, -%0, %5, %1\n\t"
                     "v_fma_f64 %2, -%0, %6, %2\n\tv_fma_f64 %3, -%0, %7, %3"
                     : "=&v"(t), "+v"(dgi), "+v"(n0), "+v"(n1) : "v"(src), "v"(mdg), "v"(m0), "v"(m1), "n"(I));
    else if constexpr (NS == 1)
        asm volatile("v_mov_b64_dpp %0, %3 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %1, -%0, %4, %1\n\t"
                     "v_fma_f64 %2, -%0, %5, %2"
                     : "=&v"(t), "+v"(dgi), "+v"(n0) : "v"(src), "v"(mdg), "v"(m0), "n"(I));
    else
        asm volatile("v_mov_b64_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %1, -%0, %3, %1"
                     : "=&v"(t), "+v"(dgi) : "v"(src), "v"(mdg), "n"(I));
}
// the scalar chain of one pivot, in links: y = ok ? 1 / sqrt(p) : 0 (fp32 seed, two Newton steps: refine_rsqrt), y2 = y^2
struct S3Piv { double p, hp, y0, t, y, y2, th; float c1; };
constexpr int S3_PIV_LINKS = 10;
template <int J>
__device__ __forceinline__ void s3_piv_link(S3Piv& q) {
    // Round 5, in-line deflation.  A column whose remaining norm ||u_k||^2 is below its threshold th = GH_RHO_MIN max(||a_k||^2, 1e-6 max_j
    // ||a_j||^2) -- numerically dependent on its predecessors; the reference's exactly-zero column (qr.cu:242-244, th = 0) is the extreme
    // case -- is SKIPPED in line, exactly as a zero column is: y = 0 (no Cholesky row, no downdate), the Householder chain sees ok = 0 and
    // leaves v_k = 0, w = 0, R_kk = B[k][k].  What is dropped is the column's remainder below the diagonal, <= sqrt(th) = 1e-4 ||a_k||; what
    // the later columns of the leaf keep too much of is row k's contribution B[k][j]^2 to their norms (the skipped downdate), a relative
    // 1 / (2 rows) on their reflectors' scale -- both inside the 1e-3 tolerance.  Before: flag the leaf, stop the pass, redo the leaf on the
    // column-by-column kernels, restart its block (1.4 - 1.7 x per ill-conditioned leaf, one restart per leaf).  (Clamping the pivot at
    // th instead and carrying on was tried first: the in-kernel updates of the top block then assume a unit reflector that the actual V is
    // not, row k of R comes out wrong by O(1): backward error 2e-3 ... 2e-2.)
    if constexpr (J == 0) { q.hp = -0.5 * q.p; q.c1 = (float)q.p; s3_pin(q.hp); s3_pin(q.c1); }
    else if constexpr (J == 1) { q.c1 = __builtin_amdgcn_rsqf(q.c1); s3_pin(q.c1); }
    else if constexpr (J == 2) { q.y0 = (double)q.c1; s3_pin(q.y0); }
    else if constexpr (J == 3) { q.t = q.y0 * q.y0; s3_pin(q.t); }
    else if constexpr (J == 4) { q.t = fma(q.hp, q.t, 1.5); s3_pin(q.t); }
    else if constexpr (J == 5) { q.y0 = q.y0 * q.t; s3_pin(q.y0); }
    else if constexpr (J == 6) { q.t = q.y0 * q.y0; s3_pin(q.t); }
    else if constexpr (J == 7) { q.t = fma(q.hp, q.t, 1.5); s3_pin(q.t); }
    else if constexpr (J == 8) {
        const bool ok = q.p > q.th && q.p < 1e30;            // false: zero / dependent / cancelled column (skipped in line), or out of range (flagged)
        q.y = ok ? q.y0 * q.t : 0.0; s3_pin(q.y);
    } else if constexpr (J == 9) { q.y2 = q.y * q.y; s3_pin(q.y2); }
}
template <int NS>
__device__ __forceinline__ void s3_chol_block(double (&dg)[16], double (&n)[16][2], int li, double& piv, double thv) {
    S3Piv cur;
    cur.p = dpp_bcast64<0, 0>(dg[0]); cur.th = dpp_bcast64<0, 0>(thv);
    static_range<0, S3_PIV_LINKS>([&](auto J) { s3_piv_link<decltype(J)::value>(cur); });
    static_for16<0>([&](auto KK) {
        constexpr int kk = decltype(KK)::value;
        const double mdg = dg[kk] * cur.y2;
        double m[2] = {0.0, 0.0};
#pragma unroll
        for (int s = 0; s < NS; s++) m[s] = n[kk][s] * cur.y2;
        piv = (li == kk) ? dg[kk] : piv;                     // the lane that owns column k keeps its pivot: checked after the block
        S3Piv nxt;
        if constexpr (kk < 15) {
            s3_chol_row<NS, kk + 1>(dg[kk + 1], n[kk + 1][0], n[kk + 1][1], dg[kk], mdg, m[0], m[1]);
            nxt.p = dpp_bcast64<kk + 1, 1>(dg[kk + 1]); nxt.th = dpp_bcast64<kk + 1, 0>(thv);
            static_for16<kk + 2>([&](auto II) {
                constexpr int i = decltype(II)::value;
                s3_chol_row<NS, i>(dg[i], n[i][0], n[i][1], dg[kk], mdg, m[0], m[1]);
                if constexpr (i - (kk + 2) < S3_PIV_LINKS) s3_piv_link<i - (kk + 2)>(nxt);
            });
            static_range<(14 - kk < S3_PIV_LINKS ? 14 - kk : S3_PIV_LINKS), S3_PIV_LINKS>([&](auto J) { s3_piv_link<decltype(J)::value>(nxt); });
        }
        dg[kk] = (li >= kk) ? dg[kk] * cur.y : 0.0;
#pragma unroll
        for (int s = 0; s < NS; s++) n[kk][s] = n[kk][s] * cur.y;
        if constexpr (kk < 15) cur = nxt;
        __builtin_amdgcn_sched_barrier(0);                   // keep the steps apart
    });
}

// ---- Householder chain (two waves).  dgR[r]: lane li holds B[k0 + r][k0 + li];  dgC[c]: lane li holds B[k0 + li][k0 + c]
// (the diagonal block in both orientations, one copy per 16-lane row, in BOTH waves);  cdg[r]: c_{k0+r}[k0 + li] (fp32).
// Arithmetic as in gh_solve_kernel / the reference's panel: u0 = B[k][k], nu = c_kk, alpha = sgn(u0) nu,
// inv = 1 / ||u + alpha e_1||,  w_j = 2 nu (c_kj + sgn(u0) B[k][j]) inv,  v_i = (B[i][k] + [i == k] alpha) inv,  B -= v (x) w.
// The scalars and the step's vectors on the diagonal block, in links (pipelined like the pivots):
struct S3Hh { float u0, nu, s, alpha, inv, d, i0, q, tnu, e, f, wdg, vdg, wo[2]; int ok; };
constexpr int S3_HH_LINKS = 11;
template <int KK, int J, int NS, int ROLE /* 1: row-panel wave (wo), 2: column-panel wave (vo) */>
__device__ __forceinline__ void s3_hh_link(S3Hh& q, const float (&dgR)[16], const float (&dgC)[16], const float (&cdg)[16],
                                           const float (&pan)[16][2], const float (&cfo)[16][2], int li) {
    if constexpr (J == 0) {
        q.s = (q.u0 >= 0.f) ? 1.f : -1.f; q.e = q.nu + fabsf(q.u0); q.tnu = 2.f * q.nu; s3_pin(q.s); s3_pin(q.e); s3_pin(q.tnu);
    } else if constexpr (J == 1) { q.d = q.ok ? q.tnu * q.e : 1.f; q.alpha = q.s * q.nu; s3_pin(q.d); s3_pin(q.alpha); }
    else if constexpr (J == 2) { q.i0 = __builtin_amdgcn_rsqf(q.d); q.q = 0.5f * q.d; s3_pin(q.i0); s3_pin(q.q); }
    else if constexpr (J == 3) { q.q = q.q * q.i0; q.e = q.s * dgR[KK]; s3_pin(q.q); s3_pin(q.e); }
    else if constexpr (J == 4) { q.q = q.q * q.i0; q.e = cdg[KK] + q.e; s3_pin(q.q); s3_pin(q.e); }
    else if constexpr (J == 5) { q.q = 1.5f - q.q; q.e = q.tnu * q.e; s3_pin(q.q); s3_pin(q.e); }
    else if constexpr (J == 6) { q.inv = q.ok ? q.i0 * q.q : 0.f; q.f = dgC[KK] + (li == KK ? q.alpha : 0.f); s3_pin(q.inv); s3_pin(q.f); }
    else if constexpr (J == 7) { q.wdg = (li > KK) ? q.e * q.inv : 0.f; q.vdg = (li >= KK) ? q.f * q.inv : 0.f; s3_pin(q.wdg); s3_pin(q.vdg); }
    else if constexpr (J == 8) {
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++) {
            if constexpr (ROLE == 1) q.wo[s2] = cfo[KK][s2] + q.s * pan[KK][s2]; else q.wo[s2] = pan[KK][s2] + 0.f;
            s3_pin(q.wo[s2]);
        }
    } else if constexpr (J == 9) {
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++) { if constexpr (ROLE == 1) q.wo[s2] = q.tnu * q.wo[s2]; s3_pin(q.wo[s2]); }
    } else if constexpr (J == 10) {
#pragma unroll
        for (int s2 = 0; s2 < NS; s2++) { q.wo[s2] = q.wo[s2] * q.inv; s3_pin(q.wo[s2]); }
    }
}
// One block of either Householder wave.  ROLE 1 (row panel): pan[r][s] = B[k0 + r][k0 + 16 + lane + 64 s], cfo likewise the
// Cholesky rows; leaves Ws[k][j] = w_j (j > k) and, in lane kk of sg / vd / td, sgn(u0), v_kk, inv of step kk (td = 1 for a
// flagged column).  ROLE 2 (column panel): pan[c][s] = B[k0 + 16 + lane + 64 s][k0 + c]; leaves Ws[i][k] = v_i (i > k: the lower
// triangle of Ws is free) and vp[kk][i] = v_i for the rows below the block (the MFMA updates read nothing else).
template <int NS, int ROLE>
__device__ __forceinline__ void s3_hh_block(float (&dgR)[16], float (&dgC)[16], float (&pan)[16][2], const float (&cdg)[16],
                                            const float (&cfo)[16][2], int okv, int lane, unsigned a_dg, const unsigned (&a_o)[2],
                                            const unsigned (&p_o)[2], unsigned dmy, float& sg, float& vd, float& td) {
    const int li = lane & 15;
    S3Hh cur;
    cur.u0 = dpp_bcast32<0, 0>(dgR[0]); cur.nu = dpp_bcast32<0, 0>(cdg[0]);
    cur.ok = __builtin_bit_cast(int, dpp_bcast32<0, 0>(__builtin_bit_cast(float, okv)));
    static_range<0, S3_HH_LINKS>([&](auto J) { s3_hh_link<0, decltype(J)::value, NS, ROLE>(cur, dgR, dgC, cdg, pan, cfo, li); });
    static_for16<0>([&](auto KK) {
        constexpr int kk = decltype(KK)::value;
        float nvdg = -cur.vdg, nwdg = -cur.wdg;
        s3_pin(nvdg); s3_pin(nwdg);
        // stores: Ws row k (w) or column k (v), and the v panel for the MFMA updates
        if constexpr (ROLE == 1) {
            const unsigned pw = (lane < 16 && li > kk) ? a_dg : dmy;
            S3F(pw + kk * TPS * 4) = cur.wdg;
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++) S3F(a_o[s2] + kk * TPS * 4) = cur.wo[s2];
            const bool me = lane == kk;
            sg = me ? cur.s : sg;
            vd = me ? (cur.u0 + cur.alpha) * cur.inv : vd;   // v_top[k]
            td = me ? (cur.ok ? cur.inv : 1.f) : td;
        } else {
            const unsigned pv = (lane < 16 && li > kk) ? a_dg : dmy;
            S3F(pv + kk * 4) = cur.vdg;
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++) { S3F(a_o[s2] + kk * 4) = cur.wo[s2]; S3F(p_o[s2] + kk * S3_VS * 4) = cur.wo[s2]; }
        }
        S3Hh nxt;
        auto row = [&](auto RR, auto NOPT) {
            constexpr int r = decltype(RR)::value;
            constexpr int nop = decltype(NOPT)::value;
            if constexpr (ROLE == 1) {
                fmac_dpp<r, nop>(dgR[r], nvdg, cur.wdg);
#pragma unroll
                for (int s2 = 0; s2 < NS; s2++) fmac_dpp<r, 0>(pan[r][s2], nvdg, cur.wo[s2]);
                fmac_dpp<r, 0>(dgC[r], nwdg, cur.vdg);
            } else {
                fmac_dpp<r, nop>(dgC[r], nwdg, cur.vdg);
#pragma unroll
                for (int s2 = 0; s2 < NS; s2++) fmac_dpp<r, 0>(pan[r][s2], nwdg, cur.wo[s2]);
                fmac_dpp<r, 0>(dgR[r], nvdg, cur.wdg);
            }
        };
        if constexpr (kk < 15) {
            row(std::integral_constant<int, kk + 1>{}, std::integral_constant<int, 1>{});
            nxt.u0 = dpp_bcast32<kk + 1, 1>(dgR[kk + 1]); nxt.nu = dpp_bcast32<kk + 1, 0>(cdg[kk + 1]);
            nxt.ok = __builtin_bit_cast(int, dpp_bcast32<kk + 1, 0>(__builtin_bit_cast(float, okv)));
            static_for16<kk + 2>([&](auto II) {
                constexpr int i = decltype(II)::value;
                row(II, std::integral_constant<int, 0>{});
                if constexpr (i - (kk + 2) < S3_HH_LINKS) s3_hh_link<kk + 1, i - (kk + 2), NS, ROLE>(nxt, dgR, dgC, cdg, pan, cfo, li);
            });
            static_range<(14 - kk < S3_HH_LINKS ? 14 - kk : S3_HH_LINKS), S3_HH_LINKS>(
                [&](auto J) { s3_hh_link<kk + 1, decltype(J)::value, NS, ROLE>(nxt, dgR, dgC, cdg, pan, cfo, li); });
            cur = nxt;
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// ---- one block of a chain wave: panels in from LDS, the 16 steps, results out.  NS = slots of 64 columns / rows beyond the
// diagonal block (a template parameter, so that each case allocates only the registers it needs)
template <int NS>
__device__ __forceinline__ void s3_ch_round(unsigned pb, int k0, int nrest, int lane, const double* thr, int* cmask, int& bad, int& ndefl) {
    asm volatile("" : "+v"(lane));                           // per-lane masks are recomputed per round, not hoisted out of the loop and spilled
    const int li = lane & 15;
    const unsigned o_dg = s3_opaque(pb + (k0 + li) * 8);
    const unsigned o_n = s3_opaque(pb + (k0 + 16 + (lane < nrest ? lane : 0)) * 8);   // slot 1: + 64 columns, inside the padded row
    double dg[16], n[16][2];
#pragma unroll
    for (int rr = 0; rr < 16; rr++) dg[rr] = S3D(o_dg + rr * S3_BS * 8);
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const bool live = lane + 64 * s < nrest;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) { const double x = S3D(o_n + (rr * S3_BS + 64 * s) * 8); n[rr][s] = live ? x : 0.0; }
    }
    double piv = 1.0;
    const double th = thr[k0 + li];
    S3T(4, 20 + 2 * (k0 >> 4));
    s3_chol_block<NS>(dg, n, li, piv, th);
    S3T(4, 21 + 2 * (k0 >> 4));
    const int okl = (piv > th && piv < 1e30) ? 1 : 0;        // lane li: the pivot of step li (the same test as inside the chain)
    bad |= (piv != piv || piv >= 1e30 || piv < -th - 1e-30) ? 1 : 0;   // flagged: NaN / Inf / out-of-range or a pivot that cancelled far below zero -- not a zero or dependent column
    ndefl += (lane < 16 && !okl && th > 1e-30) ? 1 : 0;      // deflated columns (reported: mpqr_timings.n_deflated_columns; exactly-zero columns, th = 0, are not counted)
    const unsigned q_dg = s3_opaque(pb + (k0 + li) * 8), q_n = s3_opaque(pb + (k0 + 16 + (lane < nrest ? lane : 0)) * 8);
    if (lane < 16) {
        cmask[k0 + li] = okl;                               // lane kk holds step kk's flag
#pragma unroll
        for (int rr = 0; rr < 16; rr++) S3D(q_dg + rr * S3_BS * 8) = dg[rr];
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (lane + 64 * s < nrest) {
#pragma unroll
            for (int rr = 0; rr < 16; rr++) S3D(q_n + (rr * S3_BS + 64 * s) * 8) = n[rr][s];
        }
    }
}
template <int NS>
__device__ __forceinline__ void s3_hhr_round(int b, int k0, int nrest, int lane, int w, const LeafArgs& a, const int* cmask,
                                             float* sgn, float* vdl, float* tdiag, float skip_row_max2, int* lflag) {
    asm volatile("" : "+v"(lane));
    const int li = lane & 15;
    const int lz = lane < nrest ? lane : 0;                  // lanes beyond the leaf read their slot's first column and are zeroed
    const unsigned o_r = s3_opaque(S3_OFF_BPR + (k0 + li) * 4), o_c = s3_opaque(S3_OFF_BPC + (k0 + li) * 4);
    const unsigned o_cd = s3_opaque(S3_OFF_BUF + (b & 1) * (16 * S3_BS * 8) + (k0 + li) * 8);
    const unsigned o_ro = s3_opaque(S3_OFF_BPR + (k0 + 16 + lz) * 4);
    const unsigned o_co = s3_opaque(S3_OFF_BUF + (b & 1) * (16 * S3_BS * 8) + (k0 + 16 + lz) * 8);
    float dgR[16], dgC[16], cdg[16], br[16][2], cfo[16][2];
#pragma unroll
    for (int rr = 0; rr < 16; rr++) {
        dgR[rr] = S3F(o_r + rr * 128 * 4);
        dgC[rr] = S3F(o_c + rr * S3_CS * 4);
        cdg[rr] = (float)S3D(o_cd + rr * S3_BS * 8);
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const bool live = lane + 64 * s < nrest;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
            const float x = S3F(o_ro + (rr * 128 + 64 * s) * 4);
            const float y = (float)S3D(o_co + (rr * S3_BS + 64 * s) * 8);
            br[rr][s] = live ? x : 0.f; cfo[rr][s] = live ? y : 0.f;
        }
    }
    const int okv = cmask[k0 + li];
    const unsigned dmy = s3_opaque(S3_OFF_DMY + lane * 4);
    const unsigned w_dg = s3_opaque(S3_OFF_WS + (k0 * TPS + k0 + li) * 4);
    unsigned w_o[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int col = k0 + 16 + lane + 64 * s;
        w_o[s] = s3_opaque(col < GW ? S3_OFF_WS + (k0 * TPS + col) * 4 : S3_OFF_DMY + lane * 4);
    }
    float sg = 1.f, vd = 0.f, td = 1.f;
    S3T(5, 20 + 2 * (k0 >> 4));
    s3_hh_block<NS, 1>(dgR, dgC, br, cdg, cfo, okv, lane, w_dg, w_o, w_o, dmy, sg, vd, td);
    S3T(5, 21 + 2 * (k0 >> 4));
    if (lane < 16) { sgn[k0 + lane] = sg; vdl[k0 + lane] = vd; tdiag[k0 + lane] = td; }
    // Skipped steps (ok = 0: a zero or dependent column, H_k = I): row k of R is row k of B AS IT STANDS -- the later steps of the block never
    // touch it -- not -sgn(u0) c_k (store_r, which leaves such rows alone).  It lives in this wave's registers; rare, so the stores sit here.
    const unsigned skipped = (unsigned)(__ballot(okv == 0 && lane < 16 && k0 + li < w) & 0xffffu);
    if (skipped) {
        static_for16<0>([&](auto KK) {
            constexpr int kk = decltype(KK)::value;
            if ((skipped >> kk) & 1u) {
                float* rowp = a.A + (long)(a.c0 + k0 + kk) * a.lda + a.c0;
                float r2 = 0.f;                              // largest squared entry of the dead row right of the diagonal
                if (lane < 16 && li >= kk && k0 + li < w) { rowp[k0 + li] = dgR[kk]; if (li > kk) r2 = dgR[kk] * dgR[kk]; }
#pragma unroll
                for (int s = 0; s < NS; s++) { const int col = k0 + 16 + lane + 64 * s; if (lane + 64 * s < nrest && col < w) { rowp[col] = br[kk][s]; r2 = fmaxf(r2, br[kk][s] * br[kk][s]); } }
                // The skipped step does not take its dead row out of the Gram matrix N (the Cholesky chain ran ahead and never sees B's rows):
                // the later columns of the leaf keep B[k][j]^2 too much norm, a relative B[k][j]^2 / ||u_j||^2 on their reflectors.  Measured
                // (tools/defl_rows.py): backward-error excess ~ 2.7 x that ratio per skipped column -- fine for sparse Jacobians and for leaves of
                // >= ~6000 rows, 1.3e-3 ... 4.6e-3 in all for dense leaves of 1500 - 4000 rows.  So the skip is kept only while the row is small
                // against the leaf (GH_SKIP_ROW_MAX x max_j ||a_j||^2); otherwise the leaf is flagged and redone on the column-by-column kernels
                // as before round 5.  (r2 is the row's LARGEST squared entry, ~6 x a typical one over 120 columns: excess ~ 0.4 r2 / max ||a_j||^2.)
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o));
                if (lane == 0 && r2 > skip_row_max2) *lflag = 1;
            }
        });
    }
}
template <int NS>
__device__ __forceinline__ void s3_hhc_round(int b, int k0, int nrest, int lane, const int* cmask) {
    asm volatile("" : "+v"(lane));
    const int li = lane & 15;
    const int lz = lane < nrest ? lane : 0;
    const unsigned o_r = s3_opaque(S3_OFF_BPR + (k0 + li) * 4), o_c = s3_opaque(S3_OFF_BPC + (k0 + li) * 4);
    const unsigned o_cd = s3_opaque(S3_OFF_BUF + (b & 1) * (16 * S3_BS * 8) + (k0 + li) * 8);
    const unsigned o_co = s3_opaque(S3_OFF_BPC + (k0 + 16 + lz) * 4);
    float dgR[16], dgC[16], cdg[16], bc[16][2];
#pragma unroll
    for (int rr = 0; rr < 16; rr++) {
        dgR[rr] = S3F(o_r + rr * 128 * 4);
        dgC[rr] = S3F(o_c + rr * S3_CS * 4);
        cdg[rr] = (float)S3D(o_cd + rr * S3_BS * 8);
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const bool live = lane + 64 * s < nrest;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) { const float x = S3F(o_co + (rr * S3_CS + 64 * s) * 4); bc[rr][s] = live ? x : 0.f; }
    }
    const int okv = cmask[k0 + li];
    const unsigned dmy = s3_opaque(S3_OFF_DMY + lane * 4);
    const unsigned v_dg = s3_opaque(S3_OFF_WS + ((k0 + li) * TPS + k0) * 4);
    unsigned v_o[2], p_o[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const int row = k0 + 16 + lane + 64 * s;
        v_o[s] = s3_opaque(row < GW ? S3_OFF_WS + (row * TPS + k0) * 4 : S3_OFF_DMY + lane * 4);
        p_o[s] = s3_opaque(row < GW ? S3_OFF_VPAN + ((b & 1) * 16 * S3_VS + row) * 4 : S3_OFF_DMY + lane * 4);
    }
    float sg = 1.f, vd = 0.f, td = 1.f;                      // (row-panel wave only)
    S3T(6, 20 + 2 * (k0 >> 4));
    s3_hh_block<NS, 2>(dgR, dgC, bc, cdg, bc /* unused */, okv, lane, v_dg, v_o, p_o, dmy, sg, vd, td);
    S3T(6, 21 + 2 * (k0 >> 4));
}

// rank-16 update of one 16 x 16 tile of N:  N[I0+i][J0+j] -= sum_k c[k][I0+i] c[k][J0+j]   (cb: [16][S3_BS] fp64)
// v_mfma_f64_16x16x4_f64: A[i][k] from lane i + 16 k, B[k][j] from lane j + 16 k, D[i][j] in lane j + 16 (i % 4), element i / 4
__device__ __forceinline__ void s3_n_update(double4s& acc, const double* cb, int I0, int J0, int li, int lk) {
    double av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { av[q] = -cb[(4 * q + lk) * S3_BS + I0 + li]; bv[q] = cb[(4 * q + lk) * S3_BS + J0 + li]; }
#pragma unroll
    for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
}
// rank-16 update of one 16 x 16 tile of B:  B[I0+i][J0+j] -= sum_k v_k[I0+i] w_k[J0+j]   (vp: [16][S3_VS], wrow = Ws + k0 * TPS)
// v_mfma_f32_16x16x4_f32: A[i][k] from lane i + 16 k, B[k][j] from lane j + 16 k, D[i][j] in lane j + 16 (i / 4), element i % 4
__device__ __forceinline__ void s3_b_update(float4s& acc, const float* vp, const float* wrow, int I0, int J0, int li, int lk) {
    float av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { av[q] = -vp[(4 * q + lk) * S3_VS + I0 + li]; bv[q] = wrow[(4 * q + lk) * TPS + J0 + li]; }
#pragma unroll
    for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc, 0, 0, 0);
}

__global__ __launch_bounds__(S3_THREADS) void gh_solve3_kernel(LeafArgs a, const double* __restrict__ G, float* __restrict__ Cv,
                                                               int* __restrict__ flag, SolveWait ws) {
    if (ws.stamp && threadIdx.x == 0) ws.stamp[0] = __builtin_amdgcn_s_memrealtime();
    float* Ws = (float*)(gh_smem + S3_OFF_WS);
    double* buf = (double*)(gh_smem + S3_OFF_BUF);
    float* bpr = (float*)(gh_smem + S3_OFF_BPR);
    float* bpc = (float*)(gh_smem + S3_OFF_BPC);
    float* vpan = (float*)(gh_smem + S3_OFF_VPAN);
    float* Ts = (float*)(gh_smem + S3_OFF_BUF);           // [TP][TPS] after the loop: the inverse
    __shared__ float vdl[GW], tdiag[GW], sgn[GW];
    __shared__ int cmask[GW], lflag;
    __shared__ float skip_row_max2;                        // a skipped step's dead row may hold entries up to sqrt of this (GH_SKIP_ROW_MAX x the leaf's largest ||a_j||^2)
    __shared__ double thr[GW];                            // rho thresholds GH_RHO_MIN ||a_j||^2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (round 4: swapping the roles of physical waves 1 and 3 -- wave 3 has one SIMD partner where waves 0-2 have two, if waves i, i + 4,
    //  i + 8 share a SIMD -- changed nothing: 61.5 us alone either way)
    const int li = lane & 15, lk = lane >> 4;
    const int w = a.c1 - a.c0, off = a.c0 - a.cb;
    const int nb = (w + 15) / 16, ncol = 16 * nb;
    KT_DECL; KT();
    // G and B in leaf coordinates, padded with the identity / zero: the padding decouples
    // (the loads are unconditional -- clamped index, value selected afterwards -- so that a lane's 20 - 50 of them are in flight together:
    //  behind a branch each one waited for the previous, ~20 k cycles of set-up per launch; tools/ktrace_solve.sh)
    auto Gw = [&](int i, int j) -> double {
        const double g = G[(off + min(i, w - 1)) * GW + off + min(j, w - 1)];
        return (i < w && j < w) ? g : (i == j ? 1.0 : 0.0);
    };
    static_assert((TP * TPS) % 4 == 0 && S3_OFF_WS % 16 == 0 && S3_OFF_BUF % 16 == 0, "16-byte fills of Ws / Ts");
    for (int e = tid; e < TP * TPS / 4; e += S3_THREADS) ((float4*)Ws)[e] = make_float4(0.f, 0.f, 0.f, 0.f);   // (16-byte stores: 6 rounds instead of 24)
    if (tid < GW) { vdl[tid] = 0.f; tdiag[tid] = 1.f; sgn[tid] = 1.f; cmask[tid] = 1; }
    if (tid == 0) lflag = 0;
    KT();

    // ---- output helpers of the update waves (also used after the triangular inverse, below)
    float rc_tail[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    // R[k][j] = -sgn(u0_k) c_kj (j >= k): wave u sends out rows 2u, 2u + 1 of every block.  The Cholesky rows are read (fp32)
    // in the round the Householder waves work on the block; the signs exist one barrier later, so the stores wait in
    // registers until the next round
    auto store_r = [&](int kb, const float (&rc)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int k = kb + j;
            if (k < w && cmask[k]) {                         // (a skipped step's row was stored by the Householder wave: s3_hhr_round)
                const float ns = -sgn[k];
                float* rowp = a.A + (long)(a.c0 + k) * a.lda + a.c0;
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    const int col = lane + 64 * s2;
                    if (col >= k && col < w) rowp[col] = ns * rc[j][s2];
                }
            }
        }
    };
    // V_top of block b (Ws[i][k], i > k; diagonal in vdl): A below the diagonal, the fp16 copies V and V^T -- one round after
    // the Householder waves finished the block, 16-byte stores where the leaf is 8-aligned (else: after the loop, all threads)
    const bool vec_out = (w & 7) == 0 && (a.c0 & 7) == 0;
    auto store_v = [&](int b, int u) {
        typedef half_t half4v __attribute__((ext_vector_type(4)));
        typedef half_t half8v __attribute__((ext_vector_type(8)));
        if (!vec_out) return;
        {
            const int id = u * 64 + lane, i = 16 * b + (id >> 2), k = 16 * b + 4 * (id & 3);   // 4 consecutive k: a row segment of A and Vh
            if (i < w && k < w && i >= k) {
                float t4[4]; half4v hv;
#pragma unroll
                for (int q = 0; q < 4; q++) t4[q] = Ws[i * TPS + k + q];
                float* dst = &a.A[(long)(a.c0 + i) * a.lda + a.c0 + k];
                if (i > k + 3) *(float4*)dst = make_float4(t4[0], t4[1], t4[2], t4[3]);
                else {
#pragma unroll
                    for (int q = 0; q < 4; q++) if (i > k + q) dst[q] = t4[q];
                }
                const float vd = vdl[i];
#pragma unroll
                for (int q = 0; q < 4; q++) hv[q] = i > k + q ? (half_t)t4[q] : (i == k + q ? (half_t)vd : (half_t)0.f);
                *(half4v*)&a.Vh[(long)(a.c0 + i) * a.ldvh + a.c0 + k] = hv;
            }
        }
        if (u < 4) {
            const int id = u * 64 + lane, k = 16 * b + (id >> 4), i = 8 * (id & 15);            // 8 consecutive i: a row segment of V^T
            if (k < w && i < w && i + 7 >= k) {
                half8v hv;
                const float vd = vdl[k];
#pragma unroll
                for (int q = 0; q < 8; q++) hv[q] = (i + q > k) ? (half_t)Ws[(i + q) * TPS + k] : (i + q == k ? (half_t)vd : (half_t)0.f);
                *(half8v*)&a.Vt[(long)(a.c0 + k) * a.ldvt + a.c0 + i] = hv;
            }
        }
    };
    // Every role runs the same sequence of workgroup barriers (one after the set-up, two per round); the roles' code paths are
    // separate so that each gets its own register allocation.
    if (wave < 3) __builtin_amdgcn_s_setprio(3);           // the chains first: their panel loads compete with the update waves' operand reads
    if (wave == 0) {
        // ================================================= Cholesky chain
        int bad = 0, ndefl = 0;
        {   // all 34 loads of a lane first, then the LDS stores: written as load -> store pairs the compiler kept them in program order and
            // the wave paid one memory latency per load (G was just written by gh_reduce on other CUs: misses) -- 21 k of the kernel's
            // ~155 k cycles before the first round could start (in-kernel stamps, tools/ktrace_solve.sh)
            double gv[2][17];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int col = lane + 64 * s;
                gv[s][16] = Gw(col, col);
#pragma unroll
                for (int rr = 0; rr < 16; rr++) gv[s][rr] = Gw(rr, col);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            double gmax = fmax(lane < ncol ? gv[0][16] : 0.0, lane + 64 < ncol ? gv[1][16] : 0.0);      // max_j ||a_j||^2 over the leaf's columns
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, o));
            if (lane == 0) skip_row_max2 = (float)(GH_SKIP_ROW_MAX * gmax);
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int col = lane + 64 * s;
                // pivot floor of column j: rho threshold x ||a_j||^2 (all leaf rows), never below 1e-6 of the leaf's largest column; an
                // EXACTLY zero column keeps floor 0: its pivot is 0, the step is skipped in line (v = 0, R_kk = 0) as the reference does (qr.cu:242-244)
                thr[col] = gv[s][16] == 0.0 ? 1e-30 : fmax(GH_RHO_MIN * fmax(gv[s][16], 1e-6 * gmax), 1e-30);
#pragma unroll
                for (int rr = 0; rr < 16; rr++) buf[rr * S3_BS + col] = gv[s][rr];   // row panel 0
            }
        }
        KT();
        s3_barrier();
        KT();
        for (int r = 0; r <= nb; r++) {
            S3T(0, 2 * r);
            if (r < nb) {
                const int k0 = 16 * r;
                const int nrest = ncol - (k0 + 16);           // columns right of the block: 0, 1 or 2 slots of 64
                const unsigned pb = S3_OFF_BUF + (r & 1) * (16 * S3_BS * 8);
                if (nrest > 64) s3_ch_round<2>(pb, k0, nrest, lane, thr, cmask, bad, ndefl);
                else if (nrest > 0) s3_ch_round<1>(pb, k0, nrest, lane, thr, cmask, bad, ndefl);
                else s3_ch_round<0>(pb, k0, nrest, lane, thr, cmask, bad, ndefl);
                if (bad) lflag = 1;
            }
            S3T(0, 2 * r + 1);
            s3_barrier();
            s3_barrier();
        }
        // columns deflated by the pivot clamp (s3_piv_link): counted in mapped host memory -- nothing is written on the normal path
        int tot = (lane < 16 && w > 0) ? ndefl : 0;
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
        if (lane == 0 && tot > 0 && a.deflword) __hip_atomic_fetch_add(a.deflword, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (wave == 1) {
        // ================================================= Householder chain, one block behind: wave 1 the row panel (w), wave 2
        // the column panel (v); both carry the diagonal block
        s3_barrier();
        for (int r = 0; r <= nb; r++) {
            S3T(1, 2 * r);
            if (r >= 1) {
                const int b = r - 1, k0 = 16 * b;
                const int nrest = ncol - (k0 + 16);
                if (nrest > 64) s3_hhr_round<2>(b, k0, nrest, lane, w, a, cmask, sgn, vdl, tdiag, skip_row_max2, &lflag);
                else if (nrest > 0) s3_hhr_round<1>(b, k0, nrest, lane, w, a, cmask, sgn, vdl, tdiag, skip_row_max2, &lflag);
                else s3_hhr_round<0>(b, k0, nrest, lane, w, a, cmask, sgn, vdl, tdiag, skip_row_max2, &lflag);
            }
            S3T(1, 2 * r + 1);
            s3_barrier();
            s3_barrier();
        }
    } else if (wave == 2) {
        s3_barrier();
        for (int r = 0; r <= nb; r++) {
            S3T(2, 2 * r);
            if (r >= 1) {
                const int b = r - 1, k0 = 16 * b;
                const int nrest = ncol - (k0 + 16);
                if (nrest > 64) s3_hhc_round<2>(b, k0, nrest, lane, cmask);
                else if (nrest > 0) s3_hhc_round<1>(b, k0, nrest, lane, cmask);
                else s3_hhc_round<0>(b, k0, nrest, lane, cmask);
            }
            S3T(2, 2 * r + 1);
            s3_barrier();
            s3_barrier();
        }
    } else {
        // ================================================= update waves: N and B as MFMA accumulators
        // wave u owns the tiles (I, J) with (I + J) % 8 == u: every row / column panel is spread over all eight waves.
        // B: all eight, Bacc[I], J = (u - I) & 7.  N: the upper ones (J >= I), at most five: slot t -> I = t (t <= u/2),
        // t + u - u/2 beyond.
        const int u = wave - 3, uh = u >> 1;
        double4s Nacc[5];
        float4s Bacc[8];
        int NI[5], NJ[5];
        bool NV[5]; int NP[5][4]; bool BV[8][4];
#pragma unroll
        for (int t = 0; t < 5; t++) {
            const int I = (t <= uh) ? t : t + (u - uh);
            const int J = (u - I) & 7;
            const bool valid = I < 8 && J >= I && J < nb;
            NI[t] = valid ? I : 99; NJ[t] = J;                  // 99: never selected
            NV[t] = valid;
#pragma unroll
            for (int e = 0; e < 4; e++) {                   // raw (clamped) load now, padding / masking AFTER the first barrier: a select here would wait for the load
                const int gi = 16 * min(I, 7) + lk + 4 * e, gj = 16 * J + li;
                NP[t][e] = (gi < w && gj < w) ? 2 : (gi == gj ? 1 : 0);        // 2: the loaded value, 1: identity padding, 0: zero
                Nacc[t][e] = G[(off + min(gi, w - 1)) * GW + off + min(gj, w - 1)];
            }
        }
#pragma unroll
        for (int I = 0; I < 8; I++) {
            const int J = (u - I) & 7;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int bi = 16 * I + 4 * lk + e, bj = 16 * J + li;
                BV[I][e] = I < nb && J < nb && bi < w && bj < w;
                Bacc[I][e] = a.A[(long)(a.c0 + min(bi, w - 1)) * a.lda + a.c0 + min(bj, w - 1)];
            }
        }
        // (no wait for these loads here: the first round's Cholesky chain needs wave 0's panel only and runs while they are still in
        //  flight; the accumulators are masked -- their first use -- behind the barrier, two rounds before the top block is overwritten with R)
        s3_barrier();
#pragma unroll
        for (int t = 0; t < 5; t++)
#pragma unroll
            for (int e = 0; e < 4; e++) Nacc[t][e] = !NV[t] ? 0.0 : (NP[t][e] == 2 ? Nacc[t][e] : (NP[t][e] == 1 ? 1.0 : 0.0));
#pragma unroll
        for (int I = 0; I < 8; I++)
#pragma unroll
            for (int e = 0; e < 4; e++) Bacc[I][e] = BV[I][e] ? Bacc[I][e] : 0.f;
        float rc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        for (int r = 0; r <= nb; r++) {
            if (u == 7) S3T(3, 4 * r);
            if (r >= 2) { store_r(16 * (r - 2) + 2 * u, rc); store_v(r - 2, u); }
            // ---- segment 1: the updates nobody is waiting for
#ifndef S3_DBG_NOUPD
            if (r >= 1 && r + 1 < nb) {                    // Cholesky block r-1 onto the tiles below the next panel
                const double* cb = buf + ((r - 1) & 1) * (16 * S3_BS);
#pragma unroll
                for (int t = 0; t < 5; t++)
                    if (NI[t] >= r + 1 && NI[t] < 8) s3_n_update(Nacc[t], cb, 16 * NI[t], 16 * NJ[t], li, lk);
            }
            if (r >= 2 && r < nb) {                        // Householder block r-2 onto the tiles beyond the panels of block r-1
                const float* vp = vpan + ((r - 2) & 1) * (16 * S3_VS);
                const float* wrow = Ws + 16 * (r - 2) * TPS;
#pragma unroll
                for (int I = 0; I < 8; I++) {
                    const int J = (u - I) & 7;
                    if (I >= r && J >= r && I < nb && J < nb) s3_b_update(Bacc[I], vp, wrow, 16 * I, 16 * J, li, lk);
                }
            }
#endif
            if (r >= 1) {
                const double* cb = buf + ((r - 1) & 1) * (16 * S3_BS);
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int s2 = 0; s2 < 2; s2++) rc[j][s2] = (float)cb[(2 * u + j) * S3_BS + lane + 64 * s2];
            }
            if (u == 7) S3T(3, 4 * r + 1);
            s3_barrier();
            if (u == 7) S3T(3, 4 * r + 2);
            // ---- segment 2: the panels the chains need next
            if (r + 1 < nb) {                              // row panel r+1 of N: Cholesky block r applied, handed to wave 0
                const double* cb = buf + (r & 1) * (16 * S3_BS);
                double* pn = buf + ((r + 1) & 1) * (16 * S3_BS);
#pragma unroll
                for (int t = 0; t < 5; t++)
                    if (NI[t] == r + 1) {
                        s3_n_update(Nacc[t], cb, 16 * NI[t], 16 * NJ[t], li, lk);
#pragma unroll
                        for (int e = 0; e < 4; e++) pn[(lk + 4 * e) * S3_BS + 16 * NJ[t] + li] = Nacc[t][e];
                    }
            }
            if (r < nb) {                                  // panels r of B: Householder block r-1 applied, handed to wave 1
                const float* vp = vpan + ((r - 1) & 1) * (16 * S3_VS);
                const float* wrow = Ws + 16 * (r - 1) * TPS;
#pragma unroll
                for (int I = 0; I < 8; I++) {
                    const int J = (u - I) & 7;
                    const bool rowp = (I == r && J >= r), colp = (J == r && I >= r);
                    if ((rowp || colp) && I < nb && J < nb) {
                        if (r >= 1) s3_b_update(Bacc[I], vp, wrow, 16 * I, 16 * J, li, lk);
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if (rowp) bpr[(4 * lk + e) * 128 + 16 * J + li] = Bacc[I][e];
                            if (colp) bpc[li * S3_CS + 16 * I + 4 * lk + e] = Bacc[I][e];
                        }
                    }
                }
            }
            if (u == 7) S3T(3, 4 * r + 3);
            s3_barrier();
        }
        // the last block's R rows and V_top go out AFTER the triangular inverse: a wave with global stores in flight stalls there (the
        // compiler waits for them before it reuses their registers: 4 - 6 k cycles inside the inverse's first stage, in-kernel stamps)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) rc_tail[j][s2] = rc[j][s2];
    }
    __builtin_amdgcn_s_setprio(0);
    KT();
    __syncthreads();                                       // the panels are dead: their LDS becomes Ts
    for (int e = tid; e < TP * TPS / 4; e += S3_THREADS) ((float4*)Ts)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // C = (D^-1 + W)^-1, columns of flagged reflectors zeroed (tri_inverse_128 reads only the strictly upper part of Ws)
    KT();
    tri_inverse_128(Ws, tdiag, Ts, (w + 31) / 32, tid);
    KT();
    if (wave >= 3) { store_r(16 * (nb - 1) + 2 * (wave - 3), rc_tail); store_v(nb - 1, wave - 3); }
    // ------------------------------------------------ outputs.  R went out row by row (Householder wave); V_top sits in the
    // lower triangle of Ws (Ws[i][k] = v_top^(k)[i], i > k), the diagonal entries in vdl.
    if ((w & 7) == 0 && (a.c0 & 7) == 0) {                 // usual case: the update waves sent V_top out block by block
    } else {
        for (int e = tid; e < GW * GW; e += S3_THREADS) {
            const int i = e >> 7, k = e & 127;
            if (i < w && k < w && i > k) {
                const float v = Ws[i * TPS + k];
                a.A[(long)(a.c0 + i) * a.lda + a.c0 + k] = v;
                a.Vh[(long)(a.c0 + i) * a.ldvh + a.c0 + k] = (half_t)v;
                a.Vt[(long)(a.c0 + k) * a.ldvt + a.c0 + i] = (half_t)v;
            }
        }
    }
    if (tid < w) {
        const int k = a.c0 + tid;
        const float vd = vdl[tid];
        a.vdiag[k] = vd;
        if (!((w & 7) == 0 && (a.c0 & 7) == 0)) {
            a.Vh[(long)k * a.ldvh + k] = (half_t)vd;
            a.Vt[(long)k * a.ldvt + k] = (half_t)vd;
        }
    }
    if (tid == 0 && lflag) {
        atomicOr(flag, 1);
        if (a.hostflag) __hip_atomic_store(a.hostflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the enqueuing host thread polls this word
    }

    for (int e4 = tid; e4 < GW * GW / 4; e4 += S3_THREADS) {      // window coordinates (leaf index + off), zero elsewhere; four columns per store
        const int i = (e4 >> 5) - off, k0 = (e4 & 31) * 4 - off;
        float c4[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = k0 + q;
            const bool in = i >= 0 && i <= k && k < w;
            const float t = Ts[max(i, 0) * TPS + min(max(k, 0), TP - 1)];       // unconditional LDS read (clamped), selected afterwards
            c4[q] = (in && cmask[min(max(k, 0), GW - 1)]) ? t : 0.f;
        }
        *(float4*)&Cv[4 * e4] = make_float4(c4[0], c4[1], c4[2], c4[3]);
    }
    // Round 5: the launch behind this one (leaf_a) reads columns that the T stream's deferred update of the previous leaf wrote.  As an event
    // wait between the two launches that dependency cost the chain ~6 us per leaf although the update had long finished; here ONE lane of
    // this one-workgroup kernel polls the word the T stream publishes behind that update (launch_publish_word) -- normally set tens of
    // microseconds ago.  It cannot starve the publisher (one workgroup), gives up after ws.ticks like wait_flag_kernel, and the next launch
    // starts with the usual start-of-kernel acquire.
    if (ws.flag && tid == 0 && __hip_atomic_load(ws.flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(ws.flag + ws.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ws.value) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > ws.ticks) {
                __hip_atomic_store(ws.timeout_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(ws.flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    if (ws.stamp && tid == 0) ws.stamp[1] = __builtin_amdgcn_s_memrealtime();
    KT(); KT_DUMP(7, "gh_solve3 zero|loads|barrier0|loop|out|inverse|cstore");
#ifdef MPQR_KTRACE
    __syncthreads();
    if (tid == 0 && atomicSub(&g_s3_trace_left, 1) > 0) {
        const long t0 = g_s3_trace[0][0];
        printf("s3 CH  (start,end per round):");
        for (int q = 0; q < 2 * (nb + 1); q++) printf(" %ld", g_s3_trace[0][q] - t0);
        printf("\ns3 HHr (start,end per round):");
        for (int q = 0; q < 2 * (nb + 1); q++) printf(" %ld", g_s3_trace[1][q] - t0);
        printf("\ns3 HHc (start,end per round):");
        for (int q = 0; q < 2 * (nb + 1); q++) printf(" %ld", g_s3_trace[2][q] - t0);
        printf("\ns3 CH block only (start,end per round):");
        for (int q = 20; q < 20 + 2 * nb; q++) printf(" %ld", g_s3_trace[4][q] - t0);
        printf("\ns3 HHr block only (start,end per round):");
        for (int q = 20; q < 20 + 2 * nb; q++) printf(" %ld", g_s3_trace[5][q] - t0);
        printf("\ns3 HHc block only (start,end per round):");
        for (int q = 20; q < 20 + 2 * nb; q++) printf(" %ld", g_s3_trace[6][q] - t0);
        printf("\ns3 U7  (s1 start,s1 end,s2 start,s2 end per round):");
        for (int q = 0; q < 4 * (nb + 1); q++) printf(" %ld", g_s3_trace[3][q] - t0);
        printf("\n");
    }
#endif
}

void launch_gh_solve3(const LeafArgs& a, const double* G, float* Cv, int* flag, hipStream_t s, const SolveWait* ws) {
    MPQR_ONCE_PER_DEVICE(MPQR_IGNORE(hipFuncSetAttribute((const void*)gh_solve3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS_BYTES)));
    SolveWait w{}; if (ws) w = *ws;
    hipLaunchKernelGGL(gh_solve3_kernel, dim3(1), dim3(S3_THREADS), S3_LDS_BYTES, s, a, G, Cv, flag, w);
}

}  // namespace mpqr
