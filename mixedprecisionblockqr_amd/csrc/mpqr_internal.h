// mpqr_internal.h -- internal declarations shared by the HIP translation units of libmpqr.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mpqr.h"

namespace mpqr {

typedef _Float16 half_t;

// Function attributes (hipFuncSetAttribute: the raised dynamic-LDS limit) belong to the CURRENT DEVICE's function object: a
// process-wide `static bool` sets them on the first device only, and races between the rank threads of the multi-GPU host
// (apps/mpqr_main --gpus N: one handle per device in one process).  This runs `stmt` once per (call site, device).
template <typename F>
inline void once_per_device(std::atomic<unsigned long long>& done, std::mutex& mu, F&& set) {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    std::lock_guard<std::mutex> lk(mu);
    if (done.load(std::memory_order_relaxed) & bit) return;
    set();
    done.fetch_or(bit, std::memory_order_release);
}
// A HIP call whose failure does not change what the caller can do (teardown, attribute set-up, best-effort hints): the result is still
// LOOKED AT -- a failure is reported on stderr with its call site -- instead of being cast to void (hygiene).
#define MPQR_IGNORE(call)                                                                                          \
    do {                                                                                                           \
        hipError_t ie_ = (call);                                                                                   \
        if (ie_ != hipSuccess) fprintf(stderr, "mpqr: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(ie_), __FILE__, __LINE__); \
    } while (0)
#define MPQR_ONCE_PER_DEVICE(stmt)                                                   \
    do {                                                                             \
        static std::atomic<unsigned long long> once_done_{0};                        \
        static std::mutex once_mu_;                                                  \
        ::mpqr::once_per_device(once_done_, once_mu_, [&]() { stmt; });              \
    } while (0)

// ------------------------------------------------------------------ GEMM family
// C[M x N] = A[M x K] * B[K x N], B always supplied as Bt[N][K] (k contiguous, fp16).
enum AMode {
    A_H16 = 0,   // fp16 source [M][K], k contiguous
    A_F32T = 1,  // fp32 source [K][M], m contiguous: converted (x in_scale) and transposed while staged
    A_F32 = 2,   // fp32 source [M][K], k contiguous, summed over nslab slabs while staged
    A_F32S = 3   // as A_F32, staged as fp16 hi + lo parts (two MFMAs per fragment: the operand keeps ~22 bits).  Y = X T':
                 // rounding X to fp16 before the product and Y after it rounds a dominant reflector's contribution twice
};
enum EMode {
    E_STORE_F32 = 0,  // C(slab z)[m][n] = alpha*acc          (fp32, split-K slabs)
    E_STORE_H16 = 1,  // C[m][n] = fp16(alpha*acc)
    E_SUB_F32 = 2     // C[m][n] -= alpha*acc  for n >= col_lo (fp32 read-modify-write)
};

struct GemmArgs {
    const void* A;       long lda;
    const half_t* Bt;    long ldb;
    void* C;             long ldc;
    int M, N, K;         // M multiple of 128 not required: rows >= M are zero-filled / not stored. K % 64 == 0.
    int col_lo;          // E_SUB_F32: columns < col_lo are left untouched
    float in_scale;      // A_F32T / A_F32 conversion scale
    float alpha;
    int nslab_in;        long slab_in_stride;    // A_F32: sum of nslab_in source slabs
    int nsplit;          long slab_out_stride;   // E_STORE_F32: K split over gridDim.z slabs
    // E_SUB_F32: optional fp16 shadow of the updated values, TRANSPOSED: Ct[n * ldct + m] = fp16(ct_scale * C[m][n]).  The
    // next X = C^T V then reads fp16 operands that are already k-contiguous (all LDS-DMA) instead of converting and
    // transposing fp32 in registers.  nullptr: none.
    half_t* Ct;          long ldct;   float ct_scale;
    // Bt is triangular (the compact-WY T as the second operand): 1: Bt[n][k] = 0 for k < n, 2: Bt[n][k] = 0 for k > n.
    // The 256-wide kernel skips the K tiles that are entirely zero for its output columns; other kernels ignore it.
    int tri;
    // E_STORE_H16: column n of the result is multiplied by cscale[n * cscale_ld] (fp32) before it is rounded.  The fp16
    // copies of a compact-WY T carry T[n][k] / T[n][n] (unit diagonal, exactly representable): the diagonal tau_n stays in
    // fp32 (the diagonal of the fp32 T) and is applied here.  Rounding tau_n ~ 2 to fp16 would put a coherent relative error
    // of up to 2^-11 on a whole reflector's contribution; with a dominant reflector (non-centred data) that alone moved the
    // backward error of random instances between 6e-4 and 1.2e-3.
    const float* cscale; long cscale_ld;
    // split fp16 operands (hi + lo = ~22 bits) between the two GEMMs of an update:
    half_t* C2;          // E_STORE_H16 (256-wide kernels): also store the remainder  fp16(v - fp16(v))  here (same layout as C)
    const half_t* A2;    // ping-pong kernel: a second A operand (the lo parts) accumulated over the same Bt: C = (A + A2) Bt^T
    int eye_minus;       // E_STORE_F32 (256-wide kernel): store (m == n ? 1 : 0) - alpha*acc  (Q = I - W V^T in one product)
    int nt_c;            // ping-pong kernel: C (and Ct) are streamed with the non-temporal cache policy (read-modify-write ring, fp32 stores)
};
// false: no 128-tile kernel for this (staging, epilogue) pair, or hi + lo operands requested (256-wide kernels only)
bool launch_gemm_f16(AMode am, EMode em, const GemmArgs& g, hipStream_t s);
// 256 x 256 x 64 variant for large shapes (no split-K; operands readable up to the next multiple of 256 rows)
bool launch_gemm2_f16(AMode am, EMode em, const GemmArgs& g, hipStream_t s, int config = 0);

// ------------------------------------------------------------------ fp8 (e4m3) operand path (kernels_fp8.hip)
// C (-)= alpha A[M][K] Bt[N][K]^T with 1-byte e4m3 operands (g.A / g.Bt point at bytes, lda / ldb in bytes), K % 128 == 0;
// E_SUB_F32 or E_STORE_F32 (split-K over g.nsplit slabs).  false: shape not supported.
bool launch_gemm_fp8(EMode em, const GemmArgs& g, hipStream_t s);
void launch_quant_h16_fp8(const half_t* src, long lds_, uint8_t* dst, long ldd, int rows, int cols, float scale, hipStream_t s);
// dst[c][r] = fp8(scale * src[r][c]); rows % 64 == 0
void launch_quant_transpose_f32_fp8(const float* src, long lds_, uint8_t* dst, long ldd, int rows, int cols, float scale, hipStream_t s);

// ------------------------------------------------------------------ fp32 helper GEMM (T merges, metrics)
// C[M x N] (ldc) = alpha * opA(A) * opB(B) + beta*C, plain fp32 FMA, any sizes.
// slabs: A is summed over nslab_a slabs (stride slab_a) while loading.
struct SgemmArgs {
    const float* A; long lda; int transA;
    const float* B; long ldb; int transB;
    float* C; long ldc;
    int M, N, K;
    float alpha, beta;
    int nslab_a; long slab_a;
    int nslab_b; long slab_b;   // op(B) arrives as nslab_b partial slabs (0 / 1: plain), summed while staged like A's
    int upperA, upperB;   // operand is upper triangular (op(A)[i][k]=0 for k<i / op(B)[k][j]=0 for k>j): zero K tiles skipped
    int ksplit; long slab_c;   // ksplit > 1: K is cut into ksplit ranges (blockIdx.z), range z writes alpha * its partial product to C + z * slab_c (beta ignored)
};
void launch_sgemm(const SgemmArgs& g, hipStream_t s);

// ------------------------------------------------------------------ panel (leaf) kernels
struct LeafArgs {
    float* A; long lda;         // internal fp32 matrix (R above, unshifted V below the diagonal)
    int mrows;                  // rows considered (padded rows are zero)
    int cb;                     // 32-aligned window start; leaf columns [c0, c1) lie in [cb, cb+32)
    int c0, c1;
    half_t* Vh; long ldvh;      // [row][reflector]
    half_t* Vt; long ldvt;      // [reflector][row]
    float* vdiag;               // diagonal element of each reflector
    float* P;                   // partial dot products, 2 x maxwg x 32
    int maxwg;
    int* hostflag;              // Gram-Householder leaves: word in mapped host memory that a flagged leaf also raises (or nullptr)
    int* deflword;              // ... and the word that counts the columns gh_solve deflated (pivot clamp; or nullptr)
};
void launch_leaf_factor(const LeafArgs& a, hipStream_t s);   // robust path: one workgroup (<=2048 rows) or one launch per column
// the last leaf of a (nearly) square matrix: at most 128 rows from row c0 down, up to 128 columns inside the 128-aligned window a.cb; one
// workgroup; also writes S = V^T V of the fp16-rounded reflectors (128 x 128 fp32, window coordinates) for launch_t_leaf
void launch_leaf_tail(const LeafArgs& a, float* S, hipStream_t s);
// tall leaves (up to 128 columns inside a 128-aligned window, a.cb): Gram-Householder, 4 launches; raises *flag
// when the leaf is too ill-conditioned for it
// If Sp != nullptr the apply kernel also emits the Gram matrix of the fp16 reflectors (window coordinates,
// 128 x 128 fp32, upper triangle only) into S, via per-workgroup partials Sp ((rows/64 + 2) x 16384).
// S == nullptr: the partials Sp are left for launch_gh_reduce_f32 (gh_num_partials of them) on another stream.
void launch_leaf_gram_householder(const LeafArgs& a, double* Gp /* nwg x 16384 */, double* G /* 16384 */,
                                  float* Cv /* 16384 */, int* flag, float* Sp, float* S, hipStream_t s);
int gh_num_partials(const LeafArgs& a);
// the same leaf in separately launchable steps
void launch_gh_gram(const LeafArgs& a, double* Gp, double* G, hipStream_t s);
// gh_solve can end by polling a progress word of ANOTHER stream (flag[word] >= value; flag[1] = "a wait has timed out", see wait_flag_kernel)
struct SolveWait { int* flag; int word; int value; int* timeout_word; unsigned long long ticks;
                   unsigned long long* stamp; };     // measurement aid (MPQR_DBG_STAMPS): stamp[0] / stamp[1] = s_memrealtime (100 MHz) at the kernel's start / end
void launch_gh_solve(const LeafArgs& a, const double* G, float* Cv, int* flag, hipStream_t s, const SolveWait* ws = nullptr);
// one thread stores flag[word] = value (a stream publishes "everything I have run so far is complete")
void launch_publish_word(int* flag, int word, int value, hipStream_t s);
void launch_gh_apply(const LeafArgs& a, const float* Cv, float* Sp, hipStream_t s);
// blocked form of gh_solve (kernels_solve.hip): same inputs and outputs
void launch_gh_solve3(const LeafArgs& a, const double* G, float* Cv, int* flag, hipStream_t s, const SolveWait* ws = nullptr);
// Y[M1 x 128] (fp16, ld ldy) = fp16(sum of nslab X slabs, M1 x 128 fp32) * T', Bt[n][k] = T'[k][n] (fp16, ld ldb), tri as GemmArgs::tri
void launch_leaf_xt(const float* X, int nslab, long slab_stride, int M1, const half_t* Bt, long ldb, int tri, half_t* Y, long ldy,
                    const float* cscale, long cscale_ld, hipStream_t s, int* pub_flag = nullptr, int pub_value = 0);
// cross-stream dependency without an event: stream s goes on once *flag >= value (published by leaf_xt_kernel: pub_flag / pub_value)
// flag[0]: the published word, flag[1]: set once a wait has timed out (ticks of 100 MHz)
// word: which of the chain's two progress words to wait for (flag[0]: past the leaf's T, flag[2]: past the leaf's reflectors)
void launch_wait_flag(int* flag, int value, int* timeout_word, unsigned long long ticks, hipStream_t s, int word = 0);
void launch_gh_reduce_f32(const float* Sp, int nslab, float* S, hipStream_t s);

// T of a leaf (<= 128 reflectors) from its Gram slabs S (row stride lds_, aligned range starting at a0)
// ld: leading dimension of T / Th / Tth (0: ldt, the node's own contiguous T)
// One launch for the middle of a Gram-Householder leaf (kernels_panel.hip, leaf_mid_kernel): the split-K GEMM g1 (A_F32T, E_STORE_F32,
// N <= 128) beside the sum of the nslab partial Gram matrices Sp -> S and the leaf's T from it (arguments as launch_t_leaf, S read at
// (sh, sh) of its 128 x 128 window).  counter: one device int, zero between launches.
constexpr int LEAF_MID_MAX_GROUPS = 4;      // S must hold this many 128 x 128 windows
void launch_leaf_mid(const GemmArgs& g1, const float* Sp, int nslab, float* S, int sh, int* counter, int a0, int c0, int c1,
                     float* T, half_t* Th, half_t* Tth, int ldt, int ld, hipStream_t s);
// Fused leaf (round 5, kernels_panel.hip): the chain stream's work between two solves in three launches that touch only the NEXT leaf's 128
// columns [nb, nb + 128).  leaf_a = gh_apply + partial X = (in_scale P)^T V over the same row blocks (Xp: one 128 x 128 fp32 partial per
// workgroup, gh_num_partials(a) of them, like Sp); leaf_m = sums of the partials, T of the leaf and Y = fp16(X T') (Y: 128 x 128 fp16,
// [column of the next panel][reflector]); leaf_b = P -= alpha V Y^T and, if do_gram, the next leaf's partial Gram matrices (fl_gram_partials(a) of
// them in Gp, for launch_gh_gram_reduce + launch_gh_solve); it also publishes pub_value in *pub_flag (launch_wait_flag's word).
void launch_leaf_a(const LeafArgs& a, const float* Cv, float* Sp, float* Xp, int nb, float in_scale, hipStream_t s);
void launch_leaf_m(const float* Sp, const float* Xp, int nslab, float* S, float* Xs, int* counter, int sh, int a0, int c0, int c1,
                   float* T, half_t* Th, half_t* Tth, int ldt, int ld, half_t* Y, hipStream_t s, int* pub_flag = nullptr, int pub_value = 0);
void launch_leaf_b(const LeafArgs& a, int nb, const half_t* Y, float alpha, double* Gp, bool do_gram, int* pub_flag, int pub_value, hipStream_t s);
int fl_gram_partials(const LeafArgs& a);
void launch_gh_gram_reduce(const double* Gp, int nwg, double* G, hipStream_t s);
void launch_t_leaf(const float* S, int nslab, long slab_stride, int lds_, int a0, int c0, int c1,
                   float* T, half_t* Th, half_t* Tth, int ldt, hipStream_t s, int ld = 0);
// fp16 copies (plain and transposed) of column block [c, c+w) x rows [0, rows) of a T with leading dimension ld
void launch_t_colblock_h16(float* T, half_t* Th, half_t* Tth, int ld, int rows, int c, int w, hipStream_t s,
                           const float* P = nullptr, int nz = 0, long slab = 0, int ldp = 0);   // P: split-K partials of the column block, summed into T first
// assemble a parent T from its children and T_LR
// one diagonal block of the back substitution R X = Y (in place in Y), kb <= 128
void launch_trsm_diag(const float* R, long ldr, int k0, int kb, float* Y, long ldy, int nrhs, hipStream_t s);
// T_LR = -T_L (S T_R), children of <= 128 reflectors: one workgroup, exact-f32 MFMA out of LDS
void launch_t_merge(const float* S, int ldl, int ldr, const float* TL, const float* TR, float* TLR, hipStream_t s);
void launch_t_assemble(float* T, half_t* Th, half_t* Tth, int ldt, int A0,
                       const float* TL, int ldl, int aL0, int c0, int cm,
                       const float* TR, int ldr, int aR0, int c1,
                       const float* TLR, int ldlr, hipStream_t s, int nz = 1, long zslab = 0);   // TLR as nz partial slabs

// ------------------------------------------------------------------ misc kernels
void launch_generate(float* A, long lda, int m, int n, uint64_t seed, int nglob, int block, int world, int rank,
                     hipStream_t s);
void launch_set_identity(float* Q, long ldq, int rows, int cols, hipStream_t s);
void launch_set_identity_h16(half_t* Q, long ldq, int n, hipStream_t s);   // diagonal of a zeroed fp16 matrix
void launch_identity_cyclic_h16(half_t* Qt, long ldqt, int m, int qloc, int block, int world, int rank, hipStream_t s);
void launch_pack_factor(const float* A, long lda, const float* vdiag, float* out, int m, int n, hipStream_t s);
void launch_pack_factor_rows(const float* A, long lda, const float* vdiag, float* out, int m, int n, int r0, int r1, hipStream_t s);
void launch_unpack_factor(const float* in, int m, int n, int c0, int c1, float* A, long lda, float* vdiag,
                          half_t* Vh, long ldvh, half_t* Vt, long ldvt, hipStream_t s);
void launch_strip_r(const float* A, long lda, float* R, int m, int n, hipStream_t s);
void launch_slab_reduce(const float* src, int nslab, long stride, long n_elems, float* dst, hipStream_t s);
void launch_extract_vf(const float* A, long lda, const float* vdiag, float* Vf, long ldvf, int rows, int c0, int c1, hipStream_t s);
void launch_identity_cyclic(float* Q, long ldq, int m, int qloc, int block, int world, int rank, hipStream_t s);
void launch_pack_factor_cyclic(const float* A, long lda, const float* vdiag, float* out, int m, int nloc, int block,
                               int world, int rank, hipStream_t s);
void launch_transpose_h16(const half_t* src, long lds_, half_t* dst, long ldd, int rows, int cols, hipStream_t s);
void launch_absmax(const float* A, long lda, int m, int n, float* out /*1*/, hipStream_t s);
void launch_copy_absmax(const float* src, float* dst, long ld, int rows, int m, int n, float* out /*1*/, hipStream_t s);
// metrics reductions: out[0] += sum (A - B)^2, out[1] += sum A^2
void launch_diff_norms(const float* A, long lda, const float* B, long ldb, int m, int n, double* out, hipStream_t s);
// out[0] += sum (G - I)^2, out[1] = max signed (G - I)
void launch_gram_minus_identity(const float* G, long ldg, int m, double* out, hipStream_t s);
void launch_lower_norm(const float* R, long ldr, int m, int n, double* out, hipStream_t s);

// measurement aid: bare MFMA loop on random operands (shape 0: 32x32x16, 1: 16x16x32), nwg workgroups of 8 waves; clk[2 b], clk[2 b + 1] =
// shader-clock and 100 MHz ticks workgroup b spent in the loop
void launch_mfma_peak(int shape, const half_t* src /* 2^20 halves */, float* out /* nwg x 512 */, long* clk /* 2 nwg */, int nwg, int iters, hipStream_t s);

// fp64 column-major Householder QR (C++/main.cpp path)
void launch_qr_f64(double* A, double* Q, int m, int n, double* work, hipStream_t s);

}  // namespace mpqr
