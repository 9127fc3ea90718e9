/*
 * mpqr.h -- C ABI of the MI355X-native mixed-precision block QR (libmpqr.so).
 *
 * Drop-in boundary for ONE path of jaidonlybbert/MixedPrecisionBlockQR: the block
 * Householder QR  A -> (Q, R)  of Cuda/qr.cu (panel + WY + trailing update + Q
 * accumulation) and its host block-loop driver.  The reference has no FFI layer;
 * its "operator API" is a set of free C++ functions (Cuda/qr.cuh:68-137).  Every
 * entry point below names the reference interface it replaces (file:line relative
 * to the upstream repo).  include/mpqr_reference_api.hpp re-exports them under the
 * reference's exact C++ names and signatures.
 *
 * Plain pointers and sizes only; no C++ or torch types.  All functions return an
 * int status (MPQR_OK == 0) and never call exit() (the reference exits on CUDA
 * errors, Cuda/helper_cuda.h:582-595).  There is NO CPU fallback: without a HIP
 * device every compute entry point returns MPQR_ERR_NO_DEVICE.
 *
 * Storage conventions are the reference's (Cuda/qr.cu:198-293, :1866-1875):
 *   A : float, row-major, (m+1) x n.  In: matrix in rows 0..m-1, row m zero.
 *       Out: R in the upper triangle (i <= j); below it the unit-2-norm Householder
 *       vectors shifted one row down (v_k[j] at row k+1+j), H_k = I - 2 v_k v_k^T,
 *       sign rule sgn(0)=+1, R[k][k] = -sgn(u0)||u||, last column of a square matrix
 *       IS reflected, exactly-zero columns are skipped.
 *   Q : float, row-major, m x m.  In: ignored (reference requires identity, :1873).
 *       Out: the full orthogonal factor, A_in = Q R.
 */
#ifndef MPQR_H
#define MPQR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MPQR_VERSION_MAJOR 0
#define MPQR_VERSION_MINOR 1

enum {
    MPQR_OK = 0,
    MPQR_ERR_INVALID = 1,     /* bad argument / shape */
    MPQR_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime unusable */
    MPQR_ERR_HIP = 3,         /* a HIP call failed (see mpqr_last_error) */
    MPQR_ERR_ALLOC = 4,
    MPQR_ERR_IO = 5,
    MPQR_ERR_STATE = 6        /* call order violated (e.g. factor before plan) */
};

/* operand precision of the trailing-update / Q GEMMs */
enum {
    MPQR_PREC_FP16 = 0,       /* fp16 operands, fp32 accumulate on MFMA (default; Cuda/qr.cu:1049 twin) */
    MPQR_PREC_FP32 = 1,       /* fp32 operands, exact-f32 MFMA (Cuda/qr.cu:958 dev_block_qr_wy twin) */
    MPQR_PREC_FP8 = 2         /* as FP16, but the far trailing updates (K = outer block) take fp8 e4m3 operands on the
                                 block-scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate (BASELINE config 5;
                                 the dtype-templated WMMA GEMM of Cuda/mmult.cuh:252-300).  Panel fp32, in-block updates
                                 and Q formation fp16.  4 significant bits per operand: see DESIGN.md for the measured error */
};

typedef struct mpqr_opts {
    int precision;      /* MPQR_PREC_*                                                    */
    int outer_block;    /* reflectors aggregated per far trailing update (0 = auto)       */
    int form_q;         /* 1: form the full m x m Q (reference contract); 0: R + V only   */
    int lookahead;      /* 1: factor the next block on a second stream (0 = off)          */
    int reserved[12];
} mpqr_opts;

typedef struct mpqr_metrics {
    double backward_error;      /* ||A - Q R||_F / ||A||_F            Cuda/qr.cu:115-135 */
    double q_error_max_signed;  /* max_ij (Q^T Q - I)_ij (signed)     Cuda/qr.cu:137-171 */
    double lower_trapezoid;     /* ||strict lower(R)||_F              Cuda/qr.cu:173-196 */
    double q_error_fro;         /* ||Q^T Q - I||_F  (north-star metric)                  */
    double a_norm;              /* ||A||_F                                               */
} mpqr_metrics;

typedef struct mpqr_timings {
    float ms_total;       /* factor (+ form_q) on the device, HIP events                            */
    float ms_factor;      /* copy-in + max|a| pass (and its host round trip), panels + trailing updates; on a matrix that needed
                             the robust fallback: every pass, the stopped ones included (n_passes, restart_block)             */
    float ms_form_q;      /* backward accumulation of Q behind the last pass (a speculative Q formation behind a pass that was
                             then repaired is inside ms_factor's interval)                                                   */
    float ms_trailing;    /* sum over the far trailing updates (3 GEMM launches each)               */
    float ms_panel;       /* the panel chain timed on its own stream: leaves, in-block updates, T merges
                             (with look-ahead the far updates overlap it on the second stream)       */
    float ms_far_tn;      /* sum over the far  X = A2^T V       launches (K = rows)                 */
    float ms_far_nn;      /* sum over the far  A2 -= V Y^T      launches (K = outer_block)          */
    int   n_far_launches; /* number of far updates (each = one tn + one small + one nn launch)      */
    double flops_far_tn;  /* flops executed by the tn launches (2 M N K each)                       */
    double flops_far_nn;  /* flops executed by the nn launches                                      */
    float ms_chain_wait;  /* ms_factor - ms_panel: the chain stream waiting for far updates of its columns      */
    int   n_passes;       /* block-loop passes of the last mpqr_factor (> 1: flagged leaves were redone)          */
    int   n_robust_leaves;/* tall leaves factored on the column-by-column (robust) kernels in the last pass       */
    float ms_q_tn;        /* Q formation: sum over its  X = Q2^T V   launches                                      */
    float ms_q_nn;        /* Q formation: sum over its  Q2 -= V Y^T  launches (same kernel as the far update)        */
    int   n_q_launches;   /* number of Q-formation applies (blocks or pairs of blocks)                               */
    float tflop_q;        /* flops (in units of 1e12) executed by the tn launches of Q formation (= by the nn ones)  */
    float ms_host_enqueue;/* host time to enqueue the factorisation's launches (before its one synchronisation)    */
    double gbytes_far_nn; /* algorithmic HBM bytes (1e9) of the far A2 -= V Y^T launches: fp32 C read + write, fp16 operands once */
    double gbytes_q_nn;   /* the same for Q formation's Q2 -= V Y^T launches (+ the fp16 shadow they write)          */
    int   n_gh_leaves;    /* Gram-Householder leaves (gram / solve / apply launches) of the last mpqr_factor, all passes */
    float us_gh_solve;    /* one gh_solve launch at this plan's leaf width, timed alone (mpqr_bench_leaf_solve; 0 = not measured) */
    int   n_q_ident_rows; /* Q formation: rows of X = Q2^T V copied from V because their columns of Q were still identity columns */
    int   restart_block;  /* the top-level block the LAST pass started from (n_passes > 1: a flagged leaf's block; the blocks left of it were kept) */
    int   n_fused_leaves; /* leaves of the last pass whose chain-stream work between two solves ran as the three fused launches (leaf_a / leaf_m / leaf_b) */
    int   n_tpoll_retries;/* 1: a polling wait of the T stream timed out and the factorisation was repeated with event hand-offs (stderr says so) */
    int   n_deflated_columns; /* columns whose Gram-Householder pivot was clamped at its threshold (rho_k < 1e-8: numerically dependent on their
                             predecessors) in the last factorisation's last pass: handled in line, R_kk off by <= 1e-4 ||a_k||, no restart  */
    float tflop_q_tn;     /* flops (1e12) the X = Q2^T V launches of Q formation EXECUTED: identity columns of Q are copied and the zero rows of the
                             untouched columns are skipped, so this is less than tflop_q (which the Q2 -= V Y^T launches execute in full)     */
} mpqr_timings;

typedef struct mpqr_handle_s* mpqr_handle_t;

/* ---------------- lifecycle ---------------- */
const char* mpqr_version(void);
/* sizeof(mpqr_opts), sizeof(mpqr_metrics), sizeof(mpqr_timings) as compiled into the library (bindings check their layout) */
void        mpqr_abi_sizes(int out[3]);
void        mpqr_default_opts(mpqr_opts* o);
int         mpqr_create(mpqr_handle_t* h, int device);          /* one handle per GPU; owns its streams */
int         mpqr_destroy(mpqr_handle_t h);
const char* mpqr_last_error(mpqr_handle_t h);                    /* h may be NULL: last error of a failed create */

/* ---------------- the reference drivers ---------------- */
/* replaces  void dev_mixed_precision_block_qr(float* A, float* Q, int m, int n, int r)
 *           Cuda/qr.cuh:133, Cuda/qr.cu:1049-1226   (opts->precision = MPQR_PREC_FP16)
 *      and  void dev_block_qr_wy(float* A, float* Q, int m, int n, int r)
 *           Cuda/qr.cuh:131, Cuda/qr.cu:958-1047    (opts->precision = MPQR_PREC_FP32)
 * Host pointers in/out; all device memory is owned by the handle and reused across calls. */
int mpqr_block_qr_f32(mpqr_handle_t h, float* A, float* Q, int m, int n, int r, const mpqr_opts* opts);

/* Same, on a process-wide default handle (device 0), for callers that keep the reference's
 * handle-less call shape.  Status is returned instead of exiting. */
int mpqr_dev_mixed_precision_block_qr(float* A, float* Q, int m, int n, int r);   /* qr.cuh:133 */
int mpqr_dev_block_qr_wy(float* A, float* Q, int m, int n, int r);                /* qr.cuh:131 */

/* ---------------- device-resident form of the same driver ---------------- */
/* plan: allocate / reuse workspace for an m x n problem with panel width r */
int mpqr_plan(mpqr_handle_t h, int m, int n, int r, const mpqr_opts* opts);
/* load the input matrix (m x n, row-major, leading dimension ld) into HBM */
int mpqr_set_matrix_host(mpqr_handle_t h, const float* A, long ld);
int mpqr_set_matrix_device(mpqr_handle_t h, const float* dA, long ld);
/* U[0,1) synthetic input generated on the device; bit-identical to the oracle's
 * generator (replaces h_generate_random_matrix<float>, Cuda/mmult.cuh:38-64, with a fixed seed) */
int mpqr_generate_matrix(mpqr_handle_t h, uint64_t seed);
/* Factor the loaded input: copy it into the working matrix (the input itself is kept in HBM, so the call
 * can be repeated and the metrics can be evaluated), A -> R,V, then (if form_q) Q.  The factorisation is
 * complete when the call returns; Q formation may still be running on the handle's stream. */
int mpqr_factor(mpqr_handle_t h);
int mpqr_sync(mpqr_handle_t h);
int mpqr_get_timings(mpqr_handle_t h, mpqr_timings* t);
/* measurement aid (bench.py: breakdown_ms.ms_gh_solve = n_gh_leaves x this): launches the serial core of a Gram-Householder leaf
 * (the w x w solve, one workgroup) `iters` times on scratch data and returns the mean launch time in microseconds; the value is
 * also reported by later mpqr_get_timings calls.  Replaces nothing in the reference (its panel runs on the host, qr.cu:1080). */
int mpqr_bench_leaf_solve(mpqr_handle_t h, int w, int iters, float* us_per_launch);
/* test aid: C (-)= A B through ONE of the library's MFMA GEMM kernels, so that the edge tiles (M, N not multiples of the tile,
 * K padded to the kernel's k step with zeros) can be driven directly -- the counterpart of the reference's kernel-level sweep
 * test_template_tensorcore_mmult_tiled (Cuda/mmult.cuh:387-435, iterated by Cuda/qr.cu:1944-1959) against h_mmult.
 * A: M x K, B: K x N, C: M x N, row-major fp32 on the host; A and B are rounded to the kernel's operand type.
 * kernel: 1 = 128-tile fp16 kernel, 2 = 256-tile kernel with the fp32 source converted and transposed while staged
 *         (the far X = A2^T V), 6 = 256-tile ping-pong kernel (both operands fp16 by LDS-DMA), 8 = e4m3 kernel.
 * mode:   0: C = A B,  2: C -= A B (read-modify-write epilogue; kernels 1, 6, 8; M % 32 == 0 -- the library's own callers pad the
 *         row count of the matrix being updated, the epilogue treats a 32-row sub-tile as valid or invalid as a whole). */
int mpqr_gemm_test_f32(mpqr_handle_t h, const float* A, const float* B, float* C, int M, int N, int K, int kernel, int mode);
/* measurement aid (bench.py --gemm, tools/bench_gemm.py): ONE of the large-shape GEMM kernels alone on the GPU, on device-resident random
 * operands, `iters` launches between two HIP events on the handle's chain stream -> mean milliseconds per launch.  The kernel-alone
 * rate next to the rate the same kernel reaches beside the panel chain (mpqr_get_timings) separates kernel quality from contention.
 * kernel: 6 = ping-pong kernel, fp16 A[M][K] and Bt[N][K] by LDS-DMA; 2 = fp32 A stored [K][M], converted + transposed while staged
 *         (the far X = A2^T V);  16 = kernel 6's loop on v_mfma_f32_16x16x32_f16 (experiment).
 * mode:   0: C = A B (fp32 store), 1: fp16 store, 2: C -= A B (fp32 read-modify-write), 3: mode 2 + the transposed fp16 shadow store.
 * M, N multiples of 256, K of 64.  Replaces nothing in the reference (its GEMM tests check values only, Cuda/mmult.cuh:387-435). */
int mpqr_bench_gemm(mpqr_handle_t h, int kernel, int mode, int M, int N, int K, int iters, float* ms_per_launch);
/* measurement aid (bench.py --dump-records, tools/pmc_traffic.py): the recorded read-modify-write GEMM launches  C -= V Y^T  of the last
 * factorisation IN LAUNCH ORDER -- the far trailing updates (is_q = 0; one per launch of the far-update stream) and Q formation's applies
 * (is_q = 1) -- with their algorithmic flops (2 M N K), algorithmic HBM bytes (fp32 C read + write, the fp16 shadow where it is written,
 * both fp16 operands once) and dims[3 i .. 3 i + 2] = M (rows), N (columns), K.  *n = number of records (also when cap is smaller).  A PMC
 * pass of the same command is matched against this list launch by launch, so that measured and algorithmic bytes cover the SAME launches. */
int mpqr_get_update_records(mpqr_handle_t h, int cap, double* flops, double* bytes, int* dims, int* is_q, int* n);
/* measurement aid (bench.py: roofline.mfma_measured): a bare MFMA loop on random fp16 operands held in registers, one 512-thread
 * workgroup per CU, shape 0 = v_mfma_f32_32x32x16_f16, 1 = v_mfma_f32_16x16x32_f16 -> TFLOP/s of the whole device and the shader clock
 * it holds meanwhile (GHz).  The dense peak of MI355X_MICROARCH.md (2.5 PFLOP/s) is width x 2.4 GHz; under load the chip lowers its
 * clock, differently per shape: this is the attainable denominator next to the nominal one (SURVEY.md 8d asks for the cross-check). */
int mpqr_bench_mfma_peak(mpqr_handle_t h, int shape, float* tflops, float* ghz);
/* results: A_out is (m+1) x n in the reference's shifted-reflector layout, Q is m x m */
int mpqr_get_factor_host(mpqr_handle_t h, float* A_out);
int mpqr_get_q_host(mpqr_handle_t h, float* Q);
int mpqr_get_r_host(mpqr_handle_t h, float* R);      /* m x n, strict lower part zero: h_strip_R_from_A qr.cu:85-100 */
/* three metrics of the reference's testers, computed on the device against the retained input */
int mpqr_metrics_device(mpqr_handle_t h, mpqr_metrics* out);

/* ---------------- stage-level entry points (parity tests) ---------------- */
/* replaces h_householder_qr(float* A,int m,int n,int global_offset,int panel_width)  Cuda/qr.cu:198-293
 * A: host (m+1) x n, factored in place on the GPU (columns [go, go+pw) only) */
int mpqr_householder_qr_f32(mpqr_handle_t h, float* A, int m, int n, int global_offset, int panel_width,
                            int precision /* MPQR_PREC_FP32 = the reference's arithmetic; FP16: in-panel updates on the fp16 MFMA path */);
/* replaces h_wy_transform / dev_wy_transform  Cuda/qr.cu:337-426, :535-600.
 * Compact form: T (pw x pw, row-major, upper) of Q_panel = I - V T V^T built from the reflectors
 * stored in A.  If Qpanel != NULL also returns the dense (m-go)^2 matrix the reference materialises. */
int mpqr_wy_transform_f32(mpqr_handle_t h, const float* A, int m, int n, int global_offset, int panel_width,
                          float* T, float* Qpanel);
/* replaces h_q_backward_accumulation(float* h_A, float** h_Q, int m, int n)  Cuda/qr.cu:296-335
 * (Q is caller-allocated, m x m) */
int mpqr_q_backward_accumulation_f32(mpqr_handle_t h, const float* A, float* Q, int m, int n, int precision);
/* replaces the trailing update  A[l:,tau:] <- Q_panel^T A[l:,tau:]
 *   shared_mem_mmult_in_place_transpose_a + dev_cpy_strided_array  Cuda/mmult.cu:236-288, qr.cu:1098-1106
 * with the compact-WY MFMA apply, reflectors of columns [go, go+pw) taken from A itself */
int mpqr_apply_panel_to_trailing_f32(mpqr_handle_t h, float* A, int m, int n, int global_offset, int panel_width,
                                     int precision);
/* replaces h_backward_error / h_q_error / h_lower_trapezoid_error  Cuda/qr.cu:115-196 (host buffers) */
int mpqr_metrics_f32(mpqr_handle_t h, const float* A, const float* R, const float* Q, int m, int n, mpqr_metrics* out);
/* replaces h_q_error(float* Q, int m, ...)  Cuda/qr.cu:137-171 alone: fills q_error_max_signed and q_error_fro */
int mpqr_q_error_f32(mpqr_handle_t h, const float* Q, int m, mpqr_metrics* out);
/* pass criterion of the reference's testers: err <= 2^-precision_bits * m  (qr.cu:120,127) */
int mpqr_error_passes(double err, int m, int precision_bits);

/* ---------------- C++/main.cpp path ---------------- */
/* replaces void qr_factorization(MatrixXd& A, MatrixXd& Q)  C++/main.cpp:16-43.
 * column-major doubles (Eigen MatrixXd storage), m x n with m >= n, A -> R in place,
 * Q (m x m) out.  Runs on the GPU in fp64. */
int mpqr_qr_factorization_f64(mpqr_handle_t h, double* A, double* Q, int m, int n);

/* ---------------- least squares on top of the factorisation ---------------- */
/* B <- Q^T B for host B (m x nrhs, row-major, ld = nrhs), applied IMPLICITLY from the stored reflectors (V, T of every
 * column block; Q is never formed): step 2 of Golub & Van Loan Alg. 5.3.2, the algorithm named by the reference's
 * dev_QR_Solver stub (Cuda/QR/Solver/solver.cu:39-87).  Needs a factored handle (mpqr_factor / mpqr_block_qr_f32). */
int mpqr_apply_qt_host(mpqr_handle_t h, float* B, int nrhs);
/* X (n x nrhs, row-major, ld = nrhs) = argmin ||A X - B||_F = R^-1 (Q^T B)[0:n]  for host B (m x nrhs);
 * replaces linear_least_square(A, y)  python/linear_least_sqare.py:5-22 (its pinv(Q) y is Q^T y, the back
 * substitution is its loop :17-21).  Needs a factored handle; a zero R_ii gives x_i = 0. */
int mpqr_solve_ls_host(mpqr_handle_t h, const float* B, int nrhs, float* X);
/* replaces void dev_QR_Solver(float* A, float* b, float* x, int m, int n)  Cuda/QR/Solver/solver.cu:39-87 (a stub in
 * the reference): factor A (m x n row-major host, r-column panels, Q not formed), x = argmin ||A x - b||. */
int mpqr_qr_solver_f32(mpqr_handle_t h, const float* A, const float* b, float* x, int m, int n, int r);
int mpqr_dev_qr_solver(const float* A, const float* b, float* x, int m, int n);    /* default handle, r = 128 */

/* ---------------- host-side helpers of the path (no GPU needed) ---------------- */
/* replaces read_euroc_jacobian(std::string, int*, int*, float**)  Cuda/qr.cu:696-776.
 * *matrix is malloc'd by the callee (as in the reference); free with mpqr_free_host. */
int  mpqr_read_euroc_jacobian(const char* path, int* rows, int* cols, float** matrix);
int  mpqr_write_euroc_jacobian(const char* path, int rows, int cols, const float* matrix);
void mpqr_free_host(void* p);
/* replaces h_write_results_to_log  Cuda/qr.cu:58-83: appends "rows,cols,runtime,flops,error" CSV */
int  mpqr_write_results_to_log(const char* dir, const char* file_name, int height, int width, float time_ms,
                               float flops_per_second, float backward_error);
/* replaces h_qr_flops_per_second  Cuda/qr.cu:102-113 (fp32 arithmetic kept on purpose) */
float  mpqr_qr_flops_per_second(float time_ms, int m, int n);
/* algorithmic flop counts used for roofline accounting */
double mpqr_flops_geqrf(int m, int n);                 /* 2 m n^2 - 2/3 n^3                          */
double mpqr_flops_form_q(int m, int n);                /* 4 (m^2 n - m n^2 + n^3/3)                  */
double mpqr_flops_trailing(int m, int n, int r);       /* sum_j 4 W r nk + r^2 nk over the block loop */
double mpqr_flops_panel(int m, int n, int r);          /* sum_j 2 W r^2                               */
/* host-side U[0,1) generator, bit-identical to mpqr_generate_matrix */
void   mpqr_generate_matrix_host(float* A, int m, int n, uint64_t seed);

/* ---------------- multi-GPU: 1-D block-cyclic column shards ---------------- */
/* Column blocks of width `block` are dealt round-robin: block j lives on rank j % world. */
int  mpqr_part_owner(int col, int block, int world);
int  mpqr_part_local_cols(int n, int block, int world, int rank);
int  mpqr_part_local_index(int col, int block, int world);          /* column index inside its owner's shard */
int  mpqr_part_global_index(int lcol, int block, int world, int rank);

/* Distributed driver, one process (and one handle) per GPU.  Column superblocks of `outer_block` columns
 * (mpqr_dist_block) are dealt round-robin (block s -> rank s % world); every rank holds all m rows of its
 * columns.  Q is sharded by columns in the same pattern.  The library never communicates: per block the owner
 * packs one contiguous device buffer, the CALLER broadcasts it (torch.distributed / RCCL over xGMI in bench.py),
 * and every rank unpacks and updates its own trailing columns:
 *
 *     mpqr_dist_plan; mpqr_dist_generate_matrix | mpqr_dist_set_local_matrix_host
 *     mpqr_dist_local_absmax -> all-reduce(max) -> mpqr_dist_begin(global_absmax)
 *     for s in 0 .. mpqr_dist_num_blocks-1:
 *         owner:  mpqr_dist_factor_block(s); mpqr_dist_pack_block(s, buf)
 *         all:    broadcast(buf, root = mpqr_dist_block_owner(s)); mpqr_dist_unpack_block(s, buf); mpqr_dist_update(s)
 *     mpqr_dist_form_q                      (no communication)
 *     mpqr_sync; mpqr_dist_flags -> all-reduce(max): a flagged (ill-conditioned tall leaf) factorisation is repeated after
 *     mpqr_dist_set_robust(1) on every rank -- no rank synchronises its host inside the block loop
 */
int  mpqr_dist_plan(mpqr_handle_t h, int m, int n, int r, int world, int rank, const mpqr_opts* opts);
int  mpqr_dist_block(mpqr_handle_t h);              /* columns per distributed block                     */
int  mpqr_dist_num_blocks(mpqr_handle_t h);
int  mpqr_dist_block_owner(mpqr_handle_t h, int s);
int  mpqr_dist_local_cols(mpqr_handle_t h);         /* columns of A held by this rank                    */
int  mpqr_dist_local_q_cols(mpqr_handle_t h);       /* columns of Q held by this rank                    */
int  mpqr_dist_set_local_matrix_host(mpqr_handle_t h, const float* A_local, long ld);   /* m x local_cols */
int  mpqr_dist_generate_matrix(mpqr_handle_t h, uint64_t seed);   /* this rank's columns of the global matrix */
int  mpqr_dist_local_absmax(mpqr_handle_t h, float* out);
int  mpqr_dist_begin(mpqr_handle_t h, float global_absmax);
int  mpqr_dist_factor_block(mpqr_handle_t h, int s);
int  mpqr_dist_flags(mpqr_handle_t h, int* any);      /* synchronises; 1: a Gram-Householder leaf of this rank flagged itself since mpqr_dist_begin */
int  mpqr_dist_set_robust(mpqr_handle_t h, int on);   /* every tall leaf on the column-by-column kernels from now on (reset by a new input)         */
long mpqr_dist_block_bytes(mpqr_handle_t h, int s);
int  mpqr_dist_pack_block(mpqr_handle_t h, int s, void* device_buf);
int  mpqr_dist_unpack_block(mpqr_handle_t h, int s, const void* device_buf);
/* the same without the host synchronisation: the copies are enqueued on the handle's chain stream (mpqr_dist_chain_stream returns it
 * as a hipStream_t) and the caller orders its broadcasts against them with events -- pack -> event -> communication stream waits;
 * broadcast -> event -> chain stream waits -> unpack -> event -> the next broadcast into the same buffer waits */
int  mpqr_dist_pack_block_async(mpqr_handle_t h, int s, void* device_buf);
int  mpqr_dist_unpack_block_async(mpqr_handle_t h, int s, const void* device_buf);
int  mpqr_dist_chain_stream(mpqr_handle_t h, void** hip_stream);
int  mpqr_dist_update(mpqr_handle_t h, int s);               /* = mpqr_dist_update_part(h, s, 2) */
/* look-ahead form: part 0 = the columns of block s+1 only (its owner factors that block next), chain stream;
 * part 1 = the local columns right of block s+1, far-update stream, runs beside the factorisation of block s+1;
 * part 2 = everything, chain stream.  Schedule: mixedprecisionblockqr_amd/dist.py:factor, apps/mpqr_main.cpp --gpus N. */
int  mpqr_dist_update_part(mpqr_handle_t h, int s, int part);
int  mpqr_dist_form_q(mpqr_handle_t h);
/* results of this rank: (m+1) x local_cols in the reference's shifted layout; m x local_q_cols; the input */
int  mpqr_dist_get_local_factor_host(mpqr_handle_t h, float* A_local);
int  mpqr_dist_get_local_q_host(mpqr_handle_t h, float* Q_local);
int  mpqr_dist_get_local_input_host(mpqr_handle_t h, float* A_local);
#ifdef __cplusplus
}
#endif
#endif
