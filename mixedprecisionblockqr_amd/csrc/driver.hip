// driver.hip -- host block-loop driver + C ABI of libmpqr.so.
//
// Replaces the reference's host drivers dev_mixed_precision_block_qr (Cuda/qr.cu:1049-1226),
// dev_block_qr_wy (:958-1047) and the testers' glue (:1856-1908): the whole factorisation runs on
// the device-resident matrix -- no per-panel PCIe round trips (qr.cu:1082,1215), no per-panel
// cudaMalloc/cudaFree (:1086,1115-1133,1209-1213), no device-wide sync after each launch.
//
// Algorithm (results identical to the reference's up to rounding: same Householder vectors, same
// sign rule, same R, same Q):
//   * columns are organised in a binary tree of ranges; leaves are <=32-column pieces of the
//     caller's r-wide panels, inner nodes merge panels up to `outer_block` reflectors;
//   * leaf: fp32 Householder (kernels_panel.hip), T from the Gram matrix of the fp16 reflectors;
//   * inner node [c0,c1) = L + R: factor L, apply (I - V_L T_L^T V_L^T) to R's columns with the MFMA
//     GEMMs, factor R, T_LR = -T_L (V_L^T V_R) T_R;
//   * top-level node: one far trailing update  A2 -= V (T^T (V^T A2))  with K = outer_block;
//   * Q = H_1 ... H_n by backward accumulation over the top-level nodes (G&VL 5.1.5 blocked), which
//     is what h_q_backward_accumulation (qr.cu:296-335) computes and equals the reference's forward
//     product Q <- Q Q_panel (qr.cu:1109-1207) up to rounding.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <mutex>

#include <rocprofiler-sdk-roctx/roctx.h>

#include "mpqr_internal.h"
#include <atomic>

using namespace mpqr;

namespace {

struct Node {
    int c0, c1;        // reflector (column) range
    int a0, a1;        // 64-aligned range containing it
    int ldt;           // a1 - a0
    int left, right;   // children or -1
    size_t toff;       // offset into the T arenas (elements)
    int id;            // index in the node list (per-node events / flags)
    int tld;           // leading dimension of this node's T (ldt: own contiguous T; larger: diagonal block of a block-level T)
};

// roctx range over the host-side enqueue of one phase (the reference brackets h_householder_qr, h_wy_transform,
// dev_wy_transform and the WMMA GEMM launcher with NVTX ranges: Cuda/qr.cu:207,292,339,425,536,599, mmult.cuh:324,383)
struct Range {
    explicit Range(const char* name) { roctxRangePushA(name); }
    ~Range() { roctxRangePop(); }
};

inline int rdown(int x, int a) { return (x / a) * a; }
inline int rup(int x, int a) { return ((x + a - 1) / a) * a; }

}  // namespace

struct mpqr_handle_s {
    int device = 0;
    hipStream_t s0 = nullptr;   // panel chain (high priority)
    hipStream_t s1 = nullptr;   // far trailing updates / Q formation when opts.lookahead (low priority)
    hipStream_t sT = nullptr;   // compact-WY T construction (leaf T, merges): runs beside the chain's next V-only GEMM
    hipStream_t sD = nullptr;   // the drop-in call's read-back of R / V, beside Q formation (created on first use)
    std::vector<hipEvent_t> ev_T;     // per node: T of that node is complete (recorded on sT)
    hipEvent_t ev_v = nullptr, ev_join = nullptr;   // chain -> T stream (reflectors written), T stream -> chain (join)
    hipEvent_t wait_after_first_leaf = nullptr;     // look-ahead: the rest of the block's columns become valid with this event
    int ext_c1 = 0;                     // flat schedule under look-ahead: in-block updates run on to this column (the NEXT block's first leaf)
    bool defer_join = false;            // block loop with look-ahead: the chain does not wait for a block's T (its users wait on ev_T)
    std::vector<hipEvent_t> ev_cols2;
    bool tq_on = false;         // this plan builds its T's on sT (fp16 mode)
    std::function<int()> far_hook;          // run_block_loop: the previous block's far update, enqueued behind the next block's first gh_gram
    hipStream_t node_done_stream = nullptr;   // stream on which the last factor_node left the block's reflectors complete
    hipStream_t inblock_stream = nullptr;     // apply_node, lane 0: run on this stream instead of s0
    hipStream_t op1_stream = nullptr;   // set by factor_block_flat: X = C2^T V (+ its slab sum) of the next apply runs there
    hipEvent_t ev_x = nullptr;          // ... and this event orders it before the apply's second GEMM
    hipEvent_t ev_dist_chain = nullptr, ev_dist_far = nullptr;   // distributed look-ahead: chain -> far stream, far -> chain stream
    std::string err;

    bool planned = false;
    int m = 0, n = 0, r = 0, m_pad = 0, n_pad = 0;
    long lda = 0, ldq = 0, ldvh = 0, ldvt = 0;
    mpqr_opts opts;
    int Ko = 0;

    float* dA = nullptr;      // working matrix: R above the diagonal, reflectors below
    float* dA0 = nullptr;     // the input as given (kept: metrics, re-runs, fallback to the robust panel path)
    bool have_input = false;
    float* dQ = nullptr;
    half_t* Vh = nullptr;
    half_t* Vt = nullptr;
    // MPQR_PREC_FP8 only: e4m3 operands of the far trailing update (kernels_fp8.hip)
    uint8_t* V8n = nullptr;   // [row][reflector of the block]  (2^8 V), ld = ld8k
    uint8_t* V8t = nullptr;   // [reflector of the block][row]  (2^8 V), ld = m_pad
    uint8_t* A8t = nullptr;   // [column][row]  (s A2)^T of the columns being updated, ld = m_pad
    uint8_t* Y8 = nullptr;    // [column][reflector]  (2^-2 Y), ld = ld8k
    long ld8k = 0; int v8_node = -1;          // node whose reflectors V8n / V8t currently hold
    float* Vf = nullptr;      // MPQR_PREC_FP32 only: fp32 reflectors [row][reflector], zero above the diagonal
    float* Yf = nullptr;      // MPQR_PREC_FP32 only: fp32 Y = X T'
    float* vdiag = nullptr;
    float* Xt = nullptr;   size_t xt_elems = 0;
    half_t* Yt = nullptr;  size_t yt_elems = 0;
    float* Xt1 = nullptr;  half_t* Yt1 = nullptr; size_t xt1_elems = 0;     // scratch of the far-update stream (look-ahead)
    float* Xt2 = nullptr;  half_t* Yt2 = nullptr; size_t xt2_elems = 0;   // scratch of the deferred in-block updates (apply_node lane 2, T stream)
    int pre_leaves = 0;                 // flat block: its first pre_leaves leaves were brought up to date by the previous block (run_block_loop)
    hipEvent_t ev_def = nullptr;        // chain -> T stream: a pre-updated leaf's reflectors and T are complete (deferred update may start)
    hipEvent_t ev_rest = nullptr;       // T stream -> chain: a leaf's deferred update of the rest of its block is complete (leaf-level look-ahead)
    bool rest_pending = false;          // ... and the chain stream has not waited for the last one yet
    std::vector<hipEvent_t> ev_node, ev_cols;          // per top-level node: reflectors ready / columns up to date
    float* S = nullptr;    size_t s_elems = 0;
    float* Sleaf = nullptr;       // 128 x 128 Gram of a leaf's fp16 reflectors (chain stream; S itself is used by the T stream)
    int* mid_counter = nullptr;   // arrival counter of leaf_mid_kernel's reduce workgroups (zero between launches)
    struct MidT { const float* Sp; int nslab; int sh, a0, c0, c1; float* T; half_t* Th; half_t* Tth; int ldt, ld; };
    const MidT* mid = nullptr;    // set by the flat schedule around one apply_node call: its X GEMM and the leaf's T go out as ONE launch
    bool leaf_mid = true;         // MPQR_LEAF_MID=0: X on the side stream, Gram sum and T as two launches on the chain (before round 4)
    bool v_clean = false;         // the reflector stores (Vh, Vt, vdiag) need no clearing before the next factorisation of this plan: they are
                                  // zero (fresh plan) or hold the reflectors of a clean single-pass factorisation of the same plan, every one of
                                  // which the next factorisation rewrites before it reads it
    // fused leaf (round 5): the chain touches only the next leaf's columns, in three launches (leaf_a / leaf_m / leaf_b, kernels_panel.hip)
    bool fused_leaf = true;       // MPQR_FUSED_LEAF=0: the seven-launch leaf of round 4
    float* Xp = nullptr;          // per-workgroup partials of X = (s P)^T V for the next panel (leaf_a), like Sp
    float* Xs = nullptr;          // their sum (LEAF_MID_MAX_GROUPS windows)
    half_t* Yfl = nullptr;        // Y = fp16(X T') for the next panel, 128 x 128
    int gram_ready_c0 = -1, gram_ready_rows = 0, gram_ready_n = 0;   // leaf_b left the partial Gram matrices (n of them) of the leaf that starts at this column, over this many rows
    int n_fused_leaves = 0;       // of the last mpqr_factor
    unsigned long long* dbg_stamps = nullptr; int dbg_stamps_n = 0;   // MPQR_DBG_STAMPS=1: start / end device times of every flat-schedule gh_solve (mapped host memory)
    std::vector<double> far_flops_tn;  // flops the X GEMM of the same record EXECUTED (Q formation skips identity / zero parts of Q: less than far_flops)
    int rest_seq = 0;             // value the T stream last published in tflag[3] behind a deferred update the chain waits for (polling plans)
    bool rest_split = false;      // ... and it was announced behind its first 128 columns only (fused leaf)
    bool next_block_flat = false; // run_block_loop: the block after the one being factored takes the flat schedule (its first leaf handles a pending deferred update)
    bool rest_in_solve = false;   // ... and the gh_solve launched last polls it at its end (no event wait in front of the next launch)
    int rest_first_cols = 0;      // apply_node, lane 2: the update's first this-many columns as a launch of their own, ev_rest recorded behind it
    bool rest_recorded = false;   // ... done: the caller does not record ev_rest again
    half_t* Wq = nullptr;         // Q formation: W = V T of every block pair ([rows from the pair's first 64-aligned row][K], fp16), left by merge_pair
    std::vector<long> wq_off;     // ... offset by node id, -1: none
    std::vector<char> wq_ready;   // ... W of this node matches its current T (set by pair_w, cleared by merge_pair)
    // drop-in call (mpqr_block_qr_f32): finished rows of the packed factor go to the caller's buffer while the factorisation is still running
    float* stream_out = nullptr;  // host (m+1) x n image, or nullptr
    std::vector<hipEvent_t> ev_rows;   // per top-level block t: recorded on the far stream behind far update t
    std::vector<char> rows_rec;        // ... recorded in this pass
    int rows_streamed = 0;             // packed rows [0, rows_streamed) are in the caller's buffer ...
    bool rows_valid = false;           // ... and still describe the result (single pass, no retry)
    bool defer_pair_w = false;    // merge_pair leaves W = V T to its caller (run_block_loop: a block later, where the far stream has room)
    bool q_all_ident = false;     // apply_node, Q formation: the matrix the node is applied to is still the identity (the first apply)
    int lane2_twait = 0;          // apply_node, lane 2: wait for this value of the chain's T word in front of Y = X T (0: nothing to wait for)
    int* tflag = nullptr;         // device word the chain publishes its progress in (leaf_xt_kernel / leaf_b_kernel), polled by the T stream (wait_flag_kernel)
    int tseq = 0, xt_pub = 0;     // last published value; value the next leaf_xt launch of apply_node is to publish (0: none)
    bool tpoll = true;            // MPQR_TPOLL=0: the T stream follows the chain through an event (costs the chain ~4 us per leaf)
    unsigned long long tpoll_ticks = 500000000ull;   // how long a wait_flag_kernel polls before it gives up (100 MHz ticks: 5 s; MPQR_TPOLL_TIMEOUT_MS)
    int n_tpoll_retries = 0;      // factorisations repeated with event hand-offs after a polling wait timed out (last mpqr_factor)
    float* P = nullptr;    int maxwg = 0;
    double* Gp = nullptr; double* Gs = nullptr; float* Cv = nullptr; int* dflag = nullptr;   // Gram-Householder leaf workspace
    float* Sp = nullptr;          // per-workgroup partial Grams of the fp16 reflectors (fused into gh_apply)
    bool robust = false;          // true: EVERY tall leaf is factored column by column instead of by Gram-Householder
    std::vector<char> leaf_robust;   // per tree node: this tall leaf was flagged by gh_solve and takes the robust path
    int nflag = 0;                // ints in dflag: one flag per tree node (gh_solve raises dflag[node id])
    int* hflag_host = nullptr;    // mapped host memory, one word per top-level block: a flagged Gram-Householder leaf raises its block's word,
    int* hflag_dev = nullptr;     // the thread that enqueues the block loop polls the words and stops enqueuing (run_block_loop)
    int flag_words = 0;           // number of words
    int cur_block = 0;            // the block whose leaves are being enqueued (their flags go to word cur_block)
    bool watch_flags = false;     // only mpqr_factor's block loop reacts to the word
    bool copied_in = false;       // mpqr_factor has just copied the input into the working matrix (together with the scale pass)
    bool pass_aborted = false;    // the last pass of the block loop stopped enqueuing at a flagged leaf
    int n_passes = 0, n_robust_leaves = 0;   // of the last mpqr_factor
    int n_gh_leaves = 0;                     // Gram-Householder leaves launched by the last mpqr_factor (all passes)
    int restart_block = 0;                   // top-level block the last pass of mpqr_factor started from
    int n_q_ident_rows = 0;                  // rows of X copied from V in the last Q formation (identity columns of Q)
    hipError_t async_err = hipSuccess; const char* async_what = "";   // first failed enqueue call of the current C-ABI call (HIPQ)
    bool dispatch_error = false;             // a GEMM enqueued on behalf of this handle found no kernel (gemm_dispatch)
    bool q_inited = false;                   // Q = I (and its fp16 shadow) was set up early, on the far stream beside the first panels
    std::vector<size_t> far_mark, chain_mark;   // event-pool positions at the start of every top-level block (a restart rewinds to them)
    float us_gh_solve = 0.f;                 // mpqr_bench_leaf_solve's last result
    float* rbTf = nullptr; half_t* rbTh = nullptr; half_t* rbTth = nullptr; size_t rb_elems = 0;   // T arena of a robust leaf's sub-tree
    bool force32 = false;         // tree building: only 32-column leaves (sub-tree of a robustly factored tall leaf)
    int gh_min_rows = 128;        // leaves with more rows than this below their first column use Gram-Householder
    bool tail_leaf = true;        // the last <= 128 rows as one leaf_tail_kernel leaf (MPQR_TAIL_LEAF=0: 32-column leaves, as before round 4)
    float* tmp1 = nullptr; float* tmp2 = nullptr; size_t tmp_elems = 0;
    float* Tf = nullptr; half_t* Th = nullptr; half_t* Tth = nullptr; size_t t_elems = 0;
    double* dmetric = nullptr;   // 8 doubles
    float* dscalar = nullptr;    // 4 floats
    float* dstage = nullptr; size_t stage_elems = 0;   // packed-factor staging for D2H

    std::vector<Node> nodes;
    std::vector<int> tops;
    // Q formation over PAIRS of top-level blocks (K = 2 outer_block: the GEMMs run at a higher rate; the pair's T is merged
    // in the background, T_LR = -T_L (V_L^T V_R) T_R, on the far-update stream with its own scratch)
    std::vector<int> qpair;       // per top index t: id of the pair node whose RIGHT child is top t, else -1
    bool pairs_ready = false;     // the pair T's of the current factorisation are (enqueued to be) complete
    float host_enqueue_ms = 0.f;  // host time the last block loop took to enqueue (everything before its one synchronisation)
    // tall matrices (m >= 3 n): Q = I - (V T) V^T in ONE product over all reflectors instead of the backward accumulation
    // (2 m^2 n + m n^2 flops instead of ~4 m^2 n, and no read-modify-write): needs the T of ALL reflectors, built block
    // column by block column in the background (merge_prefix)
    int qroot = -1;                               // node id of the full-width T (-1: backward accumulation)
    std::vector<std::vector<int>> qmerge_after;   // per top index t: prefix nodes that can be completed once block t is factored
    half_t* Wh = nullptr;                         // W = V T, fp16 [row][reflector]
    size_t q_first = (size_t)-1;  // first far_ev slot used by Q formation (its applies are timed like the far updates)
    float* S2 = nullptr; size_t s2_elems = 0; float* tmp1b = nullptr; float* tmp2b = nullptr; size_t tmpb_elems = 0;
    // transposed fp16 shadow of Q, Qt[column][row], kept up to date by the epilogue of Q -= V Y^T: the next X = Q2^T V
    // reads it with LDS-DMA like any fp16 operand (the fp32 operand path converts and transposes in registers: 620 TFLOP/s)
    half_t* Qt = nullptr; long ldqt = 0;
    half_t* shadow = nullptr; long ldshadow = 0;      // set by form_q around its applies (apply_node, lane 0)
    bool shadow_write = true;                         // false: this apply still reads the shadow but does not update it
    int q_ident_cols = 0;                             // form_q: this many leading columns of the apply's range are still identity columns
    // the same for the trailing matrix: At[column][row] = fp16(a_scale * A), written by every far update's epilogue, read by
    // the NEXT far update's X = A2^T V (far update 0 reads the fp32 matrix: nothing has written the shadow yet)
    half_t* At = nullptr; long ldat = 0; bool at_read = false;
    // X = C2^T V leaves its GEMM as fp16 hi + lo parts (per stream lane) and Y = X T' takes both on the ping-pong kernel
    half_t* Xh = nullptr; half_t* Xl = nullptr; half_t* Xh1 = nullptr; half_t* Xl1 = nullptr;
    // 1-D block-cyclic column distribution (world == 1: everything local)
    int world = 1, rank = 0;
    int nloc = 0;        // local columns of A
    int qloc = 0;        // local columns of Q
    float* Aeff = nullptr;   // dA shifted so that Aeff[row*lda + GLOBAL column] addresses the column being factored
    float a_scale = 1.f;
    bool factored = false, q_formed = false;

    // timing
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> far_ev;   // pool, 4 per recorded far update: around op1 / op3
    size_t far_used = 0;
    std::vector<double> far_flops;
    std::vector<double> far_bytes;     // algorithmic HBM bytes of the same launches' C -= V Y^T
    std::vector<int> far_dims;         // M, N, K of the same launches (3 per record): mpqr_get_update_records
    std::vector<hipEvent_t> chain_ev; // pool, 2 per top-level block: around factor_node on the chain stream
    size_t chain_used = 0;
    mpqr_timings last_t;
};

namespace {

thread_local std::string g_create_err;

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            char buf_[512];                                                                      \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (h)->err = buf_;                                                                     \
            return MPQR_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

// enqueue calls inside the block loop (event records / waits, device-to-device copies) cannot return from the middle of the schedule:
// the FIRST failure is kept in the handle and reported by the C-ABI entry that enqueued it (mpqr_factor, mpqr_sync, the mpqr_dist_* steps)
#define HIPQ(h, call)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess && (h)->async_err == hipSuccess) { (h)->async_err = e_; (h)->async_what = #call; } \
    } while (0)

int fail(mpqr_handle_t h, int code, const char* msg) {
    if (h) h->err = msg;
    return code;
}

template <typename T>
int dalloc(mpqr_handle_t h, T** p, size_t elems) {
    *p = nullptr;
    if (elems == 0) elems = 1;
    hipError_t e = hipMalloc((void**)p, elems * sizeof(T));
    if (e != hipSuccess) {
        h->err = std::string("hipMalloc failed: ") + hipGetErrorString(e);
        return MPQR_ERR_ALLOC;
    }
    return MPQR_OK;
}

void free_plan(mpqr_handle_t h) {
    void* ptrs[] = {h->dA, h->dA0, h->dQ, h->Vh, h->Vt, h->vdiag, h->Xt, h->Yt, h->S, h->P, h->tmp1, h->tmp2,
                    h->Tf, h->Th, h->Tth, h->dstage, h->Gp, h->Gs, h->Cv, h->dflag, h->Vf, h->Yf, h->Xt1, h->Yt1, h->Sp,
                    h->rbTf, h->rbTh, h->rbTth, h->Sleaf, h->mid_counter, h->tflag, h->V8n, h->V8t, h->A8t, h->Y8, h->Xt2, h->Yt2, h->Xp, h->Xs, h->Yfl,
                    h->S2, h->tmp1b, h->tmp2b, h->Qt, h->At, h->Xh, h->Wh, h->Xl, h->Xh1, h->Xl1, h->Wq};
    for (void* p : ptrs) if (p) MPQR_IGNORE(hipFree(p));
    if (h->hflag_host) MPQR_IGNORE(hipHostFree(h->hflag_host));
    if (h->dbg_stamps) { MPQR_IGNORE(hipHostFree(h->dbg_stamps)); h->dbg_stamps = nullptr; h->dbg_stamps_n = 0; }
    h->hflag_host = h->hflag_dev = nullptr; h->flag_words = 0; h->cur_block = 0;
    h->S2 = h->tmp1b = h->tmp2b = nullptr; h->s2_elems = 0; h->Qt = nullptr; h->shadow = nullptr; h->At = nullptr; h->at_read = false; h->Xh = nullptr; h->Xl = h->Xh1 = h->Xl1 = nullptr; h->Wh = nullptr; h->qroot = -1; h->qmerge_after.clear(); h->qpair.clear(); h->pairs_ready = false;
    h->dA = h->dA0 = h->dQ = nullptr; h->Vh = h->Vt = nullptr; h->vdiag = nullptr; h->Xt = nullptr; h->Yt = nullptr;
    h->S = nullptr; h->P = nullptr; h->tmp1 = h->tmp2 = nullptr; h->Tf = nullptr; h->Th = h->Tth = nullptr;
    h->dstage = nullptr; h->stage_elems = 0;
    h->Gp = nullptr; h->Gs = nullptr; h->Cv = nullptr; h->dflag = nullptr; h->Vf = nullptr; h->Yf = nullptr;
    h->Xt1 = nullptr; h->Yt1 = nullptr; h->Wq = nullptr; h->wq_off.clear(); h->wq_ready.clear(); h->Sp = nullptr; h->Xt2 = nullptr; h->Yt2 = nullptr; h->xt2_elems = 0;
    h->rbTf = nullptr; h->rbTh = h->rbTth = nullptr; h->rb_elems = 0; h->nflag = 0; h->leaf_robust.clear(); h->Sleaf = nullptr; h->mid_counter = nullptr; h->tflag = nullptr; h->Xp = nullptr; h->Xs = nullptr; h->Yfl = nullptr; h->gram_ready_c0 = -1;
    h->V8n = h->V8t = h->A8t = h->Y8 = nullptr; h->v8_node = -1;
    for (hipEvent_t e : h->chain_ev) MPQR_IGNORE(hipEventDestroy(e));
    h->chain_ev.clear(); h->chain_used = 0; h->far_used = 0;
    for (hipEvent_t e : h->ev_node) MPQR_IGNORE(hipEventDestroy(e));
    for (hipEvent_t e : h->ev_cols) MPQR_IGNORE(hipEventDestroy(e));
    for (hipEvent_t e : h->ev_cols2) MPQR_IGNORE(hipEventDestroy(e));
    h->ev_node.clear(); h->ev_cols.clear(); h->ev_cols2.clear();
    for (hipEvent_t e : h->far_ev) MPQR_IGNORE(hipEventDestroy(e));
    h->far_ev.clear();
    h->nodes.clear(); h->tops.clear();
    h->planned = false; h->have_input = false; h->factored = false; h->q_formed = false; h->q_inited = false;
    h->far_mark.clear(); h->chain_mark.clear();
}

// ---- column-range tree
// A node starting at column c0 is "tall" when more than gh_min_rows rows lie below its diagonal: tall leaves are
// up to 128 columns wide (Gram-Householder), short ones up to 32 (one workgroup, register resident).
// The last <= 128 rows of a (nearly) square matrix have no tall part: one 128-wide "tail" leaf (leaf_tail_kernel, plain Householder in
// one workgroup) instead of four 32-column leaves and their merges.
bool is_tail(mpqr_handle_t h, int c0) { return !h->force32 && h->tail_leaf && h->m - c0 <= 128; }
int leaf_width(mpqr_handle_t h, int c0) { return (!h->force32 && (h->m - c0 > h->gh_min_rows || is_tail(h, c0))) ? 128 : 32; }
bool is_leaf(mpqr_handle_t h, int c0, int c1) {
    const int lw = leaf_width(h, c0);
    return (c1 - c0) <= lw && (c0 / lw) == ((c1 - 1) / lw) && (lw == 32 || h->m - c1 >= 1 || is_tail(h, c0));
}

int pick_split(mpqr_handle_t h, int c0, int c1, int r) {
    const int mid = (c0 + c1) / 2;
    auto nearest_multiple = [&](int a) -> int {
        int lo = rup(c0 + 1, a), best = -1;
        for (int x = lo; x < c1; x += a)
            if (best < 0 || abs(x - mid) < abs(best - mid)) best = x;
        return best;
    };
    // Tall ranges are cut at multiples of 128 whatever the caller's r: a Gram-Householder leaf costs nearly the same for 64 columns as
    // for 128 (its launches and hand-offs, not its arithmetic), so r = 64 on 2048 rows (BASELINE config 2) took 32 leaves where 16 do.
    // r is the reference's blocking parameter, not part of the result: the reflectors, R and Q are the same for every blocking
    // (Cuda/qr.cu:1075-1076 only groups the same Householder steps).  Short ranges keep the caller's r-wide panels.
    int s = leaf_width(h, c0) == 128 ? nearest_multiple(128) : -1;
    if (s < 0) s = nearest_multiple(r);
    if (s < 0) s = nearest_multiple(32);
    if (s < 0) s = mid;
    return s;
}

int build_tree(mpqr_handle_t h, int c0, int c1) {
    Node nd;
    nd.c0 = c0; nd.c1 = c1; nd.a0 = rdown(c0, 64); nd.a1 = rup(c1, 64); nd.ldt = nd.a1 - nd.a0;
    nd.left = nd.right = -1; nd.toff = 0; nd.tld = nd.ldt;
    const int id = (int)h->nodes.size();
    nd.id = id;
    h->nodes.push_back(nd);
    if (!is_leaf(h, c0, c1)) {
        const int cm = pick_split(h, c0, c1, h->r);
        const int l = build_tree(h, c0, cm);
        const int rr = build_tree(h, cm, c1);
        h->nodes[id].left = l; h->nodes[id].right = rr;
    }
    return id;
}

// ---- GEMM wrappers -------------------------------------------------------------
// large shapes go to the 256 x 256 tile kernel, everything else to the 128 x 128 one
// 256 x 256 output tiles from which the 256-wide kernels take a GEMM (test hook: MPQR_GEMM2_MIN_TILES=1 forces them early).
// apply_node derives from the SAME number whether an update may hand X over as fp16 hi + lo parts, which only those kernels honour.
static long gemm2_min_tiles() {
    static const long v = []() { const char* e = getenv("MPQR_GEMM2_MIN_TILES"); return e ? atol(e) : 48L; }();
    return v;
}
// a GEMM that finds no kernel marks ITS handle (thread-local "current handle", set by every C-ABI entry that enqueues GEMMs):
// mpqr_factor / form_q / the mpqr_dist_* steps report it instead of returning garbage, and one rank thread of the multi-GPU host
// can neither consume nor inherit another handle's error
static thread_local mpqr_handle_t t_dispatch_handle = nullptr;
void gemm_dispatch(AMode am, EMode em, const GemmArgs& g, hipStream_t s) {
    const long tiles = (long)(g.M / 256) * (g.N / 256);
    if ((g.nsplit <= 1 || (em == E_STORE_F32 && am == A_F32T)) && (g.nslab_in <= 1) && g.M >= 256 && g.N >= 256 &&
        tiles >= gemm2_min_tiles() && (g.K % 64) == 0 &&
        launch_gemm2_f16(am, em, g, s, 0))
        return;
    if (!launch_gemm_f16(am, em, g, s) && t_dispatch_handle) t_dispatch_handle->dispatch_error = true;
}

int choose_split(int M, int N, int K, size_t cap_elems, long slab) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const int ktiles = K / 64;
    int ns = 1;
    static const int wgs = []() { const char* e = getenv("MPQR_SPLIT_WGS"); return e ? std::max(64, atoi(e)) : 512; }();   // tuning hook
    if (tiles < 128) ns = std::min(ktiles / 2, std::max(1, wgs / tiles));   // >= 2 k-tiles per slice, ~2 workgroups per CU
    else {
        // large-tile (256 x 256, one workgroup per CU) regime: keep every CU busy when the output has few tiles
        const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
        if (t256 < 192) ns = (int)std::min<long>(ktiles / 16, (256 + t256 - 1) / t256);
    }
    static const int cap = []() { const char* e = getenv("MPQR_SPLIT_CAP"); return e ? std::max(1, atoi(e)) : 32; }();   // tuning hook (64: 49.5, 32: 48.6, 16: 49.0 ms at 16384^2)
    ns = std::min(ns, cap);  // slabs are summed by launch_slab_reduce right after the producer
    while (ns > 1 && (size_t)ns * (size_t)slab > cap_elems) ns--;
    return std::max(ns, 1);
}

// S (slabs, KrL x KrR, ld = KrR) = V_L^T V_R over rows >= 64-aligned start of R
int gram(mpqr_handle_t h, const Node& L, const Node& R, int* nslab, long* slab, hipStream_t st, float* Sbuf = nullptr,
         size_t s_cap = 0) {
    if (!Sbuf) { Sbuf = h->S; s_cap = h->s_elems; }
    const int rlo = rdown(R.c0, 64);
    if (h->opts.precision == MPQR_PREC_FP32) {            // exact-f32 products of the fp32 reflectors
        SgemmArgs g{};
        g.A = h->Vf + (long)rlo * h->n_pad + L.a0; g.lda = h->n_pad; g.transA = 1;
        g.B = h->Vf + (long)rlo * h->n_pad + R.a0; g.ldb = h->n_pad; g.transB = 0;
        g.C = Sbuf; g.ldc = R.ldt; g.M = L.ldt; g.N = R.ldt; g.K = h->m_pad - rlo; g.alpha = 1.f; g.beta = 0.f; g.nslab_a = 1;
        launch_sgemm(g, st);
        *nslab = 1; *slab = (long)L.ldt * R.ldt;
        return MPQR_OK;
    }
    GemmArgs g{};
    g.A = h->Vt + (long)L.a0 * h->ldvt + rlo;  g.lda = h->ldvt;
    g.Bt = h->Vt + (long)R.a0 * h->ldvt + rlo; g.ldb = h->ldvt;
    g.C = Sbuf; g.ldc = R.ldt;
    g.M = L.ldt; g.N = R.ldt; g.K = h->m_pad - rlo;
    g.alpha = 1.f; g.in_scale = 1.f;
    *slab = (long)L.ldt * R.ldt;
    g.nsplit = choose_split(g.M, g.N, g.K, s_cap, *slab);
    g.slab_out_stride = *slab;
    launch_gemm_f16(A_H16, E_STORE_F32, g, st);
    if (g.nsplit > 1) launch_slab_reduce(Sbuf, g.nsplit, *slab, *slab, Sbuf, st);
    *nslab = 1;
    return MPQR_OK;
}

static void rest_done(mpqr_handle_t h, hipStream_t st);
// C[rows >= rdown(nd.c0,64)][cols clo..chi) <- (I - V T' V^T) C,  T' = T^T (trans_t) or T
void apply_node(mpqr_handle_t h, const Node& nd, float* C, long ldc, int clo, int chi, bool trans_t, float in_scale,
                bool record, int lane = 0, bool far = false) {
    if (chi <= clo) return;
    static const int dbg_nofar = []() { const char* e = getenv("MPQR_DBG_NOFAR"); return e ? atoi(e) : 0; }();
    if (dbg_nofar && lane == 1) return;                     // timing experiment only (results are garbage): the chain without far updates beside it
    // lane 1: far-update stream with its own scratch; lane 2: a deferred in-block update, all of it on the T stream, own scratch
    hipStream_t st = lane == 1 ? h->s1 : lane == 2 ? h->sT : (h->inblock_stream ? h->inblock_stream : h->s0);
    float* const Xt = lane == 1 ? h->Xt1 : lane == 2 ? h->Xt2 : h->Xt;
    half_t* const Yt = lane == 1 ? h->Yt1 : lane == 2 ? h->Yt2 : h->Yt;
    const int rlo = rdown(nd.c0, 64);
    const int Kw = h->m_pad - rlo;
    if (h->opts.precision == MPQR_PREC_FP32) {            // dev_block_qr_wy twin: every product on the exact-f32 MFMA
        const int M1 = chi - clo, Kr = nd.ldt;
        const float* Vn = h->Vf + (long)rlo * h->n_pad + nd.a0;
        SgemmArgs x{};                                     // X[M1 x Kr] = C2^T V
        x.A = C + (long)rlo * ldc + clo; x.lda = ldc; x.transA = 1;
        x.B = Vn; x.ldb = h->n_pad; x.transB = 0;
        x.C = h->Xt; x.ldc = Kr; x.M = M1; x.N = Kr; x.K = Kw; x.alpha = 1.f; x.beta = 0.f; x.nslab_a = 1;
        launch_sgemm(x, h->s0);
        SgemmArgs y{};                                     // Y = X T'   (T' = T for the trailing update, T^T for Q)
        y.A = h->Xt; y.lda = Kr; y.transA = 0; y.nslab_a = 1;
        y.B = h->Tf + nd.toff; y.ldb = Kr; y.transB = trans_t ? 0 : 1;
        y.C = h->Yf; y.ldc = Kr; y.M = M1; y.N = Kr; y.K = Kr; y.alpha = 1.f; y.beta = 0.f;
        launch_sgemm(y, h->s0);
        SgemmArgs u{};                                     // C2 -= V Y^T
        u.A = Vn; u.lda = h->n_pad; u.transA = 0; u.nslab_a = 1;
        u.B = h->Yf; u.ldb = Kr; u.transB = 1;
        u.C = C + (long)rlo * ldc + clo; u.ldc = ldc; u.M = Kw; u.N = M1; u.K = Kr; u.alpha = -1.f; u.beta = 1.f;
        launch_sgemm(u, h->s0);
        (void)record;
        return;
    }
    const int clo_al = rdown(clo, 32);
    const int M1 = chi - clo_al;
    const int Kr = nd.ldt;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    if (record && h->far_used + 4 > h->far_ev.size()) record = false;      // pool exhausted (sized in plan_common)
    if (record) {
        e0 = h->far_ev[h->far_used]; e1 = h->far_ev[h->far_used + 1]; e2 = h->far_ev[h->far_used + 2]; e3 = h->far_ev[h->far_used + 3];
        h->far_used += 4;
    }
    // op1: Xt[M1 x Kr] = (in_scale * C2)^T V
    GemmArgs g1{};
    g1.A = C + (long)rlo * ldc + clo_al; g1.lda = ldc;
    g1.Bt = h->Vt + (long)nd.a0 * h->ldvt + rlo; g1.ldb = h->ldvt;
    g1.C = Xt; g1.ldc = Kr;
    g1.M = M1; g1.N = Kr; g1.K = Kw;
    g1.in_scale = in_scale; g1.alpha = 1.f;
    const long slab = (long)M1 * Kr;
    g1.nsplit = choose_split(M1, Kr, Kw, lane == 2 ? h->xt2_elems : h->xt_elems, slab);
    g1.slab_out_stride = slab;
    if (record) HIPQ(h, hipEventRecord(e0, st));
    hipStream_t st1 = (h->op1_stream && lane == 0) ? h->op1_stream : st;     // flat schedule: X on the side stream, T on the chain
    // MPQR_PREC_FP8: the two large GEMMs of a FAR update take e4m3 operands (kernels_fp8.hip); shapes the fp8 kernel does
    // not cover (K not a multiple of 128) stay on the fp16 path
    const bool f8 = far && h->V8n && (Kr % 128) == 0 && (Kw % 128) == 0 && (rlo % 16) == 0 && nd.a0 == nd.c0;
    const bool a_shadow = h->At && lane == 1 && far && !f8 && C == h->dA;   // fp16 shadow of the trailing matrix (far updates)
    // X keeps ~22 bits through Y = X T' (fp16 hi + lo parts, MPQR_XSPLIT=0: one fp16 rounding as in round 1); the fp16-X
    // shortcut of Q formation (MPQR_X16=1) is the opposite trade
    constexpr int xsplit = 1, x16_env = 1;                  // (round 1's single-rounding X and the fp32 X through memory lost for two rounds: removed)
    // big unsplit updates: X goes from its GEMM to the next one as fp16 (hi, and lo when xsplit) instead of fp32
    half_t* const Xhi = lane == 1 ? h->Xh1 : lane == 2 ? nullptr : h->Xh;
    // (Q formation: Q has no dominant component, the lo part buys 1.5 % of backward error and 7 % of ||Q^T Q - I|| for
    // 0.7 ms at 16384^2 -- MPQR_QSPLIT=1 turns it on there as well)
    static const int qsplit = []() { const char* e = getenv("MPQR_QSPLIT"); return e ? atoi(e) : 0; }();
    const bool q_apply = h->shadow && lane == 0 && !far;
    half_t* const Xlo = (xsplit && (!q_apply || qsplit)) ? (lane == 1 ? h->Xl1 : lane == 2 ? nullptr : h->Xl) : nullptr;
    // Q formation's first applies are small (2048^2, 4096^2 at 16384^2: 64 / 128 tiles of 256^2): split-K sent them to the 128-tile kernel
    // with fp32 slabs and Y = X T' to the fp32-staging kernel -- 265 + 400 us for 40 + 160 us of GEMM work (tools/trace_q.sh).  One K range on
    // the 256-tile kernels instead, even with half the CUs idle.
    if (q_apply && Xhi && g1.nsplit > 1 && M1 >= 256 && Kr >= 256 && (Kr % 64) == 0 && (long)(M1 / 256) * (Kr / 256) >= gemm2_min_tiles()) g1.nsplit = 1;
    const bool x16 = x16_env && Xhi && (!xsplit || Xlo || q_apply) && !f8 && g1.nsplit == 1 && M1 >= 256 && Kr >= 256 && (Kr % 64) == 0 &&
                     (long)(M1 / 256) * (Kr / 256) >= gemm2_min_tiles() && h->opts.precision != MPQR_PREC_FP32;
    // one leaf (128 reflectors) onto a few columns: slab sum and Y = X T' in one small kernel (leaf_xt_kernel)
    const bool fuse_xt = !f8 && !far && lane != 1 && Kr == 128 && M1 <= 4096 && !(h->shadow && lane == 0);
    if (f8) {
        if (h->v8_node != nd.id) {                          // the block's reflectors, once per block: 2^8 V in both layouts
            launch_quant_h16_fp8(h->Vh + (long)rlo * h->ldvh + nd.a0, h->ldvh, h->V8n, h->ld8k, Kw, Kr, 256.f, st);
            launch_quant_h16_fp8(h->Vt + (long)nd.a0 * h->ldvt + rlo, h->ldvt, h->V8t, h->m_pad, Kr, Kw, 256.f, st);
            h->v8_node = nd.id;
        }
        launch_quant_transpose_f32_fp8(C + (long)rlo * ldc + clo_al, ldc, h->A8t, h->m_pad, Kw, M1, in_scale, st);   // (s A2)^T
        GemmArgs f1 = g1;
        f1.A = h->A8t; f1.lda = h->m_pad; f1.Bt = (const half_t*)h->V8t; f1.ldb = h->m_pad; f1.alpha = 1.f / 256.f;
        const int t256 = ((M1 + 255) / 256) * ((Kr + 255) / 256);
        f1.nsplit = 1;
        if (t256 < 192) f1.nsplit = std::max(1, std::min({(Kw / 128) / 8, (256 + t256 - 1) / t256, 16}));
        while (f1.nsplit > 1 && (size_t)f1.nsplit * (size_t)slab > h->xt_elems) f1.nsplit--;
        launch_gemm_fp8(E_STORE_F32, f1, st);
        if (f1.nsplit > 1) launch_slab_reduce(Xt, f1.nsplit, slab, slab, Xt, st);
    } else {
    if (a_shadow && h->at_read) {                          // trailing matrix: the previous far update left fp16(s A2)^T
        g1.A = h->At + (long)clo_al * h->ldat + rlo; g1.lda = h->ldat;
        g1.in_scale = 1.f;
        if (x16) { g1.C = Xhi; g1.C2 = Xlo; gemm_dispatch(A_H16, E_STORE_H16, g1, st1); }
        else gemm_dispatch(A_H16, E_STORE_F32, g1, st1);
    } else
    if (h->shadow && lane == 0 && !far) {                  // fp16 operand, already [column][row]: C2^T = shadow rows
        g1.A = h->shadow + (long)clo_al * h->ldshadow + rlo; g1.lda = h->ldshadow;
        g1.in_scale = 1.f;
        if (x16) {                                          // X leaves as fp16 hi (+ lo)
            g1.C = Xhi; g1.C2 = Xlo;
            // backward accumulation: the node's own columns of Q are still columns of the identity, so their rows of
            // X = Q2^T V are rows of V (exact in fp16): copied, and the GEMM starts behind them
            const int idc = (h->q_ident_cols > 0 && clo == clo_al && nd.a0 == nd.c0 && rlo == nd.c0 && clo == nd.c0) ?
                            std::min({h->q_ident_cols, M1 - 256, (nd.c1 - nd.c0) / 256 * 256}) : 0;
            bool x_done = false;
            if (h->q_all_ident && clo == clo_al && nd.a0 == nd.c0 && rlo == nd.c0 && clo == nd.c0) {
                // the FIRST apply of Q formation: Q2 = I, so X = Q2^T V = V, exact in fp16 -- a copy, no GEMM
                HIPQ(h, hipMemcpy2DAsync(Xhi, (size_t)Kr * sizeof(half_t), h->Vh + (long)nd.c0 * h->ldvh + nd.a0, h->ldvh * sizeof(half_t),
                                       (size_t)Kr * sizeof(half_t), M1, hipMemcpyDeviceToDevice, st1));
                if (Xlo) HIPQ(h, hipMemsetAsync(Xlo, 0, (size_t)M1 * Kr * sizeof(half_t), st1));
                h->n_q_ident_rows += M1;
                g1.M = 0; x_done = true;
            } else
            if (idc >= 256 && (long)((M1 - idc) / 256) * (Kr / 256) >= gemm2_min_tiles()) {   // (the rest still goes to the 256-wide kernel)
                HIPQ(h, hipMemcpy2DAsync(Xhi, (size_t)Kr * sizeof(half_t), h->Vh + (long)nd.c0 * h->ldvh + nd.a0, h->ldvh * sizeof(half_t),
                                       (size_t)Kr * sizeof(half_t), idc, hipMemcpyDeviceToDevice, st1));
                if (Xlo) HIPQ(h, hipMemsetAsync(Xlo, 0, (size_t)idc * Kr * sizeof(half_t), st1));
                g1.A = h->shadow + (long)(clo_al + idc) * h->ldshadow + rlo;
                g1.C = Xhi + (long)idc * Kr; if (Xlo) g1.C2 = Xlo + (long)idc * Kr;
                g1.M = M1 - idc;
                h->n_q_ident_rows += idc;
                // ... and the OTHER columns of Q (right of the node's own) are still zero in the node's own rows [c0, c1): every apply so far
                // touched rows >= its own first column >= c1 only.  Their K range starts at row c1 (round 5: 14 % of Q formation's X flops)
                if (idc == nd.c1 - nd.c0 && (idc % 64) == 0 && Kw - idc >= 64) {
                    g1.A = (const half_t*)g1.A + idc; g1.Bt += idc; g1.K = Kw - idc;
                }
            }
            if (!x_done) gemm_dispatch(A_H16, E_STORE_H16, g1, st1);
        }
        else gemm_dispatch(A_H16, E_STORE_F32, g1, st1);
    } else if (x16) {
        g1.C = Xhi; g1.C2 = Xlo;
        gemm_dispatch(A_F32T, E_STORE_H16, g1, st1);
    } else if (h->mid && lane == 0 && fuse_xt && st1 == st) {
        // flat schedule: X beside the leaf's Gram sum and T, one launch (kernels_panel.hip: leaf_mid_kernel); one workgroup per CU there
        const int gx = (M1 + 127) / 128;
        static const int mid_wgs = []() { const char* e = getenv("MPQR_MID_WGS"); return e ? std::max(8, atoi(e)) : 128; }();   // tuning hook (48 / 64 / 96 / 128 / 176: 35.7 / 35.4 / 35.0 / 35.0 / 35.1-35.2 ms)
        // tall leaves with few column tiles (leaf-level look-ahead: ONE tile): the general split cap (32) leaves every workgroup ~30 K tiles of
        // 64 at 65536 rows, 45 us of load -> LDS -> MFMA round trips beside a 26 us T path; ~12 K tiles per workgroup instead
        g1.nsplit = std::max(g1.nsplit, std::min(Kw / 64 / 12, mid_wgs / gx));
        while (g1.nsplit > 1 && (size_t)g1.nsplit * (size_t)slab > h->xt_elems) g1.nsplit--;
        while (g1.nsplit > 1 && gx * g1.nsplit > mid_wgs) g1.nsplit--;
        const mpqr_handle_s::MidT& mt = *h->mid;
        launch_leaf_mid(g1, mt.Sp, mt.nslab, h->Sleaf, mt.sh, h->mid_counter, mt.a0, mt.c0, mt.c1, mt.T, mt.Th, mt.Tth, mt.ldt, mt.ld, st);
        h->mid = nullptr;                                   // consumed
    } else
    gemm_dispatch(A_F32T, E_STORE_F32, g1, st1);
    if (g1.nsplit > 1 && !fuse_xt) launch_slab_reduce(Xt, g1.nsplit, slab, slab, Xt, st1);
    }
    if (st1 != st) { HIPQ(h, hipEventRecord(h->ev_x, st1)); HIPQ(h, hipStreamWaitEvent(st, h->ev_x, 0)); }
    if (record) HIPQ(h, hipEventRecord(e1, st));
    if (lane == 2 && h->lane2_twait) {                      // fused leaf: T_j comes from the chain stream's leaf_m; its progress word is published by leaf_b
        launch_wait_flag(h->tflag, h->lane2_twait, h->hflag_dev + h->flag_words - 1, h->tpoll_ticks, st, 0);
        h->lane2_twait = 0;
    }
    const bool use_w = q_apply && x16 && !Xlo && !trans_t && h->Wq && nd.id >= 0 && nd.id < (int)h->wq_off.size() && h->wq_off[nd.id] >= 0 &&
                       h->wq_ready[nd.id] && nd.a0 == nd.c0 && !fuse_xt;
    // op2: Yt[M1 x Kr] = fp16( Xt * T' ) -- the first use of T: it was built on the T stream beside op1
    if (st1 == st && h->tq_on && nd.id >= 0 && nd.id < (int)h->ev_T.size()) HIPQ(h, hipStreamWaitEvent(st, h->ev_T[nd.id], 0));
    GemmArgs g2{};
    g2.A = Xt; g2.lda = Kr; g2.nslab_in = 1; g2.slab_in_stride = slab;
    g2.Bt = (trans_t ? h->Tth : h->Th) + nd.toff; g2.ldb = nd.tld;
    g2.C = Yt; g2.ldc = Kr;
    g2.M = M1; g2.N = Kr; g2.K = Kr;
    g2.in_scale = 1.f; g2.alpha = 1.f; g2.nsplit = 1;
    g2.tri = trans_t ? 2 : 1;                              // T is upper triangular: T^T rows end at the diagonal, T rows start there
    g2.cscale = h->Tf + nd.toff; g2.cscale_ld = (long)nd.tld + 1;   // tau_n: the fp16 T's have a unit diagonal
    if (fuse_xt) {
        launch_leaf_xt(Xt, g1.nsplit, slab, M1, g2.Bt, g2.ldb, g2.tri, Yt, Kr, g2.cscale, g2.cscale_ld, st, h->xt_pub ? h->tflag : nullptr, h->xt_pub);
        h->xt_pub = 0;
    }
    else if (use_w) { /* Q formation with W = V T at hand: Q2 -= W X^T below, no Y */ }
    else if (x16) { g2.A = Xhi; g2.A2 = Xlo; gemm_dispatch(A_H16, E_STORE_H16, g2, st); }
    else
    gemm_dispatch(A_F32S, E_STORE_H16, g2, st);
    // op3: C2 -= (1/in_scale) V Yt^T
    GemmArgs g3{};
    g3.A = h->Vh + (long)rlo * h->ldvh + nd.a0; g3.lda = h->ldvh;
    g3.Bt = Yt; g3.ldb = Kr;
    g3.C = C + (long)rlo * ldc + clo_al; g3.ldc = ldc;
    g3.M = Kw; g3.N = M1; g3.K = Kr;
    g3.col_lo = clo - clo_al; g3.alpha = 1.0f / in_scale; g3.in_scale = 1.f; g3.nsplit = 1;
    if (use_w) { g3.A = h->Wq + h->wq_off[nd.id]; g3.lda = Kr; g3.Bt = Xhi; }
    if (a_shadow) { g3.Ct = h->At + (long)clo_al * h->ldat + rlo; g3.ldct = h->ldat; g3.ct_scale = in_scale; }
    {   // Far updates and Q formation stream their C tiles (and the fp16 shadow) once per launch, ~2 GB each: read and written with the
        // non-temporal cache policy they leave the chain's working set (panel columns, V, X, T) in the caches.  16384^2: far nn
        // 4.75 -> 4.57 ms, chain -0.2 ms, Q formation 8.0 -> 7.85 ms; 65536 x 8192: factorisation -0.7 ms (A/B pairs on one box).
        // MPQR_NT_C: bit 0 far updates, bit 1 Q formation (A/B hook).
        static const int nt_env = []() { const char* e = getenv("MPQR_NT_C"); return e ? atoi(e) : 3; }();
        if ((nt_env & 1) && far) g3.nt_c = 1;
        if ((nt_env & 2) && q_apply) g3.nt_c = 1;
        if ((nt_env & 4) && lane == 2) g3.nt_c = 1;        // (A/B hook: the deferred in-block updates of the T stream)
    }
    if (h->shadow && h->shadow_write && lane == 0 && !far) { g3.Ct = h->shadow + (long)clo_al * h->ldshadow + rlo; g3.ldct = h->ldshadow; g3.ct_scale = in_scale; }
    if (record) HIPQ(h, hipEventRecord(e2, st));
    if (f8) {
        launch_quant_h16_fp8(Yt, Kr, h->Y8, h->ld8k, M1, Kr, 0.25f, st);                       // 2^-2 Y
        GemmArgs f3 = g3;
        f3.A = h->V8n; f3.lda = h->ld8k; f3.Bt = (const half_t*)h->Y8; f3.ldb = h->ld8k;
        f3.alpha = g3.alpha * (4.f / 256.f);
        launch_gemm_fp8(E_SUB_F32, f3, st);
    } else if (lane == 2 && h->rest_first_cols > 0 && clo == clo_al && !g3.Ct && M1 == h->rest_first_cols) {
        gemm_dispatch(A_H16, E_SUB_F32, g3, st);            // (this call's range IS those columns: a pre-updated leaf's first piece)
        rest_done(h, st); h->rest_recorded = true;
    } else if (lane == 2 && h->rest_first_cols > 0 && clo == clo_al && !g3.Ct && M1 > h->rest_first_cols) {
        // fused leaf: the chain waits for the deferred update of the NEXT leaf's next panel only (the first columns of this range) -- they go
        // out as a launch of their own with the event behind it, the other columns follow (under a far update this range's one launch took
        // 80 us instead of 15 and the chain stood still for it: kernel trace)
        const int n1 = h->rest_first_cols;
        GemmArgs ga = g3; ga.N = n1;
        gemm_dispatch(A_H16, E_SUB_F32, ga, st);
        rest_done(h, st); h->rest_recorded = true;
        GemmArgs gb = g3; gb.N = M1 - n1; gb.Bt = Yt + (long)n1 * Kr; gb.C = (float*)g3.C + n1; gb.col_lo = 0;
        gemm_dispatch(A_H16, E_SUB_F32, gb, st);
        h->rest_first_cols = 0;
    } else
    gemm_dispatch(A_H16, E_SUB_F32, g3, st);
    if (lane == 2) h->rest_first_cols = 0;
    if (record) {
        HIPQ(h, hipEventRecord(e3, st));
        h->far_flops.push_back(2.0 * M1 * (double)Kr * Kw);
        h->far_bytes.push_back((g3.Ct ? 10.0 : 8.0) * M1 * (double)Kw + 2.0 * Kr * ((double)M1 + Kw));
        h->far_dims.push_back(Kw); h->far_dims.push_back(M1); h->far_dims.push_back(Kr);
        h->far_flops_tn.push_back(2.0 * (double)g1.M * (double)Kr * (double)g1.K);
    }
}

int factor_node(mpqr_handle_t h, int id, bool do_panel);
int factor_rec(mpqr_handle_t h, int id, bool do_panel);
int robust_tall_leaf(mpqr_handle_t h, const Node nd, bool do_panel);

// A Gram-Householder leaf that finds its columns too ill conditioned raises its device flag AND a word in mapped host memory.  The
// thread that enqueues mpqr_factor's block loop reads that word before every leaf (a plain load, nothing on the device) and stops
// enqueuing: everything downstream of the flagged leaf is redone anyway, so the sooner the queue ends the less is wasted.
constexpr int MPQR_ABORT_PASS = -1000;                   // internal: factor_node's "stop enqueuing", never returned to callers
// any flag word of blocks [0, upto) raised?
static inline bool flag_words_set(mpqr_handle_t h, int upto) {
    if (!h->hflag_host) return false;
    for (int b = 0; b < upto && b < h->flag_words; b++)
        if (__atomic_load_n(h->hflag_host + b, __ATOMIC_RELAXED) != 0) return true;
    return false;
}
// Checked before every leaf: any raised word ends the pass (everything downstream of a flagged leaf is redone anyway).
static inline bool pass_is_flagged(mpqr_handle_t h) {
    return h->watch_flags && flag_words_set(h, h->flag_words - 2);      // (the last two words: the T stream's time-out word, the deflated-columns count)
}

// Robust path for a tall (<=128-column) leaf: factor it through a temporary sub-tree of 32-column leaves
// (column-by-column kernels + MFMA updates inside the leaf) and keep only the sub-tree's root T.  The sub-tree's T
// arena is part of the plan (rbTf/rbTh/rbTth): no allocation, no host synchronisation here.
int robust_tall_leaf(mpqr_handle_t h, const Node nd, bool do_panel) {
    Range rg("mpqr:robust_leaf");
    std::vector<Node> saved_nodes = h->nodes; std::vector<int> saved_tops = h->tops;
    float* oTf = h->Tf; half_t* oTh = h->Th; half_t* oTth = h->Tth;
    h->nodes.clear(); h->tops.clear();
    h->force32 = true;                                   // stays on while the sub-tree is factored
    const int root = build_tree(h, nd.c0, nd.c1);
    size_t toff = 0;
    for (Node& x : h->nodes) { x.toff = toff; toff += (size_t)x.ldt * x.ldt; }
    int rc = MPQR_OK;
    if (!h->rbTf || toff > h->rb_elems) rc = fail(h, MPQR_ERR_STATE, "robust-leaf T arena missing or too small");
    else {
        h->Tf = h->rbTf; h->Th = h->rbTh; h->Tth = h->rbTth;
        rc = factor_node(h, root, do_panel);
        const Node rt = h->nodes[root];
        const size_t el = (size_t)rt.ldt * rt.ldt;      // same aligned range as nd
        if (hipMemcpyAsync(oTf + nd.toff, h->rbTf + rt.toff, el * sizeof(float), hipMemcpyDeviceToDevice, h->s0) != hipSuccess ||
            hipMemcpyAsync(oTh + nd.toff, h->rbTh + rt.toff, el * sizeof(half_t), hipMemcpyDeviceToDevice, h->s0) != hipSuccess ||
            hipMemcpyAsync(oTth + nd.toff, h->rbTth + rt.toff, el * sizeof(half_t), hipMemcpyDeviceToDevice, h->s0) != hipSuccess)
            rc = fail(h, MPQR_ERR_HIP, "robust leaf: copying the root T failed");
    }
    h->force32 = false;
    h->Tf = oTf; h->Th = oTh; h->Tth = oTth;
    h->nodes = saved_nodes; h->tops = saved_tops;
    return rc;
}

// chain stream -> T stream: everything enqueued on s0 so far (reflectors of the nodes below) is visible to sT
static void t_stream_follows_chain(mpqr_handle_t h) {
    HIPQ(h, hipEventRecord(h->ev_v, h->s0));
    HIPQ(h, hipStreamWaitEvent(h->sT, h->ev_v, 0));
}

// T stream: "the deferred update the chain stream waits for is complete" -- an event, or (polling plans) the progress word tflag[3]
static inline bool rest_polls(mpqr_handle_t h) { return h->tpoll && h->tflag && h->hflag_dev && h->flag_words > 0 && h->fused_leaf; }
static void rest_done(mpqr_handle_t h, hipStream_t st) {
    if (rest_polls(h)) launch_publish_word(h->tflag, 3, ++h->rest_seq, st);
    else HIPQ(h, hipEventRecord(h->ev_rest, st));
}
// chain stream: wait for it (unless the last gh_solve already did, at its end).  A fused leaf's deferred update announces itself behind its FIRST
// 128 columns -- all the next fused leaf's chain work reads; the other columns are ordered by the T stream itself, where the next leaf's
// deferred update follows.  A leaf that is NOT fused (robust path, tail, ragged end, tree-scheduled block) updates the whole range on the
// chain stream: it waits for everything the T stream has been given so far (need_all; an event recorded now -- rare, so its cost does not matter).
static void rest_wait(mpqr_handle_t h, bool need_all) {
    if (!h->rest_pending) return;
    static const int dbg_norestwait = []() { const char* e = getenv("MPQR_DBG_NORESTWAIT"); return e ? atoi(e) : 0; }();   // timing experiment only (a race)
    if (need_all && h->rest_split) {
        HIPQ(h, hipEventRecord(h->ev_rest, h->sT));
        HIPQ(h, hipStreamWaitEvent(h->s0, h->ev_rest, 0));
    } else if (!dbg_norestwait) {
        if (!rest_polls(h)) HIPQ(h, hipStreamWaitEvent(h->s0, h->ev_rest, 0));
        else if (!h->rest_in_solve) launch_wait_flag(h->tflag, h->rest_seq, h->hflag_dev + h->flag_words - 1, h->tpoll_ticks, h->s0, 3);
    }
    h->rest_pending = false; h->rest_in_solve = false; h->rest_split = false;
}
// the Gram matrix of a Gram-Householder leaf: gh_gram + gh_reduce, or the reduction alone when the previous leaf's leaf_b has already left
// the partials (fused leaf)
static void leaf_gram(mpqr_handle_t h, const LeafArgs& a) {
    if (h->gram_ready_c0 == a.c0 && h->gram_ready_rows == a.mrows - a.c0 && h->gram_ready_n > 0)
        launch_gh_gram_reduce(h->Gp, h->gram_ready_n, h->Gs, h->s0);
    else
        launch_gh_gram(a, h->Gp, h->Gs, h->s0);
    h->gram_ready_c0 = -1;
}

int factor_rec(mpqr_handle_t h, int id, bool do_panel) {
    const Node nd = h->nodes[id];
    int rc;
    // compact-WY T's are built on their own stream (sT): nothing on the chain needs T before op2 of the next apply,
    // so the leaf's T (Gram reduction + triangular inverse) and the merges T_LR = -T_L (V_L^T V_R) T_R run beside
    // that apply's X = A2^T V.  fp32 twin: everything stays on the chain stream.
    const bool tq = h->tq_on && id < (int)h->ev_T.size();
    hipStream_t st = tq ? h->sT : h->s0;
    if (nd.left < 0) {
        if (do_panel && pass_is_flagged(h)) return MPQR_ABORT_PASS;
        const bool tail = is_tail(h, nd.c0);             // (plain Householder: never flagged, nothing to make robust)
        const bool tall = leaf_width(h, nd.c0) == 128 && !tail;
        const bool robust_leaf = h->robust || (id < (int)h->leaf_robust.size() && h->leaf_robust[id]);
        if (do_panel && robust_leaf && !h->force32 && tall) {
            rc = robust_tall_leaf(h, nd, do_panel);
            if (tq) { t_stream_follows_chain(h); HIPQ(h, hipEventRecord(h->ev_T[id], h->sT)); }
            return rc;
        }
        bool have_s = false;
        LeafArgs a{};
        if (do_panel) {
            Range rg("mpqr:panel");
            a.A = h->Aeff; a.lda = h->lda; a.mrows = h->m; a.cb = rdown(nd.c0, tall || tail ? 128 : 32); a.c0 = nd.c0; a.c1 = nd.c1;
            a.Vh = h->Vh; a.ldvh = h->ldvh; a.Vt = h->Vt; a.ldvt = h->ldvt; a.vdiag = h->vdiag;
            a.P = h->P; a.maxwg = h->maxwg; a.hostflag = h->hflag_dev + h->cur_block; a.deflword = h->hflag_dev + h->flag_words - 2;
            const bool fused = tall && !h->Vf;    // fp16 mode: the Gram of the rounded reflectors comes out of gh_apply
            int* flag = h->dflag + (id < h->nflag ? id : 0);
            if (tall) {
                leaf_gram(h, a);
                launch_gh_solve(a, h->Gs, h->Cv, flag, h->s0);
                launch_gh_apply(a, h->Cv, fused ? h->Sp : nullptr, h->s0);
                h->n_gh_leaves++;
            }
            else if (tail) launch_leaf_tail(a, h->Sleaf, h->s0);
            else launch_leaf_factor(a, h->s0);
            if (h->Vf) launch_extract_vf(h->Aeff, h->lda, h->vdiag, h->Vf, h->n_pad, h->m, nd.c0, nd.c1, h->s0);
            have_s = fused;
            if (h->wait_after_first_leaf) {          // look-ahead: the block's other columns arrive with this event
                HIPQ(h, hipStreamWaitEvent(h->s0, h->wait_after_first_leaf, 0));
                h->wait_after_first_leaf = nullptr;
            }
        }
        Range rg("mpqr:wy_T");
        if (tq) t_stream_follows_chain(h);
        if (do_panel && tail && !h->Vf) {                  // the tail kernel left S itself (window coordinates, like the reduced partials below)
            const int sh = nd.a0 - a.cb;
            launch_t_leaf(h->Sleaf + (long)sh * 128 + sh, 1, 0, 128, nd.a0, nd.c0, nd.c1, h->Tf + nd.toff, h->Th + nd.toff,
                          h->Tth + nd.toff, nd.ldt, st);
        } else if (have_s) {
            // S = sum of gh_apply's partial Grams, in window coordinates (128 x 128 from a.cb); the node's aligned
            // range starts at a0 >= cb
            launch_gh_reduce_f32(h->Sp, gh_num_partials(a), h->S, st);
            const int sh = nd.a0 - a.cb;
            launch_t_leaf(h->S + (long)sh * 128 + sh, 1, 0, 128, nd.a0, nd.c0, nd.c1, h->Tf + nd.toff, h->Th + nd.toff,
                          h->Tth + nd.toff, nd.ldt, st);
        } else {
            int nslab; long slab;
            gram(h, nd, nd, &nslab, &slab, st);
            launch_t_leaf(h->S, nslab, slab, nd.ldt, nd.a0, nd.c0, nd.c1, h->Tf + nd.toff, h->Th + nd.toff, h->Tth + nd.toff,
                          nd.ldt, st);
        }
        if (tq) HIPQ(h, hipEventRecord(h->ev_T[id], h->sT));
        return MPQR_OK;
    }
    const Node L = h->nodes[nd.left], R = h->nodes[nd.right];
    if ((rc = factor_rec(h, nd.left, do_panel))) return rc;
    if (do_panel) {
        Range rg("mpqr:in_block_update");
        apply_node(h, L, h->Aeff, h->lda, R.c0, R.c1, true, h->a_scale, false);
    }
    if ((rc = factor_rec(h, nd.right, do_panel))) return rc;
    // T_LR = -T_L (V_L^T V_R) T_R
    Range rg("mpqr:wy_T_merge");
    if (tq) t_stream_follows_chain(h);        // V_R is complete on the chain; T_L, T_R are earlier work of sT itself
    int nslab; long slab;
    gram(h, L, R, &nslab, &slab, st);
    if (L.ldt <= 128 && R.ldt <= 128) {
        launch_t_merge(h->S, L.ldt, R.ldt, h->Tf + L.toff, h->Tf + R.toff, h->tmp2, st);
    } else {
        SgemmArgs s1{};
        s1.A = h->S; s1.lda = R.ldt; s1.transA = 0; s1.nslab_a = nslab; s1.slab_a = slab;
        s1.B = h->Tf + R.toff; s1.ldb = R.ldt; s1.transB = 0;
        s1.C = h->tmp1; s1.ldc = R.ldt; s1.M = L.ldt; s1.N = R.ldt; s1.K = R.ldt; s1.alpha = 1.f; s1.beta = 0.f;
        s1.upperB = 1;
        launch_sgemm(s1, st);
        SgemmArgs s2{};
        s2.A = h->Tf + L.toff; s2.lda = L.ldt; s2.transA = 0; s2.nslab_a = 1;
        s2.B = h->tmp1; s2.ldb = R.ldt; s2.transB = 0;
        s2.C = h->tmp2; s2.ldc = R.ldt; s2.M = L.ldt; s2.N = R.ldt; s2.K = L.ldt; s2.alpha = -1.f; s2.beta = 0.f;
        s2.upperA = 1;
        launch_sgemm(s2, st);
    }
    launch_t_assemble(h->Tf + nd.toff, h->Th + nd.toff, h->Tth + nd.toff, nd.ldt, nd.a0, h->Tf + L.toff, L.ldt, L.a0,
                      nd.c0, L.c1, h->Tf + R.toff, R.ldt, R.a0, nd.c1, h->tmp2, R.ldt, st);
    if (tq) HIPQ(h, hipEventRecord(h->ev_T[id], h->sT));
    return MPQR_OK;
}

// ---- right-looking factorisation of one top-level block with a block-level T ("flat" mode)
// Inside a top-level block the tree schedule above puts merged T's on the critical path: after the last leaf of a
// sub-tree the chain waits for leaf T -> merge -> merge (-> merge) before the next apply can run its second GEMM (kernel
// trace: 100-350 us per sub-tree end, ~15 of the 37 ms the chain takes at 16384^2).  Here every leaf j is applied to the
// block's remaining columns on its own (K = 128, needs only the leaf's T_j, which is built beside the apply's first
// GEMM), and the block's T is completed in the background, one column block per leaf (LAPACK larft order):
//     T[0:o, o:o+w] = -T[0:o, 0:o] (V_P^T V_j) T_j,    P = the block's earlier reflectors,  o = their count,
// which only the far update and Q formation need.  Same flops as the tree (every column still receives every
// reflector of the block once), K = 128 instead of 128 / 256 / 512 for the in-block updates.
static void collect_leaves(mpqr_handle_t h, int id, std::vector<int>& out) {
    const Node& nd = h->nodes[id];
    if (nd.left < 0) { out.push_back(id); return; }
    collect_leaves(h, nd.left, out); collect_leaves(h, nd.right, out);
}
static bool flat_block_ok(mpqr_handle_t h, int top, std::vector<int>& leaves) {
    if (h->Vf || h->force32) return false;
    leaves.clear();
    collect_leaves(h, top, leaves);
    if (leaves.size() < 2) return false;
    const Node& tp = h->nodes[top];
    if (tp.a0 != tp.c0) return false;
    for (int id : leaves) {
        const Node& lf = h->nodes[id];
        if (leaf_width(h, lf.c0) != 128 || lf.a0 != lf.c0 || (lf.c0 % 64) != 0) return false;
    }
    return true;
}
int factor_block_flat(mpqr_handle_t h, int top, const std::vector<int>& leaves) {
    const Node tp = h->nodes[top];
    const int ld = tp.ldt;
    const bool tq = h->tq_on && top < (int)h->ev_T.size();
    hipStream_t st = tq ? h->sT : h->s0;
    int rc;
    // column block of the block's T for leaf `lf` (o = reflectors of the block before it): on the side stream
    auto t_column_block = [&](const Node& lf, int o) {
        if (o <= 0) return;
        Range rg("mpqr:wy_T_merge");
        Node P = tp; P.c1 = lf.c0; P.a1 = lf.c0; P.ldt = o;                     // the earlier reflectors of the block
        int nslab; long slab;
        gram(h, P, lf, &nslab, &slab, st);                                         // S = V_P^T V_j  (o x ldt_j, ld = ldt_j)
        SgemmArgs s1{};                                                            // tmp1 = S T_j
        s1.A = h->S; s1.lda = lf.ldt; s1.transA = 0; s1.nslab_a = nslab; s1.slab_a = slab;
        s1.B = h->Tf + lf.toff; s1.ldb = ld; s1.transB = 0;
        s1.C = h->tmp1; s1.ldc = lf.ldt; s1.M = o; s1.N = lf.ldt; s1.K = lf.ldt; s1.alpha = 1.f; s1.beta = 0.f; s1.upperB = 1;
        launch_sgemm(s1, st);
        SgemmArgs s2{};                                                            // T[0:o, o:o+w] = -T_P tmp1
        s2.A = h->Tf + tp.toff; s2.lda = ld; s2.transA = 0; s2.nslab_a = 1;
        s2.B = h->tmp1; s2.ldb = lf.ldt; s2.transB = 0;
        s2.C = h->Tf + tp.toff + o; s2.ldc = ld; s2.M = o; s2.N = lf.ldt; s2.K = o; s2.alpha = -1.f; s2.beta = 0.f; s2.upperA = 1;
        // K = o (up to 896) in one workgroup per 64 x 64 tile is a 70 us latency chain of 56 load / barrier / MFMA steps, and the
        // last leaf's column block sits on the path to the block's far update: K is cut into ranges of 128, the partial products
        // (S is free again: tmp1 = S T_j has been formed) are summed by t_colblock_h16 on its way to the fp16 copies
        const int nz = std::min(8, o / 128);
        if (nz > 1 && (size_t)nz * o * lf.ldt <= h->s_elems) {
            s2.C = h->S; s2.ldc = lf.ldt; s2.ksplit = nz; s2.slab_c = (long)o * lf.ldt;
            launch_sgemm(s2, st);
            launch_t_colblock_h16(h->Tf + tp.toff, h->Th + tp.toff, h->Tth + tp.toff, ld, o, o, lf.ldt, st, h->S, nz, s2.slab_c, lf.ldt);
        } else {
            launch_sgemm(s2, st);
            launch_t_colblock_h16(h->Tf + tp.toff, h->Th + tp.toff, h->Tth + tp.toff, ld, o, o, lf.ldt, st);
        }
    };
    Node prev{}; int prev_o = -1;                         // leaf whose column block of T is still to be built
    for (size_t j = 0; j < leaves.size(); j++) {
        if (pass_is_flagged(h)) return MPQR_ABORT_PASS;
        const int id = leaves[j];
        Node lf = h->nodes[id];
        const int o = lf.c0 - tp.c0;                       // reflectors of the block before this leaf
        lf.toff = tp.toff + (size_t)o * (ld + 1);          // the leaf's T = diagonal block of the block's T
        lf.tld = ld;
        const bool tail = is_tail(h, lf.c0);
        const bool robust_leaf = !tail && (h->robust || (id < (int)h->leaf_robust.size() && h->leaf_robust[id]));
        bool mid_leaf = false, gh_leaf = false, far_wait_pending = false;
        int gh_partials = 0, gh_sh = 0;
        LeafArgs gh_args{};
        mpqr_handle_s::MidT mid_desc{};
        if (tail) {
            // the matrix's last <= 128 rows: plain Householder in one workgroup, S for its T from the same kernel
            Range rg("mpqr:panel");
            if (h->far_hook) { auto fh = std::move(h->far_hook); h->far_hook = nullptr; if ((rc = fh())) return rc; }
            LeafArgs a{};
            a.A = h->Aeff; a.lda = h->lda; a.mrows = h->m; a.cb = rdown(lf.c0, 128); a.c0 = lf.c0; a.c1 = lf.c1;
            a.Vh = h->Vh; a.ldvh = h->ldvh; a.Vt = h->Vt; a.ldvt = h->ldvt; a.vdiag = h->vdiag;
            a.P = h->P; a.maxwg = h->maxwg; a.hostflag = nullptr;
            if (h->wait_after_first_leaf && (!tq || (int)j >= std::min<int>(h->pre_leaves, (int)leaves.size()))) {
                HIPQ(h, hipStreamWaitEvent(h->s0, h->wait_after_first_leaf, 0));      // its columns still wait for the previous block's far update
                h->wait_after_first_leaf = nullptr;
            }
            launch_leaf_tail(a, h->Sleaf, h->s0);
            Range rt("mpqr:wy_T");
            if (tq) t_stream_follows_chain(h);
            const int sh = lf.a0 - a.cb;
            launch_t_leaf(h->Sleaf + (long)sh * 128 + sh, 1, 0, 128, lf.a0, lf.c0, lf.c1, h->Tf + lf.toff, h->Th + lf.toff,
                          h->Tth + lf.toff, lf.ldt, h->s0, ld);
        } else if (robust_leaf) {
            if (h->far_hook) { auto fh = std::move(h->far_hook); h->far_hook = nullptr; if ((rc = fh())) return rc; }   // (see below)
            // column-by-column kernels through a private sub-tree; its root T (contiguous, ldt^2) goes into the diagonal block
            const size_t keep = h->nodes[id].toff;
            if ((rc = robust_tall_leaf(h, h->nodes[id], true))) return rc;
            HIPQ(h, hipMemcpy2DAsync(h->Tf + lf.toff, (size_t)ld * 4, h->Tf + keep, (size_t)lf.ldt * 4, (size_t)lf.ldt * 4, lf.ldt, hipMemcpyDeviceToDevice, h->s0));
            HIPQ(h, hipMemcpy2DAsync(h->Th + lf.toff, (size_t)ld * 2, h->Th + keep, (size_t)lf.ldt * 2, (size_t)lf.ldt * 2, lf.ldt, hipMemcpyDeviceToDevice, h->s0));
            HIPQ(h, hipMemcpy2DAsync(h->Tth + lf.toff, (size_t)ld * 2, h->Tth + keep, (size_t)lf.ldt * 2, (size_t)lf.ldt * 2, lf.ldt, hipMemcpyDeviceToDevice, h->s0));
            if (tq) t_stream_follows_chain(h);
        } else {
            Range rg("mpqr:panel");
            LeafArgs a{};
            a.A = h->Aeff; a.lda = h->lda; a.mrows = h->m; a.cb = rdown(lf.c0, 128); a.c0 = lf.c0; a.c1 = lf.c1;
            a.Vh = h->Vh; a.ldvh = h->ldvh; a.Vt = h->Vt; a.ldvt = h->ldvt; a.vdiag = h->vdiag;
            a.P = h->P; a.maxwg = h->maxwg; a.hostflag = h->hflag_dev + h->cur_block; a.deflword = h->hflag_dev + h->flag_words - 2;
            leaf_gram(h, a);
            // the previous block's far update is enqueued HERE (run_block_loop): its stream starts when this leaf's gh_solve does,
            // so its first GEMMs fill the other 255 CUs during the solve instead of holding them while gh_gram wants them
            if (h->far_hook) { auto fh = std::move(h->far_hook); h->far_hook = nullptr; if ((rc = fh())) return rc; }
            unsigned long long* stamp = (h->dbg_stamps && 2 * (lf.c0 / 128) + 1 < h->dbg_stamps_n) ? h->dbg_stamps + 2 * (lf.c0 / 128) : nullptr;
            if (h->rest_pending && rest_polls(h)) {            // this solve ends by polling the T stream's word: nothing to wait for behind it
                const SolveWait ws{h->tflag, 3, h->rest_seq, h->hflag_dev + h->flag_words - 1, h->tpoll_ticks, stamp};
                launch_gh_solve(a, h->Gs, h->Cv, h->dflag + (id < h->nflag ? id : 0), h->s0, &ws);
                h->rest_in_solve = true;
            } else if (stamp) {
                const SolveWait ws{nullptr, 0, 0, nullptr, 0, stamp};
                launch_gh_solve(a, h->Gs, h->Cv, h->dflag + (id < h->nflag ? id : 0), h->s0, &ws);
            } else
            launch_gh_solve(a, h->Gs, h->Cv, h->dflag + (id < h->nflag ? id : 0), h->s0);
            gh_args = a;                                     // (gh_apply, or the fused leaf's leaf_a, is launched below once the leaf's form is known)
            h->n_gh_leaves++;
            if (h->wait_after_first_leaf && !tq) {       // look-ahead: the block's other columns arrive with this event
                HIPQ(h, hipStreamWaitEvent(h->s0, h->wait_after_first_leaf, 0));
                h->wait_after_first_leaf = nullptr;
            }
            gh_leaf = true; gh_partials = gh_num_partials(a); gh_sh = lf.a0 - a.cb;
            far_wait_pending = h->wait_after_first_leaf != nullptr;
        }
        // the chain: this leaf alone onto the rest of the block -- and, under look-ahead, onto the next block's first
        // leaves as well (ext_c1): those leaves then need nothing from this block's far update and the chain crosses the block
        // boundary without waiting for the block's T and a skinny far update (~350 us per boundary at 16384^2)
        const int upd_end = std::max(tp.c1, h->ext_c1);
        // The first P leaves of this block are already up to date (the previous block's in-block updates reached them); the
        // block's other columns become valid with the event wait_after_first_leaf (part (a) of the previous block's far update).
        // Leaves before the last pre-updated one update only the other pre-updated leaves on the chain; their update of the rest is
        // DEFERRED: it runs on the T stream (apply_node lane 2: X, Y and the update there, own scratch) as soon as the event has
        // fired, beside the chain's work on the next leaf.  The last pre-updated leaf's own update follows them in that stream's
        // order (its X GEMM runs there), so the chain first meets the far update's columns one leaf later than the block boundary.
        const int P = std::min<int>(h->pre_leaves, (int)leaves.size());
        const int cpre = P > 0 ? h->nodes[leaves[P - 1]].c1 : lf.c1;
        // Leaf-level look-ahead (MPQR_LEAF_LA; round 3 measured it slower with its two events per leaf): on the chain a leaf updates only the NEXT
        // leaf's columns (what the next gh_gram reads); its update of everything else (`rest`) runs on the T stream (lane 2) as soon
        // as T_j exists, i.e. beside the next leaf's gh_solve.  Order: the T stream runs X_urgent(j), rest(j), X_urgent(j+1), ...
        // so rest(j) has written the columns X_urgent(j+1) reads, and the chain's own update of them waits for that X (ev_x).
        // Round 4: with the one-launch middle and the polling T stream no event is left on the chain for it but one wait for the previous
        // leaf's rest (long finished): on for leaves with >= 20480 rows, where the update it moves off the chain is large enough --
        // 24576 x 8192: 29.4 -> 28.7 ms, 32768 x 4096: 18.8 -> 18.3, 49152 x 4096: 32.2 -> 30.8, 65536 x 8192: factorisation 44.2 -> 41.8;
        // 16384^2: 34.8 either way, 2048^2: 2.95 -> 3.01 (MPQR_LEAF_LA=0 / 1 forces it off / on for every leaf; same bits either way).
        static const int leaf_la_env = []() { const char* e = getenv("MPQR_LEAF_LA"); return e ? atoi(e) : -1; }();
        const int leaf_la = leaf_la_env >= 0 ? leaf_la_env : (h->m - lf.c0 >= 20480 ? 1 : 0);
        const int next_c1 = j + 1 < leaves.size() ? h->nodes[leaves[j + 1]].c1 : lf.c1 + 128;
        const bool lane2_ok = h->Xt2 != nullptr && h->Yt2 != nullptr;      // (allocated with opts.lookahead only: ADVICE round 3)
        // Fused leaf (round 5): the chain works on the NEXT 128 columns only, in three launches (leaf_a: gh_apply + partial X; leaf_m: T and Y;
        // leaf_b: the update + the next leaf's partial Gram matrices), the rest of the block follows on the T stream as under leaf-level look-ahead.
        // Needs a full 128-column leaf, 128 further columns inside the update range, and the polling T stream (leaf_b publishes the word).
        // (not for leaves of >= 49152 rows, where a workgroup of leaf_a / leaf_b takes four row blocks: 65536 x 8192 factors in 38.8 ms on the
        //  round-4 launches with leaf-level look-ahead, 39.3 fused)
        const bool fl = h->fused_leaf && gh_leaf && tq && lane2_ok && h->Xp && h->tpoll && h->tflag && h->hflag_dev && h->flag_words > 0 && h->m - lf.c1 < 49152 &&
                        lf.ldt == 128 && lf.a0 == lf.c0 && lf.c1 - lf.c0 == 128 && (lf.c0 % 128) == 0 && lf.c1 + 128 <= upd_end && lf.c1 + 128 <= h->n &&
                        !h->shadow && !h->Vf && h->opts.precision != MPQR_PREC_FP32;
        const bool la_split = fl || (tq && lane2_ok && leaf_la && lf.c1 < upd_end && std::min(next_c1, upd_end) < upd_end);
        const bool pre_split = tq && lane2_ok && P >= 2 && (int)j < P - 1 && cpre < upd_end;      // a pre-updated leaf before the last one
        if ((int)j == P - 1 || P == 0 || (int)j >= P || !tq) {
            if (h->wait_after_first_leaf) {
                // Only X = C2^T V_j reads those columns first, and it runs on the side stream: that stream waits, the chain stream goes
                // on with T_j and meets the dependency through the X event (the robust leaf path, which works on the chain stream, waits there)
                HIPQ(h, hipStreamWaitEvent(tq && !robust_leaf ? h->sT : h->s0, h->wait_after_first_leaf, 0));
                if (tq && (robust_leaf || fl)) HIPQ(h, hipStreamWaitEvent(robust_leaf ? h->sT : h->s0, h->wait_after_first_leaf, 0));   // (fused leaf: leaf_a reads those columns on the chain stream)
                h->wait_after_first_leaf = nullptr;
            }
        }
        const int own_end = fl ? lf.c1 + 128 : la_split ? std::min(next_c1, upd_end) : (pre_split ? cpre : upd_end);
        if (gh_leaf && !fl) launch_gh_apply(gh_args, h->Cv, h->Sp, h->s0);
        const bool have_rest = (la_split || pre_split) && own_end < upd_end;
        if (gh_leaf) {
            // T_j = (Gram of the fp16 reflectors: gh_apply's partials, summed)^-1.  Round 4: when this leaf has an in-block update of the plain
            // kind, its X = C2^T V, the Gram sum and T_j go out as ONE launch on the chain stream (leaf_mid_kernel, from apply_node below): no
            // side stream for X, no hand-offs around it; the T stream then follows the chain AFTER the update (it only builds the previous
            // leaf's column block of T).  Not for a leaf whose X still waits for the previous block's far update: there the chain goes on
            // with T_j while the side stream waits.
            mid_leaf = !fl && h->leaf_mid && tq && (!la_split || h->tpoll) && !far_wait_pending && lf.ldt == 128 && lf.c1 < own_end &&
                       own_end - rdown(lf.c1, 32) <= 4096 && !h->shadow && !h->Vf && h->opts.precision != MPQR_PREC_FP32;
            if (fl) {
                // (T_j comes out of leaf_m below)
            } else if (mid_leaf) {
                mid_desc = mpqr_handle_s::MidT{h->Sp, gh_partials, gh_sh, lf.a0, lf.c0, lf.c1, h->Tf + lf.toff, h->Th + lf.toff, h->Tth + lf.toff, lf.ldt, ld};
            } else {
                Range rt("mpqr:wy_T");
                // ONE chain -> side-stream hand-off per leaf: it publishes the leaf's reflectors (for the apply's first GEMM and for this
                // leaf's column block of T) and, being later in the chain stream than the previous leaf's T, that T as well
                if (tq) t_stream_follows_chain(h);
                // T_j (Gram reduction + triangular inverse, ~22 us) stays on the chain stream, the apply's X = C2^T V goes to the side stream
                launch_gh_reduce_f32(h->Sp, gh_partials, h->Sleaf, h->s0);
                launch_t_leaf(h->Sleaf + (long)gh_sh * 128 + gh_sh, 1, 0, 128, lf.a0, lf.c0, lf.c1, h->Tf + lf.toff, h->Th + lf.toff,
                              h->Tth + lf.toff, lf.ldt, h->s0, ld);
            }
        }
        // leaf-level look-ahead with the one-launch middle: T_j comes out of the urgent apply below, and the rest (T stream) starts behind
        // the word its leaf_xt publishes -- no event on the chain stream for it
        const bool la_mid = la_split && (mid_leaf || fl);
        if (have_rest && la_split && !la_mid) HIPQ(h, hipEventRecord(h->ev_def, h->s0));   // T_j and V_j are complete here: the rest may start beside the urgent part
        // the previous leaf's deferred update (T stream) wrote columns this leaf's X reads and the next gh_gram needs: it ran beside this
        // leaf's gh_solve, which polled its progress word at its end (or: an event wait / a one-wave polling kernel here)
        rest_wait(h, !fl);
        if (fl) {
            Range rg("mpqr:fused_leaf");
            // next leaf: does it take the partial Gram matrices leaf_b can leave?  (a Gram-Householder leaf: not the tail, not on the robust path)
            bool next_gh = !is_tail(h, lf.c1) && !h->robust;
            if (next_gh && j + 1 < leaves.size()) { const int nid = leaves[j + 1]; next_gh = !(nid < (int)h->leaf_robust.size() && h->leaf_robust[nid]); }
            launch_leaf_a(gh_args, h->Cv, h->Sp, h->Xp, lf.c1, h->a_scale, h->s0);
            static const int dbg_nopub_fl = []() { const char* e = getenv("MPQR_DBG_NOPUB"); return e ? atoi(e) : 0; }();   // test hook: the words are never published -> the waiter times out
            ++h->tseq;
            // two progress words for the T stream: leaf_m (behind leaf_a in this stream) publishes "the leaf's reflectors are complete" -- the
            // T stream's X = C_rest^T V_j starts then, beside leaf_m --, leaf_b publishes "T_j is complete" for the Y = X T_j behind it
            launch_leaf_m(h->Sp, h->Xp, gh_partials, h->Sleaf, h->Xs, h->mid_counter, gh_sh, lf.a0, lf.c0, lf.c1, h->Tf + lf.toff, h->Th + lf.toff,
                          h->Tth + lf.toff, lf.ldt, ld, h->Yfl, h->s0, h->tflag + 2, dbg_nopub_fl ? -1 : h->tseq);
            launch_leaf_b(gh_args, lf.c1, h->Yfl, 1.0f / h->a_scale, h->Gp, next_gh, h->tflag, dbg_nopub_fl ? -1 : h->tseq, h->s0);
            if (next_gh) { h->gram_ready_c0 = lf.c1; h->gram_ready_rows = h->m - lf.c1; h->gram_ready_n = fl_gram_partials(gh_args); }
            h->n_fused_leaves++;
            // the T stream goes on with this leaf's update of the rest of the block: X once the chain is past V_j, Y = X T_j once it is past T_j
            // (apply_node, lane 2, launches that second wait in front of its op2)
            launch_wait_flag(h->tflag, h->tseq, h->hflag_dev + h->flag_words - 1, h->tpoll_ticks, h->sT, 2);
            h->lane2_twait = h->tseq;
            if (!have_rest) { launch_wait_flag(h->tflag, h->tseq, h->hflag_dev + h->flag_words - 1, h->tpoll_ticks, h->sT, 0); h->lane2_twait = 0; }
        } else
        if (lf.c1 < own_end) {
            Range rg("mpqr:in_block_update");
            h->op1_stream = (tq && !mid_leaf) ? h->sT : nullptr;
            h->mid = mid_leaf ? &mid_desc : nullptr;
            const bool poll = mid_leaf && h->tpoll && h->tflag && h->hflag_dev && h->flag_words > 0;
            h->xt_pub = poll ? ++h->tseq : 0;                  // leaf_xt (behind the leaf's T in the chain stream) publishes this value
            static const int dbg_nopub = []() { const char* e = getenv("MPQR_DBG_NOPUB"); return e ? atoi(e) : 0; }();   // test hook: the word is never
            const bool swallow = poll && dbg_nopub;                                                                      // published -> the waiter must time out
            if (swallow) h->xt_pub = -1;                       // (leaf_xt then stores -1: never >= a sequence number)
            apply_node(h, lf, h->Aeff, h->lda, lf.c1, own_end, true, h->a_scale, false);
            h->op1_stream = nullptr;
            if (h->mid || h->xt_pub) { h->mid = nullptr; h->xt_pub = 0; h->dispatch_error = true; }   // (apply_node took another path than predicted)
            // the T stream may go on (with the previous leaf's column block of T) once the chain is past this leaf's T: it polls the
            // published word instead of waiting for an event the chain stream would have to record
            if (poll) launch_wait_flag(h->tflag, h->tseq, h->hflag_dev + h->flag_words - 1, h->tpoll_ticks, h->sT);
            else if (mid_leaf && tq) t_stream_follows_chain(h);
        }
        if (have_rest) {                                   // the rest, on the T stream
            Range rg("mpqr:in_block_update_deferred");
            if (!la_mid) {
                if (!la_split) HIPQ(h, hipEventRecord(h->ev_def, h->s0));
                HIPQ(h, hipStreamWaitEvent(h->sT, h->ev_def, 0));
            }                                               // (la_mid: the T stream is already behind wait_flag_kernel for this leaf)
            int lo = own_end;
            h->rest_recorded = false;
            h->rest_first_cols = fl ? 128 : 0;             // (the next leaf's leaf_a reads exactly these columns)
            if (pre_split && lo < cpre) {                   // columns the previous block already brought up to date: no far update to wait for
                apply_node(h, lf, h->Aeff, h->lda, lo, cpre, true, h->a_scale, false, 2);
                lo = cpre;
            }
            if (pre_split && h->wait_after_first_leaf) HIPQ(h, hipStreamWaitEvent(h->sT, h->wait_after_first_leaf, 0));   // (kept: the last pre-updated leaf waits too)
            apply_node(h, lf, h->Aeff, h->lda, lo, upd_end, true, h->a_scale, false, 2);
            // EVERY deferred update is announced to the chain stream: the next leaf may take the one-launch middle, whose X runs on the
            // chain stream and reads the columns this update writes (a leaf on the two-stream form is ordered behind it by the T stream
            // itself; the extra wait is harmless there).  Without it the first one-launch leaf behind a block's two-stream leaves raced
            // with the second leaf's rest: R differed from run to run from that leaf's successor on (tools/determinism_check.py).
            if (!h->rest_recorded) rest_done(h, h->sT);
            h->rest_pending = true; h->rest_in_solve = false; h->rest_first_cols = 0;
            h->rest_split = h->rest_recorded;               // announced behind its first columns only
        }
        // background, behind this leaf's X GEMM in the side stream's queue: the PREVIOUS leaf's column block of T
        if (prev_o >= 0) t_column_block(prev, prev_o);
        prev = lf; prev_o = o;
        if (!tq) { t_column_block(prev, prev_o); prev_o = -1; }          // single stream: nothing to defer
    }
    if (!h->next_block_flat) rest_wait(h, true);           // (a flat successor's first leaf takes the wait over: its gh_solve polls at its end, or the leaf waits as a non-fused one)
    if (prev_o >= 0) {                                     // the last leaf's column block needs its T (chain stream)
        if (tq) t_stream_follows_chain(h);
        t_column_block(prev, prev_o);
    }
    if (tq) {
        // the block's T first: the far update waits for it (every event operation costs its stream ~6 us); the leaves' T's
        // are diagonal blocks of it
        HIPQ(h, hipEventRecord(h->ev_T[top], h->sT));
        if (!h->defer_join) for (int id : leaves) HIPQ(h, hipEventRecord(h->ev_T[id], h->sT));
    }
    return MPQR_OK;
}

// factor the sub-tree `id`; on return every T below it is ordered before whatever is enqueued on the chain stream next
int factor_node(mpqr_handle_t h, int id, bool do_panel) {
    if (h->tq_on) {
        while (h->ev_T.size() < h->nodes.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(h, MPQR_ERR_HIP, "hipEventCreate failed");
            h->ev_T.push_back(e);
        }
    }
    std::vector<int> leaves;
    const bool is_top = std::find(h->tops.begin(), h->tops.end(), id) != h->tops.end();
    h->node_done_stream = h->s0;
    const bool flat = do_panel && is_top && flat_block_ok(h, id, leaves);
    const int rc = flat ? factor_block_flat(h, id, leaves) : factor_rec(h, id, do_panel);
    if (h->tq_on && !(h->defer_join && !h->force32)) {
        HIPQ(h, hipEventRecord(h->ev_join, h->sT));
        HIPQ(h, hipStreamWaitEvent(h->s0, h->ev_join, 0));
    }
    return rc;
}

// zero the reflector stores from reflector c0 on (c0 = 0: everything; a restart keeps the reflectors of the blocks left of c0)
int clear_reflectors(mpqr_handle_t h, int c0 = 0) {
    if (c0 <= 0) {
        HIPCHK(h, hipMemsetAsync(h->Vh, 0, (size_t)(h->m_pad + 256) * h->ldvh * sizeof(half_t), h->s0));
        HIPCHK(h, hipMemsetAsync(h->Vt, 0, (size_t)(h->n_pad + 256) * h->ldvt * sizeof(half_t), h->s0));
        HIPCHK(h, hipMemsetAsync(h->vdiag, 0, (size_t)h->n_pad * sizeof(float), h->s0));
        if (h->Vf) HIPCHK(h, hipMemsetAsync(h->Vf, 0, (size_t)(h->m_pad + 256) * h->n_pad * sizeof(float), h->s0));
        return MPQR_OK;
    }
    HIPCHK(h, hipMemset2DAsync(h->Vh + c0, (size_t)h->ldvh * sizeof(half_t), 0, (size_t)(h->ldvh - c0) * sizeof(half_t), (size_t)h->m_pad + 256, h->s0));
    HIPCHK(h, hipMemsetAsync(h->Vt + (size_t)c0 * h->ldvt, 0, (size_t)(h->n_pad + 256 - c0) * h->ldvt * sizeof(half_t), h->s0));
    HIPCHK(h, hipMemsetAsync(h->vdiag + c0, 0, (size_t)(h->n_pad - c0) * sizeof(float), h->s0));
    if (h->Vf) HIPCHK(h, hipMemset2DAsync(h->Vf + c0, (size_t)h->n_pad * sizeof(float), 0, (size_t)(h->n_pad - c0) * sizeof(float), (size_t)h->m_pad + 256, h->s0));
    return MPQR_OK;
}

// power-of-two scale so that fp16 operands stay in range (and out of the subnormal range): column norms are
// bounded by sqrt(m) max|a|, which the scale brings to [2^7, 2^8)
// copy_to != nullptr: the same pass also copies the whole padded buffer there (the factorisation's copy-in)
int compute_scale(mpqr_handle_t h, const float* src, float* copy_to = nullptr) {
    if (copy_to) launch_copy_absmax(src, copy_to, h->lda, h->m_pad, h->m, h->n, h->dscalar, h->s0);
    else launch_absmax(src, h->lda, h->m, h->n, h->dscalar, h->s0);
    float mx = 0.f;
    HIPCHK(h, hipMemcpyAsync(&mx, h->dscalar, sizeof(float), hipMemcpyDeviceToHost, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    float s = 1.f;
    if (mx > 0.f && std::isfinite(mx)) {
        const float nrm = mx * sqrtf((float)h->m);
        int e; frexpf(nrm, &e);                  // nrm = f * 2^e, f in [0.5,1)
        s = ldexpf(1.f, 8 - e);                  // exact power of two: scaling itself adds no rounding
    }
    h->a_scale = s;
    return MPQR_OK;
}

// T of a pair of top-level blocks from its children's: T_LR = -T_L (V_L^T V_R) T_R  (enqueued on `st`, own scratch)
static void pair_w(mpqr_handle_t h, int pid, hipStream_t st);
static void merge_pair(mpqr_handle_t h, int pid, hipStream_t st) {
    Range rg("mpqr:wy_T_pair");
    const Node nd = h->nodes[pid];
    const Node L = h->nodes[nd.left], R = h->nodes[nd.right];
    if (h->tq_on) {                                       // the children's T's come from the T stream (blocks factored here)
        if (L.id < (int)h->ev_T.size()) HIPQ(h, hipStreamWaitEvent(st, h->ev_T[L.id], 0));
        if (R.id < (int)h->ev_T.size()) HIPQ(h, hipStreamWaitEvent(st, h->ev_T[R.id], 0));
    }
    int nslab; long slab;
    gram(h, L, R, &nslab, &slab, st, h->S2, h->s2_elems);
    SgemmArgs s1{};
    s1.A = h->S2; s1.lda = R.ldt; s1.transA = 0; s1.nslab_a = nslab; s1.slab_a = slab;
    s1.B = h->Tf + R.toff; s1.ldb = R.ldt; s1.transB = 0;
    s1.C = h->tmp1b; s1.ldc = R.ldt; s1.M = L.ldt; s1.N = R.ldt; s1.K = R.ldt; s1.alpha = 1.f; s1.beta = 0.f; s1.upperB = 1;
    // both products are 1024^3 on 256 workgroups whose K loops (64 load / barrier / MFMA steps) take ~85 us each: K is cut into
    // ranges of 256, the partial slabs are summed by the next consumer's operand staging / by t_assemble
    const long zs = (long)L.ldt * R.ldt;
    int nz = std::max(1, std::min(4, std::min(L.ldt, R.ldt) / 256));
    if ((size_t)nz * (size_t)zs > h->tmpb_elems) nz = 1;
    if (nz > 1) { s1.ksplit = nz; s1.slab_c = zs; }
    launch_sgemm(s1, st);
    SgemmArgs s2{};
    s2.A = h->Tf + L.toff; s2.lda = L.ldt; s2.transA = 0; s2.nslab_a = 1;
    s2.B = h->tmp1b; s2.ldb = R.ldt; s2.transB = 0;
    s2.C = h->tmp2b; s2.ldc = R.ldt; s2.M = L.ldt; s2.N = R.ldt; s2.K = L.ldt; s2.alpha = -1.f; s2.beta = 0.f; s2.upperA = 1;
    if (nz > 1) { s2.nslab_b = nz; s2.slab_b = zs; s2.ksplit = nz; s2.slab_c = zs; }
    launch_sgemm(s2, st);
    launch_t_assemble(h->Tf + nd.toff, h->Th + nd.toff, h->Tth + nd.toff, nd.ldt, nd.a0, h->Tf + L.toff, L.ldt, L.a0,
                      nd.c0, L.c1, h->Tf + R.toff, R.ldt, R.a0, nd.c1, h->tmp2b, R.ldt, st, nz, zs);
    if (h->tq_on && nd.id < (int)h->ev_T.size()) HIPQ(h, hipEventRecord(h->ev_T[nd.id], st));
    if (pid < (int)h->wq_ready.size()) h->wq_ready[pid] = 0;      // (a new T: the W of an earlier pass no longer matches)
    if (!h->defer_pair_w) pair_w(h, pid, st);
}

// W[rows x K] = V T of a block pair for Q formation (T upper triangular, the fp16 T has a unit diagonal: column n of the product takes tau_n)
static void pair_w(mpqr_handle_t h, int pid, hipStream_t st) {
    const Node nd = h->nodes[pid];
    if (h->Wq && pid < (int)h->wq_off.size() && h->wq_off[pid] >= 0 && nd.a0 == nd.c0) {
        const int rlo = rdown(nd.c0, 64);
        GemmArgs w{};
        w.A = h->Vh + (long)rlo * h->ldvh + nd.a0; w.lda = h->ldvh;
        w.Bt = h->Tth + nd.toff; w.ldb = nd.tld;
        w.C = h->Wq + h->wq_off[pid]; w.ldc = nd.ldt;
        w.M = h->m_pad - rlo; w.N = nd.ldt; w.K = nd.ldt; w.alpha = 1.f; w.in_scale = 1.f; w.nsplit = 1; w.tri = 2;
        w.cscale = h->Tf + nd.toff; w.cscale_ld = (long)nd.tld + 1;
        gemm_dispatch(A_H16, E_STORE_H16, w, st);
        h->wq_ready[pid] = 1;
    }
}

// prefix step: T of blocks 0..k from T of blocks 0..k-1 (in place, leading block of the same arena) and T of block k
static void merge_prefix(mpqr_handle_t h, int pid, hipStream_t st) {
    Range rg("mpqr:wy_T_prefix");
    const Node nd = h->nodes[pid];
    const Node L = h->nodes[nd.left], R = h->nodes[nd.right];
    const int ld = nd.tld, o = L.ldt, w = R.ldt;
    float* const Tr = h->Tf + nd.toff; half_t* const Thr = h->Th + nd.toff; half_t* const Tthr = h->Tth + nd.toff;
    if (h->tq_on) {                                       // (a distributed rank that has not factored a block yet has no T events)
        if (L.id < (int)h->ev_T.size()) HIPQ(h, hipStreamWaitEvent(st, h->ev_T[L.id], 0));
        if (R.id < (int)h->ev_T.size()) HIPQ(h, hipStreamWaitEvent(st, h->ev_T[R.id], 0));
    }
    if (L.tld != ld) {                                    // first step: block 0's own T becomes the leading block
        HIPQ(h, hipMemcpy2DAsync(Tr, (size_t)ld * 4, h->Tf + L.toff, (size_t)L.tld * 4, (size_t)o * 4, o, hipMemcpyDeviceToDevice, st));
        launch_t_colblock_h16(Tr, Thr, Tthr, ld, o, 0, o, st);
    }
    int nslab; long slab;
    gram(h, L, R, &nslab, &slab, st, h->S2, h->s2_elems);  // S = V_P^T V_k  (o x w)
    SgemmArgs s1{};                                        // tmp1 = S T_k
    s1.A = h->S2; s1.lda = w; s1.transA = 0; s1.nslab_a = nslab; s1.slab_a = slab;
    s1.B = h->Tf + R.toff; s1.ldb = R.tld; s1.transB = 0;
    s1.C = h->tmp1b; s1.ldc = w; s1.M = o; s1.N = w; s1.K = w; s1.alpha = 1.f; s1.beta = 0.f; s1.upperB = 1;
    launch_sgemm(s1, st);
    SgemmArgs s2{};                                        // T[0:o, o:o+w] = -T_P tmp1
    s2.A = Tr; s2.lda = ld; s2.transA = 0; s2.nslab_a = 1;
    s2.B = h->tmp1b; s2.ldb = w; s2.transB = 0;
    s2.C = Tr + o; s2.ldc = ld; s2.M = o; s2.N = w; s2.K = o; s2.alpha = -1.f; s2.beta = 0.f; s2.upperA = 1;
    launch_sgemm(s2, st);
    HIPQ(h, hipMemcpy2DAsync(Tr + (size_t)o * ld + o, (size_t)ld * 4, h->Tf + R.toff, (size_t)R.tld * 4, (size_t)w * 4, w,
                           hipMemcpyDeviceToDevice, st));  // diagonal block = T_k
    launch_t_colblock_h16(Tr, Thr, Tthr, ld, o + w, o, w, st);   // fp16 T and T^T of the new column block (diagonal block included)
    if (h->tq_on && nd.id < (int)h->ev_T.size()) HIPQ(h, hipEventRecord(h->ev_T[nd.id], st));
}

// Q = I - (V T) V^T over all reflectors at once (tall matrices; the tree of merged T's is complete: pairs_ready)
static int form_q_one_shot(mpqr_handle_t h) {
    const Node rt = h->nodes[h->qroot];
    const int Kr = rt.ldt, rlo = rdown(rt.c0, 64);
    if (h->tq_on && rt.id < (int)h->ev_T.size()) HIPQ(h, hipStreamWaitEvent(h->s0, h->ev_T[rt.id], 0));
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
    const bool rec = h->factored && h->far_used + 4 <= h->far_ev.size();
    if (rec) {
        e0 = h->far_ev[h->far_used]; e1 = h->far_ev[h->far_used + 1]; e2 = h->far_ev[h->far_used + 2]; e3 = h->far_ev[h->far_used + 3];
        h->q_first = h->far_used; h->far_used += 4;
        h->far_flops.push_back(2.0 * (double)h->m * (double)h->m * Kr);
        h->far_bytes.push_back(4.0 * (double)h->m * h->m + 2.0 * Kr * (2.0 * h->m));
        h->far_dims.push_back(h->m); h->far_dims.push_back(h->m); h->far_dims.push_back(Kr);
        h->far_flops_tn.push_back(0.0);
    }
    GemmArgs w{};                                         // W[m x Kr] = V T   (T upper triangular: k <= n)
    w.A = h->Vh + (long)rlo * h->ldvh + rt.a0; w.lda = h->ldvh;
    w.Bt = h->Tth + rt.toff; w.ldb = rt.tld;
    w.C = h->Wh; w.ldc = h->n_pad;
    w.M = h->m_pad - rlo; w.N = Kr; w.K = Kr; w.alpha = 1.f; w.in_scale = 1.f; w.nsplit = 1; w.tri = 2;
    w.cscale = h->Tf + rt.toff; w.cscale_ld = (long)rt.tld + 1;
    if (rec) HIPQ(h, hipEventRecord(e0, h->s0));
    gemm_dispatch(A_H16, E_STORE_H16, w, h->s0);
    if (rec) { HIPQ(h, hipEventRecord(e1, h->s0)); HIPQ(h, hipEventRecord(e2, h->s0)); }
    GemmArgs q{};                                         // Q[m x m] = I - W V^T   (V[j][k] = 0 for k > j)
    q.A = h->Wh; q.lda = h->n_pad;
    q.Bt = h->Vh + (long)rlo * h->ldvh + rt.a0; q.ldb = h->ldvh;
    q.C = h->dQ + (long)rlo * h->ldq + rlo; q.ldc = h->ldq;
    q.M = h->m - rlo; q.N = h->m - rlo; q.K = Kr; q.alpha = 1.f; q.in_scale = 1.f; q.nsplit = 1; q.tri = 2; q.eye_minus = 1;
    { static const int nt_env = []() { const char* e = getenv("MPQR_NT_C"); return e ? atoi(e) : 3; }(); q.nt_c = (nt_env & 2) ? 1 : 0; }   // Q is written once: streamed
    gemm_dispatch(A_H16, E_STORE_F32, q, h->s0);
    if (rec) HIPQ(h, hipEventRecord(e3, h->s0));
    h->q_formed = true;
    return MPQR_OK;
}

// Q = I and its transposed fp16 shadow = I (1.6 GB of stores at 16384^2).  run_block_loop issues this on the far stream at the START of
// the factorisation, where that stream has nothing to do until the first block is factored, instead of in front of Q formation.
static int init_q(mpqr_handle_t h, hipStream_t st) {
    HIPCHK(h, hipMemsetAsync(h->dQ, 0, (size_t)h->m_pad * h->ldq * sizeof(float), st));
    launch_set_identity(h->dQ, h->ldq, h->m, h->m, st);
    if (h->Qt && h->world == 1) {
        HIPCHK(h, hipMemsetAsync(h->Qt, 0, (size_t)(h->ldq + 256) * h->ldqt * sizeof(half_t), st));
        launch_set_identity_h16(h->Qt, h->ldqt, h->m, st);
    }
    return MPQR_OK;
}

int form_q(mpqr_handle_t h) {
    Range rg("mpqr:form_q");
    if (h->qroot >= 0 && h->pairs_ready && h->world == 1) return form_q_one_shot(h);
    if (!h->q_inited) { int rc = init_q(h, h->s0); if (rc) return rc; }
    h->q_inited = false;                                  // (consumed: the applies below overwrite the identity)
    if (h->Qt && h->world == 1) { h->shadow = h->Qt; h->ldshadow = h->ldqt; }
    const bool rec = h->world == 1 && h->factored;        // timed like the far updates (mpqr_get_timings: ms_q_*)
    h->n_q_ident_rows = 0;
    h->q_first = rec ? h->far_used : (size_t)-1;
    bool first = h->shadow != nullptr;                    // (Q = I and, with the shadow, Qt = I: init_q)
    for (int t = (int)h->tops.size() - 1; t >= 0; t--) {
        if (h->pairs_ready && t < (int)h->qpair.size() && h->qpair[t] >= 0) {      // two blocks at once, K = 2 outer_block
            const Node& pr = h->nodes[h->qpair[t]];
            h->shadow_write = t - 1 > 0;                    // nobody reads the shadow after the last apply
            h->q_ident_cols = pr.c1 - pr.c0;
            h->q_all_ident = first; first = false;
            apply_node(h, pr, h->dQ, h->ldq, pr.c0, h->m, false, 1.f, rec);
            h->q_ident_cols = 0; h->q_all_ident = false;
            t--;
            continue;
        }
        const Node& nd = h->nodes[h->tops[t]];
        h->shadow_write = t > 0;
        h->q_ident_cols = nd.c1 - nd.c0;
        h->q_all_ident = first; first = false;
        apply_node(h, nd, h->dQ, h->ldq, nd.c0, h->m, false, 1.f, rec);
        h->q_ident_cols = 0; h->q_all_ident = false;
    }
    h->shadow = nullptr;
    h->q_formed = true;
    return MPQR_OK;
}

hipError_t create_update_stream(hipStream_t* st, int prio) {
    // The far-update stream keeps off ONE of the 8 CUs of every shader engine (round 4): its GEMM workgroups hold a CU for ~85 us each, and
    // with the last CU of every engine free the chain's small kernels start at once wherever the dispatcher sends them
    // (16384^2: 35.9-36.1 -> 35.6 ms, two A/B pairs on one box; 6 / 5 / 4 of 8: 35.9 / 36.1 / 37.7; configs 3 and 5: no change).
    // Mask layout (tools/probe_cumask.hip): word i of the 8 x 32-bit mask = CU i of all 32 shader engines (8 XCCs x 4), bit s = engine s.
    // Rounds 2 and 3 repeated one 32-bit pattern over the 8 words, which selects ENGINES, not CUs -- the runtime ignores such masks
    // unless every XCC keeps an engine, and leaving whole engines to the chain does not help it (0x0000ffff: 40.1 ms): that, not the idea,
    // is what "CU masks do not help" measured.  MPQR_UPDATE_CU_ROWS=n (1..7) keeps n CUs per engine, 8: no mask (low-priority stream).
    int n = 7;
    if (const char* r = getenv("MPQR_UPDATE_CU_ROWS")) n = atoi(r);
    if (n >= 1 && n < 8) {
        uint32_t mask[8];
        for (int i = 0; i < 8; i++) mask[i] = i < n ? 0xffffffffu : 0u;
        if (hipExtStreamCreateWithCUMask(st, 8, mask) == hipSuccess) return hipSuccess;
    }
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio);
}

int check_shape(mpqr_handle_t h, int m, int n, int r) {
    if (!h) return MPQR_ERR_INVALID;
    t_dispatch_handle = h;
    if (m < 1 || n < 1 || r < 1) return fail(h, MPQR_ERR_INVALID, "m, n, r must be >= 1");
    if (n > m) return fail(h, MPQR_ERR_INVALID, "m >= n required (Householder QR of a tall or square matrix)");
    return MPQR_OK;
}

std::mutex g_default_mu;
mpqr_handle_t g_default = nullptr;

}  // namespace

// =====================================================================================
extern "C" {

const char* mpqr_version(void) { return "mpqr 0.1 (gfx950)"; }

void mpqr_abi_sizes(int out[3]) {
    if (!out) return;
    out[0] = (int)sizeof(mpqr_opts); out[1] = (int)sizeof(mpqr_metrics); out[2] = (int)sizeof(mpqr_timings);
}

void mpqr_default_opts(mpqr_opts* o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->precision = MPQR_PREC_FP16;
    o->outer_block = 0;
    o->form_q = 1;
    o->lookahead = 1;
}

int mpqr_create(mpqr_handle_t* out, int device) {
    if (!out) return MPQR_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_err = "no HIP device available (libmpqr has no CPU fallback)";
        return MPQR_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) { g_create_err = "device ordinal out of range"; return MPQR_ERR_INVALID; }
    mpqr_handle_t h = new mpqr_handle_s();
    h->device = device;
    mpqr_default_opts(&h->opts);
    memset(&h->last_t, 0, sizeof h->last_t);
    int prio_lo = 0, prio_hi = 0;
    MPQR_IGNORE(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithPriority(&h->s0, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        create_update_stream(&h->s1, prio_lo) != hipSuccess ||
        hipStreamCreateWithPriority(&h->sT, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_v, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_x, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_def, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_rest, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_dist_chain, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_dist_far, hipEventDisableTiming) != hipSuccess) {
        g_create_err = "hipSetDevice/hipStreamCreate failed";
        delete h;
        return MPQR_ERR_HIP;
    }
    for (int i = 0; i < 4; i++) MPQR_IGNORE(hipEventCreate(&h->ev[i]));
    if (hipMalloc((void**)&h->dmetric, 8 * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&h->dscalar, 4 * sizeof(float)) != hipSuccess) {
        g_create_err = "hipMalloc failed";
        delete h;
        return MPQR_ERR_ALLOC;
    }
    *out = h;
    return MPQR_OK;
}

int mpqr_destroy(mpqr_handle_t h) {
    if (!h) return MPQR_ERR_INVALID;
    MPQR_IGNORE(hipSetDevice(h->device));
    MPQR_IGNORE(hipStreamSynchronize(h->s0));
    if (h->s1) MPQR_IGNORE(hipStreamSynchronize(h->s1));
    if (h->sT) MPQR_IGNORE(hipStreamSynchronize(h->sT));
    free_plan(h);
    if (h->dmetric) MPQR_IGNORE(hipFree(h->dmetric));
    if (h->dscalar) MPQR_IGNORE(hipFree(h->dscalar));
    for (int i = 0; i < 4; i++) if (h->ev[i]) MPQR_IGNORE(hipEventDestroy(h->ev[i]));
    for (hipEvent_t e : h->ev_T) MPQR_IGNORE(hipEventDestroy(e));
    for (hipEvent_t e : h->ev_rows) MPQR_IGNORE(hipEventDestroy(e));
    if (h->ev_v) MPQR_IGNORE(hipEventDestroy(h->ev_v));
    if (h->ev_join) MPQR_IGNORE(hipEventDestroy(h->ev_join));
    if (h->sD) MPQR_IGNORE(hipStreamDestroy(h->sD));
    if (h->ev_x) MPQR_IGNORE(hipEventDestroy(h->ev_x));
    if (h->ev_def) MPQR_IGNORE(hipEventDestroy(h->ev_def));
    if (h->ev_rest) MPQR_IGNORE(hipEventDestroy(h->ev_rest));
    if (h->ev_dist_chain) MPQR_IGNORE(hipEventDestroy(h->ev_dist_chain));
    if (h->ev_dist_far) MPQR_IGNORE(hipEventDestroy(h->ev_dist_far));
    MPQR_IGNORE(hipStreamDestroy(h->s0));
    if (h->s1) MPQR_IGNORE(hipStreamDestroy(h->s1));
    if (h->sT) MPQR_IGNORE(hipStreamDestroy(h->sT));
    delete h;
    return MPQR_OK;
}

const char* mpqr_last_error(mpqr_handle_t h) { return h ? h->err.c_str() : g_create_err.c_str(); }

static int plan_common(mpqr_handle_t h, int m, int n, int r, const mpqr_opts* opts, int world, int rank) {
    int rc = check_shape(h, m, n, r);
    if (rc) return rc;
    if (world < 1 || rank < 0 || rank >= world) return fail(h, MPQR_ERR_INVALID, "bad world/rank");
    HIPCHK(h, hipSetDevice(h->device));
    mpqr_opts o;
    if (opts) o = *opts; else mpqr_default_opts(&o);
    if (o.precision != MPQR_PREC_FP16 && o.precision != MPQR_PREC_FP32 && o.precision != MPQR_PREC_FP8)
        return fail(h, MPQR_ERR_INVALID, "unknown precision");
    if (h->planned && h->m == m && h->n == n && h->r == r && h->world == world && h->rank == rank &&
        memcmp(&o, &h->opts, sizeof o) == 0)
        return MPQR_OK;
    HIPCHK(h, hipStreamSynchronize(h->s0));
    free_plan(h);
    h->m = m; h->n = n; h->r = r; h->opts = o; h->world = world; h->rank = rank;
    {   // The compact-WY T's are built on a third stream beside the chain's next V-only GEMM.  With the tree schedule alone this did not pay (the serial leaf T -> merge -> merge chain at the end of a
        // sub-tree is longer than the GEMM it hides behind: 60.9 vs 59.0 ms at 16384^2); the flat block schedule
        // (factor_block_flat) needs only the leaf's own T on the chain and is what the T stream is for (53.0 ms).
        h->tq_on = h->sT != nullptr && o.precision != MPQR_PREC_FP32;
    }
    if (const char* e = getenv("MPQR_GH_MIN_ROWS")) h->gh_min_rows = atoi(e);      // tuning hook
    if (const char* e = getenv("MPQR_TAIL_LEAF")) h->tail_leaf = atoi(e) != 0;       // A/B hook
    if (const char* e = getenv("MPQR_LEAF_MID")) h->leaf_mid = atoi(e) != 0;         // A/B hook
    if (const char* e = getenv("MPQR_TPOLL")) h->tpoll = atoi(e) != 0;               // A/B hook
    h->m_pad = rup(m, 256); h->n_pad = rup(n, 256);
    int Ko = o.outer_block > 0 ? o.outer_block : 1024;
    Ko = std::max(Ko, 32);
    if (r >= Ko) Ko = r; else if (Ko % 128 != 0) Ko = (Ko / r) * r;   // (128-aligned top-level blocks stay: tall leaves are 128-column windows)
    h->Ko = Ko;
    h->nloc = (world == 1) ? n : mpqr_part_local_cols(n, Ko, world, rank);
    h->qloc = (world == 1) ? m : mpqr_part_local_cols(m, Ko, world, rank);
    h->lda = rup(std::max(h->nloc, 1), 256); h->ldq = rup(std::max(h->qloc, 1), 256);
    h->ldvh = h->n_pad; h->ldvt = h->m_pad;
    // tree over the GLOBAL columns (every rank builds the same one)
    for (int c = 0; c < n; c += Ko) h->tops.push_back(build_tree(h, c, std::min(n, c + Ko)));
    size_t toff = 0; int max_ldt = 64;
    for (Node& nd : h->nodes) { nd.toff = toff; toff += (size_t)nd.ldt * nd.ldt; max_ldt = std::max(max_ldt, nd.ldt); }
    // pair nodes for Q formation (single GPU, fp16 / fp8 operands, both blocks 64-aligned): appended after the tree
    h->qpair.assign(h->tops.size(), -1);
    int q_ldt = 0;
    size_t q_half = 0;                                    // largest L.ldt x R.ldt of a tree merge (scratch of merge_pair)
    h->qroot = -1;
    {
        static const int one_on = []() { const char* e = getenv("MPQR_QONESHOT"); return e ? atoi(e) : 1; }();
        bool aligned = h->tops.size() >= 2;
        for (size_t p = 0; p + 1 < h->tops.size() && aligned; p++) {
            const Node& L = h->nodes[h->tops[p]]; const Node& R = h->nodes[h->tops[p + 1]];
            aligned = L.a0 == L.c0 && R.a0 == R.c0 && L.a1 == R.a0;
        }
        h->qmerge_after.assign(h->tops.size(), std::vector<int>());
        if (one_on && aligned && o.form_q && o.precision != MPQR_PREC_FP32 && (long)m >= 3L * n && m >= 2048) {
            // prefix nodes P_k = blocks 0..k, all in ONE arena of the full width (P_k's T is the leading principal block of
            // the root's): step k adds the column block  T[0:o, o:o+w] = -T_{P_{k-1}} (V_P^T V_k) T_k  (LAPACK larft order, as
            // the flat block schedule does for the leaves of a block).  One arena instead of a tree of them; measured the same
            // step time as a balanced binary tree of merges (111 ms at 65536 x 8192: the tree leaves a 7 ms merge behind
            // the chain's end, the prefix steps cost the chain as much through contention while they run beside it)
            const Node first = h->nodes[h->tops[0]], last = h->nodes[h->tops.back()];
            const int root_ldt = last.a1 - first.a0;
            const size_t root_toff = toff;
            toff += (size_t)root_ldt * root_ldt;
            int prev = h->tops[0];
            for (size_t k = 1; k < h->tops.size(); k++) {
                const Node L = h->nodes[prev], R = h->nodes[h->tops[k]];
                Node pr;
                pr.c0 = first.c0; pr.c1 = R.c1; pr.a0 = first.a0; pr.a1 = R.a1; pr.ldt = pr.a1 - pr.a0;
                pr.left = prev; pr.right = h->tops[k]; pr.toff = root_toff; pr.tld = root_ldt; pr.id = (int)h->nodes.size();
                q_half = std::max(q_half, (size_t)L.ldt * (size_t)R.ldt);
                h->nodes.push_back(pr);
                h->qmerge_after[k].push_back(pr.id);
                prev = pr.id;
            }
            h->qroot = prev;
        } else
        if (o.form_q && o.precision != MPQR_PREC_FP32) {             // (every rank of a distributed plan builds the same pairs)
            for (size_t p = 0; p + 1 < h->tops.size(); p += 2) {
                const Node L = h->nodes[h->tops[p]], R = h->nodes[h->tops[p + 1]];
                if (L.a0 != L.c0 || R.a0 != R.c0 || L.a1 != R.a0 || L.ldt < 256 || R.ldt < 256) continue;
                Node pr;
                pr.c0 = L.c0; pr.c1 = R.c1; pr.a0 = L.a0; pr.a1 = R.a1; pr.ldt = pr.a1 - pr.a0;
                pr.left = h->tops[p]; pr.right = h->tops[p + 1]; pr.toff = toff; pr.tld = pr.ldt; pr.id = (int)h->nodes.size();
                toff += (size_t)pr.ldt * pr.ldt;
                q_ldt = std::max(q_ldt, pr.ldt);
                h->qpair[p + 1] = pr.id;
                h->nodes.push_back(pr);
            }
        }
    }
    h->t_elems = toff;
    const size_t maxdim = (size_t)std::max(h->m_pad, h->n_pad);
    const int x_ldt = std::max(max_ldt, q_ldt);
    const int t_ldt = h->qroot >= 0 ? std::max(x_ldt, h->nodes[h->qroot].ldt) : x_ldt;   // slack rows of the T arenas
    h->xt_elems = std::max(maxdim * (size_t)x_ldt, (size_t)64 * max_ldt * max_ldt);
    h->yt_elems = maxdim * (size_t)x_ldt;
    h->s_elems = (size_t)64 * max_ldt * max_ldt;
    h->tmp_elems = (size_t)max_ldt * max_ldt;
    h->maxwg = h->m_pad / 256 + 2;
    // +1024 floats / +256 rows of slack: the 256-wide GEMM tiles load unmasked (results past M, N are masked at the store)
    if ((rc = dalloc(h, &h->dA, (size_t)h->m_pad * h->lda + 1024))) return rc;
    if ((rc = dalloc(h, &h->dQ, (size_t)h->m_pad * h->ldq + 1024))) return rc;
    if ((rc = dalloc(h, &h->Vh, (size_t)(h->m_pad + 256) * h->ldvh))) return rc;
    if ((rc = dalloc(h, &h->Vt, (size_t)(h->n_pad + 256) * h->ldvt))) return rc;
    if ((rc = dalloc(h, &h->vdiag, (size_t)h->n_pad))) return rc;
    if (o.precision == MPQR_PREC_FP32) {
        if (world != 1) return fail(h, MPQR_ERR_INVALID, "MPQR_PREC_FP32 is single-GPU only");
        if ((rc = dalloc(h, &h->Vf, (size_t)(h->m_pad + 256) * h->n_pad))) return rc;
        if ((rc = dalloc(h, &h->Yf, maxdim * (size_t)max_ldt + (size_t)256 * max_ldt))) return rc;
    }
    if (o.precision == MPQR_PREC_FP8) {
        if (world != 1) return fail(h, MPQR_ERR_INVALID, "MPQR_PREC_FP8 is single-GPU only");
        h->ld8k = rup(max_ldt, 128);
        const size_t e1 = (size_t)(h->m_pad + 256) * h->ld8k, e2 = (size_t)(h->ld8k + 256) * h->m_pad;
        const size_t e3 = (size_t)(h->n_pad + 256) * h->m_pad, e4 = (size_t)(h->n_pad + 256) * h->ld8k;
        if ((rc = dalloc(h, &h->V8n, e1)) || (rc = dalloc(h, &h->V8t, e2)) || (rc = dalloc(h, &h->A8t, e3)) || (rc = dalloc(h, &h->Y8, e4))) return rc;
        HIPCHK(h, hipMemsetAsync(h->V8n, 0, e1, h->s0)); HIPCHK(h, hipMemsetAsync(h->V8t, 0, e2, h->s0));
        HIPCHK(h, hipMemsetAsync(h->A8t, 0, e3, h->s0)); HIPCHK(h, hipMemsetAsync(h->Y8, 0, e4, h->s0));
    }
    if (o.precision != MPQR_PREC_FP32) {
        const size_t xe = h->yt_elems + (size_t)256 * x_ldt;
        if ((rc = dalloc(h, &h->Xh, xe)) || (rc = dalloc(h, &h->Xl, xe))) return rc;
        HIPCHK(h, hipMemsetAsync(h->Xh, 0, xe * sizeof(half_t), h->s0)); HIPCHK(h, hipMemsetAsync(h->Xl, 0, xe * sizeof(half_t), h->s0));
        if (o.lookahead) {
            if ((rc = dalloc(h, &h->Xh1, xe)) || (rc = dalloc(h, &h->Xl1, xe))) return rc;
            HIPCHK(h, hipMemsetAsync(h->Xh1, 0, xe * sizeof(half_t), h->s0)); HIPCHK(h, hipMemsetAsync(h->Xl1, 0, xe * sizeof(half_t), h->s0));
        }
    }
    if ((rc = dalloc(h, &h->Xt, h->xt_elems + (size_t)256 * x_ldt))) return rc;
    if ((rc = dalloc(h, &h->Yt, h->yt_elems + (size_t)256 * x_ldt))) return rc;
    HIPCHK(h, hipMemsetAsync(h->Xt, 0, (h->xt_elems + (size_t)256 * x_ldt) * sizeof(float), h->s0));
    HIPCHK(h, hipMemsetAsync(h->Yt, 0, (h->yt_elems + (size_t)256 * x_ldt) * sizeof(half_t), h->s0));
    if (o.lookahead && o.precision != MPQR_PREC_FP32) {
        h->xt1_elems = h->xt_elems;
        if ((rc = dalloc(h, &h->Xt1, h->xt1_elems + (size_t)256 * x_ldt))) return rc;
        if ((rc = dalloc(h, &h->Yt1, h->yt_elems + (size_t)256 * x_ldt))) return rc;
        HIPCHK(h, hipMemsetAsync(h->Xt1, 0, (h->xt1_elems + (size_t)256 * x_ldt) * sizeof(float), h->s0));
        HIPCHK(h, hipMemsetAsync(h->Yt1, 0, (h->yt_elems + (size_t)256 * x_ldt) * sizeof(half_t), h->s0));
        {   // one leaf (128 reflectors) onto at most a block's columns + the next block's pre-updated leaves, <= 32 split-K slabs
            int wmax = 0;
            for (int tp : h->tops) wmax = std::max(wmax, h->nodes[tp].c1 - h->nodes[tp].c0);
            h->xt2_elems = (size_t)(wmax + 1024) * 128 * 32;
            if ((rc = dalloc(h, &h->Xt2, h->xt2_elems + (size_t)256 * 128))) return rc;
            if ((rc = dalloc(h, &h->Yt2, h->xt2_elems + (size_t)256 * 128))) return rc;
            HIPCHK(h, hipMemsetAsync(h->Xt2, 0, (h->xt2_elems + (size_t)256 * 128) * sizeof(float), h->s0));
            HIPCHK(h, hipMemsetAsync(h->Yt2, 0, (h->xt2_elems + (size_t)256 * 128) * sizeof(half_t), h->s0));
        }
        for (size_t t = 0; t < h->tops.size(); t++) {
            hipEvent_t e1, e2;
            HIPCHK(h, hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            h->ev_node.push_back(e1); h->ev_cols.push_back(e2);
            hipEvent_t e3;
            HIPCHK(h, hipEventCreateWithFlags(&e3, hipEventDisableTiming));
            h->ev_cols2.push_back(e3);
        }
    }
    if ((rc = dalloc(h, &h->S, h->s_elems))) return rc;
    if ((rc = dalloc(h, &h->Sleaf, (size_t)128 * 128 * LEAF_MID_MAX_GROUPS))) return rc;
    if ((rc = dalloc(h, &h->mid_counter, (size_t)1))) return rc;
    HIPCHK(h, hipMemsetAsync(h->mid_counter, 0, sizeof(int), h->s0));
    if ((rc = dalloc(h, &h->tflag, (size_t)4))) return rc;                               // [0] the chain is past this leaf's T, [1] "a wait has timed out", [2] ... past this leaf's reflectors
    HIPCHK(h, hipMemsetAsync(h->tflag, 0, 4 * sizeof(int), h->s0));
    if (const char* e = getenv("MPQR_TPOLL_TIMEOUT_MS")) h->tpoll_ticks = (unsigned long long)std::max(1, atoi(e)) * 100000ull;
    h->tseq = 0;
    if ((rc = dalloc(h, &h->P, (size_t)2 * h->maxwg * 32))) return rc;
    if ((rc = dalloc(h, &h->Gp, (size_t)(h->m_pad / 64 + 2) * 16384))) return rc;      // (leaf_b: one partial per 64 rows)
    if ((rc = dalloc(h, &h->Gs, (size_t)16384))) return rc;
    if ((rc = dalloc(h, &h->Sp, (size_t)(h->m_pad / 64 + 4) * 16384))) return rc;
    if ((rc = dalloc(h, &h->Cv, (size_t)16384))) return rc;
    if (const char* e = getenv("MPQR_FUSED_LEAF")) h->fused_leaf = atoi(e) != 0;     // A/B hook
    if (h->fused_leaf && o.precision != MPQR_PREC_FP32) {
        if ((rc = dalloc(h, &h->Xp, (size_t)(h->m_pad / 64 + 4) * 16384))) return rc;
        if ((rc = dalloc(h, &h->Xs, (size_t)16384 * LEAF_MID_MAX_GROUPS))) return rc;
        if ((rc = dalloc(h, &h->Yfl, (size_t)16384))) return rc;
    }
    h->nflag = (int)h->nodes.size() + 1024;               // one flag per tree node (+ room for the stage calls' private trees)
    if ((rc = dalloc(h, &h->dflag, (size_t)h->nflag))) return rc;
    {   // the flag word the enqueuing thread polls: mapped, coherent host memory (a flagged leaf stores 1 into it, system scope)
        void* hp = nullptr; void* dp = nullptr;
        h->flag_words = (int)h->tops.size() + 16;
        if (hipHostMalloc(&hp, (size_t)h->flag_words * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { h->err = "hipHostMalloc failed"; return MPQR_ERR_ALLOC; }
        h->hflag_host = (int*)hp; memset(h->hflag_host, 0, (size_t)h->flag_words * sizeof(int));
        HIPCHK(h, hipHostGetDevicePointer(&dp, hp, 0));
        h->hflag_dev = (int*)dp;
    }
    {
        static const int stamps_env = []() { const char* e = getenv("MPQR_DBG_STAMPS"); return e ? atoi(e) : 0; }();
        if (stamps_env && !h->dbg_stamps) {
            void* hp = nullptr;
            h->dbg_stamps_n = 2 * (h->n_pad / 128 + 2);
            if (hipHostMalloc(&hp, (size_t)h->dbg_stamps_n * 8, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) { h->dbg_stamps = (unsigned long long*)hp; memset(hp, 0, (size_t)h->dbg_stamps_n * 8); }
        }
    }
    h->leaf_robust.assign(h->nodes.size(), 0);
    // T arena of a robustly factored tall leaf (sub-tree of 32-column leaves over <= 128 columns: 7 nodes, ldt <= 192)
    h->rb_elems = (size_t)8 * 192 * 192;
    if ((rc = dalloc(h, &h->rbTf, h->rb_elems + (size_t)256 * 256))) return rc;
    if ((rc = dalloc(h, &h->rbTh, h->rb_elems + (size_t)256 * 256))) return rc;
    if ((rc = dalloc(h, &h->rbTth, h->rb_elems + (size_t)256 * 256))) return rc;
    HIPCHK(h, hipMemsetAsync(h->rbTh, 0, (h->rb_elems + (size_t)256 * 256) * sizeof(half_t), h->s0));
    HIPCHK(h, hipMemsetAsync(h->rbTth, 0, (h->rb_elems + (size_t)256 * 256) * sizeof(half_t), h->s0));
    // event pools (nothing is created inside the timed region): 2 recorded far updates per block, 2 chain events per block
    for (size_t i = 0; i < 16 * h->tops.size() + 8; i++) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); h->far_ev.push_back(e); }   // + 1 per block for Q formation
    for (size_t i = 0; i < 2 * h->tops.size() + 2; i++) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); h->chain_ev.push_back(e); }
    if ((rc = dalloc(h, &h->dA0, (size_t)h->m_pad * h->lda + 1024))) return rc;
    if ((rc = dalloc(h, &h->tmp1, h->tmp_elems))) return rc;
    if ((rc = dalloc(h, &h->tmp2, h->tmp_elems))) return rc;
    {
        if (o.form_q && o.precision != MPQR_PREC_FP32 && h->qroot < 0) {
            h->ldqt = h->m_pad;                             // one row per LOCAL column of Q (all of them on a single GPU)
            if ((rc = dalloc(h, &h->Qt, (size_t)(h->ldq + 256) * h->ldqt))) return rc;
            // W = V T of every block pair, formed behind the pair's T on the far stream (merge_pair): Q formation's apply is then
            // X = Q2^T V and Q2 -= W X^T, without the Y = X T^T GEMM between them (0.8 ms of the 7.2 at 16384^2)
            static const int wq_on = []() { const char* e = getenv("MPQR_QW"); return e ? atoi(e) : 1; }();
            if (wq_on && world == 1) {
                h->wq_off.assign(h->nodes.size(), -1); h->wq_ready.assign(h->nodes.size(), 0);
                size_t tot = 0;
                for (int pid : h->qpair) if (pid >= 0) {
                    const Node& pr = h->nodes[pid];
                    h->wq_off[pid] = (long)tot;
                    tot += (size_t)(h->m_pad - rdown(pr.c0, 64) + 256) * pr.ldt;
                }
                if (tot > 0) {
                    if ((rc = dalloc(h, &h->Wq, tot))) return rc;
                    HIPCHK(h, hipMemsetAsync(h->Wq, 0, tot * sizeof(half_t), h->s0));
                } else h->wq_off.clear();
            }

        }
    }
    {
        // fp16 shadow of the trailing matrix, At[column][row] = fp16(s A): every far update's epilogue writes it, the next far update's
        // X = A2^T V reads it by LDS-DMA like any fp16 operand (round 4 default; MPQR_ASHADOW=0: X converts and transposes the fp32 matrix
        // while staging it).  Alone on the GPU the fp32-staging kernel runs at 640-660 TFLOP/s, the all-DMA one at 920, and the shadow stores
        // cost the update 830 -> 750 (tools/bench_gemm.py); beside the chain: far X 6.8 -> 5.2 ms, far update 4.1 -> 4.5 ms, step 39.25 ->
        // 38.97 ms (round 2, older kernels: even).  0.5 GB at 16384^2.
        static const int as_on = []() { const char* e = getenv("MPQR_ASHADOW"); return e ? atoi(e) : 1; }();
        if (as_on && world == 1 && o.lookahead && o.precision == MPQR_PREC_FP16) {
            h->ldat = h->m_pad;
            if ((rc = dalloc(h, &h->At, (size_t)(h->n_pad + 256) * h->ldat))) return rc;
            HIPCHK(h, hipMemsetAsync(h->At, 0, (size_t)(h->n_pad + 256) * h->ldat * sizeof(half_t), h->s0));
        }
    }
    if (q_ldt || h->qroot >= 0) {                         // scratch of the pair / tree merges (they run on the far-update stream)
        const size_t half = std::max(q_half, (size_t)max_ldt * max_ldt);
        h->s2_elems = std::max((size_t)16 * max_ldt * max_ldt, 2 * half);
        h->tmpb_elems = 4 * half;                          // (up to four split-K slabs of a pair merge's products)
        if ((rc = dalloc(h, &h->S2, h->s2_elems)) || (rc = dalloc(h, &h->tmp1b, h->tmpb_elems)) || (rc = dalloc(h, &h->tmp2b, h->tmpb_elems))) return rc;
    }
    if (h->qroot >= 0) {
        if ((rc = dalloc(h, &h->Wh, (size_t)(h->m_pad + 256) * h->n_pad))) return rc;
        HIPCHK(h, hipMemsetAsync(h->Wh, 0, (size_t)(h->m_pad + 256) * h->n_pad * sizeof(half_t), h->s0));
    }
    if ((rc = dalloc(h, &h->Tf, h->t_elems))) return rc;
    HIPCHK(h, hipMemsetAsync(h->Tf, 0, h->t_elems * sizeof(float), h->s0));
    if ((rc = dalloc(h, &h->Th, h->t_elems + (size_t)256 * t_ldt))) return rc;
    if ((rc = dalloc(h, &h->Tth, h->t_elems + (size_t)256 * t_ldt))) return rc;
    HIPCHK(h, hipMemsetAsync(h->Th, 0, (h->t_elems + (size_t)256 * t_ldt) * sizeof(half_t), h->s0));
    HIPCHK(h, hipMemsetAsync(h->Tth, 0, (h->t_elems + (size_t)256 * t_ldt) * sizeof(half_t), h->s0));
    HIPCHK(h, hipMemsetAsync(h->dA0, 0, (size_t)h->m_pad * h->lda * sizeof(float), h->s0));
    HIPCHK(h, hipMemsetAsync(h->dflag, 0, (size_t)h->nflag * sizeof(int), h->s0));
    HIPCHK(h, hipMemsetAsync(h->dA, 0, (size_t)h->m_pad * h->lda * sizeof(float), h->s0));
    HIPCHK(h, hipMemsetAsync(h->dQ, 0, (size_t)h->m_pad * h->ldq * sizeof(float), h->s0));
    if ((rc = clear_reflectors(h))) return rc;
    h->v_clean = true;
    HIPCHK(h, hipStreamSynchronize(h->s0));
    h->Aeff = h->dA;
    h->planned = true;
    return MPQR_OK;
}

int mpqr_plan(mpqr_handle_t h, int m, int n, int r, const mpqr_opts* opts) {
    return plan_common(h, m, n, r, opts, 1, 0);
}

static bool dispatch_failed(mpqr_handle_t h) {
    bool f = h->dispatch_error; h->dispatch_error = false;
    if (h->async_err != hipSuccess) {                       // an enqueue call failed: say which one (the caller's message names the phase)
        fprintf(stderr, "mpqr: %s failed: %s\n", h->async_what, hipGetErrorString(h->async_err));
        h->async_err = hipSuccess; f = true;
    }
    return f;
}
static int need_plan(mpqr_handle_t h, bool dist = false) {
    if (!h) return MPQR_ERR_INVALID;
    t_dispatch_handle = h;                                  // GEMMs enqueued by this call report to this handle
    if (!h->planned) return fail(h, MPQR_ERR_STATE, "mpqr_plan has not been called");
    if (!dist && h->world != 1) return fail(h, MPQR_ERR_STATE, "handle holds a distributed plan: use the mpqr_dist_* calls");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(h, MPQR_ERR_HIP, "hipSetDevice failed");
    return MPQR_OK;
}

int mpqr_set_matrix_host(mpqr_handle_t h, const float* A, long ld) {
    int rc = need_plan(h); if (rc) return rc;
    if (!A || ld < h->n) return fail(h, MPQR_ERR_INVALID, "bad matrix pointer / leading dimension");
    HIPCHK(h, hipMemcpy2DAsync(h->dA0, h->lda * sizeof(float), A, ld * sizeof(float), (size_t)h->n * sizeof(float), h->m,
                               hipMemcpyHostToDevice, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    h->factored = false; h->q_formed = false; h->have_input = true; h->robust = false;
    std::fill(h->leaf_robust.begin(), h->leaf_robust.end(), 0);
    return MPQR_OK;
}

int mpqr_set_matrix_device(mpqr_handle_t h, const float* dA, long ld) {
    int rc = need_plan(h); if (rc) return rc;
    if (!dA || ld < h->n) return fail(h, MPQR_ERR_INVALID, "bad matrix pointer / leading dimension");
    HIPCHK(h, hipMemcpy2DAsync(h->dA0, h->lda * sizeof(float), dA, ld * sizeof(float), (size_t)h->n * sizeof(float), h->m,
                               hipMemcpyDeviceToDevice, h->s0));
    h->factored = false; h->q_formed = false; h->have_input = true; h->robust = false;
    std::fill(h->leaf_robust.begin(), h->leaf_robust.end(), 0);
    return MPQR_OK;
}

int mpqr_generate_matrix(mpqr_handle_t h, uint64_t seed) {
    int rc = need_plan(h); if (rc) return rc;
    launch_generate(h->dA0, h->lda, h->m, h->n, seed, h->n, 1, 1, 0, h->s0);
    h->factored = false; h->q_formed = false; h->have_input = true; h->robust = false;
    std::fill(h->leaf_robust.begin(), h->leaf_robust.end(), 0);
    return MPQR_OK;
}

// enqueue copy-in + the whole block loop; flags[id] != 0: the Gram-Householder leaf `id` was too ill-conditioned
// start > 0 (restart after a flagged leaf): blocks [0, start) of the previous pass are kept -- their reflectors, T's and columns of
// R -- and the columns right of them are brought back to the state in which block `start` found them: the input with the far
// updates of the kept blocks applied again (GEMM work only, no panel chain).  The pass then runs blocks start .. end.
static int run_block_loop(mpqr_handle_t h, int start = 0) {
    int rc;
    h->v8_node = -1;
    h->gram_ready_c0 = -1; h->lane2_twait = 0; h->rest_pending = false; h->rest_in_solve = false; h->rest_split = false; h->next_block_flat = false;
    if (start <= 0) h->n_fused_leaves = 0;
    if (h->stream_out) { h->rows_rec.assign(h->tops.size(), 0); if (start > 0) h->rows_valid = false; }
    h->pairs_ready = false; h->q_first = (size_t)-1;
    const auto host_t0 = std::chrono::steady_clock::now();  // host time to enqueue the block loop (ms_host_enqueue)
    if (start > 0 && (size_t)start < h->far_mark.size()) {
        // restart: the aborted pass's recordings for blocks >= start describe work that is thrown away -- rewind the event pools to
        // where block `start` began, so that ms_panel / ms_far_* / the roofline figures cover only work that is kept
        h->far_used = h->far_mark[start]; h->chain_used = h->chain_mark[start];
        h->far_flops.resize(h->far_used / 4); h->far_bytes.resize(h->far_used / 4); h->far_dims.resize(3 * (h->far_used / 4)); h->far_flops_tn.resize(h->far_used / 4);
    }
    if (start <= 0) {
        start = 0;
        h->far_used = 0; h->far_flops.clear(); h->far_bytes.clear(); h->far_dims.clear(); h->far_flops_tn.clear(); h->chain_used = 0;
        if (!h->copied_in)                                  // (a later pass from block 0; the first one found the copy made by mpqr_factor)
            HIPCHK(h, hipMemcpyAsync(h->dA, h->dA0, (size_t)h->m_pad * h->lda * sizeof(float), hipMemcpyDeviceToDevice, h->s0));
        h->copied_in = false;
        // 1.1 GB of memsets (0.25-0.3 ms at 16384^2) only when the stores may hold something a kernel could read before this pass has
        // rewritten it: after a flagged / restarted / robust pass, a stage call or a distributed factorisation.  A clean factorisation
        // leaves finite reflectors exactly where this one writes its own before any kernel reads them; what no kernel ever writes (the
        // entries above the diagonal, the pads) is still zero from the plan.  The only stale values a kernel can meet are the next
        // leaf's reflectors inside a 64-aligned widened range of a SHORT leaf, where T is zero-padded (finite x 0).
        if (!h->v_clean) { if ((rc = clear_reflectors(h))) return rc; }
        h->v_clean = false;                                 // (mpqr_factor declares the stores clean again after a clean single pass)
    } else {
        Range rg("mpqr:restart_replay");
        const int cs = h->nodes[h->tops[start]].c0;
        HIPCHK(h, hipMemcpy2DAsync(h->dA + cs, (size_t)h->lda * sizeof(float), h->dA0 + cs, (size_t)h->lda * sizeof(float),
                                   (size_t)(h->lda - cs) * sizeof(float), (size_t)h->m_pad, hipMemcpyDeviceToDevice, h->s0));
        if ((rc = clear_reflectors(h, cs))) return rc;
        h->at_read = false;                                 // (the fp16 shadow of the trailing matrix is stale)
        for (int s = 0; s < start; s++) apply_node(h, h->nodes[h->tops[s]], h->dA, h->lda, cs, h->n, true, h->a_scale, false, 0, true);
    }
    HIPCHK(h, hipMemsetAsync(h->dflag, 0, (size_t)h->nflag * sizeof(int), h->s0));
    for (int b = 0; b < h->flag_words; b++) __atomic_store_n(h->hflag_host + b, 0, __ATOMIC_RELAXED);   // (no leaf of an earlier pass is still running: mpqr_factor waits for every pass's last leaf)
    h->watch_flags = true;
    bool aborted = false;
    const bool la = h->Xt1 != nullptr;                    // look-ahead: far updates on s1, panel chain on s0
    if (la) {
        // s1 must see the copy-in / clears issued on s0
        HIPCHK(h, hipEventRecord(h->ev[3], h->s0));
        HIPCHK(h, hipStreamWaitEvent(h->s1, h->ev[3], 0));
        // Q = I now, on the idle far stream (form_q then starts with its first GEMM); tall matrices store Q in one product, no init
        if (start == 0 && h->opts.form_q && h->qroot < 0 && h->world == 1 && !h->q_inited) { if ((rc = init_q(h, h->s1))) return rc; h->q_inited = true; }
    }
    // Look-ahead.  Far update t (stream s1) is released in two parts: (a) the columns the chain needs next, (b) the rest,
    // which overlaps the next block's panels.  With the flat block schedule (ext[t]) block t's own in-block updates also
    // cover the first leaf of block t+1, so the chain runs across the boundary and (a) = the other columns of block t+1
    // plus the first leaf of block t+2 (when that block is flat too), waited for after block t+1's first leaf is enqueued.
    // Tree-scheduled blocks keep the older split: first leaf of block t+1 first (ev_cols, waited for before the block).
    const size_t nt = h->tops.size();
    h->far_mark.resize(nt + 1); h->chain_mark.resize(nt + 1);
    // MPQR_EXT_LEAVES (default 2): how many leading leaves of block t+1 the in-block updates of a flat block t reach.  With 2 the
    // chain meets the columns of part (a) one leaf after the boundary: the first leaf's update of the rest of its block is deferred
    // to the T stream (factor_block_flat), and part (a) -- which needs the block's complete T -- has a whole leaf to arrive in.
    std::vector<char> ext(nt + 1, 0), flat(nt + 1, 0);
    std::vector<int> cfirst(nt + 2, h->n);                  // end of the leaves of block t that the previous block's updates reach
    std::vector<int> npre(nt + 2, 0);                       // ... and how many they are (1: the first leaf only, also for tree blocks)
    {
        std::vector<int> lv;
        static const int ext_on = []() { const char* e = getenv("MPQR_EXT_LOOKAHEAD"); return e ? atoi(e) : 1; }();
        // (round 5: four leaves with the fused leaf -- the chain stream then works on ONE next panel per leaf whatever the range, the wider
        //  range only costs the T stream's deferred updates a few columns, and the wait for part (a) moves from the block's second leaf,
        //  where gh_solve's final poll sat 100 - 250 us in every block (MPQR_DBG_STAMPS), to its fourth: 33.45 -> 33.03 ms; 5 / 6 / 8: 33.4 / 34.0 / 35.2)
        static const int ext_env = []() { const char* e = getenv("MPQR_EXT_LEAVES"); return e ? std::max(1, atoi(e)) : 0; }();
        const int ext_short = ext_env ? ext_env : ((h->fused_leaf && h->Xp) ? 4 : 2);
        for (size_t t = 0; t < nt; t++) {
            flat[t] = flat_block_ok(h, h->tops[t], lv);
            ext[t] = la && ext_on && h->tq_on && t + 1 < nt && flat[t];
        }
        for (size_t t = 0; t < nt; t++) {
            std::vector<int> leaves_t;
            collect_leaves(h, h->tops[t], leaves_t);
            // (at most half of the block's leaves: a 512-column block keeps the two leaves of rounds 3 - 4)
            // (tall leaves -- >= 20480 rows, where the deferred updates are large -- keep two: 65536 x 8192 factors in 39.3 ms with two, 41.5 with four)
            const int ext_leaves = ext_env ? ext_env : (h->m - h->nodes[h->tops[t]].c0 >= 20480 ? 2 : ext_short);
            npre[t] = (t > 0 && ext[t - 1] && flat[t]) ? std::min<int>(ext_leaves, std::max<int>(ext_env ? (int)leaves_t.size() : (int)leaves_t.size() / 2, 1)) : 1;
            cfirst[t] = h->nodes[leaves_t[npre[t] - 1]].c1;
        }
    }
    std::vector<int> lvtmp;
    h->defer_join = la && h->tq_on;
    // far updates beyond the next two blocks are taken pairwise (K = 2 outer blocks: the GEMMs run ~25 % faster and there
    // are half as many), with the pair T that Q formation needs anyway
    const bool far_pair = la && h->opts.precision == MPQR_PREC_FP16 && h->opts.form_q && h->S2 && h->qroot < 0;
    bool deferred = false, pair_merged = false;
    int pending_far = -1;                                   // block whose far update waits for the next block's first leaf
    // A large pair update is enqueued in two column halves, the second one a block later BEHIND that block's urgent updates (its part (a),
    // the even block's extra columns).  The far stream works in order: the pair (0, 1) at 16384^2 takes 2.7 ms, longer than block 2's chain
    // beside it, and block 3 sat ~0.5 ms at its third leaf waiting for a 0.25 ms part (a) queued behind it (MPQR_DBG_STAMPS / MPQR_DBG_BLOCKS).
    // Only where the update is that long (>= 12000 columns: the first pair of 16384^2; 33.0 -> 32.6 ms): the X GEMM of a half has half the
    // tiles, and from the second pair on that costs more than the queueing did (8192: 33.4 ms; halves of 33 / 66 / 10 %: 33.0).
    struct HalfB { int node = -1, lo = 0, hi = 0; bool at_read = false; } half_b;
    static const int split_min = []() { const char* e = getenv("MPQR_PAIR_SPLIT_MIN"); return e ? atoi(e) : 12000; }();
    auto flush_half_b = [&]() {
        if (half_b.node < 0) return;
        h->at_read = half_b.at_read;
        apply_node(h, h->nodes[half_b.node], h->dA, h->lda, half_b.lo, half_b.hi, true, h->a_scale, true, 1, true);
        half_b.node = -1;
    };
    // W = V T of a pair (Q formation) is not urgent: it goes out behind the NEXT block's far update, where the far stream has room -- in
    // front of the pair's own update it pushed that update into the next block's part (a) (ms_panel + 0.7 ms)
    int pending_w = -1;
    h->defer_pair_w = la;
    auto flush_w = [&]() { if (pending_w >= 0) { pair_w(h, pending_w, h->s1); pending_w = -1; } };
    auto far_update = [&](size_t t) -> int {
        const Node nd = h->nodes[h->tops[t]];
        Range rg("mpqr:far_update");
        h->at_read = t >= 1 && !(start > 0 && (int)t == start);   // far update t-1 wrote the shadow of every column this one reads (not after a restart)
        HIPCHK(h, hipEventRecord(h->ev_node[t], h->node_done_stream ? h->node_done_stream : h->s0));
        HIPCHK(h, hipStreamWaitEvent(h->s1, h->ev_node[t], 0));
        if (t + 1 < nt) {
            const Node nx = h->nodes[h->tops[t + 1]];
            const int cf1 = cfirst[t + 1];
            const int a_end = ext[t + 1] ? cfirst[t + 2] : nx.c1;
            if (!ext[t]) {
                apply_node(h, nd, h->dA, h->lda, nx.c0, cf1, true, h->a_scale, false, 1, true);   // next block's first leaf first ...
                HIPCHK(h, hipEventRecord(h->ev_cols[t + 1], h->s1));
            }
            apply_node(h, nd, h->dA, h->lda, cf1, a_end, true, h->a_scale, true, 1, true);        // ... then what its first apply touches ...
            HIPCHK(h, hipEventRecord(h->ev_cols2[t + 1], h->s1));
            if (far_pair && (t % 2) == 0 && h->qpair[t + 1] >= 0) {
                // even block of a pair: only the columns the NEXT block's part (a) will need get this block's update now
                // (K = outer block); everything beyond waits for the pair's T and takes both blocks at K = 2 outer blocks
                const int e_next = (t + 2 < nt) ? (ext[t + 2] ? cfirst[t + 3] : h->nodes[h->tops[t + 2]].c1) : h->n;
                apply_node(h, nd, h->dA, h->lda, a_end, std::min(e_next, h->n), true, h->a_scale, true, 1, true);
                deferred = e_next < h->n;
                flush_half_b();                             // the previous pair's second half, behind this block's urgent columns
            } else if (deferred) {
                // odd block: the pair (t-1, t) onto everything neither of them has reached yet
                flush_half_b();                             // (cannot be pending here: the even block between took it)
                merge_pair(h, h->qpair[t], h->s1); pair_merged = true;
                flush_w(); if (t + 1 < nt) pending_w = h->qpair[t];
                h->at_read = t >= 3;                        // the first pair's columns have not been written by a far update yet
                // first half: at least everything the next block's far update touches (its part (a) and extra columns end at cfirst[t + 4])
                int mid = h->n;
                if (split_min > 0 && t + 2 < nt && h->n - a_end >= split_min) {
                    const int need = cfirst[std::min(t + 4, nt + 1)];
                    mid = std::max(need, a_end + rup((h->n - a_end) / 2, 256));
                    if (h->n - mid < 256) mid = h->n;
                }
                apply_node(h, h->nodes[h->qpair[t]], h->dA, h->lda, a_end, mid, true, h->a_scale, true, 1, true);
                if (mid < h->n) { half_b.node = h->qpair[t]; half_b.lo = mid; half_b.hi = h->n; half_b.at_read = t >= 3; }
                deferred = false;
            } else
            apply_node(h, nd, h->dA, h->lda, a_end, h->n, true, h->a_scale, true, 1, true);       // ... the rest overlaps its panels
        }
        if ((t % 2) == 0 || t + 1 >= nt) flush_half_b();       // (an even block that took another branch; the last block)
        if ((t % 2) == 0 || t + 1 >= nt) flush_w();
        // Q formation works on pairs of blocks: the pair's T behind this block's far update, beside the next panels
        // (the LAST pair gets no W: its product would sit between the chain's end and Q formation's first apply, where Y = X T^T of a
        //  2048-column apply is cheaper -- and a matrix of one pair keeps the arithmetic of rounds 1 - 4)
        if (h->opts.form_q && h->qpair[t] >= 0 && h->S2 && !pair_merged) { merge_pair(h, h->qpair[t], h->s1); flush_w(); if (t + 1 < nt) pending_w = h->qpair[t]; }
        pair_merged = false;
        if (h->opts.form_q && h->qroot >= 0) for (int id : h->qmerge_after[t]) merge_prefix(h, id, h->s1);   // ... or the T of all blocks so far
        if (h->stream_out && start == 0 && t < h->ev_rows.size()) {              // (the rows of block t-2 are final behind this: stream_rows)
            HIPCHK(h, hipEventRecord(h->ev_rows[t], h->s1));
            h->rows_rec[t] = 1;
        }
        return MPQR_OK;
    };
    for (size_t t = (size_t)start; t < nt; t++) {
        const Node nd = h->nodes[h->tops[t]];
        h->far_mark[t] = h->far_used; h->chain_mark[t] = h->chain_used;
        h->cur_block = (int)t;
        if (la && t > 0) {
            if (!ext[t - 1]) HIPCHK(h, hipStreamWaitEvent(h->s0, h->ev_cols[t], 0));
            h->wait_after_first_leaf = h->ev_cols2[t];
        }
        h->ext_c1 = ext[t] ? cfirst[t + 1] : 0;
        h->pre_leaves = (la && t > (size_t)start) ? npre[t] : 0;    // (the first block of a pass finds all of its columns up to date)
        const bool timed = h->chain_used + 2 <= h->chain_ev.size();
        if (timed) HIPCHK(h, hipEventRecord(h->chain_ev[h->chain_used], h->s0));
        if (pending_far >= 0) { const size_t tp = (size_t)pending_far; pending_far = -1; h->far_hook = [&far_update, tp]() { return far_update(tp); }; }
        h->next_block_flat = t + 1 < nt && flat[t + 1] && flat[t];
        rc = factor_node(h, h->tops[t], true);
        h->next_block_flat = false;
        h->ext_c1 = 0;
        if (rc == MPQR_ABORT_PASS) {          // a leaf of this pass is flagged: nothing enqueued from here on would be kept
            aborted = true;
            h->wait_after_first_leaf = nullptr; h->op1_stream = nullptr; h->far_hook = nullptr;
            break;
        }
        if (!rc && h->far_hook) { auto fh = std::move(h->far_hook); h->far_hook = nullptr; rc = fh(); }   // (no Gram-Householder leaf took it)
        h->far_hook = nullptr;
        if (rc) { h->defer_join = false; h->watch_flags = false; return rc; }
        if (h->wait_after_first_leaf) {       // (no leaf launched: cannot happen, but never leave the wait pending)
            HIPCHK(h, hipStreamWaitEvent(h->s0, h->wait_after_first_leaf, 0));
            h->wait_after_first_leaf = nullptr;
        }
        if (timed) { HIPCHK(h, hipEventRecord(h->chain_ev[h->chain_used + 1], h->s0)); h->chain_used += 2; }
        if (!la) {
            apply_node(h, nd, h->dA, h->lda, nd.c1, h->n, true, h->a_scale, true, 0, true);
            if (h->opts.form_q && h->qpair[t] >= 0 && h->S2) merge_pair(h, h->qpair[t], h->s0);
            if (h->opts.form_q && h->qroot >= 0) for (int id : h->qmerge_after[t]) merge_prefix(h, id, h->s0);
            continue;
        }
        // look-ahead: the far update of block t.  When the chain runs on into the next block (ext[t]) it is enqueued from inside that
        // block's first leaf, behind its gh_gram (factor_block_flat, far_hook): ev_node[t] is then recorded on the chain stream
        // behind that launch, the far stream's first GEMMs start together with the leaf's gh_solve and fill the CUs the solve
        // leaves idle, instead of taking every CU while gh_gram (139 KB of LDS per workgroup: whole CUs) wants them.
        if (ext[t] && t + 1 < nt && flat[t + 1]) { pending_far = (int)t; continue; }   // (a flat block takes the hook in its first leaf)
        if ((rc = far_update(t))) { h->defer_join = false; h->watch_flags = false; return rc; }
    }
    if (!aborted) flush_half_b();                           // (every far_update has run; nothing can be pending)
    flush_w();                                              // (also of an aborted pass: the pair's T stays valid when the restart begins behind it)
    h->defer_pair_w = false;
    h->watch_flags = false; h->pass_aborted = aborted; h->pre_leaves = 0; h->cur_block = 0;
    h->pairs_ready = !aborted && h->opts.form_q && h->S2 != nullptr;     // (pairs or the whole tree)
    if (h->defer_join) {                                    // the T stream's work of every block, once
        h->defer_join = false;
        HIPCHK(h, hipEventRecord(h->ev_join, h->sT));
        HIPCHK(h, hipStreamWaitEvent(h->s0, h->ev_join, 0));
    }
    if (la) {   // join: everything after this point (Q formation, read-backs) is ordered after both streams
        HIPCHK(h, hipEventRecord(h->ev[3], h->s1));
        HIPCHK(h, hipStreamWaitEvent(h->s0, h->ev[3], 0));
    }
    HIPCHK(h, hipEventRecord(h->ev[1], h->s0));
    h->host_enqueue_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    HIPCHK(h, hipGetLastError());
    return MPQR_OK;                                         // nothing is synchronised here: mpqr_factor waits for ev[1] AFTER it has enqueued Q formation
}

// Drop-in call: packed rows of finished blocks to the caller's buffer while the rest of the factorisation and Q formation run.  Called by
// mpqr_factor's first pass when everything is enqueued and the host would only wait.  The rows of block b are final behind far update b + 2
// (its own pair's second half goes out with that one; later blocks touch later rows only); behind the block loop everything is.  A D2H copy
// to pageable memory blocks this thread, which has nothing else to do.  If the pass turns out flagged or timed out, the repair passes change
// far columns of finished rows at rounding level: rows_valid is dropped and the caller copies everything again.
static int stream_rows(mpqr_handle_t h) {
    const int nt = (int)h->tops.size();
    h->rows_streamed = 0; h->rows_valid = false;
    if (!h->sD || !h->dstage || h->pass_aborted || h->stage_elems < (size_t)(h->m + 1) * h->n) return MPQR_OK;
    auto rows_out = [&](int r1) -> int {
        const int r0 = h->rows_streamed;
        if (r1 <= r0) return MPQR_OK;
        launch_pack_factor_rows(h->dA, h->lda, h->vdiag, h->dstage, h->m, h->n, r0, r1, h->sD);
        HIPCHK(h, hipMemcpyAsync(h->stream_out + (size_t)r0 * h->n, h->dstage + (size_t)r0 * h->n, (size_t)(r1 - r0) * h->n * sizeof(float),
                                 hipMemcpyDeviceToHost, h->sD));
        HIPCHK(h, hipStreamSynchronize(h->sD));
        h->rows_streamed = r1;
        return MPQR_OK;
    };
    auto pass_unusable = [&]() {                           // leaf flags, or the T stream's time-out word (not the deflated-columns count)
        return flag_words_set(h, h->flag_words - 2) ||
               (h->hflag_host && h->flag_words > 0 && __atomic_load_n(h->hflag_host + h->flag_words - 1, __ATOMIC_RELAXED) != 0);
    };
    int rc;
    for (int b = 0; b + 2 < nt; b++) {
        if (!h->rows_rec[b + 2]) break;
        HIPCHK(h, hipEventSynchronize(h->ev_rows[b + 2]));
        if (pass_unusable()) return MPQR_OK;                         // a flagged leaf or a timed-out wait: this pass will not be kept
        // (packed row r holds row r of R and row r - 1 of the reflectors: up to the block's last row)
        if ((rc = rows_out(std::min(h->nodes[h->tops[b]].c1, h->m)))) return rc;
    }
    HIPCHK(h, hipEventSynchronize(h->ev[1]));              // the block loop is done (Q formation goes on)
    if (pass_unusable()) return MPQR_OK;
    if ((rc = rows_out(h->m + 1))) return rc;
    h->rows_valid = true;
    return MPQR_OK;
}

int mpqr_factor(mpqr_handle_t h) {
    int rc = need_plan(h); if (rc) return rc;
    if (!h->have_input) return fail(h, MPQR_ERR_STATE, "no input matrix has been set");
    // timing starts here; the copy-in of the resident input and the maximum behind the power-of-two scale are one pass over it
    HIPCHK(h, hipEventRecord(h->ev[0], h->s0));
    if ((rc = compute_scale(h, h->dA0, h->dA))) return rc;
    h->copied_in = true;
    // Gram-Householder leaves that gh_solve flags (a column that cannot be reflected, or rho < 1e-8) are redone on the
    // column-by-column kernels.  A flagged leaf also raises a word in mapped host memory that the enqueuing thread reads before
    // every leaf: the pass stops being enqueued there (run_block_loop), the queue drains, and the next pass RESTARTS at the
    // top-level block that holds the flagged leaf, with that leaf on the robust path (everything left of the block is final).
    // Only the FIRST flagged leaf (in column order) is believed: the leaves behind it worked on columns it had spoiled, so their
    // flags say nothing.  Six restarts at most; after that every tall leaf goes robust and the pass runs from block 0.
    // n_passes counts passes (stopped ones included), restart_block is the block the last pass started from.
    // No host synchronisation between the block loop and Q formation (round 4): Q formation is enqueued right behind the pass, THEN the
    // host waits for the event behind the block loop and reads the mapped flag word (zero unless a leaf was flagged: the normal case
    // returns with Q formation still running, as before).  A flagged pass has a speculative Q formation in the queue; it is simply
    // formed again after the repair passes.
    std::vector<int> flags;
    h->n_passes = 0; h->n_tpoll_retries = 0;
    const bool tpoll_saved = h->tpoll; bool tpoll_retry = false;
    int start = 0, gh_total = 0;
    const int nblocks = (int)h->tops.size();
    for (int pass = 0; pass < 18; pass++) {
        h->n_passes++;
        h->n_gh_leaves = 0;
        h->restart_block = start;
        if ((rc = run_block_loop(h, start))) return rc;
        gh_total += h->n_gh_leaves;
        h->factored = true;
        if (h->opts.form_q && !h->pass_aborted) { if ((rc = form_q(h))) return rc; }
        HIPCHK(h, hipEventRecord(h->ev[2], h->s0));
        if (h->stream_out) { if (pass == 0) { if ((rc = stream_rows(h))) return rc; } else h->rows_valid = false; }
        HIPCHK(h, hipEventSynchronize(h->ev[1]));           // the block loop (not Q formation) is done: the flag word is final
        HIPCHK(h, hipGetLastError());
        if (h->hflag_host && h->flag_words > 0 && __atomic_load_n(h->hflag_host + h->flag_words - 1, __ATOMIC_RELAXED) != 0) {
            // A polling wait of the T stream gave up: that stream ran ahead of the chain, the pass is garbage.  A slow chain cannot be told
            // from a lost publish, so the factorisation is repeated ONCE with event hand-offs (no polling, no fused leaves) before failing.
            HIPCHK(h, hipStreamSynchronize(h->s0));
            if (h->s1) HIPCHK(h, hipStreamSynchronize(h->s1));
            if (h->sT) HIPCHK(h, hipStreamSynchronize(h->sT));
            fprintf(stderr, "mpqr: the T stream's wait for the chain stream timed out (wait_flag_kernel)%s\n", (h->tpoll && !tpoll_retry) ? "; repeating the factorisation with event hand-offs" : "");
            if (!h->tpoll || tpoll_retry) { h->tpoll = tpoll_saved; return fail(h, MPQR_ERR_HIP, "the T stream's wait for the chain stream timed out (wait_flag_kernel)"); }
            __atomic_store_n(h->hflag_host + h->flag_words - 1, 0, __ATOMIC_RELAXED);
            HIPCHK(h, hipMemsetAsync(h->tflag, 0, 4 * sizeof(int), h->s0)); h->tseq = 0;
            tpoll_retry = true; h->tpoll = false; h->n_tpoll_retries++;
            h->factored = false; h->q_formed = false; h->q_inited = false; h->v_clean = false;
            start = 0;
            continue;
        }
        if (!flag_words_set(h, h->flag_words - 2) && !h->pass_aborted) break;
        h->factored = false; h->q_formed = false;
        flags.assign(h->nodes.size(), 0);
        HIPCHK(h, hipMemcpyAsync(flags.data(), h->dflag, flags.size() * sizeof(int), hipMemcpyDeviceToHost, h->s0));
        HIPCHK(h, hipStreamSynchronize(h->s0));
        // Only the FIRST flagged leaf in column order is believed.  Round 4 tried believing every flagged leaf of the first flagged
        // block (one restart per block instead of one per leaf): on the rank-deficient Jacobian stand-in all 8 leaves
        // of the block flag once the first one has, so 8 leaves went to the column-by-column kernels where 1 was ill conditioned
        // (4.9 x the full-rank time instead of 1.3 x) -- also when the flagged leaf is made to leave an identity transform behind
        // (C = 0, V_top = 0), i.e. it is not overflowing garbage that trips them: a leaf's flag is only meaningful once every leaf
        // before it has been applied.  Every restart therefore repairs exactly one leaf, at the cost of re-running its block.
        int bad = -1;                                     // the flagged leaf with the smallest first column
        for (size_t id = 0; id < flags.size(); id++)
            if (flags[id] && !h->leaf_robust[id] && (bad < 0 || h->nodes[id].c0 < h->nodes[bad].c0)) bad = (int)id;
        if (bad < 0 && h->pass_aborted) return fail(h, MPQR_ERR_STATE, "the block loop stopped at a flagged leaf, but no leaf flag is set");
        if (bad < 0) return fail(h, MPQR_ERR_STATE, "the flag word is raised, but no leaf flag is set");
        if (h->robust) { h->factored = true; break; }       // (every tall leaf already on the robust path: nothing left to repair)
        int first = 0;
        for (int t = 0; t < nblocks; t++) {
            const Node& tp = h->nodes[h->tops[t]];
            if (h->nodes[bad].c0 >= tp.c0 && h->nodes[bad].c0 < tp.c1) { first = t; break; }
        }
        h->leaf_robust[bad] = 1;
        start = first;
        if (pass >= 15) { h->robust = true; start = 0; }    // (16 repaired leaves: the matrix is better served by the robust kernels everywhere)
    }
    h->n_gh_leaves = gh_total;
    h->n_robust_leaves = 0;
    for (char c : h->leaf_robust) h->n_robust_leaves += c ? 1 : 0;
    h->tpoll = tpoll_saved;
    h->v_clean = false;
    if (dispatch_failed(h)) return fail(h, MPQR_ERR_STATE, "a GEMM of the factorisation found no kernel for its operand staging / epilogue, or an enqueue call failed (stderr)");
    h->v_clean = h->n_passes == 1 && !h->robust && h->n_robust_leaves == 0 && h->world == 1;      // (only a factorisation that returns MPQR_OK leaves the stores clean)
    return MPQR_OK;
}

int mpqr_sync(mpqr_handle_t h) {
    if (!h) return MPQR_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    if (h->s1) HIPCHK(h, hipStreamSynchronize(h->s1));
    if (h->sT) HIPCHK(h, hipStreamSynchronize(h->sT));
    if (dispatch_failed(h)) return fail(h, MPQR_ERR_STATE, "a GEMM enqueued for this handle found no kernel for its operand staging / epilogue");
    return MPQR_OK;
}

int mpqr_get_timings(mpqr_handle_t h, mpqr_timings* t) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (!t) return MPQR_ERR_INVALID;
    if (!h->factored) return fail(h, MPQR_ERR_STATE, "nothing has been factored");
    HIPCHK(h, hipStreamSynchronize(h->s0));
    memset(t, 0, sizeof *t);
    HIPCHK(h, hipEventElapsedTime(&t->ms_factor, h->ev[0], h->ev[1]));
    HIPCHK(h, hipEventElapsedTime(&t->ms_form_q, h->ev[1], h->ev[2]));
    t->ms_total = t->ms_factor + t->ms_form_q;
    double f = 0;
    float tr = 0;
    double fq = 0, fq_tn = 0;
    for (size_t i = 0; i + 3 < h->far_used; i += 4) {
        float a = 0, b = 0, x = 0;
        HIPCHK(h, hipEventElapsedTime(&a, h->far_ev[i], h->far_ev[i + 1]));
        HIPCHK(h, hipEventElapsedTime(&b, h->far_ev[i + 2], h->far_ev[i + 3]));
        HIPCHK(h, hipEventElapsedTime(&x, h->far_ev[i], h->far_ev[i + 3]));     // whole far update (op1..op3)
        if (i >= h->q_first) {                                                    // Q formation's applies
            t->ms_q_tn += a; t->ms_q_nn += b; t->n_q_launches++;
            fq += h->far_flops[i / 4];
            if (i / 4 < h->far_flops_tn.size()) fq_tn += h->far_flops_tn[i / 4];
            if (i / 4 < h->far_bytes.size()) t->gbytes_q_nn += h->far_bytes[i / 4] * 1e-9;
            continue;
        }
        if (i / 4 < h->far_bytes.size()) t->gbytes_far_nn += h->far_bytes[i / 4] * 1e-9;
        static const int dbg_far = []() { const char* e = getenv("MPQR_DBG_BLOCKS"); return e ? atoi(e) : 0; }();
        if (dbg_far) fprintf(stderr, "mpqr: far update %2zu: %7.1f GFLOP per GEMM, tn %7.1f us (%6.0f TFLOP/s), nn %7.1f us (%6.0f TFLOP/s), whole %7.1f us\n",
                             i / 4, h->far_flops[i / 4] * 1e-9, a * 1e3f, h->far_flops[i / 4] / (a * 1e9), b * 1e3f, h->far_flops[i / 4] / (b * 1e9), x * 1e3f);
        t->ms_far_tn += a; t->ms_far_nn += b; tr += x;
        f += h->far_flops[i / 4];
        t->n_far_launches++;
    }
    t->tflop_q = (float)(fq * 1e-12);
    t->tflop_q_tn = (float)(fq_tn * 1e-12);
    t->ms_host_enqueue = h->host_enqueue_ms;
    t->flops_far_tn = f; t->flops_far_nn = f;
    t->ms_trailing = tr;
    // the panel chain timed on its own stream: leaves, in-block updates, T merges of every top-level block (with
    // look-ahead the far updates run beside it on the second stream, so ms_panel + ms_trailing may exceed ms_factor)
    float ch = 0;
    for (size_t i = 0; i + 1 < h->chain_used; i += 2) {
        float x = 0;
        HIPCHK(h, hipEventElapsedTime(&x, h->chain_ev[i], h->chain_ev[i + 1]));
        ch += x;
        static const int dbg_blocks = []() { const char* e = getenv("MPQR_DBG_BLOCKS"); return e ? atoi(e) : 0; }();
        if (dbg_blocks) {                                  // measurement aid: chain time of every top-level block + the gap before the next one
            float gap = 0;
            if (i + 2 < h->chain_used) MPQR_IGNORE(hipEventElapsedTime(&gap, h->chain_ev[i + 1], h->chain_ev[i + 2]));
            fprintf(stderr, "mpqr: block %2zu chain %8.1f us, then %6.1f us before the next block\n", i / 2, x * 1e3f, gap * 1e3f);
        }
    }
    if (h->dbg_stamps) {                                   // measurement aid: unprofiled leaf periods (gh_solve start to start) and the time between two solves
        fprintf(stderr, "mpqr: leaf solve-start period / solve time / time to the next solve (us), 8 leaves per line:\n");
        const int nl = h->n / 128;
        for (int l = 0; l + 1 < nl; l++) {
            const unsigned long long s0_ = h->dbg_stamps[2 * l], e0 = h->dbg_stamps[2 * l + 1], s1_ = h->dbg_stamps[2 * l + 2];
            if (!s0_ || !s1_) continue;
            fprintf(stderr, "%s%5.0f/%3.0f/%4.0f", (l % 8) ? " " : "mpqr:  ", (s1_ - s0_) * 0.01, (e0 - s0_) * 0.01, (s1_ - e0) * 0.01);
            if ((l % 8) == 7) fprintf(stderr, "\n");
        }
        fprintf(stderr, "\n");
    }
    t->ms_panel = ch;
    t->ms_chain_wait = t->ms_factor - ch;
    t->n_passes = h->n_passes;
    t->n_robust_leaves = h->n_robust_leaves;
    t->n_gh_leaves = h->n_gh_leaves;
    t->us_gh_solve = h->us_gh_solve;
    t->n_q_ident_rows = h->n_q_ident_rows;
    t->restart_block = h->restart_block;
    t->n_fused_leaves = h->n_fused_leaves; t->n_tpoll_retries = h->n_tpoll_retries;
    t->n_deflated_columns = (h->hflag_host && h->flag_words >= 2) ? __atomic_load_n(h->hflag_host + h->flag_words - 2, __ATOMIC_RELAXED) : 0;
    h->last_t = *t;
    return MPQR_OK;
}

// measurement aid: the serial core of a Gram-Householder leaf (gh_solve, one workgroup) timed alone on scratch data -- an
// identity-dominated Gram matrix and a unit top block; the kernel's instruction stream does not depend on the values.
int mpqr_get_update_records(mpqr_handle_t h, int cap, double* flops, double* bytes, int* dims, int* is_q, int* n) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (!n) return MPQR_ERR_INVALID;
    const size_t nr = std::min({h->far_flops.size(), h->far_bytes.size(), h->far_dims.size() / 3, h->far_used / 4});
    *n = (int)nr;
    for (size_t i = 0; i < nr && (int)i < cap; i++) {
        if (flops) flops[i] = h->far_flops[i];
        if (bytes) bytes[i] = h->far_bytes[i];
        if (dims) { dims[3 * i] = h->far_dims[3 * i]; dims[3 * i + 1] = h->far_dims[3 * i + 1]; dims[3 * i + 2] = h->far_dims[3 * i + 2]; }
        if (is_q) is_q[i] = (4 * i >= h->q_first) ? 1 : 0;
    }
    return MPQR_OK;
}

int mpqr_bench_leaf_solve(mpqr_handle_t h, int w, int iters, float* us_per_launch) {
    if (!h || !us_per_launch || w < 1 || w > 128 || iters < 1) return MPQR_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    const int lda = 128;
    std::vector<double> G(128 * 128, 0.25);
    std::vector<float> B(128 * lda, 0.125f);
    for (int i = 0; i < 128; i++) { G[i * 128 + i] = 1000.0; B[i * lda + i] = 1.f; }
    double* dG = nullptr; float *dA = nullptr, *dCv = nullptr, *dvd = nullptr; half_t *dVh = nullptr, *dVt = nullptr; int* dfl = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = MPQR_OK;
    auto bad = [&](hipError_t e) { if (e != hipSuccess && rc == MPQR_OK) rc = fail(h, MPQR_ERR_HIP, hipGetErrorString(e)); return e != hipSuccess; };
    if (!bad(hipMalloc(&dG, G.size() * 8)) && !bad(hipMalloc(&dA, B.size() * 4)) && !bad(hipMalloc(&dCv, 128 * 128 * 4)) &&
        !bad(hipMalloc(&dvd, 128 * 4)) && !bad(hipMalloc(&dVh, 128 * lda * 2)) && !bad(hipMalloc(&dVt, 128 * lda * 2)) &&
        !bad(hipMalloc(&dfl, 4)) && !bad(hipEventCreate(&e0)) && !bad(hipEventCreate(&e1)) &&
        !bad(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice)) && !bad(hipMemset(dfl, 0, 4))) {
        LeafArgs a{};
        a.A = dA; a.lda = lda; a.mrows = 128; a.cb = 0; a.c0 = 0; a.c1 = w;
        a.Vh = dVh; a.ldvh = lda; a.Vt = dVt; a.ldvt = lda; a.vdiag = dvd;
        float ms = 0.f;
        for (int it = -2; it < iters && rc == MPQR_OK; it++) {        // two warm-up launches; the top block is an input AND an output
            if (bad(hipMemcpyAsync(dA, B.data(), B.size() * 4, hipMemcpyHostToDevice, h->s0))) break;
            if (bad(hipEventRecord(e0, h->s0))) break;
            launch_gh_solve(a, dG, dCv, dfl, h->s0);
            if (bad(hipEventRecord(e1, h->s0)) || bad(hipStreamSynchronize(h->s0))) break;
            float x = 0.f;
            if (bad(hipEventElapsedTime(&x, e0, e1))) break;
            if (it >= 0) ms += x;
        }
        if (rc == MPQR_OK) { h->us_gh_solve = ms * 1000.f / iters; *us_per_launch = h->us_gh_solve; }
    }
    if (e0) MPQR_IGNORE(hipEventDestroy(e0));
    if (e1) MPQR_IGNORE(hipEventDestroy(e1));
    MPQR_IGNORE(hipFree(dG)); MPQR_IGNORE(hipFree(dA)); MPQR_IGNORE(hipFree(dCv)); MPQR_IGNORE(hipFree(dvd)); MPQR_IGNORE(hipFree(dVh)); MPQR_IGNORE(hipFree(dVt)); MPQR_IGNORE(hipFree(dfl));
    return rc;
}

// test aid: one GEMM through a chosen kernel (include/mpqr.h).  Operands are rounded on the host / by the library's own
// quantisation kernel and zero-padded to the sizes the kernels may read (256 rows, the k step).
int mpqr_gemm_test_f32(mpqr_handle_t h, const float* A, const float* B, float* C, int M, int N, int K, int kernel, int mode) {
    if (!h || !A || !B || !C || M < 1 || N < 1 || K < 1 || (mode != 0 && mode != 2)) return MPQR_ERR_INVALID;
    if (kernel != 1 && kernel != 2 && kernel != 6 && kernel != 8 && kernel != 16) return MPQR_ERR_INVALID;
    if (kernel == 2 && mode != 0) return MPQR_ERR_INVALID;
    if (mode == 2 && (M % 32) != 0) return fail(h, MPQR_ERR_INVALID, "mpqr_gemm_test_f32: the read-modify-write epilogue needs M % 32 == 0");
    HIPCHK(h, hipSetDevice(h->device));
    const int Mp = rup(M, 256), Np = rup(N, 256), Kp = rup(K, kernel == 8 ? 128 : 64);
    std::vector<half_t> Ah((size_t)Mp * Kp, (half_t)0.f), Bt((size_t)Np * Kp, (half_t)0.f);
    std::vector<float> Af, Cp((size_t)Mp * Np, 0.f);
    for (int i = 0; i < M; i++) for (int k = 0; k < K; k++) Ah[(size_t)i * Kp + k] = (half_t)A[(size_t)i * K + k];
    for (int k = 0; k < K; k++) for (int j = 0; j < N; j++) Bt[(size_t)j * Kp + k] = (half_t)B[(size_t)k * N + j];
    if (kernel == 2) {                                       // fp32 source [K][M]: the kernel rounds it to fp16 while staging
        Af.assign((size_t)Kp * Mp, 0.f);
        for (int i = 0; i < M; i++) for (int k = 0; k < K; k++) Af[(size_t)k * Mp + i] = A[(size_t)i * K + k];
    }
    for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) Cp[(size_t)i * Np + j] = C[(size_t)i * N + j];
    half_t *dA = nullptr, *dB = nullptr; float *dAf = nullptr, *dC = nullptr; uint8_t *dA8 = nullptr, *dB8 = nullptr;
    int rc = MPQR_OK;
    auto bad = [&](hipError_t e) { if (e != hipSuccess && rc == MPQR_OK) rc = fail(h, MPQR_ERR_HIP, hipGetErrorString(e)); return e != hipSuccess; };
    do {
        if (bad(hipMalloc(&dA, Ah.size() * 2)) || bad(hipMalloc(&dB, Bt.size() * 2)) || bad(hipMalloc(&dC, Cp.size() * 4))) break;
        if (bad(hipMemcpy(dA, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice)) || bad(hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice)) ||
            bad(hipMemcpy(dC, Cp.data(), Cp.size() * 4, hipMemcpyHostToDevice))) break;
        GemmArgs g{};
        g.A = dA; g.lda = Kp; g.Bt = dB; g.ldb = Kp; g.C = dC; g.ldc = Np;
        g.M = M; g.N = N; g.K = Kp; g.in_scale = 1.f; g.alpha = 1.f; g.nsplit = 1; g.nslab_in = 1;
        const EMode em = mode == 2 ? E_SUB_F32 : E_STORE_F32;
        bool ok = true;
        if (kernel == 1) ok = launch_gemm_f16(A_H16, em, g, h->s0);
        else if (kernel == 6 || kernel == 16) ok = launch_gemm2_f16(A_H16, em, g, h->s0, kernel == 16 ? 16 : 32);   // 32: the 32x32x16 form of every epilogue
        else if (kernel == 2) {
            if (bad(hipMalloc(&dAf, Af.size() * 4)) || bad(hipMemcpy(dAf, Af.data(), Af.size() * 4, hipMemcpyHostToDevice))) break;
            g.A = dAf; g.lda = Mp;
            ok = launch_gemm2_f16(A_F32T, E_STORE_F32, g, h->s0, 0);
        } else {
            if (bad(hipMalloc(&dA8, Ah.size())) || bad(hipMalloc(&dB8, Bt.size()))) break;
            launch_quant_h16_fp8(dA, Kp, dA8, Kp, Mp, Kp, 1.f, h->s0);
            launch_quant_h16_fp8(dB, Kp, dB8, Kp, Np, Kp, 1.f, h->s0);
            g.A = dA8; g.lda = Kp; g.Bt = (const half_t*)dB8; g.ldb = Kp;
            ok = launch_gemm_fp8(em, g, h->s0);
        }
        if (!ok) { rc = fail(h, MPQR_ERR_INVALID, "mpqr_gemm_test_f32: the kernel does not take this shape / mode"); break; }
        if (bad(hipStreamSynchronize(h->s0)) || bad(hipGetLastError())) break;
        if (bad(hipMemcpy(Cp.data(), dC, Cp.size() * 4, hipMemcpyDeviceToHost))) break;
        for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) C[(size_t)i * N + j] = Cp[(size_t)i * Np + j];
    } while (0);
    MPQR_IGNORE(hipFree(dA)); MPQR_IGNORE(hipFree(dB)); MPQR_IGNORE(hipFree(dC)); MPQR_IGNORE(hipFree(dAf)); MPQR_IGNORE(hipFree(dA8)); MPQR_IGNORE(hipFree(dB8));
    return rc;
}

// measurement aid: what the matrix pipes of this device deliver on random fp16 operands (include/mpqr.h)
int mpqr_bench_mfma_peak(mpqr_handle_t h, int shape, float* tflops, float* ghz) {
    if (!h || !tflops || !ghz || (shape != 0 && shape != 1)) return MPQR_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
    const int nwg = prop.multiProcessorCount, iters = 8192, reps = 6;
    half_t* dsrc = nullptr; float* dout = nullptr; long* dclk = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = MPQR_OK;
    auto bad = [&](hipError_t e) { if (e != hipSuccess && rc == MPQR_OK) rc = fail(h, MPQR_ERR_HIP, hipGetErrorString(e)); return e != hipSuccess; };
    do {
        std::vector<half_t> hb((size_t)1 << 20);
        uint64_t x = 88172645463325252ull;
        for (auto& v : hb) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (half_t)(((float)(x >> 40) / 16777216.f - 0.5f) * 2.f); }
        if (bad(hipMalloc((void**)&dsrc, hb.size() * 2 + 4096)) || bad(hipMalloc((void**)&dout, (size_t)nwg * 512 * 4)) || bad(hipMalloc((void**)&dclk, (size_t)nwg * 16))) break;
        if (bad(hipMemcpy(dsrc, hb.data(), hb.size() * 2, hipMemcpyHostToDevice)) || bad(hipEventCreate(&e0)) || bad(hipEventCreate(&e1))) break;
        for (int r = 0; r < 3; r++) launch_mfma_peak(shape, dsrc, dout, dclk, nwg, iters, h->s0);          // let the clock settle
        if (bad(hipEventRecord(e0, h->s0))) break;
        for (int r = 0; r < reps; r++) launch_mfma_peak(shape, dsrc, dout, dclk, nwg, iters, h->s0);
        float ms = 0.f;
        if (bad(hipEventRecord(e1, h->s0)) || bad(hipStreamSynchronize(h->s0)) || bad(hipGetLastError()) || bad(hipEventElapsedTime(&ms, e0, e1))) break;
        std::vector<long> clk((size_t)2 * nwg);
        if (bad(hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost))) break;
        double g = 0;
        for (int b = 0; b < nwg; b++) g += (double)clk[2 * b] / (double)std::max<long>(clk[2 * b + 1], 1) * 0.1;   // s_memrealtime: 100 MHz
        *ghz = (float)(g / nwg);
        *tflops = (float)((double)reps * nwg * 8 * (double)iters * 2.0 * 128 * 64 * 32 / (ms * 1e-3) / 1e12);
    } while (0);
    if (e0) MPQR_IGNORE(hipEventDestroy(e0));
    if (e1) MPQR_IGNORE(hipEventDestroy(e1));
    MPQR_IGNORE(hipFree(dsrc)); MPQR_IGNORE(hipFree(dout)); MPQR_IGNORE(hipFree(dclk));
    return rc;
}

// measurement aid: one large-shape GEMM kernel alone on the device (include/mpqr.h)
int mpqr_bench_gemm(mpqr_handle_t h, int kernel, int mode, int M, int N, int K, int iters, float* ms_per_launch) {
    if (!h || !ms_per_launch || iters < 1 || M < 256 || N < 256 || K < 64 || (M % 256) || (N % 256) || (K % 64) || mode < 0 || mode > 3)
        return MPQR_ERR_INVALID;
    if (kernel != 6 && kernel != 2 && kernel != 16) return MPQR_ERR_INVALID;
    if (kernel == 2 && mode > 1) return fail(h, MPQR_ERR_INVALID, "mpqr_bench_gemm: kernel 2 has store epilogues only");
    HIPCHK(h, hipSetDevice(h->device));
    t_dispatch_handle = h;
    const size_t a_el = (size_t)(M + 256) * K, b_el = (size_t)(N + 256) * K, c_el = (size_t)M * N + 1024;
    void* dA = nullptr; half_t* dB = nullptr; float* dC = nullptr; half_t* dCt = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = MPQR_OK;
    auto bad = [&](hipError_t e) { if (e != hipSuccess && rc == MPQR_OK) rc = fail(h, MPQR_ERR_HIP, hipGetErrorString(e)); return e != hipSuccess; };
    do {
        if (bad(hipMalloc(&dA, a_el * (kernel == 2 ? 4 : 2))) || bad(hipMalloc((void**)&dB, b_el * 2)) || bad(hipMalloc((void**)&dC, c_el * 4))) break;
        if (mode == 3 && bad(hipMalloc((void**)&dCt, (size_t)(N + 256) * M * 2))) break;
        if (bad(hipEventCreate(&e0)) || bad(hipEventCreate(&e1))) break;
        // random operands (the chip holds a lower clock on random data than on zeros: tools/ubench_mfma.hip), generated by the library's own
        // U[0,1) generator into an fp32 staging matrix and narrowed by a GEMM-free path: the fp32 buffer itself for kernel 2, else halves
        {
            std::vector<half_t> hb(std::max(a_el, b_el));
            uint64_t x = 88172645463325252ull;
            for (auto& v : hb) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (half_t)(((float)(x >> 40) / 16777216.f - 0.5f) * 0.25f); }
            if (kernel == 2) {
                std::vector<float> ha(a_el);
                for (size_t i = 0; i < a_el; i++) ha[i] = (float)hb[i];
                if (bad(hipMemcpy(dA, ha.data(), a_el * 4, hipMemcpyHostToDevice))) break;
            } else if (bad(hipMemcpy(dA, hb.data(), a_el * 2, hipMemcpyHostToDevice))) break;
            if (bad(hipMemcpy(dB, hb.data(), b_el * 2, hipMemcpyHostToDevice))) break;
            if (bad(hipMemset(dC, 0, c_el * 4))) break;
        }
        GemmArgs g{};
        g.A = dA; g.lda = kernel == 2 ? M : K; g.Bt = dB; g.ldb = K; g.C = dC; g.ldc = N;
        g.M = M; g.N = N; g.K = K; g.in_scale = 1.f; g.alpha = 1.f; g.nsplit = 1; g.nslab_in = 1;
        if (mode == 3) { g.Ct = dCt; g.ldct = M; g.ct_scale = 1.f; }
        const EMode em = mode == 0 ? E_STORE_F32 : mode == 1 ? E_STORE_H16 : E_SUB_F32;
        float ms = 0.f;
        for (int it = -2; it < iters; it++) {
            if (it == 0 && bad(hipEventRecord(e0, h->s0))) break;
            const bool ok = kernel == 2 ? launch_gemm2_f16(A_F32T, em, g, h->s0, 0) : launch_gemm2_f16(A_H16, em, g, h->s0, kernel == 16 ? 16 : 32);
            if (!ok) { rc = fail(h, MPQR_ERR_INVALID, "mpqr_bench_gemm: no kernel for this (kernel, mode)"); break; }
        }
        if (rc) break;
        if (bad(hipEventRecord(e1, h->s0)) || bad(hipStreamSynchronize(h->s0)) || bad(hipGetLastError()) || bad(hipEventElapsedTime(&ms, e0, e1))) break;
        *ms_per_launch = ms / iters;
    } while (0);
    if (e0) MPQR_IGNORE(hipEventDestroy(e0));
    if (e1) MPQR_IGNORE(hipEventDestroy(e1));
    MPQR_IGNORE(hipFree(dA)); MPQR_IGNORE(hipFree(dB)); MPQR_IGNORE(hipFree(dC)); MPQR_IGNORE(hipFree(dCt));
    return rc;
}

// any Gram-Householder flag raised since the last clear?  (synchronises the chain stream)
static int any_leaf_flag(mpqr_handle_t h, int* out) {
    std::vector<int> f((size_t)h->nflag, 0);
    HIPCHK(h, hipMemcpyAsync(f.data(), h->dflag, f.size() * sizeof(int), hipMemcpyDeviceToHost, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    *out = 0;
    for (int v : f) if (v) { *out = 1; break; }
    return MPQR_OK;
}
static int clear_leaf_flags(mpqr_handle_t h) {
    HIPCHK(h, hipMemsetAsync(h->dflag, 0, (size_t)h->nflag * sizeof(int), h->s0));
    return MPQR_OK;
}

static int ensure_stage(mpqr_handle_t h, size_t elems) {
    if (h->stage_elems >= elems) return MPQR_OK;
    if (h->dstage) MPQR_IGNORE(hipFree(h->dstage));
    h->dstage = nullptr; h->stage_elems = 0;
    int rc = dalloc(h, &h->dstage, elems);
    if (rc) return rc;
    h->stage_elems = elems;
    return MPQR_OK;
}

int mpqr_get_factor_host(mpqr_handle_t h, float* A_out) {
    int rc = need_plan(h); if (rc) return rc;
    if (!A_out) return MPQR_ERR_INVALID;
    if (!h->factored) return fail(h, MPQR_ERR_STATE, "nothing has been factored");
    const size_t el = (size_t)(h->m + 1) * h->n;
    if ((rc = ensure_stage(h, el))) return rc;
    launch_pack_factor(h->dA, h->lda, h->vdiag, h->dstage, h->m, h->n, h->s0);
    HIPCHK(h, hipMemcpyAsync(A_out, h->dstage, el * sizeof(float), hipMemcpyDeviceToHost, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    return MPQR_OK;
}

int mpqr_get_r_host(mpqr_handle_t h, float* R) {
    int rc = need_plan(h); if (rc) return rc;
    if (!R) return MPQR_ERR_INVALID;
    if (!h->factored) return fail(h, MPQR_ERR_STATE, "nothing has been factored");
    const size_t el = (size_t)h->m * h->n;
    if ((rc = ensure_stage(h, el))) return rc;
    launch_strip_r(h->dA, h->lda, h->dstage, h->m, h->n, h->s0);
    HIPCHK(h, hipMemcpyAsync(R, h->dstage, el * sizeof(float), hipMemcpyDeviceToHost, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    return MPQR_OK;
}

int mpqr_get_q_host(mpqr_handle_t h, float* Q) {
    int rc = need_plan(h); if (rc) return rc;
    if (!Q) return MPQR_ERR_INVALID;
    if (!h->q_formed) return fail(h, MPQR_ERR_STATE, "Q has not been formed (opts.form_q = 0?)");
    HIPCHK(h, hipMemcpy2DAsync(Q, (size_t)h->m * sizeof(float), h->dQ, h->ldq * sizeof(float), (size_t)h->m * sizeof(float),
                               h->m, hipMemcpyDeviceToHost, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    return MPQR_OK;
}

// ||A0 - Q R||/||A0||, Q^T Q - I, strict-lower(R): exact-f32 FMA products, double reductions.
// dmetric layout: [0] sum (A-QR)^2  [1] sum A^2  [2] sum (G-I)^2  [3] float bits of max signed (G-I)  [4] sum lower(R)^2
static int metrics_core(mpqr_handle_t h, const float* A0, long lda0, const float* R, long ldr, const float* Q, long ldq,
                        int m, int n, float* work /* m*max(m,n) */, mpqr_metrics* out) {
    HIPCHK(h, hipMemsetAsync(h->dmetric, 0, 8 * sizeof(double), h->s0));
    if (A0 && R) {                                        // A0 == NULL: only the Q metrics (h_q_error, qr.cu:137-171)
        SgemmArgs g{};
        g.A = Q; g.lda = ldq; g.transA = 0; g.nslab_a = 1;
        g.B = R; g.ldb = ldr; g.transB = 0;
        g.C = work; g.ldc = n; g.M = m; g.N = n; g.K = m; g.alpha = 1.f; g.beta = 0.f;
        launch_sgemm(g, h->s0);
        launch_diff_norms(A0, lda0, work, n, m, n, h->dmetric, h->s0);
        launch_lower_norm(R, ldr, m, n, h->dmetric + 4, h->s0);
    }
    SgemmArgs q{};
    q.A = Q; q.lda = ldq; q.transA = 1; q.nslab_a = 1;
    q.B = Q; q.ldb = ldq; q.transB = 0;
    q.C = work; q.ldc = m; q.M = m; q.N = m; q.K = m; q.alpha = 1.f; q.beta = 0.f;
    launch_sgemm(q, h->s0);
    launch_gram_minus_identity(work, m, m, h->dmetric + 2, h->s0);
    double hm[8];
    HIPCHK(h, hipMemcpyAsync(hm, h->dmetric, sizeof hm, hipMemcpyDeviceToHost, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    out->a_norm = sqrt(hm[1]);
    out->backward_error = hm[1] > 0 ? sqrt(hm[0]) / sqrt(hm[1]) : 0.0;
    out->q_error_fro = sqrt(hm[2]);
    float mx; memcpy(&mx, &hm[3], 4);
    out->q_error_max_signed = mx;
    out->lower_trapezoid = sqrt(hm[4]);
    return MPQR_OK;
}

int mpqr_metrics_device(mpqr_handle_t h, mpqr_metrics* out) {
    int rc = need_plan(h); if (rc) return rc;
    if (!out) return MPQR_ERR_INVALID;
    if (!h->factored || !h->q_formed) return fail(h, MPQR_ERR_STATE, "factor with form_q=1 first");
    const int m = h->m, n = h->n;
    float* R = nullptr; float* work = nullptr;
    if ((rc = dalloc(h, &R, (size_t)m * n))) return rc;
    if ((rc = dalloc(h, &work, (size_t)m * std::max(m, n)))) { MPQR_IGNORE(hipFree(R)); return rc; }
    launch_strip_r(h->dA, h->lda, R, m, n, h->s0);
    rc = metrics_core(h, h->dA0, h->lda, R, n, h->dQ, h->ldq, m, n, work, out);
    MPQR_IGNORE(hipFree(R)); MPQR_IGNORE(hipFree(work));
    return rc;
}

int mpqr_metrics_f32(mpqr_handle_t h, const float* A, const float* R, const float* Q, int m, int n, mpqr_metrics* out) {
    if (!h || !A || !R || !Q || !out || m < 1 || n < 1) return MPQR_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    float *dA = nullptr, *dR = nullptr, *dQ = nullptr, *work = nullptr;
    int rc;
    if ((rc = dalloc(h, &dA, (size_t)m * n)) || (rc = dalloc(h, &dR, (size_t)m * n)) ||
        (rc = dalloc(h, &dQ, (size_t)m * m)) || (rc = dalloc(h, &work, (size_t)m * std::max(m, n)))) {
        if (dA) MPQR_IGNORE(hipFree(dA)); if (dR) MPQR_IGNORE(hipFree(dR)); if (dQ) MPQR_IGNORE(hipFree(dQ));
        return rc;
    }
    hipError_t e1 = hipMemcpyAsync(dA, A, (size_t)m * n * 4, hipMemcpyHostToDevice, h->s0);
    hipError_t e2 = hipMemcpyAsync(dR, R, (size_t)m * n * 4, hipMemcpyHostToDevice, h->s0);
    hipError_t e3 = hipMemcpyAsync(dQ, Q, (size_t)m * m * 4, hipMemcpyHostToDevice, h->s0);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) rc = fail(h, MPQR_ERR_HIP, "H2D copy failed");
    else rc = metrics_core(h, dA, n, dR, n, dQ, m, m, n, work, out);
    MPQR_IGNORE(hipFree(dA)); MPQR_IGNORE(hipFree(dR)); MPQR_IGNORE(hipFree(dQ)); MPQR_IGNORE(hipFree(work));
    return rc;
}

int mpqr_q_error_f32(mpqr_handle_t h, const float* Q, int m, mpqr_metrics* out) {
    if (!h || !Q || !out || m < 1) return MPQR_ERR_INVALID;
    HIPCHK(h, hipSetDevice(h->device));
    float *dQ = nullptr, *work = nullptr;
    int rc;
    if ((rc = dalloc(h, &dQ, (size_t)m * m)) || (rc = dalloc(h, &work, (size_t)m * m))) { if (dQ) MPQR_IGNORE(hipFree(dQ)); return rc; }
    if (hipMemcpyAsync(dQ, Q, (size_t)m * m * 4, hipMemcpyHostToDevice, h->s0) != hipSuccess) rc = fail(h, MPQR_ERR_HIP, "H2D copy failed");
    else rc = metrics_core(h, nullptr, 0, nullptr, 0, dQ, m, m, m, work, out);
    MPQR_IGNORE(hipFree(dQ)); MPQR_IGNORE(hipFree(work));
    return rc;
}

int mpqr_error_passes(double err, int m, int precision_bits) {
    return (err <= pow(2.0, -precision_bits) * m) ? 1 : 0;
}

// ---------------------------------------------------------------- the reference drivers
int mpqr_block_qr_f32(mpqr_handle_t h, float* A, float* Q, int m, int n, int r, const mpqr_opts* opts) {
    int rc = check_shape(h, m, n, r); if (rc) return rc;
    if (!A) return fail(h, MPQR_ERR_INVALID, "A is NULL");
    mpqr_opts o; if (opts) o = *opts; else mpqr_default_opts(&o);
    if (o.form_q && !Q) return fail(h, MPQR_ERR_INVALID, "Q is NULL but form_q is set");
    if ((rc = mpqr_plan(h, m, n, r, &o))) return rc;
    if ((rc = mpqr_set_matrix_host(h, A, n))) return rc;
    // R and the reflectors leave for the caller's buffer block row by block row while the factorisation runs (stream_rows; the input has
    // been copied to the device, A is free to be overwritten).  MPQR_STREAM_OUT=0: one copy behind the block loop, as in round 4.
    static const int stream_env = []() { const char* e = getenv("MPQR_STREAM_OUT"); return e ? atoi(e) : 1; }();
    const size_t el_out = (size_t)(h->m + 1) * h->n;
    if (stream_env && h->world == 1) {
        if (!h->sD) HIPCHK(h, hipStreamCreateWithFlags(&h->sD, hipStreamNonBlocking));
        if ((rc = ensure_stage(h, el_out))) return rc;
        if (h->ev_rows.size() < h->tops.size()) {
            const size_t have = h->ev_rows.size();
            h->ev_rows.resize(h->tops.size());
            for (size_t i = have; i < h->ev_rows.size(); i++) HIPCHK(h, hipEventCreateWithFlags(&h->ev_rows[i], hipEventDisableTiming));
        }
        h->stream_out = A; h->rows_valid = false; h->rows_streamed = 0;
    }
    rc = mpqr_factor(h);
    h->stream_out = nullptr;
    if (rc) return rc;
    const bool streamed = h->rows_valid && h->rows_streamed == h->m + 1 && h->n_passes == 1 && h->n_tpoll_retries == 0;
    h->rows_valid = false;
    if (!o.form_q) return streamed ? MPQR_OK : mpqr_get_factor_host(h, A);
    if (streamed) return mpqr_get_q_host(h, Q);
    // mpqr_factor has enqueued Q formation and returned: R and the reflectors are final since the event behind the block loop, so
    // their read-back (1 GB at 16384^2, ~19 ms of PCIe) runs on a stream of its own beside Q formation (~9 ms) instead of after it
    if (!h->sD) HIPCHK(h, hipStreamCreateWithFlags(&h->sD, hipStreamNonBlocking));
    const size_t el = (size_t)(h->m + 1) * h->n;
    if ((rc = ensure_stage(h, el))) return rc;
    HIPCHK(h, hipStreamWaitEvent(h->sD, h->ev[1], 0));
    launch_pack_factor(h->dA, h->lda, h->vdiag, h->dstage, h->m, h->n, h->sD);
    HIPCHK(h, hipMemcpyAsync(A, h->dstage, el * sizeof(float), hipMemcpyDeviceToHost, h->sD));
    HIPCHK(h, hipStreamSynchronize(h->sD));
    return mpqr_get_q_host(h, Q);
}

static int default_handle(mpqr_handle_t* out) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) { int rc = mpqr_create(&g_default, 0); if (rc) return rc; }
    *out = g_default;
    return MPQR_OK;
}

int mpqr_dev_mixed_precision_block_qr(float* A, float* Q, int m, int n, int r) {
    mpqr_handle_t h; int rc = default_handle(&h); if (rc) return rc;
    mpqr_opts o; mpqr_default_opts(&o); o.precision = MPQR_PREC_FP16;
    return mpqr_block_qr_f32(h, A, Q, m, n, r, &o);
}

int mpqr_dev_block_qr_wy(float* A, float* Q, int m, int n, int r) {
    mpqr_handle_t h; int rc = default_handle(&h); if (rc) return rc;
    mpqr_opts o; mpqr_default_opts(&o); o.precision = MPQR_PREC_FP32;
    return mpqr_block_qr_f32(h, A, Q, m, n, r, &o);
}

// ---------------------------------------------------------------- stage-level entry points
// load a host (m+1) x n buffer whose columns [c0,c1) hold shifted reflectors (others: plain data)
static int stage_load(mpqr_handle_t h, const float* A, int m, int n, int r, int c0, int c1, int precision = MPQR_PREC_FP16) {
    mpqr_opts o; mpqr_default_opts(&o); o.precision = precision;
    int rc = mpqr_plan(h, m, n, r, &o); if (rc) return rc;
    const size_t el = (size_t)(m + 1) * n;
    if ((rc = ensure_stage(h, el))) return rc;
    HIPCHK(h, hipMemsetAsync(h->dA, 0, (size_t)h->m_pad * h->lda * sizeof(float), h->s0));
    if ((rc = clear_reflectors(h))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->dstage, A, el * sizeof(float), hipMemcpyHostToDevice, h->s0));
    launch_unpack_factor(h->dstage, m, n, c0, c1, h->dA, h->lda, h->vdiag, h->Vh, h->ldvh, h->Vt, h->ldvt, h->s0);
    if (h->Vf && c1 > c0) launch_extract_vf(h->dA, h->lda, h->vdiag, h->Vf, h->n_pad, h->m, c0, c1, h->s0);
    HIPCHK(h, hipStreamSynchronize(h->s0));
    h->Aeff = h->dA; h->v8_node = -1; h->q_inited = false; h->v_clean = false;       // (the stores now hold the caller's reflectors c0..c1)
    h->factored = false; h->q_formed = false; h->have_input = false;
    return MPQR_OK;
}

// a private tree over [c0,c1) appended to the plan's node list (T storage taken from the arena tail is not
// available, so stage calls build their own small arena)
struct StageTree {
    std::vector<Node> saved_nodes; std::vector<int> saved_tops; std::vector<char> saved_robust;
    float* Tf = nullptr; half_t* Th = nullptr; half_t* Tth = nullptr;
    float *oTf = nullptr; half_t *oTh = nullptr, *oTth = nullptr;
    int root = -1;
};
static int stage_tree_begin(mpqr_handle_t h, StageTree& st, int c0, int c1, int r) {
    st.saved_nodes = h->nodes; st.saved_tops = h->tops; st.saved_robust = h->leaf_robust;
    st.oTf = h->Tf; st.oTh = h->Th; st.oTth = h->Tth;
    h->nodes.clear(); h->tops.clear(); h->leaf_robust.clear();
    h->pairs_ready = false;                               // pair nodes belong to the planned tree
    const int saved_r = h->r; h->r = r;
    st.root = build_tree(h, c0, c1);
    h->r = saved_r;
    size_t toff = 0; int max_ldt = 64;
    for (Node& nd : h->nodes) { nd.toff = toff; toff += (size_t)nd.ldt * nd.ldt; max_ldt = std::max(max_ldt, nd.ldt); }
    if ((size_t)max_ldt * max_ldt > h->tmp_elems || (size_t)64 * max_ldt * max_ldt > h->s_elems ||
        (size_t)std::max(h->m_pad, h->n_pad) * max_ldt > h->yt_elems)
        return fail(h, MPQR_ERR_INVALID, "panel too wide for the planned workspace");
    if ((int)h->nodes.size() > h->nflag) return fail(h, MPQR_ERR_INVALID, "panel tree larger than the planned flag array");
    // + 256 rows of slack, zeroed: the 256-wide GEMM tiles load Bt (= T, T^T) unmasked up to the next multiple of 256 rows
    const size_t tel = toff + (size_t)256 * max_ldt;
    int rc;
    if ((rc = dalloc(h, &st.Tf, tel)) || (rc = dalloc(h, &st.Th, tel)) || (rc = dalloc(h, &st.Tth, tel))) return rc;
    HIPCHK(h, hipMemsetAsync(st.Th, 0, tel * sizeof(half_t), h->s0));
    HIPCHK(h, hipMemsetAsync(st.Tth, 0, tel * sizeof(half_t), h->s0));
    h->Tf = st.Tf; h->Th = st.Th; h->Tth = st.Tth;
    return MPQR_OK;
}
static void stage_tree_end(mpqr_handle_t h, StageTree& st) {
    MPQR_IGNORE(hipStreamSynchronize(h->s0));
    if (st.Tf) MPQR_IGNORE(hipFree(st.Tf)); if (st.Th) MPQR_IGNORE(hipFree(st.Th)); if (st.Tth) MPQR_IGNORE(hipFree(st.Tth));
    h->Tf = st.oTf; h->Th = st.oTh; h->Tth = st.oTth;
    h->nodes = st.saved_nodes; h->tops = st.saved_tops; h->leaf_robust = st.saved_robust;
}

int mpqr_householder_qr_f32(mpqr_handle_t h, float* A, int m, int n, int go, int pw, int precision) {
    int rc = check_shape(h, m, n, 1); if (rc) return rc;
    if (!A || go < 0 || go >= n || pw < 1) return fail(h, MPQR_ERR_INVALID, "bad panel range");
    if (precision != MPQR_PREC_FP16 && precision != MPQR_PREC_FP32) return fail(h, MPQR_ERR_INVALID, "unknown precision");
    const int c0 = go, c1 = std::min(n, go + pw);           // qr.cu:210: r = min(go+pw, n)
    h->robust = false;
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = stage_load(h, A, m, n, std::max(1, c1 - c0), 0, 0, precision))) return rc;
        if ((rc = compute_scale(h, h->dA))) return rc;
        if ((rc = clear_leaf_flags(h))) return rc;
        StageTree st;
        if ((rc = stage_tree_begin(h, st, c0, c1, c1 - c0))) { stage_tree_end(h, st); return rc; }
        rc = factor_node(h, st.root, true);
        int f = 0;
        const int rcf = any_leaf_flag(h, &f);
        stage_tree_end(h, st);          // synchronises the stream
        if (rc) return rc;
        if (rcf) return rcf;
        if (!f || h->robust) break;
        h->robust = true;               // ill-conditioned tall leaf: redo on the column-by-column kernels
    }
    // write back only the panel columns, in the reference's shifted layout
    std::vector<float> tmp((size_t)m * n), vd(n);
    HIPCHK(h, hipMemcpy2D(tmp.data(), (size_t)n * 4, h->dA, h->lda * 4, (size_t)n * 4, m, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(vd.data(), h->vdiag, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (int c = c0; c < c1; c++) {
        for (int rr = 0; rr <= c && rr < m; rr++) A[(size_t)rr * n + c] = tmp[(size_t)rr * n + c];
        if (c < m) A[(size_t)(c + 1) * n + c] = vd[c];
        for (int rr = c + 1; rr < m; rr++) A[(size_t)(rr + 1) * n + c] = tmp[(size_t)rr * n + c];
    }
    // columns right of the panel inside [c0,c1) were updated in place; columns outside are untouched by the
    // reference too (h_householder_qr only touches A[k:m, k:r], qr.cu:264-280)
    return MPQR_OK;
}

int mpqr_wy_transform_f32(mpqr_handle_t h, const float* A, int m, int n, int go, int pw, float* T, float* Qpanel) {
    int rc = check_shape(h, m, n, 1); if (rc) return rc;
    if (!A || go < 0 || go >= n || pw < 1 || go + pw > n || (!T && !Qpanel)) return fail(h, MPQR_ERR_INVALID, "bad arguments");
    const int c0 = go, c1 = go + pw;
    if ((rc = stage_load(h, A, m, n, pw, c0, c1))) return rc;
    StageTree st;
    if ((rc = stage_tree_begin(h, st, c0, c1, pw))) { stage_tree_end(h, st); return rc; }
    if ((rc = factor_node(h, st.root, false))) { stage_tree_end(h, st); return rc; }
    const Node root = h->nodes[st.root];
    std::vector<float> Tpad((size_t)root.ldt * root.ldt);
    hipError_t e = hipMemcpyAsync(Tpad.data(), h->Tf + root.toff, Tpad.size() * 4, hipMemcpyDeviceToHost, h->s0);
    stage_tree_end(h, st);
    if (e != hipSuccess) return fail(h, MPQR_ERR_HIP, "D2H of T failed");
    std::vector<float> Tl((size_t)pw * pw);
    const int off = c0 - root.a0;
    for (int i = 0; i < pw; i++) for (int j = 0; j < pw; j++) Tl[(size_t)i * pw + j] = Tpad[(size_t)(off + i) * root.ldt + off + j];
    if (T) memcpy(T, Tl.data(), Tl.size() * 4);
    if (Qpanel) {
        // dense Q_panel = I - V T V^T (what the reference materialises, qr.cu:402-418); exact-f32 products on the GPU
        const int W = m - go;
        std::vector<float> V((size_t)W * pw, 0.f);
        for (int i = 0; i < W; i++) for (int j = 0; j < pw && j <= i; j++) V[(size_t)i * pw + j] = A[(size_t)(go + i + 1) * n + go + j];
        float *dV = nullptr, *dT = nullptr, *dW = nullptr, *dQp = nullptr;
        if ((rc = dalloc(h, &dV, V.size())) || (rc = dalloc(h, &dT, Tl.size())) || (rc = dalloc(h, &dW, V.size())) ||
            (rc = dalloc(h, &dQp, (size_t)W * W))) { if (dV) MPQR_IGNORE(hipFree(dV)); if (dT) MPQR_IGNORE(hipFree(dT)); if (dW) MPQR_IGNORE(hipFree(dW)); return rc; }
        HIPQ(h, hipMemcpyAsync(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice, h->s0));
        HIPQ(h, hipMemcpyAsync(dT, Tl.data(), Tl.size() * 4, hipMemcpyHostToDevice, h->s0));
        HIPQ(h, hipMemsetAsync(dQp, 0, (size_t)W * W * 4, h->s0));
        launch_set_identity(dQp, W, W, W, h->s0);
        SgemmArgs a{}; a.A = dV; a.lda = pw; a.B = dT; a.ldb = pw; a.C = dW; a.ldc = pw; a.M = W; a.N = pw; a.K = pw;
        a.alpha = 1.f; a.beta = 0.f; a.nslab_a = 1;
        launch_sgemm(a, h->s0);
        SgemmArgs b{}; b.A = dW; b.lda = pw; b.B = dV; b.ldb = pw; b.transB = 1; b.C = dQp; b.ldc = W; b.M = W; b.N = W; b.K = pw;
        b.alpha = -1.f; b.beta = 1.f; b.nslab_a = 1;
        launch_sgemm(b, h->s0);
        hipError_t e2 = hipMemcpyAsync(Qpanel, dQp, (size_t)W * W * 4, hipMemcpyDeviceToHost, h->s0);
        MPQR_IGNORE(hipStreamSynchronize(h->s0));
        MPQR_IGNORE(hipFree(dV)); MPQR_IGNORE(hipFree(dT)); MPQR_IGNORE(hipFree(dW)); MPQR_IGNORE(hipFree(dQp));
        if (e2 != hipSuccess) return fail(h, MPQR_ERR_HIP, "D2H of Q_panel failed");
    }
    return MPQR_OK;
}

int mpqr_apply_panel_to_trailing_f32(mpqr_handle_t h, float* A, int m, int n, int go, int pw, int precision) {
    int rc = check_shape(h, m, n, 1); if (rc) return rc;
    if (!A || go < 0 || pw < 1 || go + pw > n) return fail(h, MPQR_ERR_INVALID, "bad panel range");
    if (precision != MPQR_PREC_FP16 && precision != MPQR_PREC_FP32 && precision != MPQR_PREC_FP8) return fail(h, MPQR_ERR_INVALID, "unknown precision");
    const int c0 = go, c1 = go + pw;
    if ((rc = stage_load(h, A, m, n, pw, c0, c1, precision))) return rc;
    if ((rc = compute_scale(h, h->dA))) return rc;
    StageTree st;
    if ((rc = stage_tree_begin(h, st, c0, c1, pw))) { stage_tree_end(h, st); return rc; }
    if ((rc = factor_node(h, st.root, false))) { stage_tree_end(h, st); return rc; }
    apply_node(h, h->nodes[st.root], h->dA, h->lda, c1, n, true, h->a_scale, false, 0, true);
    stage_tree_end(h, st);
    if (c1 < n) {
        // only A[go:m, tau:n] changes (Cuda/qr.cu:1098-1106)
        HIPCHK(h, hipMemcpy2D(A + (size_t)go * n + c1, (size_t)n * 4, h->dA + (size_t)go * h->lda + c1, h->lda * 4,
                              (size_t)(n - c1) * 4, m - go, hipMemcpyDeviceToHost));
    }
    return MPQR_OK;
}

int mpqr_q_backward_accumulation_f32(mpqr_handle_t h, const float* A, float* Q, int m, int n, int precision) {
    int rc = check_shape(h, m, n, 1); if (rc) return rc;
    if (!A || !Q) return fail(h, MPQR_ERR_INVALID, "NULL argument");
    if (precision != MPQR_PREC_FP16 && precision != MPQR_PREC_FP32) return fail(h, MPQR_ERR_INVALID, "unknown precision");
    const int r = std::min(n, 128);
    if ((rc = stage_load(h, A, m, n, r, 0, n, precision))) return rc;
    for (size_t t = 0; t < h->tops.size(); t++) if ((rc = factor_node(h, h->tops[t], false))) return rc;
    if ((rc = form_q(h))) return rc;
    return mpqr_get_q_host(h, Q);
}

// ---------------------------------------------------------------- C++/main.cpp path
// ---------------------------------------------------------------------------- least squares
namespace {
// dB (m_pad x ldb, fp32, device) <- Q^T dB from the stored reflectors, block by block (the trailing-update operator)
int apply_qt_device(mpqr_handle_t h, float* dB, long ldb, int nrhs) {
    // power-of-two scale that keeps the fp16 operand in range, as for the matrix itself
    launch_absmax(dB, ldb, h->m, nrhs, h->dscalar, h->s0);
    float mx = 0.f;
    HIPCHK(h, hipMemcpyAsync(&mx, h->dscalar, sizeof(float), hipMemcpyDeviceToHost, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    float sc = 1.f;
    if (mx > 0.f && std::isfinite(mx)) { int e; frexpf(mx * sqrtf((float)h->m), &e); sc = ldexpf(1.f, 8 - e); }
    for (size_t t = 0; t < h->tops.size(); t++) apply_node(h, h->nodes[h->tops[t]], dB, ldb, 0, nrhs, true, sc, false, 0);
    return MPQR_OK;
}
// dB[0:n] <- R^-1 dB[0:n]: blocked back substitution from the last diagonal block up (R = upper part of the factor)
int back_substitute_device(mpqr_handle_t h, float* dB, long ldb, int nrhs) {
    const int n = h->n;
    for (int k1 = n; k1 > 0;) {
        const int k0 = ((k1 - 1) / 128) * 128, kb = k1 - k0;
        launch_trsm_diag(h->dA, h->lda, k0, kb, dB, ldb, nrhs, h->s0);
        if (k0 > 0) {                                         // Y[0:k0] -= R[0:k0, k0:k1] X[k0:k1], exact-f32 MFMA
            SgemmArgs u{};
            u.A = h->dA + k0; u.lda = h->lda; u.transA = 0; u.nslab_a = 1;
            u.B = dB + (long)k0 * ldb; u.ldb = ldb; u.transB = 0;
            u.C = dB; u.ldc = ldb; u.M = k0; u.N = nrhs; u.K = kb; u.alpha = -1.f; u.beta = 1.f;
            launch_sgemm(u, h->s0);
        }
        k1 = k0;
    }
    HIPCHK(h, hipGetLastError());
    return MPQR_OK;
}
int ls_core(mpqr_handle_t h, const float* B, int nrhs, float* out, bool solve) {
    int rc = need_plan(h); if (rc) return rc;
    if (!h->factored) return fail(h, MPQR_ERR_STATE, "factor first");
    if (h->world != 1) return fail(h, MPQR_ERR_INVALID, "single-GPU handles only");
    if (h->opts.precision == MPQR_PREC_FP32) return fail(h, MPQR_ERR_INVALID, "MPQR_PREC_FP16 handles only");
    if (!B || !out || nrhs < 1) return fail(h, MPQR_ERR_INVALID, "bad arguments");
    if (nrhs > std::max(h->m_pad, h->n_pad))      // Xt / Yt scratch holds max(m_pad, n_pad) right-hand sides
        return fail(h, MPQR_ERR_INVALID, "too many right-hand sides for the planned workspace (nrhs <= max(m, n) rounded up to 256)");
    const long ldb = ((long)nrhs + 31) / 32 * 32;
    float* dB = nullptr;
    if ((rc = dalloc(h, &dB, (size_t)(h->m_pad + 256) * ldb))) return rc;
    hipError_t e0 = hipMemsetAsync(dB, 0, (size_t)(h->m_pad + 256) * ldb * sizeof(float), h->s0);
    hipError_t e1 = hipMemcpy2DAsync(dB, ldb * sizeof(float), B, (size_t)nrhs * sizeof(float), (size_t)nrhs * sizeof(float),
                                     h->m, hipMemcpyHostToDevice, h->s0);
    if (e0 != hipSuccess || e1 != hipSuccess) { MPQR_IGNORE(hipFree(dB)); return fail(h, MPQR_ERR_HIP, "copy-in failed"); }
    rc = apply_qt_device(h, dB, ldb, nrhs);
    if (!rc && solve) rc = back_substitute_device(h, dB, ldb, nrhs);
    if (!rc) {
        const int rows = solve ? h->n : h->m;
        hipError_t e2 = hipMemcpy2DAsync(out, (size_t)nrhs * sizeof(float), dB, ldb * sizeof(float), (size_t)nrhs * sizeof(float),
                                         rows, hipMemcpyDeviceToHost, h->s0);
        hipError_t e3 = hipStreamSynchronize(h->s0);
        if (e2 != hipSuccess || e3 != hipSuccess) rc = fail(h, MPQR_ERR_HIP, "copy-out failed");
    }
    MPQR_IGNORE(hipStreamSynchronize(h->s0));
    MPQR_IGNORE(hipFree(dB));
    return rc;
}
}  // namespace

int mpqr_apply_qt_host(mpqr_handle_t h, float* B, int nrhs) { return ls_core(h, B, nrhs, B, false); }
int mpqr_solve_ls_host(mpqr_handle_t h, const float* B, int nrhs, float* X) { return ls_core(h, B, nrhs, X, true); }

int mpqr_qr_solver_f32(mpqr_handle_t h, const float* A, const float* b, float* x, int m, int n, int r) {
    int rc = check_shape(h, m, n, r); if (rc) return rc;
    if (!A || !b || !x) return fail(h, MPQR_ERR_INVALID, "null buffer");
    mpqr_opts o; mpqr_default_opts(&o);
    o.form_q = 0;                                             // Q is applied implicitly
    if ((rc = mpqr_plan(h, m, n, r, &o))) return rc;
    if ((rc = mpqr_set_matrix_host(h, A, n))) return rc;
    if ((rc = mpqr_factor(h))) return rc;
    return mpqr_solve_ls_host(h, b, 1, x);
}

int mpqr_dev_qr_solver(const float* A, const float* b, float* x, int m, int n) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) { int rc = mpqr_create(&g_default, 0); if (rc) return rc; }
    return mpqr_qr_solver_f32(g_default, A, b, x, m, n, 128);
}

int mpqr_qr_factorization_f64(mpqr_handle_t h, double* A, double* Q, int m, int n) {
    int rc = check_shape(h, m, n, 1); if (rc) return rc;
    if (!A || !Q) return fail(h, MPQR_ERR_INVALID, "NULL argument");
    HIPCHK(h, hipSetDevice(h->device));
    double *dA = nullptr, *dQ = nullptr, *dw = nullptr;
    if ((rc = dalloc(h, &dA, (size_t)m * n)) || (rc = dalloc(h, &dQ, (size_t)m * m)) || (rc = dalloc(h, &dw, (size_t)m))) {
        if (dA) MPQR_IGNORE(hipFree(dA)); if (dQ) MPQR_IGNORE(hipFree(dQ));
        return rc;
    }
    std::vector<double> I((size_t)m * m, 0.0);
    for (int i = 0; i < m; i++) I[(size_t)i * m + i] = 1.0;     // reference requires Q = identity on entry
    hipError_t e1 = hipMemcpyAsync(dA, A, (size_t)m * n * 8, hipMemcpyHostToDevice, h->s0);
    hipError_t e2 = hipMemcpyAsync(dQ, I.data(), (size_t)m * m * 8, hipMemcpyHostToDevice, h->s0);
    launch_qr_f64(dA, dQ, m, n, dw, h->s0);
    hipError_t e3 = hipMemcpyAsync(A, dA, (size_t)m * n * 8, hipMemcpyDeviceToHost, h->s0);
    hipError_t e4 = hipMemcpyAsync(Q, dQ, (size_t)m * m * 8, hipMemcpyDeviceToHost, h->s0);
    hipError_t e5 = hipStreamSynchronize(h->s0);
    MPQR_IGNORE(hipFree(dA)); MPQR_IGNORE(hipFree(dQ)); MPQR_IGNORE(hipFree(dw));
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess)
        return fail(h, MPQR_ERR_HIP, "fp64 path: HIP call failed");
    return MPQR_OK;
}

}  // extern "C"

// =====================================================================================
// Multi-GPU: 1-D block-cyclic columns, one process per GPU.  The reference is single-GPU; this is
// the sharded form of the same block loop.  Column superblocks (outer_block columns, the unit the far update
// works with) are dealt round-robin; every rank holds all m rows of its columns and the same column tree.
//   step s : owner factors superblock s            mpqr_dist_factor_block
//            owner packs [V^T | T | T^T] (fp16)    mpqr_dist_pack_block     -> caller broadcasts the buffer (RCCL)
//            everyone unpacks + updates its own trailing columns            mpqr_dist_unpack_block / _update
//   Q      : columns of Q sharded the same way; each rank applies all (V,T) to its columns, no communication.
// The broadcast itself stays outside the library (torch.distributed / RCCL in bench.py and dist.py), so the
// same schedule can be exercised on CPU ranks with a test double.
extern "C" {

int mpqr_dist_plan(mpqr_handle_t h, int m, int n, int r, int world, int rank, const mpqr_opts* opts) {
    return plan_common(h, m, n, r, opts, world, rank);
}
int mpqr_dist_local_cols(mpqr_handle_t h) { return (h && h->planned) ? h->nloc : -1; }
int mpqr_dist_local_q_cols(mpqr_handle_t h) { return (h && h->planned) ? h->qloc : -1; }
int mpqr_dist_block(mpqr_handle_t h) { return (h && h->planned) ? h->Ko : -1; }
int mpqr_dist_num_blocks(mpqr_handle_t h) { return (h && h->planned) ? (int)h->tops.size() : -1; }
int mpqr_dist_block_owner(mpqr_handle_t h, int s) {
    if (!h || !h->planned || s < 0 || s >= (int)h->tops.size()) return -1;
    return s % h->world;
}

int mpqr_dist_set_local_matrix_host(mpqr_handle_t h, const float* A_local, long ld) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (h->nloc > 0) {
        if (!A_local || ld < h->nloc) return fail(h, MPQR_ERR_INVALID, "bad local matrix / leading dimension");
        HIPCHK(h, hipMemcpy2DAsync(h->dA0, h->lda * sizeof(float), A_local, ld * sizeof(float),
                                   (size_t)h->nloc * sizeof(float), h->m, hipMemcpyHostToDevice, h->s0));
    }
    HIPCHK(h, hipStreamSynchronize(h->s0));
    h->factored = false; h->q_formed = false; h->have_input = true; h->robust = false;
    return MPQR_OK;
}

int mpqr_dist_generate_matrix(mpqr_handle_t h, uint64_t seed) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (h->nloc > 0) launch_generate(h->dA0, h->lda, h->m, h->nloc, seed, h->n, h->Ko, h->world, h->rank, h->s0);
    h->factored = false; h->q_formed = false; h->have_input = true; h->robust = false;
    return MPQR_OK;
}

int mpqr_dist_local_absmax(mpqr_handle_t h, float* out) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (!out) return MPQR_ERR_INVALID;
    *out = 0.f;
    if (h->nloc > 0) {
        launch_absmax(h->dA0, h->lda, h->m, h->nloc, h->dscalar, h->s0);
        HIPCHK(h, hipMemcpyAsync(out, h->dscalar, sizeof(float), hipMemcpyDeviceToHost, h->s0));
    }
    HIPCHK(h, hipStreamSynchronize(h->s0));
    return MPQR_OK;
}

// copy the input into the working matrix and reset the reflector storage; absmax = GLOBAL max |a_ij|
int mpqr_dist_begin(mpqr_handle_t h, float absmax) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (!h->have_input) return fail(h, MPQR_ERR_STATE, "no input matrix has been set");
    float sc = 1.f;
    if (absmax > 0.f && std::isfinite(absmax)) { int e; frexpf(absmax * sqrtf((float)h->m), &e); sc = ldexpf(1.f, 8 - e); }
    h->a_scale = sc;
    h->far_used = 0; h->far_flops.clear(); h->far_bytes.clear(); h->far_dims.clear(); h->far_flops_tn.clear(); h->chain_used = 0; h->v8_node = -1;
    h->pairs_ready = false; h->n_passes = 1; h->n_robust_leaves = 0; h->n_gh_leaves = 0;
    HIPCHK(h, hipEventRecord(h->ev[0], h->s0));
    HIPCHK(h, hipMemcpyAsync(h->dA, h->dA0, (size_t)h->m_pad * h->lda * sizeof(float), hipMemcpyDeviceToDevice, h->s0));
    if (h->Xt1) {                                   // far stream starts behind the copy-in; its "done" event starts signalled
        HIPCHK(h, hipEventRecord(h->ev_dist_chain, h->s0));
        HIPCHK(h, hipStreamWaitEvent(h->s1, h->ev_dist_chain, 0));
        HIPCHK(h, hipEventRecord(h->ev_dist_far, h->s1));
    }
    if ((rc = clear_leaf_flags(h))) return rc;
    for (int b = 0; b < h->flag_words; b++) __atomic_store_n(h->hflag_host + b, 0, __ATOMIC_RELAXED);   // (the previous factorisation was synchronised: mpqr_dist_flags)
    h->cur_block = 0; h->watch_flags = false;
    if ((rc = clear_reflectors(h))) return rc;
    h->v_clean = false;                                     // (this rank's stores will hold received panels)
    h->factored = false; h->q_formed = false;
    return MPQR_OK;
}

// owner only: factor superblock s of the (already updated) local columns.  Nothing is synchronised: an ill-conditioned tall leaf raises
// its flag word (mapped host memory) and the factorisation carries on; the host asks ONCE, after the last block (mpqr_dist_flags), and a
// flagged factorisation is repeated with every tall leaf on the column-by-column kernels (mpqr_dist_set_robust) -- rare, and the same on
// every rank.  (Rounds 1-3 read the flags back after every block: one host synchronisation per block on the owner, 16 per factorisation,
// each of them a bubble in front of the next broadcast; the forced-N=1 path ran 12 % behind the single-GPU path for it.)
int mpqr_dist_factor_block(mpqr_handle_t h, int s) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (s < 0 || s >= (int)h->tops.size()) return fail(h, MPQR_ERR_INVALID, "bad block index");
    if (s % h->world != h->rank) return fail(h, MPQR_ERR_STATE, "this rank does not own that block");
    const Node nd = h->nodes[h->tops[s]];
    const int lc0 = mpqr_part_local_index(nd.c0, h->Ko, h->world);
    h->Aeff = h->dA + (lc0 - nd.c0);
    const bool timed = h->chain_used + 2 <= h->chain_ev.size();
    if (timed) HIPCHK(h, hipEventRecord(h->chain_ev[h->chain_used], h->s0));
    rc = factor_node(h, h->tops[s], true);
    if (timed) { HIPCHK(h, hipEventRecord(h->chain_ev[h->chain_used + 1], h->s0)); h->chain_used += 2; }
    h->Aeff = h->dA;
    if (rc) return rc;
    HIPCHK(h, hipGetLastError());
    return MPQR_OK;
}

// after mpqr_sync: did any Gram-Householder leaf of this rank's blocks flag itself since mpqr_dist_begin?
int mpqr_dist_flags(mpqr_handle_t h, int* any) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (!any) return MPQR_ERR_INVALID;
    HIPCHK(h, hipStreamSynchronize(h->s0));
    if (h->hflag_host && h->flag_words > 0 && __atomic_load_n(h->hflag_host + h->flag_words - 1, __ATOMIC_RELAXED) != 0)
        return fail(h, MPQR_ERR_HIP, "the T stream's wait for the chain stream timed out (wait_flag_kernel): this pass's results are invalid");
    *any = flag_words_set(h, h->flag_words - 2) ? 1 : 0;
    return MPQR_OK;
}
// on != 0: every tall leaf of the following factorisations takes the column-by-column kernels (all ranks must agree)
int mpqr_dist_set_robust(mpqr_handle_t h, int on) {
    int rc = need_plan(h, true); if (rc) return rc;
    h->robust = on != 0;
    return MPQR_OK;
}

// bytes of the broadcast buffer of block s: V^T rows (fp16, columns >= the block's first 64-aligned row) | T | T^T
long mpqr_dist_block_bytes(mpqr_handle_t h, int s) {
    if (!h || !h->planned || s < 0 || s >= (int)h->tops.size()) return -1;
    const Node& nd = h->nodes[h->tops[s]];
    const long Kr = nd.ldt, Wc = h->m_pad - rdown(nd.c0, 64);
    return (Kr * Wc + 2 * Kr * Kr) * (long)sizeof(half_t) + Kr * Kr * (long)sizeof(float);   // V^T | T | T^T (fp16) | T (fp32: pair merges)
}

int mpqr_dist_pack_block_async(mpqr_handle_t h, int s, void* dbuf) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (s < 0 || s >= (int)h->tops.size() || !dbuf) return fail(h, MPQR_ERR_INVALID, "bad arguments");
    const Node& nd = h->nodes[h->tops[s]];
    const long Kr = nd.ldt; const int rlo = rdown(nd.c0, 64); const long Wc = h->m_pad - rlo;
    half_t* out = (half_t*)dbuf;
    HIPCHK(h, hipMemcpy2DAsync(out, Wc * sizeof(half_t), h->Vt + (size_t)nd.a0 * h->ldvt + rlo, h->ldvt * sizeof(half_t),
                               Wc * sizeof(half_t), Kr, hipMemcpyDeviceToDevice, h->s0));
    HIPCHK(h, hipMemcpyAsync(out + Kr * Wc, h->Th + nd.toff, Kr * Kr * sizeof(half_t), hipMemcpyDeviceToDevice, h->s0));
    HIPCHK(h, hipMemcpyAsync(out + Kr * Wc + Kr * Kr, h->Tth + nd.toff, Kr * Kr * sizeof(half_t), hipMemcpyDeviceToDevice, h->s0));
    HIPCHK(h, hipMemcpyAsync(out + Kr * Wc + 2 * Kr * Kr, h->Tf + nd.toff, Kr * Kr * sizeof(float), hipMemcpyDeviceToDevice, h->s0));
    return MPQR_OK;
}
int mpqr_dist_pack_block(mpqr_handle_t h, int s, void* dbuf) {
    int rc = mpqr_dist_pack_block_async(h, s, dbuf); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->s0));
    return MPQR_OK;
}
// the HIP stream the pack / unpack copies are enqueued on (hipStream_t): a host that orders its broadcasts with events
// (hipStreamWaitEvent) instead of synchronising uses the _async forms
int mpqr_dist_chain_stream(mpqr_handle_t h, void** stream) {
    if (!h || !stream) return MPQR_ERR_INVALID;
    *stream = (void*)h->s0;
    return MPQR_OK;
}

int mpqr_dist_unpack_block_async(mpqr_handle_t h, int s, const void* dbuf) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (s < 0 || s >= (int)h->tops.size() || !dbuf) return fail(h, MPQR_ERR_INVALID, "bad arguments");
    if (s % h->world == h->rank) return MPQR_OK;            // the owner already holds everything
    const Node& nd = h->nodes[h->tops[s]];
    const long Kr = nd.ldt; const int rlo = rdown(nd.c0, 64); const long Wc = h->m_pad - rlo;
    const half_t* in = (const half_t*)dbuf;
    HIPCHK(h, hipMemcpy2DAsync(h->Vt + (size_t)nd.a0 * h->ldvt + rlo, h->ldvt * sizeof(half_t), in, Wc * sizeof(half_t),
                               Wc * sizeof(half_t), Kr, hipMemcpyDeviceToDevice, h->s0));
    HIPCHK(h, hipMemcpyAsync(h->Th + nd.toff, in + Kr * Wc, Kr * Kr * sizeof(half_t), hipMemcpyDeviceToDevice, h->s0));
    HIPCHK(h, hipMemcpyAsync(h->Tth + nd.toff, in + Kr * Wc + Kr * Kr, Kr * Kr * sizeof(half_t), hipMemcpyDeviceToDevice, h->s0));
    HIPCHK(h, hipMemcpyAsync(h->Tf + nd.toff, in + Kr * Wc + 2 * Kr * Kr, Kr * Kr * sizeof(float), hipMemcpyDeviceToDevice, h->s0));
    // Vh[rlo + c][a0 + k] = Vt[a0 + k][rlo + c]
    launch_transpose_h16(in, Wc, h->Vh + (size_t)rlo * h->ldvh + nd.a0, h->ldvh, (int)Kr, (int)Wc, h->s0);
    return MPQR_OK;
}
int mpqr_dist_unpack_block(mpqr_handle_t h, int s, const void* dbuf) {
    int rc = mpqr_dist_unpack_block_async(h, s, dbuf); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->s0));              // the caller may refill the buffer (double-buffered broadcasts)
    return MPQR_OK;
}

// apply block s's reflectors to this rank's columns right of it.
//   part 0: only the columns of block s+1 (non-empty on its owner), on the CHAIN stream: the owner then factors block s+1
//   part 1: the local columns right of block s+1, on the FAR stream: runs beside that factorisation (look-ahead)
//   part 2: everything on the chain stream (no look-ahead)
int mpqr_dist_update_part(mpqr_handle_t h, int s, int part) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (s < 0 || s >= (int)h->tops.size() || part < 0 || part > 2) return fail(h, MPQR_ERR_INVALID, "bad block index / part");
    const Node nd = h->nodes[h->tops[s]];
    const bool last = s + 1 == (int)h->tops.size();
    const int c_next = last ? h->n : h->nodes[h->tops[s + 1]].c1;                         // global end of block s+1
    const int lc0 = mpqr_part_local_cols(std::min(h->n, nd.c1), h->Ko, h->world, h->rank);      // local columns left of nd.c1
    const int lc1 = mpqr_part_local_cols(std::min(h->n, c_next), h->Ko, h->world, h->rank);     // ... left of the end of block s+1
    const bool two_streams = h->Xt1 != nullptr;
    if (part == 2 || !two_streams) {
        if (part == 0) { /* covered by part 1 below */ }
        else {
            if (two_streams) HIPQ(h, hipStreamWaitEvent(h->s0, h->ev_dist_far, 0));         // earlier far-stream parts first
            apply_node(h, nd, h->dA, h->lda, lc0, h->nloc, true, h->a_scale, true, 0, true);
        }
    } else if (part == 0) {
        // block s+1's columns carry every earlier update once the previous far-stream part is done
        HIPQ(h, hipStreamWaitEvent(h->s0, h->ev_dist_far, 0));
        apply_node(h, nd, h->dA, h->lda, lc0, lc1, true, h->a_scale, false, 0, true);
    } else {
        HIPQ(h, hipEventRecord(h->ev_dist_chain, h->s0));                                   // the block's V, T (unpack / factor) are on the chain stream
        HIPQ(h, hipStreamWaitEvent(h->s1, h->ev_dist_chain, 0));
        apply_node(h, nd, h->dA, h->lda, lc1, h->nloc, true, h->a_scale, true, 1, true);
        HIPQ(h, hipEventRecord(h->ev_dist_far, h->s1));
    }
    if (part != 0 && h->opts.form_q && h->S2 && s < (int)h->qpair.size() && h->qpair[s] >= 0) {
        // Q formation works on pairs of blocks: T of the pair (s-1, s), behind this rank's update with block s
        hipStream_t ms = (part == 1 && two_streams) ? h->s1 : h->s0;
        if (ms == h->s1) {                                 // V, T of block s are on the chain stream (factor / unpack)
            HIPQ(h, hipEventRecord(h->ev_dist_chain, h->s0)); HIPQ(h, hipStreamWaitEvent(h->s1, h->ev_dist_chain, 0));
        }
        merge_pair(h, h->qpair[s], ms);
        if (ms == h->s1) HIPQ(h, hipEventRecord(h->ev_dist_far, h->s1));
    }
    if (part != 0 && h->opts.form_q && h->S2 && h->qroot >= 0 && s < (int)h->qmerge_after.size() && !h->qmerge_after[s].empty()) {
        // tall matrices: T of blocks 0..s from T of blocks 0..s-1 (every rank builds the whole T itself: it holds every block's V and T),
        // for the one-shot Q formation of its column shard (mpqr_dist_form_q)
        hipStream_t ms = (part == 1 && two_streams) ? h->s1 : h->s0;
        if (ms == h->s1) { HIPQ(h, hipEventRecord(h->ev_dist_chain, h->s0)); HIPQ(h, hipStreamWaitEvent(h->s1, h->ev_dist_chain, 0)); }
        for (int id : h->qmerge_after[s]) merge_prefix(h, id, ms);
        if (ms == h->s1) HIPQ(h, hipEventRecord(h->ev_dist_far, h->s1));
    }
    if (last && part != 0) {
        h->pairs_ready = h->opts.form_q && h->S2 != nullptr;
        if (two_streams) HIPQ(h, hipStreamWaitEvent(h->s0, h->ev_dist_far, 0));
        HIPCHK(h, hipEventRecord(h->ev[1], h->s0)); h->factored = true;
    }
    HIPCHK(h, hipGetLastError());
    if (dispatch_failed(h)) return fail(h, MPQR_ERR_STATE, "a GEMM of the trailing update found no kernel for its operand staging / epilogue");
    return MPQR_OK;
}
int mpqr_dist_update(mpqr_handle_t h, int s) { return mpqr_dist_update_part(h, s, 2); }

int mpqr_dist_form_q(mpqr_handle_t h) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (!h->factored) return fail(h, MPQR_ERR_STATE, "factor first");
    if (h->Xt1) HIPQ(h, hipStreamWaitEvent(h->s0, h->ev_dist_far, 0));
    HIPCHK(h, hipMemsetAsync(h->dQ, 0, (size_t)h->m_pad * h->ldq * sizeof(float), h->s0));
    launch_identity_cyclic(h->dQ, h->ldq, h->m, h->qloc, h->Ko, h->world, h->rank, h->s0);
    if (h->Qt) {                                           // transposed fp16 shadow of the local columns (as in form_q)
        HIPCHK(h, hipMemsetAsync(h->Qt, 0, (size_t)(h->ldq + 256) * h->ldqt * sizeof(half_t), h->s0));
        launch_identity_cyclic_h16(h->Qt, h->ldqt, h->m, h->qloc, h->Ko, h->world, h->rank, h->s0);
        h->shadow = h->Qt; h->ldshadow = h->ldqt;
    }
    if (h->qroot >= 0 && h->pairs_ready) {
        // Tall matrices (m >= 3n), as on one GPU: Q = I - (V T) V^T over all reflectors at once instead of the backward accumulation.
        // Every rank forms W = V T itself (m x n, 4 ms at 65536 x 8192) and then, for each of its column superblocks J of Q,
        // Q[:, J] -= W V[J, :]^T into the identity it already holds (read-modify-write: every GEMM kernel of the library has that epilogue).
        const Node rt = h->nodes[h->qroot];
        const int Kr = rt.ldt;
        if (h->tq_on && rt.id < (int)h->ev_T.size()) HIPQ(h, hipStreamWaitEvent(h->s0, h->ev_T[rt.id], 0));
        GemmArgs w{};                                       // W[m x Kr] = V T   (T upper triangular: k <= n)
        w.A = h->Vh + rt.a0; w.lda = h->ldvh;
        w.Bt = h->Tth + rt.toff; w.ldb = rt.tld;
        w.C = h->Wh; w.ldc = h->n_pad;
        w.M = h->m_pad; w.N = Kr; w.K = Kr; w.alpha = 1.f; w.in_scale = 1.f; w.nsplit = 1; w.tri = 2;
        w.cscale = h->Tf + rt.toff; w.cscale_ld = (long)rt.tld + 1;
        gemm_dispatch(A_H16, E_STORE_H16, w, h->s0);
        for (int l0 = 0; l0 < h->qloc; l0 += h->Ko) {
            const int gc0 = mpqr_part_global_index(l0, h->Ko, h->world, h->rank);      // global column of Q = row of V
            const int nc = std::min(h->Ko, h->qloc - l0);
            GemmArgs q{};
            q.A = h->Wh; q.lda = h->n_pad;
            q.Bt = h->Vh + (long)gc0 * h->ldvh + rt.a0; q.ldb = h->ldvh;
            q.C = h->dQ + l0; q.ldc = h->ldq;
            q.M = h->m_pad; q.N = nc; q.K = Kr; q.alpha = 1.f; q.in_scale = 1.f; q.nsplit = 1;   // (padded rows: W is zero there)
            gemm_dispatch(A_H16, E_SUB_F32, q, h->s0);
        }
        h->shadow = nullptr; h->shadow_write = true;
        h->q_formed = true;
        HIPCHK(h, hipEventRecord(h->ev[2], h->s0));
        HIPCHK(h, hipGetLastError());
        if (dispatch_failed(h)) return fail(h, MPQR_ERR_STATE, "a GEMM of the Q formation found no kernel for its operand staging / epilogue");
        return MPQR_OK;
    }
    const bool pairs = h->pairs_ready && h->qpair.size() == h->tops.size();
    for (int t = (int)h->tops.size() - 1; t >= 0; t--) {
        const bool pr = pairs && h->qpair[t] >= 0;          // two blocks at once, K = 2 outer blocks
        const Node& nd = pr ? h->nodes[h->qpair[t]] : h->nodes[h->tops[t]];
        const int lq = mpqr_part_local_cols(std::min(h->m, nd.c0), h->Ko, h->world, h->rank);
        h->shadow_write = (pr ? t - 1 : t) > 0;
        apply_node(h, nd, h->dQ, h->ldq, lq, h->qloc, false, 1.f, false);
        if (pr) t--;
    }
    h->shadow = nullptr; h->shadow_write = true;
    h->q_formed = true;
    HIPCHK(h, hipEventRecord(h->ev[2], h->s0));
    HIPCHK(h, hipGetLastError());
    if (dispatch_failed(h)) return fail(h, MPQR_ERR_STATE, "a GEMM of the Q formation found no kernel for its operand staging / epilogue");
    return MPQR_OK;
}

int mpqr_dist_get_local_factor_host(mpqr_handle_t h, float* A_local) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (!h->factored) return fail(h, MPQR_ERR_STATE, "nothing has been factored");
    if (h->nloc == 0) return MPQR_OK;
    if (!A_local) return MPQR_ERR_INVALID;
    const size_t el = (size_t)(h->m + 1) * h->nloc;
    if ((rc = ensure_stage(h, el))) return rc;
    launch_pack_factor_cyclic(h->dA, h->lda, h->vdiag, h->dstage, h->m, h->nloc, h->Ko, h->world, h->rank, h->s0);
    HIPCHK(h, hipMemcpyAsync(A_local, h->dstage, el * sizeof(float), hipMemcpyDeviceToHost, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    return MPQR_OK;
}

int mpqr_dist_get_local_q_host(mpqr_handle_t h, float* Q_local) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (!h->q_formed) return fail(h, MPQR_ERR_STATE, "Q has not been formed");
    if (h->qloc == 0) return MPQR_OK;
    if (!Q_local) return MPQR_ERR_INVALID;
    HIPCHK(h, hipMemcpy2DAsync(Q_local, (size_t)h->qloc * sizeof(float), h->dQ, h->ldq * sizeof(float),
                               (size_t)h->qloc * sizeof(float), h->m, hipMemcpyDeviceToHost, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    return MPQR_OK;
}

int mpqr_dist_get_local_input_host(mpqr_handle_t h, float* A_local) {
    int rc = need_plan(h, true); if (rc) return rc;
    if (!h->have_input) return fail(h, MPQR_ERR_STATE, "no input");
    if (h->nloc == 0) return MPQR_OK;
    if (!A_local) return MPQR_ERR_INVALID;
    HIPCHK(h, hipMemcpy2DAsync(A_local, (size_t)h->nloc * sizeof(float), h->dA0, h->lda * sizeof(float),
                               (size_t)h->nloc * sizeof(float), h->m, hipMemcpyDeviceToHost, h->s0));
    HIPCHK(h, hipStreamSynchronize(h->s0));
    return MPQR_OK;
}

}  // extern "C"
