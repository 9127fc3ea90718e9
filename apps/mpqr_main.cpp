// mpqr_main.cpp -- C++ entry point with the run list of the reference's Cuda/main.cu:11-26, on the MI355X path.
//
//   test_qr_by_random_matrix(f)  : the 20 fixed (m, n, r) shapes of Cuda/qr.cu:1762-1783, U[0,1) input
//                                  (fixed seed instead of srand(time(0)), Cuda/mmult.cuh:43-44)
//   test_qr(f)                   : Euroc Jacobian files A_%09d.txt, r = 16 (Cuda/qr.cu:1794-1804), if a directory is given
// for f in { householder + backward accumulation, fp32 block QR, mixed-precision block QR }, printing the three
// error lines and appending the reference's CSV log rows (log/cpu_householder.txt, log/gpu_block.txt).
//
//   usage: mpqr_main [--jacobians DIR [--skip-random]] [--m M --n N --r R] [--seed S] [--dtype fp16|fp32|fp8] [--gpus N [--steps K] [--no-check]]
//   --gpus N (with --m --n --r): the multi-GPU host driver -- one host thread per GPU of this node, column superblocks dealt
//            round-robin (mpqr_dist_* step functions of the C ABI), V,T of every block broadcast with ncclBroadcast (RCCL over
//            xGMI) on a communication stream of its own, look-ahead schedule of SURVEY.md 8e.  The reference is single-GPU.
// build: make -C apps    (g++, links ../mixedprecisionblockqr_amd/libmpqr.so, librccl, libamdhip64)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../include/mpqr_reference_api.hpp"

static uint64_t g_seed = 1234;

static float* generate(int m, int n) {
    float* A = (float*)malloc((size_t)m * n * sizeof(float));
    mpqr_generate_matrix_host(A, m, n, g_seed++);
    return A;
}

template <typename F>
static void run_case(const char* title, const char* logname, int bits, int m, int n, int r, float* A_in, F qr) {
    printf("\nTesting %s...\nDimensions of A (m, n, r): (%d, %d, %d)\n", title, m, n, r);
    std::vector<float> Q((size_t)m * m, 0.f), R((size_t)m * n), A_out((size_t)(m + 1) * n, 0.f);
    for (int i = 0; i < m; i++) Q[(size_t)i * m + i] = 1.f;                       // h_identity_mtx (qr.cu:1873)
    memcpy(A_out.data(), A_in, (size_t)m * n * sizeof(float));                    // h_matrix_cpy  (qr.cu:1875)
    auto t0 = std::chrono::high_resolution_clock::now();
    qr(A_out.data(), Q.data(), m, n, r);
    auto t1 = std::chrono::high_resolution_clock::now();
    const float ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    const float flops = h_qr_flops_per_second(ms, m, n);
    h_strip_R_from_A(A_out.data(), R.data(), m, n);
    const float be = h_backward_error(A_in, R.data(), Q.data(), m, n, bits);
    h_q_error(Q.data(), m, bits);
    h_lower_trapezoid_error(R.data(), m, n, bits);
    h_write_results_to_log(m, n, ms, flops / 1e9f, be * 1e8f, logname);           // GFLOP/s and error x 1e8 (qr.cu:1896-1898)
    printf("%s finished in %.2f ms, averaged %.2f GFLOPs (incl. host<->device copies, reference formula)\n", title, ms, flops / 1e9f);
    free(A_in);                                                                    // testers own A_in (qr.cu:1907)
}

static void test_h_householder_qr(int m, int n, int r, float* A_in) {
    run_case("GPU householder QR + backward accumulation", "cpu_householder", 23, m, n, r, A_in,
             [](float* A, float* Q, int m_, int n_, int) {
                 h_householder_qr(A, m_, n_, 0, n_);
                 float* Qn = nullptr; h_q_backward_accumulation(A, &Qn, m_, n_);
                 memcpy(Q, Qn, (size_t)m_ * m_ * sizeof(float)); free(Qn);
             });
}
static void test_dev_block_qr(int m, int n, int r, float* A_in) {
    run_case("GPU block QR", "gpu_block", 23, m, n, r, A_in, [](float* A, float* Q, int m_, int n_, int r_) { dev_block_qr_wy(A, Q, m_, n_, r_); });
}
static void test_dev_mixed_precision_block_qr(int m, int n, int r, float* A_in) {
    run_case("GPU mixed-precision block QR", "gpu_block", 11, m, n, r, A_in,
             [](float* A, float* Q, int m_, int n_, int r_) { dev_mixed_precision_block_qr(A, Q, m_, n_, r_); });
}

static void test_dev_fp8_block_qr(int m, int n, int r, float* A_in) {
    // fp8 operands: 4 significant bits -> criterion 2^-4 * m (the reference's form, qr.cu:120, with the operand precision)
    run_case("GPU fp8 block QR (BASELINE config 5 arithmetic)", "gpu_block", 4, m, n, r, A_in,
             [](float* A, float* Q, int m_, int n_, int r_) {
                 mpqr_opts o; mpqr_default_opts(&o); o.precision = MPQR_PREC_FP8;
                 mpqr_ref::check(mpqr_block_qr_f32(mpqr_ref::handle(), A, Q, m_, n_, r_, &o), "fp8 block QR");
             });
}

static void test_qr_by_random_matrix(QR_FUNC f) {
    static const QRProblemSize dims[20] = {{6, 4, 2}, {6, 4, 1}, {6, 4, 3}, {12, 8, 4}, {12, 8, 5}, {12, 8, 6}, {12, 8, 2},
        {12, 8, 8}, {12, 8, 3}, {24, 16, 8}, {24, 16, 12}, {60, 40, 8}, {60, 40, 16}, {80, 80, 16}, {97, 90, 16},
        {100, 80, 16}, {128, 80, 16}, {129, 80, 16}, {240, 160, 16}, {600, 400, 16}};
    for (const auto& d : dims) f(d.m, d.n, d.r, generate(d.m, d.n));
}

static void test_qr(QR_FUNC f, const char* dir) {
    std::vector<MatrixInfo> list;
    for (int i = 100; i <= 22500; i += 100) {
        char name[512]; snprintf(name, sizeof name, "%s/A_%09d.txt", dir, i);
        FILE* fp = fopen(name, "r");
        if (!fp) continue;
        MatrixInfo mi; mi.filePath = name;
        if (fscanf(fp, "%d %d", &mi.m, &mi.n) == 2) list.push_back(mi);
        fclose(fp);
    }
    std::sort(list.begin(), list.end(), [](const MatrixInfo& a, const MatrixInfo& b) { return a.m < b.m; });
    int used = 0;
    for (size_t i = 0; i < list.size() && used < 30; i += 2, used++) {            // every 2nd, at most 30 (qr.cu:1752-1757)
        int m, n; float* A_in;
        read_euroc_jacobian(list[i].filePath, &m, &n, &A_in);
        if (m < n) { printf("skipping %s: m < n\n", list[i].filePath.c_str()); free(A_in); continue; }
        f(m, n, 16, A_in);
    }
    if (list.empty()) printf("no Jacobian files under %s (the reference's data blob is a git-LFS pointer)\n", dir);
}

// ---------------------------------------------------------------- multi-GPU host driver (RCCL)
namespace {
// all rank threads meet here (absmax exchange, start / stop of the timed region).  abort(): a rank that failed wakes everybody,
// wait() then returns false and every rank leaves through its one exit path -- nobody is left blocked in a barrier or a collective
struct HostBarrier {
    std::mutex mu; std::condition_variable cv; int count = 0, gen = 0, n; bool aborted = false;
    explicit HostBarrier(int n_) : n(n_) {}
    bool wait() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const int g = gen;
        if (++count == n) { count = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g || aborted; });
        return !aborted;
    }
    void abort() { std::lock_guard<std::mutex> lk(mu); aborted = true; cv.notify_all(); }
};
}  // namespace

static int run_multi_gpu(int world, int m, int n, int r, uint64_t seed, int steps, const char* dtype, bool check) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < world) { fprintf(stderr, "error: %d GPUs requested, %d visible\n", world, ndev); return 1; }
    std::vector<int> devs(world);
    for (int i = 0; i < world; i++) devs[i] = i;
    std::vector<ncclComm_t> comms(world);
    if (ncclCommInitAll(comms.data(), world, devs.data()) != ncclSuccess) { fprintf(stderr, "error: ncclCommInitAll failed\n"); return 1; }
    const bool comm_on = world > 1 || getenv("MPQR_MG_FORCE_BCAST") != nullptr;   // the env switch exercises RCCL on a 1-GPU box
    HostBarrier bar(world);
    std::vector<float> amax(world, 0.f);
    std::vector<int> flagged(world, 0);
    std::vector<double> ms(world, 0.0);
    std::atomic<bool> failed{false};
    // --check (default): every rank hands its column shards of the input, of the factor and of Q to the host after the timed loop;
    // the three reference criteria (Cuda/qr.cu:115-196) are then evaluated on the assembled matrices like in the single-GPU testers
    std::vector<float> Ain, Fac, Qfull;
    if (check) { Ain.assign((size_t)m * n, 0.f); Fac.assign((size_t)(m + 1) * n, 0.f); Qfull.assign((size_t)m * m, 0.f); }
    mpqr_opts o; mpqr_default_opts(&o);
    o.precision = !strcmp(dtype, "fp32") ? MPQR_PREC_FP32 : (!strcmp(dtype, "fp8") ? MPQR_PREC_FP8 : MPQR_PREC_FP16);
    if (world > 1) o.outer_block = std::max(r, std::min(1024, (n / (2 * world)) / r * r));   // >= 2 blocks per rank (see dist.py)
    auto rank_main = [&](int rank) {
        mpqr_handle_t h = nullptr;
        hipStream_t cs = nullptr, chain = nullptr;                      // communication stream: broadcasts only; the library's chain stream
        hipEvent_t ev_pack = nullptr, ev_bcast = nullptr, ev_unp[2] = {nullptr, nullptr};
        void* bufs[2] = {nullptr, nullptr}; long cap[2] = {0, 0};
        // first failure: message, flag, wake the barrier waiters, abort the communicators (a peer may sit inside a broadcast)
        auto fail = [&](const char* what, const char* why) {
            fprintf(stderr, "rank %d: %s failed: %s\n", rank, what, why);
            if (!failed.exchange(true)) { bar.abort(); for (auto& c : comms) (void)ncclCommAbort(c); }
            return false;
        };
        auto ok_q = [&](int rc, const char* what) { return rc == MPQR_OK ? !failed.load() : fail(what, mpqr_last_error(h)); };
        auto ok_h = [&](hipError_t e, const char* what) { return e == hipSuccess ? !failed.load() : fail(what, hipGetErrorString(e)); };
        auto ok_n = [&](ncclResult_t e, const char* what) { return e == ncclSuccess ? !failed.load() : fail(what, ncclGetErrorString(e)); };
        auto buffer = [&](int s) -> void* {
            const long need = mpqr_dist_block_bytes(h, s);
            if (cap[s & 1] < need) {
                if (bufs[s & 1]) { (void)hipFree(bufs[s & 1]); bufs[s & 1] = nullptr; cap[s & 1] = 0; }
                if (hipMalloc(&bufs[s & 1], (size_t)need) != hipSuccess) { fail("hipMalloc(payload)", "out of memory"); return nullptr; }
                cap[s & 1] = need;
            }
            return bufs[s & 1];
        };
        int nb = 0;
        // one factorisation.  The host never waits for a broadcast: pack -> ev_pack -> the communication stream waits; broadcast ->
        // ev_bcast -> the chain stream waits -> unpack -> ev_unp[buffer] -> the next broadcast into that buffer waits.
        auto factor_once = [&]() -> bool {
            if (!ok_q(mpqr_dist_local_absmax(h, &amax[rank]), "mpqr_dist_local_absmax")) return false;
            if (!bar.wait()) return false;
            float gmax = 0.f; for (float v : amax) gmax = std::max(gmax, v);
            if (!bar.wait()) return false;
            if (!ok_q(mpqr_dist_begin(h, gmax), "mpqr_dist_begin")) return false;
            auto pack = [&](int s) -> bool {
                void* b = buffer(s);
                return b && ok_q(mpqr_dist_pack_block_async(h, s, b), "mpqr_dist_pack_block_async") && ok_h(hipEventRecord(ev_pack, chain), "hipEventRecord");
            };
            if (mpqr_dist_block_owner(h, 0) == rank) { if (!ok_q(mpqr_dist_factor_block(h, 0), "mpqr_dist_factor_block") || (comm_on && !pack(0))) return false; }
            for (int s = 0; s < nb; s++) {
                const int owner = mpqr_dist_block_owner(h, s);
                if (comm_on) {
                    void* b = buffer(s);
                    if (!b) return false;
                    if (owner == rank && !ok_h(hipStreamWaitEvent(cs, ev_pack, 0), "hipStreamWaitEvent")) return false;
                    if (s >= 2 && !ok_h(hipStreamWaitEvent(cs, ev_unp[s & 1], 0), "hipStreamWaitEvent")) return false;     // the buffer is free again
                    if (!ok_n(ncclBroadcast(b, b, (size_t)mpqr_dist_block_bytes(h, s), ncclUint8, owner, comms[rank], cs), "ncclBroadcast")) return false;
                    if (!ok_h(hipEventRecord(ev_bcast, cs), "hipEventRecord") || !ok_h(hipStreamWaitEvent(chain, ev_bcast, 0), "hipStreamWaitEvent")) return false;
                    if (!ok_q(mpqr_dist_unpack_block_async(h, s, b), "mpqr_dist_unpack_block_async") || !ok_h(hipEventRecord(ev_unp[s & 1], chain), "hipEventRecord")) return false;
                }
                if (s + 1 < nb && mpqr_dist_block_owner(h, s + 1) == rank) {     // look-ahead: my block first, the rest beside its factorisation
                    if (!ok_q(mpqr_dist_update_part(h, s, 0), "mpqr_dist_update_part") || !ok_q(mpqr_dist_update_part(h, s, 1), "mpqr_dist_update_part") ||
                        !ok_q(mpqr_dist_factor_block(h, s + 1), "mpqr_dist_factor_block") || (comm_on && !pack(s + 1))) return false;
                } else if (!ok_q(mpqr_dist_update_part(h, s, 1), "mpqr_dist_update_part")) return false;
            }
            return ok_q(mpqr_dist_form_q(h), "mpqr_dist_form_q") && ok_h(hipStreamSynchronize(cs), "hipStreamSynchronize") && ok_q(mpqr_sync(h), "mpqr_sync");
        };
        // the leaf flags are asked once, after the block loop (no host synchronisation per block); any rank's flag makes every rank
        // repeat the factorisation with its tall leaves on the column-by-column kernels (as mixedprecisionblockqr_amd/dist.py::factor)
        auto factor = [&]() -> bool {
            if (!factor_once()) return false;
            if (!ok_q(mpqr_dist_flags(h, &flagged[rank]), "mpqr_dist_flags") || !bar.wait()) return false;
            bool any = false; for (int v : flagged) any = any || v != 0;
            if (!bar.wait()) return false;
            if (!any) return true;
            return ok_q(mpqr_dist_set_robust(h, 1), "mpqr_dist_set_robust") && factor_once();
        };
        do {                                                             // one exit path: everything below the loop is released once
            if (!ok_h(hipSetDevice(rank), "hipSetDevice") || !ok_q(mpqr_create(&h, rank), "mpqr_create") ||
                !ok_q(mpqr_dist_plan(h, m, n, r, world, rank, &o), "mpqr_dist_plan") ||
                !ok_h(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking), "hipStreamCreateWithFlags") ||
                !ok_q(mpqr_dist_chain_stream(h, (void**)&chain), "mpqr_dist_chain_stream") ||
                !ok_h(hipEventCreateWithFlags(&ev_pack, hipEventDisableTiming), "hipEventCreate") || !ok_h(hipEventCreateWithFlags(&ev_bcast, hipEventDisableTiming), "hipEventCreate") ||
                !ok_h(hipEventCreateWithFlags(&ev_unp[0], hipEventDisableTiming), "hipEventCreate") || !ok_h(hipEventCreateWithFlags(&ev_unp[1], hipEventDisableTiming), "hipEventCreate") ||
                !ok_q(mpqr_dist_generate_matrix(h, seed), "mpqr_dist_generate_matrix")) break;
            nb = mpqr_dist_num_blocks(h);
            if (!factor()) break;                                        // warm-up
            if (!bar.wait()) break;
            const auto t0 = std::chrono::high_resolution_clock::now();
            bool good = true;
            for (int it = 0; it < steps && good; it++) good = factor();
            if (!good || !bar.wait()) break;
            ms[rank] = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count() / steps;
            if (check) {                                                 // shards -> their global columns of the host copies (disjoint per rank)
                const int ko = mpqr_dist_block(h), nloc = mpqr_dist_local_cols(h), qloc = mpqr_dist_local_q_cols(h);
                std::vector<float> a((size_t)m * std::max(nloc, 1)), f((size_t)(m + 1) * std::max(nloc, 1)), q((size_t)m * std::max(qloc, 1));
                if (!ok_q(mpqr_dist_get_local_input_host(h, a.data()), "mpqr_dist_get_local_input_host") ||
                    !ok_q(mpqr_dist_get_local_factor_host(h, f.data()), "mpqr_dist_get_local_factor_host") ||
                    !ok_q(mpqr_dist_get_local_q_host(h, q.data()), "mpqr_dist_get_local_q_host")) break;
                for (int lc = 0; lc < nloc; lc++) {
                    const int gc = mpqr_part_global_index(lc, ko, world, rank);
                    for (int i = 0; i < m; i++) Ain[(size_t)i * n + gc] = a[(size_t)i * nloc + lc];
                    for (int i = 0; i <= m; i++) Fac[(size_t)i * n + gc] = f[(size_t)i * nloc + lc];
                }
                for (int lc = 0; lc < qloc; lc++) {
                    const int gc = mpqr_part_global_index(lc, ko, world, rank);
                    for (int i = 0; i < m; i++) Qfull[(size_t)i * m + gc] = q[(size_t)i * qloc + lc];
                }
            }
        } while (0);
        for (void* b : bufs) if (b) (void)hipFree(b);
        for (hipEvent_t e : {ev_pack, ev_bcast, ev_unp[0], ev_unp[1]}) if (e) (void)hipEventDestroy(e);
        if (cs) (void)hipStreamDestroy(cs);
        if (h) (void)mpqr_destroy(h);
    };
    std::vector<std::thread> th;
    for (int rk = 0; rk < world; rk++) th.emplace_back(rank_main, rk);
    for (auto& t : th) t.join();
    if (!failed) for (auto& c : comms) (void)ncclCommDestroy(c);         // (aborted communicators are gone already)
    if (failed) { fprintf(stderr, "multi-GPU run failed\n"); return 1; }
    double worst = 0; for (double v : ms) worst = std::max(worst, v);
    const double geqrf = 2.0 * m * (double)n * n - 2.0 / 3.0 * (double)n * n * n;
    printf("multi-GPU block QR: %d GPU(s), %d x %d, r = %d, outer block %d, %s: %.2f ms per factorisation incl. Q, %.1f GFLOP/s (GEQRF-equivalent flops)\n",
           world, m, n, r, o.outer_block ? o.outer_block : 1024, dtype, worst, geqrf / (worst * 1e-3) / 1e9);
    if (check) {
        // the reference's three criteria on the gathered result with the reference's own two precisions (qr.cu:1367: p = 23 for the fp32
        // paths, qr.cu:1889: p = 11 for the mixed path -- the e4m3 far updates are a mixed path and are held to the same p = 11)
        const int bits = !strcmp(dtype, "fp32") ? 23 : 11;
        std::vector<float> R((size_t)m * n);
        h_strip_R_from_A(Fac.data(), R.data(), m, n);
        printf("Dimensions of A (m, n, r): (%d, %d, %d)\n", m, n, r);
        h_backward_error(Ain.data(), R.data(), Qfull.data(), m, n, bits);
        h_q_error(Qfull.data(), m, bits);
        h_lower_trapezoid_error(R.data(), m, n, bits);
    }
    return 0;
}

int main(int argc, char** argv) {
    const char* jac = nullptr; int m = 0, n = 0, r = 0, gpus = 0, steps = 3; const char* dtype = "fp16"; bool random_list = true, check = true;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--jacobians") && i + 1 < argc) jac = argv[++i];
        else if (!strcmp(argv[i], "--m") && i + 1 < argc) m = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--n") && i + 1 < argc) n = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--r") && i + 1 < argc) r = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) g_seed = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dtype") && i + 1 < argc) dtype = argv[++i];
        else if (!strcmp(argv[i], "--no-check")) check = false;               // --gpus: skip the gather + the three error criteria
        else if (!strcmp(argv[i], "--skip-random")) random_list = false;      // with --jacobians: only the Jacobian run list (qr.cu:1794-1804)
        else { fprintf(stderr, "usage: %s [--jacobians DIR [--skip-random]] [--m M --n N --r R] [--seed S] [--dtype fp16|fp32|fp8] [--gpus N [--steps K] [--no-check]]\n", argv[0]); return 2; }
    }
    try {
        if (gpus > 0) {
            if (!(m > 0 && n > 0 && r > 0)) { fprintf(stderr, "--gpus needs --m --n --r\n"); return 2; }
            return run_multi_gpu(gpus, m, n, r, g_seed, std::max(1, steps), dtype, check);
        }
        if (m > 0 && n > 0 && r > 0) {
            if (!strcmp(dtype, "fp32")) test_dev_block_qr(m, n, r, generate(m, n));
            else if (!strcmp(dtype, "fp8")) test_dev_fp8_block_qr(m, n, r, generate(m, n));
            else test_dev_mixed_precision_block_qr(m, n, r, generate(m, n));
            return 0;
        }
        if (random_list) {
            test_qr_by_random_matrix(test_h_householder_qr);
            test_qr_by_random_matrix(test_dev_block_qr);
            test_qr_by_random_matrix(test_dev_mixed_precision_block_qr);
        }
        if (jac) { test_qr(test_h_householder_qr, jac); test_qr(test_dev_block_qr, jac); test_qr(test_dev_mixed_precision_block_qr, jac); }
    } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
