// tools/probe_graph_handoff.hip -- what does a cross-stream hand-off cost on the critical path, eager vs. captured into a hipGraph?
// The panel chain pays two hand-offs per leaf (chain -> side stream after gh_apply, side stream -> chain before leaf_xt).  Pattern:
//   s0: A(k) -> [event] -> s1: B(k) -> [event] -> s0: A(k+1) ...      (every kernel ~5 us, one workgroup)
// against the same 2 N kernels on ONE stream.  The difference per kernel pair is the price of two hand-offs.
// Development tool; build: hipcc --offload-arch=gfx950 -O3 -o probe_graph_handoff.bin probe_graph_handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>

__global__ void spin_kernel(long* out, int cycles) {
    const long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) out[0] = t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int N = 200, cyc = 10000;          // ~5 us per kernel at 2.1 GHz
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    long* d; CK(hipMalloc(&d, 64));
    hipEvent_t ea[N], eb[N];
    for (int i = 0; i < N; i++) { CK(hipEventCreateWithFlags(&ea[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eb[i], hipEventDisableTiming)); }
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    auto one_stream = [&]() { for (int i = 0; i < 2 * N; i++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, d, cyc); };
    auto two_streams = [&]() {
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, d, cyc);
            (void)hipEventRecord(ea[i], s0); (void)hipStreamWaitEvent(s1, ea[i], 0);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, d, cyc);
            (void)hipEventRecord(eb[i], s1); (void)hipStreamWaitEvent(s0, eb[i], 0);
        }
    };
    auto timed = [&](const char* name, auto fn) {
        fn(); (void)hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; r++) {
            (void)hipEventRecord(t0, s0); fn(); (void)hipEventRecord(t1, s0); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, t0, t1); best = ms < best ? ms : best;
        }
        printf("%-34s %8.2f us per kernel pair\n", name, best * 1e3f / N);
        return best;
    };
    // fork / join with real concurrency: s0: A, C, D   s1: B beside C.   ideal per iteration: A + max(B, C) + D = 3 kernels; serial: 4
    auto fork_join = [&]() {
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, d, cyc);                       // A
            (void)hipEventRecord(ea[i], s0); (void)hipStreamWaitEvent(s1, ea[i], 0);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, d + 1, cyc);                   // B (side)
            (void)hipEventRecord(eb[i], s1);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, d + 2, cyc);                   // C (chain, beside B)
            (void)hipStreamWaitEvent(s0, eb[i], 0);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, d + 3, cyc);                   // D (needs B and C)
        }
    };
    auto four_serial = [&]() { for (int i = 0; i < 4 * N; i++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, d, cyc); };
    // what does an event RECORD cost the recording stream?  (a) nobody waits for it; (b) a second stream waits and runs its own kernel
    auto record_only = [&]() {
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, d, cyc);
            (void)hipEventRecord(ea[i], s0);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, d, cyc);
        }
    };
    auto record_and_side_waiter = [&]() {
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, d, cyc);
            (void)hipEventRecord(ea[i], s0); (void)hipStreamWaitEvent(s1, ea[i], 0);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, d + 1, cyc / 2);          // side work nobody on s0 waits for
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, d, cyc);
        }
    };
    const float e1 = timed("eager, one stream", one_stream);
    const float r1 = timed("eager, one stream + record per pair", record_only);
    const float r2 = timed("eager, record + side stream waits", record_and_side_waiter);
    printf("event record on the chain stream: %.2f us (nobody waits), %.2f us (a side stream waits for it)\n", (r1 - e1) * 1e3f / N, (r2 - e1) * 1e3f / N);
    const float e2 = timed("eager, two streams + 2 hand-offs", two_streams);
    // the same two patterns captured into graphs (origin stream s0; s1 joins through the event waits and joins back at the end)
    hipGraph_t g1, g2; hipGraphExec_t x1, x2;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal)); one_stream(); CK(hipStreamEndCapture(s0, &g1));
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal)); two_streams(); CK(hipStreamEndCapture(s0, &g2));
    CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0)); CK(hipGraphInstantiate(&x2, g2, nullptr, nullptr, 0));
    const float g1t = timed("graph, one stream", [&]() { (void)hipGraphLaunch(x1, s0); });
    const float g2t = timed("graph, two streams + 2 hand-offs", [&]() { (void)hipGraphLaunch(x2, s0); });
    printf("two hand-offs cost: eager %.2f us, graph %.2f us per pair\n", (e2 - e1) * 1e3f / N, (g2t - g1t) * 1e3f / N);
    {
        const float f0 = timed("eager, 4 kernels serial / iteration", four_serial);
        const float f1 = timed("eager, fork-join (ideal = 3 kernels)", fork_join);
        hipGraph_t g3; hipGraphExec_t x3;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal)); fork_join(); CK(hipStreamEndCapture(s0, &g3));
        CK(hipGraphInstantiate(&x3, g3, nullptr, nullptr, 0));
        const float f2 = timed("graph, fork-join (ideal = 3 kernels)", [&]() { (void)hipGraphLaunch(x3, s0); });
        printf("fork-join per iteration: serial %.2f, eager %.2f, graph %.2f us (one kernel = %.2f us)\n", f0 * 1e3f / N, f1 * 1e3f / N, f2 * 1e3f / N, f0 * 1e3f / N / 4);
    }
    auto h0 = std::chrono::steady_clock::now(); two_streams(); auto h1 = std::chrono::steady_clock::now(); (void)hipDeviceSynchronize();
    printf("host time to enqueue the eager two-stream pattern: %.2f us per pair\n", std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
    return 0;
}
