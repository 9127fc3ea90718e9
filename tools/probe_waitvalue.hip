// Probe: cost of a cross-stream hand-off through (a) hipEventRecord + hipStreamWaitEvent and (b) a counter the producer
// kernel bumps + hipStreamWaitValue64 on the consumer stream.  Chain stream s0: [solve-like 100 us] -> producer -> consumer2;
// side stream s1: consumer1 (needs producer) ; s0's consumer2 needs consumer1.  Reports the time per iteration.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_waitvalue.hip -o gpurun_out/probe_waitvalue
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin_kernel(long cycles) { const long t0 = clock64(); while (clock64() - t0 < cycles) {} }
__global__ void work_kernel(float* p, unsigned long long* counter) {
    p[blockIdx.x * blockDim.x + threadIdx.x] += 1.f;
    if (counter && threadIdx.x == 0) { __threadfence(); atomicAdd(counter, 1ULL); }
}
int main() {
    int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    float* buf; CK(hipMalloc(&buf, 256 * 256 * 4)); CK(hipMemset(buf, 0, 256 * 256 * 4));
    unsigned long long *cA = nullptr, *cX = nullptr;
    CK(hipExtMallocWithFlags((void**)&cA, 8, hipMallocSignalMemory)); CK(hipExtMallocWithFlags((void**)&cX, 8, hipMallocSignalMemory));
    hipEvent_t ev, ex; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ex, hipEventDisableTiming));
    const int iters = 200, wgs = 128;
    for (int mode = 0; mode < 3; mode++) {          // 0: single stream (no hand-off), 1: events, 2: wait-value
        CK(hipMemset(cA, 0, 8)); CK(hipMemset(cX, 0, 8)); CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        unsigned long long nA = 0, nX = 0;
        for (int i = 0; i < iters; i++) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s0, 200000L);                 // ~85 us
            hipLaunchKernelGGL(work_kernel, dim3(wgs), dim3(256), 0, s0, buf, mode == 2 ? cA : nullptr);   // producer
            if (mode == 0) { hipLaunchKernelGGL(work_kernel, dim3(wgs), dim3(256), 0, s0, buf, (unsigned long long*)nullptr); }
            else if (mode == 1) {
                CK(hipEventRecord(ev, s0)); CK(hipStreamWaitEvent(s1, ev, 0));
                hipLaunchKernelGGL(work_kernel, dim3(wgs), dim3(256), 0, s1, buf, (unsigned long long*)nullptr);
                CK(hipEventRecord(ex, s1)); CK(hipStreamWaitEvent(s0, ex, 0));
            } else {
                nA += wgs; CK(hipStreamWaitValue64(s1, cA, nA, hipStreamWaitValueGte, 0xffffffffffffffffULL));
                hipLaunchKernelGGL(work_kernel, dim3(wgs), dim3(256), 0, s1, buf, cX);
                nX += wgs; CK(hipStreamWaitValue64(s0, cX, nX, hipStreamWaitValueGte, 0xffffffffffffffffULL));
            }
            hipLaunchKernelGGL(work_kernel, dim3(wgs), dim3(256), 0, s0, buf, (unsigned long long*)nullptr);       // consumer2
        }
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        printf("mode %d: %.1f us per iteration\n", mode, us);
    }
    return 0;
}
