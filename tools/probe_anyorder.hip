// tools/probe_anyorder.hip -- does hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) let a kernel start while the previous kernel of
// the SAME stream is still running on this device / runtime?  K1 spins ~200 us on one workgroup, K2 (any-order or normal) records when
// it started; a third, normal kernel records when it started (does it wait for both?).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void spin(long long* t, int slot, long long ticks) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) t[2 * slot] = t0;
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) t[2 * slot + 1] = wall_clock64();
}
int main() {
    long long* d; CK(hipMalloc(&d, 64 * 8));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));   // kHz
    const long long us200 = (long long)rate * 200 / 1000, us50 = (long long)rate * 50 / 1000;
    for (int mode = 0; mode < 2; mode++) {
        CK(hipMemset(d, 0, 64 * 8));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 0, us200);
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(200), dim3(256), 0, st, d, 1, us50);
        else hipExtLaunchKernelGGL(spin, dim3(200), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, d, 1, us50);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, d, 2, us50 / 10);
        CK(hipStreamSynchronize(st));
        long long h[6]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        auto us = [&](long long x) { return (double)(x - h[0]) * 1000.0 / rate; };
        printf("%s: K1 %.1f..%.1f us, K2 start %.1f end %.1f, K3 start %.1f  -> K2 %s K1\n", mode ? "any-order" : "normal   ",
               us(h[0]), us(h[1]), us(h[2]), us(h[3]), us(h[4]), h[2] < h[1] ? "OVERLAPS" : "waits for");
    }
    return 0;
}
