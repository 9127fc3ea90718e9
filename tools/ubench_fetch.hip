// tools/ubench_fetch.hip -- is a long straight-line VALU stream fetch-bound on gfx950?  One wave per SIMD, each with its own
// 16 KiB body (2048 x 8-byte v_fma_f64), executed 20 times; compared with a 512-byte loop body.  Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))
#define R1024(x) R4(R256(x))
#define BODY4 "v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3\n"
__global__ void k_long(long* out, double* sink, int nwaves) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double d0 = lane, d1 = 1.0, d2 = 2.0, d3 = 3.0, a = 0.5, b = 0.25;
    __syncthreads();
    long t0 = clock64();
    for (int it = 0; it < 20; it++) {
        if (wave == 0) asm volatile(R256(BODY4) R256(BODY4) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
        else if (wave == 1) asm volatile(R256(BODY4) R256(BODY4) "s_nop 0\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
        else if (wave == 2) asm volatile(R256(BODY4) R256(BODY4) "s_nop 1\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
        else asm volatile(R256(BODY4) R256(BODY4) "s_nop 2\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
    }
    long t1 = clock64();
    if (lane == 0) out[wave] = t1 - t0;
    sink[threadIdx.x] = d0 + d1 + d2 + d3;
}
__global__ void k_short(long* out, double* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double d0 = lane, d1 = 1.0, d2 = 2.0, d3 = 3.0, a = 0.5, b = 0.25;
    __syncthreads();
    long t0 = clock64();
    for (int it = 0; it < 20 * 32; it++) asm volatile(R16(BODY4) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
    long t1 = clock64();
    if (lane == 0) out[wave] = t1 - t0;
    sink[threadIdx.x] = d0 + d1 + d2 + d3;
}
int main() {
    long* dout; double* dsink; hipMalloc(&dout, 64); hipMalloc(&dsink, 8192);
    for (int nw = 1; nw <= 4; nw++) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_long, dim3(1), dim3(64 * nw), 0, 0, dout, dsink, nw);
        long h[4]; hipMemcpy(h, dout, 32, hipMemcpyDeviceToHost);
        printf("straight-line 16 KiB body, %d wave(s):", nw);
        for (int w = 0; w < nw; w++) printf("  %.2f", (double)h[w] / (20.0 * 2048));
        printf("  cycles per v_fma_f64\n");
    }
    for (int nw = 1; nw <= 4; nw += 3) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_short, dim3(1), dim3(64 * nw), 0, 0, dout, dsink);
        long h[4]; hipMemcpy(h, dout, 32, hipMemcpyDeviceToHost);
        printf("512-byte loop body, %d wave(s):", nw);
        for (int w = 0; w < nw; w++) printf("  %.2f", (double)h[w] / (20.0 * 2048));
        printf("  cycles per v_fma_f64\n");
    }
    return 0;
}
