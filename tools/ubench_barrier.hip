// microbenchmark: cost of one {LDS write -> __syncthreads -> LDS read -> dependent math} phase in a single workgroup
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(float* out, int iters) {
    __shared__ float sh[1024];
    __shared__ double shd[128];
    const int tid = threadIdx.x;
    float acc = tid * 0.001f;
    double dacc = 1.0 + tid * 1e-3;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {            // barrier only
            __syncthreads();
        } else if (MODE == 1) {     // LDS write, barrier, LDS read
            sh[tid] = acc;
            __syncthreads();
            acc += sh[(tid + 1) % blockDim.x];
            __syncthreads();
        } else if (MODE == 2) {     // + fp64 dependent chain (rsqrt-like: 12 dependent fma)
            if (tid < 128) shd[tid] = dacc;
            __syncthreads();
            double x = shd[it & 127];
#pragma unroll
            for (int q = 0; q < 12; q++) x = x * 0.999 + 0.001;
            dacc += x;
            __syncthreads();
        } else {                    // fp64 sqrt + div
            if (tid < 128) shd[tid] = dacc;
            __syncthreads();
            double x = shd[it & 127];
            dacc += 1.0 / sqrt(x + 2.0);
            __syncthreads();
        }
    }
    out[tid] = acc + (float)dacc;
}
template <int MODE> void run(int threads, float* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode %d threads %4d : %.3f us / iteration\n", MODE, threads, ms * 1e3 / iters);
}
int main() {
    float* d; hipMalloc(&d, 4096);
    for (int t : {64, 256, 512, 1024}) { run<0>(t, d); run<1>(t, d); run<2>(t, d); run<3>(t, d); }
    return 0;
}
