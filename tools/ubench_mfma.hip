// tools/ubench_mfma.hip -- what the matrix pipes of THIS box deliver on random data: the denominator behind roofline.frac.
//
// The 2.5 PFLOP/s dense fp16 figure is clock x width at 2.4 GHz; under an MFMA-dense load the chip lowers its clock
// (MI355X_MICROARCH.md, DVFS give-back: 1.9-2.0 GHz on random operands), and the two fp16 MFMA shapes do not hold the same clock.
// Three loops, every CU busy, 512-thread workgroups (two waves per SIMD), random fp16 operands:
//   reg32   v_mfma_f32_32x32x16_f16, operands in registers (the ceiling of gemm6's MFMA shape)
//   reg16   v_mfma_f32_16x16x32_f16, operands in registers
//   lds32 / lds16: the same with every operand fragment re-read from LDS (ds_read_b128, issued one iteration ahead of its MFMAs, one
//           s_waitcnt lgkmcnt(0) per iteration) as a 128 x 64 wave tile does per K step of 16 / 32: the ceiling of ANY kernel with this wave tile
// Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).  Development tool; build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p; }
template <int OFF> __device__ __forceinline__ half8 ds_read16(unsigned a) {
    half8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
    return v;
}

constexpr int ITERS = 4096;

// one wave: 128 x 64 output = 4 x 2 tiles of 32x32, K step 16 per MFMA round: 8 MFMAs per round, 4 A + 2 B fragments
template <bool LDS>
__global__ __launch_bounds__(512) void k32(const _Float16* __restrict__ src, float* __restrict__ out, long* __restrict__ clk) {
    __shared__ __attribute__((aligned(16))) _Float16 sm[8][6][64][8];      // per wave: 6 fragments x 64 lanes x 8 halves = 6 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    half8 a[4], b[2];
    for (int i = 0; i < 4; i++) a[i] = *(const half8*)(src + ((blockIdx.x * 512 + tid) * 6 + i) * 8 % (1 << 20));
    for (int j = 0; j < 2; j++) b[j] = *(const half8*)(src + ((blockIdx.x * 512 + tid) * 6 + 4 + j) * 8 % (1 << 20));
    if (LDS) {
        for (int i = 0; i < 4; i++) *(half8*)&sm[wave][i][lane][0] = a[i];
        for (int j = 0; j < 2; j++) *(half8*)&sm[wave][4 + j][lane][0] = b[j];
        __syncthreads();
    }
    floatx16 acc[4][2];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const long t0 = __builtin_amdgcn_s_memtime(); const long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned base = lds_addr(&sm[wave][0][lane][0]);
    for (int it = 0; it < ITERS; it++) {
        half8 an[4], bn[2];
        if (LDS) {                                                          // next iteration's fragments: in flight behind this one's MFMAs
            an[0] = ds_read16<0 * 1024>(base); an[1] = ds_read16<1 * 1024>(base); an[2] = ds_read16<2 * 1024>(base); an[3] = ds_read16<3 * 1024>(base);
            bn[0] = ds_read16<4 * 1024>(base); bn[1] = ds_read16<5 * 1024>(base);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (LDS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = an[i];
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = bn[j];
        }
    }
    const long t1 = __builtin_amdgcn_s_memtime(); const long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// the same wave tile on 16x16x32: 8 x 4 tiles, K step 32 per round: 32 MFMAs per round, 8 A + 4 B fragments
template <bool LDS>
__global__ __launch_bounds__(512) void k16(const _Float16* __restrict__ src, float* __restrict__ out, long* __restrict__ clk) {
    __shared__ __attribute__((aligned(16))) _Float16 sm[8][12][64][8];     // 12 KiB per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    half8 a[8], b[4];
    for (int i = 0; i < 8; i++) a[i] = *(const half8*)(src + ((blockIdx.x * 512 + tid) * 12 + i) * 8 % (1 << 20));
    for (int j = 0; j < 4; j++) b[j] = *(const half8*)(src + ((blockIdx.x * 512 + tid) * 12 + 8 + j) * 8 % (1 << 20));
    if (LDS) {
        for (int i = 0; i < 8; i++) *(half8*)&sm[wave][i][lane][0] = a[i];
        for (int j = 0; j < 4; j++) *(half8*)&sm[wave][8 + j][lane][0] = b[j];
        __syncthreads();
    }
    floatx4 acc[8][4];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;
    const long t0 = __builtin_amdgcn_s_memtime(); const long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned base = lds_addr(&sm[wave][0][lane][0]);
    for (int it = 0; it < ITERS / 2; it++) {                               // one round = K 32 = two rounds of the 32x32x16 loop
        half8 an[8], bn[4];
        if (LDS) {
            an[0] = ds_read16<0 * 1024>(base); an[1] = ds_read16<1 * 1024>(base); an[2] = ds_read16<2 * 1024>(base); an[3] = ds_read16<3 * 1024>(base);
            an[4] = ds_read16<4 * 1024>(base); an[5] = ds_read16<5 * 1024>(base); an[6] = ds_read16<6 * 1024>(base); an[7] = ds_read16<7 * 1024>(base);
            bn[0] = ds_read16<8 * 1024>(base); bn[1] = ds_read16<9 * 1024>(base); bn[2] = ds_read16<10 * 1024>(base); bn[3] = ds_read16<11 * 1024>(base);
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (LDS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = an[i];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = bn[j];
        }
    }
    const long t1 = __builtin_amdgcn_s_memtime(); const long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; i++) for (int j = 0; j < 4; j++) for (int e = 0; e < 4; e++) s += acc[i][j][e];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename K>
static void run(const char* name, K kern, const _Float16* dsrc, float* dout, long* dclk, int nwg) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 0, 0, dsrc, dout, dclk);      // warm-up: let the clock settle
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 0, 0, dsrc, dout, dclk);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long> clk(2 * nwg);
    hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost);
    double ghz = 0; for (int b = 0; b < nwg; b++) ghz += (double)clk[2 * b] / (double)clk[2 * b + 1] * 0.1;      // s_memrealtime ticks at 100 MHz
    ghz /= nwg;
    const double flops = (double)reps * nwg * 8 /*waves*/ * ITERS * 8 /*MFMA*/ * 2.0 * 32 * 32 * 16;
    printf("%-8s %8.1f TFLOP/s   in-kernel clock %.2f GHz   (%.3f ms per launch, %d workgroups of 8 waves)\n", name, flops / (ms * 1e-3) / 1e12, ghz, ms / reps, nwg);
}

int main() {
    const int nwg = 256;
    std::vector<_Float16> h(1 << 20);
    srand(1234);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    _Float16* dsrc; float* dout; long* dclk;
    hipMalloc(&dsrc, h.size() * 2 + 4096); hipMalloc(&dout, (size_t)nwg * 512 * 4); hipMalloc(&dclk, (size_t)nwg * 16);
    hipMemcpy(dsrc, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("fp16 MFMA rates on random operands, 256 CUs x 8 waves (dense peak at 2.4 GHz: 2516 TFLOP/s)\n");
    run("reg32", k32<false>, dsrc, dout, dclk, nwg);
    run("reg16", k16<false>, dsrc, dout, dclk, nwg);
    run("lds32", k32<true>, dsrc, dout, dclk, nwg);
    run("lds16", k16<true>, dsrc, dout, dclk, nwg);
    hipMemset(dsrc, 0, h.size() * 2);
    printf("all-zero operands:\n");
    run("reg32", k32<false>, dsrc, dout, dclk, nwg);
    run("reg16", k16<false>, dsrc, dout, dclk, nwg);
    return 0;
}
