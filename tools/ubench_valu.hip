// tools/ubench_valu.hip -- single-wave issue costs on gfx950 for the instruction mixes of the in-wave panel chains
// (v_fma_f64, v_readlane + use, DPP row broadcasts, LDS broadcast reads).  Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define BENCH(name, n_per_iter, setup, body)                                                         \
    __global__ void k_##name(long* out, float* sink) {                                              \
        __shared__ double lds[1024];                                                                 \
        const int lane = threadIdx.x;                                                                \
        lds[lane] = lane; lds[lane + 64] = lane; lds[lane + 128] = 1.0; lds[lane + 192] = 2.0;      \
        __syncthreads();                                                                             \
        double d0 = lane, d1 = 1.0 + lane, d2 = 2.0, d3 = 3.0, d4 = 1.0, d5 = 1.5, d6 = 0.5, d7 = 0.25; \
        float f0 = lane, f1 = 1.f, f2 = 2.f, f3 = 3.f, f4 = 4.f, f5 = 5.f, f6 = 6.f, f7 = 7.f;      \
        unsigned la = (unsigned)(size_t)lds; (void)la;                                               \
        setup;                                                                                       \
        long t0 = clock64();                                                                         \
        for (int it = 0; it < 200; it++) { body; }                                                   \
        long t1 = clock64();                                                                         \
        if (lane == 0) out[0] = t1 - t0;                                                             \
        sink[lane] = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7; \
    }                                                                                                \
    static void run_##name(long* dout, float* dsink) {                                              \
        hipLaunchKernelGGL(k_##name, dim3(1), dim3(64), 0, 0, dout, dsink);                         \
        hipLaunchKernelGGL(k_##name, dim3(1), dim3(64), 0, 0, dout, dsink);                         \
        long h; hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost);                                      \
        printf("%-44s %7.2f cycles per instruction group (%d groups/iter)\n", #name, (double)h / (200.0 * n_per_iter), n_per_iter); \
    }

// 1: independent v_fma_f64 (8 accumulators)
BENCH(fma64_indep, 64, , asm volatile(R16("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n")
      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(d6), "v"(d7)))
// 2: dependent v_fma_f64
BENCH(fma64_dep, 64, , asm volatile(R64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(d0) : "v"(d4), "v"(d7)))
// 3: independent / dependent v_fma_f32
BENCH(fma32_indep, 64, , asm volatile(R16("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n")
      : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f6), "v"(f7)))
BENCH(fma32_dep, 64, , asm volatile(R64("v_fma_f32 %0, %0, %1, %2\n") : "+v"(f0) : "v"(f1), "v"(f7)))
// 4: v_readlane_b32 + v_fma_f32 with the SGPR  (group = readlane + fma)
BENCH(readlane_fma32, 64, , asm volatile(R64("v_readlane_b32 s20, %1, 3\n v_fma_f32 %0, s20, %2, %0\n") : "+v"(f0) : "v"(f1), "v"(f7) : "s20"))
// 4b: the same, 4 readlanes issued first, then 4 fmas
BENCH(readlane4_fma32x4, 16, , asm volatile(R16("v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %4, 4\n v_readlane_b32 s22, %4, 5\n v_readlane_b32 s23, %4, 6\n"
      "v_fma_f32 %0, s20, %5, %0\n v_fma_f32 %1, s21, %5, %1\n v_fma_f32 %2, s22, %5, %2\n v_fma_f32 %3, s23, %5, %3\n")
      : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f4), "v"(f7) : "s20", "s21", "s22", "s23"))
// 5: two readlanes + v_fma_f64 with the SGPR pair (group = 2 readlane + 1 fma)
BENCH(readlane2_fma64, 64, , asm volatile(R64("v_readlane_b32 s20, %1, 3\n v_readlane_b32 s21, %2, 3\n s_nop 1\n v_fma_f64 %0, s[20:21], %3, %0\n")
      : "+v"(d0) : "v"(f1), "v"(f2), "v"(d7) : "s20", "s21"))
// 5b: 4 pairs first, then 4 fma64 on different accumulators
BENCH(readlane8_fma64x4, 16, , asm volatile(R16("v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 3\n v_readlane_b32 s22, %4, 4\n v_readlane_b32 s23, %5, 4\n"
      "v_readlane_b32 s24, %4, 5\n v_readlane_b32 s25, %5, 5\n v_readlane_b32 s26, %4, 6\n v_readlane_b32 s27, %5, 6\n"
      "v_fma_f64 %0, s[20:21], %6, %0\n v_fma_f64 %1, s[22:23], %6, %1\n v_fma_f64 %2, s[24:25], %6, %2\n v_fma_f64 %3, s[26:27], %6, %3\n")
      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(f1), "v"(f2), "v"(d7) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"))
// 6: DPP row_newbcast mov + fma32 (group = mov_dpp + fma)
BENCH(dpp_newbcast_fma32, 64, , asm volatile(R64("v_mov_b32_dpp %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fma_f32 %0, %1, %3, %0\n")
      : "+v"(f0), "+v"(f1) : "v"(f2), "v"(f7)))
// 6b: fma32 with a DPP operand directly (v_fmac_f32_dpp)
BENCH(fmac32_dpp, 64, , asm volatile(R64("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(f0) : "v"(f2), "v"(f7)))
// 7: LDS broadcast reads (all lanes the same address), 8 x ds_read_b128 then one wait  (group = one ds_read_b128)
BENCH(ds_read_b128_bcast, 8, , asm volatile("ds_read_b128 v[100:103], %0\n ds_read_b128 v[104:107], %0 offset:16\n ds_read_b128 v[108:111], %0 offset:32\n ds_read_b128 v[112:115], %0 offset:48\n"
      "ds_read_b128 v[116:119], %0 offset:64\n ds_read_b128 v[120:123], %0 offset:80\n ds_read_b128 v[124:127], %0 offset:96\n ds_read_b128 v[128:131], %0 offset:112\n s_waitcnt lgkmcnt(0)\n"
      : : "v"(la) : "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131", "memory"))
// 7b: one ds_read_b64 + wait: round-trip latency
BENCH(ds_read_b64_latency, 1, , asm volatile("ds_read_b64 v[100:101], %0\n s_waitcnt lgkmcnt(0)\n" : : "v"(la) : "v100", "v101", "memory"))
// 7c: ds_write_b64 (per lane) then ds_read_b128 broadcast of it, wait: the broadcast round trip
BENCH(ds_write_read_roundtrip, 1, unsigned lw = la + 8 * lane, asm volatile("ds_write_b64 %1, %2\n s_waitcnt lgkmcnt(0)\n ds_read_b128 v[100:103], %0\n s_waitcnt lgkmcnt(0)\n" : : "v"(la), "v"(lw), "v"(d7) : "v100", "v101", "v102", "v103", "memory"))
// 7d: the same without waiting for the write (LDS ops of one wave execute in order)
BENCH(ds_write_read_nowait, 1, unsigned lw = la + 8 * lane, asm volatile("ds_write_b64 %1, %2\n ds_read_b128 v[100:103], %0\n s_waitcnt lgkmcnt(0)\n" : : "v"(la), "v"(lw), "v"(d7) : "v100", "v101", "v102", "v103", "memory"))
// 8: v_mul_f64, rsq, cvt chain as in the Cholesky step
BENCH(rsqrt64_chain, 1, , asm volatile("v_cvt_f32_f64 %1, %0\n v_rsq_f32 %1, %1\n v_cvt_f64_f32 %2, %1\n v_mul_f64 %3, %2, %2\n v_fma_f64 %3, %4, %3, %5\n v_mul_f64 %2, %3, %2\n v_mul_f64 %3, %2, %2\n v_fma_f64 %3, %4, %3, %5\n v_mul_f64 %2, %2, %3\n v_mul_f64 %0, %0, %2\n"
      : "+v"(d0), "+v"(f0), "+v"(d1), "+v"(d2) : "v"(d6), "v"(d5)))
// 9: v_rsq_f64 directly
BENCH(rsq_f64, 16, , asm volatile(R16("v_rsq_f64 %0, %1\n") : "+v"(d0) : "v"(d1)))
// 10: v_mov_b32 plain
BENCH(mov32, 64, , asm volatile(R64("v_mov_b32 %0, %1\n") : "+v"(f0) : "v"(f1)))
// 11: fma64 with a DPP-broadcast operand built from two v_mov_dpp (group = 2 mov_dpp + fma64)
BENCH(dpp2_fma64, 64, , asm volatile(R64("v_mov_b32_dpp v100, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v101, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fma_f64 %0, v[100:101], %3, %0\n")
      : "+v"(d0) : "v"(f1), "v"(f2), "v"(d7) : "v100", "v101"))
// 12: v_mov_b64 with DPP row_newbcast
BENCH(mov64_dpp_fma64, 64, , asm volatile(R64("v_mov_b64_dpp v[100:101], %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fma_f64 %0, v[100:101], %2, %0\n")
      : "+v"(d0) : "v"(d1), "v"(d7) : "v100", "v101"))

int main() {
    long* dout; float* dsink;
    hipMalloc(&dout, 64); hipMalloc(&dsink, 1024);
    run_fma64_indep(dout, dsink); run_fma64_dep(dout, dsink); run_fma32_indep(dout, dsink); run_fma32_dep(dout, dsink);
    run_readlane_fma32(dout, dsink); run_readlane4_fma32x4(dout, dsink); run_readlane2_fma64(dout, dsink); run_readlane8_fma64x4(dout, dsink);
    run_dpp_newbcast_fma32(dout, dsink); run_fmac32_dpp(dout, dsink);
    run_ds_read_b128_bcast(dout, dsink); run_ds_read_b64_latency(dout, dsink); run_ds_write_read_roundtrip(dout, dsink); run_ds_write_read_nowait(dout, dsink);
    run_rsqrt64_chain(dout, dsink); run_rsq_f64(dout, dsink); run_mov32(dout, dsink); run_dpp2_fma64(dout, dsink); run_mov64_dpp_fma64(dout, dsink);
    if (hipDeviceSynchronize() != hipSuccess) printf("error\n");
    return 0;
}
