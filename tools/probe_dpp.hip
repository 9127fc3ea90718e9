// tools/probe_dpp.hip -- does row_newbcast:n deliver lane n of the own 16-lane row for v_mov_b32_dpp, v_fmac_f32_dpp and v_mov_b64_dpp?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
    const int l = threadIdx.x;
    float a = 100.f + l, one = 1.f, acc = 0.f, m;
    asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(a));
    asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(one));
    double d = 1000.0 + l, r;
    asm volatile("s_nop 1\n v_mov_b64_dpp %0, %1 row_newbcast:13 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(d));
    o[l] = m; o[64 + l] = acc; o[128 + l] = (float)r;
}
int main() {
    float* d; hipMalloc(&d, 192 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad[3] = {0, 0, 0};
    for (int l = 0; l < 64; l++) {
        const int row = l & ~15;
        if (h[l] != 100.f + row + 5) bad[0]++;
        if (h[64 + l] != 100.f + row + 9) bad[1]++;
        if (h[128 + l] != 1000.f + row + 13) bad[2]++;
    }
    printf("v_mov_b32_dpp row_newbcast: %s   v_fmac_f32_dpp: %s   v_mov_b64_dpp: %s\n", bad[0] ? "WRONG" : "ok", bad[1] ? "WRONG" : "ok", bad[2] ? "WRONG" : "ok");
    if (bad[0] || bad[1] || bad[2]) for (int l = 0; l < 64; l += 7) printf(" lane %d: %g %g %g\n", l, h[l], h[64 + l], h[128 + l]);
    return 0;
}
