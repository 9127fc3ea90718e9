// tools/probe_cumask.hip -- does hipExtStreamCreateWithCUMask restrict where workgroups run on this device?  Launches a kernel that
// records (XCC_ID, HW_ID) per workgroup on an unmasked stream and on streams with several masks, and counts the distinct CUs used.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void where(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
    const int nwg = 4096;
    uint32_t* d; CK(hipMalloc(&d, nwg * 8));
    std::vector<uint32_t> h(2 * nwg);
    const uint32_t pats[] = {0xffffffffu, 0x7f7f7f7fu, 0x55555555u, 0x0000ffffu, 0x000000ffu, 0x1u};
    for (int p = -1; p < 6 + 3; p++) {
        hipStream_t st;
        if (p < 0) { CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); }
        else {
            uint32_t mask[8]; for (int i = 0; i < 8; i++) mask[i] = pats[p < 6 ? p : 0];
            // word i = CU i of every shader engine?  rows 0..6 / 0..5 / 0..3 of the 8 CUs of each engine
            if (p >= 6) { const int n = p == 6 ? 7 : p == 7 ? 6 : 4; for (int i = 0; i < 8; i++) mask[i] = i < n ? 0xffffffffu : 0u; }
            hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
            if (e != hipSuccess) { printf("mask %08x: create failed: %s\n", pats[p], hipGetErrorString(e)); continue; }
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(where, dim3(nwg), dim3(256), 0, st, d, 20000);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost));
        std::set<uint32_t> cus, xccs;
        for (int i = 0; i < nwg; i++) {
            const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu); xccs.insert(xcc);
        }
        if (p < 0) printf("no mask      : %3zu distinct CUs on %zu XCCs, %.3f ms\n", cus.size(), xccs.size(), ms);
        else if (p < 6) printf("mask %08x: %3zu distinct CUs on %zu XCCs, %.3f ms\n", pats[p], cus.size(), xccs.size(), ms);
        else printf("CU rows 0..%d of every engine: %3zu distinct CUs on %zu XCCs, %.3f ms\n", (p == 6 ? 7 : p == 7 ? 6 : 4) - 1, cus.size(), xccs.size(), ms);
        CK(hipStreamDestroy(st));
    }
    return 0;
}
