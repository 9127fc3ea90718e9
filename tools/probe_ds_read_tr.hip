// Probe of ds_read_b64_tr_b16 on gfx950: which 16-bit elements does each lane receive?
// LDS is filled with element id = byte_offset/2; every lane passes the address of "its" 8-byte chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(int* out, int row_stride_halfs) {
    __shared__ short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // lane l points at row (l % 16), column chunk (l / 16) of a [16 rows][row_stride] halfs matrix: 4 halfs per chunk
    const short* p = &lds[(l & 15) * row_stride_halfs + (l >> 4) * 4];
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)p);
    for (int q = 0; q < 4; q++) out[l * 4 + q] = v[q];
}
int main() {
    int* d; hipMalloc(&d, 256 * 4);
    const int rs = 64;
    probe<<<1, 64>>>(d, rs);
    int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) {
        printf("lane %2d:", l);
        for (int q = 0; q < 4; q++) printf("  (r%2d,c%2d)", h[l * 4 + q] / rs, h[l * 4 + q] % rs);
        printf("\n");
    }
    return 0;
}
