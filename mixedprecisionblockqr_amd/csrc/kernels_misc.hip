// kernels_misc.hip -- data movement, layout conversion, metric reductions, fp64 plumbing path.
#include "mpqr_internal.h"

namespace mpqr {

// ------------------------------------------------------------------ synthetic input
// splitmix64(seed) + linear index -> splitmix64 -> top 24 bits -> [0,1).  Identical to
// mpqr_generate_matrix_host and to the oracle's generator (oracle_qr.c), so CPU and GPU
// agree bit for bit.  Replaces rand()/RAND_MAX of Cuda/mmult.cuh:38-64.
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// local column lc of rank `rank` maps to global column mpqr_part_global_index(lc); world==1 -> identity
__global__ void generate_kernel(float* A, long lda, int m, int nloc, int nglob, uint64_t base, int block, int world,
                                int rank) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)m * nloc) return;
    const int i = (int)(e / nloc), lc = (int)(e % nloc);
    int gc = lc;
    if (world > 1) gc = ((lc / block) * world + rank) * block + (lc % block);
    const uint64_t idx = (uint64_t)i * (uint64_t)nglob + (uint64_t)gc;
    A[(long)i * lda + lc] = (float)(splitmix64(base + idx) >> 40) * (1.0f / 16777216.0f);
}
void launch_generate(float* A, long lda, int m, int n, uint64_t seed, int nglob, int block, int world, int rank,
                     hipStream_t s) {
    const long tot = (long)m * n;
    hipLaunchKernelGGL(generate_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, lda, m, n, nglob,
                       splitmix64(seed), block, world, rank);
}

// the caller has zeroed Q (hipMemsetAsync); only the diagonal is written (a full-matrix elementwise launch would
// need >= 2^32 threads at m = 65536, which HIP rejects)
__global__ void identity_kernel(float* Q, long ldq, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Q[(long)i * ldq + i] = 1.f;
}
void launch_set_identity(float* Q, long ldq, int rows, int cols, hipStream_t s) {
    const int n = rows < cols ? rows : cols;
    hipLaunchKernelGGL(identity_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Q, ldq, n);
}

__global__ void identity_h16_kernel(half_t* Q, long ldq, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Q[(long)i * ldq + i] = (half_t)1.f;
}
void launch_set_identity_h16(half_t* Q, long ldq, int n, hipStream_t s) {
    hipLaunchKernelGGL(identity_h16_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Q, ldq, n);
}

// ------------------------------------------------------------------ boundary layout (Cuda/qr.cu:283-285)
// internal: R on/above the diagonal, v_k[1:] below it in natural rows, v_k[0] in vdiag[k]
// boundary: (m+1) x n, reflector k shifted one row down (v_k[j] at row k+1+j)
__global__ void pack_factor_kernel(const float* A, long lda, const float* vdiag, float* out, int m, int n) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)(m + 1) * n) return;
    const int r = (int)(e / n), c = (int)(e % n);
    float v;
    if (r <= c) v = (r < m) ? A[(long)r * lda + c] : 0.f;
    else if (r - 1 == c) v = vdiag[c];
    else v = A[(long)(r - 1) * lda + c];
    out[e] = v;
}
void launch_pack_factor(const float* A, long lda, const float* vdiag, float* out, int m, int n, hipStream_t s) {
    const long tot = (long)(m + 1) * n;
    hipLaunchKernelGGL(pack_factor_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, lda, vdiag, out, m, n);
}
// rows [r0, r1) of the same (m+1) x n image (the drop-in call streams finished block rows out while the factorisation goes on)
__global__ void pack_factor_rows_kernel(const float* A, long lda, const float* vdiag, float* out, int m, int n, int r0, int r1) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)(r1 - r0) * n) return;
    const int r = r0 + (int)(e / n), c = (int)(e % n);
    float v;
    if (r <= c) v = (r < m) ? A[(long)r * lda + c] : 0.f;
    else if (r - 1 == c) v = vdiag[c];
    else v = A[(long)(r - 1) * lda + c];
    out[(long)r * n + c] = v;
}
void launch_pack_factor_rows(const float* A, long lda, const float* vdiag, float* out, int m, int n, int r0, int r1, hipStream_t s) {
    const long tot = (long)(r1 - r0) * n;
    if (tot <= 0) return;
    hipLaunchKernelGGL(pack_factor_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, lda, vdiag, out, m, n, r0, r1);
}

// boundary -> internal for columns [c0,c1) that already hold reflectors; other columns are copied as plain data
__global__ void unpack_factor_kernel(const float* in, int m, int n, int c0, int c1, float* A, long lda, float* vdiag,
                                     half_t* Vh, long ldvh, half_t* Vt, long ldvt) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)m * n) return;
    const int r = (int)(e / n), c = (int)(e % n);
    if (c >= c0 && c < c1) {
        if (r <= c) {
            A[(long)r * lda + c] = in[(long)r * n + c];
            if (r == c) {
                const float d = in[(long)(r + 1) * n + c];
                vdiag[c] = d;
                Vh[(long)r * ldvh + c] = (half_t)d;
                Vt[(long)c * ldvt + r] = (half_t)d;
            }
        } else {
            const float v = in[(long)(r + 1) * n + c];
            A[(long)r * lda + c] = v;
            Vh[(long)r * ldvh + c] = (half_t)v;
            Vt[(long)c * ldvt + r] = (half_t)v;
        }
    } else {
        A[(long)r * lda + c] = in[(long)r * n + c];
    }
}
void launch_unpack_factor(const float* in, int m, int n, int c0, int c1, float* A, long lda, float* vdiag, half_t* Vh,
                          long ldvh, half_t* Vt, long ldvt, hipStream_t s) {
    const long tot = (long)m * n;
    hipLaunchKernelGGL(unpack_factor_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, in, m, n, c0, c1, A,
                       lda, vdiag, Vh, ldvh, Vt, ldvt);
}

// ---- distributed layout helpers (1-D block-cyclic columns: local column lc <-> global column gc)
__device__ __forceinline__ int cyc_global(int lc, int block, int world, int rank) {
    return ((lc / block) * world + rank) * block + (lc % block);
}
__global__ void identity_cyclic_kernel(float* Q, long ldq, int m, int qloc, int block, int world, int rank) {
    const int lc = blockIdx.x * blockDim.x + threadIdx.x;
    if (lc >= qloc) return;
    const int gc = cyc_global(lc, block, world, rank);
    if (gc < m) Q[(long)gc * ldq + lc] = 1.f;
}
void launch_identity_cyclic(float* Q, long ldq, int m, int qloc, int block, int world, int rank, hipStream_t s) {
    if (qloc <= 0) return;
    hipLaunchKernelGGL(identity_cyclic_kernel, dim3((qloc + 255) / 256), dim3(256), 0, s, Q, ldq, m, qloc, block, world, rank);
}
// the same diagonal in the transposed fp16 shadow Qt[local column][row]
__global__ void identity_cyclic_h16_kernel(half_t* Qt, long ldqt, int m, int qloc, int block, int world, int rank) {
    const int lc = blockIdx.x * blockDim.x + threadIdx.x;
    if (lc >= qloc) return;
    const int gc = cyc_global(lc, block, world, rank);
    if (gc < m) Qt[(long)lc * ldqt + gc] = (half_t)1.f;
}
void launch_identity_cyclic_h16(half_t* Qt, long ldqt, int m, int qloc, int block, int world, int rank, hipStream_t s) {
    if (qloc <= 0) return;
    hipLaunchKernelGGL(identity_cyclic_h16_kernel, dim3((qloc + 255) / 256), dim3(256), 0, s, Qt, ldqt, m, qloc, block, world, rank);
}
// boundary layout ((m+1) x nloc, shifted reflectors) of this rank's columns
__global__ void pack_factor_cyclic_kernel(const float* A, long lda, const float* vdiag, float* out, int m, int nloc,
                                          int block, int world, int rank) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)(m + 1) * nloc) return;
    const int r = (int)(e / nloc), lc = (int)(e % nloc);
    const int gc = cyc_global(lc, block, world, rank);
    float v;
    if (r <= gc) v = (r < m) ? A[(long)r * lda + lc] : 0.f;
    else if (r - 1 == gc) v = vdiag[gc];
    else v = A[(long)(r - 1) * lda + lc];
    out[e] = v;
}
void launch_pack_factor_cyclic(const float* A, long lda, const float* vdiag, float* out, int m, int nloc, int block,
                               int world, int rank, hipStream_t s) {
    const long tot = (long)(m + 1) * nloc;
    if (tot <= 0) return;
    hipLaunchKernelGGL(pack_factor_cyclic_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, lda, vdiag,
                       out, m, nloc, block, world, rank);
}
// dst[c][r] = src[r][c]  (fp16, 64 x 64 tiles through LDS; rows/cols need not be multiples of 64)
__global__ __launch_bounds__(256) void transpose_h16_kernel(const half_t* __restrict__ src, long lds_, half_t* __restrict__ dst,
                                                             long ldd, int rows, int cols) {
    __shared__ half_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(long)(r0 + r) * lds_ + c0 + c] : (half_t)0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int c = e >> 6, r = e & 63;
        if (r0 + r < rows && c0 + c < cols) dst[(long)(c0 + c) * ldd + r0 + r] = tile[r][c];
    }
}
void launch_transpose_h16(const half_t* src, long lds_, half_t* dst, long ldd, int rows, int cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(transpose_h16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, s, src, lds_, dst, ldd, rows, cols);
}

// fp32 copy of the reflectors of columns [c0,c1) (MPQR_PREC_FP32): zero above the diagonal, v_k[0] from vdiag
__global__ void extract_vf_kernel(const float* A, long lda, const float* vdiag, float* Vf, long ldvf, int rows, int c0, int c1) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int w = c1 - c0;
    if (e >= (long)rows * w) return;
    const int r = (int)(e / w), c = c0 + (int)(e % w);
    Vf[(long)r * ldvf + c] = (r > c) ? A[(long)r * lda + c] : (r == c ? vdiag[c] : 0.f);
}
void launch_extract_vf(const float* A, long lda, const float* vdiag, float* Vf, long ldvf, int rows, int c0, int c1, hipStream_t s) {
    const long tot = (long)rows * (c1 - c0);
    if (tot <= 0) return;
    hipLaunchKernelGGL(extract_vf_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, lda, vdiag, Vf, ldvf, rows, c0, c1);
}

// dst[e] = sum_q src[q * stride + e] in slab order (deterministic split-K reduction; dst may alias slab 0)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ src, int nslab, long stride, long n4, float* dst) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    float4 s = ((const float4*)src)[e];
    for (int q = 1; q < nslab; q++) {
        const float4 v = ((const float4*)(src + (long)q * stride))[e];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    ((float4*)dst)[e] = s;
}
void launch_slab_reduce(const float* src, int nslab, long stride, long n_elems, float* dst, hipStream_t s) {
    if (nslab <= 1 && src == dst) return;
    const long n4 = n_elems / 4;                       // callers keep M*N a multiple of 4 (N is a multiple of 64)
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, src, nslab, stride, n4, dst);
}

// h_strip_R_from_A, Cuda/qr.cu:85-100
__global__ void strip_r_kernel(const float* A, long lda, float* R, int m, int n) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)m * n) return;
    const int r = (int)(e / n), c = (int)(e % n);
    R[e] = (r <= c) ? A[(long)r * lda + c] : 0.f;
}
void launch_strip_r(const float* A, long lda, float* R, int m, int n, hipStream_t s) {
    const long tot = (long)m * n;
    hipLaunchKernelGGL(strip_r_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, lda, R, m, n);
}

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// out[0] = max |A[i][j]| (bit pattern; values >= 0 so uint order == float order)
__global__ void absmax_kernel(const float* A, long lda, int m, int n, float* out) {
    float mx = 0.f;
    const long tot = (long)m * n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / n), c = (int)(e % n);
        mx = fmaxf(mx, fabsf(A[(long)r * lda + c]));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)out, __float_as_uint(mx));
}
void launch_absmax(const float* A, long lda, int m, int n, float* out, hipStream_t s) {
    MPQR_IGNORE(hipMemsetAsync(out, 0, sizeof(float), s));
    hipLaunchKernelGGL(absmax_kernel, dim3(2048), dim3(256), 0, s, A, lda, m, n, out);
}

// dst = src over the whole padded buffer (rows x ld floats, ld % 4 == 0) and out[0] = max |src[i][j]| over i < m, j < n, in one
// pass: the factorisation's copy-in of the resident input and the maximum its power-of-two scale needs (mpqr_factor)
__global__ __launch_bounds__(256) void copy_absmax_kernel(const float* __restrict__ src, float* __restrict__ dst, long ld, int rows,
                                                          int m, int n, float* out) {
    float mx = 0.f;
    const int ld4 = (int)(ld >> 2);
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const float4* s4 = (const float4*)(src + (long)r * ld);
        float4* d4 = (float4*)(dst + (long)r * ld);
        for (int c4 = threadIdx.x; c4 < ld4; c4 += 256) {
            const float4 v = s4[c4];
            d4[c4] = v;
            if (r < m) {
                const int c = 4 * c4;
                if (c + 0 < n) mx = fmaxf(mx, fabsf(v.x));
                if (c + 1 < n) mx = fmaxf(mx, fabsf(v.y));
                if (c + 2 < n) mx = fmaxf(mx, fabsf(v.z));
                if (c + 3 < n) mx = fmaxf(mx, fabsf(v.w));
            }
        }
    }
    // one atomic per workgroup, and only if it can raise the maximum (a stale read costs an atomic, never a wrong result): four per workgroup
    // unconditionally were 8192 serialised atomics on one address at 2048^2 -- 80 of the kernel's 98 us (tools/trace_all.py)
    __shared__ float wmax[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (mx > __uint_as_float(__atomic_load_n((unsigned int*)out, __ATOMIC_RELAXED))) atomicMax((unsigned int*)out, __float_as_uint(mx));
    }
}
void launch_copy_absmax(const float* src, float* dst, long ld, int rows, int m, int n, float* out, hipStream_t s) {
    MPQR_IGNORE(hipMemsetAsync(out, 0, sizeof(float), s));
    hipLaunchKernelGGL(copy_absmax_kernel, dim3(std::min(rows, 4096)), dim3(256), 0, s, src, dst, ld, rows, m, n, out);
}

// out[0] += sum (A-B)^2 ; out[1] += sum A^2   (double accumulation)
__global__ void diff_norms_kernel(const float* A, long lda, const float* B, long ldb, int m, int n, double* out) {
    double d2 = 0, a2 = 0;
    const long tot = (long)m * n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / n), c = (int)(e % n);
        const double a = A[(long)r * lda + c], b = B[(long)r * ldb + c];
        d2 += (a - b) * (a - b); a2 += a * a;
    }
    d2 = wave_sum(d2); a2 = wave_sum(a2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], d2); atomicAdd(&out[1], a2); }
}
void launch_diff_norms(const float* A, long lda, const float* B, long ldb, int m, int n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(diff_norms_kernel, dim3(1024), dim3(256), 0, s, A, lda, B, ldb, m, n, out);
}

// out[0] += sum (G - I)^2 ; low 4 bytes of out[1] = float bits of max SIGNED (G - I)  (h_q_error takes the max of signed values, qr.cu:152)
__global__ void gram_minus_identity_kernel(const float* G, long ldg, int m, double* out) {
    double s2 = 0; float mx = 0.f;
    const long tot = (long)m * m;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / m), c = (int)(e % m);
        const float d = G[(long)r * ldg + c] - ((r == c) ? 1.f : 0.f);
        s2 += (double)d * d; mx = fmaxf(mx, d);
    }
    s2 = wave_sum(s2); mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], s2);
        atomicMax((unsigned int*)&out[1], __float_as_uint(mx));   // mx >= 0: bit order == value order
    }
}
void launch_gram_minus_identity(const float* G, long ldg, int m, double* out, hipStream_t s) {
    hipLaunchKernelGGL(gram_minus_identity_kernel, dim3(1024), dim3(256), 0, s, G, ldg, m, out);
}

__global__ void lower_norm_kernel(const float* R, long ldr, int m, int n, double* out) {
    double s2 = 0;
    const long tot = (long)m * n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        const int r = (int)(e / n), c = (int)(e % n);
        if (c < r) { const double v = R[(long)r * ldr + c]; s2 += v * v; }
    }
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[0], s2);
}
void launch_lower_norm(const float* R, long ldr, int m, int n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(lower_norm_kernel, dim3(256), dim3(256), 0, s, R, ldr, m, n, out);
}

// ------------------------------------------------------------------ fp64 plumbing path (C++/main.cpp:5-43)
// One workgroup, 1024 threads = 16 waves.  Column-major doubles.  For each column i:
//   sigma = (float)||u||  (truncated to float exactly as main.cpp:6), w = (u - s sigma e1)/||.||,
//   A <- H A over ALL columns (the reference multiplies the full matrix, so R_ii and the ~1e-7
//   sub-diagonal residue come out of the same arithmetic), Q <- Q H.
// Intended for plumbing-sized problems (config 1: 256 x 256); cost O(m n (m + n)) on one CU.
__device__ __forceinline__ double wave_sum_d(double v) { return wave_sum(v); }

__global__ __launch_bounds__(1024) void qr_f64_kernel(double* A, double* Q, int m, int n, double* w) {
    __shared__ double red[16];
    __shared__ double sh_scalar;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = 0; i < n; i++) {
        const int len = m - i;
        double* col = A + (long)i * m + i;
        // ||u||^2
        double s = 0;
        for (int k = tid; k < len; k += 1024) s += col[k] * col[k];
        s = wave_sum_d(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (tid == 0) { double t = 0; for (int q = 0; q < 16; q++) t += red[q]; sh_scalar = t; }
        __syncthreads();
        const float sigma = (float)sqrt(sh_scalar);
        const double sgn = (col[0] >= 0) ? -1.0 : 1.0;
        __syncthreads();
        for (int k = tid; k < len; k += 1024) w[k] = col[k] - ((k == 0) ? sgn * (double)sigma : 0.0);
        __syncthreads();
        s = 0;
        for (int k = tid; k < len; k += 1024) s += w[k] * w[k];
        s = wave_sum_d(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (tid == 0) { double t = 0; for (int q = 0; q < 16; q++) t += red[q]; sh_scalar = sqrt(t); }
        __syncthreads();
        const double nw = sh_scalar;
        for (int k = tid; k < len; k += 1024) w[k] /= nw;
        __syncthreads();
        // A <- H A : wave per column
        for (int c = wave; c < n; c += 16) {
            double* x = A + (long)c * m + i;
            double d = 0;
            for (int k = lane; k < len; k += 64) d += w[k] * x[k];
            d = wave_sum_d(d);
            for (int k = lane; k < len; k += 64) x[k] -= 2.0 * w[k] * d;
        }
        // Q <- Q H : wave per row r, elements Q(r, i+k) at (i+k)*m + r
        for (int r = wave; r < m; r += 16) {
            double d = 0;
            for (int k = lane; k < len; k += 64) d += Q[(long)(i + k) * m + r] * w[k];
            d = wave_sum_d(d);
            for (int k = lane; k < len; k += 64) Q[(long)(i + k) * m + r] -= 2.0 * d * w[k];
        }
        __syncthreads();
    }
}
void launch_qr_f64(double* A, double* Q, int m, int n, double* work, hipStream_t s) {
    hipLaunchKernelGGL(qr_f64_kernel, dim3(1), dim3(1024), 0, s, A, Q, m, n, work);
}

// ------------------------------------------------------------------ MFMA rate of this box on random operands (measurement aid)
// The 2.5 PFLOP/s dense fp16 figure is width x 2.4 GHz; under an MFMA-dense load the chip lowers its clock, by an amount that depends on
// the operand values and on the MFMA shape (MI355X_MICROARCH.md, DVFS give-back; tools/ubench_mfma.hip).  One wave = a 128 x 64 tile's
// worth of independent accumulators, operands in registers, 8 waves per workgroup, one workgroup per CU: what the matrix pipes deliver
// when nothing else limits them.  SHAPE 0: v_mfma_f32_32x32x16_f16, 1: v_mfma_f32_16x16x32_f16.
typedef half_t half8m __attribute__((ext_vector_type(8)));
typedef float floatx16m __attribute__((ext_vector_type(16)));
typedef float floatx4m __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_peak_kernel(const half_t* __restrict__ src, float* __restrict__ out, long* __restrict__ clk, int iters) {
    const int tid = threadIdx.x;
    constexpr int NA = SHAPE ? 8 : 4, NB = SHAPE ? 4 : 2;
    half8m a[NA], b[NB];
#pragma unroll
    for (int i = 0; i < NA; i++) a[i] = *(const half8m*)(src + ((((long)blockIdx.x * 512 + tid) * 12 + i) * 8) % (1 << 20));
#pragma unroll
    for (int j = 0; j < NB; j++) b[j] = *(const half8m*)(src + ((((long)blockIdx.x * 512 + tid) * 12 + 8 + j) * 8) % (1 << 20));
    floatx16m acc32[SHAPE ? 1 : 4][SHAPE ? 1 : 2];
    floatx4m acc16[SHAPE ? 8 : 1][SHAPE ? 4 : 1];
#pragma unroll
    for (int i = 0; i < (SHAPE ? 1 : 4); i++)
#pragma unroll
        for (int j = 0; j < (SHAPE ? 1 : 2); j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc32[i][j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < (SHAPE ? 8 : 1); i++)
#pragma unroll
        for (int j = 0; j < (SHAPE ? 4 : 1); j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc16[i][j][e] = 0.f;
    const long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {                         // one iteration = 128 x 64 x 32 MACs per wave in either shape
        if (SHAPE) {
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc16[SHAPE ? i : 0][SHAPE ? j : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[SHAPE ? i : 0], b[SHAPE ? j : 0], acc16[SHAPE ? i : 0][SHAPE ? j : 0], 0, 0, 0);
        } else {
#pragma unroll
            for (int rep = 0; rep < 2; rep++)
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc32[SHAPE ? 0 : i][SHAPE ? 0 : j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc32[SHAPE ? 0 : i][SHAPE ? 0 : j], 0, 0, 0);
        }
    }
    const long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < (SHAPE ? 1 : 4); i++)
#pragma unroll
        for (int j = 0; j < (SHAPE ? 1 : 2); j++)
#pragma unroll
            for (int e = 0; e < 16; e++) s += acc32[i][j][e];
#pragma unroll
    for (int i = 0; i < (SHAPE ? 8 : 1); i++)
#pragma unroll
        for (int j = 0; j < (SHAPE ? 4 : 1); j++)
#pragma unroll
            for (int e = 0; e < 4; e++) s += acc16[i][j][e];
    out[(long)blockIdx.x * 512 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
void launch_mfma_peak(int shape, const half_t* src, float* out, long* clk, int nwg, int iters, hipStream_t s) {
    if (shape) hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(nwg), dim3(512), 0, s, src, out, clk, iters);
    else hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(nwg), dim3(512), 0, s, src, out, clk, iters);
}

}  // namespace mpqr
