// host_util.cpp -- host-only pieces of the path (no HIP): Jacobian reader, CSV result log,
// FLOP models, block-cyclic partition arithmetic, host twin of the device generator.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <string>

#include "../../include/mpqr.h"

extern "C" {

// replaces read_euroc_jacobian (Cuda/qr.cu:696-776).  First line "<rows> <cols>", then
// "<row> <col> <value>" triples separated by blanks (leading blanks allowed); unspecified entries
// are zero, later duplicates overwrite.  The reference asserts on a missing file; here: MPQR_ERR_IO.
int mpqr_read_euroc_jacobian(const char* path, int* rows, int* cols, float** matrix) {
    if (!path || !rows || !cols || !matrix) return MPQR_ERR_INVALID;
    *matrix = NULL;
    FILE* f = fopen(path, "r");
    if (!f) return MPQR_ERR_IO;
    char line[1024];
    if (!fgets(line, sizeof line, f)) { fclose(f); return MPQR_ERR_IO; }
    char* p = line;
    long r = strtol(p, &p, 10), c = strtol(p, &p, 10);
    if (r <= 0 || c <= 0 || r > (1 << 24) || c > (1 << 24)) { fclose(f); return MPQR_ERR_IO; }
    float* M = (float*)calloc((size_t)r * (size_t)c, sizeof(float));
    if (!M) { fclose(f); return MPQR_ERR_ALLOC; }
    while (fgets(line, sizeof line, f)) {
        char* q = line;
        char* e;
        long ri = strtol(q, &e, 10);
        if (e == q) continue;                       // blank line
        q = e;
        long ci = strtol(q, &e, 10);
        if (e == q) continue;
        q = e;
        float v = strtof(q, &e);                    // std::stof at qr.cu:768
        if (e == q) continue;
        if (ri < 0 || ri >= r || ci < 0 || ci >= c) { free(M); fclose(f); return MPQR_ERR_IO; }
        M[(size_t)ri * (size_t)c + (size_t)ci] = v;
    }
    fclose(f);
    *rows = (int)r; *cols = (int)c; *matrix = M;
    return MPQR_OK;
}

int mpqr_write_euroc_jacobian(const char* path, int rows, int cols, const float* M) {
    if (!path || !M || rows < 1 || cols < 1) return MPQR_ERR_INVALID;
    FILE* f = fopen(path, "w");
    if (!f) return MPQR_ERR_IO;
    fprintf(f, "%d %d\n", rows, cols);
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) {
            const float v = M[(size_t)r * cols + c];
            if (v != 0.f) fprintf(f, "%d %d %.9g\n", r, c, (double)v);
        }
    fclose(f);
    return MPQR_OK;
}

void mpqr_free_host(void* p) { free(p); }

// replaces h_write_results_to_log (Cuda/qr.cu:58-83): header "rows,cols,runtime,flops,error" when the
// file is new, then one row of std::to_string(double)-formatted values ("%f").
int mpqr_write_results_to_log(const char* dir, const char* file_name, int height, int width, float time_ms,
                              float flops_per_second, float backward_error) {
    if (!file_name) return MPQR_ERR_INVALID;
    std::string d = dir ? dir : "log";
    mkdir(d.c_str(), 0755);
    std::string path = d + "/" + file_name + ".txt";
    FILE* probe = fopen(path.c_str(), "r");
    const bool fresh = (probe == NULL);
    if (probe) fclose(probe);
    FILE* f = fopen(path.c_str(), "a");
    if (!f) return MPQR_ERR_IO;
    if (fresh) fputs("rows,cols,runtime,flops,error\n", f);
    fprintf(f, "%f,%f,%f,%f,%f\n", (double)height, (double)width, (double)time_ms, (double)flops_per_second,
            (double)backward_error);
    fclose(f);
    return MPQR_OK;
}

// replaces h_qr_flops_per_second (Cuda/qr.cu:102-113), fp32 arithmetic on purpose
float mpqr_qr_flops_per_second(float time_ms, int m, int n) {
    float mf = (float)m, nf = (float)n;
    float flops = 4.0f * powf(mf, 2.f) * nf;
    flops -= mf * powf(nf, 2.f);
    flops += powf(nf, 3.f) / 3.0f;
    flops /= time_ms / 1000.0f;
    return flops;
}

double mpqr_flops_geqrf(int m, int n) { return 2.0 * m * (double)n * n - (2.0 / 3.0) * (double)n * n * n; }
double mpqr_flops_form_q(int m, int n) {
    return 4.0 * ((double)m * m * n - (double)m * n * n + (double)n * n * n / 3.0);
}
double mpqr_flops_trailing(int m, int n, int r) {
    double f = 0;
    for (int l = 0; l < n; l += r) {
        const int t = (l + r < n) ? l + r : n;
        const double W = m - l, rr = t - l, nk = n - t;
        f += 4.0 * W * rr * nk + rr * rr * nk;
    }
    return f;
}
double mpqr_flops_panel(int m, int n, int r) {
    double f = 0;
    for (int l = 0; l < n; l += r) {
        const int t = (l + r < n) ? l + r : n;
        const double W = m - l, rr = t - l;
        f += 2.0 * W * rr * rr;
    }
    return f;
}

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
void mpqr_generate_matrix_host(float* A, int m, int n, uint64_t seed) {
    const uint64_t base = splitmix64(seed);
    for (size_t i = 0; i < (size_t)m * (size_t)n; i++)
        A[i] = (float)(splitmix64(base + i) >> 40) * (1.0f / 16777216.0f);
}

// 1-D block-cyclic column distribution: block j = columns [j*block, (j+1)*block) on rank j % world
int mpqr_part_owner(int col, int block, int world) { return (col / block) % world; }
int mpqr_part_local_cols(int n, int block, int world, int rank) {
    int cnt = 0;
    for (int c = 0; c < n; c += block)
        if ((c / block) % world == rank) cnt += (c + block <= n) ? block : n - c;
    return cnt;
}
int mpqr_part_local_index(int col, int block, int world) { return ((col / block) / world) * block + col % block; }
int mpqr_part_global_index(int lcol, int block, int world, int rank) {
    return ((lcol / block) * world + rank) * block + lcol % block;
}

}  // extern "C"
