"""tools/bench_gemm.py -- the library's large-shape GEMM kernels ALONE on the GPU (mpqr_bench_gemm), at the shapes the 16384^2
factorisation launches them with: the kernel-alone rate next to what bench.py reports beside the panel chain.
    python tools/bench_gemm.py [--quick]        -> one line per (kernel, epilogue, shape): ms, TFLOP/s, fraction of 2.5 PFLOP/s"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixedprecisionblockqr_amd as mp

KN = {6: "gemm6 (fp16 x fp16, LDS-DMA ping-pong, 32x32x16)", 2: "gemm2<A_F32T> (fp32 A^T converted while staged)", 16: "gemm6 on 16x16x32 (experiment)"}
MD = {0: "store f32", 1: "store f16", 2: "C -= (fp32 RMW)", 3: "C -= + fp16 shadow"}
CASES = [
    # (label, kernel, mode, M, N, K)
    ("far nn  A2 -= V Y^T, K=1024", 6, 2, 16384, 13312, 1024),
    ("far nn  A2 -= V Y^T, K=2048 (pair)", 6, 2, 16384, 13312, 2048),
    ("far nn  K=2048 + shadow", 6, 3, 16384, 13312, 2048),
    ("Q   nn  Q2 -= V Y^T, K=2048 + shadow", 6, 3, 16384, 16384, 2048),
    ("far nn  K=4096 (quad, not built)", 6, 2, 16384, 12288, 4096),
    ("far tn  X = A2^T V from fp32, N=1024", 2, 1, 13312, 1024, 16384),
    ("far tn  X = A2^T V from fp32, N=2048", 2, 1, 13312, 2048, 16384),
    ("far tn  from an fp16 shadow, N=1024", 6, 1, 13312, 1024, 16384),
    ("far tn  from an fp16 shadow, N=2048", 6, 1, 13312, 2048, 16384),
    ("Q   tn  X = Q2^T V (shadow), N=2048", 6, 1, 16384, 2048, 16384),
    ("tall Q  Q = I - W V^T store, K=8192", 6, 0, 16384, 16384, 8192),
    ("16x16x32: store f32, K=8192", 16, 0, 16384, 16384, 8192),
    ("16x16x32: Q tn shape, store f16", 16, 1, 16384, 2048, 16384),
    ("16x16x32: far nn K=2048, register RMW", 16, 2, 16384, 13312, 2048),
    # the read-modify-write epilogue nearly alone (K = 64: one K tile): one workgroup, one per CU, thirteen per CU
    ("epilogue probe: 1 tile, DMA ring", 6, 2, 256, 256, 64),
    ("epilogue probe: 256 tiles, DMA ring", 6, 2, 4096, 4096, 64),
    ("epilogue probe: 3328 tiles, DMA ring", 6, 2, 16384, 13312, 64),
    ("epilogue probe: 1 tile, 16-B registers", 16, 2, 256, 256, 64),
    ("epilogue probe: 256 tiles, 16-B registers", 16, 2, 4096, 4096, 64),
    ("epilogue probe: 3328 tiles, 16-B registers", 16, 2, 16384, 13312, 64),
    ("store probe: 3328 tiles f32 store", 6, 0, 16384, 13312, 64),
    ("store probe: 3328 tiles f32 store 16-B", 16, 0, 16384, 13312, 64),
]


def main():
    quick = "--quick" in sys.argv
    h = mp.Handle(0)
    rows = []
    for label, k, md, M, N, K in CASES:
        if quick and K > 64:
            continue
        ms = h.bench_gemm(k, md, M, N, K, iters=8)
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        rows.append({"case": label, "kernel": KN[k], "epilogue": MD[md], "M": M, "N": N, "K": K, "ms": ms, "tflops": tf, "frac_of_2500": tf / 2500.0})
        print(f"{label:42s} {M:6d} x {N:6d} x {K:6d}  {ms:8.3f} ms  {tf:7.1f} TFLOP/s  {tf / 25.0:5.1f} % of 2.5 PF   [{KN[k]}; {MD[md]}]", flush=True)
    h.close()
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
