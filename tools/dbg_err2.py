"""First-leaf check: GPU rows 0..127 of R right of the first leaf against (a) the exact update with the GPU's own fp16
reflectors and an exact T, (b) an fp16-operand emulation of the same update."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixedprecisionblockqr_amd import api
f16 = lambda x: x.astype(np.float16).astype(np.float32)
m = n = 3072; r = 128
for seed in (12, 8, 1, 2):
    A = api.generate_matrix(m, n, seed=seed)
    Ab = np.zeros((m + 1, n), np.float32); Ab[:m] = A
    Q = np.zeros((m, m), np.float32)
    api.dev_mixed_precision_block_qr(Ab, Q, m, n, r)
    # reflectors of the first leaf from the boundary layout (shifted one row down)
    V = np.zeros((m, 128), np.float64)
    for k in range(128):
        V[k:, k] = Ab[k + 1:m + 1, k]
    Vh = f16(V.astype(np.float32)).astype(np.float64)
    S = Vh.T @ Vh
    T = np.linalg.inv(np.triu(S, 1) + np.diag(np.diag(S)) / 2)
    A2 = A[:, 128:].astype(np.float64)
    exact = A2 - Vh @ (T.T @ (Vh.T @ A2))                       # H^T A2 with the fp16 reflectors, exact arithmetic
    Rgpu = Ab[:128, 128:].astype(np.float64)
    X = f16(A2.astype(np.float32)).astype(np.float64).T @ Vh
    Y = f16((f16(X.astype(np.float32)).astype(np.float64) @ f16(T.astype(np.float32)).astype(np.float64)).astype(np.float32)).astype(np.float64)
    emu = A2 - Vh @ Y.T
    nrm = np.linalg.norm(exact[:128])
    sv = np.linalg.norm(V, axis=0)
    print("seed %2d  |Rgpu-exact|/|exact| %.3e   |emu-exact| %.3e   |Rgpu-emu| %.3e   ||v_k|| min %.6f max %.6f  S00 %.6f" %
          (seed, np.linalg.norm(Rgpu - exact[:128]) / nrm, np.linalg.norm(emu[:128] - exact[:128]) / nrm,
           np.linalg.norm(Rgpu - emu[:128]) / nrm, sv.min(), sv.max(), S[0, 0]), flush=True)
