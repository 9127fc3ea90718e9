"""Debug driver for the Gram-level look-ahead: small factorisations with MPQR_LA=1 vs 0, errors and timing."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixedprecisionblockqr_amd as mp
from mixedprecisionblockqr_amd import api

def run(m, n, r, seed=1):
    rng = np.random.default_rng(seed)
    A = rng.random((m, n), dtype=np.float32)
    Ab = np.zeros((m + 1, n), np.float32); Ab[:m] = A
    Q = np.zeros((m, m), np.float32)
    api.dev_mixed_precision_block_qr(Ab, Q, m, n, r)
    R = api.h_strip_R_from_A(Ab, m, n)
    be = np.linalg.norm(A.astype(np.float64) - Q.astype(np.float64) @ R.astype(np.float64)) / np.linalg.norm(A)
    oe = np.abs(Q.astype(np.float64).T @ Q - np.eye(m)).max()
    low = np.abs(np.tril(R[:n], -1)).max()
    return be, oe, low

for (m, n) in [(2560, 512), (4096, 2048), (3000, 1280)]:
    be, oe, low = run(m, n, 128)
    print("m=%d n=%d  backward %.3e  max|QtQ-I| %.3e  lower %.1e  LA=%s" % (m, n, be, oe, low, os.environ.get("MPQR_LA", "1")), flush=True)
