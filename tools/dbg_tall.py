"""Tall matrices (m >= 3 n): one-shot Q formation vs the backward accumulation (MPQR_QONESHOT=0), errors."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixedprecisionblockqr_amd import api
for (m, n, r) in [(8192, 2048, 128), (6400, 2000, 128), (12288, 3072, 128)]:
    A = api.generate_matrix(m, n, seed=5)
    Ab = np.zeros((m + 1, n), np.float32); Ab[:m] = A
    Q = np.zeros((m, m), np.float32)
    api.dev_mixed_precision_block_qr(Ab, Q, m, n, r)
    R = api.h_strip_R_from_A(Ab, m, n)
    A64 = A.astype(np.float64); Q64 = Q.astype(np.float64)
    be = np.linalg.norm(A64 - Q64 @ R.astype(np.float64)) / np.linalg.norm(A64)
    oe = np.linalg.norm(Q64.T @ Q64 - np.eye(m))
    print("m=%d n=%d backward %.3e  ||QtQ-I||_F %.3e (bound %.3e)  oneshot=%s" % (m, n, be, oe, 2e-3 * np.sqrt(m), os.environ.get("MPQR_QONESHOT", "1")), flush=True)
