"""Localise instance-dependent backward error: seeds sweep at a small size, then the per-leaf column profile of A - QR."""
import os, sys, math, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixedprecisionblockqr_amd import api
m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
r = 128
res = []
for seed in range(1, 13):
    h = api.Handle(); h.plan(m, n, r); h.generate(seed); h.factor(); mt = h.metrics()
    res.append((mt["backward_error"], seed)); del h
print(" ".join("%d:%.2e" % (s, e) for e, s in res), flush=True)
worst = max(res)[1]; best = min(res)[1]
for seed in (worst, best):
    A = api.generate_matrix(m, n, seed=seed)
    Ab = np.zeros((m + 1, n), np.float32); Ab[:m] = A
    Q = np.zeros((m, m), np.float32)
    api.dev_mixed_precision_block_qr(Ab, Q, m, n, r)
    R = api.h_strip_R_from_A(Ab, m, n)
    E = A.astype(np.float64) - Q.astype(np.float64) @ R.astype(np.float64)
    prof = [np.linalg.norm(E[:, c:c + 128]) / np.linalg.norm(A[:, c:c + 128]) for c in range(0, n, 128)]
    print("seed %d total %.3e per-leaf:" % (seed, np.linalg.norm(E) / np.linalg.norm(A)), " ".join("%.1e" % p for p in prof), flush=True)
    # row profile too (1024-row bands)
    rp = [np.linalg.norm(E[i:i + 512, :]) / np.linalg.norm(A[i:i + 512, :]) for i in range(0, m, 512)]
    print("   per 512-row band:", " ".join("%.1e" % p for p in rp), flush=True)
