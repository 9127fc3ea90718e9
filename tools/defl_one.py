import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mixedprecisionblockqr_amd as mp
m, n = 3072, 2048
rng = np.random.default_rng(3)
A = rng.standard_normal((m, n)).astype(np.float32)
A[:, 137] = 0.5 * A[:, 132]
for fused in (1, 1, 1, 1, 0, 0):
    os.environ["MPQR_FUSED_LEAF"] = str(fused)
    h = mp.Handle(0); h.plan(m, n, 64); h.set_matrix(A); h.factor(); h.sync()
    t = h.timings(); mt = h.metrics()
    Q = h.q().astype(np.float64); R = h.r_matrix().astype(np.float64); h.close()
    E = A.astype(np.float64) - Q @ R
    cn = np.linalg.norm(E, axis=0) / np.linalg.norm(A, axis=0)
    print(f"fused {fused}: passes {t['n_passes']} robust {t['n_robust_leaves']} be {mt['backward_error']:.2e}; err by 128-col leaf:", [float(f"{np.linalg.norm(E[:, i:i+128]) / np.linalg.norm(A[:, i:i+128]):.1e}") for i in range(0, n, 128)], flush=True)
