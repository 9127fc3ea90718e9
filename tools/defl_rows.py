import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mixedprecisionblockqr_amd as mp
for (m, n, dist) in ((2320, 1980, "gauss"), (2320, 1980, "unif"), (3072, 2048, "gauss"), (4096, 2048, "gauss"), (2048, 2048, "unif"), (1536, 1024, "gauss")):
    rng = np.random.default_rng(3)
    A0 = (rng.standard_normal((m, n)) if dist == "gauss" else rng.random((m, n))).astype(np.float32)
    for nd in (0, 1, 7):
        A = A0.copy()
        for q in range(nd):
            A[:, 130 + 12 * q + 7] = 0.5 * A[:, 130 + 12 * q + 2]
        h = mp.Handle(0); h.plan(m, n, 64); h.set_matrix(A); h.factor(); h.sync()
        t = h.timings(); mt = h.metrics(); h.close()
        print(f"{m}x{n} {dist} dependent {nd}: passes {t['n_passes']} deflated {t['n_deflated_columns']} be {mt['backward_error']:.2e} qf {mt['q_error_fro']:.3f}", flush=True)
