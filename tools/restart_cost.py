"""Cost of the robust fallback (a flagged Gram-Householder leaf): factor the rank-deficient Jacobian stand-in a few times.
Run under rocprofv3 --kernel-trace --stats to see which kernels the extra time goes to (profiles/README.md)."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, mixedprecisionblockqr_amd as mp
rd = int(sys.argv[1]) if len(sys.argv) > 1 else 7
M = mp.synthetic_jacobian(rank_deficiency=rd)
m, n = M.shape
hh = mp.Handle(0); hh.plan(m, n, 64)
for _ in range(4):
    hh.set_matrix(M); hh.factor(); hh.sync(); t = hh.timings()
    print({k: t[k] for k in ("n_passes", "n_robust_leaves", "restart_block", "ms_factor", "ms_host_enqueue", "n_gh_leaves") if k in t})
hh.close()
