#!/usr/bin/env python3
"""In-line deflation of dependent columns (gh_solve3's pivot clamp) on the ill-conditioned test matrices: passes, robust leaves, deflated
columns, errors, time against the full-rank twin."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mixedprecisionblockqr_amd as mp

def run(M, r, **kw):
    m, n = M.shape
    h = mp.Handle(0)
    h.plan(m, n, r, **kw)
    best = None
    for _ in range(4):
        h.set_matrix(M); h.factor(); h.sync()
        t = h.timings(); best = t["ms_factor"] if best is None else min(best, t["ms_factor"])
    mt = h.metrics(); R = h.r_matrix(); h.close()
    return best, t, mt, R

def show(name, M, M0, r, **kw):
    ms0, t0, mt0, _ = run(M0, r, **kw)
    ms, t, mt, R = run(M, r, **kw)
    d = np.sort(np.abs(np.diag(R)[:M.shape[1]]))
    print(f"{name}: passes {t['n_passes']} robust {t['n_robust_leaves']} deflated {t['n_deflated_columns']}  be {mt['backward_error']:.2e} (full rank {mt0['backward_error']:.2e})  "
          f"qf {mt['q_error_fro']:.3f} ({mt0['q_error_fro']:.3f})  smallest |R_kk| {d[:3]}  {ms:.2f} ms vs {ms0:.2f} ms = {ms / ms0:.2f}", flush=True)

rng = np.random.default_rng(11)
A0 = rng.standard_normal((1500, 480)).astype(np.float32); A = A0.copy()
A[:, 7] = A[:, 3]; A[:, 20] = A[:, 5] + 1e-6 * A[:, 6]; A[:, 300] = 0
show("1500x480 dependent + zero column", A, A0, 32)
show("Jacobian stand-in, 7 dependent", mp.synthetic_jacobian(rank_deficiency=7), mp.synthetic_jacobian(rank_deficiency=0), 64)
M0 = np.random.default_rng(5).random((6144, 4096), dtype=np.float32)
for at in (0, 3584):
    M = M0.copy()
    for q in range(7): M[:, at + 12 * q + 7] = 0.5 * M[:, at + 12 * q + 2]
    show(f"6144x4096 7 dependent at {at}", M, M0, 128, outer_block=512)
M6 = np.random.default_rng(6).random((6144, 4096), dtype=np.float32)
for spread, starts in (("leaves", [0, 128, 256]), ("blocks", [0, 3 * 512 + 128, 7 * 512])):
    M = M6.copy()
    for c in starts:
        for q in range(3): M[:, c + 12 * q + 7] = 0.5 * M[:, c + 12 * q + 2]
    show(f"6144x4096 spread over {spread}", M, M6, 128, outer_block=512)
