import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mixedprecisionblockqr_amd as mp
rng = np.random.default_rng(11)
m, n = 1500, 480
A = rng.standard_normal((m, n)).astype(np.float32)
A[:, 300] = 0
h = mp.Handle(0); h.plan(m, n, 32); h.set_matrix(A); h.factor(); h.sync()
Q = h.q().astype(np.float64); R = h.r_matrix().astype(np.float64)
F = h.factor_out()
E = A.astype(np.float64) - Q @ R
print("be", np.linalg.norm(E) / np.linalg.norm(A), "passes", h.timings()["n_passes"], "defl", h.timings()["n_deflated_columns"])
QtA = Q.T @ A.astype(np.float64)
k = 300
print("R[k, k:k+8]   ", np.round(R[k, k:k+8], 4))
print("QtA[k, k:k+8] ", np.round(QtA[k, k:k+8], 4))
print("R[k+1, k:k+8]   ", np.round(R[k+1, k:k+8], 4))
print("QtA[k+1, k:k+8] ", np.round(QtA[k+1, k:k+8], 4))
print("rows of QtA below diag norm (should be 0):", np.linalg.norm(np.tril(QtA[:n], -1)), " row k of tril:", np.linalg.norm(np.tril(QtA[:n], -1)[k]), "col k:", np.linalg.norm(QtA[k+1:, k]))
print("orth", np.abs(Q.T @ Q - np.eye(m)).max())
V = np.tril(F[1:m+1, :n], 0)   # shifted reflectors
print("||v_k||, v_k top entries", np.linalg.norm(V[:, k]), V[k:k+3, k], " ||v_{k+1}||", np.linalg.norm(V[:, k+1]))
# which rows of E are bad
rw = np.linalg.norm(E, axis=1); print("worst rows:", np.argsort(-rw)[:5], np.round(np.sort(-rw)[:5], 3))
print("E[k, k:k+8]", np.round(E[k, k:k+8], 4), " E[k, 470:480]", np.round(E[k, 470:480], 3))
