"""Child process of test_opt_in_schedules_and_kernels: the library reads its MPQR_* switches once per process, so every
opt-in path (alternative schedules and kernels kept as measured experiments) is exercised in a fresh interpreter.
Prints one JSON line: backward error, orthogonality, |diag R| (sign-free, comparable across schedules)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixedprecisionblockqr_amd import api

m, n, r = (int(x) for x in sys.argv[1:4])
outer_block = int(sys.argv[4]) if len(sys.argv) > 4 else 0
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 77
dump = sys.argv[6] if len(sys.argv) > 6 else None         # .npz path: the factor and Q for an element-wise comparison in the parent
A = api.generate_matrix(m, n, seed=seed)
Ab = np.zeros((m + 1, n), np.float32); Ab[:m] = A
Q = np.zeros((m, m), np.float32)
api.dev_mixed_precision_block_qr(Ab, Q, m, n, r, outer_block=outer_block)
tm = api.default_handle().timings()
if dump:
    np.savez(dump, Ab=Ab, Q=Q)
R = api.h_strip_R_from_A(Ab, m, n)
A64 = A.astype(np.float64)
be = float(np.linalg.norm(A64 - Q.astype(np.float64) @ R.astype(np.float64)) / np.linalg.norm(A64))
oe = float(np.abs(Q.astype(np.float64).T @ Q.astype(np.float64) - np.eye(m)).max())
print(json.dumps({"backward_error": be, "orth_max": oe, "absdiag": [float(abs(x)) for x in np.diag(R[:n])],
                  "n_far_launches": tm["n_far_launches"], "gbytes_far_nn": tm["gbytes_far_nn"], "n_q_ident_rows": tm["n_q_ident_rows"], "n_q_launches": tm["n_q_launches"],
                  "n_fused_leaves": tm["n_fused_leaves"], "n_tpoll_retries": tm["n_tpoll_retries"]}))
