"""Child process of test_opt_in_schedules_and_kernels: the library reads its MPQR_* switches once per process, so every
opt-in path (alternative schedules and kernels kept as measured experiments) is exercised in a fresh interpreter.
Prints one JSON line: backward error, orthogonality, |diag R| (sign-free, comparable across schedules)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixedprecisionblockqr_amd import api

m, n, r = (int(x) for x in sys.argv[1:4])
A = api.generate_matrix(m, n, seed=77)
Ab = np.zeros((m + 1, n), np.float32); Ab[:m] = A
Q = np.zeros((m, m), np.float32)
api.dev_mixed_precision_block_qr(Ab, Q, m, n, r)
R = api.h_strip_R_from_A(Ab, m, n)
A64 = A.astype(np.float64)
be = float(np.linalg.norm(A64 - Q.astype(np.float64) @ R.astype(np.float64)) / np.linalg.norm(A64))
oe = float(np.abs(Q.astype(np.float64).T @ Q.astype(np.float64) - np.eye(m)).max())
print(json.dumps({"backward_error": be, "orth_max": oe, "absdiag": [float(abs(x)) for x in np.diag(R[:n])]}))
