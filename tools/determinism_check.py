"""tools/determinism_check.py M N R [REPS] -- factor the same device-resident matrix REPS times, compare R and the backward error bit for bit."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixedprecisionblockqr_amd as mp
m, n, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
h = mp.Handle(0); h.plan(m, n, r); h.generate(1234)
ref = None; bad = 0
for i in range(reps):
    h.factor(); h.sync()
    R = h.r_matrix(); be = h.metrics()["backward_error"]
    if ref is None: ref = R.copy()
    else:
        d = np.flatnonzero((R != ref).any(axis=0))
        if len(d): bad += 1; print(f"  run {i}: R differs from run 0 in {len(d)} columns, first {d[0]}, backward {be:.10e}")
print(os.environ.get("LABEL", ""), f"{m}x{n}: {reps} runs, {bad} differ from the first")
h.close()
