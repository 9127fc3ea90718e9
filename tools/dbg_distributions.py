"""Backward error on other input distributions than the U[0,1) of the benchmarks (Gaussian, scaled, log-normal)."""
import os, sys, math, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixedprecisionblockqr_amd import api
rng = np.random.default_rng(3)
for (m, n, kind) in [(4096, 4096, "randn"), (4096, 4096, "uniform"), (8192, 2048, "randn"), (4096, 4096, "randn*1e3"), (4096, 4096, "lognormal")]:
    if kind == "randn": A = rng.standard_normal((m, n), dtype=np.float32)
    elif kind == "uniform": A = rng.random((m, n), dtype=np.float32)
    elif kind == "randn*1e3": A = (rng.standard_normal((m, n)) * 1e3).astype(np.float32)
    else: A = np.exp(rng.standard_normal((m, n))).astype(np.float32)
    h = api.Handle(); h.plan(m, n, 128); h.set_matrix(A); h.factor(); mt = h.metrics(); tm = h.timings()
    print("%5d x %-5d %-10s backward %.3e  qfro %.3e  passes %d" % (m, n, kind, mt["backward_error"], mt["q_error_fro"], tm["n_passes"]), flush=True)
    del h
