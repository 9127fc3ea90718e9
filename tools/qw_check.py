#!/usr/bin/env python3
"""Backward error of synthetic Jacobians of growing size (this process's MPQR_* settings).  usage: qw_check.py"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixedprecisionblockqr_amd import api
for cams, pts, seed in ((40, 580, 1234), (40, 580, 7), (100, 1500, 1234), (100, 1500, 5), (30, 1200, 3)):
    J = api.synthetic_jacobian(cams=cams, points=pts, seed=seed)
    m, n = J.shape
    Ab = np.zeros((m + 1, n), np.float32); Ab[:m] = J
    Q = np.zeros((m, m), np.float32)
    api.dev_mixed_precision_block_qr(Ab, Q, m, n, 64)
    R = api.h_strip_R_from_A(Ab, m, n)
    A64 = J.astype(np.float64)
    be = float(np.linalg.norm(A64 - Q.astype(np.float64) @ R.astype(np.float64)) / np.linalg.norm(A64))
    oe = float(np.abs(Q.astype(np.float64).T @ Q.astype(np.float64) - np.eye(m)).max())
    print(json.dumps({"env": os.environ.get("MPQR_QW", "1"), "m": m, "n": n, "seed": seed, "backward_error": be, "orth_max": oe}), flush=True)
