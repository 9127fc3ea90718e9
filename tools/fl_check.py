#!/usr/bin/env python3
"""Fused leaf (MPQR_FUSED_LEAF) against the seven-launch leaf of round 4 on the same inputs: backward error, orthogonality, R differences,
step times.  usage: fl_check.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mixedprecisionblockqr_amd as mp
from mixedprecisionblockqr_amd.api import Handle

shapes = [(2048, 2048, 64), (4096, 4096, 128), (2320, 1980, 64), (6144, 4096, 128), (3000, 1500, 128), (8192, 2048, 128)]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    shapes = [(16384, 16384, 128)]
for (m, n, r) in shapes:
    res = {}
    for fused in (0, 1):
        os.environ["MPQR_FUSED_LEAF"] = str(fused)
        h = Handle()
        h.plan(m, n, r)
        h.generate(1234)
        h.factor(); h.sync()
        t0 = time.time()
        for _ in range(3):
            h.factor(); h.sync()
        ms = (time.time() - t0) / 3 * 1e3
        mt = h.metrics(); tm = h.timings()
        R = h.r_matrix() if m <= 8192 else None
        res[fused] = (mt, tm, R, ms)
        h.close()
    (m0, t0_, R0, ms0), (m1, t1_, R1, ms1) = res[0], res[1]
    dr = float(np.linalg.norm(R1 - R0) / np.linalg.norm(R0)) if R0 is not None else -1
    print(f"{m}x{n} r={r}: old be={m0['backward_error']:.3e} qf={m0['q_error_fro']:.3e} {ms0:.2f} ms panel {t0_['ms_panel']:.2f} | "
          f"fused be={m1['backward_error']:.3e} qf={m1['q_error_fro']:.3e} {ms1:.2f} ms panel {t1_['ms_panel']:.2f} passes {t1_['n_passes']} | dR={dr:.2e}", flush=True)
