"""Shape sweep through the schedules added in round 2 (flat blocks, look-ahead across block boundaries, pairwise far
updates, Q formation on pairs with the fp16 shadow, one-shot Q for tall matrices): odd / even block counts, ragged last
blocks and leaves, row counts that are not multiples of the tile sizes.  Device-side metrics only (no host copies)."""
import os, sys, math, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixedprecisionblockqr_amd as mp
from mixedprecisionblockqr_amd import api

SHAPES = [(3072, 3072, 128), (5120, 5120, 128), (5000, 4900, 128), (4200, 3100, 128), (7168, 7168, 128), (6000, 5200, 128),
          (9000, 2900, 128), (10240, 3072, 128), (8200, 2100, 128), (12000, 2560, 128), (4096, 4096, 64), (6144, 5120, 256),
          (3500, 3400, 100), (2304, 2304, 128), (16384, 4096, 128)]
bad = 0
for (m, n, r) in SHAPES:
    h = api.Handle()
    h.plan(m, n, r)
    h.generate(42)
    h.factor()
    mt = h.metrics(); tm = h.timings()
    ok = mt["backward_error"] <= 1e-3 and mt["q_error_fro"] <= 2e-3 * math.sqrt(m) and mt["lower_trapezoid"] == 0.0
    bad += 0 if ok else 1
    print("%6d x %-6d r=%-3d  backward %.3e  ||QtQ-I||_F %.3e (bound %.3e)  passes %d  %s" %
          (m, n, r, mt["backward_error"], mt["q_error_fro"], 2e-3 * math.sqrt(m), tm["n_passes"], "ok" if ok else "FAIL"), flush=True)
    del h
print("failures:", bad)
sys.exit(1 if bad else 0)
