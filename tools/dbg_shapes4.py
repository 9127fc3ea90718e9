import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixedprecisionblockqr_amd import api
for (m, n, r, seed) in [(6144, 5120, 128, 1), (6144, 5120, 128, 2), (6144, 5120, 128, 3), (7168, 5120, 128, 1), (7168, 5120, 128, 2), (16384, 4096, 128, 1), (16384, 4096, 128, 2), (5120, 5120, 128, 1), (5120,5120,128,2)]:
    h = api.Handle(); h.plan(m, n, r); h.generate(seed); h.factor(); mt = h.metrics(); tm = h.timings()
    print("%6d x %-6d seed %d  backward %.4e  qfro %.3e passes %d robust %d" % (m, n, seed, mt["backward_error"], mt["q_error_fro"], tm["n_passes"], tm["n_robust_leaves"]), flush=True)
    del h
