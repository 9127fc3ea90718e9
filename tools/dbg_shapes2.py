import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixedprecisionblockqr_amd import api
for (m, n, r) in [(6000, 5200, 128), (6144, 5120, 256), (16384, 4096, 128), (6144, 5120, 128)]:
    h = api.Handle(); h.plan(m, n, r); h.generate(42); h.factor(); mt = h.metrics()
    print("%6d x %-6d r=%-3d  backward %.4e  qfro %.3e" % (m, n, r, mt["backward_error"], mt["q_error_fro"]), flush=True)
    del h
