import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixedprecisionblockqr_amd import api
shapes = [(m, 5120, 128) for m in (5120, 5376, 5632, 6144, 7168, 8192, 10240, 15360)] + [(16384, n, 128) for n in (2048, 4096, 8192, 12288)]
for (m, n, r) in shapes:
    h = api.Handle(); h.plan(m, n, r); h.generate(42); h.factor(); mt = h.metrics()
    print("%6d x %-6d r=%-3d  backward %.4e  qfro %.3e" % (m, n, r, mt["backward_error"], mt["q_error_fro"]), flush=True)
    del h
