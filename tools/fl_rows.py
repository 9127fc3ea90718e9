import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mixedprecisionblockqr_amd.api import Handle
m, n, r = 2048, 2048, 64
Rs = []
for fused in (0, 1):
    os.environ["MPQR_FUSED_LEAF"] = str(fused)
    h = Handle(); h.plan(m, n, r); h.generate(1234); h.factor(); h.sync()
    Rs.append(h.r_matrix()); h.close()
d = np.linalg.norm(Rs[1] - Rs[0], axis=1) / (np.linalg.norm(Rs[0], axis=1) + 1e-30)
bad = np.nonzero(d > 1e-2)[0]
print("rows differing > 1e-2:", len(bad), bad[:40])
ds = np.linalg.norm(np.abs(Rs[1]) - np.abs(Rs[0]), axis=1) / (np.linalg.norm(Rs[0], axis=1) + 1e-30)
print("after abs(): rows > 1e-2:", int((ds > 1e-2).sum()), "max", ds.max())
