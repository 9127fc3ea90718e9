import sys, numpy as np
sys.path.insert(0, ".")
import mixedprecisionblockqr_amd as mp
from oracle import pyoracle as po
def relF(X, Y): return np.linalg.norm(X - Y) / np.linalg.norm(Y)
m, n, go, pw = 1024, 768, 128, 128
A = po.generate(m, n, seed=3)
Ac = po.padded(A); po.lib().orc_householder_qr(Ac, m, n, go, pw)
h = mp.Handle(0)
res = {}
for name, prec in (("fp16", mp.PREC_FP16), ("fp8", mp.PREC_FP8)):
    Ag = Ac.copy(); mp.apply_panel_to_trailing(Ag, m, n, go, pw, precision=prec, handle=h); res[name] = Ag[go:m, go + pw:].astype(np.float64)
V = po.extract_V(Ac, m, n, go, pw).astype(np.float64)
A2 = Ac[go:m, go + pw:].astype(np.float64)
exact = A2 - V @ (po.compact_T(Ac, m, n, go, pw).astype(np.float64).T @ (V.T @ A2))
for k, g in res.items():
    du, de = A2 - g, A2 - exact
    c = (du * de).sum() / (de * de).sum()
    print(k, "relF vs exact", relF(g, exact), "update ratio", c, "relF of update after rescale", relF(du / c, de))
