import sys, time, numpy as np
sys.path.insert(0, '.')
import mixedprecisionblockqr_amd as mp
for (m, n, r) in [(16384, 16384, 128), (65536, 8192, 256)]:
    h = mp.Handle(0)
    h.plan(m, n, r, form_q=False)
    h.generate(1234); h.factor(); h.sync()
    t = h.timings()
    rng = np.random.default_rng(1)
    for nrhs in (1, 16):
        Y = rng.standard_normal((m, nrhs)).astype(np.float32)
        mp.solve_ls(Y, handle=h)
        t0 = time.perf_counter(); X = mp.solve_ls(Y, handle=h); dt = time.perf_counter() - t0
        print(m, n, r, "factor(no Q) ms", round(t["ms_factor"], 2), "nrhs", nrhs, "solve ms (incl. host copies)", round(dt * 1e3, 2), "finite", bool(np.isfinite(X).all()))
    h.close()
