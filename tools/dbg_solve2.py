import sys, numpy as np
sys.path.insert(0, ".")
import torch; torch.cuda.init()
import mixedprecisionblockqr_amd as mp
from mixedprecisionblockqr_amd import dist as mpdist
h = mp.Handle(0)
for (m, n, r, ko) in [(2600, 1536, 128, 512), (2600, 1536, 128, 0), (2600, 1536, 128, 256), (2048, 1024, 128, 512)]:
    h.plan(m, n, r, outer_block=ko); h.generate(1234); h.factor(); h.sync()
    mt = h.metrics(); t = h.timings()
    print("single", m, n, r, ko, "be %.3e qe %.3e passes %d robust %d" % (mt["backward_error"], mt["q_error_fro"], t["n_passes"], t["n_robust_leaves"]))
for world in (1, 2):
    m, n, r, ko = 2600, 1536, 128, 512
    engines = [mpdist.GpuEngine(0, m, n, r, world, rk, outer_block=ko) for rk in range(world)]
    for e in engines: e.generate(1234)
    sys.path.insert(0, "tests")
    import test_gpu_parity as T
    T._run_lockstep(mp, engines, lookahead=True)
    F = np.zeros((m + 1, n), np.float32); Q = np.zeros((m, m), np.float32)
    A = np.zeros((m, n), np.float32)
    for e in engines:
        c = mpdist.global_columns(n, e.block(), world, e.rank)
        F[:, c] = e.local_factor(); A[:, c] = e.local_input()
        Q[:, mpdist.global_columns(m, e.block(), world, e.rank)] = e.local_q()
    R = np.triu(F[:m])
    print("dist world", world, "be %.3e" % (np.linalg.norm(A - Q.astype(np.float64) @ R) / np.linalg.norm(A)), "tim", engines[0].timings()["n_passes"])
    for e in engines: e.close()
