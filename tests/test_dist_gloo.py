"""N > 1 path on CPU: two gloo ranks run the SAME schedule (mixedprecisionblockqr_amd.dist.factor) that the GPU
ranks run, with the oracle-backed test double as the engine, and must reproduce the single-process factors."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, m, n, r, ko, q, la=True, flag_rank=-1):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mixedprecisionblockqr_amd import dist as mpdist
    from dist_test_double import OracleEngine
    eng = OracleEngine(m, n, r, world, rank, outer_block=ko, flag_once_on_rank=flag_rank)
    eng.generate(1234)
    comm = mpdist.TorchComm()
    mpdist.factor(eng, comm, lookahead=la)
    chk = mpdist.residual_check(eng, comm)
    q.put((rank, eng.cols, eng.qcols, eng.local_factor(), eng.local_q(), chk, eng.log))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("m,n,r,ko,world,la", [(96, 80, 16, 32, 2, True), (130, 100, 8, 32, 2, True), (64, 64, 16, 32, 3, True),
                                                (130, 100, 8, 32, 2, False)])
def test_two_rank_schedule_matches_serial(po, m, n, r, ko, world, la):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + n + (7 if la else 0)
    procs = [ctx.Process(target=_worker, args=(rk, world, port, m, n, r, ko, q, la)) for rk in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    A = po.generate(m, n, seed=1234)
    A0, Q0, R0 = po.householder_qr(A)
    F = np.zeros((m + 1, n), np.float32); Q = np.zeros((m, m), np.float32)
    seen_a, seen_q = [], []
    for rank, cols, qcols, Fl, Ql, chk, log in got:
        F[:, cols] = Fl; Q[:, qcols] = Ql
        seen_a += list(cols); seen_q += list(qcols)
        assert chk["backward_error_est"] < 1e-5 and chk["q_shard_orth_fro"] < 1e-4
        if la:
            # look-ahead order on the owner of block s+1: its own block's columns first, the rest of update s
            # ENQUEUED before the factorisation of block s+1, which is packed before anybody can ask for broadcast s+1
            nb = (n + ko - 1) // ko
            for s in range(nb - 1):
                if (s + 1) % world != rank:
                    assert ("factor_block", s + 1) not in log
                    assert ("update_part", s, 0) not in log          # non-owners have nothing of block s+1
                    continue
                i0, i1 = log.index(("update_part", s, 0)), log.index(("update_part", s, 1))
                i2, i3 = log.index(("factor_block", s + 1)), log.index(("pack", s + 1))
                assert i0 < i1 < i2 < i3, (rank, s, log)
                if s + 2 < nb:
                    assert i3 < log.index(("unpack", s + 1))
    assert sorted(seen_a) == list(range(n)) and sorted(seen_q) == list(range(m))
    # the N > 1 error report IS the north-star metric ||A - QR||_F / ||A||_F (Cuda/qr.cu:115-135), estimated with 32 Gaussian probes:
    # it must agree with the exact value of the assembled result (round 3 reported it divided by sqrt(n))
    R = np.triu(F[:m]).astype(np.float64)
    exact = np.linalg.norm(A - Q.astype(np.float64) @ R) / np.linalg.norm(A)
    for _, _, _, _, _, chk, _ in got:
        assert chk["probes"] >= 16 and 0.6 * exact <= chk["backward_error_est"] <= 1.6 * exact, (chk, exact)
    np.testing.assert_allclose(F, A0, atol=3e-5 * np.sqrt(m))          # R and shifted reflectors, all columns
    np.testing.assert_allclose(Q, Q0, atol=3e-5 * np.sqrt(m))


def test_single_rank_degenerates_to_serial(po):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mixedprecisionblockqr_amd import dist as mpdist
    from dist_test_double import OracleEngine
    eng = OracleEngine(70, 50, 8, 1, 0, outer_block=32)
    eng.generate(5)
    mpdist.factor(eng, mpdist.NullComm())
    A0, Q0, _ = po.householder_qr(po.generate(70, 50, seed=5))
    np.testing.assert_allclose(eng.local_factor(), A0, atol=2e-5)
    np.testing.assert_allclose(eng.local_q(), Q0, atol=2e-5)


def test_flagged_rank_makes_every_rank_repeat_the_factorisation_on_the_robust_kernels(po):
    """No rank synchronises inside the block loop (round 4): the leaf flags are asked once, after the loop, and ANY rank's flag
    (all-reduce max) makes EVERY rank repeat the factorisation with its tall leaves on the column-by-column kernels -- the broadcasts
    are collective, so the ranks must take the second pass together."""
    m, n, r, ko, world = 96, 80, 16, 32, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + 333
    procs = [ctx.Process(target=_worker, args=(rk, world, port, m, n, r, ko, q, True, 1)) for rk in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    A = po.generate(m, n, seed=1234)
    A0, Q0, _ = po.householder_qr(A)
    F = np.zeros((m + 1, n), np.float32)
    for rank, cols, qcols, Fl, Ql, chk, log in got:
        F[:, cols] = Fl
        assert log.count(("flagged",)) == 1 and ("set_robust", True) in log, (rank, log)         # both ranks, although only rank 1 flagged
        assert sum(1 for e in log if e[0] == "factor_block") == 2 * sum(1 for s in range((n + ko - 1) // ko) if s % world == rank)   # two passes
        assert chk["backward_error_est"] < 1e-5
    np.testing.assert_allclose(F, A0, atol=3e-5 * np.sqrt(m))
