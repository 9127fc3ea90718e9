"""N > 1 path on CPU: two gloo ranks run the SAME schedule (mixedprecisionblockqr_amd.dist.factor) that the GPU
ranks run, with the oracle-backed test double as the engine, and must reproduce the single-process factors."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, m, n, r, ko, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mixedprecisionblockqr_amd import dist as mpdist
    from dist_test_double import OracleEngine
    eng = OracleEngine(m, n, r, world, rank, outer_block=ko)
    eng.generate(1234)
    comm = mpdist.TorchComm()
    mpdist.factor(eng, comm)
    chk = mpdist.residual_check(eng, comm)
    q.put((rank, eng.cols, eng.qcols, eng.local_factor(), eng.local_q(), chk))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("m,n,r,ko,world", [(96, 80, 16, 32, 2), (130, 100, 8, 32, 2), (64, 64, 16, 32, 3)])
def test_two_rank_schedule_matches_serial(po, m, n, r, ko, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + n
    procs = [ctx.Process(target=_worker, args=(rk, world, port, m, n, r, ko, q)) for rk in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    A = po.generate(m, n, seed=1234)
    A0, Q0, R0 = po.householder_qr(A)
    F = np.zeros((m + 1, n), np.float32); Q = np.zeros((m, m), np.float32)
    seen_a, seen_q = [], []
    for rank, cols, qcols, Fl, Ql, chk in got:
        F[:, cols] = Fl; Q[:, qcols] = Ql
        seen_a += list(cols); seen_q += list(qcols)
        assert chk["randomized_residual"] < 1e-5 and chk["q_shard_orth_fro"] < 1e-4
    assert sorted(seen_a) == list(range(n)) and sorted(seen_q) == list(range(m))
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
