"""Pin the CPU oracle (oracle/oracle_qr.c) before anything trusts it.

Anchors, in order of strength:
  1. outputs of the REAL reference C++/main.cpp (built by oracle/Makefile from
     /root/reference; committed as vectors in tests/golden/pyref_qr.npz and, when
     oracle/_ref/libref_cppmain.so is present, called live);
  2. outputs of the reference's python/qr.py + wy.py (same golden file);
  3. the known answer of the reference's Cuda/qr.cu host path recorded in SURVEY.md 8c.
"""
import numpy as np
import pytest

from conftest import REF_SWEEP


def _flip_last_col_if_square(Q, R):
    """python/qr.py:48-50 skips the last reflector of a square matrix; the CUDA/C++
    paths apply it (R[n-1,n-1] and the last column of Q change sign)."""
    m, n = R.shape
    if m == n:
        Q = Q.copy(); R = R.copy()
        Q[:, n - 1] *= -1
        R[n - 1, :] *= -1
    return Q, R


def test_known_answer_cuda_host_path(po):
    # reference h_householder_qr + h_q_backward_accumulation on the classic 3x3
    A = np.array([[12, -51, 4], [6, 167, -68], [-4, 24, -41]], np.float32)
    Ao, Q, R = po.householder_qr(A)
    np.testing.assert_allclose(R, [[-14, -21, 14], [0, -175, 70], [0, 0, 35]], atol=2e-4)
    np.testing.assert_allclose(Q[:, 2], [-0.331429, 0.034286, -0.942857], atol=2e-6)
    np.testing.assert_allclose(Ao[3], [-0.148250, 0.055470, -1.0], atol=2e-6)   # shifted storage, extra row


def test_f64_port_matches_cppmain_golden(po, golden):
    for name in golden["cppmain_names"]:
        A = golden[f"{name}__A"]
        Q, R = po.qr_factorization_f64(A)
        np.testing.assert_allclose(Q, golden[f"cppmain__{name}__Q"], atol=1e-12, err_msg=name)
        np.testing.assert_allclose(R, golden[f"cppmain__{name}__R"], atol=1e-10, err_msg=name)


def test_f64_port_config1_256(po, golden):
    A = po.generate(256, 256, seed=1234).astype(np.float64)
    Q, R = po.qr_factorization_f64(A)
    np.testing.assert_allclose(np.diag(R), golden["cppmain__c1_256__diagR"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(R[0], golden["cppmain__c1_256__R_row0"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(Q[:, 0], golden["cppmain__c1_256__Q_col0"], atol=1e-12)
    # float-truncated sigma leaves ~1e-6 below the diagonal (C++/main.cpp:6), as the reference does
    assert abs(np.linalg.norm(np.tril(R, -1)) - golden["cppmain__c1_256__err"][2]) < 1e-9


def test_f64_port_matches_live_reference(po):
    if po.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    for n, seed in ((3, 1), (17, 2), (64, 3)):
        A = po.generate(n, n, seed=seed).astype(np.float64) - 0.5
        Qr, Rr = po.ref_qr_factorization(A)
        Q, R = po.qr_factorization_f64(A)
        np.testing.assert_allclose(Q, Qr, atol=1e-12)
        np.testing.assert_allclose(R, Rr, atol=1e-12)


def test_fp32_path_matches_cppmain_golden(po, golden):
    # same sign convention (last column reflected) -> element-wise comparable
    for name in golden["cppmain_names"]:
        A = golden[f"{name}__A"].astype(np.float32)
        _, Q, R = po.householder_qr(A)
        s = np.abs(A).max()
        np.testing.assert_allclose(Q, golden[f"cppmain__{name}__Q"], atol=2e-5, err_msg=name)
        np.testing.assert_allclose(R, np.triu(golden[f"cppmain__{name}__R"]), atol=2e-5 * s * len(A), err_msg=name)


def test_fp32_path_matches_python_reference(po, golden):
    for name in golden["names"]:
        A = golden[f"{name}__A"]
        m, n = A.shape
        if name in ("rank1_3x3", "zero_rows_3x3", "diag_3x3"):
            continue  # python skips allclose-zero columns with a tolerance (qr.py:54); covered below
        Ao, Q, R = po.householder_qr(A.astype(np.float32))
        Qp, Rp = _flip_last_col_if_square(golden[f"{name}__Q"], golden[f"{name}__R"])
        tol = 3e-5 * max(1.0, np.abs(A).max()) * max(m, n) ** 0.5
        np.testing.assert_allclose(Q, Qp, atol=tol, err_msg=name)
        np.testing.assert_allclose(R, np.triu(Rp), atol=tol, err_msg=name)
        # reflectors: python V[:, j] is the padded unit vector of column j
        Vp = golden[f"{name}__V"]
        V = po.extract_V(Ao, m, n, 0, n)
        k = Vp.shape[1]
        np.testing.assert_allclose(V[:, :k], Vp, atol=tol, err_msg=name)


def test_wy_matches_python_reference(po, golden):
    # python/wy.py: Q = I - W Y^T with W[:,0] = 2 v0, z = 2 (I - W Y^T) v  == Cuda/qr.cu:337-426
    for name in ("int5x3", "u12x8", "u60x40", "u129x80"):
        A = golden[f"{name}__A"].astype(np.float32)
        m, n = A.shape
        Ao = po.padded(A)
        po.lib().orc_householder_qr(Ao, m, n, 0, n)
        Qp = po.wy_transform(Ao, m, n, 0, n)
        W, Y = golden[f"{name}__W"], golden[f"{name}__Y"]
        np.testing.assert_allclose(Qp, np.eye(m) - W @ Y.T, atol=2e-5, err_msg=name)
        # compact-WY restatement: W = V T
        T = po.compact_T(Ao, m, n, 0, n)
        V = po.extract_V(Ao, m, n, 0, n)
        np.testing.assert_allclose(V @ T, W, atol=2e-5, err_msg=name)
        assert np.allclose(np.diag(T), 2.0, atol=1e-5) and np.allclose(np.tril(T, -1), 0)


def test_degenerate_inputs(po, golden):
    # python/test_data.py:38-57: rank deficient, diagonal, zero rows.  The CUDA host path
    # skips exactly-zero columns (qr.cu:242-244) and must stay finite.
    for name in ("rank1_3x3", "diag_3x3", "zero_rows_3x3"):
        A = golden[f"{name}__A"].astype(np.float32)
        for variant in ("dense", "compact32"):
            _, Q, R = po.block_qr(A, 2, variant)
            assert np.isfinite(Q).all() and np.isfinite(R).all()
            assert np.abs(Q @ R - A).max() < 1e-5
            assert np.abs(Q.T @ Q - np.eye(3)).max() < 1e-5
    Z = np.zeros((5, 3), np.float32)
    _, Q, R = po.block_qr(Z, 2, "compact32")
    assert (R == 0).all() and np.allclose(Q, np.eye(5))


@pytest.mark.parametrize("m,n,r", [c for c in REF_SWEEP if c[0] <= 240])
def test_block_variants_agree_and_meet_reference_criteria(po, m, n, r):
    A = po.generate(m, n, seed=1234)
    _, Q0, R0 = po.householder_qr(A)
    for variant, bits in (("dense", 23), ("compact32", 23), ("mixed", 11), ("compact16", 11)):
        _, Q, R = po.block_qr(A, r, variant)
        mt = po.metrics(A, R, Q)
        for key in ("backward_error", "q_error_max_signed", "lower_trapezoid"):
            assert po.lib().orc_error_passes(mt[key], m, bits), (variant, key, mt)
        tol = 2e-5 if bits == 23 else 6e-3
        assert np.abs(Q - Q0).max() < tol * max(1, m ** 0.5 / 4), variant
        assert np.abs(R - R0).max() < tol * (m ** 0.5), variant
    # north-star tolerance for the fp16-operand path
    _, Q, R = po.block_qr(A, r, "compact16")
    assert po.metrics(A, R, Q)["backward_error_f64"] <= 1e-3


def test_flops_model_and_fp16_rounding(po):
    L = po.lib()
    # Cuda/qr.cu:102-113 in fp32: (4 m^2 n - m n^2 + n^3/3)/s
    got = L.orc_qr_flops_per_second(2.0, 600, 400)
    want = (4 * 600.0 ** 2 * 400 - 600 * 400.0 ** 2 + 400.0 ** 3 / 3) / 2e-3
    assert abs(got - want) / want < 1e-6
    xs = np.concatenate([np.random.default_rng(0).standard_normal(2000).astype(np.float32) * s
                         for s in (1e-8, 1e-5, 1e-3, 1, 100, 7e4)])
    with np.errstate(over="ignore"):
        ours = np.array([L.orc_round_fp16(float(x)) for x in xs], np.float32)
        ref = xs.astype(np.float16).astype(np.float32)
    assert np.array_equal(ours, ref)


def test_jacobian_reader_roundtrip(po, tmp_path):
    # format of Cuda/qr.cu:696-776: "<rows> <cols>" then "<row> <col> <val>", leading blanks ok,
    # later duplicates overwrite
    p = tmp_path / "A_000000100.txt"
    p.write_text("4 3\n  0 0 1.5\n1 2 -2.25e-3\n   3 1 7\n1 2 4.0\n")
    M = po.read_euroc_jacobian(str(p))
    want = np.zeros((4, 3), np.float32); want[0, 0] = 1.5; want[1, 2] = 4.0; want[3, 1] = 7
    assert np.array_equal(M, want)
    q = tmp_path / "B.txt"
    assert po.lib().orc_write_euroc_jacobian(str(q).encode(), 4, 3, want) == 0
    assert np.array_equal(po.read_euroc_jacobian(str(q)), want)
    with pytest.raises(IOError):
        po.read_euroc_jacobian(str(tmp_path / "missing.txt"))
