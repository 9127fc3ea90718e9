"""Generate tests/golden/pyref_qr.npz by importing the reference's Python prototype.

Run once in the build container (the reference tree does not travel):
    python tests/golden/gen_golden.py
It imports /root/reference/python/{qr,wy}.py (NumPy only) and records, for a set
of fixed inputs, the outputs of
    qr.householder_qr(A, mode='complete')   python/qr.py:27-70   -> Q (m x m), R (m x n)
    qr.householder_qr(A, mode='raw')        python/qr.py:69-70   -> V (list of padded unit reflectors), B (=2.0)
    wy.wy_representation(V, B)              python/wy.py:3-29    -> W, Y with Q = I - W Y^T
Inputs: the integer matrices the reference's own tests use (python/test_data.py:6-30,
:40-55 -- data, typed here), plus seeded uniform matrices at the shapes of the
reference's CUDA sweep (Cuda/qr.cu:1762-1783).  Only inputs and outputs are stored.
Conventions differ from the CUDA/C++ path in ONE place (python/qr.py:48-50): the last
column of a square matrix is not reflected; tests account for that.
"""
import os
import sys

import numpy as np

REF = "/root/reference/python"
sys.path.insert(0, REF)
import qr as refqr      # noqa: E402
import wy as refwy      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

FIXED = {
    "int5x3": [[1, 2, 3], [4, 5, 6], [7, 8, 7], [4, 2, 3], [4, 2, 2]],
    "int3x3_zero_lead": [[0, 3, 1], [0, 4, -2], [2, 1, 1]],
    "int3x3_classic": [[12, -51, 4], [6, 167, -68], [-4, 24, -41]],
    "int6x6": [[10, 20, 30, 40, 50, 60], [32, 32, 44, 55, 66, 35], [23, 66, 74, 64, 45, 65],
               [67, 28, 46, 26, 46, 42], [95, 95, 52, 88, 65, 11], [75, 53, 96, 47, 32, 32]],
    "rank1_3x3": [[1, 2, 3], [1, 2, 3], [1, 2, 3]],
    "diag_3x3": [[1, 0, 0], [0, 2, 0], [0, 0, 3]],
    "zero_rows_3x3": [[1, 2, 3], [0, 0, 0], [0, 0, 0]],
}
SEEDED = {"u6x4": (6, 4, 11), "u12x8": (12, 8, 12), "u24x16": (24, 16, 13), "u60x40": (60, 40, 14),
          "u80x80": (80, 80, 15), "u97x90": (97, 90, 16), "u129x80": (129, 80, 17)}


def main():
    out = {}
    names = []
    mats = {k: np.array(v, dtype=np.float64) for k, v in FIXED.items()}
    for k, (m, n, seed) in SEEDED.items():
        # float32-representable uniform values so fp32 and fp64 paths see identical inputs
        mats[k] = np.random.default_rng(seed).random((m, n), dtype=np.float32).astype(np.float64)
    for name, A in mats.items():
        Q, R = refqr.householder_qr(A.copy(), mode="complete")
        V, B = refqr.householder_qr(A.copy(), mode="raw")
        out[name + "__A"] = A
        out[name + "__Q"] = Q
        out[name + "__R"] = R
        if len(V):
            Vm = np.stack(V, axis=1)                 # m x k, column j = reflector j (zero padded)
            W, Y = refwy.wy_representation(V, B)
            out[name + "__V"] = Vm
            out[name + "__W"] = W
            out[name + "__Y"] = Y
        names.append(name)
    out["names"] = np.array(names)
    # ---- the real C++/main.cpp (oracle/_ref/libref_cppmain.so, built by oracle/Makefile
    # from /root/reference/C++/main.cpp + the vendored Eigen): qr_factorization outputs
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import pyoracle as po
    cpp = []
    for name in ("int3x3_zero_lead", "int3x3_classic", "int6x6", "u80x80"):
        Q, R = po.ref_qr_factorization(mats[name])
        out["cppmain__" + name + "__Q"] = Q
        out["cppmain__" + name + "__R"] = R
        cpp.append(name)
    out["cppmain_names"] = np.array(cpp)
    # config 1 of BASELINE.json: 256x256 (generator seed 1234); keep diag(R), row 0 of R,
    # column 0 of Q and the two error figures only (full matrices would be 1 MiB)
    A256 = po.generate(256, 256, seed=1234).astype(np.float64)
    Q, R = po.ref_qr_factorization(A256)
    out["cppmain__c1_256__diagR"] = np.diag(R).copy()
    out["cppmain__c1_256__R_row0"] = R[0].copy()
    out["cppmain__c1_256__Q_col0"] = Q[:, 0].copy()
    out["cppmain__c1_256__err"] = np.array([np.linalg.norm(A256 - Q @ R) / np.linalg.norm(A256),
                                            np.linalg.norm(Q.T @ Q - np.eye(256)),
                                            np.linalg.norm(np.tril(R, -1))])
    np.savez_compressed(os.path.join(HERE, "pyref_qr.npz"), **out)
    print("wrote", os.path.join(HERE, "pyref_qr.npz"), len(names), "cases")
    precision_fixtures()


def precision_fixtures():
    """inputs of the reference's precision experiment (python/performance_test.py:22-33) made by ITS
    generator python/utils.py:13-24 (generate_matrix, legacy global RNG seeded here), with the backward errors of ITS
    householder_qr in float32 / float64 (python/qr.py:27-70) on exactly these matrices -> pyref_precision.npz.
    The float16 column is not reproduced: explicit-H NumPy in fp16 takes minutes per matrix (duration.md)."""
    import utils as refutils          # /root/reference/python/utils.py
    out = {}
    cases = [(10, 3), (10, 5), (10, 7), (100, 3), (100, 5), (100, 7), (500, 5)]
    np.random.seed(20240)
    for n, p in cases:
        A = refutils.generate_matrix(n, 10.0 ** p)
        key = "n%d_c%d" % (n, p)
        out[key + "__A"] = A
        errs = []
        for dt in (np.float32, np.float64):
            Q, R = refqr.householder_qr(A.copy(), dtype=dt)
            errs.append(refutils.get_error(A, Q, R))
        out[key + "__ref_err_f32_f64"] = np.array(errs)
        print(key, "cond %.2e" % np.linalg.cond(A), "reference fp32 / fp64 error: %.2e %.2e" % tuple(errs))
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "pyref_precision.npz"), **out)
    print("wrote", os.path.join(HERE, "pyref_precision.npz"))


if __name__ == "__main__":
    main()
