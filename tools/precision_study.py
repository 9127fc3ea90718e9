#!/usr/bin/env python3
"""Precision study: backward error ||A - QR||_F / ||A||_F versus condition number for the three
arithmetic paths of the library -- mixed (fp16 MFMA operands, fp32 panel), fp32 twin, fp64 (C++/main.cpp path) -- next to
LAPACK in fp64.  Same experiment and table layout as the reference's python/performance_test_result/error.md
(n in {10, 100, 500}, condition 1e3 .. 1e7).  Test matrices: symmetric positive definite with a prescribed condition
number, A = P P^T, P = U diag(s) V^T with log-spaced s (construction of Bierlaire, Toint, Tuyttens 1991, the one the
reference's generate_matrix follows), fixed seed.

usage (GPU box):  python tools/precision_study.py [out.md]"""
import sys

import numpy as np

sys.path.insert(0, ".")


def spd_with_condition(n, cond, rng):
    s = np.exp(np.linspace(-np.log(cond) / 4.0, np.log(cond) / 4.0, n))       # cond(P) = sqrt(cond), cond(P P^T) = cond
    U, _ = np.linalg.qr(rng.standard_normal((n, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    P = (U * s) @ V.T
    return P @ P.T


def backward_error(A, Q, R):
    A = A.astype(np.float64)
    return float(np.linalg.norm(A - Q.astype(np.float64) @ R.astype(np.float64)) / np.linalg.norm(A))


def main():
    import mixedprecisionblockqr_amd as mp
    rng = np.random.default_rng(2024)
    rows = []
    for n in (10, 100, 500):
        for p in (3, 4, 5, 6, 7):
            A = spd_with_condition(n, 10.0 ** p, rng)
            A32 = A.astype(np.float32)
            r = min(32, n)
            errs = []
            for fn in (mp.dev_mixed_precision_block_qr, mp.dev_block_qr_wy):
                Ao = np.zeros((n + 1, n), np.float32); Ao[:n] = A32
                Q = np.zeros((n, n), np.float32)
                fn(Ao, Q, n, n, r)
                errs.append(backward_error(A32, Q, mp.h_strip_R_from_A(Ao, n, n)))
            Q64, R64 = mp.qr_factorization(A)
            errs.append(backward_error(A, Q64, R64))
            Qn, Rn = np.linalg.qr(A)
            errs.append(backward_error(A, Qn, Rn))
            rows.append((n, p, errs))
    lines = ["| (n, condition_num) | mixed (fp16 MFMA) | fp32 twin | fp64 (GPU) | numpy(lapack) qr float64 |",
             "|:------------------:|:-----------------:|:---------:|:----------:|:------------------------:|"]
    for n, p, e in rows:
        lines.append("| (%d, '10^%d') | %.2e | %.2e | %.2e | %.2e |" % (n, p, e[0], e[1], e[2], e[3]))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(
            "# Precision study (tools/precision_study.py; layout of the reference's error.md)\n\n"
            "Backward error ||A - QR||_F / ||A||_F; SPD test matrices with prescribed condition number, seed 2024, r = min(32, n).\n"
            "The reference's fp16 column is NaN from condition 1e6 on; the mixed path here scales its operands by a power of two\n"
            "and keeps the panel in fp32, so it stays at the fp16 operand accuracy (~1e-3) for every condition number.\n\n" + text + "\n")


if __name__ == "__main__":
    main()
