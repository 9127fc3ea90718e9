#!/usr/bin/env python3
"""Numerical prototype (numpy, CPU) of the Gram-level look-ahead of the panel chain.

Baseline = what the library does today inside a block: leaf j is factored from the Gram matrix of the ACTUAL
(fp16-GEMM-updated) columns; then every remaining column of the block is updated through fp16 operands.

Look-ahead = leaf j+1's inputs (N = Gram of its remaining rows, B = its top block) are PREDICTED from leaf j's small
results and a Gram pass over data that does not contain leaf j's update yet:
    Z = V_top^T E + C^T Glow_x,   Y = T^T Z,   Rx = E - V_top Y,   B' = D - (F C) Y,   N' = Glow_d + E^T E - Rx^T Rx
and the tall rows of the next leaf's columns receive exactly that update (A_low -= V_low Y, fp32) while the other
columns keep the fp16 path.  The question this answers: does the predicted N stay consistent with the data (backward
error, orthogonality) at fp16-level targets.  usage: proto_lookahead.py [m n leaf]"""
import sys
import numpy as np

f16 = lambda x: x.astype(np.float16).astype(np.float32)


def solve_model(N, B):
    """exact-arithmetic model of gh_solve: N = Gram of all leaf rows (fp64), B = top w x w block.
    returns R (w x w), V_top (w x w, unit 2-norm^2 = 2 convention dropped: LAPACK tau form folded into T later), C."""
    w = B.shape[0]
    B64 = B.astype(np.float64)
    Glow = N - B64.T @ B64
    L = np.linalg.cholesky(Glow).T                      # A_low = Q_low L
    M = np.vstack([B64, L])                             # 2w x w, same R and same reflector coefficients
    V = np.zeros_like(M); R = M.copy()
    for k in range(w):
        x = R[k:, k].copy()
        nu = np.linalg.norm(x)
        s = 1.0 if x[0] >= 0 else -1.0
        u = x.copy(); u[0] += s * nu
        u /= np.linalg.norm(u) / np.sqrt(2.0)           # ||v||^2 = 2:  H = I - v v^T
        V[k:, k] = u
        R[k:, :] -= np.outer(u, u @ R[k:, :])
    Rtop = np.triu(R[:w])
    Vtop = V[:w]
    C = np.linalg.solve(L, V[w:])                       # V_low = A_low C
    return Rtop.astype(np.float32), Vtop.astype(np.float32), C.astype(np.float32)


def t_from_v(Vh):
    S = Vh.astype(np.float64).T @ Vh.astype(np.float64)
    return np.linalg.inv(np.triu(S, 1) + np.diag(np.diag(S)) / 2).astype(np.float32)


def update_fp16(A2, Vh, T):
    """A2 <- (I - V T V^T)^T A2 with fp16 operands, fp32 accumulation (the library's op1..op3)."""
    X = f16(A2).T @ Vh                                  # op1
    Y = f16(X @ T)                                      # op2: Y = X T   (columns x reflectors)
    return A2 - Vh @ Y.T                                # op3


def run(A0, w, lookahead):
    m, n = A0.shape
    A = A0.copy()
    nl = n // w
    Vh_all = np.zeros((m, n), np.float32); Ts = []
    Npred = Bpred = None
    for j in range(nl):
        c0, c1 = j * w, (j + 1) * w
        if Npred is None:
            rows = A[c0:, c0:c1].astype(np.float64)
            N = rows.T @ rows; B = A[c0:c1, c0:c1].copy()
        else:
            N, B = Npred, Bpred
        R, Vtop, C = solve_model(N, B)
        Alow = A[c1:, c0:c1].copy()
        Vlow = (Alow @ C).astype(np.float32)
        V = np.vstack([Vtop, Vlow])
        Vh = f16(V)
        T = t_from_v(Vh); Ts.append(T)
        Vh_all[c0:, c0:c1] = Vh
        A[c0:c1, c0:c1] = R; A[c1:, c0:c1] = 0
        Npred = Bpred = None
        if c1 >= n: break
        if lookahead and j + 1 < nl:
            d0, d1 = c1, c1 + w
            # Gram pass over the next leaf's columns BEFORE leaf j's update (rows below leaf j's top block)
            Anx = A[c0:, d0:d1].copy()
            E = Anx[:w]; lowx = Anx[w:].astype(np.float64)
            Glow_x = Alow.astype(np.float64).T @ lowx
            Glow_d = lowx.T @ lowx
            Glow_jj = N - B.astype(np.float64).T @ B.astype(np.float64)
            f32 = lambda x: x.astype(np.float32)
            S = f32(Vtop.T @ Vtop) + f32(C.T @ f32(f32(Glow_jj) @ C))
            Tx = np.linalg.inv(np.triu(S, 1).astype(np.float64) + np.diag(np.diag(S)) / 2).astype(np.float32)
            Z = f32(Vtop.T @ E) + f32(C.T @ f32(Glow_x))
            Y = f32(Tx.T @ Z)
            Rx = E - f32(Vtop @ Y)
            F = Alow[:w]; D = Anx[w:2 * w]
            Bn = D - f32(f32(F @ C) @ Y)
            Nn = Glow_d + E.astype(np.float64).T @ E.astype(np.float64) - Rx.astype(np.float64).T @ Rx.astype(np.float64)
            # the data: next leaf's columns get exactly this update in fp32, the rest of the block the fp16 path
            A[c0:c1, d0:d1] = Rx
            A[d0:d1, d0:d1] = Bn
            A[d1:, d0:d1] = Anx[2 * w:] - f32(Vlow[w:] @ Y)
            if d1 < n: A[c0:, d1:] = update_fp16(A[c0:, d1:], Vh, T)
            Npred, Bpred = Nn, Bn
        else:
            A[c0:, c1:] = update_fp16(A[c0:, c1:], Vh, T)
    # Q from the fp16 reflectors and their T's (fp64 product: isolates the factorisation's own error)
    Q = np.eye(m)
    for j in reversed(range(len(Ts))):
        c0 = j * w
        V = Vh_all[c0:, c0:c0 + w].astype(np.float64)
        Q[c0:, c0:] -= V @ (Ts[j].astype(np.float64) @ (V.T @ Q[c0:, c0:]))
    Rm = np.triu(A)[:m]
    be = np.linalg.norm(A0 - Q @ Rm) / np.linalg.norm(A0)
    oe = np.linalg.norm(Q.T @ Q - np.eye(m))
    return be, oe


if __name__ == "__main__":
    m, n, w = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 1024, 128)
    rng = np.random.default_rng(1)
    A0 = rng.random((m, n), dtype=np.float32) * 128
    for la in (False, True):
        be, oe = run(A0, w, la)
        print("lookahead=%d  backward error %.3e  ||Q^T Q - I||_F %.3e" % (la, be, oe))
