#!/usr/bin/env python3
"""Can the far trailing update run on e4m3 operands inside the 1e-3 tolerance?  

CPU emulation (NumPy) of ONE far update  A2 <- A2 - V T'^T (V^T A2)  on the operands the library would feed the MX-scaled MFMA
(v_mfma_scale_f32_32x32x64_f8f6f4): each operand is cut into k e4m3 TERMS (hi, hi + lo, hi + lo + lo2), every term with its own
power-of-two scale per block of 32 along K (the MX block scale), products accumulated in fp32-like (fp64 here) precision.  A k x k' term
product costs k * k' (or the leading min-order subset) fp8 MFMAs per fp16 MFMA it replaces; the e4m3 rate is 2x the fp16 rate, so a split
pays only while its product count is < 2.

Prints, for the operand splits below, the relative error of the update against the exact one, next to the fp16-operand error, and the
MFMA cost relative to fp16.  The reflectors and the matrix come from the oracle (the same seeded generator the tests use).
usage: python tools/fp8_split_study.py [m n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po


def e4m3_terms(x, nterms, axis_k):
    """x (2-D, float64) as a sum of `nterms` e4m3 arrays, each scaled per block of 32 along axis_k by a power of two (MX scaling)."""
    x = np.moveaxis(x, axis_k, -1)
    K = x.shape[-1]
    pad = (-K) % 32
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(0, pad)])
    blocks = xp.reshape(xp.shape[:-1] + (-1, 32))
    rest = blocks.copy()
    total = np.zeros_like(blocks)
    for _ in range(nterms):
        amax = np.abs(rest).max(axis=-1, keepdims=True)
        e = np.where(amax > 0, np.floor(np.log2(np.maximum(amax, 1e-300))), 0.0)
        scale = 2.0 ** (7 - e)                                  # block maximum into [128, 256): e4m3's top binade below its max (448)
        q = po.round_e4m3((rest * scale).astype(np.float32)).astype(np.float64) / scale
        total += q
        rest = rest - q
    out = total.reshape(xp.shape)[..., :K]
    return np.moveaxis(out, -1, axis_k)


def main():
    m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 1024)
    go, pw = 0, 128
    A = po.generate(m, n, seed=3)
    Ac = po.padded(A)
    po.lib().orc_householder_qr(Ac, m, n, go, pw)
    V = po.extract_V(Ac, m, n, go, pw).astype(np.float64)
    T = po.compact_T(Ac, m, n, go, pw).astype(np.float64)
    A2 = Ac[go:m, go + pw:].astype(np.float64)
    exact = A2 - V @ (T.T @ (V.T @ A2))
    nrm = np.linalg.norm(exact)

    def update(Vq, A2q, quantY):
        X = A2q.T @ Vq
        Y = X @ T
        return A2 - Vq @ quantY(Y).T

    rows = []
    Vh = po.round_fp16(V).astype(np.float64)
    rows.append(("fp16 operands (the library's default)", 1.0,
                 np.linalg.norm(update(Vh, po.round_fp16(A2).astype(np.float64), lambda Y: po.round_fp16(Y).astype(np.float64)) - exact) / nrm))
    for kv, ka, ky in ((1, 1, 1), (2, 1, 1), (2, 2, 2), (3, 2, 2), (3, 3, 3)):
        Vq = e4m3_terms(V, kv, 0)                               # V: K runs over rows in X = A2^T V, over reflectors in V Y^T (block scales along rows here)
        A2q = e4m3_terms(A2, ka, 0)
        err = np.linalg.norm(update(Vq, A2q, lambda Y: e4m3_terms(Y, ky, 1)) - exact) / nrm
        # fp8 MFMAs per fp16 MFMA: the two GEMMs need kv * ka and kv * ky term products; the e4m3 instruction does 2x the fp16 one's flops
        cost = 0.5 * (kv * ka + kv * ky) / 2.0
        rows.append((f"e4m3, {kv} term(s) of V x {ka} of A2 / {ky} of Y, per-32 block scales", cost, err))
    print(f"one far update, {m} x {n}, 128 reflectors: relative error of the update, MFMA time relative to the fp16 GEMMs")
    for name, cost, err in rows:
        print(f"  {name:72s} cost {cost:4.2f} x   error {err:.2e}")


if __name__ == "__main__":
    main()
