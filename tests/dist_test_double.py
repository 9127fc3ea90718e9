"""CPU test double of the per-rank engine used by mixedprecisionblockqr_amd.dist.factor().

TEST INFRASTRUCTURE: same step interface as GpuEngine, arithmetic from the CPU oracle (oracle/), column
partition from the C ABI's host-only helpers.  Lets the distributed schedule run on gloo ranks without GPUs.
"""
import numpy as np
import torch

from mixedprecisionblockqr_amd import _lib as L
from mixedprecisionblockqr_amd.dist import global_columns
from oracle import pyoracle as po


class OracleEngine:
    def __init__(self, m, n, r, world, rank, outer_block=64, flag_once_on_rank=-1):
        self.m, self.n, self.r, self.world, self.rank = m, n, r, world, rank
        ko = max(outer_block, 32)
        self.ko = r if r >= ko else (ko // r) * r
        self.cols = global_columns(n, self.ko, world, rank)
        self.qcols = global_columns(m, self.ko, world, rank)
        self.A0 = np.zeros((m, len(self.cols)), np.float32)
        self.VT = {}
        self.log = []                      # call sequence, for the look-ahead order test
        self.robust = False                # set_robust(): every tall leaf on the column-by-column kernels (the oracle has only those)
        self._flag_pending = flag_once_on_rank == rank     # test hook: this rank reports a flagged leaf after its first pass

    def block(self): return self.ko
    def num_blocks(self): return (self.n + self.ko - 1) // self.ko
    def owner(self, s): return s % self.world
    def local_cols(self): return len(self.cols)
    def local_q_cols(self): return len(self.qcols)

    def set_local(self, A_loc): self.A0 = np.ascontiguousarray(A_loc, np.float32)
    def generate(self, seed=1234): self.A0 = po.generate(self.m, self.n, seed)[:, self.cols].copy()
    def local_absmax(self): return float(np.abs(self.A0).max()) if self.A0.size else 0.0

    def begin(self, absmax):
        self.A = self.A0.copy(); self.VT = {}; self.vd = np.zeros(self.n, np.float32)

    def _range(self, s): return s * self.ko, min(self.n, (s + 1) * self.ko)

    def factor_block(self, s):
        self.log.append(("factor_block", s))
        c0, c1 = self._range(s)
        assert self.owner(s) == self.rank
        lc0 = L.lib().mpqr_part_local_index(c0, self.ko, self.world)
        w = c1 - c0
        tmp = np.zeros((self.m + 1, self.n), np.float32)            # oracle works on (m+1) x n with global indices
        tmp[:self.m, c0:c1] = self.A[:, lc0:lc0 + w]
        r = self.r
        for l in range(c0, c1, r):                                   # r-wide panels inside the block (Cuda/qr.cu:1075-1080)
            t = min(c1, l + r)
            po.lib().orc_householder_qr(tmp, self.m, self.n, l, t - l)
            if t < c1:
                V = po.extract_V(tmp, self.m, self.n, l, t - l); T = po.compact_T(tmp, self.m, self.n, l, t - l)
                blk = tmp[l:self.m, t:c1]
                blk -= V @ (T.T @ (V.T @ blk))
        V = po.extract_V(tmp, self.m, self.n, c0, w)                 # (m-c0) x w, unit-norm reflectors
        T = po.compact_T(tmp, self.m, self.n, c0, w)
        # local columns keep R (rows <= col) and the reflectors in natural (unshifted) rows + diagonal aside
        out = self.A[:, lc0:lc0 + w]
        for j in range(w):
            k = c0 + j
            out[:k + 1, j] = tmp[:k + 1, k]
            out[k + 1:, j] = tmp[k + 2:self.m + 1, k]
            self.vd[k] = tmp[k + 1, k] if k + 1 <= self.m else 0.0
        self.VT[s] = (V, T)

    def block_bytes(self, s):
        c0, c1 = self._range(s); w = c1 - c0
        return 4 * ((self.m - c0) * w + w * w)

    def buffer(self, nbytes): return torch.empty(nbytes, dtype=torch.uint8)

    def pack(self, s, buf):
        self.log.append(("pack", s))
        V, T = self.VT[s]
        raw = np.concatenate([V.ravel(), T.ravel()]).astype(np.float32).view(np.uint8)
        buf.copy_(torch.from_numpy(raw.copy()))

    def unpack(self, s, buf):
        self.log.append(("unpack", s))
        c0, c1 = self._range(s); w = c1 - c0
        f = buf.numpy().view(np.float32)
        V = f[:(self.m - c0) * w].reshape(self.m - c0, w).copy(); T = f[(self.m - c0) * w:].reshape(w, w).copy()
        self.VT[s] = (V, T)

    def update(self, s):
        self.update_part(s, 2)

    def update_part(self, s, part):
        """part 0: the columns of block s+1 only; part 1: the local columns right of block s+1; part 2: everything"""
        self.log.append(("update_part", s, part))
        c0, c1 = self._range(s)
        c2 = self._range(s + 1)[1] if s + 1 < self.num_blocks() else self.n
        lc0 = L.lib().mpqr_part_local_cols(min(self.n, c1), self.ko, self.world, self.rank)
        lc1 = L.lib().mpqr_part_local_cols(min(self.n, c2), self.ko, self.world, self.rank)
        lo, hi = {0: (lc0, lc1), 1: (lc1, self.A.shape[1]), 2: (lc0, self.A.shape[1])}[part]
        V, T = self.VT[s]
        blk = self.A[c0:, lo:hi]
        if blk.shape[1]:
            blk -= V @ (T.T @ (V.T @ blk))

    def form_q(self):
        Q = np.zeros((self.m, len(self.qcols)), np.float32)
        Q[self.qcols, np.arange(len(self.qcols))] = 1.0
        for s in reversed(range(self.num_blocks())):
            c0, _ = self._range(s)
            lq = L.lib().mpqr_part_local_cols(min(self.m, c0), self.ko, self.world, self.rank)
            V, T = self.VT[s]
            blk = Q[c0:, lq:]
            if blk.shape[1]:
                blk -= V @ (T @ (V.T @ blk))
        self.Q = Q

    def sync(self): pass

    def flagged(self):
        """GpuEngine.flagged(): 1 if one of this rank's Gram-Householder leaves flagged itself (here: the test hook, once)."""
        self.log.append(("flagged",))
        f = 1 if (self._flag_pending and not self.robust) else 0
        self._flag_pending = False
        return f

    def set_robust(self, on):
        self.log.append(("set_robust", bool(on)))
        self.robust = bool(on)

    def local_factor(self):
        out = np.zeros((self.m + 1, len(self.cols)), np.float32)
        for j, gc in enumerate(self.cols):
            out[:gc + 1, j] = self.A[:gc + 1, j]
            if gc + 1 <= self.m: out[gc + 1, j] = self.vd[gc]
            out[gc + 2:, j] = self.A[gc + 1:, j]
        return out

    def local_q(self): return self.Q
    def local_input(self): return self.A0
