"""Multi-GPU block QR: the host-side schedule over the step-wise C ABI (include/mpqr.h, mpqr_dist_*).

One process per GPU.  Column superblocks are dealt round-robin (1-D block-cyclic); per block the owner factors
and packs [V^T | T | T^T], the buffer is broadcast with torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests), every rank updates its own trailing columns; Q is column-sharded and needs no
communication.  The reference has no multi-GPU path: this is the sharded form of its block loop.

`factor(engine, comm)` is written against a small engine interface so that the same schedule runs
  * on GPUs   : GpuEngine (ctypes -> libmpqr.so), comm = TorchComm
  * in tests  : a CPU test double under tests/ (never shipped, never imported from here).
"""
import ctypes as C
import json
import os
import time

import numpy as np

from . import _lib as L
from .api import MpqrError, _opts


class NullComm:
    world, rank = 1, 0

    def allreduce_max(self, x):
        return x

    def broadcast(self, buf, root):
        pass

    def barrier(self):
        pass


class TorchComm:
    """torch.distributed plumbing: broadcast of the packed block buffer and two scalar reductions."""

    def __init__(self, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.device = device

    def allreduce_max(self, x):
        import torch
        t = torch.tensor([float(x)], dtype=torch.float32, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def broadcast(self, buf, root):
        self.dist.broadcast(buf, src=root)

    def barrier(self):
        self.dist.barrier()


class GpuEngine:
    """One rank's handle on its GPU (all compute through the C ABI; torch only owns the broadcast buffer)."""

    def __init__(self, device, m, n, r, world, rank, **opt_kw):
        import torch
        self.torch = torch
        self.m, self.n, self.r, self.world, self.rank = m, n, r, world, rank
        self.device = torch.device("cuda", device)
        self._h = C.c_void_p()
        rc = L.lib().mpqr_create(C.byref(self._h), int(device))
        if rc != L.OK:
            raise MpqrError(rc, L.lib().mpqr_last_error(None).decode())
        o = _opts(**opt_kw)
        self._chk(L.lib().mpqr_dist_plan(self._h, m, n, r, world, rank, C.byref(o)))
        self._bufs = {}
        self._payload = [None, None]          # two payload buffers of the largest block's size, reused by every factor() call
        sp = C.c_void_p()
        self._chk(L.lib().mpqr_dist_chain_stream(self._h, C.byref(sp)))
        self._chain = torch.cuda.ExternalStream(sp.value, device=self.device)    # the library's chain stream, as torch sees it

    def _chk(self, rc):
        if rc != L.OK:
            raise MpqrError(rc, L.lib().mpqr_last_error(self._h).decode())

    def close(self):
        if self._h:
            L.lib().mpqr_destroy(self._h)
            self._h = C.c_void_p()

    # --- partition facts
    def block(self): return L.lib().mpqr_dist_block(self._h)
    def num_blocks(self): return L.lib().mpqr_dist_num_blocks(self._h)
    def owner(self, s): return L.lib().mpqr_dist_block_owner(self._h, s)
    def local_cols(self): return L.lib().mpqr_dist_local_cols(self._h)
    def local_q_cols(self): return L.lib().mpqr_dist_local_q_cols(self._h)

    # --- data
    def generate(self, seed=1234): self._chk(L.lib().mpqr_dist_generate_matrix(self._h, seed))

    def set_local(self, A_loc):
        A_loc = np.ascontiguousarray(A_loc, np.float32)
        self._chk(L.lib().mpqr_dist_set_local_matrix_host(self._h, A_loc.ctypes.data_as(C.c_void_p), max(A_loc.shape[1], 1)))

    def local_absmax(self):
        v = C.c_float()
        self._chk(L.lib().mpqr_dist_local_absmax(self._h, C.byref(v)))
        return v.value

    # --- steps
    def begin(self, absmax): self._chk(L.lib().mpqr_dist_begin(self._h, absmax))
    def factor_block(self, s): self._chk(L.lib().mpqr_dist_factor_block(self._h, s))
    def block_bytes(self, s): return L.lib().mpqr_dist_block_bytes(self._h, s)

    def buffer(self, nbytes):
        if nbytes not in self._bufs:
            self._bufs[nbytes] = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)
        return self._bufs[nbytes]

    def payload(self, slot, nbytes):
        """View of payload buffer `slot` (0 / 1): block sizes shrink with s, the two buffers are sized once for the largest."""
        if self._payload[slot] is None:
            cap = max(self.block_bytes(s) for s in range(self.num_blocks()))
            self._payload[slot] = self.torch.empty(cap, dtype=self.torch.uint8, device=self.device)
        return self._payload[slot][:nbytes]

    # pack / unpack are enqueued on the library's chain stream; torch's current stream (on which torch.distributed orders the
    # broadcast) is ordered against it with events, the host never waits for a broadcast
    def pack(self, s, buf):
        self._chk(L.lib().mpqr_dist_pack_block_async(self._h, s, C.c_void_p(buf.data_ptr())))
        self.torch.cuda.current_stream(self.device).wait_stream(self._chain)

    def before_broadcast(self):
        self.torch.cuda.current_stream(self.device).wait_stream(self._chain)      # the buffer's previous contents have been unpacked

    def unpack(self, s, buf):
        self._chain.wait_stream(self.torch.cuda.current_stream(self.device))      # the broadcast has landed
        self._chk(L.lib().mpqr_dist_unpack_block_async(self._h, s, C.c_void_p(buf.data_ptr())))

    def flagged(self):
        """1 if a Gram-Householder leaf of this rank's blocks found its columns too ill conditioned (synchronises)."""
        v = C.c_int()
        self._chk(L.lib().mpqr_dist_flags(self._h, C.byref(v)))
        return v.value

    def set_robust(self, on): self._chk(L.lib().mpqr_dist_set_robust(self._h, 1 if on else 0))

    def update(self, s): self._chk(L.lib().mpqr_dist_update(self._h, s))
    def update_part(self, s, part): self._chk(L.lib().mpqr_dist_update_part(self._h, s, part))

    def timings(self):
        t = L.MpqrTimings()
        self._chk(L.lib().mpqr_get_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in L.MpqrTimings._fields_ if k != "reserved"}
    def form_q(self): self._chk(L.lib().mpqr_dist_form_q(self._h))
    def sync(self): self._chk(L.lib().mpqr_sync(self._h))

    # --- results
    def _get(self, fn, rows, cols):
        out = np.empty((rows, cols), np.float32)
        self._chk(fn(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def local_factor(self): return self._get(L.lib().mpqr_dist_get_local_factor_host, self.m + 1, self.local_cols())
    def local_q(self): return self._get(L.lib().mpqr_dist_get_local_q_host, self.m, self.local_q_cols())
    def local_input(self): return self._get(L.lib().mpqr_dist_get_local_input_host, self.m, self.local_cols())


def factor(engine, comm, form_q=True, lookahead=True):
    """The distributed factorisation: the block loop below, and -- only if a tall leaf somewhere flagged itself as too ill conditioned for
    the Gram-Householder kernels (asked ONCE, after the loop: no rank synchronises its host per block) -- the same loop again with
    every tall leaf on the column-by-column kernels."""
    _factor_once(engine, comm, form_q, lookahead)
    if hasattr(engine, "flagged") and comm.allreduce_max(engine.flagged()) > 0:
        # (the robust mode belongs to the INPUT: a later factor() of the same matrix would flag again, so it starts robust at once; a new
        #  input -- mpqr_dist_set_local_matrix_host / mpqr_dist_generate_matrix -- clears it in the library)
        engine.set_robust(True)
        _factor_once(engine, comm, form_q, lookahead)


def _factor_once(engine, comm, form_q=True, lookahead=True):
    """The distributed block loop.  Every rank calls this with its own engine.

    Look-ahead: after block s has been broadcast, the owner of block s+1 updates THAT block's columns
    first (update_part(s, 0), chain stream), starts the update of its other columns on the far-update stream
    (update_part(s, 1)), and factors and packs block s+1 while that update -- and everybody else's -- is still running;
    the other ranks go straight to the next broadcast and wait there with their GPUs busy.  The packed buffers are
    double-buffered.  With one rank nothing is packed or broadcast: the schedule degenerates to the single-GPU one."""
    engine.begin(comm.allreduce_max(engine.local_absmax()))
    nb = engine.num_blocks()
    multi = comm.world > 1
    bufs = {}

    def buffer(s):                                # double-buffered payloads: two buffers, reused (engines without `payload`: tests)
        nbytes = engine.block_bytes(s)
        if hasattr(engine, "payload"):
            return engine.payload(s % 2, nbytes)
        key = (s % 2, nbytes)
        if key not in bufs:
            bufs[key] = engine.buffer(nbytes)
        return bufs[key]

    if comm.rank == engine.owner(0):
        engine.factor_block(0)
        if multi:
            engine.pack(0, buffer(0))
    for s in range(nb):
        owner = engine.owner(s)
        if multi:
            buf = buffer(s)
            if hasattr(engine, "before_broadcast"):
                engine.before_broadcast()
            comm.broadcast(buf, owner)
            engine.unpack(s, buf)
        if not lookahead:
            engine.update_part(s, 2)
            if s + 1 < nb and comm.rank == engine.owner(s + 1):
                engine.factor_block(s + 1)
                if multi:
                    engine.pack(s + 1, buffer(s + 1))
            continue
        if s + 1 < nb and comm.rank == engine.owner(s + 1):
            engine.update_part(s, 0)           # the next block's columns first ...
            engine.update_part(s, 1)           # ... the rest is enqueued on the far stream and runs beside ...
            engine.factor_block(s + 1)         # ... the factorisation of the next block (chain stream)
            if multi:
                engine.pack(s + 1, buffer(s + 1))
        else:
            engine.update_part(s, 1)
    if form_q:
        engine.form_q()
    engine.sync()


def global_columns(ncols, block, world, rank):
    """Global indices of this rank's columns (local order)."""
    l = L.lib()
    k = l.mpqr_part_local_cols(ncols, block, world, rank)
    return np.array([l.mpqr_part_global_index(lc, block, world, rank) for lc in range(k)], dtype=np.int64)


def residual_check(engine, comm, nvec=32, seed=7, group=None):
    """The north-star error metric ||A - QR||_F / ||A||_F without gathering the matrices, by Gaussian probing: for x ~ N(0, I_n),
    E ||(A - QR) x||^2 = ||A - QR||_F^2, so with nvec probes X
        backward_error_est = ||A X - Q (R X)||_F / (sqrt(nvec) ||A||_F)        (relative spread ~ 1 / sqrt(2 nvec): 12 % at 32)
    is the quantity the single-GPU path computes exactly (Cuda/qr.cu:115-135); `randomized_residual` (round 3) divided by ||X||_F
    instead, i.e. reported that number / sqrt(n).  Also ||Q_loc^T Q_loc - I||_F summed over the shards.
    A x, R x and Q y are sums over column shards (all-reduced m-vectors)."""
    import torch
    import torch.distributed as dist
    m, n = engine.m, engine.n
    blk = engine.block()
    cols = global_columns(n, blk, comm.world, comm.rank)
    qcols = global_columns(m, blk, comm.world, comm.rank)
    A = torch.from_numpy(engine.local_input()).double()
    F = engine.local_factor()
    R = torch.from_numpy(np.where(np.arange(m)[:, None] <= cols[None, :], F[:m], 0.0)).double()
    Q = torch.from_numpy(engine.local_q()).double()
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, nvec, generator=g, dtype=torch.float64)
    Ax = A @ X[cols]; y = R @ X[cols]
    a2 = torch.tensor([float((A * A).sum())], dtype=torch.float64)
    if comm.world > 1:
        for t in (Ax, y, a2):
            dist.all_reduce(t, group=group)         # host tensors: `group` must be a gloo group (default group in the CPU tests)
    Qy = Q @ y[qcols]
    if comm.world > 1:
        dist.all_reduce(Qy, group=group)
    be = float(torch.linalg.norm(Ax - Qy) / (np.sqrt(nvec) * torch.sqrt(a2)))
    res = float(torch.linalg.norm(Ax - Qy) / (torch.sqrt(a2) * torch.linalg.norm(X)))
    # ||Q_loc^T Q_loc - I||_F: exactly for small shards, by Gaussian probing (E ||(Q^T Q - I) z||^2 = ||.||_F^2, 16 probes)
    # for large ones -- the exact product is 2 m q^2 flops in fp64 on the host, minutes at bench sizes
    q = Q.shape[1]
    if m * q * q <= 2e9:
        qe2 = float(((Q.T @ Q - torch.eye(q, dtype=torch.float64)) ** 2).sum())
    else:
        Z = torch.randn(q, 16, generator=g, dtype=torch.float64)
        qe2 = float(((Q.T @ (Q @ Z) - Z) ** 2).sum() / 16.0)
    qe = torch.tensor([qe2], dtype=torch.float64)
    if comm.world > 1:
        dist.all_reduce(qe, group=group)
    return {"backward_error_est": be, "probes": int(nvec), "randomized_residual": res, "q_shard_orth_fro": float(torch.sqrt(qe))}


def bench_main(args, m, n, r, world, rank, local_rank, cpu_baseline_fn=None):
    """bench.py leg for N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL); MPQR_FORCE_DIST=1 runs it at N = 1."""
    import torch
    import torch.distributed as dist
    # MPQR_DIST_REHEARSE=gloo: every rank on GPU 0 with gloo as the transport -- a one-GPU box runs the real multi-process schedule
    # (ownership, broadcast / unpack ordering against the library's streams, sharded Q) where no second GPU exists for RCCL
    rehearse = os.environ.get("MPQR_DIST_REHEARSE", "") == "gloo"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    own_pg = not dist.is_initialized()
    if world > 1 and own_pg:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    comm = TorchComm(device=dev) if world > 1 else NullComm()
    # distribution unit = the far update's aggregation width; with many ranks keep at least two blocks per rank so that
    # every rank still owns trailing columns late in the factorisation (config 5: 8192 columns on 8 GPUs -> 512)
    ko = args.outer_block
    if not ko and world > 1:
        ko = max(r, min(1024, (n // (2 * world)) // r * r))
    prec_name = getattr(args, "precision", None) or ("fp8" if getattr(args, "config", "") == "c5" else "fp16")
    precision = {"fp16": L.PREC_FP16, "fp8": L.PREC_FP8, "fp32": L.PREC_FP32}[prec_name]
    eng = GpuEngine(local_rank, m, n, r, world, rank, outer_block=ko, lookahead=not args.no_lookahead, precision=precision)
    eng.generate(1234)
    eng.sync()

    for _ in range(args.warmup):
        factor(eng, comm, lookahead=not args.no_lookahead)
    comm.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        factor(eng, comm, lookahead=not args.no_lookahead)
    comm.barrier(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if world > 1:
        dtt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(dtt, op=dist.ReduceOp.MAX)
        dt = float(dtt.item())

    # host-side verification on a gloo side group (the m-vectors stay on the host)
    chk = None
    g = None
    try:
        g = dist.new_group(backend="gloo") if world > 1 else None
        chk = residual_check(eng, comm, group=g)
    except Exception as e:  # verification must never take the benchmark down
        chk = {"error": repr(e)}
    # every rank's own HIP-event timings of the last step, gathered on rank 0 (max / sum over ranks below)
    tm = eng.timings()
    tms = [tm]
    if world > 1:
        try:
            tms = [None] * world
            dist.all_gather_object(tms, tm, group=g)
        except Exception:
            tms = [tm]
    if rank == 0:
        from . import api
        fl = api.flops(m, n, r)
        roof = None
        if tm["n_far_launches"] > 0 and tm["ms_far_nn"] > 0:
            ach = tm["flops_far_nn"] / (tm["ms_far_nn"] * 1e-3) / 1e12
            peak = 5000.0 if prec_name == "fp8" else 2500.0
            kname = ("gemm8_fp8_kernel<E_SUB_F32> (far A2 -= V*Y^T on rank 0's column shard, e4m3 x e4m3 -> fp32, K = outer block)" if prec_name == "fp8" else
                     "gemm6_f16_kernel<E_SUB_F32> (far A2 -= V*Y^T on rank 0's column shard, fp16 x fp16 -> fp32, K = outer block)")
            roof = {"bound": "mfma", "kernel": kname,
                    "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                    "launches": tm["n_far_launches"], "avg_launch_ms": tm["ms_far_nn"] / tm["n_far_launches"],
                    "tn_achieved": (tm["flops_far_tn"] / (tm["ms_far_tn"] * 1e-3) / 1e12) if tm["ms_far_tn"] > 0 else None}
        out = {
            "metric": "GFLOP/s block QR (%s MFMA trailing)" % prec_name, "value": fl["geqrf"] / dt / 1e9, "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": {"fp16": "f16xf16->f32 (fp32 panel)", "fp32": "f32 (exact-f32 MFMA)",
                      "fp8": "e4m3xe4m3->f32 far trailing update (fp32 panel; fp16 in-block updates and Q formation)"}[prec_name],
            "data": "synthetic U[0,1) fp32, seed 1234",
            "config": {"workload": f"{m}x{n} random dense, block={r}, full Q formed", "m": m, "n": n, "block": r,
                       "outer_block": eng.block(),
                       "parallelism": f"{world} gpu(s), 1-D block-cyclic column superblocks, look-ahead, RCCL broadcast of V,T per block"},
            "error": chk,
            "breakdown_ms_rank0": {k: tm[k] for k in ("ms_total", "ms_factor", "ms_form_q", "ms_trailing", "ms_panel", "ms_far_tn", "ms_far_nn")},
            # max over ranks of every phase (the step is as long as its slowest rank), and the panel chain summed over the ranks: block t + 1
            # cannot start before block t's reflectors exist, so the owners' chains add up (the quantity tools/scale_model.py budgets)
            "breakdown_ms_max_over_ranks": {k: max(t_[k] for t_ in tms) for k in ("ms_total", "ms_factor", "ms_form_q", "ms_trailing", "ms_panel", "ms_far_tn", "ms_far_nn")},
            "ms_panel_sum_over_ranks": sum(t_["ms_panel"] for t_ in tms), "ranks_reporting": len(tms),
            "comm": {"backend": (dist.get_backend() if world > 1 else "none"), "world_size": (dist.get_world_size() if world > 1 else 1),
                     "note": "backend nccl = RCCL on ROCm; one broadcast of [V^T | T | T^T] per top-level block"},
            "gflops_with_q_flops": (fl["geqrf"] + fl["form_q"]) / dt / 1e9,
            "roofline": roof,
        }
        # the panel step against the HBM roof, as at N = 1 (2 W r 4 bytes per panel): chain time = the owners' chains added up
        pbytes = sum(2.0 * (m - lam) * min(r, n - lam) * 4 for lam in range(0, n, r))
        psum = sum(t_["ms_panel"] for t_ in tms)
        if roof is not None and psum > 0:
            roof["panel"] = {"bound": "hbm", "algorithmic_bytes": pbytes, "ms": psum, "achieved": pbytes / (psum * 1e-3) / 1e9, "peak": 8000.0,
                             "unit": "GB/s", "frac": pbytes / (psum * 1e-3) / 1e9 / 8000.0, "note": "ms = sum over ranks of ms_panel (the owners' chains are serial)"}
        if cpu_baseline_fn is not None and not args.no_cpu_baseline:
            out["cpu_baseline"], port = cpu_baseline_fn()
            if port is not None:
                out["cpu_baseline_port"] = port
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if own_pg:
            dist.destroy_process_group()
