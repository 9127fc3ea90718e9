#!/usr/bin/env python3
"""bench.py -- headline benchmark: GFLOP/s of the mixed-precision block QR (A -> Q, R) on MI355X.

Workload (BASELINE.json configs[3], the configuration the metric's targets are quoted on): 16384 x 16384
uniform random fp32 matrix, panel width 128, fp32 panel + fp16-operand/fp32-accumulate MFMA trailing update,
full m x m Q formed (the reference's matrix-in / Q,R-out contract).  One "step" = copy the HBM-resident input
into the working matrix + factor + form Q.  value = (2 m n^2 - 2/3 n^3) / step time, i.e. GEQRF-equivalent flops only
(the Q-formation flops are NOT counted, so the figure is conservative); see DESIGN.md "Measurement".

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c4|c2|c1]
For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {  # name: (m, n, r)
    "c1": (256, 256, 32), "c2": (2048, 2048, 64), "c4": (16384, 16384, 128), "c5": (65536, 8192, 256),
    "c3": (2320, 1980, 64),     # synthetic bundle-adjustment Jacobian through the reference's text format (api.synthetic_jacobian)
}
PEAK_FP16_TFLOPS = 2500.0   # MI355X dense fp16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0    # MI355X dense fp8 (block-scaled MFMA), same table
PEAK_HBM_GBPS = 8000.0      # HBM3E, same table


def cpu_baseline():
    """The reference's CPU path timed on this box's host cores in the same run (north star / BASELINE.md section 4):
    the REAL C++/main.cpp:16-43 qr_factorization (oracle/_ref/libref_cppmain.so, built from the reference tree with its
    vendored Eigen; explicit m x m H per column, O(n^4), fp64, one thread) on a bounded sample of the same workload
    family -- 256 x 256 (BASELINE config 1) and 384 x 384, U[0,1) seed 1234.  If that library did not travel to this
    box the builder's own port is the only line (kind "port")."""
    import numpy as np
    from oracle import pyoracle as po
    import mixedprecisionblockqr_amd as mp
    port = cpu_baseline_port()
    if po.ref_lib() is None:
        return port, None
    times = {}
    for nn in (256, 384):
        A = po.generate(nn, nn, seed=1234).astype(np.float64)
        t0 = time.perf_counter()
        Q, R = po.ref_qr_factorization(A)
        times[nn] = time.perf_counter() - t0
        err = float(np.linalg.norm(A - Q @ R) / np.linalg.norm(A))
    nn = 384
    ref = {"value": mp.flops(nn, nn, 32)["geqrf"] / times[nn] / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "reference",
           "sample": f"C++/main.cpp qr_factorization (explicit H, fp64, Eigen 3.4.0, 1 thread): 256x256 in {times[256]:.2f} s, "
                     f"384x384 in {times[384]:.2f} s (value quoted on 384x384, GEQRF-equivalent flops; ||A-QR||/||A|| = {err:.1e})"}
    return ref, port


def cpu_baseline_port():
    """Second CPU line: the oracle's compact-WY block loop (oracle/oracle_qr.c, OpenMP over trailing columns), all host
    cores of this job's share, 2048 x 2048, r = 64 (BASELINE config 2)."""
    import numpy as np
    from oracle import pyoracle as po
    import mixedprecisionblockqr_amd as mp
    m = n = 2048; r = 64
    cores = min(len(os.sched_getaffinity(0)), 16)       # the GPU box gives one job a 16-core share
    os.environ["OMP_NUM_THREADS"] = str(cores)
    A = po.generate(m, n, seed=1234)
    Ao = po.padded(A); Q = np.eye(m, dtype=np.float32)
    t0 = time.perf_counter()
    po.lib(omp=True).orc_block_qr_compact(Ao, Q, m, n, r, 0)
    dt = time.perf_counter() - t0
    return {"value": mp.flops(m, n, r)["geqrf"] / dt / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": f"{m}x{n} r={r} fp32 compact-WY block loop incl. Q formation, {dt:.1f} s, GEQRF-equivalent flops"}


def gemm_source_sha():
    """sha256 (16 hex digits) of the sources the dominant GEMM kernel is built from: ties a committed PMC profile to a build."""
    import hashlib
    hsh = hashlib.sha256()
    # (every source the GEMM kernels are compiled from; mpqr_internal.h is left out since round 5: it also declares the panel kernels'
    #  launchers, and a change there would void a valid GEMM profile)
    for f in ("kernels_gemm2.hip", "kernels_gemm.hip", "gemm_body.h", "gemm_epilogue.h"):
        with open(os.path.join(ROOT, "mixedprecisionblockqr_amd", "csrc", f), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def _pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/*_traffic_rmw.json, tools/pmc_traffic.py); PMC counters cannot be read from inside the timed run.  Returns (bytes or None,
    source).  The figure is quoted only (1) when the profile was taken on THIS build's GEMM sources (`kernel_source_sha`) and (2) together
    with the algorithmic bytes and the launch count of exactly the launches it was measured on (the tool matches the PMC rows launch by
    launch against mpqr_get_update_records; round 4 compared 24 launches of a truncated pass with the algorithmic bytes of 37): `source`
    carries launches / algorithmic_bytes_per_launch / ratio, for the whole set and for the far updates and Q formation separately."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("traffic_rmw.json"):
                try:
                    best = (f, json.load(open(os.path.join(pdir, f))))
                except Exception:
                    pass
    if best is None:
        return None, {"file": None, "note": "no PMC profile of this build's format committed"}
    f, d = best
    sha = gemm_source_sha()
    src = {"file": "profiles/" + f, "profile_kernel_source_sha": d.get("kernel_source_sha"), "build_kernel_source_sha": sha,
           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python bench.py --steps 1 --warmup 0 --dump-records`, FETCH_SIZE x 2 "
                  "(gfx950), matched launch by launch against the library's records (tools/pmc_traffic.py)"}
    if d.get("kernel_source_sha") != sha:
        src["note"] = "profile taken on other GEMM sources than this build: not quoted"
        return None, src
    src.update({"launches": d.get("launches"), "complete": d.get("complete"), "algorithmic_bytes_per_launch": d.get("algorithmic_bytes_per_launch"),
                "ratio": d.get("ratio")})
    for k in ("far", "q"):
        if k in d:
            src[k] = {kk: d[k].get(kk) for kk in ("launches", "records", "complete", "measured_bytes", "algorithmic_bytes", "ratio", "dispatch_ids_fetch_pass")}
    return d.get("hbm_bytes_per_launch"), src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--outer-block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the PCIe-inclusive timing of the drop-in call (dropin_ms)")
    ap.add_argument("--no-alone", action="store_true", help="skip the MFMA-rate context measurement (roofline.mfma_measured)")
    ap.add_argument("--alone", action="store_true", help="also time the dominant GEMM kernels alone on the GPU at this run's shapes (roofline.kernel_alone); "
                    "off by default: those launches carry the same kernel names as the factorisation's and would enter a profiler's per-kernel averages")
    ap.add_argument("--no-lookahead", action="store_true")
    ap.add_argument("--dump-records", default=None, help="write the last step's recorded C -= V Y^T launches (launch order: M, N, K, algorithmic "
                    "flops and bytes; far updates and Q formation) to this JSON file: tools/pmc_traffic.py matches a PMC pass of the same command against it")
    ap.add_argument("--precision", default=None, choices=["fp16", "fp8", "fp32"],
                    help="operand precision of the trailing-update GEMMs (default: fp8 for c5 = BASELINE config 5, fp16 otherwise)")
    args = ap.parse_args()

    import torch
    import mixedprecisionblockqr_amd as mp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    m, n, r = CONFIGS[args.config]
    prec_name = args.precision or ("fp8" if args.config == "c5" else "fp16")
    if world > 1 or os.environ.get("MPQR_FORCE_DIST") == "1":     # the env switch lets a 1-GPU box exercise the RCCL leg
        from mixedprecisionblockqr_amd import dist as mpdist
        return mpdist.bench_main(args, m, n, r, world, rank, local_rank, cpu_baseline_fn=cpu_baseline)

    torch.cuda.set_device(0)
    h = mp.Handle(0)
    data = "synthetic U[0,1) fp32, seed 1234"
    if args.config == "c3":
        import tempfile
        J = mp.synthetic_jacobian()
        with tempfile.TemporaryDirectory() as td:               # the reference's A_%09d.txt format, write + read back
            mp.write_euroc_jacobian(os.path.join(td, "A_000000001.txt"), J)
            J = mp.read_euroc_jacobian(os.path.join(td, "A_000000001.txt"))
        m, n = J.shape
        h.plan(m, n, r, outer_block=args.outer_block, lookahead=not args.no_lookahead)
        h.set_matrix(J)
        data = "synthetic block-sparse Jacobian (40 cameras, 580 points) read from the reference's text format"
    else:
        h.plan(m, n, r, outer_block=args.outer_block, lookahead=not args.no_lookahead,
               precision={"fp16": mp.PREC_FP16, "fp8": mp.PREC_FP8, "fp32": mp.PREC_FP32}[prec_name])
        h.generate(1234)
    h.sync()

    def step():
        h.factor()        # copies the retained input into the working matrix, factors, forms Q

    for _ in range(args.warmup):
        step()
    h.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    h.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps

    fl = mp.flops(m, n, r)
    # the serial core of one leaf (gh_solve: one workgroup on 1 of 256 CUs) timed alone; x the number of such leaves = the part
    # of the panel chain no other kernel can overlap (breakdown_ms.ms_gh_solve)
    us_solve = h.bench_leaf_solve(min(r, 128), 50) if prec_name != "fp32" else 0.0
    tm = h.timings()                      # HIP-event timings of the LAST step, on the library's own stream
    if args.dump_records:
        with open(args.dump_records, "w") as fh:
            json.dump({"config": args.config, "m": m, "n": n, "r": r, "records": h.update_records(),
                       "n_far_launches": tm["n_far_launches"], "n_q_launches": tm["n_q_launches"]}, fh, indent=1)
    mt = h.metrics()
    # dominant kernel: the far trailing-update GEMM  A2 -= V Y^T  (fp16 MFMA, K = outer block); achieved =
    # algorithmic flops (2 M N K summed over its launches) / their HIP-event time on the library's update stream
    nn_t = tm["ms_far_nn"] * 1e-3
    roof = None
    if tm["n_far_launches"] > 0 and nn_t > 0:
        ach_far = tm["flops_far_nn"] / nn_t / 1e12
        ach = ach_far
        q_nn = None
        if prec_name != "fp8" and tm.get("n_q_launches", 0) > 0 and tm["ms_q_nn"] > 0:
            # the same kernel also forms Q (Q2 -= V Y^T, K = 2 outer blocks): ALL its launches are priced
            q_nn = tm["tflop_q"] / (tm["ms_q_nn"] * 1e-3)
            ach = (tm["flops_far_nn"] * 1e-12 + tm["tflop_q"]) / (nn_t + tm["ms_q_nn"] * 1e-3)
        peak = PEAK_FP8_TFLOPS if prec_name == "fp8" else PEAK_FP16_TFLOPS
        kname = ("gemm8_fp8_kernel<E_SUB_F32> (far A2 -= V*Y^T, e4m3 x e4m3 -> fp32 on v_mfma_scale_f32_32x32x64_f8f6f4, K = outer block)"
                 if prec_name == "fp8" else
                 "gemm6_f16_kernel<E_SUB_F32> (C -= V*Y^T, fp16 x fp16 -> fp32: far trailing updates, K = 2 outer blocks beyond the next two blocks, and Q formation, K = 2 outer blocks)")
        roof = {"bound": "mfma", "kernel": kname,
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "traffic": None, "launches": tm["n_far_launches"] + (tm.get("n_q_launches", 0) if q_nn else 0),
                "avg_launch_ms": (tm["ms_far_nn"] + (tm["ms_q_nn"] if q_nn else 0.0)) / (tm["n_far_launches"] + (tm.get("n_q_launches", 0) if q_nn else 0)),
                "far_update_achieved": ach_far, "q_formation_achieved": q_nn,
                "q_formation_tn_achieved": (tm["tflop_q_tn"] / (tm["ms_q_tn"] * 1e-3)) if q_nn and tm["ms_q_tn"] > 0 else None,      # executed flops (identity / zero parts of Q skipped)
                "tn_achieved": (tm["flops_far_tn"] / (tm["ms_far_tn"] * 1e-3) / 1e12) if tm["ms_far_tn"] > 0 else None}
        if args.config == "c4" and not args.outer_block and prec_name == "fp16":
            roof["traffic"], roof["traffic_source"] = _pmc_traffic()
        if prec_name == "fp16" and not args.no_alone:
            # Context for `frac` (never `value`): (1) what the matrix pipes of THIS box deliver on random operands -- a bare MFMA loop from
            # registers, both fp16 shapes, with the clock the chip holds meanwhile (the 2.5 PFLOP/s peak is width x 2.4 GHz); (2) the
            # dominant kernel ALONE on the GPU at the shapes this factorisation launches it with (mpqr_bench_gemm): the difference to
            # `achieved` is what running beside the panel chain costs it
            p32, g32 = h.bench_mfma_peak(0)
            p16, g16 = h.bench_mfma_peak(1)
            roof["mfma_measured"] = {"unit": "TFLOP/s", "32x32x16_f16": p32, "32x32x16_clock_ghz": g32, "16x16x32_f16": p16, "16x16x32_clock_ghz": g16,
                                     "note": "bare MFMA loop, random fp16 operands in registers, 8 waves per CU; nominal peak 2500 = width x 2.4 GHz"}
            if args.alone and m >= 4096 and n >= 4096:
                Mr, Nr = m // 256 * 256, max(256, (n - 3 * (args.outer_block or 1024)) // 256 * 256)
                kb = args.outer_block or 1024
                al = {}
                for name, (kern, mode, M_, N_, K_) in {"far_nn_K%d" % kb: (6, 2, Mr, Nr, kb), "far_nn_K%d" % (2 * kb): (6, 2, Mr, Nr, 2 * kb),
                                                         "q_nn_K%d_shadow" % (2 * kb): (6, 3, Mr, Mr, 2 * kb), "far_tn_fp32_N%d" % (2 * kb): (2, 1, Nr, 2 * kb, Mr),
                                                         "q_tn_N%d" % (2 * kb): (16, 1, Mr, 2 * kb, Mr)}.items():
                    ms_ = h.bench_gemm(kern, mode, M_, N_, K_, iters=4)
                    al[name] = {"M": M_, "N": N_, "K": K_, "ms": ms_, "tflops": 2.0 * M_ * N_ * K_ / (ms_ * 1e-3) / 1e12}
                roof["kernel_alone"] = al
        # The far launches against the HBM roof: algorithmic bytes = fp32 C read + write (8 M N) + both fp16 operands once
        # (2 K (M + N)), summed by the library over the far updates it launched (pairs of blocks beyond the next two).
        alg = tm["gbytes_far_nn"] * 1e9
        roof["hbm_view"] = {"set": "far updates", "algorithmic_bytes": alg, "achieved": alg / nn_t / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                            "frac": alg / nn_t / 1e9 / PEAK_HBM_GBPS, "flop_per_byte": tm["flops_far_nn"] / alg,
                            "ridge_flop_per_byte": peak * 1e3 / PEAK_HBM_GBPS}
        # algorithmic bytes per launch over the SAME launches `traffic` is averaged over (the library sums, per launch, fp32 C
        # read + write, the fp16 shadow where it is written, both fp16 operands once)
        nl = tm["n_far_launches"] + (tm.get("n_q_launches", 0) if q_nn else 0)
        roof["algorithmic_bytes_per_launch"] = ((tm["gbytes_far_nn"] + (tm["gbytes_q_nn"] if q_nn else 0.0)) * 1e9 / nl) if nl else None
    # The panel step against the HBM roof (algorithmic bytes per panel = 2 W r 4, each panel read and written once):
    # the critical path of the factorisation, latency bound -- the fraction says how far from a bandwidth-bound panel it is
    pbytes = sum(2.0 * (m - lam) * min(r, n - lam) * 4 for lam in range(0, n, r))
    if roof is not None and tm["ms_panel"] > 0:
        roof["panel"] = {"bound": "hbm", "kernels": "gh_gram / gh_reduce / gh_solve3 / gh_apply / t_panel / leaf_xt + in-block updates (the chain stream)",
                         "algorithmic_bytes": pbytes, "ms": tm["ms_panel"], "achieved": pbytes / (tm["ms_panel"] * 1e-3) / 1e9,
                         "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": pbytes / (tm["ms_panel"] * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                         "us_gh_solve_per_leaf": us_solve, "n_gh_leaves": tm["n_gh_leaves"]}
    out = {
        "metric": "GFLOP/s block QR (%s MFMA trailing)" % prec_name, "value": fl["geqrf"] / dt / 1e9, "unit": "GFLOP/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": {"fp16": "f16xf16->f32 (fp32 panel)", "fp32": "f32 (exact-f32 MFMA)",
                  "fp8": "e4m3xe4m3->f32 far trailing update (fp32 panel; fp16 in-block updates and Q formation)"}[prec_name],
        "data": data,
        "config": {"workload": f"{m}x{n} {'sparse Jacobian' if args.config == 'c3' else 'random dense'}, block={r}, full Q formed", "m": m, "n": n, "block": r,
                   "outer_block": args.outer_block or 1024, "parallelism": "1 gpu"},
        "error": {"backward_error": mt["backward_error"], "q_error_fro": mt["q_error_fro"],
                  "q_error_max_signed": mt["q_error_max_signed"]},
        "breakdown_ms": dict({k: tm[k] for k in ("ms_total", "ms_factor", "ms_form_q", "ms_trailing", "ms_panel", "ms_chain_wait",
                                                 "ms_far_tn", "ms_far_nn", "ms_q_tn", "ms_q_nn", "ms_host_enqueue", "n_passes", "n_robust_leaves")},
                             ms_gh_solve=tm["n_gh_leaves"] * us_solve * 1e-3),
        "gflops_with_q_flops": (fl["geqrf"] + fl["form_q"]) / dt / 1e9,
        "gflops_reference_formula": (4.0 * m * m * n - m * n * n + n ** 3 / 3.0) / dt / 1e9,
        "roofline": roof,
    }
    if not args.no_dropin and 4.0 * m * m <= 4e9:          # (config 5's 17 GB host Q is left out)
        # the drop-in call itself (host pointers in, host Q and R out: the contract of qr.cu:1049-1226), PCIe inclusive; never `value`
        import numpy as np
        Ah = np.zeros((m + 1, n), np.float32)
        Ah[:m] = np.random.default_rng(1234).random((m, n), dtype=np.float32)
        Qh = np.zeros((m, m), np.float32)
        A1 = Ah.copy()
        mp.dev_mixed_precision_block_qr(A1, Qh, m, n, r, handle=h, outer_block=args.outer_block, lookahead=not args.no_lookahead)   # warm-up: pins, plans
        A1[:] = Ah
        t0 = time.perf_counter()
        mp.dev_mixed_precision_block_qr(A1, Qh, m, n, r, handle=h, outer_block=args.outer_block, lookahead=not args.no_lookahead)
        out["dropin_ms"] = (time.perf_counter() - t0) * 1e3
        del Ah, Qh, A1
    if not args.no_cpu_baseline:
        out["cpu_baseline"], port = cpu_baseline()
        if port is not None:
            out["cpu_baseline_port"] = port
    print(json.dumps(out))


if __name__ == "__main__":
    main()
