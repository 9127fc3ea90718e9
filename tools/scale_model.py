#!/usr/bin/env python3
"""What the column-sharded schedule can deliver at N GPUs, derived from ONE single-GPU bench line (no multi-GPU hardware was in reach).

usage: python tools/scale_model.py <bench_line.json> [--chain-alone-ms X] [--link-gbps 153]

Model (multi-GPU section).  A step on one GPU is  chain + Q formation + fixed, with the far updates hidden behind the chain:
  * chain      = ms_panel: the panel chain of every top-level block, on the block's OWNER only.  Block t + 1 needs block t's reflectors, so
                 the owners' chains are serial: the chain does NOT shard.  What shrinks with N is what the far updates cost it: on one GPU
                 the chain runs beside ALL far GEMMs (ms_far_tn + ms_far_nn) and is slowed by them; at N ranks each rank runs 1 / N of them.
                 chain_N = chain_alone + (ms_panel - chain_alone) / N, chain_alone = the chain with the far updates switched off
                 (--chain-alone-ms: measure it with MPQR_DBG_NOFAR=1; default: ms_panel - 0.55 x far GEMM time, the ratio of round 4 / 5).
  * broadcast  = per block [V^T | T | T^T] fp16 + T fp32 over xGMI (point-to-point links, --link-gbps each, the owner feeds N - 1 links in
                 parallel): exposed only where it exceeds the next block's chain (look-ahead), i.e. never at these sizes; reported.
  * far update = (ms_far_tn + ms_far_nn) / N per rank, hidden behind the chain while it is shorter than chain_N.
  * Q formation = ms_form_q / N (column-sharded, no communication).
  * fixed      = ms_total - ms_panel - ms_form_q of the bench line (copy-in, scale pass, joins).
T_N = max(chain_N, far / N) + Q / N + fixed + exposed broadcast.   Prints T_N and T_1 / T_N for N = 1, 2, 4, 8."""
import argparse
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bench")
    ap.add_argument("--chain-alone-ms", type=float, default=None)
    ap.add_argument("--link-gbps", type=float, default=153.0)
    a = ap.parse_args()
    line = [l for l in open(a.bench) if l.startswith("{")][-1]
    d = json.loads(line)
    b, c = d["breakdown_ms"], d["config"]
    m, n, ko = c["m"], c["n"], c.get("outer_block", 1024)
    far = b["ms_far_tn"] + b["ms_far_nn"]
    chain = b["ms_panel"]
    alone = a.chain_alone_ms if a.chain_alone_ms is not None else max(chain - 0.55 * far, 0.5 * chain)
    q = b["ms_form_q"]
    fixed = max(b["ms_total"] - chain - q, 0.0)
    nblk = (n + ko - 1) // ko
    bc_bytes = [(ko * (m - t * ko) + 2 * ko * ko) * 2 + ko * ko * 4 for t in range(nblk)]
    print(f"{m} x {n}, outer block {ko}: one GPU {b['ms_total']:.2f} ms = chain {chain:.2f} (alone {alone:.2f}) + Q {q:.2f} + fixed {fixed:.2f}; far GEMMs {far:.2f} ms hidden")
    print(f"broadcast per block: {max(bc_bytes) / 2**20:.1f} MiB at most = {max(bc_bytes) / (a.link_gbps * 1e9) * 1e3:.2f} ms per link at {a.link_gbps:.0f} GB/s; chain per block {chain / nblk:.2f} ms")
    t1 = None
    for N in (1, 2, 4, 8):
        chain_n = alone + (chain - alone) / N
        bc_exposed = 0.0 if N == 1 else sum(max(0.0, x / (a.link_gbps * 1e9) * 1e3 - chain_n / nblk) for x in bc_bytes)
        tn = max(chain_n, far / N) + q / N + fixed + bc_exposed
        t1 = t1 or tn
        print(f"  N = {N}: chain {chain_n:6.2f}  far/N {far / N:5.2f}  Q/N {q / N:5.2f}  broadcast exposed {bc_exposed:4.2f}  ->  T_N = {tn:6.2f} ms   speed-up {t1 / tn:4.2f} x")


if __name__ == "__main__":
    main()
