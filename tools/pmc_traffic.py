#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch for one kernel.

Usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> [out.json [big]]
Units and gfx950 correction as prescribed by MI355X_MICROARCH.md (HBM section): both counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read on gfx950, so it is doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores (our epilogue stores 4 B per lane: treat as a lower bound).
"""
import csv
import json
import sys


def per_launch(path, counter, needle, big_only=False):
    """big_only: only the launches bench.py prices in `roofline` -- the far updates (they run on the stream that never
    launches gh_solve) and Q formation (everything after the last panel solve); the in-block launches of the same kernel
    on the chain stream are left out."""
    rows = list(csv.DictReader(open(path)))
    solve = [r for r in rows if "gh_solve" in r["Kernel_Name"]]
    chain_q = solve[0]["Queue_Id"] if solve else None
    # the chain queue's launches count from Q formation on: everything behind the factorisation's LAST gh_apply (round 4: Q = I is set
    # up at the start of the factorisation, so the identity kernels no longer mark the Q phase; the solves bench.py times alone
    # afterwards, mpqr_bench_leaf_solve, launch no gh_apply and so do not move the mark)
    applies = [int(r["Dispatch_Id"]) for r in rows if "gh_apply" in r["Kernel_Name"]]
    last_solve = max(applies) if applies else (max(int(r["Dispatch_Id"]) for r in solve) if solve else -1)
    tcol = [r for r in rows if "t_colblock_h16" in r["Kernel_Name"]]      # the T stream (round 3: it also runs the deferred in-block updates)
    t_q = tcol[0]["Queue_Id"] if tcol else None
    vals = []
    for row in rows:
        if needle in row["Kernel_Name"] and row["Counter_Name"] == counter:
            if big_only and chain_q is not None and row["Queue_Id"] == chain_q and int(row["Dispatch_Id"]) < last_solve:
                continue
            if big_only and t_q is not None and t_q != chain_q and row["Queue_Id"] == t_q:
                continue
            vals.append(float(row["Counter_Value"]))
    return vals


def main():
    fetch, write, needle = sys.argv[1:4]
    big = len(sys.argv) > 5 and sys.argv[5] == "big"
    f = per_launch(fetch, "FETCH_SIZE", needle, big)
    w = per_launch(write, "WRITE_SIZE", needle, big)
    n = min(len(f), len(w))
    out = {
        "kernel": needle, "launches": n, "set": "far updates + Q formation" if big else "all launches",
        "fetch_kib_raw_avg": sum(f) / max(len(f), 1), "write_kib_avg": sum(w) / max(len(w), 1),
        "hbm_bytes_per_launch": (2.0 * sum(f) / max(len(f), 1) + sum(w) / max(len(w), 1)) * 1024.0,
        "note": "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads); one factorisation of 16384x16384, r=128",
    }
    try:                                                       # ties the profile to the build it was taken on (bench.py: roofline.traffic_source)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        out["kernel_source_sha"] = bench.gemm_source_sha()
    except Exception:
        pass
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
