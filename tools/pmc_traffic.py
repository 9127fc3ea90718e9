#!/usr/bin/env python3
"""HBM bytes of the read-modify-write GEMM launches (C -= V Y^T) from two rocprofv3 PMC passes, matched LAUNCH BY LAUNCH against the
library's own record of those launches, so that measured and algorithmic bytes cover the same launches.

Usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <records.json> [out.json]

  * the two CSVs: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, no trace domains) of
    `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --no-alone --dump-records <records.json>`;
  * records.json: that command's --dump-records output (mpqr_get_update_records: the far updates and Q formation's applies of the one
    factorisation in launch order with M, N, K, algorithmic flops and bytes).

Units and the gfx950 correction as MI355X_MICROARCH.md prescribes (HBM / rocprofv3 section): both counters are in KiB; FETCH_SIZE reports half
of the bytes of a wide coalesced streaming read on gfx950, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores (the
epilogue's 4-byte stores: a lower bound).

Matching (round 5; round 4's tool averaged the measured bytes over whatever rows the PMC pass held -- 24 far launches of a truncated pass --
and the algorithmic bytes over all 37 launches of the bench line: VERDICT round 4):
  * rows = launches of the E_SUB_F32 kernels (`gemm6_f16_kernel<2, ...>` and the 128-tile `gemm_f16_kernel<0, 2>`), in dispatch order;
  * far set = those on the far-update queue (the queue that runs neither gh_solve nor the T stream's kernels): record i <-> row i;
  * Q set = those on the chain queue behind the last leaf: Q record i <-> row i;
  * a set whose row count differs from its record count is reported `complete: false` and only the common prefix is matched;
  * every output figure names its launches (count, first / last dispatch id) and carries the algorithmic bytes of exactly those launches.
"""
import csv
import json
import os
import sys

ESUB = ("gemm6_f16_kernel<2,", "gemm_f16_kernel<0, 2>")


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    for r in rows:
        r["_id"] = int(r["Dispatch_Id"])
    rows.sort(key=lambda r: r["_id"])
    return rows


def queues(rows):
    chain = next((r["Queue_Id"] for r in rows if "gh_solve" in r["Kernel_Name"]), None)
    tq = next((r["Queue_Id"] for r in rows if "t_colblock_h16" in r["Kernel_Name"]), None)      # (wait_flag_kernel also runs on the chain queue)
    return chain, tq


def select(rows):
    """(far rows, q rows) of the E_SUB kernels."""
    chain, tq = queues(rows)
    # the factorisation's last leaf: its leaf_a / gh_apply launch (bench.py times gh_solve alone AFTER the factorisation: mpqr_bench_leaf_solve)
    leaf_ids = [r["_id"] for r in rows if "leaf_a_kernel" in r["Kernel_Name"] or "gh_apply" in r["Kernel_Name"] or "leaf_tail" in r["Kernel_Name"]]
    last_leaf = max(leaf_ids) if leaf_ids else -1
    far, q = [], []
    for r in rows:
        if not any(k in r["Kernel_Name"] for k in ESUB):
            continue
        if r["Queue_Id"] == chain:
            if r["_id"] > last_leaf:
                q.append(r)
        elif r["Queue_Id"] != tq:
            far.append(r)
    return far, q


def match(fetch_rows, write_rows, recs, name):
    # The records describe the LAST factorisation of the process: its launches are the last len(recs) rows of the set.  (A --pmc pass
    # serialises kernels, so a polling wait of the T stream can never see the chain's word: the factorisation times out once and is
    # repeated with event hand-offs -- mpqr_factor's retry -- unless the pass is run with MPQR_TPOLL=0, as tools/collect_profiles.sh does.
    # Either way the rows of an earlier, discarded pass come first.)
    extra_f, extra_w = len(fetch_rows) - len(recs), len(write_rows) - len(recs)
    if extra_f > 0: fetch_rows = fetch_rows[extra_f:]
    if extra_w > 0: write_rows = write_rows[extra_w:]
    n = min(len(fetch_rows), len(write_rows), len(recs))
    ids = [r["_id"] for r in fetch_rows[:n]]
    fb = [2.0 * float(r["Counter_Value"]) * 1024.0 for r in fetch_rows[:n]]
    wb = [float(r["Counter_Value"]) * 1024.0 for r in write_rows[:n]]
    alg = [r["bytes"] for r in recs[:n]]

    def grid_ok(row, rec):       # the launch's grid against the record's shape: 256 x 256 tiles of 512 threads, or 128 x 128 tiles of 256
        g, M, N = int(row["Grid_Size"]), rec["M"], rec["N"]
        if "gemm6" in row["Kernel_Name"]:                    # whole 4 x 8 groups of 256 x 256 tiles, 512 threads each
            return g == -(-(-(-M // 256)) // 4) * -(-(-(-N // 256)) // 8) * 32 * 512
        return g == -(-M // 128) * -(-N // 128) * 256
    bad = [fetch_rows[i]["_id"] for i in range(n) if not grid_ok(fetch_rows[i], recs[i])]
    out = {"set": name, "launches": n, "records": len(recs), "pmc_rows_fetch": len(fetch_rows), "pmc_rows_write": len(write_rows),
           "grid_mismatches": bad, "rows_of_earlier_passes_skipped": max(extra_f, 0),
           "complete": n == len(recs) and len(fetch_rows) == len(recs) and len(write_rows) == len(recs) and not bad,
           "dispatch_ids_fetch_pass": [ids[0], ids[-1]] if ids else [],
           "measured_bytes": sum(fb) + sum(wb), "fetch_bytes": sum(fb), "write_bytes": sum(wb), "algorithmic_bytes": sum(alg),
           "ratio": (sum(fb) + sum(wb)) / sum(alg) if n and sum(alg) > 0 else None,
           "per_launch": [{"dispatch_id": ids[i], "M": recs[i]["M"], "N": recs[i]["N"], "K": recs[i]["K"], "measured": fb[i] + wb[i],
                           "algorithmic": alg[i]} for i in range(n)]}
    return out


def main():
    fetch, write, recfile = sys.argv[1:4]
    recs = json.load(open(recfile))["records"]
    far_recs = [r for r in recs if r["set"] == "far"]
    q_recs = [r for r in recs if r["set"] == "q"]
    ff, fq = select(load(fetch, "FETCH_SIZE"))
    wf, wq = select(load(write, "WRITE_SIZE"))
    far = match(ff, wf, far_recs, "far updates")
    q = match(fq, wq, q_recs, "Q formation")
    n = far["launches"] + q["launches"]
    meas = far["measured_bytes"] + q["measured_bytes"]
    alg = far["algorithmic_bytes"] + q["algorithmic_bytes"]
    out = {
        "kernel": "E_SUB_F32 GEMMs (gemm6_f16_kernel<2, ...>, gemm_f16_kernel<0, 2>): C -= V Y^T",
        "launches": n, "complete": far["complete"] and q["complete"],
        "hbm_bytes_per_launch": meas / n if n else None, "algorithmic_bytes_per_launch": alg / n if n else None,
        "ratio": meas / alg if alg > 0 else None,
        "far": far, "q": q,
        "note": "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads); KiB -> bytes; measured and algorithmic bytes cover the same "
                "launches (matched by dispatch order per queue against mpqr_get_update_records)",
    }
    try:                                                       # ties the profile to the build it was taken on (bench.py: roofline.traffic_source)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        out["kernel_source_sha"] = bench.gemm_source_sha()
    except Exception:
        pass
    slim = dict(out)
    print(json.dumps({k: (v if k not in ("far", "q") else {kk: vv for kk, vv in v.items() if kk != "per_launch"}) for k, v in slim.items()}, indent=1))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
