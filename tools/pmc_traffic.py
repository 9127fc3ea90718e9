#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch for one kernel.

Usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> [out.json]
Units and gfx950 correction as prescribed by MI355X_MICROARCH.md (HBM section): both counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read on gfx950, so it is doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores (our epilogue stores 4 B per lane: treat as a lower bound).
"""
import csv
import json
import sys


def per_launch(path, counter, needle):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return vals


def main():
    fetch, write, needle = sys.argv[1:4]
    f = per_launch(fetch, "FETCH_SIZE", needle)
    w = per_launch(write, "WRITE_SIZE", needle)
    n = min(len(f), len(w))
    out = {
        "kernel": needle, "launches": n,
        "fetch_kib_raw_avg": sum(f) / max(len(f), 1), "write_kib_avg": sum(w) / max(len(w), 1),
        "hbm_bytes_per_launch": (2.0 * sum(f) / max(len(f), 1) + sum(w) / max(len(w), 1)) * 1024.0,
        "note": "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads); one factorisation of 16384x16384, r=128",
    }
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
