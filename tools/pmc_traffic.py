#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs of the same
bench command with --output-format csv) into per-kernel HBM bytes per launch.

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE reports half the bytes of 16-B/lane streams, so it is doubled.

usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> "<command>"
"""
import csv
import json
import sys
from collections import defaultdict


def fold(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            n[r["Kernel_Name"]] += 1
    return tot, n


def main():
    fetch, nf = fold(sys.argv[1], "FETCH_SIZE")
    write, nw = fold(sys.argv[2], "WRITE_SIZE")
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `{sys.argv[4]}`",
           "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streams, "
                         "MI355X_MICROARCH.md HBM section); units KiB -> bytes x1024",
           "kernels": {}}
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
        if k.startswith("void at::") or "rocclr" in k:
            continue
        n = nf[k]
        f_kb, w_kb = fetch[k] / n, write.get(k, 0.0) / max(nw.get(k, 1), 1)
        out["kernels"][k] = {"launches": n, "fetch_size_kb_raw_per_launch": round(f_kb, 1),
                             "write_size_kb_per_launch": round(w_kb, 1),
                             "hbm_bytes_per_launch_corrected": int((2 * f_kb + w_kb) * 1024)}
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
