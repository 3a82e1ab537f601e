#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs of the same
bench command with --output-format csv) into per-kernel HBM bytes per launch, plus the whole-step total.

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE reports half the bytes of 16-B/lane streams, so it is doubled.

usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
                                   <workload key> <steps run (warmup + timed)> "<command>"
The output file is a dict keyed by workload (bench.py reads its own workload's entry).
"""
import csv
import json
import os
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
    out_path, workload, steps, cmd = sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
    ent = {"workload": workload,
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `{cmd}`",
           "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streams, "
                         "MI355X_MICROARCH.md HBM section); units KiB -> bytes x1024",
           "kernels": {}}
    total = 0.0
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
        if k.startswith("void at::") or "rocclr" in k or "elementwise_kernel" in k or "distribution" in k:
            continue                       # torch's own kernels (synthetic data generation), not the engine's
        n = nf[k]
        f_kb, w_kb = fetch[k] / n, write.get(k, 0.0) / max(nw.get(k, 1), 1)
        total += (2 * fetch[k] + write.get(k, 0.0)) * 1024
        ent["kernels"][k] = {"launches": n, "fetch_size_kb_raw_per_launch": round(f_kb, 1),
                             "write_size_kb_per_launch": round(w_kb, 1),
                             "hbm_bytes_per_launch_corrected": int((2 * f_kb + w_kb) * 1024)}
    ent["kernels"]["whole_step"] = {"launches": steps, "hbm_bytes_per_launch_corrected": int(total / max(steps, 1)),
                                    "note": "sum over every engine kernel of the run / steps run (warm-up included)"}
    doc = {}
    if os.path.exists(out_path):
        with open(out_path) as f:
            doc = json.load(f)
    doc[workload] = ent
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
