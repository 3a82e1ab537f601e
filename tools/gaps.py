#!/usr/bin/env python3
"""One two-stream step as a list: every kernel of the step with its stream (queue), start and duration, a '|' mark while no GEMM
is resident.  From a rocprofv3 --kernel-trace CSV of `bench.py`.  usage: gaps.py <kernel_trace.csv> [step index from the end]"""
import csv, re, sys
pat = re.compile(r"pconv_kernel|pwgrad|igemm_kernel|wgrad_kernel|stem_rows")
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
ends = [e for s, e, n, q in rows if "adam_kernel" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo, hi = ends[-k - 1], ends[-k]
sel = [r for r in rows if r[0] >= lo and r[1] <= hi + 1]
gem = [(s, e) for s, e, n, q in sel if pat.search(n)]
def gemm_resident(t0, t1):
    # share of [t0, t1) with a GEMM resident
    iv = sorted((max(s, t0), min(e, t1)) for s, e in gem if e > t0 and s < t1)
    tot, cur = 0, t0
    for s, e in iv:
        s = max(s, cur)
        if e > s: tot += e - s; cur = e
    return tot / max(t1 - t0, 1)
qs = sorted({q for _, _, _, q in sel})
print(f"step of {(hi - lo) / 1e6:.3f} ms; queues {qs}")
for s, e, n, q in sel:
    nm = re.sub(r"^void ", "", n).split("(")[0][:46]
    share = gemm_resident(s, e)
    mark = "G" if pat.search(n) else (" " if share > 0.9 else ("~" if share > 0.3 else "|"))
    print(f"{(s - lo) / 1e3:9.1f} us  q{qs.index(q)}  {(e - s) / 1e3:8.1f} us  {mark} {nm}")
