#!/usr/bin/env python3
"""Do the HBM-bound kernels of one stream run BESIDE the GEMMs of the other (co-resident on the CUs) or only in their tails?
From a rocprofv3 --kernel-trace CSV of the two-stream step: for every instance of a streaming kernel (BatchNorm apply / backward
passes, pooling), the share of its duration during which a GEMM of family F was running, binned by F, with the instance's
duration relative to the shortest instance of the same (kernel, grid).  usage: overlap.py <kernel_trace.csv>"""
import csv, re, sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", ""))))
rows.sort()
ends = [e for s, e, n, g in rows if "adam_kernel" in n]
lo, hi = ends[len(ends) // 2 - 1], ends[-1]
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
def fam(n):
    m = re.match(r"void (pconv_kernel<\d), \d, \d, \d, (true|false)", n)
    if m: return m.group(1) + ("TS>" if m.group(2) == "true" else "tap>")
    for k in ("pwgrad_ring", "pwgrad_kernel", "stem_rows", "wgrad_kernel", "igemm_kernel"):
        if k in n: return k
    return None
gem = [(s, e, fam(n)) for s, e, n, g in sel if fam(n)]
stream = [r for r in sel if any(k in r[2] for k in ("bn_apply_planes", "bn_bwd_apply_planes", "bn_bwd_reduce", "stem_pool"))]
best = defaultdict(lambda: 1e30)
for s, e, n, g in stream:
    key = (n.split("(")[0], g); best[key] = min(best[key], e - s)
agg = defaultdict(lambda: [0, 0.0, 0.0])          # family -> [instances, sum of stretch, sum of duration]
for s, e, n, g in stream:
    d = e - s
    ov = defaultdict(float)
    for gs, ge, gf in gem:
        if ge <= s or gs >= e: continue
        ov[gf] += min(e, ge) - max(s, gs)
    f = max(ov, key=ov.get) if ov and max(ov.values()) > 0.5 * d else "alone"
    a = agg[f]; a[0] += 1; a[1] += d / best[(n.split("(")[0], g)]; a[2] += d
n_steps = len(ends) - len(ends) // 2
print("streaming kernels by the GEMM family running beside them for more than half of their duration:")
for f, (c, st, du) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"   {f:14s} {c / n_steps:6.1f} per step, {du / n_steps / 1e6:6.3f} ms per step, duration / shortest same launch = {st / c:5.2f}")
alone = defaultdict(float)
for s, e, n, g in stream:
    d = e - s
    ov = 0.0
    for gs, ge, gf in gem:
        if ge <= s or gs >= e: continue
        ov += min(e, ge) - max(s, gs)
    if ov <= 0.5 * d: alone[n.split("(")[0][:48]] += d
print("alone, by kernel (ms per step):")
for n, t in sorted(alone.items(), key=lambda kv: -kv[1]):
    print(f"   {t / n_steps / 1e6:7.3f}  {n}")
# per-stream GEMM time and idle time of each stream inside the step (Stream_Id when the trace has it)
