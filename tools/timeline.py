#!/usr/bin/env python3
"""Where the matrix pipe idles in a two-stream step: from a rocprofv3 --kernel-trace CSV of `bench.py` (default stream mode),
the union of the GEMM kernels' [start, end) intervals per step against the step's span, and which kernels cover the gaps.
usage: timeline.py <kernel_trace.csv> [gemm name regex]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"igemm_kernel|wgrad_kernel")
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steps are delimited by the optimizer kernel
ends = [e for s, e, n in rows if "adam_kernel" in n]
if len(ends) < 4:
    sys.exit("not enough steps in the trace")
lo, hi = ends[len(ends) // 2 - 1], ends[-1]               # the second half of the run (timed steps)
nsteps = len(ends) - len(ends) // 2
sel = [(s, e, n) for s, e, n in rows if s >= lo and e <= hi]
gemm = sorted((s, e) for s, e, n in sel if pat.search(n))
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e in gemm:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((cur_e, s))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = hi - lo
print(f"{nsteps} steps, {span / nsteps / 1e6:.3f} ms per step; a GEMM kernel is running {busy / nsteps / 1e6:.3f} ms of it "
      f"({100 * busy / span:.1f} %), no GEMM {(span - busy) / nsteps / 1e6:.3f} ms")
cover = defaultdict(float)
for g0, g1 in gaps:
    for s, e, n in sel:
        if e <= g0 or s >= g1 or pat.search(n):
            continue
        cover[n.split("(")[0][:60]] += min(e, g1) - max(s, g0)
tot = sum(g1 - g0 for g0, g1 in gaps)
print(f"gaps: {len(gaps) / nsteps:.0f} per step, {tot / nsteps / 1e6:.3f} ms; kernels running inside them (ms per step, may overlap):")
for n, t in sorted(cover.items(), key=lambda kv: -kv[1])[:14]:
    print(f"   {t / nsteps / 1e6:7.3f}  {n}")
# how much GEMM time is spent with TWO GEMM kernels resident at once
ev = []
for s, e in gemm:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, two = 0, None, 0
for t, d in ev:
    if depth >= 2:
        two += t - last
    depth += d
    last = t
print(f"two or more GEMM kernels resident: {two / nsteps / 1e6:.3f} ms per step")
