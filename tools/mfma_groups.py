#!/usr/bin/env python3
"""Condensed listing of a kernel's ISA (runs of MFMAs collapsed): tools/mfma_groups.py <file.s> <kernel substring> [from] [to]"""
import sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
out, m = [], 0
for l in lines[start:end]:
    t = l.split()
    if not t or t[0].startswith(";"): continue
    if t[0].startswith("v_mfma"): m += 1; continue
    if m: out.append(f"   [mfma x{m}]"); m = 0
    out.append(l[:110])
a = int(sys.argv[3]) if len(sys.argv) > 3 else 0
b = int(sys.argv[4]) if len(sys.argv) > 4 else len(out)
for i, l in enumerate(out[a:b], a): print(i, l)
