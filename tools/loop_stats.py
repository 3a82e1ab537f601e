#!/usr/bin/env python3
"""Instruction mix of the hottest loops of a kernel in a hipcc -save-temps .s file: tools/loop_stats.py <file.s> <kernel substring>.
A loop = a backward branch; prints, for the loops with the most MFMAs, the counts per instruction class (spill traffic shows
as scratch_ / v_readlane / v_writelane inside the loop)."""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {l[:-1]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
loops = []
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def cls(op):
    for k in ("v_mfma", "ds_read", "ds_write", "buffer_load", "buffer_store", "global_load", "global_store", "scratch_", "v_readlane", "v_writelane",
              "v_readfirstlane", "s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "v_cndmask", "s_nop"):
        if op.startswith(k): return k
    return "valu_other" if op.startswith("v_") else ("salu" if op.startswith("s_") else "other")
out = []
for a, b in loops:
    c = {}
    for l in body[a:b + 1]:
        t = l.split()
        if not t or t[0].startswith(".") or t[0].startswith(";") or t[0].endswith(":"): continue
        k = cls(t[0]); c[k] = c.get(k, 0) + 1
    out.append((c.get("v_mfma", 0), a, b, c))
out.sort(key=lambda x: -x[0])
for n, a, b, c in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 3]:
    print(f"loop lines {a}..{b} ({b - a} lines):", " ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
