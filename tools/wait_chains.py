#!/usr/bin/env python3
"""Per kernel of a .s file: how many vector-memory loads are followed by a wait for ALL outstanding loads before the next load is
issued (a chain of dependent round trips: the signature of one branch per load).  tools/wait_chains.py <file.s>"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
name, out = None, {}
ld = re.compile(r"\s+(global_load|buffer_load|flat_load)\w*\s")
for l in lines:
    if l.startswith("_Z") and l.rstrip().endswith(":") or (l.startswith("_Z") and ":" in l and "@" in l):
        name = l.split(":")[0]; out[name] = [0, 0, False]; continue
    if name is None: continue
    if ld.match(l) and " lds" not in l:
        out[name][0] += 1; out[name][2] = True
    elif "s_waitcnt" in l and "vmcnt(0)" in l and out[name][2]:
        out[name][1] += 1; out[name][2] = False
for k, (n, w, _) in sorted(out.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    if n: print(f"{w:5d} full waits right behind {n:5d} loads   {k[:100]}")
