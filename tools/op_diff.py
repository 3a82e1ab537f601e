#!/usr/bin/env python3
"""side-by-side per-(op, block) times of two or more tools/op_profile.py outputs: op_diff.py <pattern> a.txt b.txt ..."""
import re, sys
pat = sys.argv[1]; files = sys.argv[2:]
tabs = []
for f in files:
    d = {}
    for line in open(f):
        m = re.match(r'(\S+@-?\d+)\s+calls\s+(\d+)\s+([\d.]+)', line)
        if m and re.search(pat, m.group(1)): d[m.group(1)] = float(m.group(3))
    tabs.append(d)
keys = sorted(set().union(*tabs), key=lambda k: (k.split('@')[0], int(k.split('@')[1])))
print(f"{'op@block':24s}" + "".join(f"{f.split('/')[-1][:14]:>15s}" for f in files))
for k in keys:
    print(f"{k:24s}" + "".join(f"{t.get(k, float('nan')):15.3f}" for t in tabs))
print(f"{'sum':24s}" + "".join(f"{sum(t.values()):15.3f}" for t in tabs))
