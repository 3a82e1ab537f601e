#!/usr/bin/env python3
"""npz_diff.py a.npz b.npz: integer arrays must be equal, float arrays within 1e-6 of their max"""
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    x, y = a[k], b[k]
    if x.dtype.kind in "iu":
        if not np.array_equal(x, y):
            print(k, "differs in", int((x != y).sum()), "of", x.size); bad += 1
    else:
        d = np.abs(x - y).max() / max(1e-30, np.abs(x).max())
        if d > 1e-6:
            print(k, "rel", d); bad += 1
print(sys.argv[2], "arrays", len(a.files), "bad", bad)
