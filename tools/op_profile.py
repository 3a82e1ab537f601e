#!/usr/bin/env python3
"""Per-op time of one EfficientNet-B0 step (fm_profile_ops): which op of which block costs what.
usage: python tools/op_profile.py [--precision bf16] [--batch 512] [--steps 3]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedmlp_amd import spec                      # noqa: E402
from fedmlp_amd.engine import Engine             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--hw", type=int, default=224)
ap.add_argument("--streams", type=int, default=0, help="engine stream mode: 1 = one stream (solo per-op times)")
a = ap.parse_args()
B, C = a.batch, 5
eng = Engine("Efficient_b0", C, a.hw, a.hw, 2 * B, precision=a.precision, streams=a.streams)
flat, cnt = spec.init_state("Efficient_b0", C, 1037)
eng.set_state(flat, cnt); eng.teacher_snapshot(); eng.adam_reset(3e-5)
g = torch.Generator(device="cuda").manual_seed(1)
x1 = torch.randn((B, 3, a.hw, a.hw), device="cuda", generator=g)
x2 = torch.randn((B, 3, a.hw, a.hw), device="cuda", generator=g)
y = (torch.rand((B, C), device="cuda", generator=g) < 0.15).float()
lo = torch.zeros(1, device="cuda")
mask = [1.0, 0, 0, 0, 0]
for _ in range(2):
    eng.step_stage1(x1, x2, y, mask, 1, B, lo)
eng.profile_ops(True, read=False)
for _ in range(a.steps):
    eng.step_stage1(x1, x2, y, mask, 1, B, lo)
rows = eng.profile_ops(False)
tot = sum(r[2] for r in rows) / a.steps
by_op = collections.defaultdict(float)
for lab, n, ms in rows:
    by_op[lab.split("@")[0] + ("/eval" if 200 <= int(lab.split("@")[1]) <= 300 else
                               "/bwd" if int(lab.split("@")[1]) >= 399 else "/fwd")] += ms / a.steps
print(f"total timed ops: {tot:.2f} ms/step")
for k, v in sorted(by_op.items(), key=lambda kv: -kv[1]):
    print(f"{k:28s} {v:8.3f} ms")
print("--- per op and block (ms/step) ---")
for lab, n, ms in sorted(rows, key=lambda r: -r[2]):
    print(f"{lab:28s} calls {n // a.steps:3d}  {ms / a.steps:8.3f}")

# ---- achieved bandwidth of the big ops (algorithmic bytes of the op's tensors / time) -------------
STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
          (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
r16 = lambda c: (c + 15) // 16 * 16
blocks, h = [], a.hw // 2
for rep, k, s, e, cin, cout in STAGES:
    for r in range(rep):
        ss, ci = (s if r == 0 else 1), (cin if r == 0 else cout)
        ho = (h + ss - 1) // ss
        blocks.append(dict(k=k, s=ss, e=e, cin=r16(ci), ce=r16(ci * e), cout=r16(cout), hin=h, hout=ho))
        h = ho
esz = 2 if a.precision == "bf16" else 4
T = 2 * B                                      # images per train-mode / teacher pass of a stage-1 step


def nbytes(op, blk):
    # labels (engine.hip): train forward @0..@15, eval forward @201..@216, backward @400..@415 (heads @100 / @300 / @500, stem @-1 / @399)
    i = blk if 0 <= blk < 16 else blk - 201 if 201 <= blk <= 216 else blk - 400 if 400 <= blk <= 415 else -1
    if not (0 <= i < 16):
        return None
    b = blocks[i]
    E, Ep, S, Sp = b["hin"] ** 2 * b["ce"], b["hout"] ** 2 * b["ce"], b["hin"] ** 2 * b["cin"], b["hout"] ** 2 * b["cout"]
    n = {"exp_fwd": S + E, "proj_fwd": Ep + Sp, "exp_dgrad": E + S, "proj_dgrad": Sp + Ep, "exp_wgrad": S + E,
         "proj_wgrad": Ep + Sp, "k_dw_fwd": E + Ep, "k_dw_dgrad": E + Ep, "k_dw_wgrad": E + Ep,
         "k_se_fwd": Ep, "k_se_bwd": 2 * Ep, "k_se_scale": 2 * Ep, "bn_fwd_tensor": Ep}.get(op)
    if op == "k_dw_fwd" and b["e"] == 1:
        n = S + Ep
    return None if n is None else n * T * esz


print("--- achieved GB/s of the big ops (algorithmic bytes of their tensors) ---")
for lab, n, ms in sorted(rows, key=lambda r: -r[2]):
    op, blk = lab.split("@")
    nb = nbytes(op, int(blk))
    if nb and ms > 0:
        print(f"{lab:22s} {ms / a.steps:7.3f} ms  {nb / 1e9:6.2f} GB  {nb / (ms / a.steps) / 1e6:7.0f} GB/s")
