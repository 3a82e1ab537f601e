#!/usr/bin/env python3
"""Per-step losses of a FixMatch round with a tail batch (tests/golden/traj_fixmatch_tails.json): engine vs the oracle in
fp32 and in fp64 on this host.  Tells a conditioning effect (the fp32 oracle is as far from fp64 as the engine) from an
engine defect (only the engine is off).  usage: python tools/diag_tail.py [4|1]"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import steps_ref as R                                    # noqa: E402
from tests.helpers import load_golden, make_args, oracle_net, data_dict       # noqa: E402
from tests.synth import class_lists, perturbed_bn                    # noqa: E402
from fedmlp_amd import spec                                          # noqa: E402
from fedmlp_amd.engine import Engine                                 # noqa: E402

tail = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = load_golden("traj_fixmatch_tails.json")[f"tail{tail}"]
C, N, hw = g["C"], g["N"], g["hw"]
args = make_args(n_classes=C, n_clients=1)
data = data_dict(N, C, hw, g["data_seed"], True)
_, neg = class_lists(data["targets"], C)


def build(dtype):
    net = oracle_net(C, g["init_seed"])
    sd = net.state_dict()
    with torch.no_grad():
        for k, v in perturbed_bn([(k, tuple(t.shape)) for k, t in sd.items()], g["bn_seed"]):
            sd[k].copy_(torch.from_numpy(v))
        net.fc.weight.mul_(g["fc_scale"])
    return net.to(dtype)


def oracle_losses(dtype):
    net = build(dtype)
    d = {k: (v.to(dtype) if torch.is_tensor(v) else v) for k, v in data.items()}
    cl = R.RefClient(args, 0, d, list(range(N)), neg, [0])
    cl.y_masked = cl.y_masked.to(dtype)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=args.base_lr, betas=(0.9, 0.999), weight_decay=5e-4)
    out, masks = [], []
    for pos in R._batches(g["order"], 32):
        _, zw = net(cl._img("image_aug_1", pos)); _, zs = net(cl._img("image_aug_2", pos))
        masks.append(R.fixmatch_mask(zw, cl.negative, 32))
        loss = R.loss_fixmatch(zw, zs, cl.y_masked[pos], [float(v) for v in cl.loss_w], [float(v) for v in cl.loss_w_unknown],
                               cl.active, cl.negative, 32, 1, C) if dtype == torch.float32 else \
            loss_fixmatch64(zw, zs, cl.y_masked[pos], cl.loss_w, cl.loss_w_unknown, cl.active, cl.negative)
        opt.zero_grad(); loss.backward(); opt.step()
        out.append(loss.item())
    return out, masks, (zw.detach(), zs.detach())


def loss_fixmatch64(zw, zs, y, pw, pwu, act, neg_):
    F = torch.nn.functional
    pw = torch.as_tensor(pw, dtype=torch.float64); pwu = torch.as_tensor(pwu, dtype=torch.float64)
    sup = F.binary_cross_entropy_with_logits(zw, y, pos_weight=pw, reduction="none")
    loss = sup[:, act].sum() / 32
    idx = R.fixmatch_mask(zw, neg_, 32)
    if not idx:
        return loss
    hard = (torch.sigmoid(zw) > 0.5).double().detach()
    uns = F.binary_cross_entropy_with_logits(zs, hard, pos_weight=pwu, reduction="none")
    return loss + uns[idx, :][:, neg_].sum() / (len(idx) * (C - 1))


l32, m32, (zw32, zs32) = oracle_losses(torch.float32)
l64, m64, (zw64, zs64) = oracle_losses(torch.float64)
net = build(torch.float32)
flat, cnt = spec.state_dict_to_flat("Resnet18", C, net.state_dict())
eng = Engine("Resnet18", C, hw, hw, 2 * 32)
eng.set_state(flat, cnt)
eng.adam_reset(args.base_lr)
cl = R.RefClient(args, 0, data, list(range(N)), neg, [0])
le = []
lo = torch.zeros(1, device="cuda")
for pos in R._batches(g["order"], 32):
    xw, xs = cl._img("image_aug_1", pos).cuda(), cl._img("image_aug_2", pos).cuda()
    eng.step_fixmatch(xw, xs, cl.y_masked[pos].cuda(), cl.loss_w, cl.loss_w_unknown, [1.0] + [0.0] * (C - 1), 1, 32, lo)
    le.append(lo.item())
print("golden mean loss", g["loss"])
print("step  engine        oracle32      oracle64      |e-64|/64   |o32-64|/64   masks32 masks64")
for i in range(len(le)):
    print(i, f"{le[i]:.7f} {l32[i]:.7f} {l64[i]:.7f}  {abs(le[i]-l64[i])/abs(l64[i]):.2e}  {abs(l32[i]-l64[i])/abs(l64[i]):.2e}",
          len(m32[i]), len(m64[i]))
print("mean", np.mean(le), np.mean(l32), np.mean(l64))
print("last batch weak logits fp32 oracle", zw32.numpy().ravel(), "fp64", zw64.numpy().ravel())
print("last batch strong logits fp32 oracle", zs32.numpy().ravel(), "fp64", zs64.numpy().ravel())
