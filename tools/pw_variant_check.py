#!/usr/bin/env python3
"""Dump outputs / statistics of every pointwise conv on fixed inputs to an .npz; run with different FM_PW_* settings and
compare: the variants must agree bit for bit in the outputs and to rounding in the statistics."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from fedmlp_amd.engine import Engine

e = Engine("Efficient_b0", 5, 64, 64, 16, precision="bf16")
dev = e.device
res = {}
for imgs, groups in ((8, 1), (6, 2)):
    for ci in range(e.debug_num_convs()):
        info = e.debug_conv_info(ci)
        if info["k"] != 1:
            continue
        M, K, h, w = info["cout_p"], info["cin_p"], info["hout"], info["wout"]
        npix = imgs * h * w
        g = torch.Generator().manual_seed(ci)
        x = torch.randn((npix, K), generator=g).to(torch.bfloat16).to(dev)
        dy = torch.randn((npix, M), generator=g).to(torch.bfloat16).to(dev)
        r = torch.randn((npix, K), generator=g).to(torch.bfloat16).to(dev)
        out = torch.empty((npix, M), dtype=torch.bfloat16, device=dev)
        stats = torch.zeros((groups, 2, M), device=dev)
        e.debug_pw(0, ci, x, None, out, imgs, groups, stats=stats)
        dx = torch.empty((npix, K), dtype=torch.bfloat16, device=dev)
        e.debug_pw(1, ci, r, dy, dx, imgs)
        sc = (torch.rand((groups, K), generator=g) + 0.5).to(dev); sh = (torch.randn((groups, K), generator=g) * 0.1).to(dev)
        gate = torch.rand((imgs, K), generator=g).to(dev)
        outp = torch.empty((npix, M), dtype=torch.bfloat16, device=dev)
        statp = torch.zeros((groups, 2, M), device=dev)
        e.debug_pw(0, ci, x, None, outp, imgs, groups, psc=sc, psh=sh, gate=gate, stats=statp)
        torch.cuda.synchronize()
        key = f"i{imgs}g{groups}c{ci}M{M}K{K}"
        res[key + "_out"] = out.cpu().view(torch.int16).numpy(); res[key + "_dx"] = dx.cpu().view(torch.int16).numpy()
        res[key + "_pro"] = outp.cpu().view(torch.int16).numpy()
        res[key + "_st"] = stats.cpu().numpy(); res[key + "_stp"] = statp.cpu().numpy()
np.savez(sys.argv[1], **res)
print("wrote", sys.argv[1], len(res))
