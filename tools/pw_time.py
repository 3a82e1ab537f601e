#!/usr/bin/env python3
"""Time every pointwise (1x1) convolution of the bf16 EfficientNet-B0 engine alone (fm_debug_pw): forward (train form with BN
partials), forward with the gate prologue, data gradient -- ms and GB/s of the op's own tensors.
usage: python tools/pw_time.py [--imgs 1024] [--reps 5]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fedmlp_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--imgs", type=int, default=1024)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--only", default="", help="comma-separated conv indices")
a = ap.parse_args()
e = Engine("Efficient_b0", 5, 224, 224, a.imgs, precision="bf16", streams=1)
dev = e.device


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.reps):
        fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / a.reps


tot = {"fwd": 0.0, "pro": 0.0, "dgrad": 0.0}
print("conv   M    K     HW | fwd ms GB/s | fwd+prologue ms GB/s | dgrad ms GB/s")
for ci in range(e.debug_num_convs()):
    info = e.debug_conv_info(ci)
    if info["k"] != 1 or (a.only and str(ci) not in a.only.split(",")):
        continue
    M, K, h, w = info["cout_p"], info["cin_p"], info["hout"], info["wout"]
    npix = a.imgs * h * w
    x = torch.randn((npix, K), device=dev).to(torch.bfloat16)
    dy = torch.randn((npix, M), device=dev).to(torch.bfloat16)
    out = torch.empty((npix, M), dtype=torch.bfloat16, device=dev)
    dx = torch.empty((npix, K), dtype=torch.bfloat16, device=dev)
    stats = torch.zeros((1, 2, M), device=dev)
    sc = torch.rand((1, K), device=dev) + 0.5; sh = torch.randn((1, K), device=dev) * 0.1
    gate = torch.rand((a.imgs, K), device=dev)
    gb = npix * (M + K) * 2 / 1e9
    t_f = timed(lambda: e.debug_pw(0, ci, x, None, out, a.imgs, 1))
    t_p = timed(lambda: e.debug_pw(0, ci, x, None, out, a.imgs, 1, psc=sc, psh=sh, gate=gate))
    t_d = timed(lambda: e.debug_pw(1, ci, None, dy, dx, a.imgs))
    tot["fwd"] += t_f; tot["pro"] += t_p; tot["dgrad"] += t_d
    print(f"{ci:3d} {M:5d} {K:5d} {h:3d}x{w:<3d} | {t_f:6.3f} {gb / t_f * 1e3:5.0f} | {t_p:6.3f} {gb / t_p * 1e3:5.0f} | {t_d:6.3f} {gb / t_d * 1e3:5.0f}")
    del x, dy, out, dx
print("totals (ms):", {k: round(v, 3) for k, v in tot.items()})
