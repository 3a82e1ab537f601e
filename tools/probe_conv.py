#!/usr/bin/env python3
"""Time single conv GEMM launches (forward / dgrad / wgrad) per ResNet-18 layer through the
C-ABI test hooks.  Used with FEDMLP_HIP_LIB=build/probeN/libfedmlp_hip.so timing-only builds
(tools/build_probes.sh) to attribute kernel time; not part of the product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fedmlp_amd.engine import Engine
from fedmlp_amd import spec

imgs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layers = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 6, 11, 16]
ops = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
model = sys.argv[4] if len(sys.argv) > 4 else "Resnet18"
e = Engine(model, 5, 224, 224, imgs)
flat, cnt = spec.init_state(model, 5, 1037)
e.set_state(flat, cnt)
if layers == [-1]:
    layers = list(range(e.debug_num_convs()))
for ci in layers:
    info = e.debug_conv_info(ci)
    x = torch.randn((imgs, info["hin"], info["win"], info["cin_p"]), device="cuda")
    dy = torch.randn((imgs, info["hout"], info["wout"], info["cout_p"]), device="cuda")
    outs = {0: torch.empty_like(dy), 1: torch.empty((imgs, info["hin"], info["win"], info["cin_p"]), device="cuda"),
            2: torch.empty((info["cout_p"], info["Kw"]), device="cuda")}
    gb = 4e-9 * imgs * (info["hin"] * info["win"] * info["cin_p"] + info["hout"] * info["wout"] * info["cout_p"])
    flops = 2.0 * info["hout"] * info["wout"] * info["cout"] * info["cin"] * info["k"] ** 2 * imgs
    for op in ops:
        if op == 1 and ci == 0:
            continue
        for _ in range(3):
            e.debug_conv(op, ci, x, dy, outs[op], imgs)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        t0.record()
        for _ in range(n):
            e.debug_conv(op, ci, x, dy, outs[op], imgs)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / n
        print(f"conv{ci:2d} op{op} cin{info['cin']:4d} cout{info['cout']:4d} k{info['k']} s{info['stride']} "
              f"hout{info['hout']:4d}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF  {gb/ms:6.2f} TB/s (x+y once)", flush=True)
