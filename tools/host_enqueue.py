#!/usr/bin/env python3
"""Host enqueue time vs GPU time of one stage-1 step: is the launch sequence ever the bottleneck?
usage: host_enqueue.py <Resnet18|Efficient_b0> <fp32|bf16> <batch>"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedmlp_amd import spec                      # noqa: E402
from fedmlp_amd.engine import Engine             # noqa: E402

model, prec, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
C = 5
eng = Engine(model, C, 224, 224, 2 * B, precision=prec) if model == "Efficient_b0" else Engine(model, C, 224, 224, 2 * B)
flat, cnt = spec.init_state(model, C, 1037)
eng.set_state(flat, cnt); eng.teacher_snapshot(); eng.adam_reset(3e-5)
g = torch.Generator(device="cuda").manual_seed(1)
x1 = torch.randn((B, 3, 224, 224), device="cuda", generator=g)
x2 = torch.randn((B, 3, 224, 224), device="cuda", generator=g)
y = (torch.rand((B, C), device="cuda", generator=g) < 0.15).float()
lo = torch.zeros(1, device="cuda")
mask = [1.0, 0, 0, 0, 0]
for _ in range(3):
    eng.step_stage1(x1, x2, y, mask, 1, B, lo)
eng.sync(); torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    eng.step_stage1(x1, x2, y, mask, 1, B, lo)
t1 = time.perf_counter()
eng.sync(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{model} {prec} bs {B}: host enqueue {1e3 * (t1 - t0) / N:.2f} ms/step, wall {1e3 * (t2 - t0) / N:.2f} ms/step "
      f"({'HOST-BOUND' if (t1 - t0) > 0.9 * (t2 - t0) else 'host runs ahead'})")
