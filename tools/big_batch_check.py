"""Largest engines that fit one MI355X (fp32 bs 512, bf16 bs 1024) with the two-stream buffer sets: a few stage-1 steps each."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fedmlp_amd.engine import Engine
for prec, mx in (("fp32", 2048), ("bf16", 4096)):
    e = Engine("Efficient_b0", 14, 224, 224, mx, precision=prec)
    B = mx // 4
    free, tot = torch.cuda.mem_get_info()
    x1 = torch.randn((B, 3, 224, 224), device="cuda"); x2 = torch.randn((B, 3, 224, 224), device="cuda")
    y = (torch.rand((B, 14), device="cuda") < 0.1).float()
    lo = torch.zeros(1, device="cuda")
    e.teacher_snapshot(); e.adam_reset(3e-5)
    for _ in range(3):
        e.step_stage1(x1, x2, y, [1.0] + [0.0] * 13, 3, B, lo)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        e.step_stage1(x1, x2, y, [1.0] + [0.0] * 13, 3, B, lo)
    torch.cuda.synchronize()
    print(prec, "max_images", mx, "B", B, "free GB after create", round(free / 2**30, 1), "ms/step", round((time.perf_counter() - t0) / 5 * 1e3, 1), "loss", lo.item())
    e.close(); del x1, x2
    torch.cuda.empty_cache()
