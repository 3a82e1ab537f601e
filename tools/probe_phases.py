#!/usr/bin/env python3
"""Per-phase cycle counts of one pconv launch (FEDMLP_HIP_LIB=tune/lib_phases.so: pconv.hip with -DPC_PHASES, tools/build_phases.sh).  Timing aid, not part of the product."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fedmlp_amd.engine import Engine
from fedmlp_amd import spec, _lib

imgs = int(sys.argv[1]); layers = [int(a) for a in sys.argv[2].split(",")]; ops = [int(a) for a in sys.argv[3].split(",")]
e = Engine("Resnet18", 5, 224, 224, imgs)
flat, cnt = spec.init_state("Resnet18", 5, 1037)
e.set_state(flat, cnt)
lib = ctypes.CDLL(_lib.LIB_PATH)
names = ["lead wait", "barrier", "prologue reads", "main loop", "decode+lead next", "fix-up", "epilogue", "gap"]
for ci in layers:
    info = e.debug_conv_info(ci)
    x = torch.randn((imgs, info["hin"], info["win"], info["cin_p"]), device="cuda")
    dy = torch.randn((imgs, info["hout"], info["wout"], info["cout_p"]), device="cuda")
    outs = {0: torch.empty_like(dy), 1: torch.empty_like(x)}
    for op in ops:
        for _ in range(3):
            e.debug_conv(op, ci, x, dy, outs[op], imgs)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); e.debug_conv(op, ci, x, dy, outs[op], imgs); t1.record(); torch.cuda.synchronize()
        buf = np.zeros(256 * 8, dtype=np.uint64)
        rc = lib.fm_debug_pconv_prof(buf.ctypes.data_as(ctypes.c_void_p))
        a = buf.reshape(256, 8).astype(np.float64)
        tot = a.sum(1)
        print(f"conv{ci} op{op}: launch {t0.elapsed_time(t1)*1e3:.1f} us (incl. plane making); cycles per block mean {tot.mean():.0f} max {tot.max():.0f}")
        for i, n in enumerate(names):
            print(f"   {n:18s} {a[:, i].mean():10.0f}  {100 * a[:, i].mean() / tot.mean():5.1f} %")
