#!/usr/bin/env python3
"""Headline benchmark: images/sec/client of the FedMLP per-client training step.

Workload (BASELINE.json configs[1]): ICH-shaped synthetic batches, fp32
[B,3,224,224], C = 5, bs = 128, ResNet-18, FedMLP stage-1 step = 2 student
train-mode forwards + 2 frozen-teacher forwards + backward through both views +
Adam (utils/local_training.py:920-967; 28.55 GFLOP/sample).  One client per GPU;
FedAvg (utils/FedAvg.py:7-14) is an RCCL all-reduce of the device-resident state
once per `--round-steps` steps and once at the end of the timed region.
"image" = one dataset sample consumed by the step (it carries two views).

usage: python bench.py --gpus N --steps K --warmup W   (N>1: launched by torchrun)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# kernel families of fm_profile_read (names as rocprofv3 prints them for the ResNet-18 workload, where every
# conv has Ci % 32 == 0 and runs the 32-k-stage instantiation)
KERNEL_NAMES = {0: "igemm_kernel<128,128,2,false,2,32>", 1: "igemm_kernel<64,256,4,false,2,32>",
                2: "igemm_kernel<64,256,4,true,4,16>", 3: "wgrad_kernel<128,128,2>", 4: "wgrad_kernel<64,256,4>"}
NFAM = len(KERNEL_NAMES)
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--classes", type=int, default=5)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--workload", default="stage1", choices=["stage1", "train", "stage2"])
    ap.add_argument("--model", default="Resnet18", choices=["Resnet18", "Efficient_b0"],
                    help="Efficient_b0 = BASELINE configs[3] (use --batch 256)")
    ap.add_argument("--round-steps", type=int, default=40, help="steps per FL round (5000/128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (torch CPU fp32 restatement) timed on this host: a bounded sample
    of the same step (small batch at the full 224x224 size), ~10-30 s of CPU work."""
    from oracle import steps_ref as R
    from tests.helpers import oracle_net
    import copy
    B = 8
    if args.model == "Efficient_b0":
        from oracle.efficientnet_ref import EfficientNetB0Ref
        net = EfficientNetB0Ref(args.classes)
    else:
        net = oracle_net(args.classes, 1037)
    glob = copy.deepcopy(net).eval()
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=3e-5, betas=(0.9, 0.999), weight_decay=5e-4)
    g = torch.Generator().manual_seed(0)
    x1 = torch.randn((B, 3, args.hw, args.hw), generator=g)
    x2 = torch.randn((B, 3, args.hw, args.hw), generator=g)
    y = (torch.rand((B, args.classes), generator=g) < 0.15).float()
    act, neg = [0], list(range(1, args.classes))

    def step():
        if args.workload == "stage1":
            _, z1 = net(x1); _, z2 = net(x2)
            with torch.no_grad():
                _, g1 = glob(x1); _, g2 = glob(x2)
            loss, _, _ = R.loss_stage1(z1, z2, g1, g2, y, act, neg, args.batch, 1)
        else:
            _, z = net(x1)
            loss = R.loss_train(z, y, [1.0] * args.classes, args.batch, args.classes)
        opt.zero_grad(); loss.backward(); opt.step()

    step()                                   # warm-up (allocator, oneDNN primitives)
    n, t0 = 0, time.perf_counter()
    while True:
        step(); n += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or n >= 20:
            break
    return {"value": round(n * B / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{n} oracle {args.workload} steps, batch {B} at 3x{args.hw}x{args.hw} "
                      f"(torch CPU fp32, same step arithmetic, no DataLoader)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from fedmlp_amd.engine import Engine
    from fedmlp_amd import spec
    from fedmlp_amd.fedavg import fedavg_allreduce

    B, C = args.batch, args.classes
    views = 2 if args.workload == "stage1" else 1
    eng = Engine(args.model, C, args.hw, args.hw, views * B, device=str(dev))
    flat, cnt = spec.init_state(args.model, C, 1037)
    eng.set_state(flat, cnt)
    eng.teacher_snapshot()
    eng.adam_reset(3e-5)
    # Efficient_b0: the engine draws drop-connect / dropout multipliers before every train step

    # synthetic client data resident in HBM (seed = reference default, utils/options.py:10)
    g = torch.Generator(device=dev).manual_seed(1037 + rank)
    npool = 4
    x1 = [torch.randn((B, 3, args.hw, args.hw), device=dev, generator=g) for _ in range(npool)]
    x2 = [torch.randn((B, 3, args.hw, args.hw), device=dev, generator=g) for _ in range(npool)]
    active = rank % C                       # 8 clients on 5 classes: class = i mod C
    mask = [1.0 if c == active else 0.0 for c in range(C)]
    ys = []
    for _ in range(npool):
        y = (torch.rand((B, C), device=dev, generator=g) < 0.15).float()
        y = y * torch.tensor(mask, device=dev)
        ys.append(y.contiguous())
    dist_mask = [(torch.rand((B, C), device=dev, generator=g) < 0.5).float() * (1 - torch.tensor(mask, device=dev))
                 for _ in range(npool)]
    losses = torch.zeros(args.steps + args.warmup, device=dev)

    def step(i, k):
        lo = losses[k:k + 1]
        j = i % npool
        if args.workload == "stage1":
            eng.step_stage1(x1[j], x2[j], ys[j], mask, 1, B, lo)
        elif args.workload == "train":
            eng.step_bce(x1[j], ys[j], [1.0] * C, B, lo)
        else:
            eng.step_stage2(x1[j], ys[j], dist_mask[j].contiguous(), lo)

    def fedavg():
        if world > 1:
            fedavg_allreduce(eng, 1.0 / world)

    for i in range(args.warmup):
        step(i, i)
    fedavg()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if not args.no_profile:
        eng.profile_enable(True)
        for f in range(NFAM):
            eng.profile_read(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, args.warmup + i)
        if (i + 1) % args.round_steps == 0 and i + 1 < args.steps:
            fedavg()
    fedavg()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    def measured_traffic(kernel_name):
        """HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 PMC
        passes (profiles/r01/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate
        passes over this same command, FETCH_SIZE doubled per MI355X_MICROARCH.md); null when the
        workload differs from the profiled one."""
        try:
            if args.workload != "stage1" or args.batch != 128 or args.hw != 224 or args.model != "Resnet18":
                return None
            with open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")) as f:
                ks = json.load(f)["kernels"]
            key = "void " + kernel_name.replace(",", ", ") + "(IgemmParams)"
            for k, v in ks.items():
                if k.replace(" ", "") == key.replace(" ", ""):
                    return v["hbm_bytes_per_launch_corrected"]
        except Exception:
            pass
        return None

    roof = None
    if not args.no_profile:
        fams = [eng.profile_read(f) for f in range(NFAM)]
        eng.profile_enable(False)
        dom = max(range(NFAM), key=lambda f: fams[f][1])
        n, ms, fl = fams[dom]
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        roof = {"bound": "mfma", "achieved": round(tf, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": measured_traffic(KERNEL_NAMES[dom]),
                "kernel": KERNEL_NAMES[dom], "launches": n, "avg_launch_ms": round(ms / max(n, 1), 5),
                "all_kernels": {KERNEL_NAMES[f]: {"launches": fams[f][0], "ms": round(fams[f][1], 3),
                                                  "tflops": round(fams[f][2] / max(fams[f][1], 1e-9) / 1e9, 3)}
                                for f in range(NFAM)}}
    lv = losses.cpu().numpy()
    assert np.isfinite(lv).all(), "non-finite loss in the benchmark"

    if rank == 0:
        total = world * B * args.steps
        out = {"metric": "images/sec/client (ICH 224x224 bs=128) at 1/2/4/8 GPUs; mAP vs ref",
               "value": round(total / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": f"FedMLP {args.workload} step, {args.model}, ICH-shaped synthetic "
                                      f"3x{args.hw}x{args.hw}, C={C}, bs={B}, one client per GPU, "
                                      f"FedAvg all-reduce every {args.round_steps} steps + once at end",
                          "images_per_sec_per_client": round(B * args.steps / dt, 3),
                          "views_per_sec": round(views * total / dt, 3), "parallelism": f"clients{world}"},
               "roofline": roof,
               "last_loss": float(lv[-1])}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
