#!/usr/bin/env python3
"""Headline benchmark: images/sec/client of the FedMLP per-client training step.

Workload (BASELINE.json configs[1]): ICH-shaped synthetic batches, fp32
[B,3,224,224], C = 5, bs = 128, ResNet-18, FedMLP stage-1 step = 2 student
train-mode forwards + 2 frozen-teacher forwards + backward through both views +
Adam (utils/local_training.py:920-967; 28.55 GFLOP/sample).  One client per GPU;
FedAvg (utils/FedAvg.py:7-14) is an RCCL all-reduce of the device-resident state
once per `--round-steps` steps and once at the end of the timed region
(`fm_fedavg_allreduce` of the C ABI).  "image" = one dataset sample consumed by
the step (it carries two views).

usage: python bench.py --gpus N --steps K --warmup W
  N > 1 without a torchrun environment: this process starts the N ranks itself
  (python -m torch.distributed.run ... bench.py, one rank per GPU) before touching
  the GPU and exits with their code.  Rank 0 prints ONE JSON line.

Arithmetic: operands, accumulators and stored tensors are fp32 (`dtype` "f32"); the conv GEMMs form every fp32 product as
six exact bf16 partial products on the bf16 matrix pipe (csrc/split3.h; as close to float64 as the fp32-MFMA
kernels, tests/test_kernels_gpu.py).  `roofline.peak` is that form's roofline (bf16 dense peak / 6); --products 0 (fm_config.reserved[2]
= 1) runs the fp32 matrix pipe instead, and the default run times that too (leg `stage1_fp32_mfma_pipe`, `value_on_fp32_mfma_pipe`).

The ONE line of a default N = 1 run carries, after the headline fields:
  sustained      the same step repeated back to back for >= --sustain-s seconds (steady clock)
  legs           every other BASELINE config and BASELINE.md section-4 leg, each with its own
                 `roofline` and `cpu_baseline` (oracle on this host): the headline step on the fp32 pipe, conv forward bs 256, stage 1
                 with C = 14, LocalUpdate.train, stage-2 step, EfficientNet-B0 fp32 bs 256 / bf16
                 bs 512, prototype pass over N = 5 000, cosine tagging + top-k at N = 5 000 for
                 C = 5 / 14, FedAvg of 8 client states.  --no-legs skips them.

Streams: the engine's default mode enqueues the frozen teacher's forward and the weight gradients on its own
side stream (same bits as one stream, +3-4 %); that is what `value` times.  Co-running kernels stretch each
other's launch windows, so the per-kernel durations behind `roofline` come from a ONE-stream engine: a short
separate pass after the timed region on rank 0 (roofline.measured_in says so), or the timed region itself with
--one-stream -- the form to run under rocprofv3, whose per-kernel averages then agree with the line.

Single legs (same JSON contract, named in config.workload):
  --workload conv_fwd --batch 256     eval-mode forward only: the north-star "conv forward at
                                       bs=256" MFMA-roofline number (928.5 GFLOP per pass)
  --model Efficient_b0 --batch 256    BASELINE configs[3] (fp32, HBM-bound)
  --model Efficient_b0 --precision bf16 --batch 512   configs[4]
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NFAM = 9                              # kernel families of fm_profile_read (kernel_names(); FM_PROFILE_FAMILIES)
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: v_mfma_f32_16x16x32_bf16, dense


def mfma_products(args=None):
    """0 = the conv GEMMs multiply on the fp32 matrix pipe; 6 / 9 = every fp32 product as that many exact bf16 partial
    products on the bf16 matrix pipe, accumulated in fp32 (csrc/split3.h).  The form is a property of the engine handle
    (fm_config.reserved[2], --products); without --products it is the library default (6)."""
    if args is not None and getattr(args, "products", None) is not None:
        return int(args.products)
    from fedmlp_amd import _lib
    return int(_lib.load().fm_mfma_products())


def kernel_names(sp=None, planes=False):
    """kernel families of fm_profile_read, named as rocprofv3 prints them for the ResNet-18 workload (every non-stem conv has
    Ci % 32 == 0: the 32-k-stage instantiations; the last two template arguments = partial products, weight planes)."""
    sp = mfma_products() if sp is None else sp
    if planes and sp:      # planes mode (csrc/pconv.hip, pwgrad.hip): both GEMM operands arrive as bf16 planes
        return {0: f"pconv_kernel<4,4,2,{sp},true>", 1: f"pconv_kernel<2,4,2,{sp},true>", 2: f"stem_rows_kernel<{sp}>",
                3: f"pwgrad_kernel<4,*,{sp}>", 4: f"pwgrad_kernel<2,*,{sp}>", 5: f"wgrad_kernel<64,192,4,3,{sp}> [7x7 stem launch]",
                6: f"pconv_kernel<4,4,2,{sp},false>", 7: f"pconv_kernel<2,4,2,{sp},false>", 8: f"pwgrad_ring_kernel<{sp}>"}
    wp = 1 if sp else 0
    return {0: f"igemm_kernel<128,128,2,0,2,32,{sp},{wp}>", 1: f"igemm_kernel<64,192,4,0,2,32,{sp},{wp}>",
            2: f"igemm_kernel<64,256,4,2,2,32,{sp},0>" if sp else "igemm_kernel<64,256,4,2,4,16,0,0>", 3: f"wgrad_kernel<128,128,2,4,{sp}>", 4: f"wgrad_kernel<64,192,4,3,{sp}>",
            5: f"wgrad_kernel<64,192,4,3,{sp}> [7x7 stem launch]", 6: "(planes mode only)", 7: "(planes mode only) ", 8: "(planes mode only)  "}


def mfma_peak(sp=None):
    """(peak TFLOP/s of fp32-equivalent work, note): the fp32 pipe's dense peak, or the bf16 pipe's divided by the partial
    products each fp32 product costs."""
    sp = mfma_products() if sp is None else sp
    if not sp:
        return PEAK_F32_MFMA_TFLOPS, "v_mfma_f32_16x16x4_f32 dense peak"
    return round(PEAK_BF16_MFMA_TFLOPS / sp, 2), (
        f"v_mfma_f32_16x16x32_bf16 dense peak {PEAK_BF16_MFMA_TFLOPS:.0f} TFLOP/s / {sp} bf16 partial products per fp32 product "
        f"(`achieved` counts algorithmic fp32 FLOPs, not the {sp}x bf16 FLOPs issued)")


ARITHMETIC = {0: "fp32 operands, fp32 products and accumulation on v_mfma_f32_16x16x4_f32",
              6: "fp32 operands and fp32 accumulation; every product as 6 exact bf16 partial products on v_mfma_f32_16x16x32_bf16 "
                 "(3-way exact split of both operands, the 3 partial products of at most 2^-24 of the product left out (at most 2 more unit roundoffs in a K-term sum that carries K); PER CONV the "
                 "error against float64 is within 1.25x of the fp32-MFMA form's: tests/test_kernels_gpu.py::test_split_products_are_fp32_accurate; over a "
                 "whole step the bf16 pipe's accumulation shows in sums of ~1e6 cancelling terms (DESIGN.md section 2); "
                 "fm_config.reserved[2] = 1 selects the fp32 pipe)",
              9: "fp32 operands and fp32 accumulation; every product as its 9 exact bf16 partial products on "
                 "v_mfma_f32_16x16x32_bf16 (fm_config.reserved[2] = 2)"}
PEAK_HBM_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)
# minimum activation bytes per image of one EfficientNet-B0 forward
EFFNET_FWD_BYTES = {"fp32": 73.8e6, "bf16": 36.9e6}
RESNET_FWD_FLOP = 3.627e9             # 2 x 1 813 561 344 conv MACs per 224x224 image
STEP_FLOP = {"stage1": 28.55e9, "train": 10.65e9, "stage2": 10.65e9}    # SURVEY.md 8(d), per sample
PMC_FILES = [os.path.join("profiles", r, "pmc_traffic.json") for r in ("r06", "r05", "r04", "r03", "r02")]
METRIC = "images/sec/client (ICH 224x224 bs=128) at 1/2/4/8 GPUs; mAP vs ref"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed steps (default ~5 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--classes", type=int, default=5)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--workload", default="stage1", choices=["stage1", "train", "stage2", "conv_fwd"])
    ap.add_argument("--model", default="Resnet18", choices=["Resnet18", "Efficient_b0"],
                    help="Efficient_b0 = BASELINE configs[3] (use --batch 256)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="activation storage of the Efficient_b0 path (configs[4]: bf16, --batch 512)")
    ap.add_argument("--round-steps", type=int, default=40, help="steps per FL round (5000/128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-legs", action="store_true",
                    help="headline line only (the legs run by default at N = 1 for the default workload)")
    ap.add_argument("--legs", default="all",
                    help="comma list of legs to run (conv_fwd_bs256, stage1_c14, train, stage2, effnet_fp32_bs256, "
                         "effnet_bf16_bs512, proto_pass, cos_tag_c5, cos_tag_c14, fedavg8) or 'all'")
    ap.add_argument("--leg-steps", type=int, default=30, help="timed steps of each step-type leg")
    ap.add_argument("--sustain-s", type=float, default=10.0,
                    help="seconds of back-to-back repeats of the headline step after the timed region (N = 1, default "
                         "workload; 0 = off)")
    ap.add_argument("--products", type=int, default=None, choices=[0, 6, 9],
                    help="product form of the fp32 conv GEMMs (fm_config.reserved[2]): 6 = library default, 0 = fp32 matrix pipe")
    ap.add_argument("--one-stream", action="store_true",
                    help="engine stream mode 1: every kernel on one stream, roofline measured inside the timed region "
                         "(use this under rocprofv3: per-kernel durations of co-running kernels describe no kernel alone)")
    ap.add_argument("--roofline-steps", type=int, default=12,
                    help="steps of the separate one-stream roofline pass (default mode only)")
    ap.add_argument("--cpu-threads", default="16,32,64,all",
                    help="thread counts tried by the cpu_baseline leg (the best one is reported)")
    ap.add_argument("--profile-every", type=int, default=4,
                    help="HIP events bracket the conv launches of every Nth timed step (each pair costs ~3 us of "
                         "stream time: on every step that is 0.7 ms of a 39-ms ResNet-18 step)")
    return ap.parse_args(argv)


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def thread_candidates(spec_str):
    cores, avail = physical_cores(), torch.get_num_threads()
    cand = []
    for t in str(spec_str).split(","):
        n = max(avail, cores) if t.strip() == "all" else int(t)
        n = max(1, min(n, max(avail, cores)))
        if n not in cand:
            cand.append(n)
    return cand


def timed_cpu(fn, units_per_call, cand, budget_s=6.0, max_calls=10):
    """best-of-thread-counts rate of `fn` on this host: one warm-up call, then calls until budget_s / max_calls"""
    avail = torch.get_num_threads()
    tried, best = {}, None
    for nt in cand:
        torch.set_num_threads(nt)
        fn()                                 # warm-up (allocator, oneDNN primitives for this thread count)
        n, t0 = 0, time.perf_counter()
        while True:
            fn(); n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or n >= max_calls:
                break
        rate = n * units_per_call / dt
        tried[str(nt)] = round(rate, 3)
        if best is None or rate > best[0]:
            best = (rate, nt, n, dt)
    torch.set_num_threads(avail)
    return best, tried


def cpu_baseline(args, cand=None, budget_s=6.0):
    """The oracle (torch CPU fp32 restatement of the same step arithmetic) timed on this host: a
    bounded sample, <= budget_s of CPU work per thread count tried; the best thread count is reported
    (all cores oversubscribe oneDNN on a 128-core host).  The micro-batch is the reference's own
    CPU-runnable batch (configs[0]: bs 32) unless --batch is smaller; the rate is per image, so it
    compares directly with `value`."""
    from oracle import steps_ref as R
    from tests.helpers import oracle_net
    B = min(32, args.batch)
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
    dist_cls = (torch.rand((B, args.classes), generator=g) < 0.5).float()
    act, neg = [0], list(range(1, args.classes))

    def step():
        if args.workload == "conv_fwd":
            with torch.no_grad():
                glob(x1)
            return
        if args.workload == "stage1":
            _, z1 = net(x1); _, z2 = net(x2)
            with torch.no_grad():
                _, g1 = glob(x1); _, g2 = glob(x2)
            loss, _, _ = R.loss_stage1(z1, z2, g1, g2, y, act, neg, args.batch, 1)
        elif args.workload == "stage2":
            _, z = net(x1)
            loss = R.loss_stage2(z, y, dist_cls)
        else:
            _, z = net(x1)
            loss = R.loss_train(z, y, [1.0] * args.classes, args.batch, args.classes)
        opt.zero_grad(); loss.backward(); opt.step()

    cand = cand or thread_candidates(args.cpu_threads)
    (rate, nt, n, dt), tried = timed_cpu(step, B, cand, budget_s)
    return {"value": round(rate, 3), "unit": "images/sec", "cores": physical_cores(), "threads": nt, "micro_batch": B,
            "images_per_sec_by_threads": tried, "kind": "port",
            "sample": f"best of thread counts {cand}: {n} oracle {args.workload} steps at micro-batch {B} (config batch "
                      f"{args.batch}; the loss normaliser is the config batch), 3x{args.hw}x{args.hw}, torch CPU fp32, same "
                      f"step arithmetic, no DataLoader; {dt:.1f} s at {nt} threads"}


def family_traffic(prefix, args):
    """launch-weighted mean HBM bytes per launch over every profiled kernel whose name contains `prefix`"""
    try:
        doc = pmc_doc(args)[0]
        norm = lambda t: t.replace(" ", "").replace("false", "0").replace("true", "1")   # (older profiles: bool STEM)
        ks = [v for k, v in doc["kernels"].items() if norm(prefix) in norm(k)]
        n = sum(v["launches"] for v in ks)
        return int(sum(v["launches"] * v["hbm_bytes_per_launch_corrected"] for v in ks) / n) if n else None
    except Exception:
        return None


def pmc_doc(args):
    """(entry of this workload, file it came from) from the newest committed rocprofv3 PMC summary that has one"""
    for rel in PMC_FILES:
        try:
            with open(os.path.join(ROOT, rel)) as f:
                doc = json.load(f).get(workload_key(args))
            if doc is not None:
                return doc, rel
        except Exception:
            continue
    return None, PMC_FILES[0]


def read_families(eng):
    return [eng.profile_read(f) for f in range(NFAM)]


def one_stream_roofline_pass(args, eng, step_fn, max_images, dev):
    """ResNet-18: per-kernel durations from a second engine in stream mode 1 (every kernel alone on the chip), same state,
    same inputs, HIP events around every conv launch of `--roofline-steps` steps.  Returns (families, ms_per_step), or
    (None, None) when a second engine does not fit in free device memory."""
    from fedmlp_amd.engine import Engine
    free_b, _ = torch.cuda.mem_get_info(torch.device(dev))
    if free_b < (24 << 30):                  # a bs-128 two-view ResNet-18 engine holds ~12 GB
        return None, None
    e1 = Engine(args.model, args.classes, args.hw, args.hw, max_images, device=dev, streams=1, products=eng.products)
    try:
        flat, cnt = eng.get_state()
        e1.set_state(flat, cnt)
        e1.teacher_snapshot()
        e1.adam_reset(3e-5)
        n, w = max(1, args.roofline_steps), 3
        for i in range(w):
            step_fn(i, i, e1)
        torch.cuda.synchronize()
        e1.profile_enable(True)
        read_families(e1)
        t0 = time.perf_counter()
        for i in range(n):
            step_fn(i, w + i, e1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fams = read_families(e1)
        e1.profile_enable(False)
    finally:
        e1.close()
    return fams, dt / n * 1e3


def measured_traffic(kernel_name, args):
    """HBM-side bytes per launch of the dominant kernel.  NOT measured in this run: read from the
    committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate passes over this
    same command, FETCH_SIZE doubled per MI355X_MICROARCH.md); null when the workload differs from
    the profiled one."""
    doc, _ = pmc_doc(args)
    if doc is None:
        return None
    norm = lambda t: t.replace(" ", "").replace("false", "0").replace("true", "1")   # (older profiles: bool STEM)
    key = norm(kernel_name)
    for k, v in doc["kernels"].items():
        if key in norm(k):
            return v["hbm_bytes_per_launch_corrected"]
    return None


def workload_key(args):
    return f"{args.model}/{args.precision}/{args.workload}/bs{args.batch}/hw{args.hw}/C{args.classes}"


STREAM_MODE_NAMES = {0: "default: frozen teacher + weight gradients on the engine's side stream",
                     1: "one stream", 2: "frozen teacher on the engine's side stream"}


def run_workload(args, rank, world, dev, dist, cpu_cand=None, cpu_budget_s=6.0):
    """One bench line (the contract of the module docstring) for `args`; returns the dict on rank 0, None elsewhere."""
    from fedmlp_amd.engine import Engine
    from fedmlp_amd import spec
    from fedmlp_amd.fedavg import fedavg_allreduce, comm_init

    B, C = args.batch, args.classes
    views = 2 if args.workload == "stage1" else 1
    eng = Engine(args.model, C, args.hw, args.hw, views * B, device=str(dev), precision=args.precision,
                 streams=1 if args.one_stream else 0, products=getattr(args, "products", None))
    sp = eng.products
    planes = eng.planes
    flat, cnt = spec.init_state(args.model, C, 1037)
    eng.set_state(flat, cnt)
    eng.teacher_snapshot()
    eng.adam_reset(3e-5)
    rccl_ranks = 1
    if world > 1:
        # RCCL communicator inside the C-ABI library (fm_comm_init, dlopen'ed librccl)
        # a record of the wrong path is worse than no record: --gpus N measures the in-library RCCL all-reduce or fails
        rccl_ranks = comm_init(eng)
        if rccl_ranks != world:
            raise SystemExit(f"[bench] library RCCL communicator has {rccl_ranks} ranks, --gpus {world}")
    # Efficient_b0: the engine draws drop-connect / dropout multipliers before every train step

    # synthetic client data resident in HBM (seed = reference default, utils/options.py:10)
    g = torch.Generator(device=dev).manual_seed(1037 + rank)
    npool = 4
    x1 = [torch.randn((B, 3, args.hw, args.hw), device=dev, generator=g) for _ in range(npool)]
    x2 = [torch.randn((B, 3, args.hw, args.hw), device=dev, generator=g) for _ in range(npool)] \
        if views == 2 else x1
    active = rank % C                       # 8 clients on 5 classes: class = i mod C
    mask = [1.0 if c == active else 0.0 for c in range(C)]
    ys = []
    for _ in range(npool):
        y = (torch.rand((B, C), device=dev, generator=g) < 0.15).float()
        y = y * torch.tensor(mask, device=dev)
        ys.append(y.contiguous())
    dist_mask = [((torch.rand((B, C), device=dev, generator=g) < 0.5).float()
                  * (1 - torch.tensor(mask, device=dev))).contiguous() for _ in range(npool)]
    sustain = args.sustain_s if (world == 1 and getattr(args, "_sustain", False)) else 0.0
    nloss = args.steps + args.warmup
    losses = torch.zeros(nloss, device=dev)
    feat_out = torch.empty((B, eng.feature_dim), device=dev)
    logit_out = torch.empty((B, C), device=dev)

    def step(i, k, en=None):
        en = en or eng
        lo = losses[k % nloss:k % nloss + 1]
        j = i % npool
        if args.workload == "stage1":
            en.step_stage1(x1[j], x2[j], ys[j], mask, 1, B, lo)
        elif args.workload == "train":
            en.step_bce(x1[j], ys[j], [1.0] * C, B, lo)
        elif args.workload == "stage2":
            en.step_stage2(x1[j], ys[j], dist_mask[j], lo)
        else:
            en.forward_eval_into(x1[j], feat_out, logit_out)

    # per-kernel HIP events inside the timed region only where the launches of one stream do not overlap others'
    in_region = (not args.no_profile) and (args.one_stream or args.workload == "conv_fwd" or args.model == "Efficient_b0")
    ar_ev = []                                   # (start, end) events around every FedAvg all-reduce of the timed region

    def fedavg(timed=False):
        if world > 1 and args.workload != "conv_fwd":
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            fedavg_allreduce(eng, 1.0 / world)
            if timed:
                b.record()
                ar_ev.append((a, b))

    for i in range(args.warmup):
        step(i, i)
    fedavg()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if in_region:
        eng.profile_enable(True)
        read_families(eng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pe = max(1, args.profile_every)
    for i in range(args.steps):
        if in_region and pe > 1:
            eng.profile_enable(i % pe == 0)
        step(i, args.warmup + i)
        if (i + 1) % args.round_steps == 0 and i + 1 < args.steps:
            fedavg(True)
    fedavg(True)
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0           # this rank's own steps + all-reduces (before the closing barrier)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        ar_ms = sum(a.elapsed_time(b) for a, b in ar_ev)
        mine = torch.tensor([dt_rank * 1e3 / args.steps, ar_ms, float(len(ar_ev))], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "ms_per_step_incl_allreduce": round(v[0].item(), 4),
                     "allreduce_ms_total": round(v[1].item(), 3), "allreduces": int(v[2].item())}
                    for r, v in enumerate(allr)]
        if args.workload != "conv_fwd":
            # outside the timed region: the ranks have just averaged their states -- they must now hold the same bits
            # (the first N > 1 run on hardware is also a correctness run of fm_fedavg_allreduce)
            from fedmlp_amd.fedavg import state_agreement
            agree, worst = state_agreement(eng)
            if not agree:
                raise SystemExit(f"[bench] rank {rank}: the ranks' states differ after the FedAvg all-reduce "
                                 f"(checksum spread {worst:.3e} of the magnitude sum)")
    fams_region = None
    if in_region:
        fams_region = read_families(eng)
        eng.profile_enable(False)
    lv = losses.cpu().numpy()
    assert np.isfinite(lv).all(), "non-finite loss in the benchmark"

    # ---- sustained repeat: the same step back to back for >= sustain seconds (steady clock; a sub-second timed region never
    # reaches the state a training run lives in, and the driver's gpu_busy sampler cannot see it)
    sustained = None
    if sustain > 0:
        chunk, n_done = 40, 0
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        first_chunk_ms = last_chunk_ms = None
        while True:
            c0 = time.perf_counter()
            for i in range(chunk):
                step(n_done + i, n_done + i)
            torch.cuda.synchronize()
            c1 = time.perf_counter()
            n_done += chunk
            last_chunk_ms = (c1 - c0) / chunk * 1e3
            if first_chunk_ms is None:
                first_chunk_ms = last_chunk_ms
            if c1 - s0 >= sustain:
                break
        sdt = time.perf_counter() - s0
        sustained = {"sustained_ms_per_step": round(sdt / n_done * 1e3, 4), "steps": n_done, "seconds": round(sdt, 2),
                     "images_per_sec": round(B * n_done / sdt, 2), "first_40_steps_ms_per_step": round(first_chunk_ms, 4),
                     "last_40_steps_ms_per_step": round(last_chunk_ms, 4),
                     "note": "same engine, same step and inputs as the timed region, one host sync per 40 steps"}
        assert np.isfinite(losses.cpu().numpy()).all(), "non-finite loss in the sustained repeat"

    roof = None
    if not args.no_profile and rank == 0:        # (ranks != 0 build no second engine: the line is rank 0's)
        sampled, one_stream_ms = None, None
        if in_region:
            fams = fams_region
            measured_in = "the timed region" + (" (engine stream mode 1: one stream)" if args.one_stream else "")
            sampled = (f"HIP events around the conv launches of every {pe}th step of the timed region" if pe > 1
                       else "HIP events around every conv launch of the timed region")
        else:
            fams, one_stream_ms = one_stream_roofline_pass(args, eng, step, views * B, str(dev))
            measured_in = (f"a separate one-stream pass of {max(1, args.roofline_steps)} steps after the timed region (second "
                           f"engine, stream mode 1, same state and inputs): in the default two-stream mode co-running kernels "
                           f"stretch each other's launch windows; `python bench.py --one-stream` times that mode itself")
            sampled = "HIP events around every conv launch of the one-stream pass"
        pmc_file = pmc_doc(args)[1]
        allk = None
        if fams is not None:
            allk = {kernel_names(sp, planes)[f]: {"launches": fams[f][0], "ms": round(fams[f][1], 3),
                                      "tflops": round(fams[f][2] / max(fams[f][1], 1e-9) / 1e9, 3)}
                    for f in range(NFAM)}
        if args.model == "Efficient_b0":
            # HBM-bound model: whole-step algorithmic activation bytes over the step time
            n_fwd = {"stage1": 4 * B, "train": B, "stage2": B, "conv_fwd": B}[args.workload]
            n_bwd = {"stage1": 2 * B, "train": B, "stage2": B, "conv_fwd": 0}[args.workload]
            alg = (n_fwd + 2 * n_bwd) * EFFNET_FWD_BYTES[args.precision]
            gbs = alg / (dt / args.steps) / 1e9
            roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": measured_traffic("whole_step", args),
                    "traffic_source": pmc_file + " (separate rocprofv3 --pmc passes, not this run)",
                    "kernel": "whole step (all kernels of one step; no single kernel dominates)",
                    "algorithmic_bytes_per_step": alg, "mfma_kernels": allk}
        elif fams is None:
            steps_alg = STEP_FLOP[args.workload] * B
            tf = steps_alg / (dt / args.steps) / 1e12
            peak, peak_note = mfma_peak(sp)
            roof = {"bound": "mfma", "achieved": round(tf, 3), "peak": peak, "unit": "TFLOP/s", "peak_basis": peak_note,
                    "frac": round(tf / peak, 4), "traffic": None, "kernel": "whole step",
                    "measured_in": "the timed region (no free device memory for the one-stream per-kernel pass)"}
        else:
            if args.workload == "conv_fwd":
                use = [0, 1, 2, 6, 7]
                n = sum(fams[f][0] for f in use); ms = sum(fams[f][1] for f in use)
                fl = sum(fams[f][2] for f in use)
                name = ("pconv_kernel<*> + the stem's igemm_kernel" if planes and sp else "igemm_kernel<*>") + " (all 20 conv forwards of the eval pass)"
            else:
                dom = max(range(NFAM), key=lambda f: fams[f][1])
                n, ms, fl = fams[dom]
                name = kernel_names(sp, planes)[dom]
            tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            peak, peak_note = mfma_peak(sp)
            roof = {"bound": "mfma", "achieved": round(tf, 3), "peak": peak, "unit": "TFLOP/s", "peak_basis": peak_note,
                    "frac": round(tf / peak, 4), "frac_of_fp32_mfma_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                    "traffic": measured_traffic(name, args),
                    "traffic_source": pmc_file + " (separate rocprofv3 --pmc passes, not this run)",
                    "kernel": name, "launches": n, "avg_launch_ms": round(ms / max(n, 1), 5),
                    "measured_in": measured_in, "sampled": sampled, "all_kernels": allk}
            if one_stream_ms is not None:
                # `frac` describes the dominant kernel alone on the chip (one-stream pass); the two-stream timed region behind
                # `value` is described by whole_step_frac
                roof["frac_one_stream"] = roof["frac"]
                roof["one_stream_ms_per_step"] = round(one_stream_ms, 4)
            if args.workload in STEP_FLOP:
                steps_alg = STEP_FLOP[args.workload] * B
                roof["whole_step_tflops"] = round(steps_alg / (dt / args.steps) / 1e12, 3)
                roof["whole_step_frac"] = round(steps_alg / (dt / args.steps) / 1e12 / peak, 4)
                roof["whole_step_frac_of_fp32_mfma_peak"] = round(steps_alg / (dt / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
            if args.workload == "conv_fwd":
                roof["traffic"] = family_traffic("pconv_kernel<" if planes and sp else "igemm_kernel<", args)     # mean over the conv launches of a pass
                roof["algorithmic_flop_per_pass"] = RESNET_FWD_FLOP * B
                roof["whole_pass_tflops"] = round(RESNET_FWD_FLOP * B / (dt / args.steps) / 1e12, 3)
    effective_mode = eng.stream_mode
    eng.close()
    del x1, x2, ys, dist_mask
    torch.cuda.empty_cache()

    if rank != 0:
        return None
    total = world * B * args.steps
    desc = {"stage1": "FedMLP stage1 step", "train": "LocalUpdate.train step", "stage2": "FedMLP stage2 step",
            "conv_fwd": "eval-mode forward pass (conv forward)"}[args.workload]
    out = {"metric": METRIC,
           "value": round(total / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16" if (args.model == "Efficient_b0" and args.precision == "bf16") else "f32",
           "data": "synthetic",
           "config": {"workload": f"{desc}, {args.model}, ICH-shaped synthetic "
                                  f"3x{args.hw}x{args.hw}, C={C}, bs={B}, one client per GPU, "
                                  f"FedAvg all-reduce every {args.round_steps} steps + once at end",
                      "images_per_sec_per_client": round(B * args.steps / dt, 3),
                      "views_per_sec": round(views * total / dt, 3), "parallelism": f"clients{world}",
                      "stream_mode": STREAM_MODE_NAMES.get(effective_mode, str(effective_mode)) +
                                     (" (--one-stream)" if args.one_stream else ""),
                      "rccl_ranks": rccl_ranks, "ranks_agree_after_allreduce": (True if world > 1 and args.workload != "conv_fwd" else None),
                      "timed_region_s": round(dt, 3),
                      "arithmetic": ("bf16 storage, fp32 accumulation" if (args.model == "Efficient_b0" and args.precision == "bf16")
                                     else ARITHMETIC[sp])},
           "roofline": roof,
           "last_loss": float(lv[-1]), "_products": sp}
    if per_rank is not None:
        out["config"]["per_rank"] = per_rank
    if sustained is not None:
        out["sustained"] = sustained
        out["sustained_ms_per_step"] = sustained["sustained_ms_per_step"]
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, cpu_cand, cpu_budget_s)
    else:
        out["cpu_baseline"] = None
    return out


# ======================================= legs beyond the headline (N = 1) ===========================================
def _gpu_time(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def leg_proto_pass(dev, cand, cpu=True, C=5, N=5000, bs=128):
    """Prototype + t pass of utils/local_training.py:971-1002 over an ICH-sized local set: eval forward in batches of 4*bs,
    per-class feature sums and the p<L / p>U counts on the device (fm_proto_accumulate), one finalize."""
    from fedmlp_amd.engine import Engine
    from fedmlp_amd import spec
    from oracle import steps_ref as R
    from tests.helpers import oracle_net
    eng = Engine("Resnet18", C, 224, 224, 4 * bs, device=str(dev))
    flat, cnt = spec.init_state("Resnet18", C, 1037)
    eng.set_state(flat, cnt)
    g = torch.Generator(device=dev).manual_seed(1037)
    pool = [torch.randn((4 * bs, 3, 224, 224), device=dev, generator=g) for _ in range(2)]
    act, neg = [1.0] + [0.0] * (C - 1), [0.0] + [1.0] * (C - 1)
    labels = (torch.rand((N, C), device=dev, generator=g) < 0.15).float() * torch.tensor(act, device=dev)
    feat = torch.empty((4 * bs, eng.feature_dim), device=dev)
    logit = torch.empty((4 * bs, C), device=dev)
    spans = [(s, min(s + 4 * bs, N)) for s in range(0, N, 4 * bs)]

    def one_pass():
        eng.proto_reset()
        for k, (a, b) in enumerate(spans):
            n = b - a
            eng.forward_eval_into(pool[k % 2][:n], feat[:n], logit[:n])
            eng.proto_accumulate(feat[:n], logit[:n], labels[a:b].contiguous(), act, neg, 0.3, 0.7)
        return eng.proto_finalize(0, N, act)

    dt = _gpu_time(one_pass, 3, 1)
    t, proto = one_pass()
    assert np.isfinite(proto[:2]).all() and np.isfinite(t).all()
    eng.close()
    tf = RESNET_FWD_FLOP * N / dt / 1e12
    peak, peak_note = mfma_peak()
    out = {"config": {"workload": f"prototype + t pass (utils/local_training.py:971-1002), ResNet-18, N={N} local samples, "
                                   f"3x224x224, C={C}, eval batches of {4 * bs}"},
           "value": round(N / dt, 2), "unit": "images/sec", "ms_per_pass": round(dt * 1e3, 3), "dtype": "f32",
           "roofline": {"bound": "mfma", "achieved": round(tf, 3), "peak": peak, "unit": "TFLOP/s", "peak_basis": peak_note,
                        "frac": round(tf / peak, 4), "frac_of_fp32_mfma_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                        "kernel": "whole pass (eval forward 3.627 GFLOP/sample + prototype sums)"},
           "cpu_baseline": None}
    if not cpu:
        return out
    # CPU: the oracle's eval forward + prototype_pass over a bounded sample of 4 batches of 32
    net = oracle_net(C, 1037).eval()
    gx = torch.Generator().manual_seed(0)
    xs = [torch.randn((32, 3, 224, 224), generator=gx) for _ in range(4)]
    ys = [(torch.rand((32, C), generator=gx) < 0.15).float() * torch.tensor(act) for _ in range(4)]

    def cpu_pass():
        def gen():
            with torch.no_grad():
                for x, y in zip(xs, ys):
                    f, z = net(x)
                    yield f, z, y
        R.prototype_pass(gen(), C, [0], list(range(1, C)), 0.3, 0.7, 128, False)
    (rate, nt, n, cdt), tried = timed_cpu(cpu_pass, 128, cand, 5.0, 6)
    out["cpu_baseline"] = {"value": round(rate, 3), "unit": "images/sec", "cores": physical_cores(), "threads": nt,
                           "kind": "port", "images_per_sec_by_threads": tried,
                           "sample": f"{n} oracle passes (eval forward + oracle.prototype_pass) over 128 samples in "
                                     f"batches of 32; {cdt:.1f} s at {nt} threads"}
    return out


def leg_cos_tag(dev, cand, C, cpu=True, N=5000, D=512):
    """Cosine tagging + stable top-/bottom-k selection of every missing class (utils/local_training.py:1052-1112,
    1417-1435; utils/utils.py:24-35) over an ICH-sized feature matrix."""
    from fedmlp_amd.engine import Engine
    from oracle import steps_ref as R
    eng = Engine("Resnet18", C, 64, 64, 8, device=str(dev), streams=1)
    g = torch.Generator(device=dev).manual_seed(7)
    f = torch.randn((N, D), device=dev, generator=g).abs().contiguous()        # features are post-ReLU pooled: >= 0
    proto = torch.randn((2 * C, D), device=dev, generator=g).abs().contiguous()
    missing = list(range(1, C))

    def tag_all():
        # one cosine-tagging launch over all missing classes, one selection launch pair, one device-to-host read
        sims = eng.cos_tag(f, proto, missing)
        return sum(len(t) + len(b) for t, b in eng.select_topk_rows(sims, [None] * len(missing), 0.005, 0.01))

    dt = _gpu_time(tag_all, 20, 3)
    picks = tag_all()
    eng.close()
    alg = len(missing) * (N * D * 4 + 2 * D * 4 + N * 4 * 2)        # per class: read f and two prototype rows, write + read sim
    gbs = alg / dt / 1e9
    out = {"config": {"workload": f"cosine tagging + stable top-k selection (utils/local_training.py:1052-1112), N={N} "
                                  f"features x D={D}, C={C}: {len(missing)} missing classes per call, picks read back"},
           "value": round(N / dt, 1), "unit": "samples/sec (all missing classes tagged)", "ms_per_call": round(dt * 1e3, 4),
           "picks": picks, "dtype": "f32",
           "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 5), "traffic": None, "algorithmic_bytes_per_call": alg,
                        "kernel": "cos_tag + count + rank-select launches of one call (all classes at once, one device-to-host "
                                  "read of the picks)"},
           "cpu_baseline": None}
    if not cpu:
        return out
    fc, pc = f.cpu(), proto.cpu()
    idx = list(range(N))

    def cpu_tag():
        for cls in missing:
            sim = R.cosine_diff(fc, pc[2 * cls], pc[2 * cls + 1])
            R.select_for_class(sim.tolist(), idx, 0.005, 0.01)
    (rate, nt, n, cdt), tried = timed_cpu(cpu_tag, N, cand[:1], 3.0, 20)
    out["cpu_baseline"] = {"value": round(rate, 1), "unit": "samples/sec (all missing classes tagged)",
                           "cores": physical_cores(), "threads": nt, "kind": "port",
                           "sample": f"{n} oracle calls (cosine_diff + select_for_class, Python sorted like utils/utils.py) "
                                     f"at the same N, D, C; {cdt:.1f} s at {nt} threads"}
    return out


def leg_fedavg8(dev, cand, cpu=True, C=5, K=8):
    """FedAvg (utils/FedAvg.py:7-14) of K = 8 ResNet-18 client states held on one GPU: fm_fedavg_fold."""
    from fedmlp_amd.engine import Engine
    from fedmlp_amd import spec
    from oracle import steps_ref as R
    eng = Engine("Resnet18", C, 64, 64, 8, device=str(dev), streams=1)
    flat, cnt = spec.init_state("Resnet18", C, 1037)
    eng.set_state(flat, cnt)
    st = eng.state_tensor()
    g = torch.Generator(device=dev).manual_seed(3)
    states = [(st + 1e-3 * torch.randn(st.shape, device=dev, generator=g)).contiguous() for _ in range(K)]
    n_k = [5000.0] * K
    out = torch.empty_like(st)
    dt = _gpu_time(lambda: eng.fedavg_fold(states, n_k, out), 50, 5)
    alg = (K + 1) * st.numel() * 4
    gbs = alg / dt / 1e9
    assert torch.isfinite(out).all()
    eng.close()
    out = {"config": {"workload": f"FedAvg of {K} ResNet-18 client states (utils/FedAvg.py:7-14; 11.19 M fp32 entries = "
                                  f"44.75 MB per client), reference order and roundings, states resident in HBM"},
           "value": round(1.0 / dt, 1), "unit": "aggregations/sec", "ms_per_aggregation": round(dt * 1e3, 4), "dtype": "f32",
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "algorithmic_bytes_per_call": alg,
                        "kernel": "fedavg_fold_kernel"},
           "cpu_baseline": None}
    if not cpu:
        return out
    sds = []
    for k in range(K):
        sd = spec.flat_to_state_dict("Resnet18", C, flat + np.float32(1e-3 * k), cnt)
        sds.append({kk: torch.as_tensor(v) for kk, v in sd.items()})
    (rate, nt, n, cdt), tried = timed_cpu(lambda: R.fedavg(sds, [5000] * K), 1, cand[:1], 3.0, 10)
    out["cpu_baseline"] = {"value": round(rate, 3), "unit": "aggregations/sec", "cores": physical_cores(), "threads": nt,
                           "kind": "port",
                           "sample": f"{n} oracle.fedavg calls over {K} state_dicts (122 entries each); {cdt:.1f} s at "
                                     f"{nt} threads"}
    return out


def run_legs(args, dev, cand):
    """Every other BASELINE config / BASELINE.md section-4 leg on this GPU, one after the other (each builds and closes its
    own engine).  A leg that fails reports its error instead of taking the headline line down with it."""
    want = None if args.legs == "all" else set(args.legs.split(","))
    cpu = not args.no_cpu_baseline

    def step_leg(**kw):
        a = copy.copy(args)
        a.steps, a.warmup, a.roofline_steps, a.one_stream, a._sustain = args.leg_steps, 5, 6, False, False
        for k, v in kw.items():
            setattr(a, k, v)
        out = run_workload(a, 0, 1, dev, None, cand, 5.0)
        for k in ("metric", "n_gpus", "higher_is_better", "scaling", "vs_baseline", "data"):
            out.pop(k, None)
        return out

    def fp32_pipe_leg():
        # the headline workload on a handle whose conv GEMMs multiply on the fp32 matrix pipe (fm_config.reserved[2] = 1)
        return step_leg(products=0)

    table = [
        ("stage1_fp32_mfma_pipe", fp32_pipe_leg),
        ("conv_fwd_bs256", lambda: step_leg(workload="conv_fwd", batch=256, steps=60)),
        ("stage1_c14", lambda: step_leg(classes=14)),
        ("train", lambda: step_leg(workload="train")),
        ("stage2", lambda: step_leg(workload="stage2")),
        ("effnet_fp32_bs256", lambda: step_leg(model="Efficient_b0", batch=256)),
        ("effnet_bf16_bs512", lambda: step_leg(model="Efficient_b0", precision="bf16", batch=512, classes=14)),
        ("proto_pass", lambda: leg_proto_pass(dev, cand, cpu)),
        ("cos_tag_c5", lambda: leg_cos_tag(dev, cand, 5, cpu)),
        ("cos_tag_c14", lambda: leg_cos_tag(dev, cand, 14, cpu)),
        ("fedavg8", lambda: leg_fedavg8(dev, cand, cpu)),
    ]
    legs = {}
    for name, fn in table:
        if want is not None and name not in want:
            continue
        t0 = time.perf_counter()
        try:
            legs[name] = fn()
        except Exception as ex:                      # noqa: BLE001
            legs[name] = {"error": f"{type(ex).__name__}: {ex}"}
        legs[name]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
        torch.cuda.empty_cache()
    return legs



# ======================================= what is printed ============================================================
# The driver keeps a stdout tail of a few KB and parses the LAST line: that line carries only the contract's fields
# (< 1 800 characters, tests/test_bench_line_cpu.py); every leg is its own short line printed BEFORE it; the full
# records (per-kernel tables, samples, notes) go to bench_legs.json in the working directory.
LINE_LIMIT = 1800
LEG_LIMIT = 1500
ARITH_SHORT = {0: "fp32 operands, products and sums on the fp32 matrix pipe",
               6: "fp32 operands+sums; products as 6 exact bf16 partials on the bf16 matrix pipe",
               9: "fp32 operands+sums; products as 9 exact bf16 partials on the bf16 matrix pipe"}
_ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "frac_of_fp32_mfma_peak", "traffic", "kernel", "launches",
              "avg_launch_ms", "whole_step_frac")
_CPU_KEYS = ("value", "unit", "cores", "threads", "kind")


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _compact_roof(r):
    if not r:
        return None
    out = {k: r[k] for k in _ROOF_KEYS if k in r}
    if "kernel" in out:
        out["kernel"] = _short(out["kernel"], 72)
    return out


def _compact_cpu(c, sample_chars=110):
    if not c:
        return None
    out = {k: c[k] for k in _CPU_KEYS if k in c}
    if "sample" in c:
        out["sample"] = _short(c["sample"], sample_chars)
    return out


def compact_line(out, products=None):
    """The ONE line the driver parses (the contract of the module docstring): contract fields only."""
    cfg = out.get("config") or {}
    c = {"workload": _short(cfg.get("workload", ""), 170)}
    for k in ("parallelism", "rccl_ranks", "ranks_agree_after_allreduce"):
        if k in cfg:
            c[k] = cfg[k]
    if "stream_mode" in cfg:
        c["stream_mode"] = _short(cfg["stream_mode"], 80)
    if "arithmetic" in cfg:
        c["arithmetic"] = _short(ARITH_SHORT.get(products, cfg["arithmetic"]) if out.get("dtype") == "f32"
                                 else cfg["arithmetic"], 80)
    if cfg.get("per_rank"):
        c["per_rank_ms_per_step"] = [r["ms_per_step_incl_allreduce"] for r in cfg["per_rank"]]
        c["per_rank_allreduce_ms"] = [r["allreduce_ms_total"] for r in cfg["per_rank"]]
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = c
    line["roofline"] = _compact_roof(out.get("roofline"))
    line["cpu_baseline"] = _compact_cpu(out.get("cpu_baseline"))
    for k in ("sustained_ms_per_step", "value_on_fp32_mfma_pipe"):
        if k in out:
            line[k] = out[k]
    return _fit(line, LINE_LIMIT, (("cpu_baseline", "sample"), ("config", "stream_mode"), ("config", "arithmetic"),
                                   ("config", "per_rank_allreduce_ms"), ("roofline", "kernel"), ("config", "per_rank_ms_per_step"),
                                   ("value_on_fp32_mfma_pipe",), ("sustained_ms_per_step",), ("cpu_baseline",)))


def _fit(line, limit, optional):
    """json.dumps(line) within `limit` characters: never let prose cost the parse (an assert here threw away minutes of
    measurement).  Optional keys go first, in the order given; then every string is halved until the line fits."""
    s = json.dumps(line)
    for path in optional:
        if len(s) <= limit:
            return s
        d = line
        for k in path[:-1]:
            d = d.get(k) or {}
        d.pop(path[-1], None)
        s = json.dumps(line)

    def shorten(o, n):
        if isinstance(o, dict):
            return {k: shorten(v, n) for k, v in o.items()}
        if isinstance(o, list):
            return [shorten(v, n) for v in o]
        return _short(o, n) if isinstance(o, str) else o
    n = 160
    while len(s) > limit and n >= 8:
        line = shorten(line, n)
        s = json.dumps(line)
        n //= 2
    return s


def compact_leg(name, leg):
    """One short line per leg (printed before the headline line)."""
    if "error" in leg:
        return json.dumps({"leg": name, "error": _short(leg["error"], 400)})
    line = {"leg": name}
    for k in ("value", "unit", "ms_per_step", "ms_per_pass", "ms_per_call", "ms_per_aggregation", "steps", "dtype", "picks",
              "leg_wall_s"):
        if k in leg:
            line[k] = leg[k]
    line["config"] = {"workload": _short((leg.get("config") or {}).get("workload", ""), 200)}
    line["roofline"] = _compact_roof(leg.get("roofline"))
    line["cpu_baseline"] = _compact_cpu(leg.get("cpu_baseline"), 90)
    return _fit(line, LEG_LIMIT, (("cpu_baseline", "sample"), ("roofline", "kernel"), ("cpu_baseline",)))


def emit(out, path="bench_legs.json"):
    """legs (short lines) first, the full record to `path`, the parsed line LAST."""
    try:                                            # the full record first: nothing below may cost it
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    except OSError as ex:
        print(f"[bench] could not write {path}: {ex}", file=sys.stderr)
    for name, leg in (out.get("legs") or {}).items():
        try:
            print(compact_leg(name, leg), flush=True)
        except Exception as ex:                     # a leg's line is a convenience, the headline line is the contract
            print(json.dumps({"leg": name, "error": _short(repr(ex), 200)}), flush=True)
    print(compact_line(out, out.get("_products")), flush=True)


def main():
    args = parse()
    from fedmlp_amd.launch import launched_by_torchrun, spawn_ranks
    if args.gpus > 1 and not launched_by_torchrun():
        # the parent never touches the GPU: its children are the ranks
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    headline = (args.workload == "stage1" and args.model == "Resnet18" and args.batch == 128 and args.classes == 5
                and args.hw == 224 and not args.one_stream)
    args._sustain = headline and world == 1 and args.sustain_s > 0
    out = run_workload(args, rank, world, dev, dist)
    if rank == 0:
        if headline and world == 1 and not args.no_legs:
            best = None
            if out.get("cpu_baseline"):
                best = [out["cpu_baseline"]["threads"]]          # the legs' CPU baselines run at the headline's best thread count
            out["legs"] = run_legs(args, dev, best or thread_candidates(args.cpu_threads)[:1])
            f32 = out["legs"].get("stage1_fp32_mfma_pipe") or {}
            if "value" in f32:
                # the headline metric with the conv GEMMs on the fp32 matrix pipe (fm_config.reserved[2] = 1), for a reader who wants
                # the figure of that arithmetic beside `value` without digging into `legs`
                out["value_on_fp32_mfma_pipe"] = {"value": f32["value"], "ms_per_step": f32["ms_per_step"],
                                                  "roofline_frac_of_fp32_mfma_peak": (f32.get("roofline") or {}).get("frac")}
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
