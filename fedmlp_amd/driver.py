#!/usr/bin/env python3
"""Thin FL driver: the FedMLP / FedAVG / FedAVG+FixMatch rows of the reference's main.py
(main.py:106-237, with the 'FeMLP' typos normalised) on the HIP engine, one process
per GPU, clients dealt round-robin over the ranks, aggregation as RCCL all-reduces
(fedmlp_amd/fedavg.py).  Data is synthetic and HBM-resident (the reference's datasets and
ImageNet weights are not available offline).

  python -m fedmlp_amd.driver --gpus 8 --exp FedMLP --n_clients 8 --rounds_warmup 4 --rounds_FedMLP_stage1 2
(with --gpus N > 1 and no torchrun environment the driver starts its N ranks itself, fedmlp_amd/launch.py)
"""
import argparse
import json
import os
import time

import numpy as np
import torch


def args_parser():
    p = argparse.ArgumentParser()      # flag names/defaults follow utils/options.py:4-81
    p.add_argument("--exp", default="FedMLP", choices=["FedMLP", "FedAVG", "FedAVG+FixMatch"])
    p.add_argument("--model", default="Resnet18")
    p.add_argument("--seed", type=int, default=1037)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--base_lr", type=float, default=3e-5)
    p.add_argument("--annotation_num", type=int, default=1)
    p.add_argument("--n_clients", type=int, default=8)            # utils/options.py:34
    p.add_argument("--n_classes", type=int, default=8)            # utils/options.py:36
    p.add_argument("--local_ep", type=int, default=1)
    p.add_argument("--rounds_warmup", type=int, default=500)      # utils/options.py:42
    p.add_argument("--rounds_FedMLP_stage1", type=int, default=50)    # utils/options.py:46
    p.add_argument("--U", type=float, default=0.7)
    p.add_argument("--L", type=float, default=0.3)
    p.add_argument("--clean_threshold", type=float, default=0.005)
    p.add_argument("--noise_threshold", type=float, default=0.01)
    p.add_argument("--feature_dim", type=int, default=0, help="0 = the model's own (512 ResNet-18, 1280 Efficient_b0)")
    p.add_argument("--n_local", type=int, default=512, help="samples per client (5000 for ICH)")
    p.add_argument("--hw", type=int, default=224)
    p.add_argument("--gpus", type=int, default=1, help="ranks = GPUs; clients are dealt round-robin over them")
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16"], help="bf16: Efficient_b0 only")
    p.add_argument("--streams", type=int, default=0, help="engine stream mode (fm_config.reserved[1]): 0 default, 1 one stream")
    p.add_argument("--pretrained", type=int, default=1,
                   help="utils/options.py:26: ImageNet checkpoint through build_model (a warning and the from-scratch init "
                        "when no checkpoint file is on this machine)")
    p.add_argument("--pretrained_path", default=None)
    p.add_argument("--init_ckpt", default=None, help="torch.save(state_dict) file to start from (main.py:361-367)")
    p.add_argument("--save_every", type=int, default=0, help="save netglob.state_dict() every N rounds (main.py:237)")
    p.add_argument("--save_dir", default=".")
    p.add_argument("--augment", type=int, default=0,
                   help="1: uint8 HBM cache + per-sample RandomAffine/HFlip/Normalize kernel (dataset/dataset.py:40-53)")
    return p.parse_args()


def AugmentedDeviceDataset(n, C, hw, seed, device):
    """synthetic uint8 images behind fedmlp_amd.augment.AugmentedDataset (uint8 HBM cache + fm_augment)"""
    from fedmlp_amd.augment import AugmentedDataset
    rs = np.random.RandomState(seed)
    imgs = rs.randint(0, 256, size=(n, 3, hw, hw)).astype(np.uint8)
    targets = (rs.uniform(size=(n, C)) < 0.15).astype(np.float32)
    return AugmentedDataset(imgs, targets, train=True, generator=torch.Generator().manual_seed(seed))


class DeviceDataset:
    """dataset/all_dataset.py:64-83 contract on synthetic data kept in HBM."""

    def __init__(self, n, C, hw, seed, device):
        g = torch.Generator(device=device).manual_seed(seed)
        self.targets = (torch.rand((n, C), device=device, generator=g) < 0.15).float().cpu().numpy()
        x1 = torch.randn((n, 3, hw, hw), device=device, generator=g)
        self._v = {"image": x1, "image_aug_1": x1,
                   "image_aug_2": x1 + 0.1 * torch.randn((n, 3, hw, hw), device=device, generator=g)}

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        return {k: v[i] for k, v in self._v.items()} | {"target": self.targets[i].copy(), "index": i}

    def device_views(self, device):
        return self._v


class RoundAccumulator:
    """One round's aggregation (main.py:216-234: FedAvg, FedAvg_tao, FedAvg_proto) split over ranks.  A rank that trains
    several clients folds them first -- state and counters with the weights n_c / sum(n), tao / prototype numerators
    with n_c per class the client misses / annotates -- then every sum over ranks is ONE all-reduce
    (fedmlp_amd/fedavg.py: the library's RCCL communicator, or torch.distributed).  With one rank it reproduces the
    reference's single-process loop; the world-2 gloo test pins both against utils/FedAvg.py's surface."""

    def __init__(self, state_like, n_counters, C, D, n_total):
        self.acc = torch.zeros_like(state_like)
        self.acc_cnt = np.zeros(n_counters, np.float64)
        self.C, self.n_total = C, float(n_total)
        self.t = np.zeros(C); self.tn = np.zeros(C)              # FedAvg_tao numerators / weights of this rank
        self.p = torch.zeros((2 * C, D)); self.pn = np.zeros(C)  # FedAvg_proto numerators / weights

    def add(self, n_c, state, counters, active_class_list, ret):
        w = n_c / self.n_total
        self.acc.add_(state, alpha=w)                            # FedAvg numerator (utils/FedAvg.py:9-13)
        self.acc_cnt += w * np.asarray(counters, np.float64)
        if len(ret) == 8:                                        # (..., t, proto) of a prototype pass
            neg = np.array([0.0 if k in active_class_list else 1.0 for k in range(self.C)])
            act = 1.0 - neg
            self.t += np.asarray(ret[6]) * n_c * neg; self.tn += n_c * neg
            pa = np.repeat(act, 2)
            self.p += torch.where(torch.from_numpy(pa > 0)[:, None], torch.as_tensor(ret[7]) * n_c, torch.zeros(()))
            self.pn += n_c * act

    def reduce(self, eng, dev, with_tao_proto):
        """-> (global num_batches_tracked counters, tao or None, Prototype or None); the averaged state is in the engine"""
        from fedmlp_amd.fedavg import fedavg_allreduce, tao_allreduce, proto_allreduce, state_agreement
        eng.state_tensor().copy_(self.acc)
        eng.counters(np.zeros(len(self.acc_cnt), np.int64))      # the counters are reduced as float64 below
        fedavg_allreduce(eng, 1.0)                               # fm_fedavg_allreduce: ncclAllReduce of the state arena
        agree, worst = state_agreement(eng)                      # every rank starts the next round from the same w_glob
        if not agree:
            raise RuntimeError(f"the ranks' states differ after the FedAvg all-reduce (checksum spread {worst:.3e})")
        cnt = np.trunc(rank_sum(self.acc_cnt, dev) + 1e-9).astype(np.int64)   # utils/FedAvg.py:13 + load_state_dict
        eng.counters(cnt)
        if not with_tao_proto:
            return cnt, None, None
        # FedAvg_tao / FedAvg_proto (utils/FedAvg.py:51-93): this rank enters with its folded clients' means and, per
        # class, the weight sum(n_c) of its clients that miss / annotate the class
        with np.errstate(invalid="ignore", divide="ignore"):
            t_rank = np.where(self.tn > 0, self.t / np.where(self.tn > 0, self.tn, 1.0), 0.0)
            p_rank = self.p.numpy() / np.repeat(np.where(self.pn > 0, self.pn, 1.0), 2)[:, None].astype(np.float32)
        tao = tao_allreduce(t_rank, 1.0, self.tn, device=dev, engine=eng)
        proto = proto_allreduce(p_rank, 1.0, self.pn, device=dev, engine=eng)
        return cnt, tao, proto


def main():
    args = args_parser()
    from fedmlp_amd.launch import launched_by_torchrun, spawn_ranks
    if args.gpus > 1 and not launched_by_torchrun():
        import sys
        sys.exit(spawn_ranks("fedmlp_amd.driver", sys.argv[1:], args.gpus, module=True))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # every RNG the flow draws from (batch orders: torch; class draws: random; numpy) is seeded like main.py seeds its
    # process, offset by the rank: a run is reproducible bit for bit
    import random
    random.seed(args.seed + rank); np.random.seed(args.seed + rank)
    torch.manual_seed(args.seed + rank); torch.cuda.manual_seed_all(args.seed + rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from fedmlp_amd import spec
    from fedmlp_amd.engine import Engine
    from fedmlp_amd.model import ResidentNet, build_model
    from fedmlp_amd.local_training import LocalUpdate
    from fedmlp_amd.fedavg import comm_init, tao_allreduce, proto_allreduce

    C, S1 = args.n_classes, args.rounds_FedMLP_stage1
    args.feature_dim = args.feature_dim or spec.FEATURE_DIM[args.model]
    eng = Engine(args.model, C, args.hw, args.hw, 4 * args.batch_size, device=str(dev), precision=args.precision,
                 streams=args.streams)
    try:
        comm_init(eng)                                    # the library's own RCCL communicator (C ABI)
    except Exception as ex:                               # noqa: BLE001  (then torch.distributed carries the sums)
        print(f"[driver] fm_comm_init failed ({ex}); using torch.distributed all_reduce", flush=True)
    # netglob = build_model(args) (main.py:73): from-scratch init, an ImageNet checkpoint, or --init_ckpt
    host = build_model(args)
    if args.init_ckpt:
        host.load_state_dict(torch.load(args.init_ckpt, map_location="cpu"))
    host._pull()
    eng.set_state(host.flat, host.counters)
    glob = eng.state_tensor().clone()                     # netglob, replicated on every rank
    glob_cnt = eng.counters().copy()                      # its num_batches_tracked counters
    net = ResidentNet(eng)

    mine = [c for c in range(args.n_clients) if c % world == rank]
    n_all = [args.n_local] * args.n_clients
    clients = {}
    for c in mine:                                       # client c annotates class c mod C
        ds = (AugmentedDeviceDataset if args.augment else DeviceDataset)(args.n_local, C, args.hw, args.seed + 1000 * c, dev)
        pos = [np.where(ds.targets[:, k] == 1)[0] for k in range(C)]
        a = argparse.Namespace(**vars(args))
        clients[c] = LocalUpdate(a, c % C, ds, list(range(args.n_local)), pos, pos, active_class_list=[c % C])
    tao, Prototype = [0] * C, None
    log = []
    for rnd in range(args.rounds_warmup):
        t0 = time.perf_counter()
        acc = RoundAccumulator(glob, len(glob_cnt), C, args.feature_dim, float(sum(n_all)))
        losses = []
        for c in mine:
            loc = clients[c]
            eng.state_tensor().copy_(glob)                # net = deepcopy(netglob)  (main.py:181-184)
            eng.counters(glob_cnt)
            if args.exp == "FedAVG":
                ret = loc.train(rnd, net, None)
            elif args.exp == "FedAVG+FixMatch":
                ret = loc.train_FixMatch(rnd, net)
            elif rnd < S1 - 1:
                ret = loc.train_FedMLP(rnd, tao, Prototype, None, None, None, net=net)
            else:
                ret = loc.train_FedMLP(rnd, tao, Prototype, None, loc.negative_class_list,
                                       loc.active_class_list, net=net)
            losses.append(float(ret[1]))
            acc.add(n_all[c], eng.state_tensor(), eng.counters(), loc.active_class_list, ret)
        # ---- aggregation (main.py:216-234): every sum over clients is an RCCL all-reduce in the library
        glob_cnt, tao_new, proto_new = acc.reduce(eng, dev, with_tao_proto=(args.exp == "FedMLP" and rnd >= S1 - 1))
        glob.copy_(eng.state_tensor())
        if tao_new is not None:
            tao, Prototype = tao_new, proto_new
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec = {"round": rnd, "sec": round(dt, 3), "mean_loss": float(np.mean(losses)) if losses else None,
               "samples_per_sec_per_gpu": round(len(mine) * args.n_local / dt, 1)}
        if rank == 0:
            print(json.dumps(rec), flush=True)
            if args.save_every and (rnd + 1) % args.save_every == 0:     # torch.save(netglob.state_dict(), ...) main.py:237
                torch.save(net.state_dict(), os.path.join(args.save_dir, f"model_{rnd}.pth"))
        log.append(rec)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
    return log


def rank_sum(x, dev):
    """float64 sum of a small host vector over the ranks"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(x, np.float64)
    t = torch.from_numpy(np.asarray(x, np.float64)).to(dev)
    dist.all_reduce(t)
    return t.cpu().numpy()


if __name__ == "__main__":
    main()
