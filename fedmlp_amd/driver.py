#!/usr/bin/env python3
"""Thin FL driver: the FedMLP / FedAVG / FedAVG+FixMatch rows of the reference's main.py
(main.py:106-237, with the 'FeMLP' typos normalised) on the HIP engine, one process
per GPU, clients dealt round-robin over the ranks, aggregation as RCCL all-reduces
(fedmlp_amd/fedavg.py).  Data is synthetic and HBM-resident (the reference's datasets and
ImageNet weights are not available offline).

  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m fedmlp_amd.driver \\
      --exp FedMLP --n_clients 8 --rounds_warmup 4 --rounds_FedMLP_stage1 2
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
    p.add_argument("--n_clients", type=int, default=5)
    p.add_argument("--n_classes", type=int, default=5)
    p.add_argument("--local_ep", type=int, default=1)
    p.add_argument("--rounds_warmup", type=int, default=4)
    p.add_argument("--rounds_FedMLP_stage1", type=int, default=2)
    p.add_argument("--U", type=float, default=0.7)
    p.add_argument("--L", type=float, default=0.3)
    p.add_argument("--clean_threshold", type=float, default=0.005)
    p.add_argument("--noise_threshold", type=float, default=0.01)
    p.add_argument("--feature_dim", type=int, default=0, help="0 = the model's own (512 ResNet-18, 1280 Efficient_b0)")
    p.add_argument("--n_local", type=int, default=512, help="samples per client (5000 for ICH)")
    p.add_argument("--hw", type=int, default=224)
    return p.parse_args()


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


def main():
    args = args_parser()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from fedmlp_amd import spec
    from fedmlp_amd.engine import Engine
    from fedmlp_amd.model import ResidentNet
    from fedmlp_amd.local_training import LocalUpdate
    from fedmlp_amd.fedavg import allreduce_weighted_, tao_allreduce, proto_allreduce

    C, S1 = args.n_classes, args.rounds_FedMLP_stage1
    args.feature_dim = args.feature_dim or spec.FEATURE_DIM[args.model]
    eng = Engine(args.model, C, args.hw, args.hw, 4 * args.batch_size, device=str(dev))
    flat, cnt = spec.init_state(args.model, C, args.seed)
    eng.set_state(flat, cnt)
    glob = eng.state_tensor().clone()                     # netglob, replicated on every rank
    net = ResidentNet(eng)

    mine = [c for c in range(args.n_clients) if c % world == rank]
    n_all = [args.n_local] * args.n_clients
    clients = {}
    for c in mine:                                       # client c annotates class c mod C
        ds = DeviceDataset(args.n_local, C, args.hw, args.seed + 1000 * c, dev)
        pos = [np.where(ds.targets[:, k] == 1)[0] for k in range(C)]
        a = argparse.Namespace(**vars(args))
        clients[c] = LocalUpdate(a, c % C, ds, list(range(args.n_local)), pos, pos, active_class_list=[c % C])
    tao, Prototype = [0] * C, None
    log = []
    for rnd in range(args.rounds_warmup):
        t0 = time.perf_counter()
        acc = torch.zeros_like(glob)
        t_num = np.zeros(C); t_den = np.zeros(C)
        p_num = torch.zeros((2 * C, args.feature_dim)); p_den = np.zeros(2 * C)
        losses = []
        for c in mine:
            loc = clients[c]
            eng.state_tensor().copy_(glob)                # net = deepcopy(netglob)  (main.py:181-184)
            w = n_all[c] / float(sum(n_all))
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
            acc.add_(eng.state_tensor(), alpha=w)         # FedAvg numerator (utils/FedAvg.py:9-13)
            if len(ret) == 8:                             # FedAvg_tao / FedAvg_proto numerators
                neg = np.array([0.0 if k in loc.active_class_list else 1.0 for k in range(C)])
                act = 1.0 - neg
                t_num += ret[6] * n_all[c] * neg; t_den += n_all[c] * neg
                pa = np.repeat(act, 2)
                p_num += torch.where(torch.from_numpy(pa > 0)[:, None], ret[7] * n_all[c], torch.zeros(()))
                p_den += n_all[c] * pa
        if dist is not None:
            dist.all_reduce(acc)
        glob.copy_(acc)
        if args.exp == "FedMLP" and rnd >= S1 - 1:
            buf = torch.from_numpy(np.concatenate([t_num, t_den])).to(dev)
            pn, pd = p_num.to(dev), torch.from_numpy(p_den).to(dev)
            if dist is not None:
                dist.all_reduce(buf); dist.all_reduce(pn); dist.all_reduce(pd)
            buf = buf.cpu().numpy()
            tao = np.where(buf[C:] == 0, 1.0, buf[:C] / np.where(buf[C:] == 0, 1.0, buf[C:]))
            with np.errstate(invalid="ignore", divide="ignore"):
                Prototype = torch.from_numpy(pn.cpu().numpy() / pd.cpu().numpy()[:, None].astype(np.float32))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec = {"round": rnd, "sec": round(dt, 3), "mean_loss": float(np.mean(losses)) if losses else None,
               "samples_per_sec_per_gpu": round(len(mine) * args.n_local / dt, 1)}
        if rank == 0:
            print(json.dumps(rec), flush=True)
        log.append(rec)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
    return log


if __name__ == "__main__":
    main()
