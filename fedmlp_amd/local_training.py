"""LocalUpdate drop-in: the reference's per-client trainer surface
(utils/local_training.py:26-55 ctor, :628-703 train, :771-825 train_FixMatch,
:904-1256 train_FedMLP) driven through the HIP engine.

Same constructor and method signatures, same return tuples (positions 3-4 are
the reference's junk `_` placeholders -> None), same persistent per-client
state (traindata_idx, idxss, class_num_list, loss_w, iter_num, epoch).  What
changes is where the arithmetic runs: every forward/backward/optimiser step,
the prototype pass, cosine tagging and top-k selection are engine calls
(include/fedmlp_hip.h); this file only orders batches and keeps label masks.

Data: `dataset` follows dataset/all_dataset.py:64-83 -- `.targets` float32
[N,C] and `dataset[i]` -> dict with "image" or "image_aug_1"/"image_aug_2",
"target", "index".  If the dataset exposes `device_views()` -> dict of CUDA
tensors [N,3,H,W] (the HBM-resident cache of SURVEY 8f), batches are gathered
on the GPU instead of being collated on the host.

Batch order: the reference's DataLoader(shuffle=True) draws from the global
torch RNG (:47-48, 1166-1167).  Here `torch.randperm` does (`_order`; the parity
tests replay recorded orders through a subclass, tests/helpers.ReplayLocalUpdate).
"""
import logging

import numpy as np
import torch

from .model import HipNet


def _batches(order, bs):
    return [order[i:i + bs] for i in range(0, len(order), bs)]


def mask_targets(targets, idxs, active_class_list, class_neg_idx):
    """DatasetSplit.__getitem__ label masking (utils/local_training.py:1347-1356) for the
    whole local set at once: for every NON-active class c, the label of samples listed in
    class_neg_idx[c] is zeroed.  Returns (unmasked local labels, masked copy), float32 [N,C]."""
    loc = np.asarray(targets, dtype=np.float32)[[int(i) for i in idxs]]
    y = loc.copy()
    ids = np.asarray([int(i) for i in idxs])
    for c in range(loc.shape[1]):
        if c not in active_class_list:
            y[np.isin(ids, np.asarray(list(class_neg_idx[c]), dtype=np.int64)), c] = 0.0
    return loc, y


def pseudo_targets(local_targets, idxs, active_class_list, negative_class_list, traindata_idx):
    """DatasetSplit_pseudo.__getitem__ (utils/local_training.py:1456-1477) for the whole local
    set: non-active labels zeroed; for the k-th missing class, samples selected as clean or
    noise are supervised (label 1 iff noise), all others get distill_cls = 1.
    Returns (y[N,C], distill_cls[N,C]) float32."""
    ids = np.asarray([int(i) for i in idxs])
    yp = np.array(local_targets, dtype=np.float32, copy=True)
    dist = np.zeros_like(yp)
    for c in range(yp.shape[1]):
        if c not in active_class_list:
            yp[:, c] = 0.0
    for k, cls in enumerate(negative_class_list):
        clean = np.asarray(traindata_idx[2 * k], dtype=np.int64)
        noise = np.asarray(traindata_idx[2 * k + 1], dtype=np.int64)
        in_noise = np.isin(ids, noise)
        in_any = in_noise | np.isin(ids, clean)
        yp[in_noise, cls] = 1.0
        dist[~in_any, cls] = 1.0
    return yp, dist


class LocalUpdate(object):
    def __init__(self, args, client_id, dataset, idxs, class_pos_idx, class_neg_idx,
                 active_class_list=None, student=None, teacher_neg=None, teacher_act=None,
                 dataset_test=None):
        self.args = args
        self.client_id = client_id
        self.dataset = dataset
        self.dataset_test = dataset_test
        self.idxs = [int(i) for i in idxs]
        self.student, self.teacher_neg, self.teacher_act = student, teacher_neg, teacher_act
        self.class_pos_idx, self.class_neg_idx = class_pos_idx, class_neg_idx
        C = args.n_classes
        if active_class_list is None:       # DatasetSplit.__init__ :1338-1341
            import random
            active_class_list = random.sample(list(range(C)), args.annotation_num)
        self.active_class_list = list(active_class_list)
        # DatasetSplit.__getitem__ :1347-1356: zero positives of non-active classes listed in class_neg_idx
        loc, y = mask_targets(dataset.targets, self.idxs, self.active_class_list, class_neg_idx)
        # get_num_of_each_class :1358-1362 (float64 sums of the UNMASKED labels), loss_w :40-42
        self.class_num_list = loc.astype(np.float64).sum(axis=0).tolist()
        n = len(self.idxs)
        self.loss_w = [n / i for i in self.class_num_list]
        self.loss_w_unknown = [1] * C
        self.loss_w_unknown[client_id] = n / self.class_num_list[client_id]
        logging.info(f"---> Client{client_id}, each class num: {self.class_num_list}, total num: {n}")
        self._targets_local = loc
        self._y_masked = y
        self.negative_class_list = [c for c in range(C) if c not in self.active_class_list]
        self.epoch = 0
        self.iter_num = 0
        self.lr = args.base_lr
        self.traindata_idx = []
        self.idxss = []
        self._dev = {}

    # ---- data plumbing -------------------------------------------------------------------
    def _order(self, n):
        """one shuffled pass over the local set: DataLoader(shuffle=True) of :47-48, 1166-1167"""
        return torch.randperm(n).tolist()

    def _views(self, eng):
        if "views" not in self._dev:
            v = None
            if hasattr(self.dataset, "device_views"):
                v = self.dataset.device_views(eng.device)
            self._dev["views"] = v
        return self._dev["views"]

    def _images(self, eng, key, pos):
        ds_idx = [self.idxs[p] for p in pos]
        if hasattr(self.dataset, "device_batch"):
            # HBM-resident uint8 cache + fm_augment: a fresh RandomAffine / HFlip draw per sample and per view,
            # like dataset/all_dataset.py:66-91 applies its transform inside every __getitem__
            return self.dataset.device_batch(eng, key, ds_idx)
        v = self._views(eng)
        if v is not None and key in v:
            sel = torch.as_tensor(ds_idx, device=eng.device, dtype=torch.long)
            return v[key].index_select(0, sel).contiguous()
        x = torch.stack([torch.as_tensor(self.dataset[i][key], dtype=torch.float32) for i in ds_idx])
        return x.to(eng.device).contiguous()

    def _labels_dev(self, eng, arr, name):
        t = self._dev.get(name)
        if t is None or t[0] is not arr:
            self._dev[name] = (arr, torch.from_numpy(np.ascontiguousarray(arr)).to(eng.device))
        return self._dev[name][1]

    def _rows(self, t, pos, eng):
        sel = torch.as_tensor(list(pos), device=eng.device, dtype=torch.long)
        return t.index_select(0, sel).contiguous()

    def _bind(self, net, key):
        assert isinstance(net, HipNet), "net must come from fedmlp_amd.model.build_model"
        sample = self.dataset[self.idxs[0]][key]
        h, w = int(sample.shape[-2]), int(sample.shape[-1])
        return net.bind(h, w, 4 * self.args.batch_size)

    @staticmethod
    def _sd(net):
        # a resident net (one client per GPU) keeps its state in HBM: no per-round D2H copy
        return None if getattr(net, "resident", False) else net.state_dict()

    def _mask(self, classes):
        return [1.0 if c in classes else 0.0 for c in range(self.args.n_classes)]

    # ---- LocalUpdate.train (:628-703) ----------------------------------------------------------
    def train(self, rnd, net, writer1=None):
        a = self.args
        eng = self._bind(net, "image")
        eng.adam_reset(self.lr, (0.9, 0.999), 1e-8, 5e-4)
        y = self._labels_dev(eng, self._y_masked, "y_masked")
        n = len(self.idxs)
        epoch_loss = []
        for _ in range(a.local_ep):
            batches = _batches(self._order(n), a.batch_size)
            losses = torch.zeros(len(batches), device=eng.device)
            for k, pos in enumerate(batches):
                eng.step_bce(self._images(eng, "image", pos), self._rows(y, pos, eng), self.loss_w,
                             a.batch_size, losses[k:k + 1])
                self.iter_num += 1
            self.epoch += 1
            epoch_loss.append(losses.cpu().numpy().astype(np.float64).mean())
        net.mark_trained()
        return self._sd(net), np.array(epoch_loss).mean(), None, None, \
            list(self.negative_class_list), list(self.active_class_list)

    # ---- LocalUpdate.train_FixMatch (:771-825) ---------------------------------------------------
    def train_FixMatch(self, rnd, net):
        a = self.args
        eng = self._bind(net, "image_aug_1")
        eng.adam_reset(self.lr, (0.9, 0.999), 1e-8, 5e-4)
        y = self._labels_dev(eng, self._y_masked, "y_masked")
        n = len(self.idxs)
        act = self._mask(self.active_class_list)
        epoch_loss = []
        for _ in range(a.local_ep):
            batches = _batches(self._order(n), a.batch_size)
            losses = torch.zeros(len(batches), device=eng.device)
            for k, pos in enumerate(batches):
                eng.step_fixmatch(self._images(eng, "image_aug_1", pos), self._images(eng, "image_aug_2", pos),
                                  self._rows(y, pos, eng), self.loss_w, self.loss_w_unknown, act,
                                  a.annotation_num, a.batch_size, losses[k:k + 1])
                self.iter_num += 1
            self.epoch += 1
            epoch_loss.append(losses.cpu().numpy().astype(np.float64).mean())
        net.mark_trained()
        return self._sd(net), np.array(epoch_loss).mean(), None, None, \
            list(self.negative_class_list), list(self.active_class_list)

    # ==== SURVEY 8f rank 4: the other baselines of main.py's --exp switch =========================
    # Their forward/backward/Adam run on the engine through the split step (fm_forward_train /
    # fm_backward_step); only the loss head on the [B,C] logits is host-mirror code (torch ops on the
    # device tensors, gradient by autograd on that [B,C] leaf).
    @staticmethod
    def _head_grad(z, loss_fn):
        zl = z.detach().requires_grad_(True)
        loss = loss_fn(zl)
        (dz,) = torch.autograd.grad(loss, zl)
        return loss.detach(), dz

    # ---- LocalUpdate.train_RSCFed (:705-769) ---------------------------------------------------------
    def train_RSCFed(self, rnd, net):
        a = self.args
        assert isinstance(self.teacher_neg, HipNet), "LocalUpdate(teacher_neg=build_model(...)) is required"
        eng = self._bind(net, "image_aug_1")
        # the EMA teacher lives in the engine's teacher slot for the duration of the call
        self.teacher_neg._pull()
        eng.teacher_swap()
        eng.set_state(self.teacher_neg.flat, self.teacher_neg.counters)
        eng.teacher_swap()
        eng.adam_reset(self.lr, (0.9, 0.999), 1e-8, 5e-4)
        y = self._labels_dev(eng, self._y_masked, "y_masked")
        n = len(self.idxs)
        act, neg = list(self.active_class_list), list(self.negative_class_list)
        pw = torch.tensor(self.loss_w, dtype=torch.float32, device=eng.device)
        F = torch.nn.functional
        epoch_loss = []
        for _ in range(a.local_ep):
            batches = _batches(self._order(n), a.batch_size)
            losses = torch.zeros(len(batches), device=eng.device)
            for k, pos in enumerate(batches):
                yb = self._rows(y, pos, eng)
                _, zt = eng.forward_eval(self._images(eng, "image_aug_2", pos), teacher=True)
                _, z1 = eng.forward_train(self._images(eng, "image_aug_1", pos))

                def head(z):
                    l = F.binary_cross_entropy_with_logits(z, yb, pos_weight=pw, reduction="none")
                    sup = l[:, act].sum() / (a.batch_size * a.annotation_num)
                    return sup + F.mse_loss(torch.sigmoid(z)[:, neg], torch.sigmoid(zt)[:, neg])
                loss, dz = self._head_grad(z1, head)
                eng.backward_step(dz)
                eng.teacher_axpby(1 - 0.001, 0.001)                 # :751-759
                losses[k] = loss
                self.iter_num += 1
            self.epoch += 1
            epoch_loss.append(losses.cpu().numpy().astype(np.float64).mean())
        eng.teacher_swap()
        self.teacher_neg.flat, self.teacher_neg.counters = eng.get_state()
        self.teacher_neg._version += 1
        eng.teacher_swap()
        net.mark_trained()
        return self._sd(net), np.array(epoch_loss).mean(), None, None, neg, act

    # ---- LocalUpdate.train_FedNoRo (:115-234) ----------------------------------------------------------
    def train_FedNoRo(self, id, rnd, net, writer1=None, weight_kd=None, clean_clients=None, noisy_clients=None):
        a = self.args
        if rnd >= a.rounds_FedNoRo_warmup:
            # The reference's post-warm-up branches cannot run: the clean-client branch calls
            # backward() on an unreduced [B,C] loss (:174-176) and the noisy-client branch builds
            # LA_KD without its required class lists (:201); main.py keeps both commented out (:145-148).
            raise NotImplementedError("train_FedNoRo after the warm-up rounds is not runnable in the reference")
        eng = self._bind(net, "image")
        eng.teacher_snapshot()                                      # teacher_net = deepcopy(net), frozen
        eng.adam_reset(self.lr, (0.9, 0.999), 1e-8, 5e-4)
        y = self._labels_dev(eng, self._y_masked, "y_masked")
        n = len(self.idxs)
        act, neg = list(self.active_class_list), list(self.negative_class_list)
        for c in neg:                                               # :136-137
            self.class_num_list[c] = 0
        w_kd = float(weight_kd)
        F = torch.nn.functional
        epoch_loss = []
        for _ in range(a.local_ep):
            batches = _batches(self._order(n), a.batch_size)
            losses = torch.zeros(len(batches), device=eng.device)
            for k, pos in enumerate(batches):
                x, yb = self._images(eng, "image", pos), self._rows(y, pos, eng)
                _, zt = eng.forward_eval(x, teacher=True)
                soft = torch.sigmoid(zt / 0.8)
                _, z = eng.forward_train(x)

                def head(zz):
                    p = torch.sigmoid(zz)
                    B = p.shape[0]
                    bce = F.binary_cross_entropy(p, yb, reduction="none")[:, act].sum() / (B * len(act))
                    kl = F.mse_loss(p, soft, reduction="none")[:, neg].sum() / (B * len(neg))
                    return w_kd * kl + (1 - w_kd) * bce
                loss, dz = self._head_grad(z, head)
                eng.backward_step(dz)
                losses[k] = loss
                self.iter_num += 1
            self.epoch += 1
            epoch_loss.append(losses.cpu().numpy().astype(np.float64).mean())
        net.mark_trained()
        return self._sd(net), np.array(epoch_loss).mean(), None, None, neg, act

    # ---- LocalUpdate.train_CBAFed (:236-342) -------------------------------------------------------------
    def train_CBAFed(self, rnd, net, pt=None, tao=None):
        a = self.args
        eng = self._bind(net, "image")
        eng.adam_reset(self.lr, (0.9, 0.999), 1e-8, 5e-4)
        y = self._labels_dev(eng, self._y_masked, "y_masked")
        n = len(self.idxs)
        act, neg = list(self.active_class_list), list(self.negative_class_list)
        warm = rnd < a.rounds_CBAFed_warmup
        class_num_list = torch.zeros(a.n_classes)
        data_num = 0
        F = torch.nn.functional
        epoch_loss = []
        for _ in range(a.local_ep):
            batches = _batches(self._order(n), a.batch_size)
            losses = torch.zeros(len(batches), device=eng.device)
            for k, pos in enumerate(batches):
                yb = self._rows(y, pos, eng)
                _, z = eng.forward_train(self._images(eng, "image", pos))
                labels, idx_neg = yb, []
                if warm:
                    data_num += len(pos)
                else:
                    prob = torch.sigmoid(z)
                    labels = yb.clone()
                    for i in neg:                                    # :303-316 (one host sync per class, like the reference)
                        hi, lo = prob[:, i] > tao[i], prob[:, i] < (1 - tao[i])
                        noise_num, clean_num = int(hi.sum()), int(lo.sum())
                        labels[:, i] = torch.where(hi, torch.ones_like(labels[:, i]), labels[:, i])
                        pseudo = torch.where(hi | lo)[0]
                        idx_neg.append(pseudo)
                        class_num_list[i] += len(pseudo)
                        data_num += len(pseudo)
                        self.loss_w[i] = 1 if noise_num == 0 else (noise_num + clean_num) / noise_num
                    for i in act:
                        class_num_list[i] += len(pos)
                    data_num += len(pos) * a.annotation_num
                pw = torch.tensor(self.loss_w, dtype=torch.float32, device=eng.device)

                def head(zz):
                    l = F.binary_cross_entropy_with_logits(zz, labels, pos_weight=pw, reduction="none")
                    loss = l[:, act].sum() / (a.batch_size * a.annotation_num)
                    for kk, i in enumerate(neg if not warm else []):
                        if len(idx_neg[kk]) != 0:
                            loss = loss + l[idx_neg[kk], i].sum() / len(idx_neg[kk])
                    return loss
                loss, dz = self._head_grad(z, head)
                eng.backward_step(dz)
                losses[k] = loss
                self.iter_num += 1
            if warm:
                for c in act:
                    class_num_list[c] = data_num
            self.epoch += 1
            epoch_loss.append(losses.cpu().numpy().astype(np.float64).mean())
        net.mark_trained()
        return self._sd(net), np.array(epoch_loss).mean(), None, None, neg, act, class_num_list, data_num

    # ---- prototype + t pass (:971-1002 unguarded, :1208-1250 zero-guarded) ----------------------------
    def _proto_pass(self, eng, negative_list, zero_guard):
        a = self.args
        n = len(self.idxs)
        y = self._labels_dev(eng, self._y_masked, "y_masked")
        act, neg = self._mask(self.active_class_list), self._mask(negative_list)
        eng.proto_reset()
        for pos in _batches(list(range(n)), a.batch_size * 4):
            f, z = eng.forward_eval(self._images(eng, "image_aug_1", pos))
            eng.proto_accumulate(f, z, self._rows(y, pos, eng), act, neg, a.L, a.U)
        t, proto = eng.proto_finalize(zero_guard, n, act)
        return t, torch.from_numpy(proto)

    # ---- cosine tagging + stable top-/bottom-k selection of one missing class (:1052-1112) ---------------
    def _similarity(self, eng, rnd, cls, pool_f, pool_idx, proto_dev):
        """cos(f, P0) - cos(f, P1) of every pool sample, on the device (CosineSimilarityFast :1417-1435)"""
        return eng.cos_tag(pool_f.contiguous(), proto_dev, [cls])[0]

    def _log_similarity(self, rnd, cls, pool_idx, sim_of_pool):
        """hook: the similarities of one class's pool, in pool order (a callable returning the device tensor; tests record it)"""

    def _select_all(self, eng, rnd, classes, f, ds_idx, pools, proto_dev):
        """Every missing class of the round: ONE cosine-tagging launch over all local features, one selection launch pair and one
        device-to-host read (fm_cos_tag + fm_select_topk_rows).  pools[k] = rows of f forming class k's pool, in pool order
        (None = all).  -> [(clean, noise)] dataset indices per class, with _select()'s semantics."""
        if not len(classes):
            return []
        sims = eng.cos_tag(f.contiguous(), proto_dev, list(classes))
        picks = eng.select_topk_rows(sims, pools, self.args.clean_threshold, self.args.noise_threshold)
        out = []
        for k, cls in enumerate(classes):
            rows = range(len(ds_idx)) if pools[k] is None else pools[k]
            pool_idx = [ds_idx[r] for r in rows]
            if pools[k] is None:
                self._log_similarity(rnd, cls, pool_idx, lambda k=k: sims[k])
            else:
                self._log_similarity(rnd, cls, pool_idx, lambda k=k, rows=rows: sims[k].index_select(
                    0, torch.as_tensor(list(rows), device=eng.device, dtype=torch.long)))
            top, bot = picks[k]
            if not len(top):
                # the reference's first branch tests the CLEAN list twice (`len(max_m_indices_list) == 0 and
                # len(max_m_indices_list) == 0`, :1076 / :1099): without a clean pick it keeps no noise pick either
                bot = []
            out.append(([int(pool_idx[j]) for j in top], [int(pool_idx[j]) for j in bot]))
        return out

    def _select(self, eng, rnd, cls, pool_f, pool_idx, proto_dev):
        """-> (clean, noise): dataset indices of the int(clean_threshold * #(sim >= 0)) most similar and the
        int(noise_threshold * #(sim < 0)) least similar pool samples (stable ranks, utils/utils.py:24-35)"""
        if not len(pool_idx):
            return [], []
        sim = self._similarity(eng, rnd, cls, pool_f, pool_idx, proto_dev)
        top, bot = eng.select_topk(sim, self.args.clean_threshold, self.args.noise_threshold)
        if not len(top):
            # the reference's first branch tests the CLEAN list twice (`len(max_m_indices_list) == 0 and
            # len(max_m_indices_list) == 0`, :1076 / :1099): without a clean pick it keeps no noise pick either
            bot = []
        return [int(pool_idx[j]) for j in top], [int(pool_idx[j]) for j in bot]

    # ---- LocalUpdate.train_FedMLP (:904-1256) --------------------------------------------------------
    def train_FedMLP(self, rnd, tao, Prototype, writer1, negetive_class_list, active_class_list_client_i, net):
        a = self.args
        eng = self._bind(net, "image_aug_1")
        n = len(self.idxs)
        if rnd < a.rounds_FedMLP_stage1:                                   # ---- stage 1
            eng.teacher_snapshot()                                         # glob_model = deepcopy(net) :909
            eng.adam_reset(self.lr, (0.9, 0.999), 1e-8, 5e-4)
            y = self._labels_dev(eng, self._y_masked, "y_masked")
            act = self._mask(self.active_class_list)
            epoch_loss = []
            for _ in range(a.local_ep):
                batches = _batches(self._order(n), a.batch_size)
                losses = torch.zeros(len(batches), device=eng.device)
                for k, pos in enumerate(batches):
                    eng.step_stage1(self._images(eng, "image_aug_1", pos), self._images(eng, "image_aug_2", pos),
                                    self._rows(y, pos, eng), act, a.annotation_num, a.batch_size,
                                    losses[k:k + 1])
                    self.iter_num += 1
                self.epoch += 1
                epoch_loss.append(losses.cpu().numpy().astype(np.float64).mean())
            for c in self.negative_class_list:                             # :932 "try noro"
                self.class_num_list[c] = 0
            net.mark_trained()
            ret = (self._sd(net), np.array(epoch_loss).mean(), None, None,
                   list(self.negative_class_list), list(self.active_class_list))
            if rnd == a.rounds_FedMLP_stage1 - 1:                          # first tao and proto :971
                t, proto = self._proto_pass(eng, negetive_class_list, zero_guard=False)
                ret = ret + (t, proto)
            return ret

        # ---------------------------------------------------------------------- stage 2
        first = (rnd == a.rounds_FedMLP_stage1)
        if first:
            self.traindata_idx = []
        # (a) eval-mode feature pass over the shuffled local loader (:1023-1049)
        feat_order = self._order(n)
        D = eng.feature_dim
        f = torch.empty((n, D), device=eng.device)
        row = 0
        for pos in _batches(feat_order, a.batch_size):
            fb, _ = eng.forward_eval(self._images(eng, "image_aug_1", pos))
            f[row:row + len(pos)] = fb
            row += len(pos)
        ds_idx = [self.idxs[p] for p in feat_order]                        # `class_idx`, loader order
        proto_dev = torch.as_tensor(np.asarray(Prototype, dtype=np.float32)).to(eng.device).contiguous()
        # (b)+(c) cosine tagging and stable top-/bottom-k selection per missing class (:1052-1112)
        where = {v: j for j, v in enumerate(ds_idx)}
        pools = [None if first else [where[v] for v in self.idxss[k]]      # find_indices_in_a :901-902
                 for k in range(len(negetive_class_list))]
        picked = self._select_all(eng, rnd, list(negetive_class_list), f, ds_idx, pools, proto_dev)
        for k, cls in enumerate(negetive_class_list):
            clean, noise = picked[k]
            if first:
                self.traindata_idx += [clean, noise]
            else:
                self.traindata_idx[2 * k].extend(clean)
                self.traindata_idx[2 * k + 1].extend(noise)
        for k, cls in enumerate(negetive_class_list):                      # :1117-1120
            self.class_num_list[cls] = len(self.traindata_idx[2 * k + 1])
        # :1150-1156: `loss_w = self.loss_w` aliases the list, so the per-class clean/noise ratio (or
        # 5.0) permanently replaces the pos_weight of every missing class.  Stage 2 itself trains with
        # un-weighted BCE (:1184), but a later train / train_FixMatch / train_RSCFed / train_CBAFed on
        # this LocalUpdate sees the mutated weights, exactly like the reference.
        for k, cls in enumerate(negetive_class_list):
            n_noise = len(self.traindata_idx[2 * k + 1])
            self.loss_w[cls] = len(self.traindata_idx[2 * k]) / n_noise if n_noise != 0 else 5.0
        # (d) training on pseudo-labelled targets (DatasetSplit_pseudo :1456-1477; loop :1164-1196)
        yp, dist = pseudo_targets(self._targets_local, self.idxs, active_class_list_client_i,
                                  negetive_class_list, self.traindata_idx)
        yp_d = torch.from_numpy(yp).to(eng.device)
        dist_d = torch.from_numpy(dist).to(eng.device)
        eng.adam_reset(self.lr, (0.9, 0.999), 1e-8, 5e-4)
        epoch_loss = []
        for _ in range(a.local_ep):
            batches = _batches(self._order(n), a.batch_size)
            losses = torch.zeros(len(batches), device=eng.device)
            for kb, pos in enumerate(batches):
                eng.step_stage2(self._images(eng, "image_aug_1", pos), self._rows(yp_d, pos, eng),
                                self._rows(dist_d, pos, eng), losses[kb:kb + 1])
                self.iter_num += 1
            self.epoch += 1
            epoch_loss.append(losses.cpu().numpy().astype(np.float64).mean())
        net.mark_trained()
        self.idxss = []
        for k in range(len(self.traindata_idx) // 2):                      # :1197-1204
            sel = self.traindata_idx[2 * k] + self.traindata_idx[2 * k + 1]
            self.idxss.append(list(set(self.idxs) - set(sel)))
        # (e) prototype + t pass, zero-count guarded (:1208-1250)
        t, proto = self._proto_pass(eng, negetive_class_list, zero_guard=True)
        return self._sd(net), np.array(epoch_loss).mean(), None, None, \
            negetive_class_list, list(self.active_class_list), t, proto
