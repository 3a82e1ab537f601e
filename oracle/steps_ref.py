"""Oracle restatement of the FedMLP per-client trainer arithmetic (torch CPU fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference lines it follows (paths relative to /root/reference).  Pinned
against goldens captured from the imported reference: tests/golden/.
"""
import copy

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# loss heads
# --------------------------------------------------------------------------
def bce_on_probs(p, y):
    """LogitAdjust_Multilabel.forward (utils/FedNoRo.py:15-22): despite the name
    it is element-wise BCE on probabilities (log clamped at -100 by torch)."""
    return F.binary_cross_entropy(p, y, reduction="none")


def loss_train(logits, y, pos_weight, bs_norm, n_classes):
    """LocalUpdate.train loss (utils/local_training.py:642, 664-665):
    BCEWithLogits(pos_weight) summed over ALL classes / (args.batch_size * C)."""
    l = F.binary_cross_entropy_with_logits(
        logits, y, pos_weight=torch.as_tensor(pos_weight, dtype=torch.float32), reduction="none")
    return l.sum() / (bs_norm * n_classes)


def loss_stage1(z1, z2, g1, g2, y, act, neg, bs_norm, annotation_num):
    """train_FedMLP stage-1 loss (utils/local_training.py:937-963).
    z1,z2 student logits of the two views; g1,g2 teacher logits (no grad).
    Returns (loss, loss_sup, loss_dis).  The 0.0*loss_unsup term (:961-963) is dead."""
    p1, p2 = torch.sigmoid(z1), torch.sigmoid(z2)
    q1, q2 = torch.sigmoid(g1.detach()), torch.sigmoid(g2.detach())
    dis = ((p1 - q1) ** 2 + (p2 - q2) ** 2) / 2.0
    sup = (bce_on_probs(p1, y) + bce_on_probs(p2, y)) / 2.0
    loss_sup = sup[:, act].sum() / (bs_norm * annotation_num)
    loss_dis = dis[:, neg].sum() / (bs_norm * len(neg))
    return loss_sup + loss_dis, loss_sup, loss_dis


def loss_stage2(z, y, distill_cls):
    """train_FedMLP stage-2 loss (utils/local_training.py:1172-1188):
    masked BCE on probabilities, sup_cls = 1 - distill_cls, / sum(sup_cls)."""
    sup_cls = 1.0 - (distill_cls != 0).float()
    return (bce_on_probs(torch.sigmoid(z), y) * sup_cls).sum() / sup_cls.sum()


def fixmatch_mask(z_weak, neg, bs_norm):
    """Confident-row list of train_FixMatch (utils/local_training.py:799-803):
    rows r < args.batch_size whose EVERY missing-class prob is >0.8 or <0.2.
    The reference seeds the set with range(args.batch_size), so rows are limited
    to the actual batch by the where() intersections.  Returns a sorted list."""
    p = torch.sigmoid(z_weak)
    keep = set(range(bs_norm))
    for c in neg:
        conf = set(torch.where((p[:, c] > 0.8) | (p[:, c] < 0.2))[0].tolist())
        keep &= conf
    return sorted(keep)


def loss_fixmatch(z_weak, z_strong, y, pos_w, pos_w_unknown, act, neg, bs_norm,
                  annotation_num, n_classes):
    """train_FixMatch loss (utils/local_training.py:783-815)."""
    pw = torch.as_tensor(pos_w, dtype=torch.float32)
    pwu = torch.as_tensor(pos_w_unknown, dtype=torch.float32)
    sup = F.binary_cross_entropy_with_logits(z_weak, y, pos_weight=pw, reduction="none")
    loss_sup = sup[:, act].sum() / (bs_norm * annotation_num)
    idx = fixmatch_mask(z_weak, neg, bs_norm)
    if len(idx) == 0 or len(neg) == 0:
        return loss_sup
    hard = (torch.sigmoid(z_weak) > 0.5).float().detach()
    uns = F.binary_cross_entropy_with_logits(z_strong, hard, pos_weight=pwu, reduction="none")
    loss_unsup = uns[idx, :][:, neg].sum() / (len(idx) * (n_classes - annotation_num))
    return loss_sup + loss_unsup


# --------------------------------------------------------------------------
# label masking (DatasetSplit / DatasetSplit_pseudo)
# --------------------------------------------------------------------------
# ---- loss heads of the other baselines ------------------------------------
def loss_rscfed(z_student, z_teacher, y, pos_weight, act, neg, bs_norm, annotation_num):
    """train_RSCFed loss (utils/local_training.py:719, 740-743): BCEWithLogits(pos_weight) on the
    active classes / (args.batch_size*annotation_num) + nn.MSELoss() (mean over B*|neg| elements)
    between student(view 1) and EMA-teacher(view 2) probabilities on the missing classes."""
    l = F.binary_cross_entropy_with_logits(
        z_student, y, pos_weight=torch.as_tensor(pos_weight, dtype=torch.float32, device=z_student.device),
        reduction="none")
    sup = l[:, act].sum() / (bs_norm * annotation_num)
    unsup = F.mse_loss(torch.sigmoid(z_student)[:, neg], torch.sigmoid(z_teacher.detach())[:, neg])
    return sup + unsup


def loss_la_kd(z_student, z_teacher, y, act, neg, w_kd):
    """train_FedNoRo warm-up loss = LA_KD.forward (utils/FedNoRo.py:35-38) on sigmoid(logits) with
    soft_label = sigmoid(teacher/0.8) (utils/local_training.py:143-151): both terms are divided by
    the ACTUAL batch length len(x)."""
    p = torch.sigmoid(z_student)
    soft = torch.sigmoid(z_teacher.detach() / 0.8)
    B = p.shape[0]
    bce = bce_on_probs(p, y)[:, act].sum() / (B * len(act))
    kl = F.mse_loss(p, soft, reduction="none")[:, neg].sum() / (B * len(neg))
    return w_kd * kl + (1 - w_kd) * bce


def loss_cbafed_stage1(z, y, pos_weight, act, bs_norm, annotation_num):
    """train_CBAFed warm-up loss (utils/local_training.py:247-248, 266-269)."""
    l = F.binary_cross_entropy_with_logits(
        z, y, pos_weight=torch.as_tensor(pos_weight, dtype=torch.float32, device=z.device), reduction="none")
    return l[:, act].sum() / (bs_norm * annotation_num)


def cbafed_stage2_targets(prob, y, neg, tao, loss_w):
    """Per-batch pseudo-labelling of train_CBAFed stage 2 (utils/local_training.py:303-316).
    Returns (labels, idx_neg [per negative class: LongTensor of rows], loss_w (updated copy),
    per-class pseudo counts)."""
    labels = y.clone()
    loss_w = list(loss_w)
    idx_neg, counts = [], []
    for i in neg:
        hi = prob[:, i] > tao[i]
        lo = prob[:, i] < (1 - tao[i])
        noise_num, clean_num = int(hi.sum()), int(lo.sum())
        labels[:, i] = torch.where(hi, torch.ones_like(labels[:, i]), labels[:, i])
        pseudo = torch.where(hi | lo)[0]
        idx_neg.append(pseudo)
        counts.append(len(pseudo))
        loss_w[i] = 1 if noise_num == 0 else (noise_num + clean_num) / noise_num
    return labels, idx_neg, loss_w, counts


def loss_cbafed_stage2(z, labels, idx_neg, loss_w, act, neg, bs_norm, annotation_num):
    """train_CBAFed stage-2 loss (utils/local_training.py:321-331)."""
    l = F.binary_cross_entropy_with_logits(
        z, labels, pos_weight=torch.as_tensor(loss_w, dtype=torch.float32, device=z.device), reduction="none")
    loss = l[:, act].sum() / (bs_norm * annotation_num)
    for k, i in enumerate(neg):
        if len(idx_neg[k]) != 0:
            loss = loss + l[idx_neg[k], i].sum() / len(idx_neg[k])
    return loss


def consistency_weight(rnd, begin, end):
    """get_current_consistency_weight -> sigmoid_rampup (utils/FedNoRo.py:72-81):
    exp(-5 (1 - (clip(rnd, begin, end)-begin)/(end-begin))^2), i.e. exp(-5) before `begin`, 1 after `end`."""
    cur = float(np.clip(rnd, begin, end))
    phase = 1.0 - (cur - begin) / (end - begin)
    return float(np.exp(-5.0 * phase * phase))


def class_counts(targets, idxs):
    """DatasetSplit.get_num_of_each_class (utils/local_training.py:1358-1362):
    per-class sum of the UNMASKED targets over the client's indices (float64)."""
    s = np.zeros(targets.shape[1], dtype=np.float64)
    for i in idxs:
        s += targets[i]
    return s.tolist()


def mask_targets(targets, idxs, active, class_neg_idx):
    """DatasetSplit.__getitem__ label masking (utils/local_training.py:1347-1356):
    for every non-active class c, zero the label of samples listed in
    class_neg_idx[c].  Returns a fresh [len(idxs), C] float32 array."""
    neg_sets = [set(int(v) for v in lst) for lst in class_neg_idx]
    out = np.array([targets[i] for i in idxs], dtype=np.float32).reshape(len(idxs), -1)
    for r, i in enumerate(idxs):
        for c in range(out.shape[1]):
            if c not in active and int(i) in neg_sets[c]:
                out[r, c] = 0.0
    return out


def pseudo_targets(targets, idxs, active, negative, traindata_idx):
    """DatasetSplit_pseudo.__getitem__ (utils/local_training.py:1456-1477).
    Returns (y[N,C], distill_cls[N,C]).  Non-active labels are zeroed; for the
    k-th missing class, samples in clean_k+noise_k get label 1 iff in noise_k,
    all other samples get distill_cls=1 for that class."""
    n, C = len(idxs), targets.shape[1]
    y = np.array([targets[i] for i in idxs], dtype=np.float32).reshape(n, C)
    dist = np.zeros((n, C), dtype=np.float32)
    for c in range(C):
        if c not in active:
            y[:, c] = 0.0
    for k in range(len(traindata_idx) // 2):
        clean = set(int(v) for v in traindata_idx[2 * k])
        noise = set(int(v) for v in traindata_idx[2 * k + 1])
        cls = negative[k]
        for r, i in enumerate(idxs):
            i = int(i)
            if i in clean or i in noise:
                if i in noise:
                    y[r, cls] = 1.0
            else:
                dist[r, cls] = 1.0
    return y, dist


# --------------------------------------------------------------------------
# cosine tagging + selection
# --------------------------------------------------------------------------
def cosine_sim(x1, x2):
    """CosineSimilarityFast.forward (utils/local_training.py:1421-1435):
    x1[N,D], x2[1,D] -> x1 x2^T / (|x1| |x2|), squeezed to [N]."""
    x2t = x2.t()
    num = x1.mm(x2t)
    den = x1.norm(dim=1).unsqueeze(0).t().mm(x2t.norm(dim=0).unsqueeze(0))
    return torch.squeeze(num.mul(1 / den), dim=1)


def cosine_diff(f, p0, p1):
    """sim = cos(f, proto_0) - cos(f, proto_1) (utils/local_training.py:1052-1057)."""
    return cosine_sim(f, p0.unsqueeze(0)) - cosine_sim(f, p1.unsqueeze(0))


def max_m_indices(lst, n):
    """utils/utils.py:24-28: positions of the n largest values, stable (first
    position wins ties), via a descending stable sort."""
    order = sorted(range(len(lst)), key=lambda i: lst[i], reverse=True)
    return order[:n]


def min_n_indices(lst, n):
    """utils/utils.py:31-35: positions of the n smallest values, stable."""
    order = sorted(range(len(lst)), key=lambda i: lst[i])
    return order[:n]


def select_for_class(sim, pool_idx, clean_thr, noise_thr):
    """Selection for one missing class (utils/local_training.py:1061-1087):
    'clean' = sim >= 0, 'noise' = sim < 0 (NaN -> neither);
    k_clean = int(clean_thr*|clean|), k_noise = int(noise_thr*|noise|);
    take top-k_clean / bottom-k_noise of sim over the WHOLE pool; map to dataset
    indices.  Returns (clean_dataset_idx, noise_dataset_idx)."""
    sim = [float(s) for s in sim]
    arr = np.array(sim)
    n_clean = int(np.sum(arr >= 0))
    n_noise = int(np.sum(arr < 0))
    k_clean = int(1 * clean_thr * n_clean)
    k_noise = int(1 * noise_thr * n_noise)
    top = max_m_indices(sim, k_clean)
    bot = min_n_indices(sim, k_noise)
    if len(top) == 0:
        # :1076 / :1099 test `len(max_m_indices_list) == 0 and len(max_m_indices_list) == 0` (the clean list twice):
        # a class without a clean pick keeps no noise pick either
        bot = []
    return [int(pool_idx[j]) for j in top], [int(pool_idx[j]) for j in bot]


# --------------------------------------------------------------------------
# prototype pass
# --------------------------------------------------------------------------
def prototype_pass(batches, n_classes, active, negative, L, U, n_local, zero_guard):
    """Prototype + t pass (utils/local_training.py:971-1002 unguarded,
    :1208-1250 zero-guarded).  `batches` yields (feature[b,D], logits[b,C],
    labels[b,C]) from the eval-mode net.  Returns (t float64[C], proto[2C,D])."""
    proto, cnt = None, [0] * (2 * n_classes)
    t = np.array([0] * n_classes)
    for feature, logits, labels in batches:
        if proto is None:
            proto = torch.zeros((2 * n_classes, feature.shape[1]))
        probs = torch.sigmoid(logits)
        for c in active:
            i0 = torch.where(labels[:, c] == 0)[0]
            i1 = torch.where(labels[:, c] == 1)[0]
            cnt[2 * c] += len(i0)
            cnt[2 * c + 1] += len(i1)
            proto[2 * c] = feature[i0, :].sum(0) + proto[2 * c]
            proto[2 * c + 1] = feature[i1, :].sum(0) + proto[2 * c + 1]
        for c in negative:
            t[c] += torch.sum((probs[:, c] < L) | (probs[:, c] > U)).item()
    for c in active:
        for r in (2 * c, 2 * c + 1):
            if zero_guard and cnt[r] == 0:
                continue
            proto[r] = proto[r] / cnt[r]
    return t / n_local, proto


# --------------------------------------------------------------------------
# aggregation (utils/FedAvg.py)
# --------------------------------------------------------------------------
def fedavg(w, dict_len):
    """FedAvg (utils/FedAvg.py:7-14): left-to-right sample-count-weighted mean of
    EVERY state_dict entry (BN running stats and the int64 counter included; the
    counter becomes float by true division)."""
    out = copy.deepcopy(w[0])
    for k in out.keys():
        acc = out[k] * dict_len[0]
        for i in range(1, len(w)):
            acc = acc + w[i][k] * dict_len[i]
        out[k] = acc / sum(dict_len)
    return out


def fedavg_tao(t, weight, class_active_client_list):
    """FedAvg_tao, class-masked branch (utils/FedAvg.py:60-70)."""
    C = len(t[0])
    out = np.zeros(C, dtype=np.float64)
    for cls, clients in enumerate(class_active_client_list):
        if len(clients) == 0:
            out[cls] = 1.0
            continue
        wsum = 0.0
        for i in range(len(t)):
            if i in clients:
                out[cls] += t[i][cls] * float(weight[i])
                wsum += float(weight[i])
        out[cls] = out[cls] / wsum
    return out


def fedavg_proto(protos, weight, class_active_client_list):
    """FedAvg_proto (utils/FedAvg.py:72-93): per class, weighted mean of the two
    prototype rows over that class's active clients; no client -> 0/0 = NaN."""
    out = torch.zeros((len(protos[0]), len(protos[0][0])))
    for cls, clients in enumerate(class_active_client_list):
        a0 = torch.zeros_like(protos[0][0])
        a1 = torch.zeros_like(protos[0][0])
        for cid in clients:
            a0 = protos[cid][2 * cls] * weight[cid] + a0
            a1 = protos[cid][2 * cls + 1] * weight[cid] + a1
        den = np.sum(np.array(weight)[clients])
        out[2 * cls] = a0 / den
        out[2 * cls + 1] = a1 / den
    return out


# --------------------------------------------------------------------------
# trainer flows with explicit batch orders
# --------------------------------------------------------------------------
def _adam(net, lr):
    # utils/local_training.py:636-637, 912-913, 1149-1150: fresh torch Adam every
    # round, coupled L2 weight decay 5e-4.
    return torch.optim.Adam(net.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=5e-4)


def _batches(order, bs):
    return [order[i:i + bs] for i in range(0, len(order), bs)]


class RefClient:
    """Functional restatement of LocalUpdate for the FedAVG / FedMLP / FixMatch
    flows.  `data` is a dict of torch tensors: images under "image" and/or
    "image_aug_1"/"image_aug_2" ([Ntot,3,H,W]) and numpy "targets" [Ntot,C].
    Batch orders are passed explicitly (positions into self.idxs) because the
    reference draws them from the global RNG (Q11 in SURVEY.md)."""

    def __init__(self, args, client_id, data, idxs, class_neg_idx, active_class_list):
        self.args, self.client_id, self.data = args, client_id, data
        self.idxs = [int(i) for i in idxs]
        self.active = list(active_class_list)
        self.negative = [c for c in range(args.n_classes) if c not in self.active]
        self.targets = data["targets"]
        # utils/local_training.py:38-42
        self.class_num_list = class_counts(self.targets, self.idxs)
        n = len(self.idxs)
        self.loss_w = [n / c for c in self.class_num_list]
        self.loss_w_unknown = [1] * args.n_classes
        self.loss_w_unknown[client_id] = n / self.class_num_list[client_id]
        self.y_masked = torch.from_numpy(
            mask_targets(self.targets, self.idxs, self.active, class_neg_idx))
        self.traindata_idx = []
        self.idxss = []

    def _img(self, key, pos):
        return self.data[key][[self.idxs[p] for p in pos]]

    # -- LocalUpdate.train (utils/local_training.py:628-703) -----------------
    def train(self, net, order):
        a = self.args
        net.train()
        opt = _adam(net, a.base_lr)
        losses = []
        for pos in _batches(order, a.batch_size):
            _, z = net(self._img("image", pos))
            loss = loss_train(z, self.y_masked[pos], self.loss_w, a.batch_size, a.n_classes)
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(loss.item())
        return net.state_dict(), float(np.mean(losses)), losses

    # -- LocalUpdate.train_RSCFed (utils/local_training.py:705-769) ------------
    def train_rscfed(self, net, teacher, order):
        """`teacher` (self.teacher_neg in the reference) is updated in place by the per-step EMA
        over every state_dict entry (int64 counters: float result truncated on load)."""
        a = self.args
        student = copy.deepcopy(net)
        teacher.eval(); student.train()
        opt = _adam(student, a.base_lr)
        losses = []
        for pos in _batches(order, a.batch_size):
            _, z1 = student(self._img("image_aug_1", pos))
            with torch.no_grad():
                _, zt = teacher(self._img("image_aug_2", pos))
            loss = loss_rscfed(z1, zt, self.y_masked[pos], self.loss_w, self.active, self.negative,
                               a.batch_size, a.annotation_num)
            opt.zero_grad(); loss.backward(); opt.step()
            sd1, sd2 = teacher.state_dict(), student.state_dict()
            teacher.load_state_dict({k: (1 - 0.001) * sd1[k] + 0.001 * sd2[k] for k in sd1})
            losses.append(loss.item())
        return student.state_dict(), float(np.mean(losses)), losses

    # -- LocalUpdate.train_FedNoRo, warm-up branch (utils/local_training.py:118-155) --
    def train_fednoro(self, net, order, weight_kd):
        a = self.args
        student, teacher = copy.deepcopy(net), copy.deepcopy(net)
        student.train(); teacher.eval()
        opt = _adam(student, a.base_lr)
        for i in self.negative:                 # :136-137 zeroes the missing classes' counts
            self.class_num_list[i] = 0
        losses = []
        for pos in _batches(order, a.batch_size):
            x = self._img("image", pos)
            _, z = student(x)
            with torch.no_grad():
                _, zt = teacher(x)
            loss = loss_la_kd(z, zt, self.y_masked[pos], self.active, self.negative, weight_kd)
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(loss.item())
        return student.state_dict(), float(np.mean(losses)), losses

    # -- LocalUpdate.train_CBAFed (utils/local_training.py:236-342) ---------------
    def train_cbafed(self, net, order, tao=None):
        """tao None = warm-up stage.  Returns (state_dict, mean loss, losses, class_num_list, data_num)."""
        a = self.args
        net.train()
        opt = _adam(net, a.base_lr)
        class_num = torch.zeros(a.n_classes)
        data_num = 0
        losses = []
        for pos in _batches(order, a.batch_size):
            y = self.y_masked[pos]
            _, z = net(self._img("image", pos))
            if tao is None:
                data_num += len(pos)
                loss = loss_cbafed_stage1(z, y, self.loss_w, self.active, a.batch_size, a.annotation_num)
            else:
                labels, idx_neg, self.loss_w, counts = cbafed_stage2_targets(
                    torch.sigmoid(z.detach()), y, self.negative, tao, self.loss_w)
                for k, i in enumerate(self.negative):
                    class_num[i] += counts[k]
                    data_num += counts[k]
                for i in self.active:
                    class_num[i] += len(pos)
                data_num += len(pos) * a.annotation_num
                loss = loss_cbafed_stage2(z, labels, idx_neg, self.loss_w, self.active, self.negative,
                                          a.batch_size, a.annotation_num)
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(loss.item())
        if tao is None:
            for i in self.active:
                class_num[i] = data_num
        return net.state_dict(), float(np.mean(losses)), losses, class_num.tolist(), int(data_num)

    # -- LocalUpdate.train_FixMatch (utils/local_training.py:771-825) --------
    def train_fixmatch(self, net, order):
        a = self.args
        net.train()
        opt = _adam(net, a.base_lr)
        losses = []
        for pos in _batches(order, a.batch_size):
            _, zw = net(self._img("image_aug_1", pos))
            _, zs = net(self._img("image_aug_2", pos))
            loss = loss_fixmatch(zw, zs, self.y_masked[pos], self.loss_w, self.loss_w_unknown,
                                 self.active, self.negative, a.batch_size, a.annotation_num,
                                 a.n_classes)
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(loss.item())
        return net.state_dict(), float(np.mean(losses)), losses

    # -- train_FedMLP stage 1 (utils/local_training.py:907-1004) -------------
    def stage1(self, net, order, with_proto, negative_param=None):
        a = self.args
        glob = copy.deepcopy(net).eval()
        net.train()
        opt = _adam(net, a.base_lr)
        losses = []
        for pos in _batches(order, a.batch_size):
            x1, x2 = self._img("image_aug_1", pos), self._img("image_aug_2", pos)
            _, z1 = net(x1)
            _, z2 = net(x2)
            with torch.no_grad():
                _, g1 = glob(x1)
                _, g2 = glob(x2)
            loss, _, _ = loss_stage1(z1, z2, g1, g2, self.y_masked[pos], self.active,
                                     self.negative, a.batch_size, a.annotation_num)
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(loss.item())
        for c in self.negative:          # :932 "try noro"
            self.class_num_list[c] = 0
        out = [net.state_dict(), float(np.mean(losses)), losses]
        if with_proto:
            t, proto = self._proto_pass(net, negative_param, zero_guard=False)
            out += [t, proto]
        return out

    def _proto_pass(self, net, negative_param, zero_guard):
        a = self.args
        net.eval()
        n = len(self.idxs)

        def gen():
            with torch.no_grad():
                for pos in _batches(list(range(n)), a.batch_size * 4):
                    f, z = net(self._img("image_aug_1", pos))
                    yield f, z, self.y_masked[pos]
        return prototype_pass(gen(), a.n_classes, self.active, negative_param, a.L, a.U, n,
                              zero_guard)

    # -- train_FedMLP stage 2 (utils/local_training.py:1006-1256) ------------
    def stage2(self, rnd, net, prototype, negative_param, feat_order, train_order):
        a = self.args
        n = len(self.idxs)
        net.eval()
        feats, pos_all = [], []
        with torch.no_grad():
            for pos in _batches(feat_order, a.batch_size):
                f, _ = net(self._img("image_aug_1", pos))
                feats.append(f); pos_all += list(pos)
        f = torch.cat(feats, 0)
        ds_idx = [self.idxs[p] for p in pos_all]         # `class_idx` in loader order
        first = (rnd == a.rounds_FedMLP_stage1)
        if first:
            self.traindata_idx = []
        sims = []
        for k, cls in enumerate(negative_param):
            if first:
                pool_f, pool_idx = f, ds_idx
            else:
                where = {v: j for j, v in enumerate(ds_idx)}
                rows = [where[v] for v in self.idxss[k]]   # find_indices_in_a :901-902
                pool_f, pool_idx = f[rows], [ds_idx[r] for r in rows]
            sim = cosine_diff(pool_f, prototype[2 * cls], prototype[2 * cls + 1]).tolist()
            sims.append(sim)
            clean, noise = select_for_class(sim, pool_idx, a.clean_threshold, a.noise_threshold)
            if first:
                self.traindata_idx += [clean, noise]
            else:
                self.traindata_idx[2 * k].extend(clean)
                self.traindata_idx[2 * k + 1].extend(noise)
        for k, cls in enumerate(negative_param):
            self.class_num_list[cls] = len(self.traindata_idx[2 * k + 1])
        for k, cls in enumerate(negative_param):         # :1150-1156 (`loss_w = self.loss_w` aliases the list)
            n_noise = len(self.traindata_idx[2 * k + 1])
            self.loss_w[cls] = len(self.traindata_idx[2 * k]) / n_noise if n_noise != 0 else 5.0
        # training on pseudo-labelled targets
        y, dist = pseudo_targets(self.targets, self.idxs, self.active, negative_param,
                                 self.traindata_idx)
        y, dist = torch.from_numpy(y), torch.from_numpy(dist)
        net.train()
        opt = _adam(net, a.base_lr)
        losses = []
        for pos in _batches(train_order, a.batch_size):
            _, z = net(self._img("image_aug_1", pos))
            loss = loss_stage2(z, y[pos], dist[pos])
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(loss.item())
        self.idxss = []
        for k in range(len(self.traindata_idx) // 2):
            sel = self.traindata_idx[2 * k] + self.traindata_idx[2 * k + 1]
            self.idxss.append(list(set(self.idxs) - set(sel)))     # :1197-1204
        t, proto = self._proto_pass(net, negative_param, zero_guard=True)
        return [net.state_dict(), float(np.mean(losses)), losses, t, proto, sims]
