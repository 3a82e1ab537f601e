"""globaltest drop-in (reference: utils/evaluations.py:15-73 + utils/multilabel_metrixs.py).

The eval-mode forward over the test set runs on the HIP engine (batches of 4*batch_size like
:18); the metrics are small host-side numpy restatements: per-class average precision and ROC
AUC follow scikit-learn's definitions (the reference calls average_precision_score / roc_curve
+ auc, :41-49, :60-66), BACC / R / P / F1 / Hamming follow utils/multilabel_metrixs.py.
"""
import numpy as np
import torch


def average_precision(y_true, score):
    """sklearn.metrics.average_precision_score for one binary column:
    AP = sum_n (R_n - R_{n-1}) P_n over the distinct score thresholds, descending."""
    y_true = np.asarray(y_true, dtype=np.float64)
    score = np.asarray(score, dtype=np.float64)
    order = np.argsort(-score, kind="mergesort")
    y, s = y_true[order], score[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, y.size - 1]
    tps = np.cumsum(y)[idx]
    fps = 1 + idx - tps
    precision = tps / (tps + fps)
    recall = tps / tps[-1] if tps[-1] > 0 else np.full_like(tps, np.nan)
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def roc_auc(y_true, score):
    """auc(roc_curve(y, score)) -- trapezoid under the ROC of the distinct thresholds."""
    y_true = np.asarray(y_true, dtype=np.float64)
    score = np.asarray(score, dtype=np.float64)
    order = np.argsort(-score, kind="mergesort")
    y, s = y_true[order], score[order]
    idx = np.r_[np.where(np.diff(s))[0], y.size - 1]
    tps = np.r_[0.0, np.cumsum(y)[idx]]
    fps = np.r_[0.0, (1 + idx) - np.cumsum(y)[idx]]
    tpr, fpr = tps / tps[-1], fps / fps[-1]
    return float((np.trapezoid if hasattr(np, "trapezoid") else np.trapz)(tpr, fpr))


def multilabel_metrics(all_labels, all_probs, threshold=0.5):
    """mAP, BACC, R, F1, auc, P, hamming_loss exactly as globaltest assembles them."""
    y = np.asarray(all_labels)
    p = np.asarray(all_probs)
    pred = p > threshold
    C = y.shape[1]
    aps = [average_precision(y[:, c], p[:, c]) for c in range(C)]
    yt = y.astype(bool)
    tp = np.logical_and(yt, pred).sum(0).astype(np.float64)
    npos = yt.sum(0).astype(np.float64)
    npred = pred.sum(0).astype(np.float64)
    tn = (~np.logical_or(yt, pred)).sum(0).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        recall1 = tp / npos
        recall0 = tn / (y.shape[0] - npos)
        R = float(np.sum(recall1) / C)
        bacc = float(np.sum((recall0 + recall1) / 2) / C)
        F1 = float(np.sum(2 * tp / (npos + npred)) / C)
        # Precision skips classes without predictions but still divides by C (multilabel_metrixs.py:57-64)
        P = float(np.sum(np.where(npred > 0, tp / np.where(npred > 0, npred, 1), 0.0)) / C)
    hamming = float((yt != pred).sum() / (y.shape[0] * C))
    auroc = float(np.mean([roc_auc(y[:, c], p[:, c]) for c in range(C)]))
    return {"mAP": torch.tensor(aps).mean(), "BACC": bacc, "R": R, "F1": F1, "auc": auroc, "P": P,
            "hamming_loss": hamming}


def _hw(ds):
    s = ds[0]["image"]
    return int(s.shape[-2]), int(s.shape[-1])


def globaltest(net, test_dataset, args):
    """utils/evaluations.py:15-73 with the forward on the HIP engine."""
    net.eval()
    n = len(test_dataset)
    bs = args.batch_size * 4
    from .launch import default_device
    views = test_dataset.device_views(default_device()) if hasattr(test_dataset, "device_views") else None
    probs = []
    for i in range(0, n, bs):
        idx = list(range(i, min(n, i + bs)))
        if hasattr(test_dataset, "device_batch"):
            x = test_dataset.device_batch(net.bind(*_hw(test_dataset), bs), "image", idx)
        elif views is not None and "image" in views:
            x = views["image"][i:i + len(idx)]
        else:
            x = torch.stack([torch.as_tensor(test_dataset[j]["image"], dtype=torch.float32) for j in idx])
        _, logits = net(x)
        z = logits.cpu().numpy().astype(np.float32)
        probs.append((1.0 / (1.0 + np.exp(-z.astype(np.float32)))).astype(np.float32))
    all_probs = np.concatenate(probs, 0)
    assert all_probs.shape == (n, args.n_classes)
    return multilabel_metrics(np.array(test_dataset.targets), all_probs)
