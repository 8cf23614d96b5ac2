"""CPU oracle of the validation metrics.  TEST INFRASTRUCTURE ONLY: imported by tests/ and by nothing
that ships.

What the reference computes (engines/metrics.py:125-159 `metrics_base`, :104-123 `get_cls_metrics`): a torchmetrics
MetricCollection of Accuracy(macro), F1Score(macro), AUROC(macro), Precision(macro), Recall(macro), CohenKappa and
Accuracy(micro) in the multiclass task — or, with ``--bin_metric`` and two classes, the binary task on ``logits[:, 1]`` —
optionally wrapped in a DeterministicBootStrapper (engines/metrics.py:35-78: multinomial resampling with a seeded torch
generator, mean / std over the resamples).

torchmetrics (requirements.txt:11, unpinned) is NOT in /root/reference and not installed here: the algorithms below restate
its published 1.x classification semantics —
  * format: predictions that are not all inside [0,1] go through softmax (multiclass) / sigmoid (binary); hard labels are
    arg-max (multiclass) / prob > 0.5 (binary);
  * per-class tp / fp / fn from the confusion matrix; precision tp/(tp+fp), recall tp/(tp+fn), f1 2tp/(2tp+fp+fn), each
    0 when its denominator is 0; macro = mean over the classes with tp+fp+fn > 0; macro accuracy = macro recall;
    micro accuracy = correct / n;
  * Cohen's kappa (po - pe) / (1 - pe) from the confusion matrix;
  * AUROC with thresholds=None is the exact area under the ROC curve, i.e. the Mann-Whitney statistic
    (#{pos > neg} + 0.5 #{pos == neg}) / (P N) one-vs-rest per class; macro = mean over the classes that have both
    positives and negatives.
PARITY UNPINNED against torchmetrics itself (absent).  Pinned instead against scikit-learn (installed; the textbook
definitions coincide whenever every class has support): tests/test_metrics_cpu.py.
"""
from __future__ import annotations

import numpy as np

KEYS = ("Acc", "AUC", "Precision", "Recall", "F1", "CK", "Acc_micro")


def _softmax32(x):
    x = x.astype(np.float32)
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m, dtype=np.float32)
    return (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)


def _sigmoid32(x):
    x = x.astype(np.float32)
    return (np.float32(1.0) / (np.float32(1.0) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def _auc_pairs(score, pos):
    """Exact ROC area: (#{p > n} + 0.5 #{p == n}) / (P N); None when a side is empty."""
    sp, sn = score[pos], score[~pos]
    if sp.size == 0 or sn.size == 0:
        return None
    order = np.sort(sn)
    lo = np.searchsorted(order, sp, side="left")          # negatives strictly below each positive
    hi = np.searchsorted(order, sp, side="right")
    twice = int((2 * lo + (hi - lo)).sum())               # 2 * (greater + 0.5 * ties): an integer
    return twice / (2.0 * sp.size * sn.size)


def _safe(a, b):
    return a / b if b > 0 else 0.0


def cls_metrics(logits, labels, n_classes, bin_metric=False):
    """logits [n, C] (or [n] scores when bin_metric), labels [n] ints -> dict over KEYS (python floats)."""
    logits = np.asarray(logits, dtype=np.float32)
    labels = np.asarray(labels).astype(np.int64).reshape(-1)
    n = labels.size
    if n_classes == 2 and bin_metric:
        s = logits[:, 1] if logits.ndim == 2 else logits
        if not np.all((s >= 0) & (s <= 1)):
            s = _sigmoid32(s)
        pred = (s > np.float32(0.5)).astype(np.int64)
        tp = int(((pred == 1) & (labels == 1)).sum()); fp = int(((pred == 1) & (labels == 0)).sum())
        fn = int(((pred == 0) & (labels == 1)).sum()); tn = int(((pred == 0) & (labels == 0)).sum())
        acc = _safe(tp + tn, n)
        auc = _auc_pairs(s, labels == 1)
        conf = np.array([[tn, fp], [fn, tp]], dtype=np.float64)
        out = {"Acc": acc, "AUC": 0.0 if auc is None else auc, "Precision": _safe(tp, tp + fp), "Recall": _safe(tp, tp + fn),
               "F1": _safe(2 * tp, 2 * tp + fp + fn), "CK": _kappa(conf), "Acc_micro": acc}
        return out
    C = int(n_classes)
    p = logits
    if not np.all((p >= 0) & (p <= 1)):
        p = _softmax32(p)
    pred = p.argmax(axis=1)                                # first maximum, as torch.argmax
    conf = np.zeros((C, C), dtype=np.float64)              # conf[target, pred]
    np.add.at(conf, (labels, pred), 1.0)
    tp = np.diag(conf); fp = conf.sum(0) - tp; fn = conf.sum(1) - tp
    w = ((tp + fp + fn) > 0).astype(np.float64)
    prec = np.array([_safe(tp[c], tp[c] + fp[c]) for c in range(C)])
    rec = np.array([_safe(tp[c], tp[c] + fn[c]) for c in range(C)])
    f1 = np.array([_safe(2 * tp[c], 2 * tp[c] + fp[c] + fn[c]) for c in range(C)])
    macro = lambda v: _safe(float((v * w).sum()), float(w.sum()))
    aucs = [a for a in (_auc_pairs(p[:, c], labels == c) for c in range(C)) if a is not None]
    return {"Acc": macro(rec), "AUC": float(np.mean(aucs)) if aucs else 0.0, "Precision": macro(prec), "Recall": macro(rec),
            "F1": macro(f1), "CK": _kappa(conf), "Acc_micro": _safe(float(tp.sum()), n)}


def _kappa(conf):
    n = conf.sum()
    if n == 0:
        return 0.0
    po = np.trace(conf) / n
    pe = float((conf.sum(0) * conf.sum(1)).sum()) / (n * n)
    return float((po - pe) / (1.0 - pe)) if pe != 1.0 else 0.0


def bootstrap_metrics(logits, labels, n_classes, sample_idx, bin_metric=False):
    """sample_idx [B, n] resampled row ids (engines/metrics.py:62-66) -> dict key -> (mean, std with ddof=1)."""
    rows = [cls_metrics(np.asarray(logits)[ix], np.asarray(labels)[ix], n_classes, bin_metric) for ix in np.asarray(sample_idx)]
    out = {}
    for k in KEYS:
        v = np.array([r[k] for r in rows], dtype=np.float64)
        out[k] = (float(v.mean()), float(v.std(ddof=1)) if v.size > 1 else 0.0)
    return out
