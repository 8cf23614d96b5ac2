"""Validation path of the MHIM hot path.

``validate`` mirrors ``BaseTrainer.validate`` (engines/base_engine.py:234-329): every bag goes through the engine's
``validate_func`` (-> ``MHIM.forward_test`` on the HIP kernels), the logits stay on the device, the mean cross-entropy is
the reference's ``loss_cls_meter.avg``; ``get_metric_val`` mirrors engines/metrics.py:161-262 (same return tuples and
``rowd`` keys) with the torchmetrics collection replaced by one device evaluation (``ops.cls_metrics`` ->
mhimx_cls_metrics), including the DeterministicBootStrapper's resamples (engines/metrics.py:35-78): the resample indices
come from the same seeded ``torch.multinomial`` draw, so the sets are the reference's, and all resamples are evaluated in
the same four launches.  Survival (C-index) datasets are outside the path.
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import ops

_BOOT_SEED = 7784414403328510413            # engines/metrics.py:154


def bootstrap_indices(n, num_bootstraps, seed, device):
    """The resamples the reference draws (engines/metrics.py:27-28,59-63): ONE generator seeded once, ``num_bootstraps``
    successive multinomial draws of n out of n with replacement."""
    g = torch.Generator()
    g.manual_seed(seed)
    w = torch.ones(n)
    idx = torch.stack([torch.multinomial(w, num_samples=n, replacement=True, generator=g) for _ in range(num_bootstraps)])
    return idx.to(device=device, dtype=torch.int64)


def get_cls_metrics(args, logits, labels, bootstrap, device=None):
    """engines/metrics.py:104-123.  Without bootstrap: seven python floats (Acc, AUC, Precision, Recall, F1, CK, Acc_micro);
    with: seven [mean, std] pairs in the reference's order."""
    C = int(args.n_classes)
    binm = C == 2 and bool(getattr(args, "bin_metric", False))
    logits = logits.detach().float().reshape(-1, C).contiguous()
    labels = labels.detach().reshape(-1).to(torch.int64).contiguous()
    if not bootstrap:
        m = ops.cls_metrics(logits, labels, C, binm)[0].cpu().tolist()
        return tuple(m)
    idx = bootstrap_indices(labels.numel(), int(args.num_bootstrap), int(getattr(args, "fold_curr", 0)) + _BOOT_SEED, logits.device)
    out = ops.cls_metrics(logits, labels, C, binm, sample_idx=idx).double()
    mean, std = out.mean(0).cpu().tolist(), (out.std(0) if out.shape[0] > 1 else torch.zeros(7)).cpu().tolist()
    return tuple([mean[i], std[i]] for i in range(7))


def get_metric_val(args, bag_logit, bag_labels, model, status, early_stopping, epoch, loss_avg, suffix=None):
    """engines/metrics.py:161-262 for the classification tasks."""
    boot = status in getattr(args, "bootstrap_mode", ())
    suffix = "" if suffix is None else "_" + str(suffix)
    acc, auc, prec, rec, f1, ck, acc_micro = get_cls_metrics(args, bag_logit, bag_labels, boot)
    if boot and status == "val":
        acc, auc, prec, rec, f1, ck, acc_micro = acc[0], auc[0], prec[0], rec[0], f1[0], ck[0], acc_micro[0]
    if status == "val":
        stop = False
        if early_stopping is not None:
            early_stopping(args, epoch, -(auc if getattr(args, "best_metric_index", 0) == 0 else acc), model)
            stop = early_stopping.early_stop
        rowd = OrderedDict([("acc", acc), ("precision", prec), ("recall", rec), ("fscore", f1), ("auc", auc), ("ck", ck),
                            ("acc_micro", acc_micro), ("loss", loss_avg)])
        rowd = OrderedDict((k + suffix, v) for k, v in rowd.items())
        return [auc, acc, prec, rec, f1, ck, acc_micro], stop, loss_avg, None, rowd
    if not boot:
        acc, auc, prec, rec, f1, ck, acc_micro = ([v, 0] for v in (acc, auc, prec, rec, f1, ck, acc_micro))
    rowd = OrderedDict([("acc", acc[0]), ("precision", prec[0]), ("recall", rec[0]), ("fscore", f1[0]), ("auc", auc[0]),
                        ("ck", ck[0]), ("acc_micro", acc_micro[0]), ("loss", loss_avg), ("acc_std", acc[1]), ("fscore_std", f1[1]),
                        ("auc_std", auc[1]), ("ck_std", ck[1]), ("acc_micro_std", acc_micro[1])])
    rowd = OrderedDict((k + suffix, v) for k, v in rowd.items())
    return [auc[0], acc[0], prec[0], rec[0], f1[0], ck[0], acc_micro[0], auc[1], acc[1], f1[1], ck[1], acc_micro[1]], loss_avg, rowd


def validate(engine, args, model, loader, criterion=None, early_stopping=None, epoch=None, status="val"):
    """engines/base_engine.py:234-329 for the MHIM models.  ``loader`` yields dicts with 'input' (bag [1,N,D] or [N,D]) and
    'target' ([1] int64) - the reference's batch layout (batch size 1).  Returns what BaseTrainer.validate returns."""
    model.eval()
    criterion = criterion or torch.nn.CrossEntropyLoss()
    logits_all, labels_all, loss_sum, count = [], [], None, 0
    with torch.no_grad():
        for i, batch in enumerate(loader):
            bag, label = batch["input"], batch["target"]
            dev = bag.device if bag.is_cuda else torch.device("cuda")
            bag, label = bag.to(dev, non_blocking=True), label.to(dev, non_blocking=True).reshape(-1)
            logits, labels = engine.validate_func(args, model=model, bag=bag, label=label, criterion=criterion, batch_size=label.numel(),
                                                  i=i, pos=batch.get("pos"))
            if logits is None:
                continue
            bs = logits.size(0)
            logits_all.append(logits.reshape(bs, -1))
            labels_all.append(label)
            loss = criterion(logits.view(bs, -1), labels.view(bs))
            loss_sum = loss if loss_sum is None else loss_sum + loss              # AverageMeter.update(loss, 1)
            count += 1
    bag_logit, bag_labels = torch.cat(logits_all), torch.cat(labels_all)
    loss_avg = float(loss_sum / count)
    return list(get_metric_val(args, bag_logit, bag_labels, model, status, early_stopping, epoch, loss_avg))
