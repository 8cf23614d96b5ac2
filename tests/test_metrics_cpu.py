"""Validation-metric oracle (oracle/metrics_oracle.py) against scikit-learn, the installed independent implementation of the
same textbook definitions (torchmetrics, which the reference uses, is absent: see the oracle's header)."""
import numpy as np
import pytest

from oracle import metrics_oracle as MO

sk = pytest.importorskip("sklearn.metrics")


def _data(seed, n, C, ties=False):
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, C, size=n)
    logits = rng.normal(size=(n, C)).astype(np.float32) + 1.5 * np.eye(C, dtype=np.float32)[labels] * rng.random((n, 1)).astype(np.float32)
    if ties:
        logits = np.round(logits * 2) / 2                       # many exactly equal scores
    return logits, labels


@pytest.mark.parametrize("C,ties", [(2, False), (2, True), (3, False), (4, True)])
def test_multiclass_matches_sklearn(C, ties):
    logits, labels = _data(11 + C, 400, C, ties)
    m = MO.cls_metrics(logits, labels, C)
    p = MO._softmax32(logits)
    pred = p.argmax(1)
    np.testing.assert_allclose(m["Acc"], sk.balanced_accuracy_score(labels, pred), atol=1e-12)
    np.testing.assert_allclose(m["Acc_micro"], sk.accuracy_score(labels, pred), atol=1e-12)
    np.testing.assert_allclose(m["Precision"], sk.precision_score(labels, pred, average="macro", zero_division=0), atol=1e-12)
    np.testing.assert_allclose(m["Recall"], sk.recall_score(labels, pred, average="macro", zero_division=0), atol=1e-12)
    np.testing.assert_allclose(m["F1"], sk.f1_score(labels, pred, average="macro", zero_division=0), atol=1e-12)
    np.testing.assert_allclose(m["CK"], sk.cohen_kappa_score(labels, pred), atol=1e-12)
    if C == 2:
        auc = sk.roc_auc_score(labels, p[:, 1].astype(np.float64))
    else:
        auc = np.mean([sk.roc_auc_score((labels == c).astype(int), p[:, c].astype(np.float64)) for c in range(C)])
    np.testing.assert_allclose(m["AUC"], auc, atol=1e-12)


@pytest.mark.parametrize("ties", [False, True])
def test_binary_task_matches_sklearn(ties):
    logits, labels = _data(5, 300, 2, ties)
    m = MO.cls_metrics(logits, labels, 2, bin_metric=True)
    s = MO._sigmoid32(logits[:, 1])
    pred = (s > 0.5).astype(int)
    np.testing.assert_allclose(m["Acc"], sk.accuracy_score(labels, pred), atol=1e-12)
    np.testing.assert_allclose(m["Precision"], sk.precision_score(labels, pred, zero_division=0), atol=1e-12)
    np.testing.assert_allclose(m["Recall"], sk.recall_score(labels, pred, zero_division=0), atol=1e-12)
    np.testing.assert_allclose(m["F1"], sk.f1_score(labels, pred, zero_division=0), atol=1e-12)
    np.testing.assert_allclose(m["CK"], sk.cohen_kappa_score(labels, pred), atol=1e-12)
    np.testing.assert_allclose(m["AUC"], sk.roc_auc_score(labels, s.astype(np.float64)), atol=1e-12)


def test_probabilities_are_not_transformed_again_and_absent_class():
    rng = np.random.default_rng(3)
    p = rng.random((50, 3)).astype(np.float32)
    p /= p.sum(1, keepdims=True)
    labels = rng.integers(0, 2, size=50)                          # class 2 never occurs as a target
    m = MO.cls_metrics(p, labels, 3)
    pred = p.argmax(1)
    present = sorted(set(labels.tolist()) | set(pred.tolist()))
    np.testing.assert_allclose(m["F1"], sk.f1_score(labels, pred, average="macro", labels=present, zero_division=0), atol=1e-12)
    aucs = [sk.roc_auc_score((labels == c).astype(int), p[:, c].astype(np.float64)) for c in (0, 1)]
    np.testing.assert_allclose(m["AUC"], np.mean(aucs), atol=1e-12)


def test_bootstrap_mean_std():
    logits, labels = _data(9, 120, 2)
    rng = np.random.default_rng(1)
    idx = rng.integers(0, 120, size=(16, 120))
    b = MO.bootstrap_metrics(logits, labels, 2, idx)
    rows = np.array([[MO.cls_metrics(logits[i], labels[i], 2)[k] for k in MO.KEYS] for i in idx])
    for j, k in enumerate(MO.KEYS):
        np.testing.assert_allclose(b[k][0], rows[:, j].mean(), atol=1e-12)
        np.testing.assert_allclose(b[k][1], rows[:, j].std(ddof=1), atol=1e-12)


def test_known_answer_vectors():
    """Hand-computed cases (tests/metrics_known_answers.py): torchmetrics cannot be imported here, so the oracle is also pinned to
    numbers worked out from the definitions, independent of scikit-learn and of the oracle's own code."""
    from tests.metrics_known_answers import CASES
    for name, logits, labels, C, binm, expect in CASES:
        m = MO.cls_metrics(logits, labels, C, bin_metric=binm)
        np.testing.assert_allclose([m[k] for k in MO.KEYS], expect, atol=1e-12, err_msg=name)
