"""mhimx_cls_metrics and the validation loop against the CPU oracle."""
import types

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO
from oracle import mhim_oracle as O
from mhim_mil_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from mhim_mil_amd import ops
    return ops


def _data(seed, n, C, ties=False):
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, C, size=n)
    logits = rng.normal(size=(n, C)).astype(np.float32) + 1.5 * np.eye(C, dtype=np.float32)[labels] * rng.random((n, 1)).astype(np.float32)
    if ties:
        logits = np.round(logits * 2) / 2
    return logits, labels


def _close(got, ref, n):
    ref = np.array([ref[k] for k in MO.KEYS])
    # the count-based metrics are exact; the AUC may move by a few of its ~n^2/4 pairs where the device expf and numpy's
    # exp round two nearly equal probabilities to a different order (tie-heavy inputs)
    atol = np.full(7, 2e-7)
    atol[1] = 60.0 / (n * n) + 2e-7
    assert np.all(np.abs(got - ref) <= atol + 2e-6 * np.abs(ref)), (got, ref)


@pytest.mark.parametrize("C,ties,binm,n", [(2, False, False, 500), (2, True, False, 500), (2, True, True, 333), (2, False, True, 1500),
                                           (4, True, False, 700), (3, False, False, 31)])
def test_cls_metrics_vs_oracle(C, ties, binm, n):
    ops = _ops()
    logits, labels = _data(7 * C + n, n, C, ties)
    got = ops.cls_metrics(torch.from_numpy(logits).to(DEV), torch.from_numpy(labels).to(DEV), C, binm).cpu().numpy()[0]
    _close(got, MO.cls_metrics(logits, labels, C, binm), n)


def test_cls_metrics_probability_inputs_and_absent_class():
    ops = _ops()
    rng = np.random.default_rng(3)
    p = rng.random((200, 3)).astype(np.float32)
    p /= p.sum(1, keepdims=True)
    labels = rng.integers(0, 2, size=200)
    got = ops.cls_metrics(torch.from_numpy(p).to(DEV), torch.from_numpy(labels).to(DEV), 3).cpu().numpy()[0]
    _close(got, MO.cls_metrics(p, labels, 3), 200)


def test_bootstrap_resamples_in_one_evaluation():
    ops = _ops()
    from mhim_mil_amd import validate as V
    n, C, B = 240, 2, 25
    logits, labels = _data(21, n, C)
    idx = V.bootstrap_indices(n, B, 5 + V._BOOT_SEED, DEV)
    g = torch.Generator(); g.manual_seed(5 + V._BOOT_SEED)                       # the reference's draw (engines/metrics.py:27-28,59-63)
    ref_idx = torch.stack([torch.multinomial(torch.ones(n), num_samples=n, replacement=True, generator=g) for _ in range(B)])
    assert torch.equal(idx.cpu(), ref_idx)
    got = ops.cls_metrics(torch.from_numpy(logits).to(DEV), torch.from_numpy(labels).to(DEV), C, False, sample_idx=idx).cpu().numpy()
    for b in range(B):
        ix = ref_idx[b].numpy()
        _close(got[b], MO.cls_metrics(logits[ix], labels[ix], C), n)
    args = types.SimpleNamespace(n_classes=2, bin_metric=False, num_bootstrap=B, fold_curr=5)
    pairs = V.get_cls_metrics(args, torch.from_numpy(logits).to(DEV), torch.from_numpy(labels).to(DEV), True)
    ref = MO.bootstrap_metrics(logits, labels, C, ref_idx.numpy())
    order = ("Acc", "AUC", "Precision", "Recall", "F1", "CK", "Acc_micro")
    for pr, k in zip(pairs, order):
        np.testing.assert_allclose(pr, ref[k], atol=2e-5)


def test_validate_loop_matches_oracle_forward_and_metrics():
    """BaseTrainer.validate restated: forward_test per bag on the HIP kernels, CE mean, metrics on the device - against the
    oracle's forward_test + metric restatement on the same synthetic bags."""
    from mhim_mil_amd import validate as V
    from mhim_mil_amd.engine import CommonMIL
    from mhim_mil_amd.mhim import MHIM
    from tests.test_mhim_gpu import build, V2
    state = synth.mhim_state(3, input_dim=256, merge_k=5)
    model = build(state, "auto", input_dim=256, **V2)
    rng = np.random.default_rng(0)
    bags, labels = [], []
    for b in range(24):
        n = int(rng.integers(40, 200))
        bags.append(synth.bag(100 + b, n, 256))
        labels.append(int(rng.integers(0, 2)))
    loader = [{"input": torch.from_numpy(x).unsqueeze(0), "target": torch.tensor([y])} for x, y in zip(bags, labels)]
    args = types.SimpleNamespace(model="mhim", baseline="attn", n_classes=2, bin_metric=False, bootstrap_mode=(), best_metric_index=0)
    out = V.validate(CommonMIL(args), args, model, loader, status="val")
    po = {k: torch.as_tensor(v) for k, v in state.items()}
    cfg = O.Cfg(**V2)
    ref_logits = np.stack([O.forward_test(torch.from_numpy(x), po, cfg).reshape(-1).float().numpy() for x in bags])
    ref = MO.cls_metrics(ref_logits, np.array(labels), 2)
    got = dict(zip(("AUC", "Acc", "Precision", "Recall", "F1", "CK", "Acc_micro"), out[0]))
    for k in MO.KEYS:
        np.testing.assert_allclose(got[k], ref[k], atol=1e-5, err_msg=k)
    lg = torch.from_numpy(ref_logits).double()
    ce = torch.nn.functional.cross_entropy(lg, torch.tensor(labels)).item()
    np.testing.assert_allclose(out[2], ce, rtol=2e-4)
    assert list(out[4].keys()) == ["acc", "precision", "recall", "fscore", "auc", "ck", "acc_micro", "loss"]


def test_known_answer_vectors_on_device():
    """The same hand-computed cases through mhimx_cls_metrics (csrc/metrics.hip): exact counts, AUROC as pair counts."""
    from tests.metrics_known_answers import CASES
    ops = _ops()
    for name, logits, labels, C, binm, expect in CASES:
        got = ops.cls_metrics(torch.from_numpy(logits).to(DEV), torch.from_numpy(labels).to(DEV), C, binm).cpu().numpy()[0]
        np.testing.assert_allclose(got, expect, atol=1e-6, err_msg=name)
