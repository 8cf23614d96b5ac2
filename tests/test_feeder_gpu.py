"""Bag feeder — GPU box only: both modes deliver every bag intact and in order (ragged bag sizes,
.pt files and in-memory tensors); the streaming mode's double buffer is not overwritten while a step still reads it; a trainer
driven by either mode ends in the same state."""
import os

import numpy as np
import pytest
import torch

from mhim_mil_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bags(tmp_path, n_bags=7, d=64):
    srcs, keep = [], []
    for i in range(n_bags):
        t = torch.from_numpy(synth.bag(100 + i, 300 + 37 * i, d))
        keep.append(t)
        if i % 2:
            p = os.path.join(tmp_path, f"slide{i}.pt")
            torch.save(t if i % 3 else t.numpy(), p)                  # the reference's files hold tensors or ndarrays
            srcs.append(p)
        else:
            srcs.append(t)
    return srcs, keep


@pytest.mark.parametrize("resident", [True, False])
def test_feeder_delivers_every_bag_in_order(tmp_path, resident):
    from mhim_mil_amd.feeder import BagFeeder
    srcs, keep = _bags(str(tmp_path))
    order = [3, 0, 6, 6, 1, 5, 2, 4, 0]
    f = BagFeeder(srcs, labels=[i % 2 for i in range(len(srcs))], device=DEV, resident=resident, order=order)
    seen, sums = [], []
    for x, label, idx in f:
        assert x.is_cuda and x.shape == keep[idx].shape and int(label) == idx % 2
        # a slow consumer kernel: if the copy stream overwrote this half early, the checksum below would change
        acc = x.double().sum()
        for _ in range(20):
            acc = acc + (x.double() * 1e-9).sum()
        sums.append(acc)
        seen.append(idx)
        assert torch.equal(x.cpu(), keep[idx])
    assert seen == order and len(f) == len(order)
    for s, idx in zip(sums, order):
        ref = keep[idx].double().sum() * (1 + 20e-9)
        assert abs(float(s) - float(ref)) <= 1e-9 * abs(float(ref)) + 1e-6


def test_trainer_state_is_independent_of_the_feeding_mode(tmp_path):
    from mhim_mil_amd.engine import FusedTrainer
    from mhim_mil_amd.feeder import BagFeeder
    from mhim_mil_amd.mhim import MHIM
    cfg = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
               merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)
    srcs, _ = _bags(str(tmp_path), n_bags=4)
    base = synth.mhim_state(7, input_dim=64, merge_k=5)

    def run(resident):
        def mk():
            m = MHIM(input_dim=64, n_classes=2, baseline="attn", **cfg)
            sd = dict(base)
            sd["merge.global_q"] = sd["merge.global_q_mm"]
            m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
            m = m.to(DEV).train()
            m.merge.dropout = 0.0
            return m
        torch.manual_seed(5)
        tr = FusedTrainer(mk(), mk(), aux_alpha=0.5)
        for x, label, _ in BagFeeder(srcs, [0, 1, 1, 0], device=DEV, resident=resident, order=[0, 1, 2, 3, 1]):
            tr.train_step(x, label)
        torch.cuda.synchronize()
        return tr.flat.student.cpu().numpy()

    np.testing.assert_array_equal(run(True), run(False))


def test_bag_loader_feeds_the_reference_train_loop_body():
    """BagLoader yields the batch dictionaries of the reference's loader seam (dataset_feat.py:93-111 + PrefetchLoader); the body of
    BaseTrainer.train (base_engine.py:52-93) is restated around CommonMIL.forward_func with the keyword set the reference passes."""
    import types
    from mhim_mil_amd.engine import CommonMIL
    from mhim_mil_amd.feeder import BagLoader
    from mhim_mil_amd.mhim import MHIM
    d = 64
    cfg = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True, merge_k=5,
               merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.25)
    s = MHIM(input_dim=d, n_classes=2, baseline="attn", **cfg).to(DEV).train()
    t = MHIM(input_dim=d, n_classes=2, baseline="attn", **cfg).to(DEV).train()
    t.load_state_dict(s.state_dict())
    bags = [torch.from_numpy(synth.bag(40 + i, 300 + 57 * i, d)) for i in range(4)]
    for resident in (True, False):
        loader = BagLoader(bags, [0, 1, 1, 0], device=DEV, resident=resident, return_id=True, names=[f"slide_{i}.pt" for i in range(4)])
        assert len(loader) == 4
        eng = CommonMIL(None)
        args = types.SimpleNamespace(model="mhim", baseline="attn", aux_alpha=0.5)
        eng.init_func_train(args)
        seen = []
        for batch_idx, batch in enumerate(loader):
            bag, label = batch["input"], batch["target"]                    # base_engine.py:59-64
            batch_size = label.size(0)
            pos, idx, feat = batch.get("pos", None), batch.get("idx", None), batch.get("feat", None)
            assert bag.is_cuda and bag.dim() == 3 and bag.shape[0] == 1 and label.is_cuda and label.dtype == torch.int64 and batch_size == 1
            out = eng.forward_func(args, s, t, bag, label, None, batch_size, batch_idx, 0, batch_idx, pos, loader=loader, device=DEV,
                                   others={}, idx=idx, feat=feat)
            logits, lab, aux, patch_num, keep_num = out[:5]
            assert logits.shape == (1, 2) and lab is label and patch_num == bags[batch_idx].shape[0] and 0 < keep_num < patch_num
            seen.append(idx[0])
        assert seen == [f"slide_{i}.pt" for i in range(4)]
