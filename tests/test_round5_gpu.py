"""Round-5 GPU tests: ADVICE r4 fixes and the step-level C entry point."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(mask_ratio_h=0.03, mask_ratio_hr=0.5, merge_enable=True, merge_k=5, merge_mm=0.9999, merge_ratio=0.9, act="gelu", da_act="relu",
           attn2score=True, temp_t=0.1)


def _models(D=256, dropout=0.0, seed=3):
    from mhim_mil_amd.mhim import MHIM
    torch.manual_seed(seed)
    s = MHIM(input_dim=D, n_classes=2, baseline="attn", dropout=dropout, **CFG).cuda().train()
    t = copy.deepcopy(s)
    t.merge_test = False
    return s, t.train()


def test_tea_type_same_trains_the_student():
    """--tea_type same (modules/__init__.py:211-212, base_engine.py:157-158): model_ema IS model.  The fused trainer adopts the module once,
    runs no EMA, and the student's own parameters move by Adam (they moved only through the EMA, at rate 1 - mm)."""
    from mhim_mil_amd.engine import FusedTrainer
    from mhim_mil_amd.optim import FusedAdamEMA
    s, _ = _models()
    before = {n: p.detach().clone() for n, p in s.named_parameters()}
    opt = FusedAdamEMA(s, s, lr=1e-3, mm=0.9997)
    tr = opt.trainer
    assert tr.flat.same_teacher and tr.flat.teacher is tr.flat.student
    assert len(list(s.parameters())) > 0 and not getattr(s, "_ema_owned", False)
    # the parameters are views of the STUDENT buffer (the one Adam updates)
    lo, hi = tr.flat.student.data_ptr(), tr.flat.student.data_ptr() + tr.flat.student.numel() * 4
    assert all(lo <= p.data_ptr() < hi for p in s.parameters())
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(1500, 256, device="cuda", generator=g).abs_()
    lab = torch.tensor([1], device="cuda")
    for _ in range(2):
        tr.train_step(x, lab)
    torch.cuda.synchronize()
    moved = {n: (p.detach() - before[n]).abs().max().item() for n, p in s.named_parameters()}
    # Adam's first steps move every trained weight by ~lr; an EMA-only drift would be (1 - mm) * lr ~ 3e-7
    assert moved["feature.0.weight"] > 2e-4, moved
    assert moved["online_encoder.attention.attention.0.weight"] > 2e-4, moved
    opt.close()


def test_shape_cached_falls_back_for_steps_it_cannot_capture():
    """shape_cached only captures the single-pass step; a v1 mask ratio (host read-back in get_mask) runs eagerly, every time."""
    from mhim_mil_amd.engine import FusedTrainer
    from mhim_mil_amd.mhim import MHIM
    torch.manual_seed(1)
    cfg = dict(CFG, mask_ratio_l=0.05)
    s = MHIM(input_dim=128, n_classes=2, baseline="attn", dropout=0.0, **cfg).cuda().train()
    t = copy.deepcopy(s).train()
    t.merge_test = False
    tr = FusedTrainer(s, t)
    x = torch.rand(600, 128, device="cuda")
    lab = torch.tensor([0], device="cuda")
    for _ in range(3):
        assert tr.shape_cached("train_step", x, lab) is None
        tr.train_step(x, lab)                                      # the eager path still works after the refusal
    torch.cuda.synchronize()


def test_shape_cached_key_and_lru():
    from mhim_mil_amd.engine import FusedTrainer
    s, t = _models(D=128)
    tr = FusedTrainer(s, t)
    lab = torch.tensor([1], device="cuda")
    bags = {n: torch.rand(n, 128, device="cuda") for n in (640, 704, 768)}
    for n in (640, 704, 640, 704, 640, 768, 768):                 # 640 and 704 captured; 768 evicts the least recently used (704)
        assert tr.shape_cached("train_step", bags[n], lab, cache=2) is not None
    keys = [k[1] for k in tr._shape_graphs["graphs"]]
    assert keys == [(640, 128), (768, 128)], keys
    # a changed loss weight is a different graph (it is baked into the head kernel's arguments)
    tr.aux_alpha = 0.25
    n_before = len(tr._shape_graphs["seen"])
    tr.shape_cached("train_step", bags[640], lab, cache=2)
    assert len(tr._shape_graphs["seen"]) == n_before + 1
    torch.cuda.synchronize()


def test_pinned_step_rejects_a_cpu_label():
    from mhim_mil_amd import _lib as L
    from mhim_mil_amd.engine import FusedTrainer
    s, t = _models(D=128)
    tr = FusedTrainer(s, t)
    x = torch.rand(640, 128, device="cuda")
    with pytest.raises(L.MhimxError):
        tr.train_step(x, torch.tensor([1]))
    with pytest.raises(L.MhimxError):
        tr.train_step(x, torch.tensor([1], device="cuda", dtype=torch.int32))
