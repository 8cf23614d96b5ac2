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


# ------------------------------------------------------------------------------------------------- the step behind the C-ABI (mhimx_step_run)
def _pair_of_trainers(D=512, dropout=0.25, **kw):
    from mhim_mil_amd.engine import FusedTrainer
    out = []
    for _ in range(2):
        torch.manual_seed(11)
        s, t = _models(D=D, dropout=dropout, seed=11)
        out.append(FusedTrainer(s, t, lr=1e-3, mm=0.999, aux_alpha=0.5, **kw))
    return out


def _state(tr):
    fl = tr.flat
    return [fl.student.clone(), fl.teacher.clone(), fl.m.clone(), fl.v.clone(), tr.opt_step.clone(), tr.tick.clone()]


def _same_state(tr_c, tr_p, steps, lr=1e-3):
    """Round 6: up to 16 384 rows the executor's scorer backward writes the stay rows' share of the dPRE image itself (mhimx_pool_grad.img) and
    the image is ordered [stay | merge]: the projection's weight / bias gradient is the same sum in another order, so the two paths' parameters
    agree to rounding - which Adam turns into up to ~2 lr on an element whose gradient is at rounding level - and no longer bit for bit
    (MHIMX_FUSE_DPRE=0 in the environment: the old route, bit-identical again).  Counters stay exact."""
    import os
    sc, sp = _state(tr_c), _state(tr_p)
    assert torch.equal(sc[4], sp[4]) and torch.equal(sc[5], sp[5])
    if os.environ.get("MHIMX_FUSE_DPRE", "1") == "0":
        for a, b in zip(sc[:4], sp[:4]):
            assert torch.equal(a, b)
        return
    for name, a, b in zip(("student", "teacher", "m", "v"), sc[:4], sp[:4]):
        d = (a - b).abs()
        assert float(d.mean()) <= 4e-6 * steps and float(d.max()) <= 2.2 * lr * steps, (name, float(d.mean()), float(d.max()))


def test_step_executor_equals_the_python_step_bit_for_bit():
    """mhimx_step_run issues the launches of FusedTrainer._forward_backward_nat + _apply itself: the same kernels, arguments and seeds, so
    logits, row sets, teacher scores and merged tokens of the first step agree BIT FOR BIT with the Python orchestration, parameters and
    optimiser state to rounding (_same_state: the dPRE image's row order) - over bags whose size changes every step (nothing is captured),
    dropout on."""
    tr_c, tr_p = _pair_of_trainers()
    tr_p.use_executor = False
    g = torch.Generator(device="cuda").manual_seed(9)
    sizes = [2048, 1777, 3001, 1024, 2500]
    for step, n in enumerate(sizes):
        x = torch.randn(n, 512, device="cuda", generator=g).abs_()
        lab = torch.tensor([step % 2], device="cuda")
        lc, sc = tr_c.train_step(x, lab)
        lp, sp = tr_p.train_step(x, lab)
        assert tr_c._exec is not None and tr_c.last.get("ws") is not None, "the executor did not run"
        if step == 0:
            assert torch.equal(lc, lp) and torch.equal(sc, sp), (step, lc, lp)
            assert torch.equal(tr_c.last["rows"], tr_p.last["rows"]) and torch.equal(tr_c.last["score"], tr_p.last["score"])
            assert torch.equal(tr_c.last["tokens"], tr_p.last["tokens"])
        else:
            assert torch.allclose(lc, lp, atol=2e-4) and torch.allclose(sc, sp, atol=2e-4), (step, lc, lp)
        _same_state(tr_c, tr_p, step + 1)
    assert tr_c.flat.step == tr_p.flat.step == len(sizes)


def test_step_executor_forward_backward_only_and_clip():
    """update = 0 (forward_backward: the complete gradient, no optimiser) and the --clip_grad path (the update stays outside the call)."""
    from mhim_mil_amd.engine import FusedTrainer
    tr_c, tr_p = _pair_of_trainers(clip_grad=0.5)
    tr_p.use_executor = False
    x = torch.rand(1500, 512, device="cuda")
    lab = torch.tensor([1], device="cuda")
    for tr in (tr_c, tr_p):
        tr.forward_backward(x, lab)
    gc, gp = tr_c.flat.grad, tr_p.flat.grad
    assert gc.abs().max() > 0 and float((gc - gp).abs().max()) <= 2e-6 * float(gp.abs().max())
    n_w1 = tr_c.s.feature[0].weight.numel() + tr_c.s.feature[0].bias.numel()
    assert torch.equal(gc[n_w1:], gp[n_w1:])                   # (everything but the projection's gradient pair: the same launches, the same bits)
    for tr in (tr_c, tr_p):
        tr.update()
        tr.train_step(x, lab)
    _same_state(tr_c, tr_p, 2)


def test_run_steps_equals_train_steps():
    """mhimx_step_run_many: a chunk of bags as ONE C call == train_step bag after bag."""
    tr_m, tr_1 = _pair_of_trainers()
    g = torch.Generator(device="cuda").manual_seed(4)
    bags = [torch.randn(n, 512, device="cuda", generator=g).abs_() for n in (1200, 1536, 999, 2048)]
    labels = [torch.tensor([j % 2], device="cuda") for j in range(len(bags))]
    lm, _ = tr_m.run_steps(bags, labels)
    for b, l in zip(bags, labels):
        l1, _ = tr_1.train_step(b, l)
    assert torch.equal(lm, l1)
    for a, b in zip(_state(tr_m), _state(tr_1)):
        assert torch.equal(a, b)


def test_step_executor_under_capture_and_shape_cache():
    """The executor is enqueue-only: a captured step (shape_cached: eager, captured on the second visit, replayed afterwards) through it
    equals the same sequence through the Python orchestration bit for bit (both captures freeze the same host-side seeds)."""
    tr_c, tr_p = _pair_of_trainers()
    tr_p.use_executor = False
    x = torch.rand(1800, 512, device="cuda")
    lab = torch.tensor([0], device="cuda")
    for _ in range(4):
        for tr in (tr_c, tr_p):
            assert tr.shape_cached("train_step", x, lab) is not None
    torch.cuda.synchronize()
    _same_state(tr_c, tr_p, 4)


def test_step_counts_match_the_reference_formulas():
    import ctypes as C
    from mhim_mil_amd import _lib as L
    s, _ = _models(D=256)
    for n in (64, 257, 1000, 9999, 10000, 16384):
        c = L.StepCounts()
        L.check(L.lib().mhimx_step_counts_of(n, s.mask_ratio_h, s.mask_ratio_hr, s.merge.merge_ratio, C.byref(c)))
        assert (c.k_top, c.n_sel, c.len_keep, c.Lk, c.R) == s.v2_counts(n), n


@pytest.mark.parametrize("baseline", ["selfattn", "dsmil"])
def test_shape_cached_captures_the_transmil_and_dsmil_students(baseline):
    """VERDICT r4 missing 3: shape_cached was ABMIL-only.  The TransMIL / DSMIL students (autograd over kernel-backed nodes) run every step of
    a cached shape on the trainer's capture stream: eager twice, captured on the third visit, replayed afterwards; the parameters keep
    moving and stay finite, and the device step counters advance under replay."""
    from mhim_mil_amd.engine import FusedTrainer
    from mhim_mil_amd.mhim import MHIM
    torch.manual_seed(5)
    s = MHIM(input_dim=256, n_classes=2, baseline=baseline, dropout=0.25, **CFG).cuda().train()
    t = copy.deepcopy(s).train()
    t.merge_test = False
    tr = FusedTrainer(s, t, lr=1e-3)
    x = torch.rand(700, 256, device="cuda")
    lab = torch.tensor([1], device="cuda")
    snaps = []
    for it in range(6):
        out = tr.shape_cached("train_step", x, lab)
        assert out is not None, it
        torch.cuda.synchronize()
        assert torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all()
        snaps.append(tr.flat.student.clone())
    assert len(tr._shape_graphs["graphs"]) == 1 and not tr._shape_graphs["bad"]
    for a, b in zip(snaps[:-1], snaps[1:]):
        assert not torch.equal(a, b) and torch.isfinite(b).all()
    assert int(tr.opt_step.item()) == 6 and tr.flat.step == 6


def test_persistent_projection_equals_one_workgroup_per_tile():
    """Round 5: launches of more than 256 tiles run PERSISTENT workgroups that request the next tile's first stages from inside the epilogue
    (csrc/bag_project_ws.hip).  (a) one bag of 24 000 rows (150 row tiles x 4 column tiles = 600 tiles, residual rows, bias, two heads)
    against the same rows projected in chunks of <= 252 tiles - one tile per workgroup there: bit-identical (a tile's arithmetic does not
    depend on who runs it); (b) three bags of 10 000 rows in ONE launch (768 virtual tiles, dropout, d out / d pre) against per-bag launches
    of 252 tiles: bit-identical masks and rows."""
    from mhim_mil_amd import ops
    DEV = "cuda"
    g = torch.Generator(device=DEV).manual_seed(5)
    n, d, E = 24000, 512, 512
    x = torch.randn(n, d, device=DEV, generator=g)
    wa, wb = (torch.randn(E, d, device=DEV, generator=g) * 0.05 for _ in range(2))
    ba, bb = (torch.randn(E, device=DEV, generator=g) * 0.1 for _ in range(2))
    wap, wbp = ops.pair_planes(wa), ops.pair_planes(wb)
    res = torch.randn(n, E, device=DEV, generator=g)
    full = ops.bag_project(x, [ops.ProjHead(wap, ba, resid=res), ops.ProjHead(wbp, bb)], act=0)
    chunk = 4800                                               # 30 row tiles x 4 = 120 tiles per launch
    for lo in range(0, n, chunk):
        part = ops.bag_project(x[lo:lo + chunk], [ops.ProjHead(wap, ba, resid=res[lo:lo + chunk]), ops.ProjHead(wbp, bb)], act=0)
        assert torch.equal(part[0].out, full[0].out[lo:lo + chunk]) and torch.equal(part[1].out, full[1].out[lo:lo + chunk]), lo
    # PLAIN launches (no activation / dropout / residual rows / d out-d pre): the consumer waves store their accumulators from registers, every
    # tile takes the direct prologue - against the staged epilogue of the same products (the residual head above minus its residual rows)
    plain = ops.bag_project(x[:n - 37], [ops.ProjHead(wap, ba), ops.ProjHead(wbp, None)], act=0)      # (a ragged last tile: 23 963 rows)
    assert torch.equal(plain[0].out, full[0].out[:n - 37] - res[:n - 37]) or \
        float((plain[0].out - (full[0].out[:n - 37] - res[:n - 37])).abs().max()) < 2e-6        # (x + r - r: one rounding of the residual add)
    refb = x[:n - 37].double() @ wb.double().t()
    assert float((plain[1].out.double() - refb).abs().max() / refb.abs().max()) < 1e-5
    assert torch.equal(plain[1].out + bb, full[1].out[:n - 37]) or float((plain[1].out + bb - full[1].out[:n - 37]).abs().max()) < 1e-6
    small = ops.bag_project(x[:1000], [ops.ProjHead(wap, ba)], act=0)                                   # one tile per workgroup, ragged
    assert torch.equal(small[0].out, plain[0].out[:1000])
    ref = x.double() @ wa.double().t() + ba.double() + res.double()
    assert float((full[0].out.double() - ref).abs().max() / ref.abs().max()) < 1e-5           # (three-term bf16 products, K = 512)
    for dk in (32, 64):                                        # one and two k-steps per tile: the prologues' clamped requests (persistent, plain and staged)
        xs_, wk = x[:, :dk].contiguous(), wa[:, :dk].contiguous()
        wkp = ops.pair_planes(wk)
        refk = xs_.double() @ wk.double().t()
        pk = ops.bag_project(xs_, [ops.ProjHead(wkp, None)], act=0)[0].out
        sk = ops.bag_project(xs_, [ops.ProjHead(wkp, None, resid=res)], act=0)[0].out
        assert float((pk.double() - refk).abs().max() / refk.abs().max()) < 1e-5, dk
        assert float((sk.double() - refk - res.double()).abs().max() / refk.abs().max()) < 1e-5, dk
    nb, nr, dd = 3, 10000, 256
    xs = [torch.randn(nr, dd, device=DEV, generator=g).abs_() for _ in range(nb)]
    wt, ws_ = (torch.randn(E, dd, device=DEV, generator=g) * 0.05 for _ in range(2))
    wtp, wsp = ops.pair_planes(wt), ops.pair_planes(ws_)
    tick = torch.tensor([9], dtype=torch.int64, device=DEV)
    mk = lambda b: [ops.ProjHead(wtp, ba, drop_p=0.25, drop_seed=50 + 2 * b), ops.ProjHead(wsp, bb, drop_p=0.25, drop_seed=51 + 2 * b, want_dact=True)]
    one = [ops.bag_project(xb, mk(b), act=1, drop_tick=tick) for b, xb in enumerate(xs)]
    many = ops.bag_project_multi(xs, [mk(b) for b in range(nb)], act=1, drop_tick=tick)
    for a, m in zip(one, many):
        assert torch.equal(a[0].out, m[0].out) and torch.equal(a[1].out, m[1].out) and torch.equal(a[1].dact, m[1].dact)
