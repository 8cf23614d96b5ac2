"""Parity of MHIM(baseline='dsmil') — GPU box only.

(1) the new kernels (column max / arg-max, row max, DSMIL head) vs torch statements; (2) the module against the fixture
generated from the reference import (g13: eval logits + attention, teacher, student step with every gradient, pure);
(3) the CommonMIL hook tuples and two FusedTrainer steps against the oracle's train_step.
Tolerances: logits 1e-4 abs; gradients 2e-3 of the tensor's scale; index sets exact.
"""
import json
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mhim_mil_amd import synth
from oracle import mhim_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu
DEV = "cuda"
V2 = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True,
          merge_k=5, merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)


def build(sd, **kw):
    from mhim_mil_amd.mhim import MHIM
    m = MHIM(baseline="dsmil", n_classes=2, **kw)
    sd = dict(sd)
    if "merge.global_q_mm" in sd:
        sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV)
    if kw.get("merge_enable", True):
        m.merge.dropout = 0.0
    return m


def X(seed, n, d):
    return torch.from_numpy(synth.bag(seed, n, d)).to(DEV).unsqueeze(0)


def close(got, ref, rtol, what=""):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def test_colmax_rowmax_kernels():
    from mhim_mil_amd import dsmil as DS
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5000, 3, generator=g)
    x[17, 1] = x[4000, 1] = 9.0                      # a planted tie: the lowest row wins
    xd = x.to(DEV).requires_grad_()
    vals, idx = DS.ColMax.apply(xd)
    np.testing.assert_array_equal(vals.detach().cpu().numpy(), x.max(0).values.numpy())
    assert idx.cpu().tolist() == [int(x[:, 0].argmax()), 17, int(x[:, 2].argmax())]
    (vals * torch.tensor([1.0, 2.0, 3.0], device=DEV)).sum().backward()
    ref = torch.zeros_like(x)
    ref[idx.cpu(), torch.arange(3)] = torch.tensor([1.0, 2.0, 3.0])
    np.testing.assert_array_equal(xd.grad.cpu().numpy(), ref.numpy())
    np.testing.assert_array_equal(DS.rowmax(x.to(DEV)).cpu().numpy(), x.max(1).values.numpy())


def test_dsmil_head_kernel():
    from mhim_mil_amd import dsmil as DS
    g = torch.Generator().manual_seed(2)
    lb, li = torch.randn(2, generator=g), torch.randn(2, generator=g)
    Bs, Bt = torch.randn(2, 512, generator=g), torch.randn(2, 512, generator=g) * 0.3
    lbr, lir, Bsr = lb.clone().double().requires_grad_(), li.clone().double().requires_grad_(), Bs.clone().double().requires_grad_()
    ce = F.cross_entropy((0.5 * lbr + 0.5 * lir).view(1, -1), torch.tensor([1]))
    cl = O.soft_target_ce(Bsr, Bt.double(), 0.1).mean()
    (1.0 * ce + 0.5 * cl).backward()
    losses, g_lb, g_li, g_B = DS.dsmil_head(lb.to(DEV), li.to(DEV), torch.tensor([1], device=DEV), Bs.to(DEV), Bt.to(DEV), 0.1, 1.0, 0.5)
    np.testing.assert_allclose(losses.cpu().numpy(), [float(ce + 0.5 * cl), float(ce), float(cl)], rtol=2e-5)
    close(g_lb, lbr.grad, 1e-5)
    close(g_li, lir.grad, 1e-5)
    close(g_B, Bsr.grad, 1e-4)
    # autograd form of the distillation term alone
    Bd = Bs.to(DEV).requires_grad_()
    c = DS.SoftTargetCE.apply(Bd, Bt.to(DEV), 0.1)
    assert abs(float(c) - float(cl)) < 1e-4
    (3.0 * c).backward()
    close(Bd.grad, 3.0 * Bsr.grad / 0.5, 1e-4)


def test_g13_dsmil_module_vs_reference_fixture():
    meta, a = G.load("g13_dsmil")
    d, n = meta["d"], meta["n"]
    cfgd = {k: meta[k] for k in V2 if k in meta}
    base = synth.mhim_state(meta["seed"], input_dim=d, merge_k=5, baseline="dsmil")
    x = X(meta["xseed"], n, d)
    m = build(base, input_dim=d, **cfgd).eval()
    logits, B = m.forward_test(x)
    assert logits[0].shape == (1, 2) and B.shape == (1, 2, 512)
    np.testing.assert_allclose(logits[0][0].cpu().numpy(), a["test_logits_bag"].reshape(-1), atol=1e-4, rtol=0)
    np.testing.assert_allclose(logits[1][0].cpu().numpy(), a["test_logits_ins"].reshape(-1), atol=1e-4, rtol=0)
    np.testing.assert_allclose(B[0].cpu().numpy(), a["test_B"], atol=2e-5, rtol=1e-3)
    _, attn = m.forward_test(x, return_attn=True)
    np.testing.assert_allclose(attn[0].cpu().numpy(), a["test_attn"].reshape(-1), atol=2e-5, rtol=1e-4)
    m.train()
    feat, score = m.forward_teacher(x)
    assert feat.shape == (1, 2, 512) and score.shape == (1, n)
    np.testing.assert_allclose(feat[0].cpu().numpy(), a["teacher_feat"], atol=2e-5, rtol=1e-3)
    np.testing.assert_allclose(score[0].cpu().numpy(), a["teacher_score"].reshape(-1), atol=2e-5, rtol=1e-4)
    # student step on the fixture's teacher outputs and draws
    tscore = torch.from_numpy(a["teacher_score"]).to(DEV).view(1, -1)
    tfeat = torch.from_numpy(a["teacher_feat"]).to(DEV)
    lg, cl, ps, keep = m(x, tscore, tfeat, i=0, perm=a["perm"], ids_shuffle=a["ids_shuffle"])
    assert (ps, keep) == (n, int(a["keep"]))
    np.testing.assert_allclose(lg[0][0].detach().cpu().numpy(), a["logits_bag"].reshape(-1), atol=1e-4, rtol=0)
    np.testing.assert_allclose(lg[1][0].detach().cpu().numpy(), a["logits_ins"].reshape(-1), atol=1e-4, rtol=0)
    assert abs(float(cl) - float(a["cls_loss"])) < 2e-4
    loss = F.cross_entropy(0.5 * lg[0] + 0.5 * lg[1], torch.tensor([meta["label"]], device=DEV)) + meta["aux_alpha"] * cl
    assert abs(float(loss) - float(a["loss"])) < 2e-4
    loss.backward()
    pd = dict(m.named_parameters())
    keys = json.loads(str(a["grad_keys"]))
    for k, nrm in zip(keys, a["grad_norms"]):
        got = float(pd[k].grad.norm())
        assert abs(got - nrm) <= 5e-3 * nrm + 1e-7, (k, got, nrm)
    for k, exp in G.tagged(a, "grad").items():
        if "full" in exp:
            close(pd[k].grad, torch.from_numpy(exp["full"]).view_as(pd[k].grad), 5e-3, k)
    # pure (no mask, no merge)
    mp = build(synth.mhim_state(meta["seed"], input_dim=d, baseline="dsmil", merge_enable=False), input_dim=d,
               **{**cfgd, "merge_enable": False}).train()
    pl, aux, ps2, keep2 = mp.pure(x)
    assert (aux, ps2, keep2) == (0, n, n)
    np.testing.assert_allclose(pl[0][0].detach().cpu().numpy(), a["pure_logits_bag"].reshape(-1), atol=1e-4, rtol=0)
    np.testing.assert_allclose(pl[1][0].detach().cpu().numpy(), a["pure_logits_ins"].reshape(-1), atol=1e-4, rtol=0)


def test_dsmil_hooks_and_fused_trainer_vs_oracle():
    from mhim_mil_amd.engine import CommonMIL, FusedTrainer
    d, n, lr = 64, 800, 2e-4
    base = synth.mhim_state(23, input_dim=d, merge_k=5, baseline="dsmil")
    cfg = O.Cfg(**{**V2, "baseline": "dsmil"})
    k, n_sel, _ = O.mask_count(n, V2["mask_ratio_h"], V2["mask_ratio_hr"])
    # hook tuples (common_mil.py:14-48,56-68)
    s, t = build(base, input_dim=d, **V2).train(), build(base, input_dim=d, **V2).train()
    eng = CommonMIL(None)
    x = X(900, n, d)
    label = torch.tensor([1], device=DEV)
    perm, shuf = synth.permutation(50, k), synth.permutation(60, n - n_sel)
    args = types.SimpleNamespace(model="mhim", baseline="dsmil", aux_alpha=0.5)
    r = eng.forward_func(args, s, t, x, label, None, 1, 0, 0, 0, None, perm=perm, ids_shuffle=shuf)
    with torch.no_grad():
        o_feat, o_score = O.forward_teacher(x[0].cpu(), O.as_torch(base), cfg)
    o_lg, o_cl, _, o_keep, _ = O.forward_student(x[0].cpu(), O.as_torch(base), cfg, o_score, o_feat, perm=perm, ids_shuffle=shuf)
    assert r[0].shape == (1, 2) and (r[3], r[4]) == (n, o_keep)
    np.testing.assert_allclose(r[0][0].detach().cpu().numpy(), (0.5 * o_lg[0] + 0.5 * o_lg[1]).detach().numpy(), atol=1e-4, rtol=0)
    assert abs(float(r[2]) - float(o_cl)) < 2e-4
    s.eval()
    v, _ = eng.validate_func(types.SimpleNamespace(model="mhim", baseline="dsmil"), s, x, label, None, 1, 0, None)
    o_t = O.forward_test(x[0].cpu(), O.as_torch(base), cfg)[0]
    np.testing.assert_allclose(v[0].cpu().numpy(), (0.5 * o_t[0] + 0.5 * o_t[1]).numpy(), atol=1e-4, rtol=0)
    # two fused train steps
    s, t = build(base, input_dim=d, **V2).train(), build(base, input_dim=d, **V2).train()
    tr = FusedTrainer(s, t, lr=lr, aux_alpha=0.5, mm=0.999)
    stu, tea, opt = O.as_torch(base), O.as_torch(base), {}
    for step in range(2):
        xn = synth.bag(910 + step, n, d)
        perm, shuf = synth.permutation(51 + step, k), synth.permutation(61 + step, n - n_sel)
        stu, tea, opt, info = O.train_step(torch.from_numpy(xn), step % 2, stu, tea, opt, cfg, step + 1, perm=perm, ids_shuffle=shuf,
                                           aux_alpha=0.5, mm=0.999, lr=lr)
        logits, losses = tr.forward_backward(torch.from_numpy(xn).to(DEV), torch.tensor([step % 2], device=DEV),
                                             perm=torch.from_numpy(perm).to(DEV), ids_shuffle=torch.from_numpy(shuf).to(DEV))
        assert abs(float(losses[0]) - info["loss"]) < 3e-4, (step, float(losses[0]), info["loss"])
        for key, g in info["grads"].items():
            close(tr.flat.grad_views[key], g.view_as(tr.flat.grad_views[key]), 5e-3, f"step {step} grad {key}")
        tr.update()
    for tag, mdl, ref in (("stu", s, stu), ("tea", t, tea)):
        sd = mdl.state_dict()
        for key, exp in ref.items():
            err = (sd[key].detach().cpu().double() - exp.double().view_as(sd[key])).abs()
            tol = 0.1 * 2 * lr if tag == "stu" else 1e-6
            assert err.mean().item() <= 0.1 * tol + 1e-7, (tag, key, err.mean().item())
            assert (err > tol + 1e-7).double().mean().item() < 2e-3, (tag, key, err.max().item())
