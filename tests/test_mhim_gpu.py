"""Module-level parity of the HIP MHIM (mhim_mil_amd.mhim / engine) — GPU box only.

Checked against (a) the fixtures generated from the reference import (tests/golden) and (b) the CPU
oracle on the same seeded inputs, incl. BASELINE.json's config sizes (c1 N=512, c2 N=10 000, D=1024).
Tolerances: bag logits 1e-4 abs (north_star); gradients 1e-3 of the tensor's scale; index sets exact.
"""
import types

import numpy as np
import pytest
import torch

from mhim_mil_amd import synth
from oracle import mhim_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu
DEV = "cuda"

V2 = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True,
          merge_k=5, merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)


def build(sd, prec="auto", **kw):
    from mhim_mil_amd.mhim import MHIM
    m = MHIM(baseline="attn", n_classes=2, prec=prec, **kw)
    sd = dict(sd)
    if "merge.global_q_mm" in sd:
        sd["merge.global_q"] = sd["merge.global_q_mm"]
    missing, unexpected = m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV)
    if kw.get("merge_enable", True):
        m.merge.dropout = 0.0            # parity runs: the reference tests zero the MCA dropouts too
    return m


def X(seed, n, d):
    return torch.from_numpy(synth.bag(seed, n, d)).to(DEV).unsqueeze(0)


LOGIT_TOL = {"f32": 5e-6, "f16s": 1e-4, "bf16x3": 2e-5, "auto": 1e-4}


@pytest.mark.parametrize("prec", ["auto", "f32"])
@pytest.mark.parametrize("name", G.names("g1_abmil_eval"))
def test_forward_test_and_pure_eval_golden(name, prec):
    meta, a = G.load(name)
    m = build(synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_enable=False), prec, input_dim=meta["d"],
              act=meta["act"], da_act=meta["da_act"], merge_enable=False, dropout=0.25).eval()
    x = X(meta["xseed"], meta["n"], meta["d"])
    logits, attn = m.forward_test(x, return_attn=True)
    _, raw = m.forward_test(x, return_attn=True, no_norm=True)
    assert logits.shape == (1, 2) and attn.shape == (1, meta["n"])
    np.testing.assert_allclose(logits[0].cpu().numpy(), a["logits"], atol=LOGIT_TOL[prec], rtol=0)
    f = 1 if prec == "f32" else 100
    np.testing.assert_allclose(attn[0].cpu().numpy(), a["attn"], atol=1e-7 * f, rtol=3e-5 * f)
    np.testing.assert_allclose(raw[0].cpu().numpy(), a["raw"], atol=5e-6 * f, rtol=1e-5 * f)
    np.testing.assert_allclose(m.pure(x)[0].cpu().numpy(), a["logits"], atol=LOGIT_TOL[prec], rtol=0)


def _grad_close(got, exp, name, rtol=1e-3):
    g = got.detach().cpu().numpy().astype(np.float64)
    if "full" in exp:
        ref = exp["full"].astype(np.float64).reshape(g.shape)
        scale = np.abs(ref).max() + 1e-30
        np.testing.assert_allclose(g, ref, atol=rtol * scale, rtol=rtol, err_msg=name)
    else:
        flat = g.reshape(-1)
        samp = flat[::int(exp["stride"])][:exp["sample"].shape[0]]
        ref = exp["sample"].astype(np.float64)
        scale = np.abs(ref).max() + 1e-30
        np.testing.assert_allclose(samp, ref, atol=rtol * scale, rtol=rtol, err_msg=name)
        assert abs(np.linalg.norm(flat) - float(exp["norm"])) <= rtol * float(exp["norm"]), name


@pytest.mark.parametrize("prec", ["auto", "f32"])
@pytest.mark.parametrize("name", G.names("g2_abmil_train"))
def test_pure_train_grads_golden(name, prec):
    """mhim_pure train step: logits, CE and every parameter gradient vs the reference's autograd."""
    meta, a = G.load(name)
    m = build(synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_enable=False), prec, input_dim=meta["d"],
              act=meta["act"], da_act=meta["da_act"], merge_enable=False, dropout=0.0).train()
    logits, aux, ps, keep = m.pure(X(meta["xseed"], meta["n"], meta["d"]))
    assert (aux, ps, keep) == (0, meta["n"], meta["n"])
    loss = torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([meta["label"]], device=DEV))
    loss.backward()
    np.testing.assert_allclose(logits[0].detach().cpu().numpy(), a["logits"], atol=LOGIT_TOL[prec], rtol=0)
    assert abs(loss.item() - float(a["loss"])) < 1e-4
    pd = dict(m.named_parameters())
    for k, exp in G.tagged(a, "grad").items():
        _grad_close(pd[k].grad, exp, k)


@pytest.mark.parametrize("prec", ["auto", "f32"])
@pytest.mark.parametrize("name", G.names("g4_teacher"))
def test_forward_teacher_golden(name, prec):
    meta, a = G.load(name)
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"])
    sd = synth.spread_teacher(base) if meta["family"] == "tiefree" else base
    t = build(sd, prec, input_dim=meta["d"], **{**V2, "attn2score": meta["attn2score"]}).train()
    feat, score = t.forward_teacher(X(meta["xseed"], meta["n"], meta["d"]))
    assert feat.shape == (1, 512) and score.shape == (1, meta["n"])
    f = 1 if prec == "f32" else 60
    np.testing.assert_allclose(feat[0].cpu().numpy(), a["feat"], atol=5e-6 * f, rtol=1e-5 * f)
    np.testing.assert_allclose(score[0].cpu().numpy(), a["score"], atol=1e-6 * f, rtol=3e-5 * f)


@pytest.mark.parametrize("prec", ["auto", "f32"])
def test_student_step_golden(prec):
    """MHIM.forward (mask -> merge -> encoder -> losses) + backward vs the reference, RNG draws injected."""
    meta, a = G.load("g6_student_attn")
    m = build(synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"]), prec, input_dim=meta["d"],
              **{k: meta[k] for k in V2}).train()
    x = X(meta["xseed"], meta["n"], meta["d"])
    score = torch.from_numpy(a["teacher_score"]).to(DEV).view(1, -1)
    tfeat = torch.from_numpy(a["teacher_feat"]).to(DEV).view(1, -1)
    logits, cls_loss, ps, keep = m(x, score, tfeat, i=0, perm=a["perm"], ids_shuffle=a["ids_shuffle"])
    assert (ps, keep) == (int(a["ps"]), int(a["keep"]))
    np.testing.assert_allclose(logits[0].detach().cpu().numpy(), a["logits"], atol=LOGIT_TOL[prec], rtol=0)
    assert abs(cls_loss.item() - float(a["cls_loss"])) < 2e-4
    loss = torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([meta["label"]], device=DEV)) + meta["aux_alpha"] * cls_loss
    loss.backward()
    assert abs(loss.item() - float(a["loss"])) < 2e-4
    pd = dict(m.named_parameters())
    for k, exp in G.tagged(a, "grad").items():
        _grad_close(pd[k].grad, exp, k, rtol=2e-3 if k.startswith("merge.norm") else 1e-3)
    np.testing.assert_allclose(m.merge.global_q_mm.detach().cpu().numpy(), a["global_q_after"], atol=2e-6, rtol=1e-4)


def test_forward_func_golden():
    """CommonMIL.forward_func / validate_func tuples (common_mil.py:14-68)."""
    from mhim_mil_amd.engine import CommonMIL
    meta, a = G.load("g11_forward_func")
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"])
    cfg = {k: meta[k] for k in V2}
    s = build(base, input_dim=meta["d"], **cfg).train()
    t = build(synth.spread_teacher(base), input_dim=meta["d"], **cfg).train()
    eng = CommonMIL(None)
    x = X(meta["xseed"], meta["n"], meta["d"])
    label = torch.tensor([1], device=DEV)
    q0 = s.merge.global_q_mm.detach().clone()
    for aux in (0.5, 0.0):
        args = types.SimpleNamespace(model="mhim", baseline="attn", aux_alpha=aux)
        # with every keyword the reference trainer passes (base_engine.py:77-93): none of them may reach the model
        r = eng.forward_func(args, s, t, x, label, None, 1, 0, 0, 0, None, perm=a["perm"], ids_shuffle=a["ids_shuffle"],
                             loader=None, device=DEV, others={"epoch": 0}, idx=torch.tensor([3]), feat=None)
        s.merge.global_q_mm.data.copy_(q0)
        assert len(r) == 7 and r[1] is label
        np.testing.assert_allclose(r[0][0].detach().cpu().numpy(), a[f"logits_aux{aux}"], atol=1e-4, rtol=0)
        assert abs(float(r[2]) - float(a[f"auxloss_aux{aux}"])) < 2e-4
        assert [r[3], r[4], r[5], r[6]] == list(a[f"pn_kn_aux{aux}"])
    s.eval()
    lg, lab = eng.validate_func(types.SimpleNamespace(model="mhim", baseline="attn"), s, x, label, None, 1, 0, None)
    np.testing.assert_allclose(lg[0].cpu().numpy(), a["val_logits"], atol=1e-4, rtol=0)
    pure = build(synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_enable=False), input_dim=meta["d"], act="gelu",
                 da_act="relu", merge_enable=False, dropout=0.0).train()
    r = eng.forward_func(types.SimpleNamespace(model="mhim_pure", baseline="attn", aux_alpha=0.0), pure, None, x, label, None,
                         1, 0, 0, 0, None, loader=None, device=DEV, others=None, idx=None, feat=None)
    np.testing.assert_allclose(r[0][0].detach().cpu().numpy(), a["pure_logits"], atol=1e-4, rtol=0)
    assert [float(r[2]), r[3], r[4], r[5], r[6]] == list(a["pure_tuple"])


@pytest.mark.parametrize("prec", ["auto", "f32"])
def test_fused_trainer_three_steps_golden(prec):
    """FusedTrainer (flat buffers, head kernel, fused Adam+EMA) == the reference modules stepped by torch Adam."""
    from mhim_mil_amd.engine import FusedTrainer
    meta, a = G.load("g10_train_steps")
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"])
    cfg = {k: meta[k] for k in V2}
    s = build(base, prec, input_dim=meta["d"], **cfg).train()
    t = build(synth.spread_teacher(base), prec, input_dim=meta["d"], **cfg).train()
    tr = FusedTrainer(s, t, lr=meta["lr"], weight_decay=meta["wd"], mm=meta["mm"], aux_alpha=meta["aux_alpha"])
    for step in range(meta["steps"]):
        x = X(meta["xseed0"] + step, meta["n"], meta["d"])
        label = torch.tensor([step % 2], device=DEV)
        logits, losses = tr.train_step(x, label, perm=torch.from_numpy(a[f"perm{step}"]).to(DEV),
                                       ids_shuffle=torch.from_numpy(a[f"shuf{step}"]).to(DEV))
        assert abs(float(losses[0]) - float(a["losses"][step])) < 3e-4, (step, float(losses[0]), a["losses"][step])
    # Adam's first steps move each weight by ~lr regardless of gradient scale, so parameters are compared on the
    # scale of the update (3 steps * lr = 6e-4): 5 % of that.
    for tag, mdl in (("stu", s), ("tea", t)):
        sd = mdl.state_dict()
        for k, exp in G.tagged(a, tag).items():
            got = sd[k].detach().cpu().numpy().astype(np.float64)
            if "full" in exp:
                np.testing.assert_allclose(got, exp["full"].reshape(got.shape), atol=3e-5, rtol=0, err_msg=f"{tag}:{k}")
            else:
                samp = got.reshape(-1)[::int(exp["stride"])][:exp["sample"].shape[0]]
                np.testing.assert_allclose(samp, exp["sample"], atol=3e-5, rtol=0, err_msg=f"{tag}:{k}")


@pytest.mark.parametrize("n,d,prec", [(512, 1024, "auto"), (10000, 1024, "auto"), (10000, 1024, "bf16x3"), (2000, 1536, "auto")])
def test_full_size_step_vs_oracle(n, d, prec):
    """BASELINE configs c1/c2 (and a D=1536 bag): teacher + select + student + grads against the CPU oracle."""
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    tsd = synth.spread_teacher(base)
    cfg = O.Cfg(**V2)
    x = synth.bag(1000 * 2 + n, n, d)
    xt = torch.from_numpy(x)
    k, n_sel, _ = O.mask_count(n, V2["mask_ratio_h"], V2["mask_ratio_hr"])
    perm = synth.permutation(3, k)
    ids_shuffle = synth.permutation(4, n - n_sel)
    # oracle
    torch.set_num_threads(8)
    po = O.as_torch(base)
    for kk, v in po.items():
        v.requires_grad_(kk not in O.TRAINABLE_EXCLUDE)
    with torch.no_grad():
        o_feat, o_score = O.forward_teacher(xt, O.as_torch(tsd), cfg)
    o_logits, o_cl, _, o_keep, ex = O.forward_student(xt, po, cfg, o_score, o_feat, perm=perm, ids_shuffle=ids_shuffle)
    (O.cross_entropy(o_logits, 1) + 0.5 * o_cl).backward()
    # device
    s = build(base, prec, input_dim=d, **V2).train()
    t = build(tsd, prec, input_dim=d, **V2).train()
    xd = torch.from_numpy(x).to(DEV).unsqueeze(0)
    feat, score = t.forward_teacher(xd)
    np.testing.assert_allclose(feat[0].cpu().numpy(), o_feat.numpy(), atol=2e-4, rtol=1e-3)
    np.testing.assert_allclose(score[0].cpu().numpy(), o_score.numpy(), atol=1e-4, rtol=2e-3)
    # select on IDENTICAL inputs (the oracle's score) must give identical index sets
    len_keep, mask_ids = s.get_mask(n, 0, o_score.to(DEV).view(1, -1), perm=perm)
    assert len_keep == ex["len_keep_mask"]
    assert np.array_equal(mask_ids[0].cpu().numpy(), ex["mask_ids"])
    logits, cls_loss, ps, keep = s(xd, o_score.to(DEV).view(1, -1), o_feat.to(DEV).view(1, -1), i=0, perm=perm,
                                   ids_shuffle=ids_shuffle)
    assert keep == o_keep and ps == n
    np.testing.assert_allclose(logits[0].detach().cpu().numpy(), o_logits.detach().numpy(), atol=1e-4, rtol=0)
    assert abs(cls_loss.item() - o_cl.item()) < 3e-4
    (torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([1], device=DEV)) + 0.5 * cls_loss).backward()
    for kk, p in s.named_parameters():
        if p.grad is None:
            continue
        ref = po[kk].grad.numpy()
        scale = np.abs(ref).max() + 1e-30
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2e-3 * scale, rtol=2e-3, err_msg=kk)


def test_hashed_dropout_forward_backward_consistent():
    """Dropout p=0.25 from the counter-based stream: the mask the forward applied is the one the backward uses
    (checked by handing the recovered mask to the oracle)."""
    n, d = 700, 64
    base = synth.mhim_state(7, input_dim=d, merge_enable=False)
    m = build(base, "f32", input_dim=d, act="relu", da_act="gelu", merge_enable=False, dropout=0.25).train()
    x = X(55, n, d)
    torch.manual_seed(1234)
    logits, _, _, _ = m.pure(x)
    logits.sum().backward()
    # recover the mask: rerun the feature kernel with the same seed and compare against the undropped activation
    seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + m._step * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    Hd = m._feature(x[0], None, 0.25, seed)
    H0 = m._feature(x[0], None, 0.0, 0)
    mask = ((Hd != 0) | (H0 == 0)).cpu()
    keep_rate = (Hd != 0).sum().item() / max(1, (H0 != 0).sum().item())
    assert abs(keep_rate - 0.75) < 0.01
    po = O.as_torch(base)
    for v in po.values():
        v.requires_grad_(True)
    lo = O.pure(x[0].cpu(), po, O.Cfg(act="relu", da_act="gelu", merge_enable=False, dropout=0.25), drop_mask=mask)
    lo.sum().backward()
    np.testing.assert_allclose(logits[0].detach().cpu().numpy(), lo.detach().numpy(), atol=2e-5, rtol=0)
    for kk, p in m.named_parameters():
        ref = po[kk].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=1e-4 * (np.abs(ref).max() + 1e-30), rtol=1e-3, err_msg=kk)


def test_rejects_cpu_tensors_and_missing_paths():
    from mhim_mil_amd._lib import MhimxError
    from mhim_mil_amd.mhim import MHIM
    m = MHIM(input_dim=64, baseline="attn", merge_k=1, merge_ratio=0.9, mask_ratio_h=0.03, mask_ratio_hr=0.5).to(DEV)
    with pytest.raises(MhimxError):
        m.forward_test(torch.zeros(1, 10, 64))
    with pytest.raises(NotImplementedError):
        MHIM(input_dim=64, baseline="clam")                 # not one of the reference's three MHIM baselines
    extra = MHIM(input_dim=64, baseline="attn", attn_layer=0, select_mask=False)      # tolerated kwargs
    assert isinstance(extra, torch.nn.Module)


def test_mm_schedule_under_graph_replay():
    """EMA-momentum schedule (`mm_sche`, base_engine.py:160-161) as a device table indexed by the device step counter:
    replaying ONE captured graph applies a different momentum on every step, equal to the eager trainer's and the oracle's
    EMA formula."""
    from mhim_mil_amd.engine import FusedTrainer, cosine_scheduler
    d, n = 64, 700
    base = synth.mhim_state(3, input_dim=d, merge_k=5)
    sche = cosine_scheduler(0.99, 1.0, epochs=2, niter_per_ep=3, start_warmup_value=1.0)         # 6 values, 0.99 -> ~1
    x = X(321, n, d)
    label = torch.tensor([1], device=DEV)
    k, n_sel, _ = O.mask_count(n, V2["mask_ratio_h"], V2["mask_ratio_hr"])
    perm = torch.from_numpy(synth.permutation(5, k)).to(DEV)
    shuf = torch.from_numpy(synth.permutation(6, n - n_sel)).to(DEV)

    def run(graph):
        s = build(base, input_dim=d, **V2).train()
        t = build(synth.spread_teacher(base), input_dim=d, **V2).train()
        tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.5, mm_sche=sche)
        tea0 = tr.flat.teacher.clone()
        if graph:
            g = tr.capture(x, label, warmup=0, perm=perm, ids_shuffle=shuf)          # capture runs no eager step first
            for _ in range(4):
                g.replay()
        else:
            for _ in range(4):
                tr.train_step(x, label, perm=perm, ids_shuffle=shuf)
        torch.cuda.synchronize()
        return tr, tea0

    tr_e, tea0 = run(False)
    tr_g, _ = run(True)
    np.testing.assert_allclose(tr_g.flat.teacher.cpu().numpy(), tr_e.flat.teacher.cpu().numpy(), atol=2e-6, rtol=0)
    np.testing.assert_allclose(tr_g.flat.student.cpu().numpy(), tr_e.flat.student.cpu().numpy(), atol=2e-6, rtol=0)
    # with a constant momentum of 0.5 the teacher would have moved most of the way to the student; the schedule (~0.99)
    # keeps it near its start
    moved = (tr_e.flat.teacher - tea0).abs().max().item()
    gap = (tr_e.flat.student - tea0).abs().max().item()
    assert moved < 0.1 * gap


def test_forward_eval_mode_golden():
    """MHIM.forward with the module in eval mode (the reference's class allows it, mhim.py:318-378): fixture from the reference import."""
    meta, a = G.load("g16_student_eval_attn")
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"])
    s = build(base, input_dim=meta["d"], **{k: meta[k] for k in V2}).eval()
    x = X(meta["xseed"], meta["n"], meta["d"])
    q0 = s.merge.global_q_mm.detach().clone()
    logits, cl, ps, keep = s(x, torch.from_numpy(a["teacher_score"]).to(DEV).view(1, -1), torch.from_numpy(a["teacher_feat"]).to(DEV).view(1, -1),
                             i=0, perm=a["perm"])
    np.testing.assert_allclose(logits[0].cpu().numpy(), a["logits"], atol=1e-4, rtol=0)
    assert abs(float(cl) - float(a["cls_loss"])) < 2e-4 and ps == int(a["ps"]) and keep == int(a["keep"])
    assert torch.equal(q0, s.merge.global_q_mm.detach())             # no EMA of the global queries in eval mode
