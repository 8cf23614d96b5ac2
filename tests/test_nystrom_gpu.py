"""Parity of the TransMIL / Nystrom path (BASELINE config c3) — GPU box only.

Three layers: (1) every streaming primitive and the nn-GEMM against a plain torch fp32 statement of the same op,
forward and backward; (2) the encoder against the fixtures generated from the reference import (g7/g8/g9);
(3) the student's gradients element-wise against the CPU oracle's autograd.
Tolerances: logits 1e-4 abs (north_star); attention rows 1e-3 rel; gradients 2e-3 of the tensor's scale.
"""
import json

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


def NY():
    from mhim_mil_amd import nystrom
    return nystrom


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def close(got, ref, rtol=2e-3, what=""):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


# ------------------------------------------------------------------------------------------------ primitives
@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
@pytest.mark.parametrize("shape", [(300, 256, 64), (1, 256, 256), (256, 64, 1280), (130, 72, 36)])
def test_heads_matmul_modes(mode, shape):
    ny = NY()
    M, N, K = shape
    a = rnd(3, *((M, K) if mode != "tn" else (K, M)), seed=1).requires_grad_()
    b = rnd(3, *((N, K) if mode == "nt" else (K, N)), seed=2).requires_grad_()
    ref = {"nt": lambda: a @ b.transpose(1, 2), "nn": lambda: a @ b, "tn": lambda: a.transpose(1, 2) @ b}[mode]()
    bat = lambda t: (0, t.shape[1] * t.shape[2], t.shape[2], t.shape[1], t.shape[2])
    got = ny.heads_mm(a, b, mode, bat(a), bat(b), (3, M, N), (0, M * N, N, M, N), 3)
    close(got, ref, 1e-4, "fwd")
    w = rnd(3, M, N, seed=3)
    ga, gb = torch.autograd.grad((got * w).sum(), (a, b))
    ra, rb = torch.autograd.grad((ref * w).sum(), (a, b))
    close(ga, ra, 1e-4, "dA")
    close(gb, rb, 1e-4, "dB")


def test_heads_matmul_packed_operands():
    """Heads addressed in place inside a packed [T, 3*512] buffer (64-column groups), output into [T, (h d)]."""
    ny = NY()
    T, m = 512, 256
    qkv = rnd(T, 1536, seed=4).requires_grad_()
    a1 = rnd(8, T, m, seed=5).requires_grad_()
    w2 = rnd(8, m, 64, seed=6).requires_grad_()
    out = ny.heads_mm(a1, w2, "nn", (0, T * m, m, T, m), (0, m * 64, 64, m, 64), (T, 512), (0, 64, 512, T, 64))
    ref = torch.einsum("htm,hmd->thd", a1, w2).reshape(T, 512)
    close(out, ref, 1e-4)
    s1 = ny.heads_mm(qkv, qkv, "nt", (0, 64, 1536, T, 64), (512, 64, 1536, T, 64), (8, T, T), (0, T * T, T, T, T))
    q = qkv[:, :512].reshape(T, 8, 64).permute(1, 0, 2)
    k = qkv[:, 512:1024].reshape(T, 8, 64).permute(1, 0, 2)
    close(s1, q @ k.transpose(1, 2), 1e-4)
    w = rnd(8, T, T, seed=7)
    g, = torch.autograd.grad((s1 * w).sum(), qkv)
    r, = torch.autograd.grad(((q @ k.transpose(1, 2)) * w).sum(), qkv)
    close(g, r, 1e-4, "d qkv")


@pytest.mark.parametrize("R,Lr,alpha", [(50, 256, 0.125), (7, 1280, 1.0), (2048, 256, 0.125), (3, 50176, 0.125)])
def test_softmax_rows(R, Lr, alpha):
    ny = NY()
    x = rnd(R, Lr, seed=8, scale=3.0).requires_grad_()
    y = ny.Softmax.apply(x, alpha)
    ref = torch.softmax(x * alpha, -1)
    close(y, ref, 1e-5)
    w = rnd(R, Lr, seed=9)
    g, = torch.autograd.grad((y * w).sum(), x)
    r, = torch.autograd.grad((ref * w).sum(), x)
    close(g, r, 1e-4)


@pytest.mark.parametrize("T,l", [(512, 2), (1280, 5), (256, 1)])
def test_landmarks(T, l):
    ny = NY()
    x = rnd(T, 1536, seed=10).requires_grad_()
    y = ny.Landmarks.apply(x, l)
    ref = x[:, :1024].reshape(T // l, l, 1024).mean(1)
    close(y, ref, 1e-5)
    w = rnd(T // l, 1024, seed=11)
    g, = torch.autograd.grad((y * w).sum(), x)
    r, = torch.autograd.grad((ref * w).sum(), x)
    close(g, r, 1e-5)


def test_pinv_matches_oracle_and_autograd():
    ny = NY()
    # softmax rows all sum to 1: the arg-max ROW of the init scaling is rounding noise, but a constant added to a whole
    # row of d a vanishes in the softmax backward — so the gradient is compared where it is well defined, at the logits
    x0 = rnd(8, 256, 256, seed=12, scale=2.0)
    x = x0.clone().requires_grad_()
    z = ny._pinv(ny.Softmax.apply(x, 1.0))
    xc = x0.cpu().clone().requires_grad_()
    zr = O.pinv_iter(torch.softmax(xc, -1))
    close(z, zr, 2e-4, "pinv")
    w = rnd(8, 256, 256, seed=13)
    g, = torch.autograd.grad((z * w).sum(), x)
    r, = torch.autograd.grad((zr * w.cpu()).sum(), xc)
    close(g, r, 2e-3, "d pinv")


def test_pinv_init_bwd_through_maxima():
    ny = NY()
    a0 = rnd(8, 256, 256, seed=14)          # signed entries, distinct row / column abs-sums (no arg-max ties)
    a = a0.clone().requires_grad_()
    z = ny.PinvInit.apply(a)
    ac = a0.cpu().double().requires_grad_()
    ab = ac.abs()
    zr = ac.transpose(1, 2) / (ab.sum(-1).max() * ab.sum(-2).max())
    close(z, zr, 1e-5)
    w = rnd(8, 256, 256, seed=15)
    g, = torch.autograd.grad((z * w).sum(), a)
    r, = torch.autograd.grad((zr * w.cpu().double()).sum(), ac)
    close(g, r, 1e-4)


@pytest.mark.parametrize("T", [256, 1280])
def test_resconv(T):
    ny = NY()
    qkv = rnd(T, 1536, seed=16).requires_grad_()
    w = rnd(8, 1, 33, 1, seed=17, scale=0.2).requires_grad_()
    y = ny.ResConv.apply(qkv, w)
    v = qkv[:, 1024:].reshape(T, 8, 64).permute(1, 0, 2).unsqueeze(0)          # [1,h,T,d]
    ref = F.conv2d(v, w, padding=(16, 0), groups=8)[0].permute(1, 0, 2).reshape(T, 512)
    close(y, ref, 1e-5)
    ww = rnd(T, 512, seed=18)
    g = torch.autograd.grad((y * ww).sum(), (qkv, w))
    r = torch.autograd.grad((ref * ww).sum(), (qkv, w))
    close(g[0], r[0], 1e-5, "d v")
    close(g[1], r[1], 1e-4, "d w")


@pytest.mark.parametrize("N", [47, 600, 1500])
def test_ppeg(N):
    ny = NY()
    sd = synth.mhim_state(5, input_dim=64, baseline="selfattn", merge_enable=False)
    pre = "online_encoder.pos_embedding."
    names = ["proj.weight", "proj1.weight", "proj2.weight", "proj.bias", "proj1.bias", "proj2.bias"]
    x0 = rnd(N, 512, seed=19)
    x = x0.clone().requires_grad_()
    ps = [torch.as_tensor(sd[pre + n]).to(DEV).requires_grad_() for n in names]
    y = ny.PPEG.apply(x, *ps)
    pc = {pre + n: torch.as_tensor(sd[pre + n]).clone().requires_grad_() for n in names}
    xc = x0.cpu().clone().requires_grad_()
    ref = O.ppeg(xc, pc, pre)
    close(y, ref, 1e-5)
    w = rnd(N, 512, seed=20)
    g = torch.autograd.grad((y * w).sum(), [x] + ps)
    r = torch.autograd.grad((ref * w.cpu()).sum(), [xc] + [pc[pre + n] for n in names])
    for gi, ri, nm in zip(g, r, ["x"] + names):
        close(gi, ri, 1e-4, nm)


def test_layernorm_and_linear_fn():
    ny = NY()
    x = rnd(300, 512, seed=21).requires_grad_()
    w, b = rnd(512, seed=22).requires_grad_(), rnd(512, seed=23).requires_grad_()
    y = ny.LayerNorm.apply(x, w, b)
    ref = F.layer_norm(x, (512,), w, b)
    close(y, ref, 1e-5)
    ww = rnd(300, 512, seed=24)
    for gi, ri in zip(torch.autograd.grad((y * ww).sum(), (x, w, b)), torch.autograd.grad((ref * ww).sum(), (x, w, b))):
        close(gi, ri, 1e-4)
    W, bb = rnd(384, 512, seed=25, scale=0.05).requires_grad_(), rnd(384, seed=26).requires_grad_()
    y = ny.Linear.apply(x, W, bb, 0.0, 0, None)
    ref = F.linear(x, W, bb)
    close(y, ref, 1e-4)
    ww = rnd(300, 384, seed=27)
    for gi, ri in zip(torch.autograd.grad((y * ww).sum(), (x, W, bb)), torch.autograd.grad((ref * ww).sum(), (x, W, bb))):
        close(gi, ri, 1e-4)
    # dropout: the backward re-applies exactly the forward's mask
    y = ny.Linear.apply(x, W, bb, 0.3, 99, None)
    keep = (y != 0).float()
    assert 0.6 < keep.mean().item() < 0.8
    g, = torch.autograd.grad((y * ww).sum(), x)
    r, = torch.autograd.grad((F.linear(x, W, bb) * keep / 0.7 * ww).sum(), x)
    close(g, r, 1e-4)


# ------------------------------------------------------------------------------------------------ golden fixtures
def build(sd, prec="auto", **kw):
    from mhim_mil_amd.mhim import MHIM
    m = MHIM(baseline="selfattn", n_classes=2, prec=prec, **kw)
    sd = dict(sd)
    if "merge.global_q_mm" in sd:
        sd["merge.global_q"] = sd["merge.global_q_mm"]
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV)
    if kw.get("merge_enable", True):
        m.merge.dropout = 0.0
    m.online_encoder.layer1.attn.dropout = 0.0           # parity runs: the fixtures were made with the aux dropouts zeroed
    m.online_encoder.layer2.attn.dropout = 0.0
    return m


def X(seed, n, d):
    return torch.from_numpy(synth.bag(seed, n, d)).to(DEV).unsqueeze(0)


@pytest.mark.parametrize("name", G.names("g7_nystrom"))
def test_g7_nystrom(name):
    meta, a = G.load(name)
    m = build(synth.mhim_state(meta["seed"], input_dim=64, baseline="selfattn", merge_enable=False), input_dim=64, merge_enable=False).eval()
    att = m.online_encoder.layer1.attn
    x = torch.from_numpy((synth.normal(meta["xseed"], (meta["n"], meta["dim"])) * 0.5).astype(np.float32)).to(DEV)
    with torch.no_grad():
        out, attn, v = att(x, return_attn=True)
        _, attn_raw, _ = att(x, return_attn=True, no_norm=True)
    out = out.cpu().numpy()
    np.testing.assert_allclose(out[:8], a["out_head"], atol=2e-5, rtol=2e-4)
    np.testing.assert_allclose(out[-8:], a["out_tail"], atol=2e-5, rtol=2e-4)
    np.testing.assert_allclose(out.sum(0), a["out_sum"], atol=2e-3, rtol=2e-4)
    np.testing.assert_allclose(attn.cpu().numpy(), a["attn"], atol=1e-6, rtol=2e-3)
    np.testing.assert_allclose(attn_raw.cpu().numpy(), a["attn_raw"], atol=2e-3, rtol=5e-3)
    vt = v.reshape(v.shape[0], 8, 64).permute(1, 0, 2)[:, -4:].cpu().numpy()
    np.testing.assert_allclose(vt, a["v_tail"], atol=2e-5, rtol=1e-4)     # bf16x3 GEMM: ~2^-16 of the row scale


@pytest.mark.parametrize("name", G.names("g8_sattention"))
def test_g8_sattention(name):
    meta, a = G.load(name)
    m = build(synth.mhim_state(meta["seed"], input_dim=meta["d"], baseline="selfattn", merge_enable=False),
              input_dim=meta["d"], act=meta["act"], merge_enable=False).eval()
    x = X(meta["xseed"], meta["n"], meta["d"])
    logits, attn = m.forward_test(x, return_attn=True)
    assert logits.shape == (1, 2) and attn[0].shape == (1, 8, meta["n"])
    np.testing.assert_allclose(logits[0].cpu().numpy(), a["logits"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(attn[0][0].cpu().numpy(), a["attn1"], atol=1e-6, rtol=2e-3)
    np.testing.assert_allclose(attn[1][0].cpu().numpy(), a["attn2"], atol=1e-6, rtol=2e-3)
    np.testing.assert_allclose(m.forward_test(x)[0].cpu().numpy(), a["logits"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(m.pure(x)[0].cpu().numpy(), a["logits"], atol=1e-4, rtol=0)


@pytest.mark.parametrize("name", G.names("g9_transmil_teacher"))
def test_g9_transmil_teacher(name):
    meta, a = G.load(name)
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"], baseline="selfattn")
    m = build(synth.spread_teacher(base), input_dim=meta["d"], **{**V2, "attn2score": meta["attn2score"]}).train()
    feat, score = m.forward_teacher(X(meta["xseed"], meta["n"], meta["d"]))
    np.testing.assert_allclose(feat[0].cpu().numpy(), a["feat"], atol=1e-4, rtol=1e-4)
    s = score[0].cpu().numpy()
    np.testing.assert_allclose(s, a["score"], atol=5e-6, rtol=2e-3)


def test_g9_transmil_student_golden_and_oracle_grads():
    meta, a = G.load("g9_transmil_student")
    sd = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"], baseline="selfattn")
    cfgd = {k: meta[k] for k in V2 if k in meta}
    m = build(sd, input_dim=meta["d"], **cfgd).train()
    x = X(meta["xseed"], meta["n"], meta["d"])
    score = torch.from_numpy(a["teacher_score"]).to(DEV)
    tfeat = torch.from_numpy(a["teacher_feat"]).to(DEV).view(1, -1)
    logits, cl, ps, keep = m(x, score, tfeat, perm=a["perm"], ids_shuffle=a["ids_shuffle"])
    assert keep == int(a["keep"]) and ps == meta["n"]
    np.testing.assert_allclose(logits[0].detach().cpu().numpy(), a["logits"], atol=1e-4, rtol=0)
    assert abs(cl.item() - float(a["cls_loss"])) < 1e-4
    loss = F.cross_entropy(logits.view(1, -1), torch.tensor([meta["label"]], device=DEV)) + meta["aux_alpha"] * cl
    loss.backward()
    pd = dict(m.named_parameters())
    keys = json.loads(str(a["grad_keys"]))
    for k, n in zip(keys, a["grad_norms"]):
        got = float(pd[k].grad.norm())
        assert abs(got - n) <= 5e-3 * n + 1e-7, (k, got, n)
    # element-wise against the oracle's autograd on the same inputs
    p = O.as_torch(sd)
    for k, v in p.items():
        v.requires_grad_(k not in O.TRAINABLE_EXCLUDE)
    cfg = O.Cfg(**{**cfgd, "baseline": "selfattn"})
    lo, clo, _, _, _ = O.forward_student(x[0].cpu(), p, cfg, a["teacher_score"], torch.from_numpy(a["teacher_feat"]),
                                         perm=a["perm"], ids_shuffle=a["ids_shuffle"])
    (O.cross_entropy(lo, meta["label"]) + meta["aux_alpha"] * clo).backward()
    for k in keys:
        close(pd[k].grad, p[k].grad.view_as(pd[k].grad), 5e-3, k)


def test_teacher_vote_mask_matches_oracle():
    """attn2score=False: per-head attention [1,h,N] -> vote fusion (masking.py:49-59) -> same index set as the oracle."""
    sd = synth.mhim_state(11, input_dim=64, merge_k=3, baseline="selfattn")
    kw = {**V2, "attn2score": False, "merge_k": 3}
    m = build(synth.spread_teacher(sd), input_dim=64, **kw).train()
    x = X(77, 700, 64)
    _, attn = m.forward_teacher(x)
    assert attn.shape == (1, 8, 700)
    k = int(np.ceil(700 * 0.03 / 0.5))
    perm = synth.permutation(5, k)
    lk, ids = m.get_mask(700, 0, attn, perm=perm)
    lko, idso = O.get_mask(700, attn[0].cpu().numpy(), mask_ratio_h=0.03, mask_ratio_hr=0.5, perms=(None, None, perm))
    assert lk == lko
    np.testing.assert_array_equal(ids[0].cpu().numpy(), idso)


# ------------------------------------------------------------------------------------------------ trainer + full size
def test_fused_trainer_selfattn_two_steps_vs_oracle():
    """FusedTrainer on the TransMIL student (autograd into the flat gradient buffer, head kernel, fused Adam + EMA)
    against the oracle's train_step; Adam's first steps move weights by ~lr, so parameters are compared on that scale."""
    from mhim_mil_amd.engine import FusedTrainer
    n, d, lr = 900, 64, 2e-4
    base = synth.mhim_state(21, input_dim=d, merge_k=5, baseline="selfattn")
    cfg = O.Cfg(**{**V2, "baseline": "selfattn"})
    s = build(base, input_dim=d, **V2).train()
    t = build(synth.spread_teacher(base), input_dim=d, **V2).train()
    tr = FusedTrainer(s, t, lr=lr, aux_alpha=0.5, mm=0.999)
    stu, tea, opt = O.as_torch(base), O.as_torch(synth.spread_teacher(base)), {}
    k, n_sel, _ = O.mask_count(n, V2["mask_ratio_h"], V2["mask_ratio_hr"])
    for step in range(2):
        xn = synth.bag(500 + step, n, d)
        perm, shuf = synth.permutation(30 + step, k), synth.permutation(40 + step, n - n_sel)
        stu, tea, opt, info = O.train_step(torch.from_numpy(xn), step % 2, stu, tea, opt, cfg, step + 1, perm=perm, ids_shuffle=shuf,
                                           aux_alpha=0.5, mm=0.999, lr=lr)
        logits, losses = tr.forward_backward(torch.from_numpy(xn).to(DEV), torch.tensor([step % 2], device=DEV),
                                             perm=torch.from_numpy(perm).to(DEV), ids_shuffle=torch.from_numpy(shuf).to(DEV))
        assert abs(float(losses[0]) - info["loss"]) < 3e-4, (step, float(losses[0]), info["loss"])
        np.testing.assert_allclose(logits.cpu().numpy(), info["logits"].numpy(), atol=1e-4, rtol=0)
        for key, g in info["grads"].items():                 # the flat gradient buffer before the update
            close(tr.flat.grad_views[key], g.view_as(tr.flat.grad_views[key]), 5e-3, f"step {step} grad {key}")
        tr.update()
    # Adam normalises every element to a ~lr move, so elements whose gradient is rounding noise (|g| ~ 1e-8 of the tensor's
    # scale) may legitimately move the other way: bound the mean error tightly and the count of such outliers
    for tag, mdl, ref in (("stu", s, stu), ("tea", t, tea)):
        sd = mdl.state_dict()
        for key, exp in ref.items():
            err = (sd[key].detach().cpu().double() - exp.double().view_as(sd[key])).abs()
            tol = 0.1 * 2 * lr if tag == "stu" else 1e-6          # EMA teacher: (1 - mm) of the student's moves + fp32 ulps
            assert err.mean().item() <= 0.1 * tol + 1e-7, (tag, key, err.mean().item())
            assert (err > tol + 1e-7).double().mean().item() < 2e-3, (tag, key, err.max().item())


def test_c3_size_teacher_and_student_forward_vs_oracle():
    """BASELINE config c3: MHIM(TransMIL) on one N=50 000, D=1024 bag — teacher score / feature, student logits and EVERY parameter
    gradient of (logits.sum() + cls_loss) against the CPU oracle's autograd (~25 s of CPU work)."""
    n, d = 50000, 1024
    base = synth.mhim_state(7, input_dim=d, merge_k=5, baseline="selfattn")
    tsd = synth.spread_teacher(base)
    cfg = O.Cfg(**{**V2, "baseline": "selfattn"})
    xn = synth.bag(9, n, d)
    k, n_sel, _ = O.mask_count(n, V2["mask_ratio_h"], V2["mask_ratio_hr"])
    perm, shuf = synth.permutation(3, k), synth.permutation(4, n - n_sel)
    torch.set_num_threads(16)
    with torch.no_grad():
        o_feat, o_score = O.forward_teacher(torch.from_numpy(xn), O.as_torch(tsd), cfg)
    t = build(tsd, input_dim=d, **V2).train()
    s = build(base, input_dim=d, **V2).train()
    x = torch.from_numpy(xn).to(DEV)
    feat, score = t.forward_teacher(x)
    np.testing.assert_allclose(feat[0].cpu().numpy(), o_feat.numpy(), atol=2e-4, rtol=1e-3)
    np.testing.assert_allclose(score[0].cpu().numpy(), o_score.numpy(), atol=1e-5, rtol=5e-3)
    # student on the ORACLE's teacher outputs (identical inputs => identical index sets)
    ostu = {k_: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k_, v in O.as_torch(base).items()}
    o_logits, o_cl, _, o_keep, ex = O.forward_student(torch.from_numpy(xn), ostu, cfg, o_score, o_feat, perm=perm, ids_shuffle=shuf)
    (o_logits.sum() + o_cl).backward()
    o_logits, o_cl = o_logits.detach(), o_cl.detach()
    lk, ids = s.get_mask(n, 0, o_score.to(DEV).view(1, -1), perm=perm)
    assert lk == ex["len_keep_mask"]
    np.testing.assert_array_equal(ids[0].cpu().numpy(), ex["mask_ids"])
    logits, cl, ps, keep = s(x, o_score.to(DEV).view(1, -1), o_feat.to(DEV).view(1, -1), perm=perm, ids_shuffle=shuf)
    assert keep == o_keep
    np.testing.assert_allclose(logits[0].detach().cpu().numpy(), o_logits.numpy(), atol=1e-4, rtol=0)
    assert abs(float(cl) - float(o_cl)) < 2e-4
    (logits.sum() + cl).backward()
    worst = {}
    for nme, p in s.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), nme
        ref = ostu[nme].grad
        if ref is None:                                              # (a parameter the loss does not reach in the reference either)
            assert float(p.grad.abs().max()) == 0.0, nme
            continue
        g, r = p.grad.detach().cpu().double().reshape(-1), ref.double().reshape(-1)
        scale = float(r.abs().max()) + 1e-30
        worst[nme] = float((g - r).abs().max()) / scale
    # 3-term bf16 products (~2^-16) through two Nystrom layers with a 6-step pseudo-inverse each, 48 500 tokens: measured <= 2e-3 of a
    # tensor's largest gradient entry
    bad = {k_: v for k_, v in worst.items() if v > 5e-3}
    assert not bad, bad


@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
@pytest.mark.parametrize("n,B", [(256, 8), (64, 40), (512, 2)])
def test_bmm_affine_vs_fp64(mode, n, B):
    """mhimx_bmm_affine (csrc/small_bmm.hip): ident * I + alpha * op(a, b) for batches of small square matrices."""
    from mhim_mil_amd import nystrom as NY
    g = torch.Generator().manual_seed(n + B)
    a = torch.randn((B, n, n), generator=g).to(DEV)
    b = torch.randn((B, n, n), generator=g).to(DEV)
    out = NY._bmm_affine(mode, a, b, torch.empty_like(a), -0.75, 13.0)
    ad, bd = a.double(), b.double()
    prod = {"nt": ad @ bd.transpose(1, 2), "nn": ad @ bd, "tn": ad.transpose(1, 2) @ bd}[mode]
    ref = 13.0 * torch.eye(n, device=DEV, dtype=torch.float64) - 0.75 * prod
    assert float((out.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())


def test_gemm_batched_takes_the_small_kernel_and_agrees():
    """mhimx_gemm_batched on pseudo-inverse-sized operands (8 x 256^3) dispatches to small_bmm.hip: same numbers as fp64."""
    from mhim_mil_amd import nystrom as NY
    g = torch.Generator().manual_seed(1)
    a = torch.randn((8, 256, 256), generator=g).to(DEV).requires_grad_(True)
    b = torch.randn((8, 256, 256), generator=g).to(DEV).requires_grad_(True)
    bat = (0, 256 * 256, 256, 256, 256)
    c = NY.heads_mm(a, b, "nn", bat, bat, (8, 256, 256), bat, 8)
    w = torch.randn((8, 256, 256), generator=g).to(DEV)
    (c * w).sum().backward()
    ad, bd, wd = a.detach().double(), b.detach().double(), w.double()
    for got, ref in ((c.detach(), ad @ bd), (a.grad, wd @ bd.transpose(1, 2)), (b.grad, ad.transpose(1, 2) @ wd)):
        assert float((got.double() - ref).abs().max()) <= 3e-5 * float(ref.abs().max())


def test_fused_layer_output_dropout_on_projection_kernel():
    """TransLayer (baseline.py:213-218) at n >= 2048: to_out runs on the projection kernel with ITS dropout stream (nystrom_attention.py:99-102:
    nn.Dropout(0.1) after to_out); the backward re-applies exactly that mask (mhimx_dropout_apply_proj).  Reference: the same layer with the
    dropout off, masked by what the forward kept - y = x + (y0 - x) * keep / (1 - p) - through autograd."""
    from mhim_mil_amd import nystrom as ny
    torch.manual_seed(3)
    n, p = 3000, 0.1
    layer = ny.TransLayer(512).to(DEV)
    with torch.no_grad():
        for q in layer.parameters():
            q.normal_(0, 0.05)
        layer.norm.weight.add_(1.0)
    x = rnd(n, 512, seed=41, scale=0.5).requires_grad_()
    ww = rnd(n, 512, seed=42)
    tick = torch.tensor([5], dtype=torch.int64, device=DEV)
    old = layer.attn.dropout
    try:
        layer.attn.dropout = p
        y = layer(x, seed=1234, tick=tick, training=True)
        y0 = layer(x, training=False)
        lin, lin0 = (y - x).detach(), (y0 - x)
        # the mask the forward drew = the projection kernels' dropout stream at (seed, tick), read back through mhimx_dropout_apply_proj;
        # (lin != 0 recovers it too, except where the branch's value is below half an ulp of x and y == x exactly: one element in 1.5 M
        # with the round-5 pseudo-inverse's rounding - which is how this was found)
        ones, kept = torch.ones_like(ww), torch.empty_like(ww)
        from mhim_mil_amd import _lib as L
        L.check(L.lib().mhimx_dropout_apply_proj(ny._st(), ny._ptr(ones), ny._ptr(kept), n, 512, float(p), 1234, ny._ptr(tick)), "dropout_apply_proj")
        keep = (kept != 0).float()
        assert ((lin != 0).float() - keep).abs().sum().item() <= 3 and not ((lin != 0) & (keep == 0)).any()
        assert abs(keep.mean().item() - (1 - p)) < 0.01
        thr16 = round(p * 65536)                                     # the stream's 16-bit threshold: keep probability (65536 - thr16) / 65536
        scale = 65536.0 / (65536 - thr16)
        close(lin, lin0.detach() * keep * scale, 1e-4)
        params = [x] + [q for q in layer.parameters()]
        got = torch.autograd.grad((y * ww).sum(), params)
        ref = torch.autograd.grad(((x + lin0 * keep * scale) * ww).sum(), params)
        for gi, ri in zip(got, ref):
            close(gi, ri, 2e-4)
        # the same seed and tick draw the same mask again; another tick another one
        y2 = layer(x, seed=1234, tick=tick, training=True)
        assert torch.equal(y2, y)
        y3 = layer(x, seed=1234, tick=tick + 1, training=True)
        assert not torch.equal((y3 - x) != 0, lin != 0)
    finally:
        layer.attn.dropout = old


def test_student_tokens_node_equals_the_chain_of_nodes():
    """_StudentTokensFn ([cls ; kept ; merged] and its backward as one node) against the chain it replaces (_FeatureFn -> slices ->
    _MergeFn -> two concatenations): same logits, same flat gradient, on the trainer's single-pass path."""
    from mhim_mil_amd import mhim as MH
    from mhim_mil_amd.engine import FusedTrainer
    n, d = 3000, 256
    base = synth.mhim_state(23, input_dim=d, merge_k=5, baseline="selfattn")
    x = torch.from_numpy(synth.bag(77, n, d)).to(DEV)
    k, n_sel, _ = O.mask_count(n, V2["mask_ratio_h"], V2["mask_ratio_hr"])
    perm, shuf = torch.from_numpy(synth.permutation(31, k)).to(DEV), torch.from_numpy(synth.permutation(41, n - n_sel)).to(DEV)
    res = {}
    old = MH._TOKENS_NODE
    try:
        for flag in (False, True):
            MH._TOKENS_NODE = flag
            s = build(base, input_dim=d, **V2).train()
            t = build(synth.spread_teacher(base), input_dim=d, **V2).train()
            tr = FusedTrainer(s, t, lr=2e-4, aux_alpha=0.5, mm=0.999)
            assert tr.single_pass and s.single_projection_ok(x) and t.single_projection_ok(x)
            logits, losses = tr.forward_backward(x, torch.tensor([1], device=DEV), perm=perm, ids_shuffle=shuf)
            res[flag] = (logits.clone(), tr.flat.grad.clone(), s.merge.global_q_mm.data.clone())
    finally:
        MH._TOKENS_NODE = old
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][2], res[False][2])
    assert res[False][1].abs().max().item() > 0
    close(res[True][1], res[False][1], 1e-6, "flat gradient")


def test_prepared_step_images_are_the_launched_ones():
    """prep_batch kind PREP_PAIR_T == pair_planes(transpose(w)) bit for bit, and a FusedTrainer step that takes its 12 weight images
    from the preparation launch ends in the same parameters as one that launches each image where it is used."""
    from mhim_mil_amd import engine as E, ops
    from mhim_mil_amd.engine import FusedTrainer
    g = torch.Generator(device=DEV).manual_seed(3)
    w = torch.randn(1536, 512, device=DEV, generator=g)
    img = torch.empty(512, 1536, device=DEV)
    ops.prep_batch([(ops.PREP_PAIR_T, w, img)])
    assert torch.equal(img.view(torch.int32), ops.pair_planes(ops.transpose(w)).view(torch.int32))
    n, d = 4200, 64                                             # >= 2048 rows: the layers run on the projection kernel
    base = synth.mhim_state(23, input_dim=d, merge_k=5, baseline="selfattn")
    x = torch.from_numpy(synth.bag(77, n, d)).to(DEV)
    lab = torch.tensor([1], device=DEV)
    finals = []
    for on in (True, False):
        old = E._STEP_IMAGES
        E._STEP_IMAGES = on
        try:
            s = build(base, input_dim=d, **V2).train()
            t = build(synth.spread_teacher(base), input_dim=d, **V2).train()
            tr = FusedTrainer(s, t, lr=2e-4, aux_alpha=0.5, mm=0.999)
            for _ in range(2):
                tr.train_step(x, lab)
            assert not ops._STEP_IMAGES                           # dropped by update(): the weights moved
            finals.append({k: v.detach().clone() for k, v in s.state_dict().items()})
        finally:
            E._STEP_IMAGES = old
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k
