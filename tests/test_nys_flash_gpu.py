"""The streamed Nystrom attention kernels (csrc/nys_flash.hip) against fp64 torch math of nystrom_attention.py:111-136."""
import math

import pytest
import torch

from mhim_mil_amd import ops

pytestmark = pytest.mark.gpu
LN2 = math.log(2.0)


def _case(T, seed=0, spread=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = torch.randn(T, 1536, device="cuda", generator=g) * spread
    l = T // 256 if T % 256 == 0 else None
    lm = torch.randn(256, 1024, device="cuda", generator=g) * spread if l is None else qkv[:, :1024].reshape(256, l, 1024).mean(1).contiguous()
    return qkv, lm


def _heads(x):                       # [T, 512] -> [8, T, 64] fp64
    return x.double().reshape(x.shape[0], 8, 64).permute(1, 0, 2)


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("T", [64, 256, 2112, 12800])
def test_a3v_forward_backward_and_cls_row(T):
    scale = 0.125
    qkv, lm = _case(T, seed=T)
    o = ops.NysOperands(qkv, lm, scale)
    a3v, lse3 = ops.nys_a3v_fwd(o)
    k = _heads(qkv[:, 512:1024]).requires_grad_()
    v = _heads(qkv[:, 1024:]).requires_grad_()
    ql = _heads(lm[:, :512]).requires_grad_()
    S = scale * ql @ k.transpose(1, 2)
    P = S.softmax(-1)
    ref = P @ v
    assert _rel(a3v, ref) < 2e-5
    assert float((lse3.double() - torch.logsumexp(S, -1) / LN2).abs().max()) < 1e-4
    g = torch.Generator(device="cuda").manual_seed(1)
    da = torch.randn(8, 256, 64, device="cuda", generator=g)
    ref.backward(da.double())
    dqkv = torch.full_like(qkv, float("nan"))
    dqkv[:, 1024:] = 1.0
    dlm = torch.full_like(lm, float("nan"))
    ops.nys_a3v_bwd(o, a3v, da, lse3, dqkv, dlm, accumulate_dv=True)
    assert _rel(_heads(dqkv[:, 512:1024]), k.grad) < 5e-5
    assert _rel(_heads(dqkv[:, 1024:] - 1.0), v.grad) < 5e-5
    assert _rel(_heads(dlm[:, :512]), ql.grad) < 5e-5
    dqkv2 = torch.empty_like(qkv)
    ops.nys_a3v_bwd(o, a3v, da, lse3, dqkv2, dlm, accumulate_dv=False)
    assert _rel(_heads(dqkv2[:, 1024:]), v.grad) < 5e-5
    u = torch.randn(8, 256, device="cuda", generator=g)
    r = ops.nys_cls_attn(o, lse3, u)
    assert _rel(r, (u.double().unsqueeze(1) @ P.detach()).squeeze(1)) < 2e-5


@pytest.mark.parametrize("T", [64, 256, 2112, 12800])
def test_out_forward_backward(T):
    scale = 0.125
    qkv, lm = _case(T, seed=T + 1)
    g = torch.Generator(device="cuda").manual_seed(2)
    w2 = torch.randn(8, 256, 64, device="cuda", generator=g)
    o = ops.NysOperands(qkv, lm, scale)
    out, lse1 = ops.nys_out_fwd(o, w2)
    q = _heads(qkv[:, :512]).requires_grad_()
    kl = _heads(lm[:, 512:]).requires_grad_()
    w = w2.double().requires_grad_()
    S = scale * q @ kl.transpose(1, 2)
    ref = S.softmax(-1) @ w
    assert _rel(_heads(out), ref) < 2e-5
    assert float((lse1.double() - torch.logsumexp(S, -1) / LN2).abs().max()) < 1e-4
    dout = torch.randn(T, 512, device="cuda", generator=g)
    ref.backward(_heads(dout))
    dqkv = torch.full_like(qkv, float("nan"))
    dlm = torch.full_like(lm, float("nan"))
    dw2 = ops.nys_out_bwd(o, w2, dout, lse1, dqkv, dlm)
    assert _rel(_heads(dqkv[:, :512]), q.grad) < 5e-5
    assert _rel(_heads(dlm[:, 512:]), kl.grad) < 5e-5
    assert _rel(dw2, w.grad) < 5e-5


def test_out_forward_accumulates_into_a_given_buffer():
    qkv, lm = _case(2112, seed=11)
    g = torch.Generator(device="cuda").manual_seed(4)
    w2 = torch.randn(8, 256, 64, device="cuda", generator=g)
    o = ops.NysOperands(qkv, lm, 0.125)
    plain, l0 = ops.nys_out_fwd(o, w2)
    base = torch.randn(2112, 512, device="cuda", generator=g)
    acc, l1 = ops.nys_out_fwd(o, w2, base.clone(), accumulate=True)
    assert torch.equal(acc, plain + base) and torch.equal(l0, l1)     # one fp32 add per element, after the same arithmetic
    with pytest.raises(Exception):
        ops.nys_out_fwd(o, w2, None, accumulate=True)


def test_runs_are_bitwise_repeatable():
    qkv, lm = _case(2112, seed=5)
    o = ops.NysOperands(qkv, lm, 0.125)
    a, la = ops.nys_a3v_fwd(o)
    b, lb = ops.nys_a3v_fwd(o)
    assert torch.equal(a, b) and torch.equal(la, lb)
    w2 = torch.randn(8, 256, 64, device="cuda")
    o1, l1 = ops.nys_out_fwd(o, w2)
    o2, l2 = ops.nys_out_fwd(o, w2)
    assert torch.equal(o1, o2) and torch.equal(l1, l2)


def test_peaked_scores_stay_finite():
    qkv, lm = _case(1280, seed=9, spread=6.0)       # scores of +-100s: the online max has to carry them
    o = ops.NysOperands(qkv, lm, 0.125)
    a3v, lse3 = ops.nys_a3v_fwd(o)
    k, v, ql = _heads(qkv[:, 512:1024]), _heads(qkv[:, 1024:]), _heads(lm[:, :512])
    ref = (0.125 * ql @ k.transpose(1, 2)).softmax(-1) @ v
    assert torch.isfinite(a3v).all() and _rel(a3v, ref) < 1e-4
