"""Entry points added in round 2, each against plain torch fp64 math of the reference line it replaces."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from mhim_mil_amd import _lib as L
from mhim_mil_amd import nystrom as NY
from mhim_mil_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def test_bmm_affine2_and_pair_match_torch():
    """nystrom_attention.py:21-26: xz = x @ z and 7 I - xz from one launch; two independent 256^3 products in one launch."""
    g = torch.Generator(device=DEV).manual_seed(0)
    a, b, c, d = (torch.randn(8, 256, 256, device=DEV, generator=g) * 0.1 for _ in range(4))
    az, t1 = torch.empty_like(a), torch.empty_like(a)
    gm = L.GemmNT(A=NY._ptr(a), lda=256, rows=None, B=NY._ptr(b), ldb=256, C=NY._ptr(az), ldc=256, M=256, N=256, K=256, accumulate=0, prec=L.PREC["bf16x3"])
    L.check(L.lib().mhimx_bmm_affine2(NY._st(), 1, C.byref(gm), 8, 65536, 65536, 65536, 1.0, 0.0, NY._ptr(t1), -1.0, 7.0), "bmm_affine2")
    ref = a.double() @ b.double()
    eye = torch.eye(256, device=DEV, dtype=torch.float64)
    assert _rel(az, ref) < 2e-5 and _rel(t1, 7 * eye - ref) < 2e-5
    o0, o1 = torch.empty_like(a), torch.full_like(a, 0.5)
    NY._bmm_pair(("nt", a, b, o0, 0.25, False), ("tn", c, d, o1, -1.0, True))
    assert _rel(o0, 0.25 * a.double() @ b.double().transpose(1, 2)) < 2e-5
    assert _rel(o1, 0.5 - c.double().transpose(1, 2) @ d.double()) < 2e-5


@pytest.mark.parametrize("M", [5, 700, 4099])
def test_layernorm_backward_with_residual(M):
    """d/dx of y = x + f(LayerNorm(x)) (baseline.py:213-218): LayerNorm backward + the residual branch's gradient in one pass."""
    g = torch.Generator(device=DEV).manual_seed(M)
    E = 512
    x = torch.randn(M, E, device=DEV, generator=g)
    w, b = torch.randn(E, device=DEV, generator=g), torch.randn(E, device=DEV, generator=g)
    dy, res = torch.randn(M, E, device=DEV, generator=g), torch.randn(M, E, device=DEV, generator=g)
    y = torch.empty_like(x)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    L.check(L.lib().mhimx_layernorm_fwd(NY._st(), NY._ptr(x), M, E, NY._ptr(w), NY._ptr(b), NY._ptr(y), NY._ptr(mean), NY._ptr(rstd)), "ln_fwd")
    dx, dw, db = torch.empty_like(x), torch.empty(E, device=DEV), torch.empty(E, device=DEV)
    ws = torch.empty(2 * 512 * E, device=DEV)
    L.check(L.lib().mhimx_layernorm_bwd_res(NY._st(), NY._ptr(dy), NY._ptr(x), M, E, NY._ptr(w), NY._ptr(mean), NY._ptr(rstd), NY._ptr(res),
                                            NY._ptr(dx), NY._ptr(dw), NY._ptr(db), 0, NY._ptr(ws)), "ln_bwd_res")
    xd = x.double().requires_grad_()
    wd, bd = w.double().requires_grad_(), b.double().requires_grad_()
    yr = torch.nn.functional.layer_norm(xd, (E,), wd, bd, 1e-5)
    yr.backward(dy.double())
    assert _rel(y, yr.detach()) < 1e-5
    assert _rel(dx, xd.grad + res.double()) < 2e-5 and _rel(dw, wd.grad) < 2e-5 and _rel(db, bd.grad) < 2e-5


def test_sincos_add_matches_the_reference_formula():
    """emb_position.py:5-83: 2-d sin-cos table of the W x H grid gathered at py * W + px, here evaluated per row."""
    N, Cc, W, H = 333, 512, 23, 19
    rs = np.random.RandomState(1)
    cells = rs.permutation(W * H)[:N]
    pos = torch.from_numpy(np.stack([cells % W, cells // W], 1).astype(np.int64)).to(DEV)
    x = torch.randn(N, Cc, device=DEV)
    out = torch.empty_like(x)
    L.check(L.lib().mhimx_sincos_add(NY._st(), NY._ptr(x), ops._p(pos), N, Cc, NY._ptr(out)), "sincos_add")
    omega = 1.0 / (10000 ** (torch.arange(Cc // 4, dtype=torch.float64, device=DEV) / (Cc / 4.0)))
    px, py = pos[:, 0:1].double(), pos[:, 1:2].double()
    emb = torch.cat([torch.sin(px * omega), torch.cos(px * omega), torch.sin(py * omega), torch.cos(py * omega)], 1)
    assert float((out.double() - (x.double() + emb)).abs().max()) < 2e-5


@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_over_the_instances_matches_torch(training):
    """nn.BatchNorm1d on [1, C, M] (abmil.py:206-210): batch statistics / running statistics, all three gradients."""
    from mhim_mil_amd.standalone import _bn
    torch.manual_seed(4)
    M, Cc = 777, 192
    x = (torch.randn(M, Cc, device=DEV) * 2 + 1).requires_grad_()
    ours, ref = torch.nn.BatchNorm1d(Cc).to(DEV), torch.nn.BatchNorm1d(Cc).to(DEV).double()
    with torch.no_grad():
        for m in (ours, ref):
            m.weight.copy_(torch.linspace(0.5, 1.5, Cc)); m.bias.copy_(torch.linspace(-1, 1, Cc))
            m.running_mean.copy_(torch.linspace(-0.3, 0.3, Cc)); m.running_var.copy_(torch.linspace(0.5, 2.0, Cc))
    ours.train(training); ref.train(training)
    dy = torch.randn(M, Cc, device=DEV)
    y = _bn(x, ours, ours.training)
    y.backward(dy)
    xd = x.detach().double().requires_grad_()
    yr = ref(xd.t().unsqueeze(0))[0].t()
    yr.backward(dy.double())
    assert _rel(y.detach(), yr.detach()) < 2e-5 and _rel(x.grad, xd.grad) < 1e-4
    assert _rel(ours.weight.grad, ref.weight.grad) < 2e-5 and _rel(ours.bias.grad, ref.bias.grad) < 2e-5
    assert _rel(ours.running_mean, ref.running_mean) < 1e-5 and _rel(ours.running_var, ref.running_var) < 1e-5


def test_streamed_attention_backward_with_peaked_scores():
    """Scores of +-100s (a sharpened q / k): the saved log-sum-exps carry them through the recomputing backward kernels."""
    T, scale = 1280, 0.125
    g = torch.Generator(device=DEV).manual_seed(9)
    qkv = torch.randn(T, 1536, device=DEV, generator=g) * 4.0
    lm = qkv[:, :1024].reshape(256, T // 256, 1024).mean(1).contiguous()
    o = ops.NysOperands(qkv, lm, scale)
    w2 = torch.randn(8, 256, 64, device=DEV, generator=g)
    out, lse1 = ops.nys_out_fwd(o, w2)
    H = lambda t: t.double().reshape(t.shape[0], 8, 64).permute(1, 0, 2)
    q, kl, w = H(qkv[:, :512]).requires_grad_(), H(lm[:, 512:]).requires_grad_(), w2.double().requires_grad_()
    ref = (scale * q @ kl.transpose(1, 2)).softmax(-1) @ w
    dout = torch.randn(T, 512, device=DEV, generator=g)
    ref.backward(H(dout))
    dqkv, dlm = torch.empty_like(qkv), torch.empty_like(lm)
    dw2 = ops.nys_out_bwd(o, w2, dout, lse1, dqkv, dlm)
    assert torch.isfinite(dqkv[:, :512]).all() and _rel(H(out), ref.detach()) < 1e-4
    assert _rel(H(dqkv[:, :512]), q.grad) < 2e-4 and _rel(H(dlm[:, 512:]), kl.grad) < 2e-4 and _rel(dw2, w.grad) < 2e-4


@pytest.mark.parametrize("L_", [2048, 5003])
def test_weight_gradient_pair_with_compact_dh_and_optional_dact(L_):
    """mhimx_rows_dpre_image_c (dH compact, dact gathered by the row list) and the dact-less image of any dy, both through
    mhimx_bag_wgrad: dW = (dH * dact[rows])^T X[rows]  /  dy^T x."""
    g = torch.Generator(device=DEV).manual_seed(L_)
    N, E, D = 7000, 512, 1024
    x = torch.randn(N, D, device=DEV, generator=g)
    rows = torch.randperm(N, device=DEV, generator=g)[:L_].sort().values
    dH = torch.randn(L_, E, device=DEV, generator=g) * 0.1
    dact = (torch.rand(N, E, device=DEV, generator=g) * 1.5).half()
    dW, db = ops.bag_wgrad(dH, dact, x, rows, L_, dh_compact=True)
    dpre = dH.double() * dact[rows].double()
    assert _rel(dW, dpre.t() @ x[rows].double()) < 2e-5 and _rel(db, dpre.sum(0)) < 2e-5
    dW2, _ = ops.bag_wgrad(dH, None, x, rows, L_, rows_dh=None, want_bias=False)      # no activation factor, dH compact, X gathered
    assert _rel(dW2, dH.double().t() @ x[rows].double()) < 2e-5
    y = torch.randn(L_, 1536, device=DEV, generator=g) * 0.1                           # a [L, 1536] gradient against [L, 512] rows (to_qkv)
    xs = torch.randn(L_, 512, device=DEV, generator=g)
    dW3, b3 = ops.bag_wgrad(y, None, xs, None, L_)
    assert _rel(dW3, y.double().t() @ xs.double()) < 2e-5 and _rel(b3, y.double().sum(0)) < 2e-5


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 6000, 197000])
def test_random_perm_is_a_permutation_and_depends_on_the_key(n):
    """mhimx_random_perm (what masking.py:67 and merge.py:165-170 draw with torch.randperm): a permutation of 0..n-1 for every key, another
    one for another seed or tick, the same one for the same key; with a source list, the list permuted."""
    tick = torch.tensor([4], dtype=torch.int64, device=DEV)
    p0 = ops.random_perm(n, 123, tick=tick, device=DEV)
    assert torch.equal(torch.sort(p0).values, torch.arange(n, device=DEV))
    assert torch.equal(ops.random_perm(n, 123, tick=tick, device=DEV), p0)
    if n > 7:
        assert not torch.equal(ops.random_perm(n, 124, tick=tick, device=DEV), p0)
        assert not torch.equal(ops.random_perm(n, 123, tick=tick + 1, device=DEV), p0)
        assert not torch.equal(p0, torch.arange(n, device=DEV))
    src = torch.arange(n, device=DEV) * 3 + 1
    assert torch.equal(ops.random_perm(n, 123, tick=tick, src=src), src[p0])


def test_random_perm_prefix_is_a_fair_subset():
    """Every element lands in a prefix of length m with probability m / n (the subsets the path takes are prefixes): 4000 keys, n = 600,
    m = 300 - the inclusion counts of all elements within 5 sigma of binomial, neighbours not travelling together."""
    n, m, trials = 600, 300, 4000
    counts = torch.zeros(n, device=DEV)
    together = 0
    for sd in range(trials):
        p = ops.random_perm(n, 9000 + sd, device=DEV)[:m]
        counts[p] += 1
        inc = torch.zeros(n, dtype=torch.bool, device=DEV)
        inc[p] = True
        together += int((inc[:-1] & inc[1:]).sum())
    sigma = math.sqrt(trials * 0.25)
    assert float((counts - trials * m / n).abs().max()) < 5 * sigma
    exp_pairs = trials * (n - 1) * (m / n) * ((m - 1) / (n - 1))                 # sampling without replacement
    assert abs(together - exp_pairs) < 5 * math.sqrt(exp_pairs)
