"""The projection's weight-gradient pair (csrc/wgrad.hip: mhimx_rows_dpre_image + mhimx_bag_wgrad) against fp64 math and against the
generic pair it replaces (mhimx_rows_dpre + mhimx_gemm_tn)."""
import numpy as np
import pytest
import torch

from mhim_mil_amd import _lib as L
from mhim_mil_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(N, n_rows, E, D, seed, gather=True, pitch=None):
    g = torch.Generator(device="cpu").manual_seed(seed)
    pitch = pitch or D
    x = torch.randn((N, pitch), generator=g).abs().to(DEV)[:, :D]              # (a view: rows may be wider than D)
    dH = (torch.randn((N + 3, E), generator=g) * 1e-3).to(DEV)
    dact = (torch.randn((N + 3, E), generator=g) * (torch.rand((N + 3, E), generator=g) > 0.25)).to(torch.float16).to(DEV)
    rows = torch.randperm(N, generator=g)[:n_rows].sort().values.to(DEV) if gather else None
    return x, dH, dact, rows


def _ref(x, dH, dact, rows, n_rows):
    idx = rows if rows is not None else torch.arange(n_rows, device=DEV)
    dpre = dH[idx].double() * dact[idx].double()
    return dpre.t() @ x[idx].double(), dpre.sum(0)


@pytest.mark.parametrize("N,n_rows,E,D,gather", [(10000, 9705, 512, 1024, True), (700, 37, 128, 256, True), (300, 300, 256, 512, False),
                                                (5000, 4097, 512, 1536, True), (64, 1, 128, 256, True)])
def test_bag_wgrad_vs_fp64(N, n_rows, E, D, gather):
    x, dH, dact, rows = _case(N, n_rows, E, D, seed=N + n_rows, gather=gather)
    assert ops.bag_wgrad_ok(x, E, n_rows)
    dW, db = ops.bag_wgrad(dH, dact, x, rows, n_rows)
    rW, rb = _ref(x, dH, dact, rows, n_rows)
    sW, sb = float(rW.abs().max()), float(rb.abs().max())
    assert float((dW.double() - rW).abs().max()) <= 2e-5 * sW + 1e-12
    assert float((db.double() - rb).abs().max()) <= 1e-5 * sb + 1e-12


def test_bag_wgrad_accumulate_defer_and_pitch():
    N, n_rows, E, D = 2000, 1777, 512, 1024
    x, dH, dact, rows = _case(N, n_rows, E, D, seed=5, pitch=D + 8)           # bag rows wider than D
    rW, rb = _ref(x, dH, dact, rows, n_rows)
    base_w, base_b = torch.randn((E, D), device=DEV) * 1e-3, torch.randn(E, device=DEV) * 1e-3
    out_w, out_b = base_w.clone(), base_b.clone()
    defer = ops.ReduceList()
    ops.bag_wgrad(dH, dact, x, rows, n_rows, out_w=out_w, out_b=out_b, accumulate=True, defer=defer)
    assert torch.equal(out_w, base_w)                                         # nothing lands before the flush
    ops.reduce_flush(defer)
    now_w, now_b = ops.bag_wgrad(dH, dact, x, rows, n_rows)
    assert torch.equal(out_w, base_w + now_w) and torch.equal(out_b, base_b + now_b)       # queued or not: the same bits
    assert float((now_w.double() - rW).abs().max()) <= 2e-5 * float(rW.abs().max())


def test_bag_wgrad_matches_the_generic_pair():
    N, n_rows, E, D = 3000, 2500, 512, 1024
    x, dH, dact, rows = _case(N, n_rows, E, D, seed=9)
    dW, db = ops.bag_wgrad(dH, dact, x, rows, n_rows)
    dpre, db0 = ops.rows_dpre(dH, dact, rows, n_rows)
    dW0 = ops.gemm_tn(dpre, x, rows=rows, splits=8, prec="bf16x3", M=n_rows)
    assert float((dW - dW0).abs().max()) <= 3e-5 * float(dW0.abs().max())
    assert float((db - db0).abs().max()) <= 1e-5 * float(db0.abs().max())


def test_bag_wgrad_rejects_what_it_cannot_take():
    x, dH, dact, rows = _case(100, 50, 128, 256, seed=1)
    assert not ops.bag_wgrad_ok(x[:, :192], 128, 50) and not ops.bag_wgrad_ok(x, 96, 50)
    with pytest.raises(L.MhimxError):
        ops.bag_wgrad(dH[:, :96].contiguous(), dact[:, :96].contiguous(), x, rows, 50)


def _keep_flags(N, rows):
    keep = torch.zeros(N, dtype=torch.uint8, device=DEV)
    keep[rows] = 1
    return keep


@pytest.mark.parametrize("N,n_rows,E,D", [(10000, 9705, 512, 1024), (700, 37, 128, 256), (333, 333, 256, 512), (5000, 4097, 512, 1536),
                                          (64, 1, 128, 256), (31, 20, 128, 256)])
def test_bag_wgrad_both_operands_as_images_vs_fp64(N, n_rows, E, D):
    """mhimx_bag_wgrad_args.ximg: the bag as an operand image (prep kind 9), dPRE in bag order behind keep flags - the same product."""
    x, dH, dact, rows = _case(N, n_rows, E, D, seed=N + n_rows)
    x = x.contiguous()
    dH[:N][_keep_flags(N, rows) == 0] = float("nan")                          # rows that did not take part hold garbage: never read into the sum
    ximg = torch.empty(ops.bag_ximage_floats(x), device=DEV)
    ops.prep_batch([(ops.PREP_XIMG, x, ximg)])
    dW, db = ops.bag_wgrad(dH, dact, x, None, N, ximg=ximg, keep=_keep_flags(N, rows))
    rW, rb = _ref(x, dH, dact, rows, n_rows)
    sW, sb = float(rW.abs().max()), float(rb.abs().max())
    assert float((dW.double() - rW).abs().max()) <= 2e-5 * sW + 1e-12
    assert float((db.double() - rb).abs().max()) <= 1e-5 * sb + 1e-12
    # against the gathering kernel: the same three-term products, another summation order
    oW, _ = ops.bag_wgrad(dH, dact, x, rows, n_rows)
    assert float((dW - oW).abs().max()) <= 2e-6 * sW + 1e-12


def test_bag_wgrad_images_accumulate_and_defer():
    N, n_rows, E, D = 2000, 1777, 512, 1024
    x, dH, dact, rows = _case(N, n_rows, E, D, seed=5)
    keep = _keep_flags(N, rows)
    ximg = torch.empty(ops.bag_ximage_floats(x), device=DEV)
    ops.prep_batch([(ops.PREP_XIMG, x, ximg)])
    base_w, base_b = torch.randn((E, D), device=DEV) * 1e-3, torch.randn(E, device=DEV) * 1e-3
    out_w, out_b = base_w.clone(), base_b.clone()
    defer = ops.ReduceList()
    ops.bag_wgrad(dH, dact, x, None, N, out_w=out_w, out_b=out_b, accumulate=True, defer=defer, ximg=ximg, keep=keep)
    assert torch.equal(out_w, base_w)
    ops.reduce_flush(defer)
    now_w, now_b = ops.bag_wgrad(dH, dact, x, None, N, ximg=ximg, keep=keep)
    assert torch.equal(out_w, base_w + now_w) and torch.equal(out_b, base_b + now_b)
