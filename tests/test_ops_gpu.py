"""Kernel-level parity: every C-ABI entry point against the CPU oracle / fp64 torch math (GPU box only).

Tolerances are stated per test.  Index outputs are compared exactly.
"""
import math

import numpy as np
import pytest
import torch

from mhim_mil_amd import synth
from oracle import mhim_oracle as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from mhim_mil_amd import ops
    return ops


def rnd(seed, shape, std=1.0):
    return torch.from_numpy(synth.normal(seed, shape, std=std).astype(np.float32))


TOL = {"f32": (2e-6, 2e-5), "f16s": (3e-4, 3e-3), "bf16x3": (2e-5, 2e-4)}   # (relative-to-scale atol, rtol)


@pytest.mark.parametrize("prec", ["f32", "f16s", "bf16x3"])
@pytest.mark.parametrize("M,N,K", [(5, 512, 512), (257, 130, 64), (1000, 512, 1024), (128, 128, 32), (300, 1024, 512)])
def test_gemm_nt_plain(prec, M, N, K):
    ops = _ops()
    a, b = rnd(1, (M, K)), rnd(2, (N, K), std=0.05)
    ref = (a.double() @ b.double().t()).float()
    out = ops.gemm_nt(a.to(DEV), b.to(DEV), prec=prec).cpu()
    scale = ref.abs().max().item()
    atol, rtol = TOL[prec]
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=atol * scale, rtol=rtol)


def test_gemm_nt_identity_asymmetric():
    """A = I with an asymmetric B catches a transposed fragment/C layout (guide §3)."""
    ops = _ops()
    n = 128
    a = torch.eye(n)
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 64.0
    out = ops.gemm_nt(a.to(DEV), b.to(DEV), prec="f32").cpu()
    assert torch.equal(out, b.t())


@pytest.mark.parametrize("prec", ["f32", "f16s"])
def test_gemm_nt_epilogue(prec):
    ops = _ops()
    M, N, K, R = 300, 512, 64, 1000
    x, w, bias = rnd(3, (R, K)).abs(), rnd(4, (N, K), std=0.1), rnd(5, (N,), std=0.1)
    rows = torch.from_numpy(synth.permutation(9, R)[:M].copy())
    rowv, colv = rnd(6, (M,)), rnd(7, (N,))
    pre_ref = x[rows].double() @ w.double().t() + bias.double() + rowv.double()[:, None] * colv.double()[None, :]
    ref = O._act(pre_ref, "gelu")
    base = rnd(8, (M, N))
    out = base.clone().to(DEV)
    pre = torch.empty(M, N, device=DEV)
    ops.gemm_nt(x.to(DEV), w.to(DEV), out=out, rows=rows.to(DEV), bias=bias.to(DEV), act=2, pre=pre, rowv=rowv.to(DEV),
                colv=colv.to(DEV), accumulate=True, prec=prec)
    atol, rtol = TOL[prec]
    scale = pre_ref.abs().max().item()
    np.testing.assert_allclose(pre.cpu().numpy(), pre_ref.float().numpy(), atol=atol * scale, rtol=rtol)
    np.testing.assert_allclose(out.cpu().numpy(), (ref + base.double()).float().numpy(), atol=atol * scale, rtol=rtol)


def test_gemm_nt_dropout_mask_and_hash():
    ops = _ops()
    M, N, K = 200, 512, 64
    x, w = rnd(3, (M, K)), rnd(4, (N, K), std=0.1)
    mask = (torch.from_numpy(synth.uniform(5, (M, N))) >= 0.25).to(torch.uint8)
    ref = (x.double() @ w.double().t()).float() * mask / 0.75
    out = ops.gemm_nt(x.to(DEV), w.to(DEV), drop_p=0.25, drop_mask=mask.to(DEV), prec="f32").cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=1e-5, rtol=1e-5)
    # hashed stream: keep-rate ~ 1-p, deterministic in (seed,row,col), scaled by 1/(1-p)
    full = ops.gemm_nt(x.to(DEV), w.to(DEV), prec="f32").cpu()
    h1 = ops.gemm_nt(x.to(DEV), w.to(DEV), drop_p=0.25, drop_seed=77, prec="f32").cpu()
    h2 = ops.gemm_nt(x.to(DEV), w.to(DEV), drop_p=0.25, drop_seed=77, prec="f32").cpu()
    h3 = ops.gemm_nt(x.to(DEV), w.to(DEV), drop_p=0.25, drop_seed=78, prec="f32").cpu()
    assert torch.equal(h1, h2) and not torch.equal(h1, h3)
    kept = h1 != 0
    assert abs(kept.float().mean().item() - 0.75) < 0.01
    np.testing.assert_allclose(h1[kept].numpy(), (full[kept] / 0.75).numpy(), rtol=1e-6)
    # rows/cols are not correlated: per-row and per-column keep rates stay near 0.75
    assert (kept.float().mean(0) - 0.75).abs().max() < 0.15 and (kept.float().mean(1) - 0.75).abs().max() < 0.1


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("M,K1,K2,splits", [(5, 512, 512, 1), (1000, 128, 512, 1), (3000, 512, 64, 4), (2500, 130, 200, 3),
                                            (70000, 128, 512, 2)])      # last: reduction too long for the LDS row table with 2
def test_gemm_tn(prec, M, K1, K2, splits):
    ops = _ops()
    a, b = rnd(11, (M, K1), std=1e-3), rnd(12, (M + 50, K2)).abs()
    rows = torch.from_numpy(synth.permutation(13, M + 50)[:M].copy())
    ref = (a.double().t() @ b[rows].double()).float()
    out = ops.gemm_tn(a.to(DEV), b.to(DEV), rows=rows.to(DEV), splits=splits, prec=prec).cpu()
    atol, rtol = TOL[prec]
    atol *= max(1.0, (M / 3000) ** 0.5)              # fp32 accumulation over M terms: rounding grows ~ sqrt(M)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=atol * ref.abs().max().item(), rtol=rtol)
    out2 = ops.gemm_tn(a.to(DEV), b.to(DEV), out=out.to(DEV).clone(), rows=rows.to(DEV), splits=splits, accumulate=True,
                       prec=prec).cpu()
    np.testing.assert_allclose(out2.numpy(), 2 * ref.numpy(), atol=2 * atol * ref.abs().max().item(), rtol=rtol)


@pytest.mark.parametrize("M,N,K,gather", [(300, 512, 1024, False), (1000, 128, 64, True), (10000, 512, 1024, True)])
def test_gemm_nt_paired_planes(M, N, K, gather):
    """Both operands pre-split into paired bf16 planes: same result as the in-kernel 3-term split (to fp32 rounding)."""
    ops = _ops()
    a, b = rnd(31, (M + 40, K)).abs(), rnd(32, (N, K), std=0.05)
    rows = torch.from_numpy(synth.permutation(33, M + 40)[:M].copy()) if gather else None
    ref = (a[rows] if gather else a[:M]).double() @ b.double().t()
    ap, bp = ops.pair_planes(a.to(DEV)), ops.pair_planes(b.to(DEV))
    # the pairing itself: hi + lo reproduces x to ~2^-16 relative
    v = ap.cpu().view(torch.bfloat16).view(M + 40, K // 8, 2, 8).float()
    np.testing.assert_allclose((v[:, :, 0] + v[:, :, 1]).reshape(M + 40, K).numpy(), a.numpy(), rtol=2e-5, atol=1e-30)
    out = ops.gemm_nt(ap, bp, rows=None if rows is None else rows.to(DEV), M=M, prec="bf16x3", paired=True).cpu()
    np.testing.assert_allclose(out.numpy(), ref.float().numpy(), atol=2e-5 * ref.abs().max().item(), rtol=1e-4)
    plain = ops.gemm_nt(a.to(DEV), b.to(DEV), rows=None if rows is None else rows.to(DEV), M=M, prec="bf16x3").cpu()
    np.testing.assert_allclose(out.numpy(), plain.numpy(), atol=2e-6 * ref.abs().max().item(), rtol=1e-5)


@pytest.mark.parametrize("act", [1, 2])
def test_projection_dact_output_and_mul_colsum(act):
    """The paired projection's `dact` output = d out / d pre (act' * keep/(1-p)); mul_colsum applies it and sums columns."""
    ops = _ops()
    M, N, K, p = 700, 256, 128, 0.25
    a, b, bias = rnd(41, (M, K)).abs(), rnd(42, (N, K), std=0.1), rnd(43, (N,), std=0.1)
    ap, bp = ops.pair_planes(a.to(DEV)), ops.pair_planes(b.to(DEV))
    dact = torch.empty((M, N), device=DEV)
    pre = torch.empty((M, N), device=DEV)
    out = ops.gemm_nt(ap, bp, bias=bias.to(DEV), act=act, pre=pre, drop_p=p, drop_seed=77, prec="bf16x3", paired=True, dact=dact)
    x = pre.cpu().double().requires_grad_()
    y = torch.relu(x) if act == 1 else torch.nn.functional.gelu(x)
    keep = (out.cpu() != 0) | (y.detach() == 0)                   # dropped elements are exact zeros
    gref, = torch.autograd.grad(y.sum(), x)
    ref = (gref * keep / (1 - p)).float()
    np.testing.assert_allclose(dact.cpu().numpy(), ref.numpy(), atol=2e-6, rtol=2e-5)
    g = rnd(44, (M, N))
    gd = g.to(DEV).clone()
    _, cs = ops.mul_colsum(gd, dact)
    np.testing.assert_allclose(gd.cpu().numpy(), (g * dact.cpu()).numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(cs.cpu().numpy(), (g * dact.cpu()).double().sum(0).float().numpy(), rtol=1e-4, atol=1e-5)
    with pytest.raises(Exception):                                # dact exists only on the paired projection kernel
        ops.gemm_nt(a.to(DEV), b.to(DEV), prec="bf16x3", dact=dact)


def test_transpose():
    ops = _ops()
    x = rnd(1, (130, 77))
    assert torch.equal(ops.transpose(x.to(DEV)).cpu(), x.t().contiguous())


def _scorer_params(seed, E, A, gated, bias):
    wa, wc = rnd(seed, (A, E), std=0.06), rnd(seed + 1, (1, A), std=0.3)
    wb = rnd(seed + 2, (A, E), std=0.06) if gated else None
    ba = rnd(seed + 3, (A,), std=0.1) if bias else None
    bb = rnd(seed + 4, (A,), std=0.1) if (bias and gated) else None
    bc = rnd(seed + 5, (1,), std=0.1) if bias else None
    return wa, wc, wb, ba, bb, bc


@pytest.mark.parametrize("act,gated,bias,A", [("relu", False, False, 128), ("gelu", False, False, 128),
                                               ("tanh", False, True, 128), ("tanh", True, True, 384),
                                               ("relu", True, False, 128)])
@pytest.mark.parametrize("prec", ["f32", "f16s"])
def test_pool_fwd_bwd(act, gated, bias, A, prec):
    """Scorer + softmax pool (two token segments) forward and backward vs fp64 autograd of the oracle."""
    ops = _ops()
    E, M1, M2, Cc = 512, 777, 5, 2
    T1, T2 = rnd(21, (M1, E)).abs(), rnd(22, (M2, E), std=0.5)
    wa, wc, wb, ba, bb, bc = _scorer_params(30, E, A, gated, bias)
    wp = rnd(40, (Cc, E), std=0.05)
    gz = rnd(41, (E,), std=0.01)
    # oracle in fp64
    leaves = {k: (v.double().requires_grad_(True) if v is not None else None)
              for k, v in dict(T1=T1, T2=T2, wa=wa, wc=wc, wb=wb, ba=ba, bb=bb, bc=bc).items()}
    T = torch.cat([leaves["T1"], leaves["T2"]], 0)
    s = O.scorer_logits(T, leaves["wa"], leaves["wc"], act, ba=leaves["ba"], b2=leaves["bc"], wb=leaves["wb"], bb=leaves["bb"])
    z, attn = O.softmax_pool(T, s)
    (z * gz.double()).sum().backward()
    # device
    d = lambda t: None if t is None else t.to(DEV)
    sc = ops.ScorerW(d(wa), d(wc), {"relu": 1, "gelu": 2, "tanh": 3}[act], ba=d(ba), wb=d(wb), bb=d(bb), bc=d(bc), prec=prec)
    st = ops.abmil_pool_fwd(sc, d(T1), d(T2), wp=d(wp))
    f = 1.0 if prec == "f32" else 4.0
    np.testing.assert_allclose(st.s.cpu().numpy(), s.detach().float().numpy(), atol=2e-5 * f, rtol=1e-5 * f)
    np.testing.assert_allclose(st.z.cpu().numpy(), z.detach().float().numpy(), atol=3e-6 * f, rtol=1e-5 * f)
    np.testing.assert_allclose(ops.softmax_from_stats(st.s, st.stats).cpu().numpy(), attn.detach().float().numpy(),
                               atol=1e-7 * f, rtol=3e-5 * f)
    cp_ref = (torch.cat([T1, T2]).double() @ wp.double().t()).float()
    np.testing.assert_allclose(st.cproj.cpu().numpy(), cp_ref.numpy(), atol=2e-5, rtol=1e-5)
    g = ops.abmil_pool_bwd(sc, st, d(gz), ops.transpose(d(wa)), ops.transpose(d(wb)) if gated else None, need_bias=bias,
                           splits=4)
    gtol = dict(rtol=2e-4 * f, atol=0)

    def close(name, got, ref):
        ref = ref.float().numpy()
        np.testing.assert_allclose(got.cpu().numpy().reshape(ref.shape), ref, atol=3e-5 * f * (np.abs(ref).max() + 1e-30),
                                   rtol=gtol["rtol"], err_msg=name)

    close("dT1", g["dT1"], leaves["T1"].grad)
    close("dT2", g["dT2"], leaves["T2"].grad)
    close("d_wa", g["d_wa"], leaves["wa"].grad)
    close("d_wc", g["d_wc"], leaves["wc"].grad)
    if gated:
        close("d_wb", g["d_wb"], leaves["wb"].grad)
    if bias:
        close("d_ba", g["d_ba"], leaves["ba"].grad)
        # d_bc = sum_n ds_n is identically 0 (softmax is shift invariant): absolute check only
        assert abs(float(g["d_bc"].cpu())) < 1e-6 and abs(float(leaves["bc"].grad)) < 1e-12
        if gated:
            close("d_bb", g["d_bb"], leaves["bb"].grad)


def test_pseudo_score_matches_oracle():
    ops = _ops()
    N, E, Cc = 1000, 512, 2
    h, s = rnd(51, (N, E)).abs(), rnd(52, (N,), std=2.0)
    wp, bp = rnd(53, (Cc, E), std=1.0), rnd(54, (Cc,), std=0.1)
    attn = torch.softmax(s, 0)
    ref = O.pseudo_score(h, attn, wp, bp)
    stats = torch.tensor([s.max().item(), torch.exp(s - s.max()).sum().item()])
    cproj = h @ wp.t()
    got, a = ops.pseudo_score(s.to(DEV), stats.to(DEV), cproj.to(DEV), bp.to(DEV), want_attn=True)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(a.cpu().numpy(), attn.numpy(), atol=1e-8, rtol=2e-5)


@pytest.mark.parametrize("name", G.names("g5_select_ti"))
def test_select_mask_golden(name):
    """Device select vs the reference's own output (tie-free: bit-exact; tie-heavy: vs the oracle's contract)."""
    ops = _ops()
    meta, a = G.load(name)
    n, k = meta["n"], meta["k"]
    perm = a["perm"] if a["perm"].size else None
    _, n_sel, _ = O.mask_count(n, meta["mask_ratio_h"], meta["mask_ratio_hr"])
    ids, len_keep, topk = ops.select_mask(torch.from_numpy(a["score"]).to(DEV), k, n_sel, True,
                                          None if perm is None else torch.from_numpy(perm).to(DEV), want_topk=True)
    ids = ids.cpu().numpy()
    o_len, o_ids, _ = O.select_mask(n, a["score"], True, meta["mask_ratio_h"], random_ratio=meta["mask_ratio_hr"], perm=perm)
    assert int(len_keep.item()) == o_len == int(a["len_keep"])
    assert np.array_equal(ids, o_ids)                                   # bit-exact vs the oracle, ties included
    assert np.array_equal(topk.cpu().numpy(), O.topk_indices(a["score"], k, True))
    if meta["family"] == "tiefree":
        assert np.array_equal(ids[o_len:], a["masked"])                 # and vs the reference itself
        assert np.array_equal(ids[:o_len], np.sort(a["kept"]))


@pytest.mark.parametrize("n,k,largest", [(1, 1, True), (2, 1, False), (63, 63, True), (1000, 1, True), (1025, 513, False),
                                          (5000, 5000, True), (50000, 3000, True), (200000, 12000, True), (30000, 700, False),
                                          (16385, 16384, True), (262144, 1, True)])
def test_select_mask_edges(n, k, largest):
    """Edge sizes incl. k = N, k = 1, non-multiples of the block, and BASELINE configs c3/c5 (k=3000/12000)."""
    ops = _ops()
    s = synth.uniform(n + k, (n,)).astype(np.float32)
    s[:: max(1, n // 50)] = s[0]                                         # plant ties
    s[n // 2] = -s[n // 2]                                               # and a negative value
    perm = synth.permutation(5, k)
    n_sel = int(math.ceil(k * 0.5))
    ids, len_keep, topk = ops.select_mask(torch.from_numpy(s).to(DEV), k, n_sel, largest, torch.from_numpy(perm).to(DEV),
                                          want_topk=True)
    top = O.topk_indices(s, k, largest)
    assert np.array_equal(topk.cpu().numpy(), top)
    sel = top[perm[:n_sel]]
    flag = np.zeros(n, bool); flag[sel] = True
    exp = np.concatenate([np.nonzero(~flag)[0], sel])
    assert int(len_keep.item()) == n - n_sel
    assert np.array_equal(ids.cpu().numpy(), exp)


def test_select_mask_union_and_vote():
    ops = _ops()
    meta, a = G.load("g5_getmask_v1_n1500")
    n = meta["n"]
    s = torch.from_numpy(a["score"]).to(DEV)
    # stage 1: random v1 mask (ratio .5 / .001 -> k = n, n_sel = ceil(n*.5)), stage 2: low .2, stage 3: high .01/.5
    k1, n1, _ = O.mask_count(n, meta["mask_ratio"], 0.001)
    ids1, lk1, _ = ops.select_mask(s, k1, n1, False, torch.from_numpy(a["perm1"]).to(DEV))
    m1 = ids1[n - n1:].contiguous()
    k2, n2, _ = O.mask_count(n, meta["mask_ratio_l"], 1.0)
    ids2, lk2, _ = ops.select_mask(s, k2, n2, False, None, other=m1)
    lk2 = int(lk2.item())
    m2 = ids2[lk2:].contiguous()
    k3, n3, _ = O.mask_count(n, meta["mask_ratio_h"], meta["mask_ratio_hr"])
    ids3, lk3, _ = ops.select_mask(s, k3, n3, True, torch.from_numpy(a["perm3"]).to(DEV), other=m2)
    lk3 = int(lk3.item())
    assert lk3 == int(a["len_keep"])
    assert np.array_equal(ids3[:lk3].cpu().numpy(), a["kept"])
    assert np.array_equal(ids3[lk3:].cpu().numpy(), np.sort(a["masked"]))
    # vote fusion
    meta, a = G.load("g5_select_vote_n600")
    vote = ops.vote_scores(torch.from_numpy(a["attn"]).to(DEV), meta["k"], True).cpu().numpy()
    ref = np.zeros(meta["n"], np.float32)
    for h in range(meta["heads"]):
        ref[O.topk_indices(a["attn"][h], meta["k"])] += 1
    assert np.array_equal(vote, ref)


def test_select_mask_union_large_bag():
    """The multi-workgroup form (N > 16384) with an earlier mask (masking.py:74-75): kept ids ascending, then the union ascending; an
    all-equal score vector (every key ties: lowest indices win)."""
    ops = _ops()
    n, k = 40000, 2000
    s = synth.uniform(77, (n,)).astype(np.float32)
    s[::97] = s[5]
    other = np.sort(synth.permutation(8, n)[:3000]).astype(np.int64)
    perm = synth.permutation(9, k)
    n_sel = 1200
    ids, len_keep, topk = ops.select_mask(torch.from_numpy(s).to(DEV), k, n_sel, True, torch.from_numpy(perm).to(DEV),
                                          other=torch.from_numpy(other).to(DEV), want_topk=True)
    top = O.topk_indices(s, k, True)
    assert np.array_equal(topk.cpu().numpy(), top)
    flag = np.zeros(n, bool); flag[top[perm[:n_sel]]] = True; flag[other] = True
    lk = int(len_keep.item())
    assert lk == int((~flag).sum())
    got = ids.cpu().numpy()
    assert np.array_equal(got[:lk], np.nonzero(~flag)[0]) and np.array_equal(got[lk:], np.nonzero(flag)[0])
    flat = np.full(20000, 0.25, np.float32)
    ids, len_keep, topk = ops.select_mask(torch.from_numpy(flat).to(DEV), 500, 500, True, None, want_topk=True)
    assert np.array_equal(topk.cpu().numpy(), np.arange(500)) and np.array_equal(ids.cpu().numpy()[:19500], np.arange(500, 20000))


@pytest.mark.parametrize("prec", ["f32", "f16s"])
@pytest.mark.parametrize("R,k", [(970, 5), (33, 1), (300, 10)])
def test_merge_fwd_bwd(prec, R, k):
    _merge_fwd_bwd(prec, R, k, False)


@pytest.mark.parametrize("R", [8200, 19400, 32768])
def test_merge_fwd_bwd_many_rows(R):
    """The projection-free Merge beyond 256 row tiles (the c5 bag merges ~19 400 rows): the tile partials are merged in chunks of 256."""
    _merge_fwd_bwd("bf16x3", R, 5, False)


@pytest.mark.parametrize("R,k", [(970, 5), (33, 3), (2000, 16)])
def test_merge_fwd_bwd_tile_kernel(R, k):
    """The same comparison with the prep-time fragment image of Wkv: the forward takes the one-kernel-per-row-tile form
    (mca_fused.hip) and the backward consumes the K / V / dots it stored."""
    _merge_fwd_bwd("bf16x3", R, k, True)


def _merge_fwd_bwd(prec, R, k, frag):
    """Merge.merge (LN -> MCA -> to_out -> EMA) forward/backward vs fp64 autograd of the oracle."""
    ops = _ops()
    E = 512
    sd = synth.mhim_state(7, input_dim=64, merge_k=k)
    p = {kk: torch.from_numpy(v).double() for kk, v in sd.items() if kk.startswith("merge.")}
    p["merge.norm.weight"] = p["merge.norm.weight"] + rnd(61, (E,), std=0.1).double()
    p["merge.norm.bias"] = rnd(62, (E,), std=0.1).double()
    p["merge.attn.to_out.0.bias"] = rnd(63, (E,), std=0.1).double()
    for kk in p:
        p[kk].requires_grad_(kk != "merge.global_q_mm")
    X = (rnd(64, (R, E)).abs() * 0.7).double().requires_grad_(True)
    z, g_new = O.merge_tokens(X, p, 0.9999, True)
    dz = rnd(65, (k, E), std=0.1)
    (z * dz.double()).sum().backward()
    f32 = lambda t: t.detach().float().contiguous().to(DEV)
    tr = (ops.transpose(f32(p["merge.attn.to_kv.weight"])), ops.transpose(f32(p["merge.attn.to_q.weight"])),
          ops.transpose(f32(p["merge.attn.to_out.0.weight"])))
    mw = ops.MergeW(f32(p["merge.global_q_mm"]).reshape(k, E), f32(p["merge.norm.weight"]), f32(p["merge.norm.bias"]),
                    f32(p["merge.attn.to_kv.weight"]), f32(p["merge.attn.to_q.weight"]), f32(p["merge.attn.to_out.0.weight"]),
                    f32(p["merge.attn.to_out.0.bias"]), 0.9999, prec=prec, transposes=tr)
    if frag:
        wkv = f32(p["merge.attn.to_kv.weight"])
        img = torch.empty_like(wkv)
        ops.prep_batch([(ops.PREP_FRAG, wkv, img)])
        mw = ops.MergeW(f32(p["merge.global_q_mm"]).reshape(k, E), f32(p["merge.norm.weight"]), f32(p["merge.norm.bias"]), wkv,
                        f32(p["merge.attn.to_q.weight"]), f32(p["merge.attn.to_out.0.weight"]), f32(p["merge.attn.to_out.0.bias"]),
                        0.9999, prec=prec, transposes=tr, wkv_frag=img)
    Xd = f32(X)
    zd, qn, ws = ops.merge_fwd(mw, Xd)
    f = 1.0 if prec == "f32" else 4.0
    np.testing.assert_allclose(zd.cpu().numpy(), z.detach().float().numpy(), atol=5e-6 * f, rtol=1e-5 * f)
    np.testing.assert_allclose(qn.cpu().numpy(), g_new.detach().float().numpy(), atol=1e-7, rtol=1e-6)
    g = ops.merge_bwd(mw, Xd, dz.to(DEV), ws, splits=4)

    def close(name, got, ref):
        ref = ref.float().numpy()
        np.testing.assert_allclose(got.cpu().numpy().reshape(ref.shape), ref, atol=5e-5 * f * (np.abs(ref).max() + 1e-30),
                                   rtol=3e-4 * f, err_msg=name)

    close("dX", g["dX"], X.grad)
    close("d_ln_w", g["d_ln_w"], p["merge.norm.weight"].grad)
    close("d_ln_b", g["d_ln_b"], p["merge.norm.bias"].grad)
    close("d_wkv", g["d_wkv"], p["merge.attn.to_kv.weight"].grad)
    close("d_wq", g["d_wq"], p["merge.attn.to_q.weight"].grad)
    close("d_wo", g["d_wo"], p["merge.attn.to_out.0.weight"].grad)
    close("d_bo", g["d_bo"], p["merge.attn.to_out.0.bias"].grad)


@pytest.mark.parametrize("act", ["relu", "gelu"])
def test_act_bwd_and_colsum(act):
    ops = _ops()
    M, E = 333, 512
    pre = rnd(71, (M, E))
    H = O._act(pre, act)
    dH = rnd(72, (M, E))
    x = pre.double().requires_grad_(True)
    (O._act(x, act) * dH.double()).sum().backward()
    got = ops.act_bwd(dH.clone().to(DEV), H.to(DEV), pre.to(DEV), {"relu": 1, "gelu": 2}[act]).cpu()
    np.testing.assert_allclose(got.numpy(), x.grad.float().numpy(), atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(ops.colsum(got.to(DEV)).cpu().numpy(), got.double().sum(0).float().numpy(), atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("aux", [True, False])
def test_head_fwd_bwd(aux):
    ops = _ops()
    E, Cc = 512, 2
    z = rnd(81, (E,), std=0.5).double().requires_grad_(True)
    t = rnd(82, (E,), std=0.5)
    wp = rnd(83, (Cc, E), std=0.06).double().requires_grad_(True)
    bp = rnd(84, (Cc,), std=0.1).double().requires_grad_(True)
    logits = wp @ z + bp
    ce = O.cross_entropy(logits, 1)
    cl = O.soft_target_ce(z, t.double(), 0.1) if aux else torch.zeros((), dtype=torch.float64)
    loss = 1.0 * ce + 0.5 * cl
    (loss / 2.0).backward()
    label = torch.tensor([1], device=DEV)
    f = lambda x: x.detach().float().to(DEV)
    lg, losses, gz, dwp, dbp = ops.head_fwd_bwd(f(z), t.to(DEV) if aux else None, f(wp), f(bp), label, temp_t=0.1,
                                                main_alpha=1.0, aux_alpha=0.5, inv_accum=0.5)
    np.testing.assert_allclose(lg.cpu().numpy(), logits.detach().float().numpy(), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(losses.cpu().numpy(), [loss.item(), ce.item(), cl.item()], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(gz.cpu().numpy(), z.grad.float().numpy(), atol=1e-7, rtol=2e-4)
    np.testing.assert_allclose(dwp.cpu().numpy(), wp.grad.float().numpy(), atol=1e-7, rtol=2e-4)
    np.testing.assert_allclose(dbp.cpu().numpy(), bp.grad.float().numpy(), atol=1e-7, rtol=2e-4)


def test_adam_ema_matches_torch_adam():
    ops = _ops()
    n_train, n_all = 5000, 5100
    p0, t0 = rnd(91, (n_all,), std=0.1), rnd(92, (n_all,), std=0.1)
    ref_p = torch.nn.Parameter(p0[:n_train].clone())
    opt = torch.optim.Adam([ref_p], lr=2e-4, weight_decay=1e-5)
    p, m, v, tea = p0.clone().to(DEV), torch.zeros(n_train, device=DEV), torch.zeros(n_train, device=DEV), t0.clone().to(DEV)
    ref_t = t0.clone()
    for step in range(1, 4):
        g = rnd(93 + step, (n_train,), std=1e-3)
        ref_p.grad = g.clone()
        opt.step()
        full = torch.cat([ref_p.detach(), p0[n_train:]])
        ref_t = ref_t * 0.9997 + full * (1 - 0.9997)
        gd = g.clone().to(DEV)
        ops.adam_ema(p, gd, m, v, tea, n_train, step, ema_mm=0.9997)
        assert float(gd.abs().max()) == 0.0                              # zero_grad
    np.testing.assert_allclose(p.cpu().numpy(), torch.cat([ref_p.detach(), p0[n_train:]]).numpy(), atol=1e-7, rtol=1e-5)
    np.testing.assert_allclose(tea.cpu().numpy(), ref_t.numpy(), atol=1e-7, rtol=1e-5)


def test_select_rows_fused_random_subsets():
    """mhimx_select_rows: HAM mask + Merge split with device-drawn random subsets (production path).

    Structure is checked exactly (sets, sizes, ordering, candidates = oracle top-k); the random draws are checked for
    determinism in (seed, tick) and for uniformity over 200 seeds (every candidate masked ~ n_sel/k of the time,
    every kept row merged ~ R/L of the time)."""
    ops = _ops()
    n = 10000
    s = ((synth.permutation(5, n) + 0.25 * synth.uniform(6, (n,))) / n).astype(np.float32)
    k, n_sel, _ = O.mask_count(n, 0.03, 0.5)
    L_ = n - n_sel
    Lk = int(L_ * 0.9)
    R = L_ - Lk
    top = set(O.topk_indices(s, k, True).tolist())
    sd = torch.from_numpy(s).to(DEV)
    tick = torch.zeros(1, dtype=torch.int64, device=DEV)
    rows, mask_ids = ops.select_rows(sd, k, n_sel, R, 1234, tick=tick, want_mask_ids=True)
    rows2 = ops.select_rows(sd, k, n_sel, R, 1234, tick=tick)
    assert torch.equal(rows, rows2)
    rows, mask_ids = rows.cpu().numpy(), mask_ids.cpu().numpy()
    assert rows.shape == (L_,) and len(set(rows.tolist())) == L_
    assert np.all(np.diff(rows[:Lk]) > 0) and np.all(np.diff(rows[Lk:]) > 0)
    masked = set(range(n)) - set(rows.tolist())
    assert len(masked) == n_sel and masked <= top
    assert np.array_equal(np.sort(mask_ids[:L_]), np.sort(rows)) and set(mask_ids[L_:].tolist()) == masked
    assert np.all(np.diff(mask_ids[:L_]) > 0)
    ops.tick(tick)
    rows3 = ops.select_rows(sd, k, n_sel, R, 1234, tick=tick).cpu().numpy()
    assert not np.array_equal(rows3, rows)                     # the device counter advances the stream
    cnt_mask = np.zeros(n)
    cnt_merge = np.zeros(n)
    T = 200
    for seed in range(T):
        r = ops.select_rows(sd, k, n_sel, R, 99991 * seed + 7).cpu().numpy()
        m = np.ones(n, bool); m[r] = False
        cnt_mask[m] += 1
        cnt_merge[r[Lk:]] += 1
    cand = np.array(sorted(top))
    f = cnt_mask[cand] / T
    assert abs(f.mean() - n_sel / k) < 1e-9 and f.min() > 0.3 and f.max() < 0.7          # p = 0.5, sigma = 0.035
    never = np.setdiff1d(np.arange(n), cand)
    g = cnt_merge[never] / T
    assert abs(g.mean() - R / L_) < 0.002 and g.max() < 0.25                            # p = 0.1, sigma = 0.021


@pytest.mark.parametrize("M", [31, 1000, 20000])
def test_fused_scorer_fragment_image(M):
    """The one-pass scorer (scorer_fused.hip) with the prep kind-4 fragment image of Wa == the same kernel splitting Wa on
    the fly, bit for bit; against the fp64 oracle within the bf16x3 tolerance.  M = 20000: several tiles per workgroup."""
    ops = _ops()
    E, A, Cc = 512, 128, 2
    T = rnd(51, (M, E)).abs()
    wa, wc, _, ba, _, bc = _scorer_params(52, E, A, False, True)
    wp = rnd(53, (Cc, E), std=0.05)
    s_ref = O.scorer_logits(T.double(), wa.double(), wc.double(), "tanh", ba=ba.double(), b2=bc.double())
    z_ref, _ = O.softmax_pool(T.double(), s_ref)
    d = lambda t: t.to(DEV)
    frag = torch.empty(A, E, device=DEV)
    ops.prep_batch([(ops.PREP_FRAG, d(wa), frag)])
    res = []
    for wf in (None, frag):
        sc = ops.ScorerW(d(wa), d(wc), 3, ba=d(ba), bc=d(bc), prec="bf16x3", wa_frag=wf)
        st = ops.abmil_pool_fwd(sc, d(T), wp=d(wp))
        res.append((st.s.clone(), st.z.clone(), st.cproj.clone(), st.stats.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    np.testing.assert_allclose(res[1][0].cpu().numpy(), s_ref.float().numpy(), atol=8e-5, rtol=4e-5)
    np.testing.assert_allclose(res[1][1].cpu().numpy(), z_ref.float().numpy(), atol=1.2e-5, rtol=4e-5)
    np.testing.assert_allclose(res[1][2].cpu().numpy(), (T.double() @ wp.double().t()).float().numpy(), atol=2e-5, rtol=1e-5)


def test_deferred_reductions_match_immediate():
    """mhimx_reduce_flush: a split TN GEMM and a column-sum pass with their final reductions queued and flushed in one
    launch give the same bits as the immediate two-launch forms (accumulate on and off)."""
    ops = _ops()
    M, K1, K2 = 5000, 512, 1024
    a, b = rnd(71, (M, K1)).to(DEV), rnd(72, (M, K2)).to(DEV)
    dH, dact = rnd(73, (M, K1)).to(DEV), rnd(74, (M, K1)).abs().to(DEV)
    for acc in (False, True):
        base = rnd(75, (K1, K2)).to(DEV)
        base_b = rnd(76, (K1,)).to(DEV)
        ref = ops.gemm_tn(a, b, out=base.clone(), splits=8, accumulate=acc)
        _, ref_b = ops.mul_colsum(dH.clone(), dact, colsum_out=base_b.clone(), accumulate=acc)
        lst = ops.ReduceList()
        got = ops.gemm_tn(a, b, out=base.clone(), splits=8, accumulate=acc, defer=lst)
        _, got_b = ops.mul_colsum(dH.clone(), dact, colsum_out=base_b.clone(), accumulate=acc, defer=lst)
        assert lst.c.n == 2
        ops.reduce_flush(lst)
        assert lst.c.n == 0 and not lst.keep
        assert torch.equal(got, ref) and torch.equal(got_b, ref_b)


def test_fused_scorer_backward_and_transposed_fragment_image():
    """prep kind 5 (fragment image of the transpose, made from the untransposed weight) == kind 4 of the transposed weight;
    the one-pass scorer backward with that image == the same kernel splitting Wa^T on the fly, bit for bit (the fp64
    comparison of the fused backward is test_pool_fwd_bwd's)."""
    ops = _ops()
    E, A, M = 512, 128, 3001
    wa, wc, _, ba, _, bc = _scorer_params(61, E, A, False, True)
    d = lambda t: t.to(DEV)
    wat = ops.transpose(d(wa))
    f4, f5 = torch.empty(E, A, device=DEV), torch.empty(E, A, device=DEV)
    ops.prep_batch([(ops.PREP_FRAG, wat, f4), (ops.PREP_FRAG_T, d(wa), f5)])
    assert torch.equal(f4, f5)
    T = d(rnd(62, (M, E)).abs())
    gz = d(rnd(63, (E,), std=0.01))
    sc = ops.ScorerW(d(wa), d(wc), 3, ba=d(ba), bc=d(bc), prec="bf16x3")
    st = ops.abmil_pool_fwd(sc, T)
    g0 = ops.abmil_pool_bwd(sc, st, gz, wat, need_bias=True)
    g1 = ops.abmil_pool_bwd(sc, st, gz, wat, need_bias=True, wa_t_frag=f5)
    for k in ("dT1", "d_wa", "d_wc", "d_bc", "d_ba"):
        assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize("M2", [0, 5])
def test_pool_finalize_writes_the_pseudo_score(M2):
    """pool_io.pscore (written by the pool's finalize launch) == mhimx_pseudo_score on the same s / stats / cproj, bit for bit."""
    ops = _ops()
    E, A, M1, Cc = 512, 128, 2500, 2
    wa, wc, _, _, _, _ = _scorer_params(65, E, A, False, False)
    d = lambda t: t.to(DEV)
    T1, T2 = d(rnd(66, (M1, E)).abs()), (d(rnd(67, (M2, E))) if M2 else None)
    wp, bp = d(rnd(68, (Cc, E), std=0.05)), d(rnd(69, (Cc,), std=0.1))
    sc = ops.ScorerW(d(wa), d(wc), 1, prec="bf16x3")
    st = ops.abmil_pool_fwd(sc, T1, T2, wp=wp, bp=bp)
    ref = ops.pseudo_score(st.s[:M1], st.stats, st.cproj[:M1], bp)
    assert torch.equal(st.pscore, ref)
