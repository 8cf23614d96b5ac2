"""Standalone abmil / gabmil models (mhim_mil_amd/standalone.py) against fixtures produced by
importing the reference's modules/abmil.py (oracle/gen_golden.py: g3_standalone_*, g14_standalone_train_*)."""
import numpy as np
import pytest
import torch

from mhim_mil_amd import synth
from tests import golden_util as G

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _build(meta, kind, act, train):
    from mhim_mil_amd.standalone import build_model
    if kind == "abmil":
        m = build_model("abmil", input_dim=meta["d"], n_classes=2, dropout=0.0, act=act)
    else:
        m = build_model("gabmil", input_dim=meta["d"], n_classes=2, act=act, dropout=0.)
    sd = {k: torch.from_numpy(synth.normal(meta["pseed"], tuple(shape), std=meta["std"], lane=i + 1).astype(np.float32))
          for i, (k, shape) in enumerate(zip(meta["keys"], meta["shapes"]))}
    missing, unexpected = m.load_state_dict(sd, strict=True)            # same names and shapes as the reference module
    assert not missing and not unexpected
    m = m.to(DEV)
    return m.train() if train else m.eval()


def _x(meta):
    return torch.from_numpy(synth.bag(meta["xseed"], meta["n"], meta["d"])).to(DEV).unsqueeze(0)


def test_eval_forward_fixtures():
    meta, a = G.load("g3_standalone_dattention")
    m = _build(meta, "abmil", "relu", False)
    with torch.no_grad():
        logits, attn, act = m(_x(meta), return_attn=True, return_act=True)
    np.testing.assert_allclose(logits.cpu().numpy().reshape(-1), a["logits"].reshape(-1), atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(attn.cpu().numpy().reshape(-1), a["attn"].reshape(-1), atol=1e-7, rtol=2e-4)
    assert act.shape == (meta["n"], 512)
    meta, a = G.load("g3_standalone_gated")
    m = _build(meta, "gabmil", "relu", False)
    with torch.no_grad():
        logits = m(_x(meta))
    np.testing.assert_allclose(logits.cpu().numpy().reshape(-1), a["logits"].reshape(-1), atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("name", G.names("g14_standalone_train"))
def test_train_mode_gradients(name):
    meta, a = G.load(name)
    m = _build(meta, meta["kind"], meta["act"], True)
    x = _x(meta)
    out = m(x, return_attn=True) if meta["kind"] == "abmil" else m(x)
    logits = out[0] if isinstance(out, (list, tuple)) else out
    loss = torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([meta["label"]], device=DEV))
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy().reshape(-1), a["logits"].reshape(-1), atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(loss.item(), float(a["loss"]), rtol=1e-4)
    if meta["kind"] == "abmil":
        np.testing.assert_allclose(out[1].detach().cpu().numpy().reshape(-1), a["attn"].reshape(-1), atol=1e-7, rtol=2e-4)
    grads = G.tagged(a, "grad")
    params = dict(m.named_parameters())
    assert set(grads) == set(params)
    for k, exp in grads.items():
        g = params[k].grad
        assert g is not None, k
        scale = max(float(np.abs(exp["full"]).max()) if "full" in exp else float(exp["norm"]) / np.sqrt(max(1, g.numel())), 1e-12)
        # (the scorer's output bias has an analytically ZERO gradient - softmax is shift invariant - both sides hold rounding noise)
        # ReLU embeddings: a pre-activation within rounding of 0 may fall on the other side of the kink (3-term bf16 vs fp32): one
        # instance's contribution to a weight-gradient element flips -> a looser absolute floor for those fixtures
        rel_floor = 2e-2 if meta["act"] == "relu" else 2e-4
        G.check_compact(g.cpu().numpy(), exp, rtol=2e-3, atol=max(rel_floor * scale, 1e-6), what=f"{name}:{k}")


def test_factory_and_guards():
    from mhim_mil_amd import _lib as L
    from mhim_mil_amd.mhim import MHIM
    from mhim_mil_amd.standalone import build_model
    assert isinstance(build_model("mhim", input_dim=64, n_classes=2), MHIM)
    pure = build_model("mhim_pure", input_dim=64, n_classes=2, baseline="attn")
    assert isinstance(pure, MHIM) and not pure.merge_enable
    with pytest.raises(NotImplementedError):
        build_model("clam_sb", input_dim=64, n_classes=2)
    bn = build_model("abmil", input_dim=64, n_classes=2, dropout=0.0, act="relu", mil_norm="bn").to(DEV).train()
    with pytest.raises(ValueError):                        # BatchNorm1d on the ONE pooled row in training mode: the reference raises too
        bn(torch.randn(1, 50, 64, device=DEV))
    with pytest.raises(L.MhimxError):
        build_model("abmil", input_dim=64, n_classes=2, dropout=0.0, act="relu", mil_norm="gn")
    with pytest.raises(L.MhimxError):                     # the reference constructor itself fails here (abmil.py:66)
        build_model("gabmil", input_dim=64, n_classes=2, act="relu", mil_norm="ln", embed_norm_pos=0)


@pytest.mark.parametrize("name", G.names("g17_standalone_opt"))
def test_standalone_options_fixture(name):
    """mil_norm='ln' (both positions, norm1), pos='sincos', the gated model with a LayerNorm, TransMIL with mil_norm='ln' / pos='none':
    logits and every parameter gradient vs fixtures made by importing the reference modules (oracle/gen_golden.py g17)."""
    from mhim_mil_amd.standalone import build_model
    meta, a = G.load(name)
    m = build_model(meta["kind"], input_dim=meta["d"], n_classes=2, **meta["kwargs"])
    sd = {k: torch.from_numpy(synth.normal(meta["pseed"], tuple(shape), std=meta["std"], lane=i + 1).astype(np.float32))
          for i, (k, shape) in enumerate(zip(meta["keys"], meta["shapes"]))}
    for k in meta["keys"]:                              # BatchNorm buffers: the generator's start values (positive variances, counter 0)
        if "buf0:" + k in a:
            sd[k] = torch.from_numpy(np.asarray(a["buf0:" + k]))
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    m = m.to(DEV)
    if name.endswith("_eval"):
        m = m.eval()
    elif meta["kind"] == "transmil":
        m = m.eval()
        if meta["kwargs"].get("mil_norm") == "bn":
            m.norm1.train()                             # batch statistics in the input norm, attention dropouts off (as the generator)
    else:
        m = m.train()
    fkw = {"pos": torch.from_numpy(a["pos"]).to(DEV)} if "pos" in a else {}
    logits = m(_x(meta), **fkw)
    loss = torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([meta["label"]], device=DEV))
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy().reshape(-1), a["logits"].reshape(-1), atol=1e-4, rtol=1e-3)
    np.testing.assert_allclose(loss.item(), float(a["loss"]), rtol=1e-3)
    for k, v in m.state_dict().items():                 # running statistics after the step
        if "buf1:" + k in a:
            np.testing.assert_allclose(v.detach().cpu().numpy(), np.asarray(a["buf1:" + k]), rtol=1e-4, atol=1e-6, err_msg=k)
    grads = G.tagged(a, "grad")
    params = dict(m.named_parameters())
    assert set(grads) <= set(params)
    for k in set(params) - set(grads):                  # constructed but never applied in the reference (AttentionGated.norm1): no gradient there either
        assert params[k].grad is None, k
    relu = meta["kwargs"].get("act") == "relu"
    tm = meta["kind"] == "transmil"
    for k, exp in grads.items():
        g = params[k].grad
        assert g is not None, k
        scale = max(float(np.abs(exp["full"]).max()) if "full" in exp else float(exp["norm"]) / np.sqrt(max(1, g.numel())), 1e-12)
        G.check_compact(g.cpu().numpy(), exp, rtol=5e-3 if tm else 2e-3, atol=max((2e-2 if relu else (5e-3 if tm else 2e-4)) * scale, 1e-6),
                        what=f"{name}:{k}")


def test_gated_scorer_dropouts_match_the_masked_math():
    """modules/abmil.py:96-98: Dropout(0.25) after the tanh and after the sigmoid gate, inside the scorer's row kernels
    (mhimx_scorer.gate_drop_p).  torch's Philox masks cannot be reproduced, so the kernel's own counter-based masks are read back
    through mhimx_dropout_apply on an [M, 2A] matrix of ones and the masked scorer is recomputed in fp64 (forward and every gradient)."""
    import ctypes as C
    from mhim_mil_amd import _lib as L, ops
    torch.manual_seed(3)
    M, E, A, p, seed = 700, 512, 384, 0.25, 0x1234567
    T = (torch.randn(M, E, device=DEV) * 0.5).requires_grad_()
    wa, wb = (torch.randn(A, E, device=DEV) * 0.05 for _ in range(2))
    ba, bb = (torch.randn(A, device=DEV) * 0.1 for _ in range(2))
    wc, bc = torch.randn(1, A, device=DEV) * 0.3, torch.randn(1, device=DEV)
    ones = torch.ones(M, 2 * A, device=DEV)
    mask = torch.empty_like(ones)
    L.check(L.lib().mhimx_dropout_apply(ops._stream(), ops._p(ones), ops._p(mask), M, 2 * A, p, seed, None), "dropout_apply")
    keep = float((mask > 0).float().mean())
    assert abs(keep - 0.75) < 0.01 and torch.unique(mask).numel() == 2
    sc = ops.ScorerW(wa, wc, L.ACT["tanh"], ba=ba, wb=wb, bb=bb, bc=bc, prec="bf16x3", gate_drop_p=p, gate_drop_seed=seed)
    st = ops.abmil_pool_fwd(sc, T.detach())
    g_z = torch.randn(E, device=DEV)
    g = ops.abmil_pool_bwd(sc, st, g_z, ops.transpose(wa), ops.transpose(wb), need_bias=True)
    # fp64 restatement with the same masks
    Td = T.detach().double().requires_grad_()
    P = [t.double().requires_grad_() for t in (wa, ba, wb, bb, wc)]
    md = mask.double()
    u = torch.tanh(Td @ P[0].t() + P[1]) * md[:, :A]
    gt = torch.sigmoid(Td @ P[2].t() + P[3]) * md[:, A:]
    s = (u * gt) @ P[4].t() + bc.double()
    z = (torch.softmax(s.view(-1), 0).unsqueeze(0) @ Td).view(-1)
    z.backward(g_z.double())
    rel = lambda x, y: float((x.double() - y).abs().max() / y.abs().max().clamp_min(1e-30))
    assert rel(st.z.view(-1), z.detach()) < 2e-5
    assert rel(g["dT1"], Td.grad) < 2e-4
    for key, ref in (("d_wa", P[0].grad), ("d_ba", P[1].grad), ("d_wb", P[2].grad), ("d_bb", P[3].grad), ("d_wc", P[4].grad)):
        assert rel(g[key].view(ref.shape), ref) < 5e-4, key
    # p = 0 leaves the scorer unchanged, and the standalone model trains with its dropouts on
    from mhim_mil_amd.standalone import build_model
    gm = build_model("gabmil", input_dim=64, n_classes=2, act="relu", dropout=0.25).to(DEV).train()
    x = torch.randn(1, 200, 64, device=DEV)
    l1 = gm(x)
    l1.sum().backward()
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in gm.parameters())
    gm.eval()
    with torch.no_grad():
        assert torch.equal(gm(x), gm(x))


@pytest.mark.parametrize("name", G.names("g15_standalone_transmil"))
def test_standalone_transmil_fixture(name):
    """Standalone TransMIL (wrap-padded tokens as a gather in the embedding GEMM, Nystrom layers, PPEG) vs the fixture made by
    importing modules/transmil.py: logits, both attention maps, every parameter gradient (eval mode: attention dropouts off)."""
    from mhim_mil_amd.standalone import build_model
    meta, a = G.load(name)
    _standalone_transmil_case(build_model, meta, a, name, 5e-3)


def _standalone_transmil_case(build_model, meta, a, name, grad_rtol):
    m = build_model("transmil", input_dim=meta["d"], n_classes=2, dropout=False, act=meta["act"])
    sd = {k: torch.from_numpy(synth.normal(meta["pseed"], tuple(shape), std=meta["std"], lane=i + 1).astype(np.float32))
          for i, (k, shape) in enumerate(zip(meta["keys"], meta["shapes"]))}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    m = m.to(DEV).eval()
    x = _x(meta)
    logits, attn = m(x, return_attn=True)
    loss = torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([meta["label"]], device=DEV))
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy().reshape(-1), a["logits"].reshape(-1), atol=1e-4, rtol=1e-3)
    for i, key in enumerate(("attn0", "attn1")):
        got = attn[i].detach().cpu().numpy().reshape(a[key].shape)
        np.testing.assert_allclose(got, a[key], atol=2e-6, rtol=2e-3, err_msg=key)
    grads = G.tagged(a, "grad")
    params = dict(m.named_parameters())
    assert set(grads) == set(params)
    rel_floor = 2e-2 if meta["act"] == "relu" else 5e-3
    for k, exp in grads.items():
        g = params[k].grad
        assert g is not None, k
        scale = max(float(np.abs(exp["full"]).max()) if "full" in exp else float(exp["norm"]) / np.sqrt(max(1, g.numel())), 1e-12)
        G.check_compact(g.cpu().numpy(), exp, rtol=grad_rtol, atol=max(rel_floor * scale, 1e-7) * (grad_rtol / 5e-3), what=f"{name}:{k}")
