"""The CPU oracle (oracle/mhim_oracle.py) against the fixtures produced from the reference import.

These run everywhere (no GPU, no /root/reference).  They are what pins the
oracle: SURVEY.md §8(c) — the reference itself has no tests or golden vectors.
fp32 tolerances: 2e-6 abs on logits/features (identical math, different op
order), exact equality on every index set in the tie-free families.
"""
import json

import numpy as np
import pytest
import torch

from mhim_mil_amd import synth
from oracle import mhim_oracle as O
from tests import golden_util as G

torch.set_num_threads(4)


def _x(seed, n, d):
    return torch.from_numpy(synth.bag(seed, n, d))


V2_KEYS = ("act", "da_act", "mask_ratio_h", "mask_ratio_hr", "attn2score", "merge_enable", "merge_k", "merge_mm",
           "merge_ratio", "temp_t", "dropout")


def _cfg(meta, **kw):
    d = {k: meta[k] for k in V2_KEYS if k in meta}
    d.update(kw)
    return O.Cfg(**d)


V2 = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True, merge_enable=True,
          merge_k=5, merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)


@pytest.mark.parametrize("name", G.names("g1_abmil_eval"))
def test_g1_abmil_eval(name):
    meta, a = G.load(name)
    p = O.as_torch(synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_enable=False))
    cfg = O.Cfg(act=meta["act"], da_act=meta["da_act"], merge_enable=False)
    x = _x(meta["xseed"], meta["n"], meta["d"])
    logits, attn = O.forward_test(x, p, cfg, return_attn=True)
    _, raw = O.forward_test(x, p, cfg, return_attn=True, no_norm=True)
    np.testing.assert_allclose(logits.numpy(), a["logits"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(attn.numpy(), a["attn"], atol=1e-7, rtol=2e-5)
    np.testing.assert_allclose(raw.numpy(), a["raw"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(O.pure(x, p, cfg).numpy(), a["logits"], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("name", G.names("g2_abmil_train"))
def test_g2_abmil_train_grads(name):
    meta, a = G.load(name)
    p = O.as_torch(synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_enable=False))
    for v in p.values():
        v.requires_grad_(True)
    cfg = O.Cfg(act=meta["act"], da_act=meta["da_act"], merge_enable=False)
    logits = O.pure(_x(meta["xseed"], meta["n"], meta["d"]), p, cfg)
    loss = O.cross_entropy(logits, meta["label"])
    loss.backward()
    assert abs(loss.item() - float(a["loss"])) < 2e-6
    for k, exp in G.tagged(a, "grad").items():
        G.check_compact(p[k].grad.numpy(), exp, rtol=2e-4, atol=1e-7, what=k)


@pytest.mark.parametrize("name", G.names("g3_scorer_"))
def test_g3_scorer_variants(name):
    meta, a = G.load(name)
    p = O.as_torch(synth.mhim_state(meta["seed"], input_dim=64, merge_enable=False, gated=meta["gated"]))
    h = torch.from_numpy(synth.normal(meta["hseed"], (meta["n"], 512)).astype(np.float32))
    z, attn, _ = O.dattention(h, p, meta["act"], meta["gated"])
    _, raw, _ = O.dattention(h, p, meta["act"], meta["gated"], no_norm=True)
    np.testing.assert_allclose(z.numpy(), a["z"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(attn.numpy(), a["attn"], atol=1e-7, rtol=2e-5)
    np.testing.assert_allclose(raw.numpy(), a["raw"], atol=2e-6, rtol=1e-5)


def _fill(meta):
    return {k: torch.from_numpy(synth.normal(meta["pseed"], tuple(s), std=meta["std"], lane=i + 1).astype(np.float32))
            for i, (k, s) in enumerate(zip(meta["keys"], meta["shapes"]))}


def test_g3_standalone_dattention():
    """modules/abmil.py:145-251 — tanh scorer with biases, classifier inside."""
    meta, a = G.load("g3_standalone_dattention")
    sd = _fill(meta)
    x = _x(meta["xseed"], meta["n"], meta["d"])
    h = torch.relu(x @ sd["feature.0.weight"].t() + sd["feature.0.bias"])
    s = O.scorer_logits(h, sd["attention.0.weight"], sd["attention.2.weight"], "tanh",
                        ba=sd["attention.0.bias"], b2=sd["attention.2.bias"])
    z, attn = O.softmax_pool(h, s)
    logits = z @ sd["classifier.weight"].t() + sd["classifier.bias"]
    np.testing.assert_allclose(logits.numpy(), a["logits"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(attn.numpy(), a["attn"], atol=1e-7, rtol=2e-5)


def test_g3_standalone_gated():
    """modules/abmil.py:51-143 — gated tanh*sigmoid scorer, 384 hidden units, with biases."""
    meta, a = G.load("g3_standalone_gated")
    sd = _fill(meta)
    x = _x(meta["xseed"], meta["n"], meta["d"])
    h = torch.relu(x @ sd["feature.0.weight"].t() + sd["feature.0.bias"])
    s = O.scorer_logits(h, sd["attention_a.0.weight"], sd["attention_c.weight"], "tanh", ba=sd["attention_a.0.bias"],
                        b2=sd["attention_c.bias"], wb=sd["attention_b.0.weight"], bb=sd["attention_b.0.bias"])
    z, _ = O.softmax_pool(h, s)
    logits = z @ sd["classifier.0.weight"].t() + sd["classifier.0.bias"]
    np.testing.assert_allclose(logits.numpy(), a["logits"], atol=2e-6, rtol=1e-5)


@pytest.mark.parametrize("name", G.names("g4_teacher"))
def test_g4_teacher(name):
    meta, a = G.load(name)
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"])
    sd = synth.spread_teacher(base) if meta["family"] == "tiefree" else base
    cfg = O.Cfg(**{**V2, "attn2score": meta["attn2score"]})
    feat, score = O.forward_teacher(_x(meta["xseed"], meta["n"], meta["d"]), O.as_torch(sd), cfg)
    np.testing.assert_allclose(feat.numpy(), a["feat"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(score.numpy(), a["score"], atol=3e-7, rtol=2e-5)


@pytest.mark.parametrize("name", G.names("g5_select_ti"))
def test_g5_select_2d(name):
    meta, a = G.load(name)
    perm = a["perm"] if a["perm"].size else None
    len_keep, ids, masked = O.select_mask(meta["n"], a["score"], True, meta["mask_ratio_h"],
                                          random_ratio=meta["mask_ratio_hr"], perm=perm)
    assert len_keep == int(a["len_keep"])
    if meta["family"] == "tiefree":
        # bit-exact: kept ascending, masked in (topk order o perm) order
        # (when > ~60 % of the bag is masked the reference's kept ids come out in CPython set-hash order, not
        #  ascending — SURVEY.md §8 A6; the build's contract is ascending, so compare as sets there)
        if meta["mask_ratio_h"] >= 0.6:
            assert np.array_equal(ids[:len_keep], np.sort(a["kept"]))
        else:
            assert np.array_equal(ids[:len_keep], a["kept"])
        assert np.array_equal(ids[len_keep:], a["masked"])
    else:
        # tie contract (ii): same multiset of selected VALUES; every strictly-greater element selected
        s = a["score"]
        assert np.array_equal(np.sort(s[ids[len_keep:]]), np.sort(s[a["masked"]])) or meta["mask_ratio_hr"] < 1.0
        k = meta["k"]
        top = O.topk_indices(s, k, True)
        kth = s[top[-1]]
        assert set(np.nonzero(s > kth)[0]).issubset(set(top.tolist()))
        assert len(set(ids.tolist())) == meta["n"]


def test_g5_select_low():
    meta, a = G.load("g5_select_low_n512")
    len_keep, ids, _ = O.select_mask(meta["n"], a["score"], False, meta["mask_ratio_l"])
    assert len_keep == int(a["len_keep"])
    assert np.array_equal(ids[:len_keep], a["kept"]) and np.array_equal(ids[len_keep:], a["masked"])


def test_g5_select_mean():
    """msa_fusion='mean' (masking.py:44-48): per-head top-(k // h), their sorted union, a random half of it masked."""
    meta, a = G.load("g5_select_mean_n600")
    len_keep, ids, _ = O.select_mask(meta["n"], a["attn"], True, meta["mask_ratio_h"], random_ratio=meta["mask_ratio_hr"], perm=a["perm"],
                                     msa_fusion="mean")
    assert len_keep == int(a["len_keep"]) and len(a["perm"]) == meta["union"]
    assert np.array_equal(ids[:len_keep], a["kept"]) and np.array_equal(ids[len_keep:], a["masked"])


def test_g5_select_inv():
    """select_inv (masking.py:82-84): the selected rows first, len_keep = their number."""
    meta, a = G.load("g5_select_inv_n512")
    len_keep, ids, _ = O.select_mask(meta["n"], a["score"], True, meta["mask_ratio_h"], select_inv=True)
    assert len_keep == int(a["len_keep"])
    assert np.array_equal(ids[:len_keep], a["first"]) and np.array_equal(ids[len_keep:], a["rest"])


def test_g5_select_vote():
    meta, a = G.load("g5_select_vote_n600")
    len_keep, ids, _ = O.select_mask(meta["n"], a["attn"], True, meta["mask_ratio_h"],
                                     random_ratio=meta["mask_ratio_hr"], perm=a["perm"], msa_fusion="vote")
    assert len_keep == int(a["len_keep"])
    # votes are integers in 0..8: ties are structural, so only the contract-(ii) properties hold
    vote = np.zeros(meta["n"])
    k = meta["k"]
    for h in range(meta["heads"]):
        vote[O.topk_indices(a["attn"][h], k)] += 1
    top_ref_votes = np.sort(vote)[::-1][:k]
    top = O.topk_indices(vote.astype(np.float32), k)
    assert np.array_equal(np.sort(vote[top])[::-1], top_ref_votes)
    assert len(set(ids.tolist())) == meta["n"] and np.all(np.diff(ids[:len_keep]) > 0)


def test_g5_getmask_v1():
    meta, a = G.load("g5_getmask_v1_n1500")
    len_keep, ids = O.get_mask(meta["n"], a["score"], meta["mask_ratio"], meta["mask_ratio_l"], meta["mask_ratio_h"],
                               meta["mask_ratio_hr"], perms=(a["perm1"], None, a["perm3"]))
    assert len_keep == int(a["len_keep"])
    assert np.array_equal(ids[:len_keep], a["kept"])
    assert np.array_equal(np.sort(ids[len_keep:]), np.sort(a["masked"]))


def test_g6_student_attn():
    meta, a = G.load("g6_student_attn")
    p = O.as_torch(synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"]))
    for k, v in p.items():
        v.requires_grad_(k not in O.TRAINABLE_EXCLUDE)
    cfg = _cfg(meta)
    x = _x(meta["xseed"], meta["n"], meta["d"])
    logits, cl, ps, keep, ex = O.forward_student(x, p, cfg, a["teacher_score"], torch.from_numpy(a["teacher_feat"]),
                                                 perm=a["perm"], ids_shuffle=a["ids_shuffle"])
    assert (ps, keep) == (int(a["ps"]), int(a["keep"]))
    np.testing.assert_allclose(logits.detach().numpy(), a["logits"], atol=2e-6, rtol=1e-5)
    assert abs(cl.item() - float(a["cls_loss"])) < 1e-5
    loss = O.cross_entropy(logits, meta["label"]) + meta["aux_alpha"] * cl
    assert abs(loss.item() - float(a["loss"])) < 1e-5
    loss.backward()
    for k, exp in G.tagged(a, "grad").items():
        G.check_compact(p[k].grad.numpy(), exp, rtol=5e-4, atol=2e-7, what=k)
    np.testing.assert_allclose(ex["global_q_new"].numpy(), a["global_q_after"][0], atol=1e-7, rtol=1e-6)


@pytest.mark.parametrize("name", G.names("g7_nystrom"))
def test_g7_nystrom(name):
    meta, a = G.load(name)
    p = O.as_torch(synth.mhim_state(meta["seed"], input_dim=64, baseline="selfattn", merge_enable=False))
    x = torch.from_numpy((synth.normal(meta["xseed"], (meta["n"], meta["dim"])) * 0.5).astype(np.float32))
    pre = "online_encoder.layer1.attn."
    out, attn, v = O.nystrom_attention(x, p, pre, return_attn=True)
    _, attn_raw, _ = O.nystrom_attention(x, p, pre, return_attn=True, no_norm=True)
    np.testing.assert_allclose(out[:8].numpy(), a["out_head"], atol=5e-6, rtol=1e-4)
    np.testing.assert_allclose(out[-8:].numpy(), a["out_tail"], atol=5e-6, rtol=1e-4)
    np.testing.assert_allclose(out.sum(0).numpy(), a["out_sum"], atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(attn.numpy(), a["attn"], atol=1e-6, rtol=1e-3)
    np.testing.assert_allclose(attn_raw.numpy(), a["attn_raw"], atol=5e-4, rtol=2e-3)
    np.testing.assert_allclose(v[:, -4:].numpy(), a["v_tail"], atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("name", G.names("g8_sattention"))
def test_g8_sattention(name):
    meta, a = G.load(name)
    p = O.as_torch(synth.mhim_state(meta["seed"], input_dim=meta["d"], baseline="selfattn", merge_enable=False))
    cfg = O.Cfg(act=meta["act"], baseline="selfattn", merge_enable=False)
    x = _x(meta["xseed"], meta["n"], meta["d"])
    logits, attn = O.forward_test(x, p, cfg, return_attn=True)
    np.testing.assert_allclose(logits.numpy(), a["logits"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(attn[0].numpy(), a["attn1"], atol=1e-6, rtol=1e-3)
    np.testing.assert_allclose(attn[1].numpy(), a["attn2"], atol=1e-6, rtol=1e-3)


@pytest.mark.parametrize("name", G.names("g9_transmil_teacher"))
def test_g9_transmil_teacher(name):
    meta, a = G.load(name)
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"], baseline="selfattn")
    cfg = O.Cfg(**{**V2, "baseline": "selfattn", "attn2score": meta["attn2score"]})
    feat, score = O.forward_teacher(_x(meta["xseed"], meta["n"], meta["d"]), O.as_torch(synth.spread_teacher(base)), cfg)
    np.testing.assert_allclose(feat.numpy(), a["feat"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(score.numpy(), a["score"], atol=2e-6, rtol=1e-3)


def test_g9_transmil_student():
    meta, a = G.load("g9_transmil_student")
    p = O.as_torch(synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"], baseline="selfattn"))
    for k, v in p.items():
        v.requires_grad_(k not in O.TRAINABLE_EXCLUDE)
    cfg = _cfg(meta, baseline="selfattn")
    x = _x(meta["xseed"], meta["n"], meta["d"])
    logits, cl, ps, keep, _ = O.forward_student(x, p, cfg, a["teacher_score"], torch.from_numpy(a["teacher_feat"]),
                                                perm=a["perm"], ids_shuffle=a["ids_shuffle"])
    assert keep == int(a["keep"])
    np.testing.assert_allclose(logits.detach().numpy(), a["logits"], atol=1e-5, rtol=1e-4)
    assert abs(cl.item() - float(a["cls_loss"])) < 2e-5
    (O.cross_entropy(logits, meta["label"]) + meta["aux_alpha"] * cl).backward()
    keys = json.loads(str(a["grad_keys"]))
    for k, n in zip(keys, a["grad_norms"]):
        got = float(p[k].grad.norm())
        assert abs(got - n) <= 2e-3 * n + 1e-7, (k, got, n)


def test_g10_train_steps():
    """Three trainer steps (Adam + EMA) == the reference modules stepped by torch.optim.Adam."""
    meta, a = G.load("g10_train_steps")
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"])
    stu, tea = O.as_torch(base), O.as_torch(synth.spread_teacher(base))
    cfg = _cfg(meta)
    opt = {}
    for step in range(meta["steps"]):
        x = _x(meta["xseed0"] + step, meta["n"], meta["d"])
        stu, tea, opt, info = O.train_step(x, step % 2, stu, tea, opt, cfg, step + 1, perm=a[f"perm{step}"],
                                           ids_shuffle=a[f"shuf{step}"], aux_alpha=meta["aux_alpha"], mm=meta["mm"],
                                           lr=meta["lr"], wd=meta["wd"])
        assert abs(info["loss"] - float(a["losses"][step])) < 2e-5, (step, info["loss"], a["losses"][step])
    for tag, sd in (("stu", stu), ("tea", tea)):
        for k, exp in G.tagged(a, tag).items():
            G.check_compact(sd[k].numpy(), exp, rtol=1e-5, atol=2e-6, what=f"{tag}:{k}")


def test_g18_train_accum8():
    """Two optimiser updates with accumulation_steps = 8 (base_engine.py:29,47-49,100-119): the oracle's window step with the
    reference's sequential in-forward query EMA == the reference modules; the batched contract ("window": every bag attends with the
    window's first queries, the same EMA chain runs on the tokens those forwards produced) differs in second order of (1 - merge_mm)."""
    meta, a = G.load("g18_train_accum8")
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"])
    cfg = _cfg(meta)
    acc = meta["accum"]
    res = {}
    for mode in ("sequential", "window"):
        stu, tea, opt = O.as_torch(base), O.as_torch(synth.spread_teacher(base)), {}
        for u in range(meta["updates"]):
            bs = range(u * acc, (u + 1) * acc)
            xs = [_x(int(a["xseeds"][b]), meta["n"], meta["d"]) for b in bs]
            stu, tea, opt, info = O.train_window(xs, [b % 2 for b in bs], stu, tea, opt, cfg, u + 1, perms=[a[f"perm{b}"] for b in bs],
                                                 shuffles=[a[f"shuf{b}"] for b in bs], aux_alpha=meta["aux_alpha"], mm=meta["mm"],
                                                 lr=meta["lr"], wd=meta["wd"], q_ema=mode)
            tol = 2e-5 if mode == "sequential" else 5e-5
            for j, b in enumerate(bs):
                assert abs(info["loss"][j] - float(a["losses"][b])) < tol, (mode, b, info["loss"][j], a["losses"][b])
                np.testing.assert_allclose(info["logits"][j].numpy(), a["logits"][b], atol=tol, rtol=0)
        res[mode] = (stu, tea)
    for tag, sd in (("stu", res["sequential"][0]), ("tea", res["sequential"][1])):
        for k, exp in G.tagged(a, tag).items():
            # (Adam turns a rounding-level difference of a near-zero accumulated gradient into a fraction of lr = 2e-4)
            G.check_compact(sd[k].numpy(), exp, rtol=1e-5, atol=3e-5, what=f"{tag}:{k}")
    for tag, sd in (("stu", res["window"][0]), ("tea", res["window"][1])):          # two Adam steps of lr 2e-4 each: bound by a tenth of one
        for k, exp in G.tagged(a, tag).items():
            # Adam's first steps move a weight by ~lr whatever the size of its gradient: a near-zero accumulated gradient whose sign
            # the second-order difference flips shows up as ~1.5 lr on that one element - bound the mean tightly, the worst case by 2 lr
            got = sd[k].numpy().astype(np.float64).reshape(-1)
            want = exp["full"].astype(np.float64).reshape(-1) if "full" in exp else exp["sample"].astype(np.float64)
            got = got if "full" in exp else got[::int(exp["stride"])][:want.shape[0]]
            err = np.abs(got - want)
            assert err.mean() <= 5e-6 and err.max() <= 4.1e-4, (f"window {tag}:{k}", err.mean(), err.max())


def test_clip_grad_norm_is_torchs():
    """--clip_grad (base_engine.py:115-119): the oracle's restatement against torch.nn.utils.clip_grad_norm_ itself, and inside a
    train step (a clip value below the gradient norm changes the update, one above does not)."""
    g = torch.Generator().manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(7, 5, generator=g)), torch.nn.Parameter(torch.randn(11, generator=g))]
    for p in ps:
        p.grad = torch.randn(p.shape, generator=g) * 3.0
    raw = {str(i): p.grad.clone() for i, p in enumerate(ps)}
    for max_norm in (0.5, 5.0, 1e3):
        for p, (k, gr) in zip(ps, raw.items()):
            p.grad = gr.clone()
        total = torch.nn.utils.clip_grad_norm_(ps, max_norm)
        mine, tot = O.clip_grad_norm(raw, max_norm)
        assert abs(tot - float(total)) <= 1e-5 * float(total)
        for p, k in zip(ps, raw):
            np.testing.assert_allclose(mine[k].numpy(), p.grad.numpy(), rtol=1e-6, atol=1e-7)
    meta, a = G.load("g10_train_steps")
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"])
    cfg = _cfg(meta)
    x = _x(meta["xseed0"], meta["n"], meta["d"])
    outs = []
    for clip in (None, 1e6, 1e-3):
        stu, _, _, info = O.train_step(x, 0, O.as_torch(base), O.as_torch(synth.spread_teacher(base)), {}, cfg, 1, perm=a["perm0"],
                                       ids_shuffle=a["shuf0"], clip_grad=clip)
        outs.append(stu["feature.0.weight"])
    assert torch.equal(outs[0], outs[1])
    # Adam's first step is sign-like (|update| = lr): with the gradient clipped to 1e-3 the weight-decay term decides some signs: <= 2 lr
    assert not torch.equal(outs[0], outs[2]) and (outs[0] - outs[2]).abs().max() < 4.1e-4


def test_g11_forward_func():
    """CommonMIL.forward_func 7-tuple pieces (common_mil.py:14-48) and validate_func (:56-68)."""
    meta, a = G.load("g11_forward_func")
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"])
    stu, tea = O.as_torch(base), O.as_torch(synth.spread_teacher(base))
    cfg = _cfg(meta)
    x = _x(meta["xseed"], meta["n"], meta["d"])
    feat, score = O.forward_teacher(x, tea, cfg)
    for aux in (0.5, 0.0):
        t_in = None if aux == 0.0 else feat
        logits, cl, ps, keep, _ = O.forward_student(x, stu, cfg, score, t_in, perm=a["perm"], ids_shuffle=a["ids_shuffle"])
        np.testing.assert_allclose(logits.numpy(), a[f"logits_aux{aux}"], atol=2e-6, rtol=1e-5)
        assert abs(float(cl) - float(a[f"auxloss_aux{aux}"])) < 1e-5
        assert [ps, keep, 0.0, 0.0] == list(a[f"pn_kn_aux{aux}"])
    np.testing.assert_allclose(O.forward_test(x, stu, cfg).numpy(), a["val_logits"], atol=2e-6, rtol=1e-5)
    pure = O.as_torch(synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_enable=False))
    lg = O.pure(x, pure, O.Cfg(act="gelu", da_act="relu", merge_enable=False))
    np.testing.assert_allclose(lg.numpy(), a["pure_logits"], atol=2e-6, rtol=1e-5)
    assert list(a["pure_tuple"]) == [0.0, meta["n"], meta["n"], 0.0, 0.0]


def test_g13_dsmil():
    """MHIM(baseline='dsmil') (scope row N1): the oracle's restatement vs the reference fixture — eval logits + attention,
    teacher (B, max-class score), student step (both logit vectors, distillation loss, every gradient), pure train."""
    meta, a = G.load("g13_dsmil")
    d, n = meta["d"], meta["n"]
    base = synth.mhim_state(meta["seed"], input_dim=d, merge_k=5, baseline="dsmil")
    cfg = _cfg(meta, baseline="dsmil")
    x = _x(meta["xseed"], n, d)
    lg, attn = O.forward_test(x, O.as_torch(base), cfg, return_attn=True)
    np.testing.assert_allclose(lg[0].numpy(), a["test_logits_bag"].reshape(-1) if a["test_logits_bag"].ndim > 1 and "feat" not in "test_logits_bag" else a["test_logits_bag"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(lg[1].numpy(), a["test_logits_ins"].reshape(-1) if a["test_logits_ins"].ndim > 1 and "feat" not in "test_logits_ins" else a["test_logits_ins"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(attn.numpy(), a["test_attn"].reshape(-1) if a["test_attn"].ndim > 1 and "feat" not in "test_attn" else a["test_attn"], atol=2e-6, rtol=1e-5)
    feat, score = O.forward_teacher(x, O.as_torch(base), cfg)
    np.testing.assert_allclose(feat.numpy(), a["teacher_feat"].reshape(-1) if a["teacher_feat"].ndim > 1 and "feat" not in "teacher_feat" else a["teacher_feat"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(score.numpy(), a["teacher_score"].reshape(-1) if a["teacher_score"].ndim > 1 and "feat" not in "teacher_score" else a["teacher_score"], atol=2e-6, rtol=1e-5)
    p = O.as_torch(base)
    for k, v in p.items():
        v.requires_grad_(k not in O.TRAINABLE_EXCLUDE)
    logits, cl, ps, keep, _ = O.forward_student(x, p, cfg, a["teacher_score"].reshape(-1) if a["teacher_score"].ndim > 1 and "feat" not in "teacher_score" else a["teacher_score"], torch.from_numpy(a["teacher_feat"]),
                                                perm=a["perm"], ids_shuffle=a["ids_shuffle"])
    assert keep == int(a["keep"])
    np.testing.assert_allclose(logits[0].detach().numpy(), a["logits_bag"].reshape(-1) if a["logits_bag"].ndim > 1 and "feat" not in "logits_bag" else a["logits_bag"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(logits[1].detach().numpy(), a["logits_ins"].reshape(-1) if a["logits_ins"].ndim > 1 and "feat" not in "logits_ins" else a["logits_ins"], atol=2e-6, rtol=1e-5)
    assert abs(float(cl) - float(a["cls_loss"])) < 1e-5
    loss = O.cross_entropy(0.5 * logits[0] + 0.5 * logits[1], meta["label"]) + meta["aux_alpha"] * cl
    assert abs(float(loss) - float(a["loss"])) < 1e-5
    loss.backward()
    for k, exp in G.tagged(a, "grad").items():
        # merge.norm.*: the reference's LayerNorm(global_q) backward sees the post-EMA queries (quirk H7, ~1e-4 of scale)
        scale = float(np.abs(exp["full"]).max()) if "full" in exp else float(exp["norm"]) / np.sqrt(p[k].numel())
        G.check_compact(p[k].grad.numpy(), exp, rtol=2e-4, atol=(1e-3 if k.startswith("merge.norm") else 1e-6) * scale + 1e-9, what=k)
    (lb, li), B = O.forward_test(x, O.as_torch(base), cfg)
    np.testing.assert_allclose(B.numpy(), a["test_B"], atol=2e-6, rtol=1e-5)
    pl = O.pure(x, O.as_torch(synth.mhim_state(meta["seed"], input_dim=d, baseline="dsmil", merge_enable=False)),
                _cfg(meta, baseline="dsmil", merge_enable=False))
    np.testing.assert_allclose(pl[0].numpy(), a["pure_logits_bag"].reshape(-1) if a["pure_logits_bag"].ndim > 1 and "feat" not in "pure_logits_bag" else a["pure_logits_bag"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(pl[1].numpy(), a["pure_logits_ins"].reshape(-1) if a["pure_logits_ins"].ndim > 1 and "feat" not in "pure_logits_ins" else a["pure_logits_ins"], atol=2e-6, rtol=1e-5)


def test_g16_student_eval():
    """MHIM.forward in eval mode (mask applied, Merge keeps every surviving row + k merged tokens, no EMA): reference fixture."""
    meta, a = G.load("g16_student_eval_attn")
    base = synth.mhim_state(meta["seed"], input_dim=meta["d"], merge_k=meta["merge_k"])
    cfg = O.Cfg(**{k: meta[k] for k in V2})
    x = torch.from_numpy(synth.bag(meta["xseed"], meta["n"], meta["d"]))
    logits, cl, ps, keep = O.forward_student_eval(x, O.as_torch(base), cfg, a["teacher_score"], torch.from_numpy(a["teacher_feat"]),
                                                  perm=a["perm"])
    np.testing.assert_allclose(logits.numpy().ravel(), a["logits"].ravel(), atol=2e-6, rtol=0)
    assert abs(float(cl) - float(a["cls_loss"])) < 1e-5 and ps == int(a["ps"]) and keep == int(a["keep"])
