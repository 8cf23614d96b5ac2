"""Generate tests/golden/*.npz FROM THE REFERENCE IMPORT — build container only.

    python -m oracle.gen_golden            # rewrites every fixture

Each fixture stores the generator arguments (seed, dims, hyper-parameters,
injected permutations) and the outputs the *reference's own modules* produced
for them.  Inputs and weights are regenerated on both sides from
mhim_mil_amd/synth.py, so only expectations are stored.  The fixtures are data
(numbers), not reference source.  SURVEY.md Appendix C lists the matrix.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mhim_mil_amd import synth  # noqa: E402
from oracle import _refimport  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

V2 = dict(act="gelu", da_act="relu", mask_ratio_h=0.03, mask_ratio_hr=0.5, attn2score=True,
          merge_enable=True, merge_k=5, merge_mm=0.9999, merge_ratio=0.9, temp_t=0.1, dropout=0.0)


def _save(name, meta, **arrays):
    os.makedirs(OUT, exist_ok=True)
    arrays = {k: np.asarray(v) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=np.array(json.dumps(meta)), **arrays)
    print(f"  wrote {name}.npz  ({sum(a.nbytes for a in arrays.values())/1024:.1f} KiB raw)")


def compact(prefix, arr, limit=2048, nsample=512):
    """Small tensors are stored whole; big ones as (L2 norm, sum, strided sample) — see tests/golden_util.py."""
    a = np.asarray(arr)
    if a.size <= limit:
        return {prefix + "|full": a}
    flat = a.reshape(-1).astype(np.float64)
    stride = max(1, flat.size // nsample)
    return {prefix + "|norm": np.linalg.norm(flat), prefix + "|sum": flat.sum(),
            prefix + "|sample": flat[::stride][:nsample].astype(np.float32), prefix + "|stride": stride}


def _compact_all(tag, d):
    out = {}
    for k, v in d.items():
        out.update(compact(f"{tag}:{k}", v))
    return out


def _fill_module(mod, seed, std=0.05):
    """Deterministic parameters for modules whose init is not restated in synth.mhim_state."""
    sd = {}
    for i, (k, v) in enumerate(mod.state_dict().items()):
        sd[k] = torch.from_numpy(synth.normal(seed, tuple(v.shape), std=std, lane=i + 1).astype(np.float32))
    mod.load_state_dict(sd)
    return mod


def _x(seed, n, d):
    return torch.from_numpy(synth.bag(seed, n, d)).unsqueeze(0)


def _grads(m):
    return {k: p.grad.detach().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}


def g1_abmil_eval(ns):
    for d in (64, 1024):
        for n in (1, 7, 257, 512):
            for act, da in (("relu", "gelu"), ("gelu", "relu")):
                sd = synth.mhim_state(7, input_dim=d, merge_enable=False)
                m = _refimport.build_mhim(ns, sd, input_dim=d, n_classes=2, act=act, da_act=da, baseline="attn",
                                          merge_enable=False, dropout=0.25).eval()
                x = _x(1000 + n, n, d)
                logits, attn = m.forward_test(x, return_attn=True)
                logits2 = m.pure(x)
                _, raw = m.forward_test(x, return_attn=True, no_norm=True)
                assert torch.equal(logits, logits2)
                _save(f"g1_abmil_eval_d{d}_n{n}_{act}", dict(seed=7, xseed=1000 + n, n=n, d=d, act=act, da_act=da),
                      logits=logits[0].numpy(), attn=attn[0].numpy(), raw=raw[0].numpy())


def g2_abmil_train(ns):
    n, d = 257, 64
    for act, da in (("relu", "gelu"), ("gelu", "tanh")):
        sd = synth.mhim_state(7, input_dim=d, merge_enable=False)
        m = _refimport.build_mhim(ns, sd, input_dim=d, n_classes=2, act=act, da_act=da, baseline="attn",
                                  merge_enable=False, dropout=0.0).train()
        x = _x(2000, n, d)
        logits, aux, ps, keep = m.pure(x)
        loss = torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([1]))
        loss.backward()
        g = _grads(m)
        _save(f"g2_abmil_train_{act}_{da}", dict(seed=7, xseed=2000, n=n, d=d, act=act, da_act=da, label=1),
              logits=logits[0].detach().numpy(), loss=loss.item(), **_compact_all("grad", g))


def g3_scorers(ns):
    n, e = 129, 512
    h = torch.from_numpy(synth.normal(31, (1, n, e)).astype(np.float32))
    for act in ("relu", "gelu", "tanh"):
        for gated in (False, True):
            torch.manual_seed(0)
            mod = ns.baseline.DAttention(e, act, gated=gated)
            sd = synth.mhim_state(11, input_dim=64, merge_enable=False, gated=gated)
            pre = "online_encoder."
            mod.load_state_dict({k[len(pre):]: torch.as_tensor(v) for k, v in sd.items() if k.startswith(pre)})
            z, a, actv = mod(h, return_attn=True, return_act=True)
            _, raw = mod(h, return_attn=True, no_norm=True)
            _save(f"g3_scorer_{act}_{'gated' if gated else 'plain'}", dict(seed=11, hseed=31, n=n, act=act, gated=gated),
                  z=z[0].detach().numpy(), attn=a[0].detach().numpy(), raw=raw[0].detach().numpy())
    if ns.abmil is not None:
        d = 64
        x = _x(32, n, d)
        m = _fill_module(ns.abmil.DAttention(d, 2, dropout=0.0, act="relu"), 41).eval()   # biases non-zero
        out = m(x.clone(), return_attn=True, return_act=True)
        _save("g3_standalone_dattention", dict(n=n, d=d, xseed=32, pseed=41, std=0.05,
                                               keys=list(m.state_dict().keys()),
                                               shapes=[list(v.shape) for v in m.state_dict().values()]),
              logits=out[0][0].detach().numpy(), attn=out[1][0].detach().numpy())
        mg = _fill_module(ns.abmil.AttentionGated(d, 2, act="relu"), 42).eval()
        lg = mg(x.clone())
        _save("g3_standalone_gated", dict(n=n, d=d, xseed=32, pseed=42, std=0.05,
                                          keys=list(mg.state_dict().keys()),
                                          shapes=[list(v.shape) for v in mg.state_dict().values()]),
              logits=lg[0].detach().numpy())


def _teacher(ns, sd, d, baseline="attn", attn2score=True, **kw):
    cfg = dict(V2)
    cfg.update(kw)
    cfg["attn2score"] = attn2score
    m = _refimport.build_mhim(ns, sd, input_dim=d, n_classes=2, baseline=baseline, **cfg)
    m.merge_test = False
    return _refimport.zero_aux_dropouts(m).train()


def g4_teacher(ns):
    n, d = 1000, 64
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    for fam, sd in (("tiefree", synth.spread_teacher(base)), ("tieheavy", base)):
        for a2s in (True, False):
            t = _teacher(ns, sd, d, attn2score=a2s)
            x = _x(4000, n, d)
            feat, score = t.forward_teacher(x)
            _save(f"g4_teacher_{fam}_{'score' if a2s else 'attn'}",
                  dict(seed=7, xseed=4000, n=n, d=d, family=fam, attn2score=a2s, merge_k=5),
                  feat=feat[0].numpy(), score=score[0].numpy(), n_unique=len(np.unique(score[0].numpy())))


def g5_select(ns):
    # 2-D scores: tie-free (continuous) and tie-heavy (quantised) families.
    for n in (512, 10000):
        for fam in ("tiefree", "tieheavy"):
            # tie-free: a permuted grid + jitter => all values distinct in fp32 by construction
            s = ((synth.permutation(50 + n, n) + 0.25 * synth.uniform(50 + n, (n,))) / n).astype(np.float32)
            assert len(np.unique(s)) == n
            if fam == "tieheavy":
                s = (np.floor(s * 300) / 300 * 0.01 + 0.5).astype(np.float32)
            for (mrh, hr) in ((0.01, 1.0), (0.03, 0.5), (0.05, 0.5), (0.6, 0.5)):
                eff = min(mrh / hr, 1.0)
                k = int(np.ceil(n * eff))
                torch.manual_seed(123)
                perm = torch.randperm(k).numpy() if hr < 1.0 else None
                torch.manual_seed(123)
                len_keep, ids = ns.select_mask_fn(n, torch.from_numpy(s)[None], True, mrh, len_keep_other=n,
                                                  random_ratio=hr)
                ids = ids[0].numpy()
                _save(f"g5_select_{fam}_n{n}_h{mrh}_r{hr}",
                      dict(n=n, family=fam, mask_ratio_h=mrh, mask_ratio_hr=hr, sseed=50 + n, k=k),
                      score=s, perm=perm if perm is not None else np.zeros(0, np.int64),
                      len_keep=len_keep, kept=ids[:len_keep], masked=ids[len_keep:])
    # low-attention selection (largest=False), v1
    n = 512
    s = synth.uniform(77, (n,)).astype(np.float32)
    len_keep, ids = ns.select_mask_fn(n, torch.from_numpy(s)[None], False, 0.2)
    _save("g5_select_low_n512", dict(n=n, mask_ratio_l=0.2, sseed=77), score=s, len_keep=len_keep,
          kept=ids[0, :len_keep].numpy(), masked=ids[0, len_keep:].numpy())
    # 3-D 'vote' fusion on per-head attention (attn2score off + selfattn)
    n, h = 600, 8
    a = synth.uniform(88, (h, n)).astype(np.float32)
    torch.manual_seed(5)
    k = int(np.ceil(n * min(0.05 / 0.5, 1.0)))
    perm = torch.randperm(k).numpy()
    torch.manual_seed(5)
    len_keep, ids = ns.select_mask_fn(n, torch.from_numpy(a)[None], True, 0.05, len_keep_other=n, random_ratio=0.5,
                                      msa_fusion="vote")
    _save("g5_select_vote_n600", dict(n=n, heads=h, mask_ratio_h=0.05, mask_ratio_hr=0.5, sseed=88, k=k),
          attn=a, perm=perm, len_keep=len_keep, kept=ids[0, :len_keep].numpy(), masked=ids[0, len_keep:].numpy())
    # 3-D 'mean' fusion (masking.py:44-48: per-head top-(k // h), their sorted union) with a random half of the union masked
    n, h = 600, 8
    a = ((np.stack([synth.permutation(120 + i, n) for i in range(h)]) + 0.25 * synth.uniform(89, (h, n))) / n).astype(np.float32)
    kk = int(np.ceil(n * min(0.2 / 0.5, 1.0)) // h)
    union = np.unique(np.concatenate([np.argsort(-a[i], kind="stable")[:kk] for i in range(h)]))
    torch.manual_seed(6)
    perm = torch.randperm(len(union)).numpy()
    torch.manual_seed(6)
    len_keep, ids = ns.select_mask_fn(n, torch.from_numpy(a)[None], True, 0.2, len_keep_other=n, random_ratio=0.5, msa_fusion="mean")
    _save("g5_select_mean_n600", dict(n=n, heads=h, mask_ratio_h=0.2, mask_ratio_hr=0.5, sseed=89, kk=kk, union=int(len(union))),
          attn=a, perm=perm, len_keep=len_keep, kept=ids[0, :len_keep].numpy(), masked=ids[0, len_keep:].numpy())
    # select_inv (masking.py:82-84): the SELECTED rows come first and len_keep counts them
    n = 512
    s = ((synth.permutation(131, n) + 0.25 * synth.uniform(131, (n,))) / n).astype(np.float32)
    len_keep, ids = ns.select_mask_fn(n, torch.from_numpy(s)[None], True, 0.1, select_inv=True)
    _save("g5_select_inv_n512", dict(n=n, mask_ratio_h=0.1, sseed=131), score=s, len_keep=len_keep,
          first=ids[0, :len_keep].numpy(), rest=ids[0, len_keep:].numpy())
    # v1 combined masks through MHIM.get_mask: random .5 + low .2 + high .01/.5
    n = 1500
    s = synth.uniform(99, (n,)).astype(np.float32)
    sd = synth.mhim_state(7, input_dim=64, merge_k=5)
    m = _refimport.build_mhim(ns, sd, input_dim=64, n_classes=2, baseline="attn", mask_ratio=0.5, mask_ratio_l=0.2,
                              **{**V2, "mask_ratio_h": 0.01})
    torch.manual_seed(9)
    k1 = int(np.ceil(n * 1.0))                     # ratio .5/.001 > 1 -> k = ps, random_ratio = .5
    perm1 = torch.randperm(k1).numpy()
    k3 = int(np.ceil(n * 0.02))
    perm3 = torch.randperm(k3).numpy()
    torch.manual_seed(9)
    len_keep, ids = m.get_mask(n, 0, torch.from_numpy(s)[None])
    _save("g5_getmask_v1_n1500", dict(n=n, mask_ratio=0.5, mask_ratio_l=0.2, mask_ratio_h=0.01, mask_ratio_hr=0.5,
                                      sseed=99), score=s, perm1=perm1, perm3=perm3, len_keep=len_keep,
          kept=ids[0, :len_keep].numpy(), masked=ids[0, len_keep:].numpy())


def _student_case(ns, n, d, baseline, seed_rng, xseed, fam="tiefree", wseed=7, **kw):
    base = synth.mhim_state(wseed, input_dim=d, merge_k=5, baseline=baseline)
    tsd = synth.spread_teacher(base) if fam == "tiefree" else base
    t = _teacher(ns, tsd, d, baseline=baseline, **kw)
    s = _teacher(ns, base, d, baseline=baseline, **kw)
    x = _x(xseed, n, d)
    feat, score = t.forward_teacher(x)
    cfg = {**V2, **kw}
    eff = min(cfg["mask_ratio_h"] / cfg["mask_ratio_hr"], 1.0)
    k = int(np.ceil(n * eff))
    torch.manual_seed(seed_rng)
    perm = torch.randperm(k)
    n_sel = int(np.ceil(k * cfg["mask_ratio_hr"]))
    L = n - n_sel
    ids_shuffle = torch.argsort(torch.rand(L), dim=0)
    torch.manual_seed(seed_rng)
    q0 = s.merge.global_q_mm.detach().clone()
    logits, cls_loss, ps, keep = s(x, score, feat, i=0)
    loss = torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([1])) + 0.5 * cls_loss
    loss.backward()
    return dict(t=t, s=s, x=x, feat=feat, score=score, perm=perm.numpy(), ids_shuffle=ids_shuffle.numpy(),
                logits=logits, cls_loss=cls_loss, ps=ps, keep=keep, loss=loss, q0=q0, k=k, n_sel=n_sel)


def g6_student(ns):
    n, d = 1000, 64
    c = _student_case(ns, n, d, "attn", 17, 6000)
    g = _grads(c["s"])
    _save("g6_student_attn", dict(seed=7, xseed=6000, n=n, d=d, label=1, aux_alpha=0.5, family="tiefree", **V2),
          teacher_feat=c["feat"][0].numpy(), teacher_score=c["score"][0].numpy(), perm=c["perm"],
          ids_shuffle=c["ids_shuffle"], logits=c["logits"][0].detach().numpy(), cls_loss=c["cls_loss"].item(),
          ps=c["ps"], keep=c["keep"], loss=c["loss"].item(),
          global_q_after=c["s"].merge.global_q_mm.detach().numpy(), **_compact_all("grad", g))


def g7_nystrom(ns):
    dim = 512
    for n in (257, 601, 1024):
        torch.manual_seed(21)
        att = ns.NystromAttention(dim=dim, dim_head=64, heads=8, num_landmarks=256, pinv_iterations=6,
                                  residual=True, dropout=0.0).eval()
        sd = synth.mhim_state(13, input_dim=64, baseline="selfattn", merge_enable=False)
        pre = "online_encoder.layer1.attn."
        att.load_state_dict({k[len(pre):]: torch.as_tensor(v) for k, v in sd.items() if k.startswith(pre)})
        x = torch.from_numpy((synth.normal(700 + n, (1, n, dim)) * 0.5).astype(np.float32))
        out = att(x)
        out2, attn, v = att(x, return_attn=True)
        _, attn_raw, _ = att(x, return_attn=True, no_norm=True)
        assert torch.allclose(out, out2)
        _save(f"g7_nystrom_n{n}", dict(seed=13, xseed=700 + n, n=n, dim=dim),
              out_head=out[0, :8].detach().numpy(), out_tail=out[0, -8:].detach().numpy(),
              out_sum=out[0].sum(0).detach().numpy(), attn=attn[0].detach().numpy(),
              attn_raw=attn_raw[0].detach().numpy(), v_tail=v[0, :, -4:].detach().numpy())


def g8_sattention(ns):
    for n in (20, 48, 600, 1500):                 # 20 < 37 tokens: PPEG's zero-padding to a 7 x 7 grid (emb_position.py:100-104)
        sd = synth.mhim_state(13, input_dim=64, baseline="selfattn", merge_enable=False)
        m = _refimport.build_mhim(ns, sd, input_dim=64, n_classes=2, act="gelu", baseline="selfattn",
                                  merge_enable=False, dropout=0.0).eval()
        x = _x(800 + n, n, 64)
        logits = m.forward_test(x)
        lg2, attn = m.forward_test(x, return_attn=True)
        _save(f"g8_sattention_n{n}", dict(seed=13, xseed=800 + n, n=n, d=64, act="gelu"),
              logits=logits[0].numpy(), attn1=attn[0][0].numpy(), attn2=attn[1][0].numpy())


def g9_transmil_teacher(ns):
    n, d = 600, 64
    base = synth.mhim_state(13, input_dim=d, merge_k=5, baseline="selfattn")
    for a2s in (True, False):
        t = _teacher(ns, synth.spread_teacher(base), d, baseline="selfattn", attn2score=a2s)
        x = _x(9000, n, d)
        feat, score = t.forward_teacher(x)
        _save(f"g9_transmil_teacher_{'score' if a2s else 'attn'}",
              dict(seed=13, xseed=9000, n=n, d=d, attn2score=a2s, merge_k=5),
              feat=feat[0].numpy(), score=score[0].numpy())
    c = _student_case(ns, n, d, "selfattn", 19, 9000, wseed=13)
    g = _grads(c["s"])
    keys = sorted(g)
    _save("g9_transmil_student", dict(seed=13, xseed=9000, n=n, d=d, label=1, aux_alpha=0.5, **V2),
          teacher_feat=c["feat"][0].numpy(), teacher_score=c["score"][0].numpy(), perm=c["perm"],
          ids_shuffle=c["ids_shuffle"], logits=c["logits"][0].detach().numpy(), cls_loss=c["cls_loss"].item(),
          keep=c["keep"], loss=c["loss"].item(),
          grad_norms=np.array([np.linalg.norm(g[k]) for k in keys]), grad_keys=np.array(json.dumps(keys)))


def g10_train_steps(ns):
    """3 consecutive trainer steps restated from base_engine.py:76-167 with the reference MHIM modules,
    torch.optim.Adam(lr 2e-4, wd 1e-5) (train_utils.py:58-65) and the per-parameter EMA loop."""
    n, d, mm = 257, 64, 0.9997
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    t = _teacher(ns, synth.spread_teacher(base), d)
    s = _teacher(ns, base, d)
    opt = torch.optim.Adam([p for p in s.parameters() if p.requires_grad], lr=2e-4, weight_decay=1e-5)
    perms, shuffles, losses = [], [], []
    for step in range(3):
        x = _x(10000 + step, n, d)
        label = torch.tensor([step % 2])
        feat, score = t.forward_teacher(x)
        k = int(np.ceil(n * 0.06))
        torch.manual_seed(100 + step)
        perm = torch.randperm(k)
        L = n - int(np.ceil(k * 0.5))
        ids = torch.argsort(torch.rand(L), dim=0)
        torch.manual_seed(100 + step)
        logits, cls_loss, ps, keep = s(x, score, feat, i=step)
        loss = torch.nn.functional.cross_entropy(logits.view(1, -1), label) + 0.5 * cls_loss
        loss.backward()
        opt.step()
        opt.zero_grad()
        with torch.no_grad():
            for pq, pk in zip(s.parameters(), t.parameters()):
                pk.mul_(mm).add_(pq.detach(), alpha=1.0 - mm)
        perms.append(perm.numpy()); shuffles.append(ids.numpy()); losses.append(loss.item())
    arrays = {}
    for nm, mdl in (("stu", s), ("tea", t)):
        arrays.update(_compact_all(nm, {k_: v.numpy() for k_, v in mdl.state_dict().items() if k_ != "merge.global_q"}))
    _save("g10_train_steps", dict(seed=7, n=n, d=d, mm=mm, steps=3, xseed0=10000, aux_alpha=0.5, lr=2e-4, wd=1e-5,
                                  **V2), losses=np.array(losses), perm0=perms[0], perm1=perms[1], perm2=perms[2],
          shuf0=shuffles[0], shuf1=shuffles[1], shuf2=shuffles[2], **arrays)


def g18_train_accum8(ns):
    """2 optimiser updates with --accumulation_steps 8 (16 bags), restated from base_engine.py:29,47-49,100-119,146-167 with the
    reference MHIM modules: loss / 8 per bag, gradients accumulate, ONE torch.optim.Adam step and ONE EMA-teacher update per window;
    Merge's in-forward EMA of the global queries runs bag after bag (merge.py:142-143).  SURVEY Appendix C, G10 'accumulation 8'."""
    n, d, mm, acc = 300, 256, 0.9997, 8
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    t = _teacher(ns, synth.spread_teacher(base), d)
    s = _teacher(ns, base, d)
    opt = torch.optim.Adam([p for p in s.parameters() if p.requires_grad], lr=2e-4, weight_decay=1e-5)
    perms, shuffles, losses, logits_all, q_trace = [], [], [], [], []
    opt.zero_grad()
    xseeds, cand = [], 18000
    for b in range(2 * acc):
        k = int(np.ceil(n * 0.06))
        while True:                                        # torch.topk's tie order is implementation-defined: only bags whose
            x = _x(cand, n, d)                             # k-th and (k+1)-th teacher scores are clearly apart enter the fixture
            cand += 1
            feat, score = t.forward_teacher(x)
            srt = np.sort(score[0].numpy())[::-1]
            if srt[k - 1] - srt[k] > 2e-5:
                break
        xseeds.append(cand - 1)
        label = torch.tensor([b % 2])
        torch.manual_seed(300 + b)
        perm = torch.randperm(k)
        L = n - int(np.ceil(k * 0.5))
        ids = torch.argsort(torch.rand(L), dim=0)
        torch.manual_seed(300 + b)
        logits, cls_loss, ps, keep = s(x, score, feat, i=b)
        loss = torch.nn.functional.cross_entropy(logits.view(1, -1), label) + 0.5 * cls_loss
        (loss / acc).backward()
        perms.append(perm.numpy()); shuffles.append(ids.numpy()); losses.append(loss.item())
        logits_all.append(logits[0].detach().numpy().copy())
        q_trace.append(float(s.merge.global_q_mm.detach().norm()))
        if (b + 1) % acc == 0:
            opt.step()
            opt.zero_grad()
            with torch.no_grad():
                for pq, pk in zip(s.parameters(), t.parameters()):
                    pk.mul_(mm).add_(pq.detach(), alpha=1.0 - mm)
    arrays = {}
    for nm, mdl in (("stu", s), ("tea", t)):
        arrays.update(_compact_all(nm, {k_: v.numpy() for k_, v in mdl.state_dict().items() if k_ != "merge.global_q"}))
    for b in range(2 * acc):
        arrays[f"perm{b}"], arrays[f"shuf{b}"] = perms[b], shuffles[b]
    _save("g18_train_accum8", dict(seed=7, n=n, d=d, mm=mm, accum=acc, updates=2, xseed0=18000, aux_alpha=0.5, lr=2e-4, wd=1e-5, **V2),
          losses=np.array(losses), logits=np.array(logits_all), q_norms=np.array(q_trace), xseeds=np.array(xseeds), **arrays)


def g11_forward_func(ns):
    import types
    n, d = 257, 64
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    t = _teacher(ns, synth.spread_teacher(base), d)
    s = _teacher(ns, base, d)
    eng = ns.CommonMIL(None)
    x = _x(11000, n, d)
    label = torch.tensor([1])
    out = {}
    for aux in (0.5, 0.0):
        args = types.SimpleNamespace(model="mhim", baseline="attn", aux_alpha=aux)
        torch.manual_seed(77)
        k = int(np.ceil(n * 0.06))
        perm = torch.randperm(k)
        ids = torch.argsort(torch.rand(n - int(np.ceil(k * 0.5))), dim=0)
        torch.manual_seed(77)
        q0 = s.merge.global_q_mm.detach().clone()
        r = eng.forward_func(args, s, t, x, label, None, 1, 0, 0, 0, None)
        s.merge.global_q_mm.data.copy_(q0)
        out[f"logits_aux{aux}"] = r[0][0].detach().numpy()
        out[f"auxloss_aux{aux}"] = float(r[2].detach() if torch.is_tensor(r[2]) else r[2])
        out[f"pn_kn_aux{aux}"] = np.array([r[3], r[4], r[5], r[6]], dtype=np.float64)
        out["perm"], out["ids_shuffle"] = perm.numpy(), ids.numpy()
    s.eval()
    args = types.SimpleNamespace(model="mhim", baseline="attn", aux_alpha=0.5)
    lg, lab = eng.validate_func(args, s, x, label, None, 1, 0, None)
    out["val_logits"] = lg[0].numpy()
    pure_sd = synth.mhim_state(7, input_dim=d, merge_enable=False)
    mp = _refimport.build_mhim(ns, pure_sd, input_dim=d, n_classes=2, act="gelu", da_act="relu", baseline="attn",
                               merge_enable=False, dropout=0.0).train()
    args = types.SimpleNamespace(model="mhim_pure", baseline="attn", aux_alpha=0.0)
    r = eng.forward_func(args, mp, None, x, label, None, 1, 0, 0, 0, None)
    out["pure_logits"] = r[0][0].detach().numpy()
    out["pure_tuple"] = np.array([float(r[2]), r[3], r[4], r[5], r[6]])
    _save("g11_forward_func", dict(seed=7, xseed=11000, n=n, d=d, **V2), **out)


def g13_dsmil(ns):
    """MHIM(baseline='dsmil') (scope row N1): eval forward_test (+attn), teacher, student step with gradients, pure train."""
    d, n = 64, 500
    cfg = {**V2, "merge_k": 5}
    base = synth.mhim_state(19, input_dim=d, merge_k=5, baseline="dsmil")
    x = torch.from_numpy(synth.bag(13000, n, d)).unsqueeze(0)
    m = _refimport.build_mhim(ns, base, input_dim=d, baseline="dsmil", **cfg)
    _refimport.zero_aux_dropouts(m)
    out = {}
    m.eval()
    with torch.no_grad():
        lg = m.forward_test(x)
        lg2, a = m.forward_test(x, return_attn=True)
        _, a_raw = m.forward_test(x, return_attn=True, no_norm=True)
    assert isinstance(lg, tuple) and lg[1].shape == (1, 2, 512)          # (mhim.py:262: the encoder's ([l, l_ins], B) tuple)
    out["test_logits_bag"], out["test_logits_ins"] = lg[0][0][0].numpy(), lg[0][1][0].numpy()
    out["test_B"] = lg[1][0].numpy()
    np.testing.assert_allclose(lg2[0][0].numpy(), lg[0][0][0].numpy())
    out["test_attn"] = a[0].numpy()
    np.testing.assert_allclose(a_raw[0].numpy(), a[0].numpy())          # cls_attn: the instance logits either way
    m.train()
    tsd = synth.spread_teacher(base) if False else base
    t = _refimport.build_mhim(ns, tsd, input_dim=d, baseline="dsmil", **cfg)
    _refimport.zero_aux_dropouts(t)
    t.train()
    with torch.no_grad():
        tfeat, tscore = t.forward_teacher(x)
    out["teacher_feat"], out["teacher_score"] = tfeat[0].numpy(), tscore[0].numpy()
    k = int(np.ceil(n * min(cfg["mask_ratio_h"] / cfg["mask_ratio_hr"], 1.0)))
    n_sel = int(np.ceil(k * cfg["mask_ratio_hr"]))
    torch.manual_seed(29)                                   # the student's two draws, replayed (cf. _student_case)
    perm = torch.randperm(k).numpy()
    shuf = torch.argsort(torch.rand(n - n_sel), dim=0).numpy()
    torch.manual_seed(29)
    logits, cl, ps, keep = m(x, tscore, tfeat[0], i=0)
    mixed = 0.5 * logits[0].view(1, -1) + 0.5 * logits[1].view(1, -1)
    loss = torch.nn.functional.cross_entropy(mixed, torch.tensor([1])) + 0.5 * cl
    loss.backward()
    out["logits_bag"], out["logits_ins"] = logits[0][0].detach().numpy(), logits[1][0].detach().numpy()
    out["cls_loss"], out["loss"], out["keep"] = float(cl), float(loss), keep
    keys, norms = [], []
    for k_, p_ in m.named_parameters():
        if p_.grad is not None and k_ != "merge.global_q":
            keys.append(k_)
            norms.append(float(p_.grad.norm()))
            out.update(compact(f"grad:{k_}", p_.grad.numpy()))
    out["grad_keys"], out["grad_norms"] = np.array(json.dumps(keys)), np.array(norms)
    out["perm"], out["ids_shuffle"] = perm, shuf
    # pure (no mask / merge) train-mode logits
    mp = _refimport.build_mhim(ns, synth.mhim_state(19, input_dim=d, baseline="dsmil", merge_enable=False), input_dim=d,
                               baseline="dsmil", **{**cfg, "merge_enable": False})
    mp.train()
    pl, _, _, _ = mp.pure(x)
    out["pure_logits_bag"], out["pure_logits_ins"] = pl[0][0].detach().numpy(), pl[1][0].detach().numpy()
    _save("g13_dsmil", dict(seed=19, xseed=13000, n=n, d=d, label=1, aux_alpha=0.5, **cfg), **out)


def g14_standalone_train(ns):
    """Standalone abmil / gabmil (modules/abmil.py) in train mode, dropout off: logits, attention and every parameter
    gradient of CE(logits, label) - fixtures for mhim_mil_amd/standalone.py."""
    if ns.abmil is None:
        print("  (modules/abmil.py not importable: skipped)")
        return
    n, d = 300, 64
    x = _x(33, n, d)
    for name, mod, pseed in (("abmil_gelu", ns.abmil.DAttention(d, 2, dropout=0.0, act="gelu"), 43),
                             ("abmil_relu", ns.abmil.DAttention(d, 2, dropout=0.0, act="relu"), 44),
                             ("gabmil_relu", ns.abmil.AttentionGated(d, 2, act="relu", dropout=0.), 45),
                             ("gabmil_gelu", ns.abmil.AttentionGated(d, 2, act="gelu", dropout=0.), 46)):
        m = _fill_module(mod, pseed).train()
        out = m(x.clone(), return_attn=True) if name.startswith("abmil") else m(x.clone())
        logits = out[0] if isinstance(out, (list, tuple)) else out
        loss = torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([1]))
        loss.backward()
        extra = {"attn": out[1][0].detach().numpy()} if isinstance(out, (list, tuple)) else {}
        _save(f"g14_standalone_train_{name}", dict(n=n, d=d, xseed=33, pseed=pseed, std=0.05, label=1, kind=name.split("_")[0],
                                                   act=name.split("_")[1], keys=list(m.state_dict().keys()),
                                                   shapes=[list(v.shape) for v in m.state_dict().values()]),
              logits=logits[0].detach().numpy(), loss=loss.item(), **extra, **_compact_all("grad", _grads(m)))


def g15_standalone_transmil(ns):
    """Standalone TransMIL (modules/transmil.py) in eval mode (its attention dropouts off): logits, the two cls-row attention
    maps and every parameter gradient of CE(logits, label) - fixtures for mhim_mil_amd/standalone.TransMIL."""
    if getattr(ns, "transmil", None) is None:
        print("  (modules/transmil.py not importable: skipped)")
        return
    d = 64
    # 300 -> 18 x 18 grid with 24 wrapped tokens; 441 = 21 x 21; 20 -> 5 x 5 (a grid below the 7 x 7 stencil, transmil.py:57-64)
    for n, act, pseed in ((300, "relu", 47), (441, "gelu", 48), (20, "gelu", 49)):
        x = _x(34 + n, n, d)
        m = _fill_module(ns.transmil.TransMIL(d, 2, dropout=False, act=act), pseed).eval()
        # Host-library note: this image's torch-CPU oneDNN path returns a WRONG fp32 weight gradient for the depth-wise 33-tap
        # res_conv when the padded token count is exactly 256 (taps 17..31 off by up to 5 % against fp64 and against the native ATen
        # path; 512 and up are right - tools/exp_onednn_conv.py).  Bags of fewer than 256 tokens are therefore run on ATen's own
        # convolution: same reference code, correct arithmetic.
        with torch.backends.mkldnn.flags(enabled=n + 1 > 256):
            out = m(x.clone(), return_attn=True)
            logits, attn = out[0], out[1]
            loss = torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([1]))
            loss.backward()
        _save(f"g15_standalone_transmil_n{n}_{act}", dict(n=n, d=d, xseed=34 + n, pseed=pseed, std=0.05, label=1, act=act,
                                                          keys=list(m.state_dict().keys()),
                                                          shapes=[list(v.shape) for v in m.state_dict().values()]),
              logits=logits[0].detach().numpy(), loss=loss.item(), attn0=attn[0][0].detach().numpy(), attn1=attn[1][0].detach().numpy(),
              **_compact_all("grad", _grads(m)))


def g17_standalone_options(ns):
    """The non-default options of the standalone models (VERDICT r1 N4 gaps): mil_norm='ln' at both positions and norm1 (abmil.py:171-178,
    :237), pos='sincos' (emb_position.py:5-83), the gated model with a LayerNorm (abmil.py:68-70), TransMIL with mil_norm='ln' and with
    pos='none' (transmil.py:73-74,83-84).  Train mode, dropout off; logits and every parameter gradient of CE(logits, label)."""
    if ns.abmil is None or getattr(ns, "transmil", None) is None:
        print("  (modules/abmil.py / transmil.py not importable: skipped)")
        return
    n, d = 300, 64
    x = _x(35, n, d)
    W, Hh = 23, 17
    rs = np.random.RandomState(5)
    cells = rs.permutation(W * Hh)[:n]
    pos = torch.from_numpy(np.concatenate([[[W, Hh]], np.stack([cells % W, cells // W], 1)], 0).astype(np.int64)).unsqueeze(0)
    cases = (("abmil_ln0", "abmil", dict(dropout=0.0, act="gelu", mil_norm="ln", embed_norm_pos=0), 51, {}),
             ("abmil_ln1", "abmil", dict(dropout=0.0, act="relu", mil_norm="ln", embed_norm_pos=1), 52, {}),
             ("abmil_sincos", "abmil", dict(dropout=0.0, act="gelu", pos="sincos"), 53, {"pos": pos}),
             ("gabmil_ln1", "gabmil", dict(act="gelu", dropout=0., mil_norm="ln", embed_norm_pos=1), 54, {}),
             ("transmil_ln", "transmil", dict(dropout=False, act="gelu", mil_norm="ln"), 55, {}),
             ("transmil_posnone", "transmil", dict(dropout=False, act="relu", pos="none"), 56, {}),
             # mil_norm='bn': BatchNorm1d over the instances of the bag (train mode: batch statistics; the running statistics after the
             # step are part of the fixture).  DAttention's norm1 sees ONE pooled row and raises in train mode (the reference cannot
             # train that setting): its fixture is the eval forward with non-trivial running statistics
             ("gabmil_bn0", "gabmil", dict(act="gelu", dropout=0., mil_norm="bn", embed_norm_pos=0), 57, {}),
             ("gabmil_bn1", "gabmil", dict(act="relu", dropout=0., mil_norm="bn", embed_norm_pos=1), 58, {}),
             ("transmil_bn", "transmil", dict(dropout=False, act="gelu", mil_norm="bn"), 59, {}),
             ("abmil_bn1_eval", "abmil", dict(dropout=0.0, act="gelu", mil_norm="bn", embed_norm_pos=1), 60, {}),
             # embed_feat=False: the bag rows are the tokens (input_dim == inner_dim = 512)
             ("abmil_noembed", "abmil", dict(dropout=0.0, act="gelu", embed_feat=False, mil_norm="ln", embed_norm_pos=0), 61, {"_d": 512}),
             ("transmil_noembed", "transmil", dict(dropout=False, act="gelu", embed_feat=False), 62, {"_d": 512}))
    for name, kind, kw, pseed, fkw in cases:
        cls = {"abmil": ns.abmil.DAttention, "gabmil": ns.abmil.AttentionGated, "transmil": ns.transmil.TransMIL}[kind]
        fkw = dict(fkw)
        d = fkw.pop("_d", 64)
        x = _x(35, n, d)
        m = _fill_module(cls(d, 2, **kw), pseed)
        with torch.no_grad():                                     # (filled from N(0, 0.05): keep the variances positive, the counter integral)
            for k_, v_ in m.state_dict().items():
                if k_.endswith("running_var"):
                    v_.copy_(v_.abs() + 0.5)
                if k_.endswith("num_batches_tracked"):
                    v_.zero_()
        init_sd = {k_: v_.clone() for k_, v_ in m.state_dict().items()}
        if name.endswith("_eval"):
            m = m.eval()
        elif kind == "transmil":
            m = m.eval()
            if kw.get("mil_norm") == "bn":
                m.norm1.train()                                   # batch statistics in the input norm, attention dropouts off
        else:
            m = m.train()
        out = m(x.clone(), **fkw)
        logits = out[0] if isinstance(out, (list, tuple)) else out
        loss = torch.nn.functional.cross_entropy(logits.view(1, -1), torch.tensor([1]))
        loss.backward()
        extra = {"pos": pos.numpy()} if fkw else {}
        for k_, v_ in m.state_dict().items():                     # buffers: their values before and after the step
            if "running_" in k_ or k_.endswith("num_batches_tracked"):
                extra["buf0:" + k_] = init_sd[k_].numpy()
                extra["buf1:" + k_] = v_.detach().numpy()
        _save(f"g17_standalone_opt_{name}", dict(n=n, d=d, xseed=35, pseed=pseed, std=0.05, label=1, kind=kind, kwargs=kw,
                                                 keys=list(m.state_dict().keys()), shapes=[list(v.shape) for v in m.state_dict().values()]),
              logits=logits.view(-1).detach().numpy(), loss=loss.item(), **extra, **_compact_all("grad", _grads(m)))


def g16_student_eval(ns):
    """MHIM.forward with the module in eval mode (mhim.py:318-378: mask applied, Merge keeps every surviving row and appends the k tokens
    merged from all of them, merge.py:197-203) - the branch the reference's trainer never takes but the class allows."""
    n, d = 600, 64
    base = synth.mhim_state(7, input_dim=d, merge_k=5)
    t = _teacher(ns, synth.spread_teacher(base), d).eval()
    s = _teacher(ns, base, d).eval()
    x = _x(16000, n, d)
    feat, score = t.forward_teacher(x)
    k = int(np.ceil(n * min(V2["mask_ratio_h"] / V2["mask_ratio_hr"], 1.0)))
    torch.manual_seed(23)
    perm = torch.randperm(k)
    torch.manual_seed(23)
    q0 = s.merge.global_q_mm.detach().clone()
    with torch.no_grad():
        logits, cls_loss, ps, keep = s(x, score, feat, i=0)
    assert torch.equal(q0, s.merge.global_q_mm.detach())
    _save("g16_student_eval_attn", dict(seed=7, xseed=16000, n=n, d=d, **V2), teacher_feat=feat[0].numpy(), teacher_score=score[0].numpy(),
          perm=perm.numpy(), logits=logits[0].numpy(), cls_loss=float(cls_loss), ps=ps, keep=keep)


def g12_cosine_scheduler(ns):
    """utils.cosine_scheduler (utils.py:199-210) for the two schedules the trainer builds (modules/__init__.py:72-75,177-181).
    utils.py imports the whole training stack, so only this function's AST node is compiled and run."""
    import ast
    with open(os.path.join(_refimport.REF_ROOT, "utils.py")) as f:
        tree = ast.parse(f.read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "cosine_scheduler")
    g = {"np": np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), "utils.py", "exec"), g)
    cs = g["cosine_scheduler"]
    _save("g12_cosine_scheduler", dict(cases=[[0.9997, 1.0, 5, 7, 0, 1.0], [0.03, 0.0, 4, 9, 0, 0.0], [0.9998, 1.0, 6, 5, 2, 1.0]]),
          c0=cs(0.9997, 1.0, epochs=5, niter_per_ep=7, start_warmup_value=1.0),
          c1=cs(0.03, 0.0, epochs=4, niter_per_ep=9),
          c2=cs(0.9998, 1.0, epochs=6, niter_per_ep=5, warmup_epochs=2, start_warmup_value=1.0))


def main():
    ns = _refimport.load()
    torch.set_num_threads(8)
    only = set(sys.argv[1:])                     # python -m oracle.gen_golden g14_standalone_train  -> just that family
    for fn in (g1_abmil_eval, g2_abmil_train, g3_scorers, g4_teacher, g5_select, g6_student, g7_nystrom,
               g8_sattention, g9_transmil_teacher, g10_train_steps, g18_train_accum8, g11_forward_func, g12_cosine_scheduler, g13_dsmil,
               g14_standalone_train, g15_standalone_transmil, g16_student_eval, g17_standalone_options):
        if only and fn.__name__ not in only:
            continue
        print(fn.__name__)
        fn(ns)


if __name__ == "__main__":
    main()
