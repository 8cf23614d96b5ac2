"""CPU oracle for the MHIM aggregation path — TEST INFRASTRUCTURE ONLY.

This file is a from-scratch functional restatement (torch CPU / numpy) of the
algorithm the reference executes on the hot path named by BASELINE.json's
north_star.  It is the *checker*: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The
product (mhim_mil_amd/) never imports anything under oracle/ and fails loudly
when the HIP library is missing.

Parity pin: the reference has no tests and no golden vectors, so
this restatement is pinned by (i) oracle/check_against_reference.py, which
imports /root/reference in the build container and compares every function
here against the reference's own modules on seeded inputs, and (ii) the
fixtures under tests/golden/ that oracle/gen_golden.py produced from the
*reference import* (not from this file).  tests/test_oracle_golden.py re-checks
(ii) everywhere, including the GPU box where /root/reference does not exist.

Everything is written against plain tensors and a flat ``params`` dict that uses
the reference's state_dict key names; there are no nn.Module
classes here.  All randomness (dropout masks, permutations) is injected.

Citations are into /root/reference (file:line).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- #
# small helpers
# --------------------------------------------------------------------------- #


def _act(x: torch.Tensor, name: Optional[str]) -> torch.Tensor:
    """relu / exact-erf gelu / tanh / identity (nn.GELU default is the erf form)."""
    if name is None:
        return x
    name = name.lower()
    if name == "relu":
        return torch.relu(x)
    if name == "gelu":
        return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))
    if name == "tanh":
        return torch.tanh(x)
    if name in ("none", "identity"):
        return x
    raise ValueError(name)


def _linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


DRAW_DROPOUT = False       # timing runs only (bench.py cpu_baseline): with no injected mask, draw one with torch's own CPU dropout as the
                           # reference does (mhim.py:76) - parity runs inject masks or run with p = 0


def _drop(x, mask, p):
    """Inverted dropout with an injected keep-mask (1 = keep).  p == 0 or mask None -> identity (unless DRAW_DROPOUT)."""
    if p == 0.0:
        return x
    if mask is None:
        return torch.nn.functional.dropout(x, p, True) if DRAW_DROPOUT else x
    return x * mask.to(x.dtype) / (1.0 - p)


def as_torch(sd: dict, dtype=torch.float32) -> dict:
    return {k: torch.as_tensor(np.asarray(v)).to(dtype).clone() for k, v in sd.items()}


# --------------------------------------------------------------------------- #
# A1  feature projection   (modules/mhim.py:69-76,84; used :193-194,:335-336)
# --------------------------------------------------------------------------- #


def feature(x, params, act="relu", drop_mask=None, p=0.0):
    """H = dropout(act(X W1^T + b1)).  x: [N, D] -> [N, E].

    mhim.py:69-74 builds Linear(+ReLU|GELU; any other ``act`` string adds no
    activation at all), mhim.py:76 the dropout applied right after it.
    """
    a = act.lower() if isinstance(act, str) else None
    a = a if a in ("relu", "gelu") else None
    h = _act(_linear(x, params["feature.0.weight"], params["feature.0.bias"]), a)
    return _drop(h, drop_mask, p)


# --------------------------------------------------------------------------- #
# A2  instance scorer + softmax pool   (mhim_modules/baseline.py:8-41, 43-86, 88-110)
# --------------------------------------------------------------------------- #


def scorer_logits(h, wa, w2, act="gelu", ba=None, b2=None, wb=None, bb=None):
    """Raw per-instance score s[N].

    ungated (baseline.py:14-36):  s = w2 . act(Wa h (+ba)) (+b2)
    gated   (baseline.py:49-75):  s = wc . (act(Wa h) * sigmoid(Wb h)) (+bc)
    ``act`` outside {gelu, relu, tanh} means no activation (baseline.py:17-22).
    """
    a = act.lower() if isinstance(act, str) else None
    a = a if a in ("relu", "gelu", "tanh") else None
    u = _act(_linear(h, wa, ba), a)
    if wb is not None:
        u = u * torch.sigmoid(_linear(h, wb, bb))
    return _linear(u, w2, b2).squeeze(-1)


def softmax_pool(h, s):
    """A = softmax_N(s); z = sum_n A_n h_n  (baseline.py:33-36)."""
    a = torch.softmax(s, dim=-1)
    return a @ h, a


def dattention(h, params, act="gelu", gated=False, no_norm=False, prefix="online_encoder.attention."):
    """mhim_modules.baseline.DAttention.forward with return_attn=True, return_act=True.

    Returns (z [E], attn [N] (softmax weights, or raw scores if no_norm), act = h).
    """
    if gated:
        s = scorer_logits(h, params[prefix + "attention_a.0.weight"], params[prefix + "attention_c.weight"],
                          act, wb=params[prefix + "attention_b.0.weight"],
                          ba=params.get(prefix + "attention_a.0.bias"), bb=params.get(prefix + "attention_b.0.bias"),
                          b2=params.get(prefix + "attention_c.bias"))
    else:
        s = scorer_logits(h, params[prefix + "attention.0.weight"], params[prefix + "attention.2.weight"], act,
                          ba=params.get(prefix + "attention.0.bias"), b2=params.get(prefix + "attention.2.bias"))
    z, a = softmax_pool(h, s)
    return z, (s if no_norm else a), h


# --------------------------------------------------------------------------- #
# A3  pseudo score (ABMIL)   (mhim_modules/scoring.py:37-58)
# --------------------------------------------------------------------------- #


def pseudo_score(h, attn, wp, bp):
    """score_n = max_c softmax_c( (A_n h_n) . Wp_c + bp[0] ).

    scoring.py:49 scales the features by the attention first, :52 projects on the
    predictor weight, :54 adds the *class-0* bias to every class (sic), :55 softmax
    over classes, :56 max over classes.
    """
    f = h * attn[:, None]
    cam = wp @ f.t() + bp[0]
    cam = torch.softmax(cam, dim=0)
    return cam.max(dim=0).values


# --------------------------------------------------------------------------- #
# A6  select_mask_fn   (mhim_modules/masking.py:9-88)   -- integer index work, numpy
# --------------------------------------------------------------------------- #


def topk_indices(v: np.ndarray, k: int, largest: bool = True) -> np.ndarray:
    """Indices of the k largest (smallest) values, sorted, ties -> LOWEST INDEX FIRST.

    The reference calls torch.topk (masking.py:62) whose tie order is
    implementation-defined.  This is the build's stated tie
    contract (DESIGN.md "tie contract"): order by (value desc, index asc).
    NaNs are not expected (softmax outputs).
    """
    v = np.asarray(v)
    key = -v if largest else v
    order = np.argsort(key, kind="stable")        # stable => ties keep ascending index
    return order[:k].astype(np.int64)


def mask_count(ps: int, mask_ratio: float, random_ratio: float = 1.0):
    """(k, n_sel) exactly as masking.py:30-35,61,70 computes them in float64."""
    ratio_ori = mask_ratio
    mask_ratio = mask_ratio / random_ratio
    if mask_ratio > 1:
        random_ratio = ratio_ori
        mask_ratio = 1.0
    k = int(np.ceil(ps * mask_ratio))
    n_sel = int(np.ceil(k * random_ratio)) if random_ratio < 1.0 else k
    return k, n_sel, random_ratio


def select_mask(ps: int, attn: np.ndarray, largest: bool, mask_ratio: float,
                other_masked: Optional[np.ndarray] = None, random_ratio: float = 1.0,
                perm: Optional[np.ndarray] = None, select_inv: bool = False,
                msa_fusion: str = "vote", shrink_ps: bool = False):
    """Restatement of select_mask_fn.

    attn: [N] (2-D case in the reference: [1,N]) or [h,N] (3-D case: [1,h,N]).
    other_masked: indices already masked by an earlier call (the reference passes
        them as cls_attn_topk_idx_other, mhim.py:136-139,164-167).
    shrink_ps: reproduces the branch masking.py:37-40 (mask_ids_other given but no
        explicit index list) where k is computed from ps - |other|.
    perm: the injected torch.randperm(k) of masking.py:67 (required if random_ratio<1).
    Returns (len_keep, mask_ids[N] int64 = kept ascending ++ masked), masked list.
    Kept ids are emitted ASCENDING (the reference emits CPython set order, which is
    ascending for realistic ratios).
    """
    attn = np.asarray(attn)
    ps_tmp = ps - (len(other_masked) if (shrink_ps and other_masked is not None) else 0)
    ratio_ori = mask_ratio
    mask_ratio = mask_ratio / random_ratio
    if mask_ratio > 1:
        random_ratio = ratio_ori
        mask_ratio = 1.0
    k = int(np.ceil(ps_tmp * mask_ratio))
    if attn.ndim == 2:
        h = attn.shape[0]
        if msa_fusion == "mean":                                    # masking.py:44-48
            kk = int(np.ceil(ps_tmp * mask_ratio) // h)
            idx = np.concatenate([topk_indices(attn[i], kk, largest) for i in range(h)])
            top = np.unique(idx)
        elif msa_fusion == "vote":                                  # masking.py:49-59
            vote = np.zeros(attn.shape[1], dtype=np.float32)
            for i in range(h):
                vote[topk_indices(attn[i], k, largest)] += 1.0
            top = topk_indices(vote, k, True)
        else:
            raise ValueError(msa_fusion)
    else:
        top = topk_indices(attn, k, largest)                        # masking.py:61-63
    if random_ratio < 1.0:                                          # masking.py:66-71
        n_sel = int(np.ceil(top.shape[0] * random_ratio))
        assert perm is not None and perm.shape[0] == top.shape[0]
        top = top[np.asarray(perm)[:n_sel]]
    if other_masked is not None:                                    # masking.py:74-75
        top = np.unique(np.concatenate([top, np.asarray(other_masked, dtype=np.int64)]))
    len_keep = ps - top.shape[0]                                    # masking.py:77
    flag = np.zeros(ps, dtype=bool)
    flag[top] = True
    kept = np.nonzero(~flag)[0].astype(np.int64)                    # masking.py:78-80 (ascending contract)
    if select_inv:                                                  # masking.py:82-84
        return ps - len_keep, np.concatenate([top, kept]), top
    return len_keep, np.concatenate([kept, top]), top               # masking.py:86


def get_mask(ps, attn, mask_ratio=0.0, mask_ratio_l=0.0, mask_ratio_h=0.0, mask_ratio_hr=1.0,
             perms: Sequence[Optional[np.ndarray]] = (None, None, None), msa_fusion="vote"):
    """MHIM.get_mask (mhim.py:109-179) with select_inv=False.

    perms = (perm for the random v1 mask [random_ratio 0.001], None, perm for the HAM mask).
    """
    len_keep, mask_ids, masked = ps, None, None
    if attn is not None and mask_ratio > 0.0:                       # mhim.py:123-128
        len_keep, mask_ids, masked = select_mask(ps, attn, False, mask_ratio, random_ratio=0.001,
                                                 perm=perms[0], msa_fusion=msa_fusion)
    if attn is not None and mask_ratio_l > 0.0:                     # mhim.py:132-149
        len_keep, mask_ids, masked = select_mask(ps, attn, False, mask_ratio_l, other_masked=masked,
                                                 msa_fusion=msa_fusion)
    if mask_ratio_h > 0.0:                                          # mhim.py:158-177
        len_keep, mask_ids, masked = select_mask(ps, attn, True, mask_ratio_h, other_masked=masked,
                                                 random_ratio=mask_ratio_hr, perm=perms[2],
                                                 msa_fusion=msa_fusion)
    return len_keep, mask_ids


# --------------------------------------------------------------------------- #
# A8  Merge / MCA   (mhim_modules/merge.py)
# --------------------------------------------------------------------------- #


def mca(x, q_in, params, heads=8, dim_head=64, attn_mask=None, out_mask=None, p=0.0, prefix="merge.attn."):
    """Multi-head cross attention of m query tokens over n rows (merge.py:43-65).

    x [n,E], q_in [m,E] -> [m,E].  kv = x Wkv^T, k = first half, v = second half
    (merge.py:53 chunk), heads are contiguous 64-column groups (merge.py:56-57).
    """
    inner = heads * dim_head
    kv = _linear(x, params[prefix + "to_kv.weight"])
    k, v = kv[:, :inner], kv[:, inner:]
    q = _linear(q_in, params[prefix + "to_q.weight"])
    n, m = x.shape[0], q_in.shape[0]
    k = k.reshape(n, heads, dim_head).permute(1, 0, 2)
    v = v.reshape(n, heads, dim_head).permute(1, 0, 2)
    q = q.reshape(m, heads, dim_head).permute(1, 0, 2)
    dots = (q @ k.transpose(-1, -2)) * (dim_head ** -0.5)            # merge.py:59
    a = _drop(torch.softmax(dots, dim=-1), attn_mask, p)            # merge.py:61-62
    out = (a @ v).permute(1, 0, 2).reshape(m, inner)                # merge.py:64-65
    out = _linear(out, params[prefix + "to_out.0.weight"], params[prefix + "to_out.0.bias"])
    return _drop(out, out_mask, p)


def merge_tokens(x_rest, params, mm, training=True, **kw):
    """Merge.merge (merge.py:131-144): z = MCA(LN(x), LN(global_q)); returns (z, new global_q)."""
    g = params["merge.global_q_mm"].reshape(-1, x_rest.shape[-1])
    w, b = params["merge.norm.weight"], params["merge.norm.bias"]
    z = mca(_layer_norm(x_rest, w, b), _layer_norm(g, w, b), params, **kw)
    g_new = g * mm + z.detach() * (1.0 - mm) if training else g      # merge.py:127-129,142-143
    return z, g_new


def merge_train(x, ids_shuffle, params, merge_ratio, mm, **kw):
    """Merge.forward, training branch (merge.py:146-176,189-196).

    ids_shuffle: the injected argsort(rand(L)) of merge.py:164-165.
    Returns (tokens [int(L*r)+k, E], new global_q [k,E]).
    """
    L = x.shape[0]
    len_keep = int(L * merge_ratio)                                 # merge.py:163
    ids = torch.as_tensor(np.asarray(ids_shuffle), dtype=torch.long)
    z, g_new = merge_tokens(x[ids[len_keep:]], params, mm, True, **kw)
    return torch.cat([x[ids[:len_keep]], z], dim=0), g_new


def merge_eval(x, params, **kw):
    """Merge.forward, eval branch (merge.py:197-198): cat(x, merge(x)); no EMA update."""
    z, _ = merge_tokens(x, params, 0.0, False, **kw)
    return torch.cat([x, z], dim=0)


# --------------------------------------------------------------------------- #
# A11  predictor + SoftTargetCrossEntropy   (mhim.py:97; losses.py:26-45)
# --------------------------------------------------------------------------- #


def predictor(z, params):
    return _linear(z, params["predictor.weight"], params["predictor.bias"])


def soft_target_ce(student, teacher, temp_t=1.0, temp_s=1.0):
    """-sum_e softmax(t/temp_t)_e * log_softmax(s/temp_s)_e over the E feature dims."""
    return -(torch.softmax(teacher / temp_t, dim=-1) * torch.log_softmax(student / temp_s, dim=-1)).sum(dim=-1)


# --------------------------------------------------------------------------- #
# A9  Nystrom attention   (modules/nystrom_attention.py:12-27, 65-152)
# --------------------------------------------------------------------------- #


def pinv_iter(a, iters=6):
    """Iterative Moore-Penrose pseudo-inverse; the init scale is GLOBAL over heads (nystrom:15-18)."""
    absa = a.abs()
    z = a.transpose(-1, -2) / (absa.sum(dim=-1).max() * absa.sum(dim=-2).max())
    eye = torch.eye(a.shape[-1], dtype=a.dtype)
    for _ in range(iters):
        az = a @ z
        z = 0.25 * z @ (13 * eye - az @ (15 * eye - az @ (7 * eye - az)))
    return z


def nystrom_attention(x, params, prefix, heads=8, dim_head=64, landmarks=256, iters=6,
                      return_attn=False, no_norm=False, out_mask=None, p=0.0):
    """NystromAttention.forward on one bag x [n, dim] (attn_mask=None: the masked branch is dead).

    Front zero padding to a multiple of ``landmarks`` (nystrom:70-73); pad tokens are
    ordinary (unmasked) keys with q=k=v=0 because to_qkv has no bias.
    """
    n, dim = x.shape
    m = landmarks
    pad = (m - n % m) % m
    xp = torch.cat([x.new_zeros(pad, dim), x], dim=0) if pad else x
    npad = xp.shape[0]
    qkv = _linear(xp, params[prefix + "to_qkv.weight"])
    inner = heads * dim_head
    q, k, v = (qkv[:, i * inner:(i + 1) * inner].reshape(npad, heads, dim_head).permute(1, 0, 2) for i in range(3))
    q = q * dim_head ** -0.5                                        # nystrom:89
    l = math.ceil(n / m)                                            # nystrom:93
    ql = q.reshape(heads, m, l, dim_head).sum(dim=2) / l            # nystrom:95-109
    kl = k.reshape(heads, m, l, dim_head).sum(dim=2) / l
    s1 = q @ kl.transpose(-1, -2)                                   # [h, npad, m]
    s2 = ql @ kl.transpose(-1, -2)                                  # [h, m, m]
    s3 = ql @ k.transpose(-1, -2)                                   # [h, m, npad]
    a1, a2, a3 = (t.softmax(dim=-1) for t in (s1, s2, s3))
    a2i = pinv_iter(a2, iters)
    out = (a1 @ a2i) @ (a3 @ v)                                     # nystrom:132
    w = params[prefix + "res_conv.weight"]                          # [h,1,33,1] depthwise along tokens
    out = out + F.conv2d(v.unsqueeze(0), w, padding=(w.shape[2] // 2, 0), groups=heads).squeeze(0)
    out = out.permute(1, 0, 2).reshape(npad, inner)
    out = _drop(_linear(out, params[prefix + "to_out.0.weight"], params[prefix + "to_out.0.bias"]), out_mask, p)
    out = out[-n:]
    if not return_attn:
        return out
    if no_norm:                                                     # nystrom:127-129,147-149
        r = (s1[:, -n].unsqueeze(-2) @ pinv_iter(s2, iters)) @ s3
    else:
        r = (a1[:, -n].unsqueeze(-2) @ a2i) @ a3
    return out, r[:, 0, -n + 1:], v[:, -n + 1:]                     # cls-row attention [h, n-1], v [h, n-1, d]


def ppeg(x, params, prefix="online_encoder.pos_embedding."):
    """PPEG.forward (emb_position.py:92-120) on tokens x [N, C]."""
    n, c = x.shape
    hh = int(np.ceil(np.sqrt(n)))
    add = hh * hh - n
    x = torch.cat([x, x[:add]], dim=0)
    if hh < 7:
        z = 49 - (n + add)
        x = torch.cat([x, x.new_zeros(z, c)], dim=0)
        add += z
        hh = 7
    g = x.t().reshape(1, c, hh, hh)
    y = g
    for nm, kk in (("proj", 7), ("proj1", 5), ("proj2", 3)):
        y = y + F.conv2d(g, params[prefix + nm + ".weight"], params[prefix + nm + ".bias"], padding=kk // 2, groups=c)
    y = y.reshape(c, hh * hh).t()
    return y[:-add] if add > 0 else y


def trans_layer(x, params, prefix, need_attn=False, no_norm=False, **kw):
    """TransLayer.forward (baseline.py:210-220): x + Nystrom(LN(x))."""
    xn = _layer_norm(x, params[prefix + "norm.weight"], params[prefix + "norm.bias"])
    if need_attn:
        z, attn, v = nystrom_attention(xn, params, prefix + "attn.", return_attn=True, no_norm=no_norm, **kw)
        return x + z, attn, v
    return x + nystrom_attention(xn, params, prefix + "attn.", **kw)


def sattention(h, params, return_attn=False, no_norm=False, prefix="online_encoder.", drop=None, p=0.0):
    """SAttention.forward (baseline.py:244-288), pos='ppeg', pos_pos=0.

    drop = optional (mask_layer1, mask_layer2) for the two to_out dropouts.
    Returns feat [E] or (feat, [attn_l1, attn_l2], v_l1).
    """
    d1, d2 = drop if drop is not None else (None, None)
    x = torch.cat([params[prefix + "cls_token"].reshape(1, -1), h], dim=0)
    attn = []
    if return_attn:
        x, a, v = trans_layer(x, params, prefix + "layer1.", True, no_norm, out_mask=d1, p=p)
        attn.append(a)
    else:
        x = trans_layer(x, params, prefix + "layer1.", out_mask=d1, p=p)
    x = torch.cat([x[:1], ppeg(x[1:], params, prefix + "pos_embedding.")], dim=0)   # baseline.py:265-266
    if return_attn:
        x, a, _ = trans_layer(x, params, prefix + "layer2.", True, no_norm, out_mask=d2, p=p)
        attn.append(a)
    else:
        x = trans_layer(x, params, prefix + "layer2.", out_mask=d2, p=p)
    x = _layer_norm(x, params[prefix + "norm.weight"], params[prefix + "norm.bias"])
    return (x[0], attn, v) if return_attn else x[0]


def pseudo_score_trans(v, attn, params, to_out_prefix="online_encoder.layer1.attn.", out_mask=None, p=0.0):
    """get_pseudo_score_trans (scoring.py:9-34). v [h,n,d], attn [h,n] -> score [n]."""
    h, n, d = v.shape
    f = (v * attn[:, :, None]).permute(1, 0, 2).reshape(n, h * d)
    f = _drop(_linear(f, params[to_out_prefix + "to_out.0.weight"], params[to_out_prefix + "to_out.0.bias"]), out_mask, p)
    cam = params["predictor.weight"] @ f.t() + params["predictor.bias"][0]
    return torch.softmax(cam, dim=0).max(dim=0).values


# --------------------------------------------------------------------------- #
# A12  MHIM entry points   (modules/mhim.py)
# --------------------------------------------------------------------------- #


# --------------------------------------------------------------------------- #
# N1  DSMIL encoder   (mhim_modules/baseline.py:112-194)
# --------------------------------------------------------------------------- #


def dsmil(h, params, cls_attn=True, no_norm=False, prefix="online_encoder."):
    """DSMIL.attention (baseline.py:166-186) over BClassifier.forward (:133-157), dropout_v = 0.
    h [N,E] -> (logits_bag [C], logits_ins [C], attn [N], B [C,E], A [N,C] (softmax unless no_norm))."""
    P = lambda n: params[prefix + n]
    classes = _linear(h, P("i_classifier.0.weight"), P("i_classifier.0.bias"))                   # [N,C]
    b = prefix + "b_classifier."
    qnet = lambda t: torch.tanh(_linear(torch.relu(_linear(t, params[b + "q.0.weight"], params[b + "q.0.bias"])),
                                        params[b + "q.2.weight"], params[b + "q.2.bias"]))
    V = torch.relu(_linear(h, params[b + "v.1.weight"], params[b + "v.1.bias"]))                  # [N,E]
    Q = qnet(h)                                                                                   # [N,128]
    _, m_indices = torch.sort(classes, 0, descending=True)                                        # baseline.py:139
    m_feats = h[m_indices[0]]                                                                     # critical instance per class [C,E]
    q_max = qnet(m_feats)                                                                         # [C,128]
    A_raw = (Q @ q_max.t()) / math.sqrt(Q.shape[1])                                               # [N,C]
    A = torch.softmax(A_raw, 0)
    B = A.t() @ V                                                                                 # [C,E]
    Cc = torch.einsum("ocv,cv->o", params[b + "fcc.weight"], B) + params[b + "fcc.bias"]          # Conv1d(C,C,kernel=E) on [1,C,E]
    classes_bag = classes.max(0).values
    attn = classes.max(-1).values if cls_attn else A.max(-1).values                               # attn_index == 'max'
    return Cc, classes_bag, attn, B, (A_raw if no_norm else A)


class Cfg:
    """Hyper-parameters of MHIM.__init__ (mhim.py:22-27) that the functions below read."""

    def __init__(self, act="relu", da_act="gelu", baseline="attn", dropout=0.0, mask_ratio=0.0,
                 mask_ratio_l=0.0, mask_ratio_h=0.0, mask_ratio_hr=1.0, attn2score=True,
                 merge_enable=True, merge_k=1, merge_mm=0.9998, merge_ratio=0.0, merge_test=False,
                 temp_t=1.0, gated=False, msa_fusion="vote", mca_dropout=0.0):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def _encode(h, params, cfg: Cfg, return_attn=False, no_norm=False):
    if cfg.baseline == "attn":
        z, a, act = dattention(h, params, cfg.da_act, cfg.gated, no_norm)
        return (z, a, act) if return_attn else z
    if cfg.baseline == "selfattn":
        return sattention(h, params, return_attn, no_norm)
    raise ValueError(cfg.baseline)            # 'dsmil' returns two logit vectors: handled by the entry points themselves


def forward_teacher(x, params, cfg: Cfg, drop_mask=None):
    """MHIM.forward_teacher (mhim.py:181-227).  x [N,D] -> (feat [E], score [N] or [h,N])."""
    h = feature(x, params, cfg.act, drop_mask, cfg.dropout)
    p0 = h.shape[0]
    if cfg.merge_test:                                              # mhim.py:196-200
        h = merge_eval(h, params)
    if cfg.baseline == "dsmil":                                     # mhim.py:202-205: feat = B [C,E], score = max-class logits
        _, _, attn, B, _ = dsmil(h, params, cls_attn=cfg.attn2score)
        return B, (attn[:p0] if cfg.merge_test else attn)
    z, attn, act = _encode(h, params, cfg, return_attn=True)
    if cfg.merge_test:                                              # mhim.py:207-213
        attn = [a[:, :p0] for a in attn] if isinstance(attn, list) else attn[:p0]
    if cfg.attn2score:                                              # mhim.py:215-222
        if cfg.baseline == "selfattn":
            attn = pseudo_score_trans(act, attn[0], params)
        else:
            attn = pseudo_score(act, attn, params["predictor.weight"], params["predictor.bias"])
    elif isinstance(attn, list):
        attn = attn[0]                                              # mhim.py:224-225 (attn_layer = 0)
    return z, attn


def forward_test(x, params, cfg: Cfg, return_attn=False, no_norm=False):
    """MHIM.forward_test (mhim.py:229-272), eval mode (dropout = identity)."""
    h = feature(x, params, cfg.act)
    if cfg.merge_test:
        h = merge_eval(h, params)
    if cfg.baseline == "dsmil":                                     # mhim.py:257-258,264-265: [bag logits, max instance logits]
        lb, li, a, B, _ = dsmil(h, params, cls_attn=cfg.attn2score, no_norm=no_norm)
        return ([lb, li], a) if return_attn else ([lb, li], B)      # without return_attn the encoder's tuple passes through (:262)
    if return_attn:
        z, a, _ = _encode(h, params, cfg, True, no_norm)
        return predictor(z, params), a
    return predictor(_encode(h, params, cfg), params)


def pure(x, params, cfg: Cfg, drop_mask=None):
    """MHIM.pure (mhim.py:274-298): logits [C]."""
    h = feature(x, params, cfg.act, drop_mask, cfg.dropout)
    if cfg.baseline == "dsmil":                                     # mhim.py:289-290
        lb, li, _, _, _ = dsmil(h, params, cls_attn=cfg.attn2score)
        return [lb, li]
    return predictor(_encode(h, params, cfg), params)


def forward_student(x, params, cfg: Cfg, attn, teacher_feat=None, perm=None, ids_shuffle=None,
                    drop_mask=None, mrh=None, perms=None):
    """MHIM.forward (mhim.py:318-378).

    attn: teacher score, numpy/torch [N] or [h,N].  perm: injected randperm(k) for the HAM
    subsample; ids_shuffle: injected argsort(rand(L)) for Merge.masking.
    Returns (logits [C], cls_loss scalar-or-0., ps, len_keep_after_merge, extras dict).
    """
    h = feature(x, params, cfg.act, drop_mask, cfg.dropout)
    ps = h.shape[0]
    a_np = attn.detach().cpu().numpy() if torch.is_tensor(attn) else np.asarray(attn)
    mr_h = cfg.mask_ratio_h if mrh is None else mrh
    len_keep, mask_ids = get_mask(ps, a_np, cfg.mask_ratio, cfg.mask_ratio_l, mr_h, cfg.mask_ratio_hr,
                                  perms if perms is not None else (None, None, perm), cfg.msa_fusion)
    ids_keep = torch.as_tensor(mask_ids[:len_keep], dtype=torch.long)
    hk = h[ids_keep]                                                # masking.py:107
    g_new = None
    if cfg.merge_enable:
        hk, g_new = merge_train(hk, ids_shuffle, params, cfg.merge_ratio, cfg.merge_mm)
    if cfg.baseline == "dsmil":                                     # mhim.py:355-364: distillation on B [C,E], mean over classes
        lb, li, _, B, _ = dsmil(hk, params, cls_attn=cfg.attn2score)
        cls_loss = soft_target_ce(B, teacher_feat.detach(), cfg.temp_t).mean() if teacher_feat is not None else 0.0
        return [lb, li], cls_loss, ps, hk.shape[0], {"len_keep_mask": len_keep, "mask_ids": mask_ids, "global_q_new": g_new,
                                                     "feat": B}
    z = _encode(hk, params, cfg)
    logits = predictor(z, params)
    cls_loss = soft_target_ce(z, teacher_feat.detach(), cfg.temp_t) if teacher_feat is not None else 0.0
    return logits, cls_loss, ps, hk.shape[0], {"len_keep_mask": len_keep, "mask_ids": mask_ids,
                                               "global_q_new": g_new, "feat": z}


def forward_student_eval(x, params, cfg: Cfg, attn, teacher_feat=None, perm=None):
    """MHIM.forward with the module in eval mode (mhim.py:318-378): the mask is applied whatever the mode; Merge's eval branch keeps
    every surviving row and appends the k tokens merged from all of them (merge.py:197-198); no EMA, no dropout."""
    h = feature(x, params, cfg.act, None, 0.0)
    ps = h.shape[0]
    a_np = attn.detach().cpu().numpy() if torch.is_tensor(attn) else np.asarray(attn)
    len_keep, mask_ids = get_mask(ps, a_np, cfg.mask_ratio, cfg.mask_ratio_l, cfg.mask_ratio_h, cfg.mask_ratio_hr, (None, None, perm),
                                  cfg.msa_fusion)
    hk = merge_eval(h[torch.as_tensor(mask_ids[:len_keep], dtype=torch.long)], params)
    z = _encode(hk, params, cfg)
    logits = predictor(z, params)
    cls_loss = soft_target_ce(z, teacher_feat.detach(), cfg.temp_t) if teacher_feat is not None else 0.0
    return logits, cls_loss, ps, hk.shape[0]


# --------------------------------------------------------------------------- #
# A14  trainer step semantics   (engines/base_engine.py:76-134,151,155-167; train_utils.py:58-65)
# --------------------------------------------------------------------------- #


def cross_entropy(logits, label: int):
    return -torch.log_softmax(logits, dim=-1)[label]


def adam_step(p, g, m, v, step, lr=2e-4, b1=0.9, b2=0.999, eps=1e-8, wd=1e-5):
    """torch.optim.Adam (L2 weight decay folded into the gradient), one tensor, step >= 1."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mh = m / (1 - b1 ** step)
    vh = v / (1 - b2 ** step)
    return p - lr * mh / (vh.sqrt() + eps), m, v


def ema_update(teacher, student, mm):
    """base_engine.py:166-167: pk <- mm*pk + (1-mm)*pq for every parameter, incl. merge.global_q_mm."""
    return {k: teacher[k] * mm + student[k] * (1.0 - mm) for k in teacher}


TRAINABLE_EXCLUDE = ("merge.global_q_mm",)                            # requires_grad=False (merge.py:108)


def clip_grad_norm(grads, max_norm):
    """--clip_grad (base_engine.py:115-119 -> timm dispatch_clip_grad(mode='norm') -> torch.nn.utils.clip_grad_norm_, 2-norm):
    total = || all gradients ||_2, every gradient *= min(1, max_norm / (total + 1e-6)).  Returns (clipped dict, total norm)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: g * coef for k, g in grads.items()}, float(total)


def train_step(x, label, stu, tea, opt_state, cfg: Cfg, step, perm=None, ids_shuffle=None,
               aux_alpha=0.5, main_alpha=1.0, mm=0.9997, lr=2e-4, wd=1e-5, model="mhim", score_override=None, clip_grad=None,
               drop_mask_t=None, drop_mask_s=None):
    """One CommonMIL.forward_func + BaseTrainer step (accumulation 1).  Dropout (cfg.dropout > 0) takes injected keep-masks [N, E]
    for the teacher's and the student's feature dropout (mhim.py:76; the trainer keeps the teacher in train mode, base_engine.py:37-38).

    stu/tea: dicts of fp32 tensors (reference key names).  opt_state: {name: (m, v)}.
    Returns (new_stu, new_tea, new_opt_state, info).
    """
    stu_g = {k: v.clone().requires_grad_(k not in TRAINABLE_EXCLUDE) for k, v in stu.items()}
    if model == "mhim":
        with torch.no_grad():
            t_feat, score = forward_teacher(x, tea, cfg, drop_mask_t)
        if score_override is not None:      # tests: select on the scores the device saw (its random subsets are read back as perm)
            score = score_override
        t_in = None if aux_alpha == 0.0 else t_feat                 # common_mil.py:24
        logits, cls_loss, ps, keep, ex = forward_student(x, stu_g, cfg, score, t_in, perm, ids_shuffle, drop_mask=drop_mask_s)
    else:
        logits, cls_loss, ps, keep, ex = pure(x, stu_g, cfg), 0.0, x.shape[0], x.shape[0], {}
    if isinstance(logits, list):                                    # dsmil: common_mil.py:26-28 mixes the two logit vectors
        logits = 0.5 * logits[0] + 0.5 * logits[1]
    loss = main_alpha * cross_entropy(logits, label) + aux_alpha * cls_loss   # base_engine.py:99-100
    loss.backward()
    new_stu, new_opt = {}, {}
    raw = {k: p.grad.detach().clone() for k, p in stu_g.items() if p.grad is not None}
    if clip_grad is not None:                                       # base_engine.py:115-119, right before optimizer.step()
        clipped, _ = clip_grad_norm(raw, clip_grad)
        for k, p in stu_g.items():
            if p.grad is not None:
                p.grad = clipped[k]
    for k, p in stu_g.items():
        if p.grad is None:
            new_stu[k] = p.detach()
            continue
        m, v = opt_state.get(k, (torch.zeros_like(p), torch.zeros_like(p)))
        pn, m, v = adam_step(p.detach(), p.grad, m, v, step, lr=lr, wd=wd)
        new_stu[k], new_opt[k] = pn, (m, v)
    if ex.get("global_q_new") is not None:                          # in-forward EMA of the global queries
        new_stu["merge.global_q_mm"] = ex["global_q_new"].reshape(stu["merge.global_q_mm"].shape).detach()
    new_tea = ema_update(tea, new_stu, mm) if model == "mhim" else tea
    info = {"teacher_score": score if model == "mhim" else None, "loss": float(loss.detach()), "logits": logits.detach(), "cls_loss": float(cls_loss.detach() if torch.is_tensor(cls_loss) else cls_loss), "ps": ps, "keep": keep,
            "grads": raw}
    if "mask_ids" in ex:                                            # the rows the student kept (masking.py:107), before Merge's shuffle
        info["rows"] = np.asarray(ex["mask_ids"][:ex["len_keep_mask"]])
    return new_stu, new_tea, new_opt, info


def train_window(xs, labels, stu, tea, opt_state, cfg: Cfg, step, perms=None, shuffles=None, aux_alpha=0.5, main_alpha=1.0, mm=0.9997,
                 lr=2e-4, wd=1e-5, q_ema="sequential", score_overrides=None, clip_grad=None):
    """One optimiser update over an accumulation window of len(xs) bags (--accumulation_steps; base_engine.py:29,47-49,100-102,
    146-153): every bag's loss is divided by the window length and back-propagated into the same gradients, then ONE Adam step
    and ONE EMA-teacher update (need_update, base_engine.py:47,109-119,155-167).  Dropout off.

    q_ema: how the in-forward EMA of Merge's global queries (merge.py:142-143) composes inside the window:
      "sequential"  the reference's single process: bag j sees the queries as left by bag j-1;
      "window"      the batched contract: every bag of the window attends with the window's FIRST queries q0 (the
                    k bags run through the same launches, nothing orders them), and the window ends with the same chain of EMA steps
                    applied to the tokens z_j those forwards produced:  q <- mm^k q0 + (1 - mm) sum_j mm^(k-1-j) z_j.  Differs from
                    "sequential" only through d z_j / d q over a query drift of <= k (1 - mm): second order in (1 - mm).
    Returns (new_stu, new_tea, new_opt_state, info) with per-bag lists in info.
    """
    k = len(xs)
    stu_g = {n: v.clone().requires_grad_(n not in TRAINABLE_EXCLUDE) for n, v in stu.items()}
    q0 = stu["merge.global_q_mm"].clone() if "merge.global_q_mm" in stu else None
    q_news, infos = [], {"logits": [], "loss": [], "cls_loss": [], "rows": [], "teacher_score": []}
    for j, x in enumerate(xs):
        with torch.no_grad():
            t_feat, score = forward_teacher(x, tea, cfg)
        if score_overrides is not None and score_overrides[j] is not None:
            score = score_overrides[j]
        t_in = None if aux_alpha == 0.0 else t_feat
        logits, cls_loss, ps, keep, ex = forward_student(x, stu_g, cfg, score, t_in, None if perms is None else perms[j],
                                                         None if shuffles is None else shuffles[j])
        loss = (main_alpha * cross_entropy(logits, int(labels[j])) + aux_alpha * cls_loss) / k       # base_engine.py:99-102
        loss.backward()
        infos["logits"].append(logits.detach()); infos["loss"].append(float(loss.detach()) * k)
        infos["cls_loss"].append(float(cls_loss.detach()) if torch.is_tensor(cls_loss) else float(cls_loss))
        infos["teacher_score"].append(score)
        if "mask_ids" in ex:
            infos["rows"].append(np.asarray(ex["mask_ids"][:ex["len_keep_mask"]]))
        if ex.get("global_q_new") is not None:
            qn = ex["global_q_new"].reshape(q0.shape).detach()
            q_news.append(qn)
            if q_ema == "sequential":
                with torch.no_grad():
                    stu_g["merge.global_q_mm"].copy_(qn)
    new_stu, new_opt = {}, {}
    infos["grads"] = {n: p.grad.detach().clone() for n, p in stu_g.items() if p.grad is not None}
    if clip_grad is not None:
        clipped, _ = clip_grad_norm(infos["grads"], clip_grad)
        for n, p in stu_g.items():
            if p.grad is not None:
                p.grad = clipped[n]
    for n, p in stu_g.items():
        if p.grad is None:
            new_stu[n] = p.detach()
            continue
        m, v = opt_state.get(n, (torch.zeros_like(p), torch.zeros_like(p)))
        pn, m, v = adam_step(p.detach(), p.grad, m, v, step, lr=lr, wd=wd)
        new_stu[n], new_opt[n] = pn, (m, v)
    if q_news:
        if q_ema == "sequential":
            new_stu["merge.global_q_mm"] = q_news[-1]
        else:
            g_mm = float(cfg.merge_mm)
            q = q0.double()
            for qn in q_news:                                # z_j = (q_new_j - mm q0) / (1 - mm): the tokens of bag j (merge.py:142)
                z = (qn.double() - g_mm * q0.double()) / (1.0 - g_mm)
                q = g_mm * q + (1.0 - g_mm) * z
            new_stu["merge.global_q_mm"] = q.float()
    new_tea = ema_update(tea, new_stu, mm)
    return new_stu, new_tea, new_opt, infos
