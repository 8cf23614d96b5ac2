"""mhim_modules/masking.py on the device: ``select_mask_fn`` (masking.py:9-88) with every option of the reference's signature - the two
multi-head fusions ('vote', which MHIM uses, and 'mean'), ``select_inv``, the union with an earlier mask - and ``mask_fn``
(masking.py:91-110).  MHIM.get_mask (mhim.py) composes the same kernels for the options its constructor fixes (msa_fusion='vote',
select_inv=False: mhim.py:59-60); this module is the reference's free-function interface.

Tie contract (include/mhimx.h, select): candidates are ordered by (value, index ascending); kept ids are emitted ascending - the
reference's order is torch.topk's and CPython's set order, both implementation-defined.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from . import ops


def select_mask_fn(ps, attn, largest, mask_ratio, mask_ids_other=None, len_keep_other=None, cls_attn_topk_idx_other=None,
                   random_ratio=1., select_inv=False, msa_fusion="vote", perm=None, generator=None):
    """Returns (len_keep: int, mask_ids [1, ps] int64 = kept ascending ++ masked; with select_inv the selected rows first and their count).
    ``perm``: the torch.randperm draw of masking.py:67 (parity tests); otherwise drawn from ``generator`` on the device."""
    if not (torch.is_tensor(attn) and attn.is_cuda):
        raise L.MhimxError("select_mask_fn: attn must be a GPU tensor (the HIP path has no CPU fallback)")
    dev = attn.device
    ps_tmp = ps
    ratio_ori = mask_ratio
    mask_ratio = mask_ratio / random_ratio
    if mask_ratio > 1:                                                   # masking.py:32-34
        random_ratio = ratio_ori
        mask_ratio = 1.
    other = None
    if mask_ids_other is not None:                                       # masking.py:36-39,74-75: the union happens iff mask_ids_other is given
        if cls_attn_topk_idx_other is None:
            cls_attn_topk_idx_other = mask_ids_other.reshape(-1)[len_keep_other:]
            ps_tmp = ps - cls_attn_topk_idx_other.numel()
        other = cls_attn_topk_idx_other.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
    k = int(np.ceil(ps_tmp * mask_ratio))
    heads = attn.dim() == 3 or (attn.dim() == 2 and attn.shape[0] != 1)
    if heads:
        a = attn.reshape(-1, attn.shape[-1]).contiguous().float()
        if msa_fusion == "mean":                                         # masking.py:44-48: per-head top-(k // h), their sorted union
            kk = int(np.ceil(ps_tmp * mask_ratio) // a.shape[0])
            if kk < 1:
                raise L.MhimxError("select_mask_fn(msa_fusion='mean'): the ratio leaves no candidate per head")
            score = ops.vote_scores(a, kk, largest).clamp_(max=1.0)      # 1 on the union: its ascending order is the candidate order
            kc, lg = int(score.sum().item()), True
        elif msa_fusion == "vote":                                       # masking.py:49-59
            score, kc, lg = ops.vote_scores(a, k, largest), k, True
        else:
            raise ValueError(msa_fusion)
    else:
        score, kc, lg = attn.reshape(-1).contiguous().float(), k, bool(largest)
    n_sel = kc
    pm = None
    if random_ratio < 1.:                                                # masking.py:66-71
        n_sel = int(np.ceil(kc * random_ratio))
        if perm is None:
            pm = torch.randperm(kc, device=dev, generator=generator)
        else:
            pm = perm if torch.is_tensor(perm) else torch.as_tensor(np.asarray(perm), dtype=torch.int64)
            pm = pm.to(dev).contiguous()
    ids, lk_dev, _ = ops.select_mask(score, kc, n_sel, lg, pm, other=other)
    len_keep = ps - n_sel if other is None else int(lk_dev.item())      # masking.py:74-77 (a union's size is data dependent)
    if select_inv:                                                       # masking.py:82-84
        ids = torch.cat([ids[len_keep:], ids[:len_keep]])
        len_keep = ps - len_keep
    return len_keep, ids.view(1, -1)


def mask_fn(x, ids_shuffle=None, len_keep=None):
    """x [1, L, D] -> the rows ids_shuffle[:, :len_keep] (masking.py:91-110)."""
    assert ids_shuffle is not None
    if x.dim() != 3 or x.shape[0] != 1:
        raise L.MhimxError("mask_fn: x is one bag [1, L, D]")
    rows = ids_shuffle.reshape(-1)[:len_keep].to(torch.int64).contiguous()
    return ops.shard_gather(x[0].contiguous(), rows, 0, x.shape[1]).unsqueeze(0)
