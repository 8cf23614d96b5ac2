"""mhim_modules/scoring.py on the device: the reference's two free functions with their argument shapes.  MHIM.forward_teacher computes the
same scores inside its pool / encoder launches (the class projections ride in the scorer pass, the score is written by the pool's finalize
launch; `_trans_score` for the TransMIL teacher); these are the stand-alone forms for callers of the reference's interface.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops


def _classifier_wb(classifier):
    ps = list(classifier.parameters())                      # scoring.py:27,29 / 51,53: the last two parameters = weight [C, E], bias [C]
    if len(ps) < 2:
        raise L.MhimxError("get_pseudo_score: the classifier needs a weight and a bias")
    return ps[-2].detach().contiguous().float(), ps[-1].detach().contiguous().float()


def _scaled(v, attn, dh):
    """out[t, c] = v[t, c] * attn[c // dh, t]   (mhimx_scale_heads)"""
    n, C = v.shape
    out = torch.empty((n, C), device=v.device)
    L.check(L.lib().mhimx_scale_heads(ops._stream(), ops._p(v), v.stride(0), ops._p(attn), attn.stride(0), int(dh), n, C, ops._p(out)),
            "mhimx_scale_heads")
    return out


def get_pseudo_score(classifier, feat, attention):
    """scoring.py:37-58.  feat [1, n, d], attention [1, n] (already normalised) -> [1, n]:
    max_c softmax_c((A_n feat_n) . W_c + b[0])   (the class-0 bias on every class, as the reference has it)."""
    if not (torch.is_tensor(feat) and feat.is_cuda):
        raise L.MhimxError("get_pseudo_score: feat must be a GPU tensor (the HIP path has no CPU fallback)")
    w, b = _classifier_wb(classifier)
    f = feat.reshape(-1, feat.shape[-1]).contiguous().float()
    a = attention.reshape(1, -1).contiguous().float()
    cam = ops.gemm_nt(_scaled(f, a, f.shape[1]), w, prec="bf16x3")
    return ops.pseudo_score(None, None, cam, b).view(1, -1)


def get_pseudo_score_trans(classifier, feat, attention, to_out):
    """scoring.py:9-34.  feat [1, h, n, d] (the layer's v per head), attention [1, h, n] (the cls token's attention row per head),
    to_out: the attention block's output projection (its Linear, or the Sequential(Linear, Dropout) the reference passes - the dropout acts
    only if that module is in training mode, as there) -> [1, n]."""
    if not (torch.is_tensor(feat) and feat.is_cuda):
        raise L.MhimxError("get_pseudo_score_trans: feat must be a GPU tensor (the HIP path has no CPU fallback)")
    w, b = _classifier_wb(classifier)
    _, h, n, d = feat.shape
    v = feat[0].permute(1, 0, 2).reshape(n, h * d).contiguous().float()              # 'h n d -> n (h d)' (scoring.py:24)
    a = attention.reshape(h, n).contiguous().float()
    f = _scaled(v, a, d)
    lin = to_out[0] if isinstance(to_out, torch.nn.Sequential) else to_out
    p = 0.0
    if isinstance(to_out, torch.nn.Sequential) and to_out.training:
        for m in list(to_out)[1:]:
            p = max(p, float(getattr(m, "p", 0.0)))
    bias = None if lin.bias is None else lin.bias.detach().contiguous().float()
    f = ops.gemm_nt(f, lin.weight.detach().contiguous().float(), bias=bias, drop_p=p,
                    drop_seed=int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0, prec="bf16x3")
    cam = ops.gemm_nt(f, w, prec="bf16x3")
    return ops.pseudo_score(None, None, cam, b).view(1, -1)
