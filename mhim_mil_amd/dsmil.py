"""DSMIL encoder over the HIP primitives.

Mirrors the reference's ``DSMIL`` / ``BClassifier`` (mhim_modules/baseline.py:112-194) as used inside MHIM
(modules/mhim.py:91-95,202-205,257-258,289-290,355-364): same parameter names and shapes, same math —

    classes = i_classifier(h)                         [M,C]   instance logits
    V = relu(v.1(h)),  Q = tanh(q.2(relu(q.0(h))))    [M,E], [M,128]
    critical instance per class = arg max_m classes[m,c]  ->  q_max = q(h[critical])   [C,128]
    A = softmax_m(Q q_max^T / sqrt(128)),  B = A^T V  [C,E],  bag logits = Conv1d(C,C,E)(B)
    returns ([bag logits, max_m classes], B) and, as the per-instance score, max_c classes[m,c] (cls_attn)

Every arithmetic step is a kernel of libmhimx.so wrapped in a ``torch.autograd.Function`` (GEMMs with fused bias +
activation, row softmax, column max / arg-max); torch contributes graph bookkeeping, views and the C-row gather/scatter of
the critical instances.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch import nn

from . import _lib as L
from . import nystrom as NY
from . import ops

QDIM = 128


class LinearAct(torch.autograd.Function):
    """y = act(x[rows] W^T + b), act in {none, relu, tanh} fused into the GEMM epilogue; backward through the activation,
    then dX = dPre W, dW = dPre^T x[rows], db = column sums."""

    @staticmethod
    def forward(ctx, x, w, b, act, rows):
        x = x.contiguous()
        need_pre = act == L.ACT["tanh"]
        M = x.shape[0] if rows is None else rows.shape[0]
        pre = torch.empty((M, w.shape[0]), device=x.device) if need_pre else None
        y = ops.gemm_nt(x, w, rows=rows, bias=b, act=act, pre=pre, prec=NY._PREC)
        ctx.save_for_backward(x, w, y, pre, rows)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, pre, rows = ctx.saved_tensors
        g = dy.contiguous().clone()
        if ctx.act != 0:
            ops.act_bwd(g, y, pre, ctx.act)                                   # in place: g = dy * act'(pre)
        M, N = g.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dsub = torch.empty((M, x.shape[1]), device=x.device)
            NY._gemm("nn", g, 0, N, w, 0, w.shape[1], dsub, 0, x.shape[1], M, x.shape[1], N)
            if rows is None:
                dx = dsub
            else:                                                             # C critical rows: scatter back
                dx = torch.zeros_like(x)
                dx.index_add_(0, rows, dsub)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_tn(g, x, rows=rows, splits=min(64, max(8, M // 256)) if M >= 4096 else 1, prec=NY._PREC)   # (few output tiles: split the long reduction over the chip)
        if ctx.needs_input_grad[2]:
            db = ops.colsum(g)
        return dx, dw, db, None, None


class ColMax(torch.autograd.Function):
    """(max_m x[m,c], arg max) per class; the gradient goes to the arg-max rows."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        M, Cc = x.shape
        vals = torch.empty(Cc, device=x.device)
        idx = torch.empty(Cc, device=x.device, dtype=torch.int64)
        L.check(L.lib().mhimx_colmax(ops._stream(), ops._p(x), M, Cc, ops._p(vals), ops._p(idx)), "mhimx_colmax")
        ctx.save_for_backward(idx)
        ctx.shape = (M, Cc)
        ctx.mark_non_differentiable(idx)
        return vals, idx

    @staticmethod
    def backward(ctx, dv, _):
        (idx,) = ctx.saved_tensors
        dx = torch.zeros(ctx.shape, device=dv.device)
        dx[idx, torch.arange(ctx.shape[1], device=dv.device)] = dv
        return dx


class SoftmaxCols(torch.autograd.Function):
    """softmax over the M instances of every column of x [M,C] (baseline.py:147)."""

    @staticmethod
    def forward(ctx, x, alpha):
        x = x.contiguous()
        y = torch.empty_like(x)
        L.check(L.lib().mhimx_softmax_cols(ops._stream(), ops._p(x), ops._p(y), x.shape[0], x.shape[1], float(alpha)), "mhimx_softmax_cols")
        ctx.save_for_backward(y)
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        L.check(L.lib().mhimx_softmax_cols_bwd(ops._stream(), ops._p(y), ops._p(dy), ops._p(dx), y.shape[0], y.shape[1], float(ctx.alpha)),
                "mhimx_softmax_cols_bwd")
        return dx, None


def rowmax(x):
    x = x.contiguous()
    out = torch.empty(x.shape[0], device=x.device)
    L.check(L.lib().mhimx_rowmax(ops._stream(), ops._p(x), x.shape[0], x.shape[1], ops._p(out)), "mhimx_rowmax")
    return out


class SoftTargetCE(torch.autograd.Function):
    """cl = mean_c SoftTargetCrossEntropy(Bs[c], Bt[c]) over the feature dims (mhim.py:360-362, losses.py:26-45)."""

    @staticmethod
    def forward(ctx, Bs, Bt, temp_t):
        Bs, Bt = Bs.contiguous(), Bt.contiguous()
        Cc, V = Bs.shape
        losses = torch.empty(3, device=Bs.device)
        L.check(L.lib().mhimx_dsmil_head(ops._stream(), None, None, None, ops._p(Bs), ops._p(Bt), Cc, V, float(temp_t), 0.0, 1.0, 1.0,
                                         ops._p(losses), None, None, None, None), "mhimx_dsmil_head")
        ctx.save_for_backward(Bs, Bt)
        ctx.temp_t = temp_t
        return losses[2].clone()

    @staticmethod
    def backward(ctx, g):
        Bs, Bt = ctx.saved_tensors
        Cc, V = Bs.shape
        gB, losses = torch.empty_like(Bs), torch.empty(3, device=Bs.device)
        gin = g.contiguous().view(1).float()
        L.check(L.lib().mhimx_dsmil_head(ops._stream(), None, None, None, ops._p(Bs), ops._p(Bt), Cc, V, float(ctx.temp_t), 0.0, 1.0, 1.0,
                                         ops._p(losses), None, None, ops._p(gB), ops._p(gin)), "mhimx_dsmil_head")
        return gB, None, None


def dsmil_head(lb, li, label, Bs, Bt, temp_t, main_alpha, aux_alpha, inv_accum=1.0):
    """Fused-trainer form: loss and the gradients of (bag logits, max-instance logits, B) in one launch."""
    Cc, V = Bs.shape
    dev = Bs.device
    losses, g_lb, g_li, g_B = torch.empty(3, device=dev), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev), torch.empty_like(Bs)
    L.check(L.lib().mhimx_dsmil_head(ops._stream(), ops._p(lb), ops._p(li), ops._p(label), ops._p(Bs), ops._p(Bt), Cc, V, float(temp_t),
                                     float(main_alpha), float(aux_alpha), float(inv_accum), ops._p(losses), ops._p(g_lb), ops._p(g_li),
                                     ops._p(g_B), None), "mhimx_dsmil_head")
    return losses, g_lb, g_li, g_B


# --------------------------------------------------------------------------------------------------- modules
class _Lin(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.zeros(o))
        nn.init.xavier_normal_(self.weight)                       # mhim_modules/utils.py:16-19


class _Slot(nn.Module):
    pass


class _Conv1d(nn.Module):
    """nn.Conv1d(C, C, kernel_size=E) parameters with its default init (kaiming uniform, fan_in = C*E)."""

    def __init__(self, c, e):
        super().__init__()
        w = torch.empty(c, c, e)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(c * e)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.empty(c).uniform_(-bound, bound))


class _BClassifier(nn.Module):
    def __init__(self, e, c):
        super().__init__()
        self.q = nn.Sequential(_Lin(e, QDIM), _Slot(), _Lin(QDIM, QDIM), _Slot())
        self.v = nn.Sequential(_Slot(), _Lin(e, e), _Slot())
        self.fcc = _Conv1d(c, e)


class DSMIL(nn.Module):
    """mhim_modules/baseline.DSMIL (attn_index='max')."""

    def __init__(self, n_classes=2, mlp_dim=512, cls_attn=True):
        super().__init__()
        if n_classes > 16:
            raise L.MhimxError("DSMIL kernels handle up to 16 classes")
        self.i_classifier = nn.Sequential(_Lin(mlp_dim, n_classes))
        self.b_classifier = _BClassifier(mlp_dim, n_classes)
        self.cls_attn = cls_attn
        self.n_classes, self.mlp_dim = n_classes, mlp_dim

    def _q(self, t, rows=None):
        q = self.b_classifier.q
        h1 = LinearAct.apply(t, q[0].weight, q[0].bias, L.ACT["relu"], rows)
        return LinearAct.apply(h1, q[2].weight, q[2].bias, L.ACT["tanh"], None)

    def forward(self, h, want_attn=False, no_norm=False):
        """h [M,E] -> (bag logits [C], max-instance logits [C], B [C,E], attn [M] | None)."""
        M, E = h.shape
        Cc = self.n_classes
        ic = self.i_classifier[0]
        classes = NY.Linear.apply(h, ic.weight, ic.bias, 0.0, 0, None)                          # [M,C]
        v = self.b_classifier.v[1]
        V = LinearAct.apply(h, v.weight, v.bias, L.ACT["relu"], None)                           # [M,E]  (Dropout(0) before it)
        Q = self._q(h)                                                                          # [M,128]
        logits_ins, crit = ColMax.apply(classes)                                                # [C], critical rows
        q_max = self._q(h, rows=crit)                                                           # [C,128]
        # A [M,C] keeps the instances as ROWS: every reduction over M is then a TN GEMM (any M, split over slabs) and the
        # K = C contractions of the backward are a couple of FMAs per output
        a_raw = NY.heads_mm(Q, q_max, "nt", (0, 0, QDIM, M, QDIM), (0, 0, QDIM, Cc, QDIM), (M, Cc), (0, 0, Cc, M, Cc), heads=1)
        A = SoftmaxCols.apply(a_raw, 1.0 / math.sqrt(QDIM))                                     # softmax over the M instances
        B = NY.heads_mm(A, V, "tn", (0, 0, Cc, M, Cc), (0, 0, E, M, E), (Cc, E), (0, 0, E, Cc, E), heads=1)   # A^T V  [C,E]
        fcc = self.b_classifier.fcc
        logits = NY.Linear.apply(B.reshape(1, Cc * E), fcc.weight.view(Cc, Cc * E), fcc.bias, 0.0, 0, None)[0]
        attn = None
        if want_attn:
            with torch.no_grad():
                if self.cls_attn:
                    attn = rowmax(classes)                                                      # baseline.py:176 (raw logits either way)
                else:
                    attn = rowmax((a_raw * (1.0 / math.sqrt(QDIM))) if no_norm else A)           # [M]
        return logits, logits_ins, B, attn
