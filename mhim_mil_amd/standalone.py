"""The standalone attention-MIL models of the reference's model factory on the MHIM path's kernels.

``build_model`` mirrors the branches of modules/__init__.py:71-116 that share the hot path's kernels:

    'mhim' / 'mhim_pure'  -> mhim.MHIM                     (modules/mhim.py)
    'abmil'               -> DAttention                    (modules/abmil.py:145-251: biased scorer, tanh, classifier)
    'gabmil'              -> AttentionGated                (modules/abmil.py:51-143: D = 384, tanh x sigmoid gate)
    'transmil'            -> TransMIL                      (modules/transmil.py:66-175: wrap-padded tokens, cls token, two Nystrom
                                                            TransLayers around a PPEG, LayerNorm, classifier)

Same parameter names and shapes as the reference modules (a reference checkpoint loads with ``load_state_dict``), the
reference initialisation (xavier-normal weights, zero biases: abmil.py:8-21), and trainable through autograd: the
embedding and the scorer + softmax pool are ``torch.autograd.Function``s over libmhimx.so (GEMM with fused bias /
activation / counter-based dropout; mhimx_abmil_pool_fwd / _bwd with the bias gradients).  Options: ``mil_norm`` None / 'ln'
(LayerNorm before the embedding, ``embed_norm_pos=0``, or after it, ``=1``, plus ``norm1`` on the pooled vector: abmil.py:171-178,
transmil.py:83-84), ``pos`` None / 'none' / 'sincos' (abmil, emb_position.py:5-83) and 'ppeg' / 'none' (transmil); the gated
scorer's inner dropouts (abmil.py:96-98, active in training when ``dropout`` is set) run inside the scorer's row kernels
(``mhimx_scorer.gate_drop_p``); ``mil_norm='bn'`` is a BatchNorm over the instances of ONE bag (``mhimx_bn_fwd / _bwd``; running
statistics kept in the ``nn.BatchNorm1d`` holder; DAttention's ``norm1`` on the single pooled row raises in training mode exactly as
the reference does); ``embed_feat=False`` (abmil / transmil: the bag rows are the tokens, ``input_dim == inner_dim``).
CLAM, DTFD, RRT, ... are other model families (SURVEY.md §8 out of scope).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import _lib as L
from . import nystrom as NY
from . import ops
from .mhim import MHIM


class _EmbedFn(torch.autograd.Function):
    """H = dropout(act(x[rows] W^T + b)) in one GEMM launch (``rows``: optional row gather, fused into the A loads); backward:
    act/dropout backward + bias column sums in one pass, then dW = dPre^T x[rows] (the bag x is data: no input gradient)."""

    @staticmethod
    def forward(ctx, x, w, b, act, drop_p, seed, rows=None):
        need_pre = act == L.ACT["gelu"]
        M = x.shape[0] if rows is None else rows.shape[0]
        pre = torch.empty((M, w.shape[0]), device=x.device) if need_pre else None
        H = ops.gemm_nt(x, w, rows=rows, bias=b, act=act, pre=pre, drop_p=drop_p, drop_seed=seed, prec="bf16x3")
        ctx.save_for_backward(x, H, pre, rows)
        ctx.cfg = (act, drop_p, seed, b is not None)
        ctx.w = w
        return H

    @staticmethod
    def backward(ctx, dH):
        x, H, pre, rows = ctx.saved_tensors
        act, drop_p, seed, has_b = ctx.cfg
        g = dH.contiguous().clone()
        g, db = ops.act_bwd(g, H, pre, act, drop_p, seed, None, rows, want_colsum=True)
        dw = ops.gemm_tn(g, x, rows=rows, splits=8 if g.shape[0] >= 2048 else 1, prec="bf16x3")
        dx = None
        if ctx.needs_input_grad[0]:                              # a LayerNorm in front of the embedding (mil_norm='ln'): dx = dPre W
            if rows is not None:
                raise L.MhimxError("_EmbedFn: an input gradient through a row gather is not built")
            w = ctx.w
            dx = torch.empty_like(x)
            NY._gemm("nn", g, 0, g.shape[1], w, 0, w.shape[1], dx, 0, x.shape[1], g.shape[0], x.shape[1], g.shape[1])
        return dx, dw, (db if has_b else None), None, None, None, None


class _PoolFn(torch.autograd.Function):
    """z = softmax_n(scorer(T)) T  (mhimx_abmil_pool_fwd) with every scorer weight and bias trainable."""

    @staticmethod
    def forward(ctx, T, wa, ba, wc, bc, wb, bb, act, gate_p=0.0, gate_seed=0):
        T = T.contiguous()
        sc = ops.ScorerW(wa, wc, act, ba=ba, wb=wb, bb=bb, bc=bc, prec="bf16x3", gate_drop_p=gate_p, gate_drop_seed=gate_seed)
        st = ops.abmil_pool_fwd(sc, T)
        ctx.sc, ctx.st = sc, st
        ctx.gated = wb is not None
        ctx.mark_non_differentiable(st.s, st.stats)
        return st.z, st.s, st.stats

    @staticmethod
    def backward(ctx, g_z, _gs, _gst):
        sc, st = ctx.sc, ctx.st
        wa, wc, ba, wb, bb, bc = sc.t[:6]
        g = ops.abmil_pool_bwd(sc, st, g_z.contiguous(), ops.transpose(wa), ops.transpose(wb) if ctx.gated else None,
                               need_bias=ba is not None or bc is not None)
        return (g["dT1"], g["d_wa"], g.get("d_ba") if ba is not None else None, g["d_wc"], g.get("d_bc") if bc is not None else None,
                g.get("d_wb") if ctx.gated else None, g.get("d_bb") if (ctx.gated and bb is not None) else None, None, None, None)


def _linear(i, o, bias=True):
    m = nn.Linear(i, o, bias=bias)
    nn.init.xavier_normal_(m.weight)                      # abmil.py:8-14
    if m.bias is not None:
        m.bias.data.zero_()
    return m


class _Slot(nn.Module):                                   # a parameter-free layer of the reference's nn.Sequential (keeps the indices)
    pass


class _SinCosAdd(torch.autograd.Function):
    """x + SINCOS(pos) (emb_position.py:5-83): parameter-free, the gradient passes through."""

    @staticmethod
    def forward(ctx, x, pos_xy):
        x = x.contiguous()
        out = torch.empty_like(x)
        L.check(L.lib().mhimx_sincos_add(NY._st(), NY._ptr(x), ops._p(pos_xy), x.shape[0], x.shape[1], NY._ptr(out)), "mhimx_sincos_add")
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, None


class _BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm1d over the M instances of one bag ([1, M, C] transposed to [1, C, M] in the reference: abmil.py:206-210)."""

    @staticmethod
    def forward(ctx, x, w, b, bn, training):
        x = x.contiguous()
        M, Cc = x.shape
        dev = x.device
        y = torch.empty_like(x)
        ws = torch.empty(L.lib().mhimx_bn_ws_floats(M, Cc), device=dev)
        if training:
            mean, rstd, var = (torch.empty(Cc, device=dev) for _ in range(3))
        else:
            mean, var = bn.running_mean.detach().float().contiguous(), None
            rstd = torch.rsqrt(bn.running_var.detach().float() + bn.eps).contiguous()
        L.check(L.lib().mhimx_bn_fwd(NY._st(), NY._ptr(x), M, Cc, NY._ptr(w), NY._ptr(b), float(bn.eps), int(training), NY._ptr(mean),
                                     NY._ptr(rstd), None if var is None else NY._ptr(var), NY._ptr(y), NY._ptr(ws)), "mhimx_bn_fwd")
        if training:                                             # running statistics (momentum 0.1, unbiased variance), on [C] vectors
            with torch.no_grad():
                mom = bn.momentum if bn.momentum is not None else 0.1
                bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var * (M / max(M - 1, 1)), alpha=mom)
                bn.num_batches_tracked += 1
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        M, Cc = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw, db = torch.empty(Cc, device=x.device), torch.empty(Cc, device=x.device)
        ws = torch.empty(L.lib().mhimx_bn_ws_floats(M, Cc), device=x.device)
        L.check(L.lib().mhimx_bn_bwd(NY._st(), NY._ptr(dy), NY._ptr(x), M, Cc, NY._ptr(w), NY._ptr(mean), NY._ptr(rstd), int(ctx.training),
                                     None if dx is None else NY._ptr(dx), NY._ptr(dw), NY._ptr(db), NY._ptr(ws)), "mhimx_bn_bwd")
        return dx, dw, db, None, None


def _bn(x, m, training):
    if training and x.shape[0] < 2:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")    # as nn.BatchNorm1d
    return _BatchNormFn.apply(x, m.weight, m.bias, m, bool(training))


def _layernorm(dim, bias=True):
    """nn.LayerNorm(dim, bias=mil_bias) under the reference's initialisation (weight 1, bias 0: abmil.py:15-17)."""
    m = nn.Module()
    m.weight = nn.Parameter(torch.ones(dim))
    m.bias = nn.Parameter(torch.zeros(dim)) if bias else None
    return m


def _ln(x, m):
    b = m.bias if m.bias is not None else torch.zeros_like(m.weight)
    return NY.LayerNorm.apply(x, m.weight, b)


class _AttnMILBase(nn.Module):
    def _check(self, x):
        if not x.is_cuda:
            raise L.MhimxError("standalone MIL (mhimx): the bag must be a CUDA tensor; there is no CPU path")
        if x.dim() == 3:
            if x.shape[0] != 1:
                raise L.MhimxError("one bag per call (batch_size = 1, as the reference trainer does)")
            x = x[0]
        return x.contiguous().float()

    def _seed(self):
        self._step = getattr(self, "_step", 0) + 1
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._step * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def _norm_cfg(self, mil_norm, who):
        if mil_norm not in (None, "ln", "bn"):
            raise L.MhimxError(f"{who} (mhimx): mil_norm={mil_norm!r} (None, 'ln', 'bn')")
        return mil_norm

    def _embed(self, x, rows=None):
        """self.feature: [LayerNorm(input_dim)]? + Linear + act? + Dropout?  (the reference's nn.Sequential indices)"""
        i = 1 if getattr(self, "_ln_first", False) else 0
        if getattr(self, "_no_embed", False):                      # embed_feat=False (abmil.py:180, transmil.py:87): the bag rows ARE the tokens
            x = _ln(x, self.feature[0]) if i else x
            return x if rows is None else x.index_select(0, rows)
        if i:
            if rows is not None:
                raise L.MhimxError("_embed: LayerNorm + row gather is handled by the caller")
            x = _ln(x, self.feature[0])
        f = self.feature[i]
        p = self.embed_drop if self.training else 0.0
        return _EmbedFn.apply(x, f.weight, f.bias, L.ACT[self.act], float(p), self._seed(), rows)


class DAttention(_AttnMILBase):
    """modules/abmil.py:145-251."""

    def __init__(self, input_dim, n_classes, dropout, act, mil_norm=None, mil_bias=True, mil_cls_bias=True, inner_dim=512,
                 embed_feat=True, embed_norm_pos=0, pos=None, **kwargs):
        super().__init__()
        if pos not in (None, "none", "sincos") or embed_norm_pos not in (0, 1):
            raise L.MhimxError("DAttention: pos in ('sincos', 'none', None), embed_norm_pos in (0, 1) (abmil.py:159-160)")
        self._no_embed = not embed_feat
        if self._no_embed and input_dim != inner_dim:
            raise L.MhimxError("DAttention(embed_feat=False): the bag rows are the tokens, input_dim must equal inner_dim")
        self.mil_norm, self.embed_norm_pos, self.pos = self._norm_cfg(mil_norm, "DAttention"), embed_norm_pos, pos
        if mil_bias:
            mil_cls_bias = True
        self.L, self.D, self.K = inner_dim, 128, 1
        self.act = "gelu" if act.lower() == "gelu" else "relu"
        self.embed_drop = 0.25 if dropout else 0.0                          # abmil.py:190-191: a FIXED 0.25 when dropout is set
        self._ln_first = mil_norm == "ln" and embed_norm_pos == 0
        layers = ([_layernorm(input_dim, mil_bias)] if self._ln_first else []) + \
            ([] if self._no_embed else [_linear(input_dim, inner_dim, mil_bias), _Slot()] + ([_Slot()] if dropout else []))
        if mil_norm == "ln":
            if embed_norm_pos == 1:
                self.norm = _layernorm(inner_dim, mil_bias)
            self.norm1 = _layernorm(self.L * self.K, mil_bias)
        elif mil_norm == "bn":                                              # abmil.py:167-169
            self.norm = nn.BatchNorm1d(input_dim if embed_norm_pos == 0 else inner_dim)
            self.norm1 = nn.BatchNorm1d(self.L * self.K)
        self.feature = nn.Sequential(*layers)
        self.attention = nn.Sequential(_linear(self.L, self.D, mil_bias), _Slot(), _linear(self.D, self.K, mil_bias))
        self.classifier = _linear(self.L * self.K, n_classes, mil_cls_bias)

    def forward(self, x, return_attn=False, no_norm=False, return_act=False, pos=None, return_img_feat=False, **kwargs):
        x = self._check(x)
        if self.mil_norm == "bn" and self.embed_norm_pos == 0:               # abmil.py:206-210
            x = _bn(x, self.norm, self.norm.training)
        H = self._embed(x)
        if self.pos == "sincos":                                             # abmil.py:216-217, emb_position.py:62-83
            if pos is None:
                raise L.MhimxError("DAttention(pos='sincos'): forward needs the patch coordinates `pos`")
            pp = pos[0] if pos.dim() == 3 else pos                            # [1 + N, 2]: (W, H) of the grid, then (x, y) per patch
            H = _SinCosAdd.apply(H, pp[1:].to(device=H.device, dtype=torch.int64).contiguous())
        if self.mil_norm == "ln" and self.embed_norm_pos == 1:
            H = _ln(H, self.norm)
        elif self.mil_norm == "bn" and self.embed_norm_pos == 1:             # abmil.py:219-223
            H = _bn(H, self.norm, self.norm.training)
        a0, a2 = self.attention[0], self.attention[2]
        z, s, stats = _PoolFn.apply(H, a0.weight, a0.bias, a2.weight, a2.bias, None, None, L.ACT["tanh"])
        zz = z.view(1, -1)
        # abmil.py:237 (BatchNorm1d on the ONE pooled row raises in training mode, there as here: the reference cannot train this setting)
        zc = _ln(zz, self.norm1) if self.mil_norm == "ln" else (_bn(zz, self.norm1, self.norm1.training) if self.mil_norm == "bn" else zz)
        logits = NY.Linear.apply(zc, self.classifier.weight, self.classifier.bias, 0.0, 0, None)
        out = [logits, zz.clone()] if return_img_feat else logits
        if not return_attn:
            return out
        res = [out, ops.softmax_from_stats(s, stats).view(1, -1)]            # abmil.py:236-241 (always the normalised attention)
        if return_act:
            res.append(H)
        return res


class AttentionGated(_AttnMILBase):
    """modules/abmil.py:51-143."""

    def __init__(self, input_dim, n_classes, act="relu", dropout=0., mil_norm=None, mil_bias=True, mil_cls_bias=True, inner_dim=512,
                 embed_feat=True, embed_norm_pos=0, pos=None, **kwargs):
        super().__init__()
        self.mil_norm, self.embed_norm_pos = self._norm_cfg(mil_norm, "AttentionGated"), embed_norm_pos
        if mil_norm == "ln" and embed_norm_pos == 0:
            # abmil.py:66 appends to self.feature before it exists: the reference constructor raises AttributeError for this setting
            raise L.MhimxError("AttentionGated(mil_norm='ln', embed_norm_pos=0): the reference constructor fails here (abmil.py:66); "
                               "use embed_norm_pos=1")
        self.L, self.D, self.K = inner_dim, 384, 1
        self.act = act if act in ("gelu", "relu") else "none"
        self.embed_drop = float(dropout)                                    # abmil.py:79: nn.Dropout(dropout)
        self.scorer_drop = 0.25 if dropout else 0.0                         # abmil.py:96-98
        if mil_norm == "ln":
            self.norm = _layernorm(inner_dim, mil_bias)
            self.norm1 = _layernorm(self.L * self.K, mil_bias)              # (constructed, never applied: abmil.py:111-143)
        elif mil_norm == "bn":                                              # abmil.py:61-63
            self.norm = nn.BatchNorm1d(input_dim if embed_norm_pos == 0 else inner_dim)
            self.norm1 = nn.BatchNorm1d(self.L * self.K)
        self.feature = nn.Sequential(*([_linear(input_dim, inner_dim, mil_bias)] + ([_Slot()] if act in ("gelu", "relu") else []) + [_Slot()]))
        self.attention_a = nn.Sequential(_linear(self.L, self.D, mil_bias), _Slot())
        self.attention_b = nn.Sequential(_linear(self.L, self.D, mil_bias), _Slot())
        self.attention_c = _linear(self.D, self.K, mil_bias)
        self.classifier = nn.Sequential(_linear(self.L * self.K, n_classes, mil_bias))

    def forward(self, x, **kwargs):
        x = self._check(x)
        if self.mil_norm == "bn" and self.embed_norm_pos == 0:               # abmil.py:115-119
            x = _bn(x, self.norm, self.norm.training)
        H = self._embed(x)
        if self.mil_norm == "ln":
            H = _ln(H, self.norm)
        elif self.mil_norm == "bn" and self.embed_norm_pos == 1:             # abmil.py:123-127
            H = _bn(H, self.norm, self.norm.training)
        a, b, c = self.attention_a[0], self.attention_b[0], self.attention_c
        gp = self.scorer_drop if self.training else 0.0                     # abmil.py:96-98: Dropout(0.25) after the tanh and after the gate
        z, _, _ = _PoolFn.apply(H, a.weight, a.bias, c.weight, c.bias, b.weight, b.bias, L.ACT["tanh"], float(gp), self._seed())
        cl = self.classifier[0]
        return NY.Linear.apply(z.view(1, -1), cl.weight, cl.bias, 0.0, 0, None)


class TransMIL(_AttnMILBase):
    """modules/transmil.py:66-175: embedded tokens wrap-padded to a square (transmil.py:124-128; one concatenation together with the
    cls token) - then the encoder of SURVEY rows A9/A10 (mhim_mil_amd/nystrom.py) and a classifier.
    Any bag size: this model's PPEG is told the grid (transmil.py:57-64), it does not zero-pad to 7 x 7 as emb_position.PPEG does.
    pos='none' drops the PPEG (transmil.py:73-74,145-146); mil_norm='ln' puts a LayerNorm in front of the embedding (:83-84)."""

    def __init__(self, input_dim, n_classes, dropout, act, mil_norm=None, mil_bias=True, inner_dim=512, embed_feat=True, pos="ppeg",
                 n_heads=8, **kwargs):
        super().__init__()
        if inner_dim != 512 or n_heads != 8:
            raise L.MhimxError("TransMIL (mhimx): built for inner_dim=512, 8 heads")
        self._no_embed = not embed_feat
        if self._no_embed and input_dim != inner_dim:
            raise L.MhimxError("TransMIL(embed_feat=False): the bag rows are the tokens, input_dim must equal inner_dim")
        self.mil_norm, self.pos = self._norm_cfg(mil_norm, "TransMIL"), pos
        self.act = "gelu" if act.lower() == "gelu" else ("relu" if act.lower() == "relu" else "none")
        self.embed_drop = 0.25 if dropout else 0.0
        self._ln_first = mil_norm == "ln"
        if mil_norm == "bn":
            self.norm1 = nn.BatchNorm1d(input_dim)                          # transmil.py:79-81
        self.feature = nn.Sequential(*(([_layernorm(input_dim, mil_bias)] if self._ln_first else []) + ([] if self._no_embed else (
            [_linear(input_dim, inner_dim, mil_bias)] + ([_Slot()] if self.act != "none" else []) + ([_Slot()] if dropout else [])))))
        self.cls_token = nn.Parameter(torch.randn(1, 1, inner_dim) * 1e-6)      # transmil.py:99-100
        self.layer1, self.layer2 = NY.TransLayer(inner_dim), NY.TransLayer(inner_dim)
        self.pos_layer = NY._PPEG(inner_dim) if pos != "none" else nn.Identity()
        self.norm = NY._P(weight=torch.ones(inner_dim), bias=torch.zeros(inner_dim))
        self.classifier = _linear(inner_dim, n_classes, mil_bias)
        self.n_classes = n_classes

    def forward(self, x, return_attn=False, return_act=False, **kwargs):
        x = self._check(x)
        if self.mil_norm == "bn":                       # transmil.py:112-115
            x = _bn(x, self.norm1, self.norm1.training)
        n = x.shape[0]
        side = int(math.ceil(math.sqrt(n)))             # transmil.py:124-126 (any side: its PPEG takes the grid explicitly, :57-64)
        while side * side < n:
            side += 1
        add = side * side - n
        # the reference embeds (Linear, act, Dropout) the n bag rows and THEN wraps: cat([h, h[:add]]) (transmil.py:117-127) - the wrapped
        # rows are exact copies of rows 0..add-1 including their dropout mask, so the wrap is a concatenation of embedded rows here too
        # (rows 0..add-1 receive both gradients through autograd's cat), and the embedding GEMM runs on n rows, not on side^2
        h = self._embed(x, None)
        tr = self.training
        s1, s2 = self._seed(), self._seed()
        t = torch.cat([self.cls_token.view(1, -1), h] + ([h[:add]] if add > 0 else []), 0)
        attn = []
        if return_attn:
            t, a, v = self.layer1(t, True, False, s1, None, tr)
            attn.append((a[:, :a.shape[1] - add] if add > 0 else a).unsqueeze(0))     # transmil.py:138-141
        else:
            t = self.layer1(t, False, False, s1, None, tr)
        if self.pos != "none":
            t = self.pos_layer(t, grid=side, skip=1)                     # transmil.py:62-63: cat([cls, ppeg(tokens)])
        if return_attn:
            t, a, _ = self.layer2(t, True, False, s2, None, tr)
            attn.append((a[:, :a.shape[1] - add] if add > 0 else a).unsqueeze(0))
        else:
            t = self.layer2(t, False, False, s2, None, tr)
        z = NY.LayerNorm.apply(t[:1].contiguous(), self.norm.weight, self.norm.bias)
        logits = NY.Linear.apply(z, self.classifier.weight, self.classifier.bias, 0.0, 0, None)
        if not return_attn:
            return logits
        return [logits, attn, v] if return_act else [logits, attn]


def build_teacher(model, others=None, teacher_init=None, no_tea_init=False, tea_type="ema", mm_sche=None):
    """The MHIM teacher seam of the reference factory (modules/__init__.py:176-214): ``model_tea = deepcopy(model)`` (a device-resident
    model copies on the device), the optional ``--teacher_init`` checkpoint loaded with ``strict=False`` after the ``module.`` prefix of a
    DistributedDataParallel checkpoint is added / stripped to match (an ``mhim_pure`` checkpoint has no ``merge.*`` keys: they keep the
    student's values), ``tea_type == 'same'`` -> the student itself, ``merge_test = False``.  Fills ``others['model_ema']`` and
    ``others['mm_sche']`` as the factory does and returns the teacher."""
    import copy
    others = {} if others is None else others
    others["mm_sche"] = mm_sche
    tea = copy.deepcopy(model)
    if teacher_init is not None and not no_tea_init and tea_type != "same":
        pre = torch.load(teacher_init, weights_only=True) if isinstance(teacher_init, (str, bytes)) or hasattr(teacher_init, "read") else teacher_init
        if "model" in pre:
            pre = pre["model"]
        model_has = any(k.startswith("module.") for k in tea.state_dict())
        pre_has = any(k.startswith("module.") for k in pre)
        if model_has and not pre_has:
            pre = {"module." + k: v for k, v in pre.items()}
        elif pre_has and not model_has:
            pre = {k.replace("module.", ""): v for k, v in pre.items()}
        others["teacher_init_info"] = tea.load_state_dict(pre, strict=False)
    if tea_type == "same":
        tea = model
    tea.merge_test = False
    others["model_ema"] = tea
    return tea


def build_model(model_name, **params):
    """The kernel-sharing branches of modules/__init__.py:71-116; ``params`` are the constructor arguments the reference's
    factory assembles (``genera_model_params`` / the MHIM ``model_params``)."""
    name = model_name.lower()
    if name == "mhim":
        return MHIM(**params)
    if name == "mhim_pure":
        params = dict(params)
        params.update(select_mask=False, merge_enable=False)
        return MHIM(**params)
    if name == "abmil":
        return DAttention(**params)
    if name == "gabmil":
        return AttentionGated(**params)
    if name == "transmil":
        return TransMIL(**params)
    raise NotImplementedError(f"model {model_name!r}: not a branch of the MHIM / attention-MIL hot path (SURVEY.md §8)")
