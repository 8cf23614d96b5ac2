"""MHIM — host-side mirror of the reference's ``modules/mhim.MHIM`` over the HIP kernels.

Same constructor arguments, same methods and return conventions, same state_dict keys
(SURVEY.md §8(b)), so a reference training script can import this class instead:

    reference                               this class
    ---------                               ----------
    MHIM.forward_teacher  (mhim.py:181-227) forward_teacher  -> (feat [1,E], score [1,N])
    MHIM.forward          (mhim.py:318-378) forward          -> (logit [1,C], cls_loss, ps, len_keep)
    MHIM.forward_test     (mhim.py:229-272) forward_test     -> logits | (logits, attn)
    MHIM.pure             (mhim.py:274-298) pure             -> (logits, 0, ps, ps) | logits

All math runs in libmhimx.so (hand-written gfx950 kernels behind the C-ABI); torch only owns
parameters, device memory, the stream and autograd's graph bookkeeping.  There is no eager/CPU
fallback: inputs must be CUDA tensors and the library must be built.

``baseline='attn'`` (ABMIL) runs the fused hand-derived forward/backward; ``baseline='selfattn'`` (TransMIL /
Nystrom) composes the encoder from kernel-backed autograd primitives (nystrom.py);
``baseline='dsmil'`` (scope row N1) composes the DSMIL encoder the same way (dsmil.py).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np

import torch
from torch import nn

from . import _lib as L
from . import dsmil as DS
from . import nystrom as NY
from . import ops

_FEATURE_ACTS = ("relu", "gelu")
_SCORER_ACTS = ("relu", "gelu", "tanh")
_SPLIT_POOL = os.environ.get("MHIMX_SPLIT_POOL", "1") != "0"      # the student's pool forward in two calls around the Merge tail (ops.abmil_pool_fwd_split)


# ----------------------------------------------------------------------------------------------- holders
class _Lin(nn.Module):
    """Parameter holder with nn.Linear's names/shapes/init law (no math: kernels read .weight/.bias)."""

    def __init__(self, i, o, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.zeros(o)) if bias else None
        nn.init.xavier_normal_(self.weight)          # mhim_modules/utils.py:16-19


class _Norm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _Slot(nn.Module):
    """Occupies a Sequential index that holds a parameter-free module in the reference (activation/dropout)."""


class _Attention(nn.Module):
    def __init__(self, E, gated):
        super().__init__()
        if gated:
            self.attention_a = nn.Sequential(_Lin(E, 128, bias=False), _Slot())
            self.attention_b = nn.Sequential(_Lin(E, 128, bias=False), _Slot())
            self.attention_c = _Lin(128, 1, bias=False)
        else:
            self.attention = nn.Sequential(_Lin(E, 128, bias=False), _Slot(), _Lin(128, 1, bias=False))


class _DAttention(nn.Module):
    def __init__(self, E, gated=False):
        super().__init__()
        self.gated = gated
        self.attention = _Attention(E, gated)


class _MCA(nn.Module):
    def __init__(self, E, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.to_kv = _Lin(E, 2 * inner, bias=False)
        self.to_q = _Lin(E, inner, bias=False)
        self.to_out = nn.Sequential(_Lin(inner, E), _Slot())


class _Merge(nn.Module):
    def __init__(self, E, k, mm, merge_ratio):
        super().__init__()
        self.norm = _Norm(E)
        self.attn = _MCA(E)
        val = math.sqrt(6.0 / float(3 * 16 * 16 + E))                      # merge.py:109-111
        self.global_q_mm = nn.Parameter(torch.empty(1, k, E).uniform_(-val, val), requires_grad=False)
        self.global_q = self.global_q_mm                                   # same Parameter, second name (merge.py:118)
        self.k, self.g_q_mm, self.merge_ratio = k, mm, merge_ratio
        self.dropout = 0.1                                                 # Merge(dropout=0.1) default, merge.py:72


# ----------------------------------------------------------------------------------------------- autograd
class _BagFn(torch.autograd.Function):
    """Student bag forward: X rows -> feature -> (merge) -> scorer + pool -> bag feature z [E].

    Saves only device buffers produced by the kernels; backward is hand-derived and
    produces every parameter gradient in one pass of kernel launches.  X gets no gradient.
    """

    @staticmethod
    def forward(ctx, model, x, plan, *params):
        ctx.model, ctx.plan = model, plan
        ctx.names = list(model._bag_param_names)
        ctx.merge_on = model.merge_enable
        z, saved = model._bag_forward(x, plan)
        ctx.saved = saved
        ctx.x = x
        return z.view(1, -1)

    @staticmethod
    def backward(ctx, g_z):
        model = ctx.model
        keep, model.merge_enable = model.merge_enable, ctx.merge_on
        try:
            grads = model._bag_backward(ctx.x, ctx.plan, ctx.saved, g_z.contiguous().view(-1))
        finally:
            model.merge_enable = keep
        out = [grads.get(name) for name in ctx.names]
        return (None, None, None, *out)


class _HeadFn(torch.autograd.Function):
    """(z, teacher feat) -> (logits [1,C], cls_loss scalar): predictor + SoftTargetCrossEntropy in one kernel."""

    @staticmethod
    def forward(ctx, z, t, wp, bp, temp_t):
        logits, losses, _, _, _ = ops.head_fwd_bwd(z.view(-1), None if t is None else t.contiguous().view(-1), wp, bp, None,
                                                   temp_t=temp_t)
        ctx.save_for_backward(z, t, wp, bp)
        ctx.temp_t = temp_t
        return logits.view(1, -1), losses[2].clone()

    @staticmethod
    def backward(ctx, g_logits, g_cl):
        z, t, wp, bp = ctx.saved_tensors
        g_logits = g_logits.contiguous().view(-1)
        g_cl = g_cl.contiguous().view(1) if g_cl is not None else torch.zeros(1, device=z.device)
        _, _, g_z, d_wp, d_bp = ops.head_fwd_bwd(z.view(-1), None if t is None else t.contiguous().view(-1), wp, bp, None,
                                                 temp_t=ctx.temp_t, g_logits_in=g_logits, g_cl_in=g_cl)
        return g_z.view_as(z), None, d_wp, d_bp, None


_TOKENS_NODE = os.environ.get("MHIMX_TOKENS_NODE", "1") != "0"


class _FeatureFn(torch.autograd.Function):
    """Token rows of the student: H = dropout(act(X[rows] W^T + b))  (mhim.py:68-76,337; masking.py:107 gathers the rows)."""

    @staticmethod
    def forward(ctx, model, x, plan, w, b):
        pre = getattr(plan, "pre", None)
        if pre is not None:
            # the single-pass projection (ops.bag_project) already holds the student's feature rows of ALL bag rows (bag order) and
            # d out / d pre in fp16: the token rows are a gather, the backward the matrix-core-image weight-gradient pair
            Hs, dact = pre
            H = ops.shard_gather(Hs, plan.rows, 0, Hs.shape[0]) if plan.rows is not None else Hs
            ctx.model, ctx.plan, ctx.x, ctx.dact = model, plan, x, dact
            return H
        need_pre = L.act_code(model.act, _FEATURE_ACTS) == L.ACT["gelu"]
        H = torch.empty((plan.L, model.mlp_dim), device=x.device)
        PRE = torch.empty_like(H) if need_pre else None
        p = model.dropout_p if plan.training else 0.0
        model._feature(x, plan.rows, p, plan.drop_seed, plan.drop_mask, out=H, pre_out=PRE, M=plan.L)
        ctx.model, ctx.plan, ctx.x, ctx.H, ctx.PRE = model, plan, x, H, PRE
        ctx.dact = None
        return H

    @staticmethod
    def backward(ctx, dH):
        model, plan = ctx.model, ctx.plan
        if ctx.dact is not None:
            dH = dH.contiguous()
            if plan.rows is None:
                dW, db = ops.bag_wgrad(dH, ctx.dact, ctx.x, None, plan.L)
            else:
                dW, db = ops.bag_wgrad(dH, ctx.dact, ctx.x, plan.rows, plan.L, dh_compact=True)
            return None, None, None, dW, db
        dH = dH.contiguous().clone()
        p = model.dropout_p if plan.training else 0.0
        _, db = ops.act_bwd(dH, ctx.H, ctx.PRE, L.act_code(model.act, _FEATURE_ACTS), p, plan.drop_seed, plan.drop_mask, plan.rows,
                            want_colsum=True, drop_tick=model._tick)
        if plan.L >= 2048 and model.prec != "f32" and ops.bag_wgrad_ok(ctx.x, dH.shape[1], plan.L):
            # the matrix-core-image weight-gradient pair (csrc/wgrad.hip): dH is compact, the bag rows are gathered
            dW, _ = ops.bag_wgrad(dH, None, ctx.x, plan.rows, plan.L, rows_dh=None, want_bias=False)
        else:
            dW = ops.gemm_tn(dH, ctx.x, rows=plan.rows, splits=8 if plan.L >= 2048 else 1,
                             prec="f32" if model.prec == "f32" else "bf16x3", M=plan.L)
        return None, None, None, dW, db


class _MergeFn(torch.autograd.Function):
    """Merge.merge (merge.py:131-144): rows to merge -> k merged tokens, EMA of the global queries in the forward."""

    NAMES = ("merge.norm.weight", "merge.norm.bias", "merge.attn.to_kv.weight", "merge.attn.to_q.weight",
             "merge.attn.to_out.0.weight", "merge.attn.to_out.0.bias")

    @staticmethod
    def forward(ctx, model, plan, X, *params):
        X = X.contiguous()
        z_tok, q_new, mws = ops.merge_fwd(model._merge_w(plan), X, update_q=plan.training)
        ctx.model, ctx.plan, ctx.X, ctx.mws = model, plan, X, mws
        ctx.q_old = model.merge.global_q_mm.data.clone()
        if plan.training:
            model.merge.global_q_mm.data.copy_(q_new.view_as(model.merge.global_q_mm))
        return z_tok

    @staticmethod
    def backward(ctx, dz):
        model = ctx.model
        mw = model._merge_w(ctx.plan, need_t=True, q=ctx.q_old)
        mg = ops.merge_bwd(mw, ctx.X, dz.contiguous(), ctx.mws, grads={})
        return (None, None, mg["dX"], mg["d_ln_w"], mg["d_ln_b"], mg["d_wkv"], mg["d_wq"], mg["d_wo"], mg["d_bo"])


class _StudentTokensFn(torch.autograd.Function):
    """The TransMIL student's token matrix [cls ; kept tokens ; k merged tokens] (mhim.py:335-352, merge.py:131-144,190-194, baseline.py:
    248-251) from the single-pass projection's rows (plan.pre) as ONE autograd node.  As the chain _FeatureFn -> slices -> _MergeFn -> cat ->
    cat the forward copied the kept rows three times (gather, two concatenations) and the backward assembled d tokens from two zero-filled
    slice gradients and an add (~0.25 ms per c3 step): here the gather writes the kept rows where the encoder reads them, the merged tokens
    and the cls token are written beside them, and the backward hands the weight-gradient pair ONE compact [L, E] matrix (the kept rows'
    gradient copied once, Merge's dX written into its tail)."""

    @staticmethod
    def forward(ctx, model, x, plan, cls, w, b, *mparams):
        Hs, dact = plan.pre
        Lk, R, E, N = plan.Lk, plan.R, Hs.shape[1], Hs.shape[0]
        k = model.merge.k if R > 0 else 0
        tok = torch.empty((1 + Lk + k, E), device=Hs.device)
        ops.shard_gather(Hs, plan.rows[:Lk], 0, N, out=tok[1:1 + Lk])
        jobs = [(ops.PREP_COPY, cls.reshape(1, E), tok[:1])]
        ctx.merge = None
        if R > 0:
            X = ops.shard_gather(Hs, plan.rows[Lk:Lk + R], 0, N)
            z_tok, q_new, mws = ops.merge_fwd(model._merge_w(plan), X, update_q=plan.training)
            ctx.merge = (X, mws, model.merge.global_q_mm.data.clone())
            if plan.training:
                model.merge.global_q_mm.data.copy_(q_new.view_as(model.merge.global_q_mm))
            jobs.append((ops.PREP_COPY, z_tok, tok[1 + Lk:]))
        ops.prep_batch(jobs)
        ctx.model, ctx.plan, ctx.x, ctx.dact, ctx.cls_shape = model, plan, x, dact, cls.shape
        return tok

    @staticmethod
    def backward(ctx, dtok):
        model, plan = ctx.model, ctx.plan
        Lk, R = plan.Lk, plan.R
        dtok = dtok.contiguous()
        E = dtok.shape[1]
        dH = torch.empty((Lk + R, E), device=dtok.device)
        ops.stream_copy(dtok[1:1 + Lk], dH[:Lk])
        mg = [None] * 6
        if ctx.merge is not None:
            X, mws, q_old = ctx.merge
            mw = model._merge_w(plan, need_t=True, q=q_old)
            g = ops.merge_bwd(mw, X, dtok[1 + Lk:], mws, grads={"dX": dH[Lk:]})
            mg = [g["d_ln_w"], g["d_ln_b"], g["d_wkv"], g["d_wq"], g["d_wo"], g["d_bo"]]
        dW, db = ops.bag_wgrad(dH, ctx.dact, ctx.x, plan.rows, plan.L, dh_compact=True)
        return (None, None, None, dtok[:1].reshape(ctx.cls_shape), dW, db, *mg)


class BagPlan:
    """Row bookkeeping of one student forward (all device tensors; no host sync)."""

    def __init__(self, rows=None, L=0, Lk=0, R=0, drop_seed=0, drop_mask=None, mca_seed=0, training=True, merge_first=False):
        self.rows, self.L, self.Lk, self.R = rows, L, Lk, R
        self.merge_first = merge_first            # rows = [rows to merge (R) | rows that stay (Lk)] instead of [stay | merge]
        self.drop_seed, self.drop_mask, self.mca_seed, self.training = drop_seed, drop_mask, mca_seed, training


# ----------------------------------------------------------------------------------------------- model
class MHIM(nn.Module):
    def __init__(self, input_dim=1024, mlp_dim=512, mask_ratio=0, n_classes=2, temp_t=1., dropout=0.25, act='relu',
                 mask_ratio_h=0., mrh_sche=None, mask_ratio_hr=0., mask_ratio_l=0., da_act='gelu', baseline='selfattn',
                 head=8, attn2score=True, merge_enable=True, merge_k=1, merge_mm=0.9998, merge_ratio=0.,
                 merge_test=False, attn_layer=None, select_mask=None, prec="auto", gated=False, **_ignored):
        # attn_layer / select_mask: accepted and ignored — the reference's own factory passes them although its
        # constructor rejects them (modules/__init__.py:91,102).
        super().__init__()
        self.mask_ratio, self.mask_ratio_h, self.mask_ratio_hr, self.mask_ratio_l = mask_ratio, mask_ratio_h, mask_ratio_hr, mask_ratio_l
        self.select_inv = False
        self.msa_fusion = "vote"
        self.mrh_sche = mrh_sche
        self.attn_layer = 0
        self.baseline = baseline
        self.merge_test = merge_test
        self.attn2score = attn2score
        self.head = head
        self.temp_t, self.temp_s = temp_t, 1.0
        self.input_dim, self.mlp_dim, self.n_classes = input_dim, mlp_dim, n_classes
        self.act, self.da_act = act, da_act
        self.dropout_p = float(dropout)
        self.prec = prec
        self.merge_enable = bool(merge_enable)
        if baseline not in ("attn", "selfattn", "dsmil"):
            raise NotImplementedError(f"baseline={baseline!r}: the reference knows 'attn' (ABMIL), 'selfattn' (TransMIL/Nystrom) "
                                      "and 'dsmil'")
        self.merge = _Merge(mlp_dim, merge_k, merge_mm, merge_ratio) if merge_enable else nn.Identity()
        self.feature = nn.Sequential(_Lin(input_dim, mlp_dim), _Slot())
        if baseline == "attn":
            self.online_encoder = _DAttention(mlp_dim, gated=gated)
        elif baseline == "selfattn":
            self.online_encoder = NY.SAttention(mlp_dim, head)
        else:
            self.online_encoder = DS.DSMIL(n_classes=n_classes, mlp_dim=mlp_dim, cls_attn=attn2score)      # mhim.py:91-95
        self.predictor = _Lin(mlp_dim, n_classes)
        self._step = 0
        self._tick = None          # optional device step counter (uint64 [1]) mixed into every dropout seed: set by
                                   # FusedTrainer so that a captured hipGraph draws fresh masks on each replay

    # ------------------------------------------------------------------ weights in kernel layout
    def parameters(self, recurse: bool = True):
        """nn.Module.parameters - except for a teacher whose EMA a fused optimiser owns (optim.FusedAdamEMA sets ``_ema_owned``): the
        reference trainer's per-parameter EMA loop (base_engine.py:166-167, ``zip(model.parameters(), model_ema.parameters())``) then finds
        nothing left to update.  named_parameters / state_dict / load_state_dict are untouched."""
        if getattr(self, "_ema_owned", False):
            return iter(())
        return super().parameters(recurse)

    def __deepcopy__(self, memo):
        """A copy (a best-teacher snapshot, a re-built teacher) is an ordinary module: the ``_ema_owned`` mark belongs to the instance a
        fused optimiser adopted, not to its copies (they would look parameterless to requires_grad_ / zero_grad / a second optimiser)."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for key, val in self.__dict__.items():
            if key != "_ema_owned":
                new.__dict__[key] = copy.deepcopy(val, memo)
        return new

    @property
    def _bag_param_names(self):
        names = ["feature.0.weight", "feature.0.bias"]
        if self.online_encoder.gated:
            names += ["online_encoder.attention.attention_a.0.weight", "online_encoder.attention.attention_b.0.weight",
                      "online_encoder.attention.attention_c.weight"]
        else:
            names += ["online_encoder.attention.attention.0.weight", "online_encoder.attention.attention.2.weight"]
        if self.merge_enable:
            names += ["merge.norm.weight", "merge.norm.bias", "merge.attn.to_kv.weight", "merge.attn.to_q.weight",
                      "merge.attn.to_out.0.weight", "merge.attn.to_out.0.bias"]
        return names

    def unused_parameter_names(self):
        """Parameters no entry point reads (they never receive a gradient, so the reference's Adam never moves them)."""
        return ("predictor.weight", "predictor.bias") if self.baseline == "dsmil" else ()

    def _param(self, name):
        obj = self
        for part in name.split("."):
            obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
        return obj

    def _scorer(self, wa_frag=None):
        att = self.online_encoder.attention
        act = L.act_code(self.da_act, _SCORER_ACTS)
        if self.online_encoder.gated:
            return ops.ScorerW(att.attention_a[0].weight.data, att.attention_c.weight.data, act,
                               wb=att.attention_b[0].weight.data, prec=self._op_prec)
        return ops.ScorerW(att.attention[0].weight.data, att.attention[2].weight.data, act, prec=self._op_prec, wa_frag=wa_frag)

    def prep_jobs(self, backward=True, lean_merge=False):
        """Parameter-only work of one step — the paired-plane image of the projection weight, weight transposes for the dX
        GEMMs, a snapshot of the global queries — as (jobs for ops.prep_batch, dict of their outputs).  It depends on nothing
        but the parameters, so a trainer folds the teacher's and the student's jobs and its step counters into ONE launch."""
        jobs, prep = [], {"w1p": None}
        w1 = self.feature[0].weight.data
        if self._feature_prec(1 << 20) == "bf16x3" and self.input_dim % 32 == 0:
            prep["w1p"] = torch.empty_like(w1)
            jobs.append((ops.PREP_PAIR, w1, prep["w1p"]))

        def tr(w):
            out = torch.empty((w.shape[1], w.shape[0]), device=w.device)
            jobs.append((ops.PREP_TRANSPOSE, w, out))
            return out

        if self.baseline == "attn" and not self.online_encoder.gated:
            wa = self.online_encoder.attention.attention[0].weight.data
            if wa.shape[0] % 32 == 0 and wa.shape[1] % 16 == 0:           # matrix-core image of the scorer weight (fused scorer)
                prep["wa_frag"] = torch.empty_like(wa)
                jobs.append((ops.PREP_FRAG, wa, prep["wa_frag"]))
        if backward and self.baseline == "attn":
            att = self.online_encoder.attention
            if self.online_encoder.gated:
                prep["wa_t"], prep["wb_t"] = tr(att.attention_a[0].weight.data), tr(att.attention_b[0].weight.data)
            else:
                prep["wa_t"] = tr(att.attention[0].weight.data)
                if prep["wa_t"].shape[0] % 32 == 0 and prep["wa_t"].shape[1] % 16 == 0:
                    # matrix-core image of Wa^T for the one-pass scorer backward.  It reads the transpose made by an EARLIER job
                    # of the same launch - so it is made from the weight itself, by a transposing fragment job
                    prep["wa_t_frag"] = torch.empty_like(prep["wa_t"])
                    jobs.append((ops.PREP_FRAG_T, att.attention[0].weight.data, prep["wa_t_frag"]))
            if self.merge_enable:
                m = self.merge
                wkv = m.attn.to_kv.weight.data
                if lean_merge:
                    # the projection-free Merge (mca2.hip) reads to_kv / to_q as they are; only to_out is wanted transposed
                    prep["merge_t"] = (None, None, tr(m.attn.to_out[0].weight.data))
                else:
                    if wkv.shape[0] % 32 == 0 and wkv.shape[1] % 16 == 0:     # matrix-core image of Wkv (one-kernel cross attention)
                        prep["wkv_frag"] = torch.empty_like(wkv)
                        jobs.append((ops.PREP_FRAG, wkv, prep["wkv_frag"]))
                    prep["merge_t"] = (tr(m.attn.to_kv.weight.data), tr(m.attn.to_q.weight.data), tr(m.attn.to_out[0].weight.data))
                prep["q_old"] = torch.empty_like(m.global_q_mm.data)
                jobs.append((ops.PREP_COPY, m.global_q_mm.data, prep["q_old"]))
        return jobs, prep

    def prepare_step(self, backward=True):
        jobs, prep = self.prep_jobs(backward)
        if jobs:
            ops.prep_batch(jobs)
        return prep

    def _merge_w(self, plan: Optional[BagPlan], need_t=False, q=None, tr=None, wkv_frag=None, x_rows=None, prepared=False, own=None, rep=1.0):
        m = self.merge
        if need_t and tr is None:
            tr = (ops.transpose(m.attn.to_kv.weight.data), ops.transpose(m.attn.to_q.weight.data),
                  ops.transpose(m.attn.to_out[0].weight.data))
        drop = m.dropout if (plan is not None and plan.training) else 0.0
        q = m.global_q_mm.data.view(m.k, -1) if q is None else q.view(m.k, -1)
        return ops.MergeW(q, m.norm.weight.data, m.norm.bias.data, m.attn.to_kv.weight.data,
                          m.attn.to_q.weight.data, m.attn.to_out[0].weight.data, m.attn.to_out[0].bias.data, m.g_q_mm,
                          drop_p=drop, drop_seed=plan.mca_seed if plan is not None else 0, prec=self._op_prec, transposes=tr,
                          drop_tick=self._tick, wkv_frag=wkv_frag, x_rows=x_rows, prepared=prepared, own=own, rep=rep)

    # ------------------------------------------------------------------ kernels: feature rows
    def _check_x(self, x):
        if not x.is_cuda:
            raise L.MhimxError("MHIM (mhimx): input bag must be a CUDA tensor; there is no CPU path")
        if x.dim() == 3:
            if x.shape[0] != 1:
                raise L.MhimxError("MHIM processes one bag per call (batch_size=1, as the reference trainer does)")
            x = x[0]
        return x.contiguous().float()

    def _pair(self, x):
        """Paired-plane image of the bag (8 bf16 hi | 8 bf16 lo per 8 consecutive features), made ONCE per step and shared
        by every projection that streams X: the GEMM inner loop then has no fp32 -> bf16 conversion.  None when the
        projection does not run in the 3-term bf16 form or the shape is not tileable."""
        if not self._pairable(x):
            return None
        return ops.pair_planes(x)

    def _pairable(self, x):
        return self._feature_prec(x.shape[0]) == "bf16x3" and x.shape[1] % 32 == 0 and x.shape[0] > 16

    def _feature(self, x, rows=None, drop_p=0.0, drop_seed=0, drop_mask=None, want_pre=False, out=None, pre_out=None, M=None,
                 xp=None, w1p=None, dact=None):
        f = self.feature[0]
        act = L.act_code(self.act, _FEATURE_ACTS)
        nrows = M if M is not None else (rows.shape[0] if rows is not None else x.shape[0])
        if xp is None and nrows >= 1024:
            xp = self._pair(x)
        if xp is not None and nrows > 16:
            return ops.gemm_nt(xp, w1p if w1p is not None else ops.pair_planes(f.weight.data), out=out, rows=rows, bias=f.bias.data, act=act, pre=pre_out,
                               drop_p=drop_p, drop_seed=drop_seed, drop_mask=drop_mask, prec="bf16x3", M=M, drop_tick=self._tick,
                               paired=True, dact=dact)
        if dact is not None:
            raise L.MhimxError("the dact output needs the paired-plane projection path")
        return ops.gemm_nt(x, f.weight.data, out=out, rows=rows, bias=f.bias.data, act=act, pre=pre_out, drop_p=drop_p,
                           drop_seed=drop_seed, drop_mask=drop_mask, prec=self._feature_prec(nrows), M=M, drop_tick=self._tick)

    def _feature_prec(self, nrows):
        """'auto' = 'bf16x3' (3 MFMAs per tile step, ~2^-16 relative, fp32 range): the only 16-bit form that keeps
        INSTANCE-level quantities (scores, attention, hence the top-k) at fp32-parity when attention is peaked.
        'f16s' (2 MFMAs: activation in one fp16 term, weight hi+lo) is the opt-in fast mode: bag logits stay
        within 1e-4 when attention is diffuse (errors average over instances) but per-instance scores carry
        ~1e-4 relative error, 2.7e-3 on the bag feature with a x20-sharpened scorer (measured)."""
        if self.prec != "auto":
            return self.prec
        return "bf16x3"

    @property
    def _op_prec(self):
        """Matrix-core form of the scorer / Merge GEMMs: the library runs every form but exact 'f32' as 3-term bf16 there (their
        outputs are instance scores and gradients), so 'auto' and 'f16s' mean 'bf16x3' for these operators."""
        return "f32" if self.prec == "f32" else "bf16x3"

    def _next_seed(self, teacher=False):
        """Seed of the next counter-hash stream.  ``teacher``: the stream belongs to a teacher pass (forward_teacher / the trainer's teacher
        head).  Teacher and student are built alike and call this in step, so without the role salt the trainer's teacher and student shared
        ONE feature-dropout mask (found by the c2 parity test that reads the masks back); the reference draws them independently
        (mhim.py:76 under base_engine.py:37-38)."""
        self._step += 1
        salt = 0xA24BAED4963EE407 if teacher else 0
        return (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._step * 0xD1B54A32D192ED03 + salt) & 0xFFFFFFFFFFFFFFFF

    # ------------------------------------------------------------------ student bag forward / backward
    def _bag_forward(self, x, plan: BagPlan, xp=None, prep=None):
        prep = prep or {}
        E = self.mlp_dim
        Lrows = plan.L
        dev = x.device
        need_pre = L.act_code(self.act, _FEATURE_ACTS) == L.ACT["gelu"] and plan.training
        merging = self.merge_enable and plan.R > 0
        k = self.merge.k if merging else 0
        # One buffer [rows | k merged tokens].  With plan.merge_first the rows are [merge (R) | stay (Lk)], so the pool's
        # input [stay rows | merged tokens] is ONE contiguous segment (no separate 5-row scorer / score kernels).
        Hbuf = torch.empty((Lrows + k, E), device=dev)
        H = Hbuf[:Lrows]
        # backward through act + dropout: the projection's epilogue can emit d out / d pre (act'(pre) * keep/(1-p)) instead of
        # the pre-activation — the backward is then one multiply with no erf / exp / hash (paired-plane kernel only)
        use_dact = plan.training and xp is not None and E % 128 == 0 and x.shape[1] % 32 == 0 and Lrows >= 64
        DACT = torch.empty((Lrows, E), device=dev) if use_dact else None
        PRE = torch.empty((Lrows, E), device=dev) if (need_pre and not use_dact) else None
        p = self.dropout_p if plan.training else 0.0
        self._feature(x, plan.rows, p, plan.drop_seed, plan.drop_mask, out=H, pre_out=PRE, M=Lrows, xp=xp, w1p=prep.get("w1p"),
                      dact=DACT)
        saved = {"H": H, "Hbuf": Hbuf, "PRE": PRE, "DACT": DACT, "prep": prep}
        sc = self._scorer(prep.get("wa_frag"))
        if merging:
            mw = self._merge_w(plan, need_t=False, wkv_frag=prep.get("wkv_frag"))
            q_old = prep.get("q_old")
            if q_old is None and plan.training:
                q_old = self.merge.global_q_mm.data.clone()
            # in-forward EMA of the global queries (merge.py:142-143), written in place: the pre-update values the
            # backward needs were snapshotted above
            q_param = getattr(plan, "q_out", None)                          # (a data-parallel step sends the EMA result to scratch)
            if q_param is None:
                q_param = self.merge.global_q_mm.data.view(self.merge.k, -1)
            Hm = H[:plan.R] if plan.merge_first else H[plan.Lk:]
            z_tok, _, mws = ops.merge_fwd(mw, Hm, z_out=Hbuf[Lrows:], update_q=plan.training,
                                          q_out=q_param if plan.training else None)
            if plan.merge_first:
                st = ops.abmil_pool_fwd(sc, Hbuf[plan.R:], None)
            else:
                st = ops.abmil_pool_fwd(sc, H[:plan.Lk], z_tok)
            saved.update(z_tok=z_tok, mws=mws, q_old=q_old)
        else:
            st = ops.abmil_pool_fwd(sc, H, None)
        saved["pool"] = st
        return st.z, saved

    def _bag_backward(self, x, plan: BagPlan, saved, g_z, out=None, defer=None, mid_hook=None):
        """Returns {param name: gradient}.  ``out`` may map names to preallocated (flat-buffer) views to fill.  ``defer``
        (ops.ReduceList): the final stages of the weight / bias gradient reductions are queued on it; the caller runs
        ``ops.reduce_flush`` before reading any gradient.  ``mid_hook``: called once every gradient except the projection's
        (feature.0.weight / .bias) is final - pending reductions are flushed first - so a data-parallel trainer can start
        all-reducing those while the projection's weight-gradient GEMM, the longest kernel of the backward, still runs."""
        out = out or {}
        E = self.mlp_dim
        dev = x.device
        H, PRE, st = saved["H"], saved["PRE"], saved["pool"]
        prep = saved.get("prep") or {}
        sc = self._scorer()
        att = self.online_encoder.attention
        merging = self.merge_enable and plan.R > 0
        mf = merging and plan.merge_first
        dHbuf = torch.empty_like(saved["Hbuf"])
        dH = dHbuf[:plan.L]
        grads = {}
        pool_g = {"dT1": dHbuf[plan.R:] if mf else (dH[:plan.Lk] if merging else dH)}
        pre = "online_encoder.attention."
        for key, nm in (("d_wa", "attention_a.0.weight" if self.online_encoder.gated else "attention.0.weight"),
                        ("d_wb", "attention_b.0.weight"),
                        ("d_wc", "attention_c.weight" if self.online_encoder.gated else "attention.2.weight")):
            if pre + nm in out:
                pool_g[key] = out[pre + nm]
        if self.online_encoder.gated:
            g = ops.abmil_pool_bwd(sc, st, g_z, prep.get("wa_t") if "wa_t" in prep else ops.transpose(att.attention_a[0].weight.data),
                                   prep.get("wb_t") if "wb_t" in prep else ops.transpose(att.attention_b[0].weight.data), grads=pool_g)
            grads["online_encoder.attention.attention_a.0.weight"] = g["d_wa"]
            grads["online_encoder.attention.attention_b.0.weight"] = g["d_wb"]
            grads["online_encoder.attention.attention_c.weight"] = g["d_wc"]
        else:
            g = ops.abmil_pool_bwd(sc, st, g_z, prep.get("wa_t") if "wa_t" in prep else ops.transpose(att.attention[0].weight.data),
                                   grads=pool_g, defer=defer, wa_t_frag=prep.get("wa_t_frag"))
            grads["online_encoder.attention.attention.0.weight"] = g["d_wa"]
            grads["online_encoder.attention.attention.2.weight"] = g["d_wc"]
        if self.merge_enable and plan.R > 0:
            # LayerNorm(global_q) backward uses the PRE-update queries (the reference sees post-update values through
            # an in-place .data write, a 1e-4-relative quirk: SURVEY.md §7 H7)
            mw = self._merge_w(plan, need_t=True, q=saved["q_old"], tr=prep.get("merge_t"))
            mgr = {"dX": dH[:plan.R] if mf else dH[plan.Lk:]}
            for key, nm in (("d_ln_w", "merge.norm.weight"), ("d_ln_b", "merge.norm.bias"), ("d_wkv", "merge.attn.to_kv.weight"),
                            ("d_wq", "merge.attn.to_q.weight"), ("d_wo", "merge.attn.to_out.0.weight"),
                            ("d_bo", "merge.attn.to_out.0.bias")):
                if nm in out:
                    mgr[key] = out[nm]
            mg = ops.merge_bwd(mw, H[:plan.R] if mf else H[plan.Lk:], dHbuf[plan.L:] if mf else g["dT2"], saved["mws"], grads=mgr,
                               defer=defer)
            grads["merge.norm.weight"], grads["merge.norm.bias"] = mg["d_ln_w"], mg["d_ln_b"]
            grads["merge.attn.to_kv.weight"], grads["merge.attn.to_q.weight"] = mg["d_wkv"], mg["d_wq"]
            grads["merge.attn.to_out.0.weight"], grads["merge.attn.to_out.0.bias"] = mg["d_wo"], mg["d_bo"]
        if mid_hook is not None:
            if defer is not None:
                ops.reduce_flush(defer)
            mid_hook()
        p = self.dropout_p if plan.training else 0.0
        if saved.get("DACT") is not None:
            _, db1 = ops.mul_colsum(dH, saved["DACT"], colsum_out=out.get("feature.0.bias"), defer=defer)
        else:
            _, db1 = ops.act_bwd(dH, H, PRE, L.act_code(self.act, _FEATURE_ACTS), p, plan.drop_seed, plan.drop_mask, plan.rows,
                                 colsum_out=out.get("feature.0.bias"), want_colsum=True, drop_tick=self._tick)
        splits = 8 if plan.L >= 2048 else 1
        grads["feature.0.weight"] = ops.gemm_tn(dH, x, out=out.get("feature.0.weight"), rows=plan.rows, splits=splits,
                                                prec="f32" if self.prec == "f32" else "bf16x3", M=plan.L, defer=defer)
        grads["feature.0.bias"] = db1
        return grads

    # ------------------------------------------------------------------ student bag forward / backward, bag-ordered buffers
    def single_projection_ok(self, x):
        """Shapes ops.bag_project takes for this model's feature projection (any baseline): E = 512, D % 32 == 0, the 3-term bf16 form."""
        return (self.mlp_dim == 512 and x.shape[1] % 32 == 0 and x.shape[0] >= 64 and self._feature_prec(x.shape[0]) == "bf16x3"
                and ops.bag_wgrad_ok(x, self.mlp_dim, x.shape[0]))

    def bag_ordered_ok(self, x):
        """The fused trainer's single-pass form: one projection launch for teacher AND student over the raw bag (ops.bag_project),
        feature rows kept in BAG order and every consumer gathering rows by index.  Needs the shapes the one-pass kernels are built
        for (E = 512, scorer width 128, plain scorer, 8 x 64 merge heads) and the 3-term bf16 form."""
        att = self.online_encoder.attention if self.baseline == "attn" else None
        return (self.baseline == "attn" and not self.online_encoder.gated and self.mlp_dim == 512 and x.shape[1] % 32 == 0
                and att.attention[0].weight.shape[0] == 128 and self._feature_prec(x.shape[0]) == "bf16x3" and x.shape[0] >= 64
                and self.n_classes <= 4 and L.act_code(self.da_act, _SCORER_ACTS) in (L.ACT["relu"], L.ACT["gelu"], L.ACT["tanh"], 0))

    def _bag_forward_nat(self, x, plan: BagPlan, Hbuf, DACT, rows_all, prep):
        """Student forward on bag-ordered feature rows Hbuf [N + k, E] (rows N.. receive the k merged tokens).
        rows_all int64 [R + Lk + k] = [rows to merge | rows that stay | N .. N+k-1]; None: every row takes part, no merge."""
        N = x.shape[0]
        sc = self._scorer(prep.get("wa_frag"))
        saved = {"Hbuf": Hbuf, "DACT": DACT, "rows_all": rows_all, "prep": prep}
        if rows_all is not None:
            # (prep["merge_ws"]: the parameter-only part of Merge already ran as a job of the step's preparation launch)
            mw = self._merge_w(plan, need_t=False, wkv_frag=prep.get("wkv_frag"), x_rows=rows_all[:plan.R], prepared="merge_ws" in prep)
            q_param = getattr(plan, "q_out", None)                          # (a window step sends the per-bag EMA result elsewhere)
            if q_param is None:
                q_param = self.merge.global_q_mm.data.view(self.merge.k, -1)
            k = self.merge.k
            if _SPLIT_POOL and "merge_ws" in prep and k <= 6 and rows_all.numel() - plan.R > k:
                # (round 5) the student's scorer over the rows that stay does not need the tokens, and Merge's rows pass does not need the
                # scorer: ONE launch runs both (the Merge row tiles at its front); then the Merge tail makes the tokens, and the pool's
                # finalize launch scores those k rows itself (mhimx_pool_io.phase) - the scorer launch is off the serial chain
                st, rode = ops.abmil_pool_fwd_split(sc, Hbuf, rows_all[plan.R:], k, ride_merge=(mw, Hbuf, prep["merge_ws"]))
                _, _, mws = ops.merge_fwd(mw, Hbuf, z_out=Hbuf[N:], update_q=plan.training, q_out=q_param if plan.training else None,
                                          ws=prep["merge_ws"], rows_done=rode)
                ops.abmil_pool_fwd_finish(sc, st, wa_t=prep.get("wa_t"), tail_row0=N)
            else:
                _, _, mws = ops.merge_fwd(mw, Hbuf, z_out=Hbuf[N:], update_q=plan.training, q_out=q_param if plan.training else None,
                                          ws=prep.get("merge_ws"))
                st = ops.abmil_pool_fwd(sc, Hbuf, None, rows1=rows_all[plan.R:])
            saved.update(mws=mws, q_old=prep.get("q_old"))
        else:
            st = ops.abmil_pool_fwd(sc, Hbuf[:N], None)
        saved["pool"] = st
        return st.z, saved

    def _bag_backward_nat(self, x, plan: BagPlan, saved, g_z, out, defer=None, mid_hook=None, accumulate=False, wgrad_park=None,
                          ride_tail=False):
        """Backward of _bag_forward_nat: every gradient buffer is bag-ordered too (dH [N + k, E]); the rows that took part are
        gathered once more by the activation backward (ops.rows_dpre) and the projection's weight-gradient GEMM."""
        N = x.shape[0]
        Hbuf, rows_all, st, prep = saved["Hbuf"], saved["rows_all"], saved["pool"], saved["prep"]
        sc = self._scorer()
        dHbuf = torch.empty_like(Hbuf)
        grads = {}
        pre = "online_encoder.attention.attention."
        pool_g = {"dT1": dHbuf if rows_all is not None else dHbuf[:N]}
        for key, nm in (("d_wa", "0.weight"), ("d_wc", "2.weight")):
            if pre + nm in out:
                pool_g[key] = out[pre + nm]
        mw = mgr = None
        if rows_all is not None:
            mw = self._merge_w(plan, need_t=True, q=saved["q_old"], tr=prep.get("merge_t"), x_rows=rows_all[:plan.R])
            mgr = {"dX": dHbuf}
            for key, nm in (("d_ln_w", "merge.norm.weight"), ("d_ln_b", "merge.norm.bias"), ("d_wkv", "merge.attn.to_kv.weight"),
                            ("d_wq", "merge.attn.to_q.weight"), ("d_wo", "merge.attn.to_out.0.weight"),
                            ("d_bo", "merge.attn.to_out.0.bias")):
                if nm in out:
                    mgr[key] = out[nm]
            if len(mgr) == 7:
                # the Merge backward's first stage (parameters x d tokens) rides in the pool backward's rows launch, behind a gate on the
                # tile that writes the tokens' gradient rows dHbuf[N:] (ops.merge_bwd_park)
                ops.merge_bwd_park(mw, Hbuf, dHbuf[N:], saved["mws"], mgr, accumulate=accumulate, defer=defer)
        g = ops.abmil_pool_bwd(sc, st, g_z, prep["wa_t"], grads=pool_g, defer=defer, wa_t_frag=prep.get("wa_t_frag"), accumulate=accumulate)
        grads[pre + "0.weight"], grads[pre + "2.weight"] = g["d_wa"], g["d_wc"]
        if rows_all is not None:
            mg = ops.merge_bwd(mw, Hbuf, dHbuf[N:], saved["mws"], grads=mgr, defer=defer, accumulate=accumulate)
            grads["merge.norm.weight"], grads["merge.norm.bias"] = mg["d_ln_w"], mg["d_ln_b"]
            grads["merge.attn.to_kv.weight"], grads["merge.attn.to_q.weight"] = mg["d_wkv"], mg["d_wq"]
            grads["merge.attn.to_out.0.weight"], grads["merge.attn.to_out.0.bias"] = mg["d_wo"], mg["d_bo"]
        if mid_hook is not None:
            if defer is not None:
                ops.reduce_flush(defer)
            mid_hook()
        Lr = plan.L
        rows = None if rows_all is None else rows_all[:Lr]
        if wgrad_park is not None and rows is not None and ops.bag_wgrad_ok(x, dHbuf.shape[1], Lr):
            # an accumulation window: only the dPre image (+ the bias gradient) now; the window's ONE product launch sums all its bags
            im = ops.bag_wgrad_image(dHbuf, saved["DACT"], x, rows, Lr, out_b=out.get("feature.0.bias"), accumulate=accumulate, defer=defer)
            wgrad_park.append(im)
            grads["feature.0.weight"], db1 = None, out.get("feature.0.bias")
        elif ops.bag_wgrad_ok(x, dHbuf.shape[1], Lr):
            grads["feature.0.weight"], db1 = ops.bag_wgrad(dHbuf, saved["DACT"], x, rows, Lr, out_w=out.get("feature.0.weight"),
                                                           out_b=out.get("feature.0.bias"), defer=defer, accumulate=accumulate,
                                                           ride_tail=ride_tail)
        else:
            dpre, db1 = ops.rows_dpre(dHbuf, saved["DACT"], rows, Lr, colsum_out=out.get("feature.0.bias"), defer=defer, accumulate=accumulate)
            grads["feature.0.weight"] = ops.gemm_tn(dpre, x, out=out.get("feature.0.weight"), rows=rows, splits=8 if Lr >= 2048 else 1,
                                                    prec="bf16x3", M=Lr, defer=defer, accumulate=accumulate)
        grads["feature.0.bias"] = db1
        return grads

    # ------------------------------------------------------------------ masking (mhim.py:109-179)
    def get_mask(self, ps, i, attn, mrh=None, perm=None, perms=None, generator=None, seed=None):
        """Device-side get_mask.  Returns (len_keep:int, mask_ids [1,ps] int64).  ``perm``/``perms`` inject the
        randperm draws of masking.py:67 (parity tests); otherwise they are drawn on the device."""
        if attn is None:
            return ps, None
        heads = None
        if attn.dim() == 3 or (attn.dim() == 2 and attn.shape[0] != 1):
            # per-head attention of the TransMIL teacher ([1,h,N] or [h,N]): msa_fusion='vote' (masking.py:49-59) — every
            # select call first turns the h rows into per-instance vote counts, then takes the top-k of the votes
            heads = attn.reshape(-1, attn.shape[-1]).contiguous().float()
            score = None
        else:
            score = attn.reshape(-1).contiguous().float()
        dev = attn.device
        perms = list(perms) if perms is not None else [None, None, perm]
        masked, n_masked, len_keep, mask_ids = None, 0, ps, None
        n_draws = [0]

        def run(largest, ratio, rratio, pm):
            nonlocal masked, n_masked, len_keep, mask_ids
            eff = ratio / rratio
            if eff > 1:
                rratio, eff = ratio, 1.0
            k = int(np.ceil(ps * eff))
            n_sel = int(np.ceil(k * rratio)) if rratio < 1.0 else k
            if rratio < 1.0 and pm is None:
                if generator is not None:
                    pm = torch.randperm(k, device=dev, generator=generator)
                else:                         # masking.py:67's torch.randperm as ONE element-wise launch, reproducible from (seed, tick)
                    n_draws[0] += 1
                    pm = ops.random_perm(k, (self._next_seed() if seed is None else seed) + 0x51ED270B * n_draws[0], tick=self._tick, device=dev)
            elif pm is not None and not torch.is_tensor(pm):
                pm = torch.as_tensor(np.asarray(pm), dtype=torch.int64, device=dev)
            sc, lg = score, largest
            if heads is not None:
                sc, lg = ops.vote_scores(heads, k, largest), True
            ids, lk_dev, _ = ops.select_mask(sc, k, n_sel, lg, pm if rratio < 1.0 else None, other=masked)
            if masked is None:
                len_keep = ps - n_sel
            else:
                len_keep = int(lk_dev.item())      # union size is data dependent (v1 recipes only): one host sync
            mask_ids = ids
            masked = ids[len_keep:].contiguous()

        if self.mask_ratio > 0.:
            run(False, self.mask_ratio, 0.001, perms[0])
        if self.mask_ratio_l > 0.:
            run(False, self.mask_ratio_l, 1.0, None)
        mask_ratio_h = self.mask_ratio_h
        if self.mrh_sche is not None:
            mask_ratio_h = self.mrh_sche[i]
        if mrh is not None:
            mask_ratio_h = mrh
        if mask_ratio_h > 0.:
            run(True, mask_ratio_h, self.mask_ratio_hr, perms[2])
        return len_keep, (None if mask_ids is None else mask_ids.view(1, -1))

    def v2_counts(self, ps, i=None, mrh=None):
        """(k, n_sel, len_keep, L_keep, R) of the v2 recipe (HAM mask only: masking.py:30-35,61 + merge.py:163), or None when a
        v1 ratio makes the number of masked rows data dependent."""
        mask_ratio_h = self.mask_ratio_h
        if self.mrh_sche is not None and i is not None:
            mask_ratio_h = self.mrh_sche[i]
        if mrh is not None:
            mask_ratio_h = mrh
        if not (self.mask_ratio == 0 and self.mask_ratio_l == 0 and mask_ratio_h > 0):
            return None
        eff, rr = mask_ratio_h / self.mask_ratio_hr, self.mask_ratio_hr
        if eff > 1:
            rr, eff = mask_ratio_h, 1.0
        k = int(np.ceil(ps * eff))
        n_sel = int(np.ceil(k * rr)) if rr < 1.0 else k
        len_keep = ps - n_sel
        Lk = int(len_keep * self.merge.merge_ratio)
        return k, n_sel, len_keep, Lk, len_keep - Lk

    def device_draw_ok(self, ps, i=None, mrh=None):
        """True when student_rows draws both random subsets inside the select kernel (no torch.randperm, no generator): the v2
        recipe on an order-free pool (ABMIL / DSMIL) with ps <= 16384 and k <= 4096."""
        c = self.v2_counts(ps, i, mrh)
        return c is not None and self.baseline in ("attn", "dsmil") and ps <= 16384 and c[0] <= 4096

    def student_rows(self, ps, i, attn, perm=None, ids_shuffle=None, mrh=None, generator=None, merge_first=False, rows_out=None, seed=None):
        """Row list of one student forward: get_mask (mhim.py:341) + Merge.masking (merge.py:158-176) composed.

        Returns (rows int64 [L] = [rows that stay (L_keep) | rows to merge (R)], L, L_keep, R); with ``merge_first`` the
        two groups are swapped ([merge | stay]: the fused trainer's layout, see _bag_forward).
        Production (no injected draws, v2 recipe, N <= 16384, ABMIL whose pool is order-free): ONE kernel draws both
        random subsets on the device (mhimx_select_rows).  The TransMIL encoder sees the token ORDER (landmark means,
        PPEG grid), so it always takes the literal two-stage form with a device-drawn shuffle.  With injected ``perm`` / ``ids_shuffle`` (parity tests) or v1 masks the reference's
        two-stage form is followed literally: select_mask -> mask_ids, then ids_keep[ids_shuffle].
        ``seed``: the device draw's seed (default: this model's own counter, which starts from the PROCESS seed - replicas of an
        instance-sharded bag pass a seed they share).
        """
        mask_ratio_h = self.mask_ratio_h
        if self.mrh_sche is not None:
            mask_ratio_h = self.mrh_sche[i]
        if mrh is not None:
            mask_ratio_h = mrh
        v2 = self.mask_ratio == 0 and self.mask_ratio_l == 0 and mask_ratio_h > 0
        if (v2 and self.baseline in ("attn", "dsmil") and perm is None and ids_shuffle is None and generator is None and attn is not None
                and attn.numel() == ps and ps <= 16384):
            eff, rr = mask_ratio_h / self.mask_ratio_hr, self.mask_ratio_hr
            if eff > 1:
                rr, eff = mask_ratio_h, 1.0
            k = int(np.ceil(ps * eff))
            n_sel = int(np.ceil(k * rr)) if rr < 1.0 else k
            if k <= 4096:
                len_keep = ps - n_sel
                Lk = int(len_keep * self.merge.merge_ratio)
                R = len_keep - Lk
                if R == 0:
                    raise L.MhimxError("merge_ratio leaves no rows to merge (int(L*merge_ratio) == L)")
                rows = ops.select_rows(attn.reshape(-1).contiguous().float(), k, n_sel, R, self._next_seed() if seed is None else seed, tick=self._tick,
                                       merge_first=merge_first, out=None if rows_out is None else rows_out[:len_keep])
                return rows, len_keep, Lk, R
        if generator is None and seed is None and (perm is None or ids_shuffle is None):
            seed = self._next_seed()                           # one seed for this bag's device draws
        len_keep, mask_ids = self.get_mask(ps, i, attn, mrh=mrh, perm=perm, generator=generator, seed=seed)
        if mask_ids is None:
            raise AssertionError("MHIM.forward needs a mask (mask_ratio_h > 0 or a v1 ratio), as the reference does "
                                 "(masking.py:104)")
        Lk = int(len_keep * self.merge.merge_ratio)                         # merge.py:163
        R = len_keep - Lk
        if R == 0:
            raise L.MhimxError("merge_ratio leaves no rows to merge (int(L*merge_ratio) == L)")
        dev = mask_ids.device
        if ids_shuffle is None and generator is None:
            # merge.py:165-170's shuffle of the kept rows: rows = ids_keep[pi], pi a keyed pseudo-random permutation - one launch, no sort
            rows = ops.random_perm(len_keep, seed ^ 0x3C6EF372FE94F82B, tick=self._tick, src=mask_ids.view(-1)[:len_keep].contiguous())
        else:
            if ids_shuffle is None:
                ids_shuffle = torch.randperm(len_keep, device=dev, generator=generator)   # == argsort(rand(L)) in distribution
            elif not torch.is_tensor(ids_shuffle):
                ids_shuffle = torch.as_tensor(np.asarray(ids_shuffle), dtype=torch.int64, device=dev)
            rows = ops.compose_ids(mask_ids.view(-1), ids_shuffle.contiguous())
        if merge_first:
            rows = torch.cat([rows[Lk:], rows[:Lk]])
        if rows_out is not None:
            rows_out[:len_keep].copy_(rows)
            rows = rows_out[:len_keep]
        return rows, len_keep, Lk, R

    # ------------------------------------------------------------------ reference entry points
    @torch.no_grad()
    def forward_teacher(self, x, drop_mask=None, xp=None, w1p=None, wa_frag=None, H=None, tok_full=None):
        x = self._check_x(x)
        p = self.dropout_p if self.training else 0.0           # the trainer keeps the teacher in train mode
        if H is None:                                          # (H: the teacher's feature rows from the trainer's single-pass projection)
            H = self._feature(x, None, p, self._next_seed(teacher=True), drop_mask, xp=xp, w1p=w1p)
        p0 = H.shape[0]
        T2 = None
        if self.merge_test:                                    # eval-mode merge over all rows (mhim.py:196-200)
            mw = self._merge_w(None)
            T2, _, _ = ops.merge_fwd(mw, H, update_q=False)
        if self.baseline == "dsmil":                           # mhim.py:202-205: feature = B [1,C,E], score = instance score
            tok = H if T2 is None else torch.cat([H, T2], 0)
            _, _, B, attn = self.online_encoder(tok, want_attn=True)
            return B.unsqueeze(0), attn[:p0].view(1, -1)
        if self.baseline == "selfattn":
            if tok_full is not None and T2 is None:            # H = tok_full[1:]: the projection wrote under a free first row - no concatenation
                tok_full[:1].copy_(self.online_encoder.cls_token.data.view(1, -1))
                z, attn, v = self._encode(tok_full, return_attn=True, has_cls=True)
            else:
                tok = H if T2 is None else torch.cat([H, T2], 0)
                z, attn, v = self._encode(tok, return_attn=True)
            if self.attn2score:
                score = self._trans_score(v[:p0], attn[0][:, :p0]).view(1, -1)
            else:
                score = attn[self.attn_layer][:, :p0].contiguous().unsqueeze(0)          # [1,h,N] (mhim.py:224-225)
            return z.view(1, -1), score
        wp = self.predictor.weight.data if self.attn2score else None
        st = ops.abmil_pool_fwd(self._scorer(wa_frag), H, T2, wp=wp, bp=self.predictor.bias.data if self.attn2score else None)
        if self.attn2score:
            score = st.pscore                                  # written by the pool's finalize launch (mhimx_pseudo_score's bits)
        else:
            score = ops.softmax_from_stats(st.s, st.stats)[:p0]
        return st.z.view(1, -1), score.view(1, -1)

    @torch.no_grad()
    def forward_test(self, x, return_attn=False, no_norm=False, return_act=False, **kwargs):
        x = self._check_x(x)
        p = self.dropout_p if self.training else 0.0
        H = self._feature(x, None, p, self._next_seed())
        T2 = None
        if self.merge_test:
            T2, _, _ = ops.merge_fwd(self._merge_w(None), H, update_q=False)
        if self.baseline == "dsmil":                           # mhim.py:257-265: no predictor; ([bag, max-instance], B | attn)
            tok = H if T2 is None else torch.cat([H, T2], 0)
            lb, li, B, attn = self.online_encoder(tok, want_attn=return_attn, no_norm=no_norm)
            logits = [lb.view(1, -1), li.view(1, -1)]
            return (logits, attn.view(1, -1)) if return_attn else (logits, B.unsqueeze(0))
        if self.baseline == "selfattn":
            tok = H if T2 is None else torch.cat([H, T2], 0)
            pw, pb = self.predictor.weight.data, self.predictor.bias.data
            if not return_attn:
                return ops.gemm_nt(self._encode(tok).view(1, -1), pw, bias=pb, prec="f32")
            z, attn, v = self._encode(tok, return_attn=True, no_norm=no_norm)
            logits = ops.gemm_nt(z.view(1, -1), pw, bias=pb, prec="f32")
            attn = [a.contiguous().unsqueeze(0) for a in attn]
            if return_act:
                return logits, [attn, v.reshape(v.shape[0], NY.HEADS, NY.DH).permute(1, 0, 2).unsqueeze(0)]
            return logits, attn
        st = ops.abmil_pool_fwd(self._scorer(), H, T2)
        logits = ops.gemm_nt(st.z.view(1, -1), self.predictor.weight.data, bias=self.predictor.bias.data, prec="f32")
        if not return_attn:
            return logits
        a = st.s.clone() if no_norm else ops.softmax_from_stats(st.s, st.stats)
        a = a.view(1, -1)
        if return_act:
            act = H if T2 is None else torch.cat([H, T2], 0)
            return logits, [a, act]
        return logits, a

    # ------------------------------------------------------------------ TransMIL (selfattn) pieces
    def _encode(self, tok, return_attn=False, no_norm=False, has_cls=False):
        """SAttention over the token rows [n, E] (has_cls: row 0 already holds the cls token); train-mode to_out dropouts use the
        counter-based stream."""
        return self.online_encoder(tok, return_attn, no_norm, seeds=(self._next_seed(), self._next_seed()), tick=self._tick,
                                   training=self.training, has_cls=has_cls)

    def _trans_score(self, v, attn0):
        """get_pseudo_score_trans (scoring.py:9-34): v [n, (h d)] (a strided view into the packed qkv rows), attn0 [h, n]
        -> score [n] = max_c softmax_c(Wp to_out(v * attn) + bp[0])."""
        n = v.shape[0]
        f = torch.empty((n, NY.INNER), device=v.device)
        L.check(L.lib().mhimx_scale_heads(ops._stream(), ops._p(v), v.stride(0), ops._p(attn0), attn0.stride(0), NY.DH, n, NY.INNER,
                                          ops._p(f)), "mhimx_scale_heads")
        to_out = self.online_encoder.layer1.attn.to_out[0]
        p = self.online_encoder.layer1.attn.dropout if self.training else 0.0
        if n >= 2048:                                           # on the projection kernel (its own dropout stream; nothing re-applies it)
            f = ops.bag_project(f, [ops.ProjHead(ops.pair_planes(to_out.weight.data), to_out.bias.data, drop_p=p,
                                                 drop_seed=self._next_seed())], act=0, drop_tick=self._tick if p > 0 else None)[0].out
        else:
            f = ops.gemm_nt(f, to_out.weight.data, bias=to_out.bias.data, drop_p=p, drop_seed=self._next_seed(), drop_tick=self._tick,
                            prec="bf16x3")
        cam = ops.gemm_nt(f, self.predictor.weight.data, prec="bf16x3")
        return ops.pseudo_score(None, None, cam, self.predictor.bias.data)

    def _dsmil_student(self, x, plan):
        H = _FeatureFn.apply(self, x, plan, self.feature[0].weight, self.feature[0].bias)
        if self.merge_enable and plan.R > 0:
            z_tok = _MergeFn.apply(self, plan, H[plan.Lk:], *[self._param(n) for n in _MergeFn.NAMES])
            H = torch.cat([H[:plan.Lk], z_tok], 0)
        lb, li, B, _ = self.online_encoder(H)
        return lb, li, B

    def _selfattn_student(self, x, plan):
        if (getattr(plan, "pre", None) is not None and plan.rows is not None and _TOKENS_NODE and (plan.R == 0 or self.merge_enable)
                and plan.L == plan.Lk + plan.R):
            tok = _StudentTokensFn.apply(self, x, plan, self.online_encoder.cls_token, self.feature[0].weight, self.feature[0].bias,
                                         *[self._param(n) for n in _MergeFn.NAMES])
            return self._encode(tok, has_cls=True)
        H = _FeatureFn.apply(self, x, plan, self.feature[0].weight, self.feature[0].bias)
        if self.merge_enable and plan.R > 0:
            z_tok = _MergeFn.apply(self, plan, H[plan.Lk:], *[self._param(n) for n in _MergeFn.NAMES])
            H = torch.cat([H[:plan.Lk], z_tok], 0)                  # merge.py:190-194: kept tokens ++ k merged tokens
        return self._encode(H)

    def _plan_all_rows(self, n):
        return BagPlan(rows=None, L=n, Lk=n, R=0, drop_seed=self._next_seed(), mca_seed=0, training=self.training)

    def _head(self, z, teacher_feat):
        t = None if teacher_feat is None else teacher_feat.detach()
        return _HeadFn.apply(z, t, self.predictor.weight, self.predictor.bias, float(self.temp_t))

    def pure(self, x):
        x = self._check_x(x)
        ps = x.shape[0]
        if not self.training:
            return self.forward_test(x)
        if self.baseline == "dsmil":                           # mhim.py:289-290
            plan = self._plan_all_rows(ps)
            H = _FeatureFn.apply(self, x, plan, self.feature[0].weight, self.feature[0].bias)
            lb, li, _, _ = self.online_encoder(H)
            return [lb.view(1, -1), li.view(1, -1)], 0, ps, ps
        if self.baseline == "selfattn":
            plan = self._plan_all_rows(ps)
            H = _FeatureFn.apply(self, x, plan, self.feature[0].weight, self.feature[0].bias)
            logits, _ = self._head(self._encode(H).view(1, -1), None)
            return logits, 0, ps, ps
        merge_enable, self.merge_enable = self.merge_enable, False     # pure(): no masking, no merging (mhim.py:274-298)
        try:
            plan = self._plan_all_rows(ps)
            z = _BagFn.apply(self, x, plan, *[self._param(n) for n in self._bag_param_names])
        finally:
            self.merge_enable = merge_enable
        logits, _ = self._head(z, None)
        return logits, 0, ps, ps

    def forward_loss(self, student_cls_feat, teacher_cls_feat):
        if teacher_cls_feat is None:
            return 0.
        return self._head(student_cls_feat, teacher_cls_feat)[1]

    @torch.no_grad()
    def _forward_eval(self, x, ps, i, attn, teacher_cls_feat, perm):
        """MHIM.forward with the module in eval mode (mhim.py:318-378 as written: the mask is applied whatever the mode; Merge in
        eval keeps EVERY surviving row and appends the k tokens merged from all of them, merge.py:197-203; no EMA of the global
        queries, no dropout).  No gradient path (the reference's trainer never calls it in eval mode; validation uses forward_test)."""
        len_keep, mask_ids = self.get_mask(ps, i, attn, perm=perm)
        if mask_ids is None:
            raise AssertionError("MHIM.forward needs a mask (mask_ratio_h > 0 or a v1 ratio), as the reference does (masking.py:104)")
        rows = mask_ids.view(-1)[:len_keep].contiguous()
        H = self._feature(x, rows, 0.0, 0, M=len_keep)
        z_tok, _, _ = ops.merge_fwd(self._merge_w(None), H, update_q=False)
        k = z_tok.shape[0]
        if self.baseline == "dsmil":
            lb, li, B, _ = self.online_encoder(torch.cat([H, z_tok], 0))
            cls_loss = 0.
            if teacher_cls_feat is not None:
                cls_loss = DS.SoftTargetCE.apply(B, teacher_cls_feat.detach().reshape(B.shape).float(), float(self.temp_t))
            return [lb.view(1, -1), li.view(1, -1)], cls_loss, ps, len_keep + k
        if self.baseline == "selfattn":
            z = self._encode(torch.cat([H, z_tok], 0)).view(1, -1)
        else:
            z = ops.abmil_pool_fwd(self._scorer(), H, z_tok).z.view(1, -1)
        logits, cls_loss = self._head(z, teacher_cls_feat)
        if teacher_cls_feat is None:
            cls_loss = 0.
        return logits, cls_loss, ps, len_keep + k

    def forward(self, x, attn=None, teacher_cls_feat=None, i=None, pos=None, perm=None, ids_shuffle=None, drop_mask=None):
        x = self._check_x(x)
        ps = x.shape[0]
        if not self.merge_enable:
            raise TypeError("MHIM.forward requires merge_enable=True (the reference's Identity merge rejects the "
                            "second argument, mhim.py:82,351)")
        if not self.training:
            return self._forward_eval(x, ps, i, attn, teacher_cls_feat, perm)
        rows, len_keep, Lk, R = self.student_rows(ps, i, attn, perm=perm, ids_shuffle=ids_shuffle)
        plan = BagPlan(rows=rows, L=len_keep, Lk=Lk, R=R, drop_seed=self._next_seed(), drop_mask=drop_mask,
                       mca_seed=self._next_seed(), training=self.training)
        if self.baseline == "dsmil":                           # mhim.py:355-364
            lb, li, B = self._dsmil_student(x, plan)
            cls_loss = 0.
            if teacher_cls_feat is not None:
                cls_loss = DS.SoftTargetCE.apply(B, teacher_cls_feat.detach().reshape(B.shape).float(), float(self.temp_t))
            return [lb.view(1, -1), li.view(1, -1)], cls_loss, ps, Lk + self.merge.k
        if self.baseline == "selfattn":
            z = self._selfattn_student(x, plan).view(1, -1)
        else:
            z = _BagFn.apply(self, x, plan, *[self._param(n) for n in self._bag_param_names])
        logits, cls_loss = self._head(z, teacher_cls_feat)
        if teacher_cls_feat is None:
            cls_loss = 0.
        return logits, cls_loss, ps, Lk + self.merge.k
