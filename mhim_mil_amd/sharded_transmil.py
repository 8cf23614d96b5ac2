"""MHIM(TransMIL) train step on ONE bag whose rows are split across the ranks (third line; the ABMIL form is sharded.py).

Rank r holds the bag rows [lo_r, lo_r + n_r).  The encoder (mhim_modules/baseline.py:244-288) wants the TOKEN sequence
[zeros(pad) | cls | tokens] in aligned blocks of T / W rows (nystrom_sharded.py), and the tokens of a masked bag are the kept rows in the
order merge.py:165-170 shuffled them into, followed by the k merged tokens - so between the instance shards and the encoder stands a
re-balancing exchange:

  teacher   feature rows of the local bag rows -> ALL-TO-ALL into the token blocks of [pad | cls | rows 0 .. N-1]
            sequence-parallel encoder with the cls token's attention row over the local tokens (ShardedTransLayerFn need_attn)
            pseudo score of the local tokens (scoring.py:9-34 is per token)             -> all-gather of T floats
  select    replicated (shared seed / injected draws): the same row list on every rank
  student   feature rows of the local KEPT rows; the rows to merge are summed into a replicated [R, E] block (one non-zero contributor per
            row) and Merge runs replicated; the rows that stay go to the rank that owns their token position (ALL-TO-ALL; who sends what
            where follows from the replicated row list: no index traffic), the merged tokens are placed by the rank(s) owning the tail
            sequence-parallel encoder -> cls feature on every rank; head replicated
  backward  the exchanges in reverse (all-to-all of d tokens; d merged tokens summed over the owners; the merge rows' gradient is read
            off the replicated dX); parameter gradients are local partial sums (replicated terms counted on rank 0 only) and ride in ONE
            all-reduce of the flat gradient; Adam + EMA replicated.
The PPEG between the layers works on a band of its token grid per rank - own grid rows, three halo rows either side, the wrap cells: one
all-to-all each way (nystrom_sharded.ShardedPPEGFn).  Dropout streams are
keyed by rank-local element indices with rank-salted seeds: valid, independent masks, not the single-process ones (the equality tests run
with dropout 0).  One host read-back per exchange plan (the W x W row counts).
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import nystrom as NY
from . import ops
from .mhim import BagPlan, _FeatureFn, _MergeFn, _FEATURE_ACTS
from .nystrom_sharded import sharded_sattention


def _all_to_all_rows(comm, send, send_counts, recv_counts):
    return comm.all_to_all_rows(send, send_counts, recv_counts)


class ExchangePlan:
    """Who sends which token row where, from replicated data only.  ``ids`` [n_tok] int64: the bag row behind token j (token j sits at
    sequence row pad + 1 + j); ``bounds``: the shards' first bag rows (W + 1 entries); blocks of Tr sequence rows per rank.
    A rank's send buffer is its rows in ascending token order (= grouped by destination); what it receives comes grouped by source, each
    group in ascending token order - ``place`` [n_recv] are the block rows they go to."""

    def __init__(self, ids, bounds, pad, Tr, comm):
        W, r, dev = comm.world, comm.rank, ids.device
        self.Tr, self.comm, self.run = Tr, comm, None
        if W == 1:                                          # everything stays: token j -> block row pad + 1 + j, no device work, no read-back
            n = ids.numel()
            self.send_counts, self.recv_counts = [n], [n]
            self.run, self.place, self.local = (pad + 1, pad + 1 + n), None, None
            return
        b = torch.as_tensor(bounds[1:-1], dtype=torch.int64, device=dev)
        owner = torch.bucketize(ids, b, right=True)                      # shard of the row behind every token
        pos = pad + 1 + torch.arange(ids.numel(), device=dev)
        dest = torch.div(pos, Tr, rounding_mode="floor")
        cnt = torch.bincount(owner * W + dest, minlength=W * W).view(W, W).tolist()          # the ONE host read-back
        self.send_counts = cnt[r]
        self.recv_counts = [cnt[q][r] for q in range(W)]
        mine = torch.nonzero(dest == r).view(-1)                         # tokens of my block, ascending
        order = torch.sort(owner[mine], stable=True).indices             # ... grouped by source rank
        self.place = (pos[mine][order] - r * Tr).contiguous()
        self.local = torch.nonzero(owner == r).view(-1)                  # my rows' token indices, ascending (= the send order)


class IdentityPlan:
    """ExchangePlan for tokens in BAG order (the teacher: token j = bag row j): everything follows from the shard bounds on the host - no
    device work, no read-back - and what a rank receives is ONE contiguous run of its block (sources ascending = positions ascending)."""

    def __init__(self, N, bounds, pad, Tr, comm):
        W, r = comm.world, comm.rank

        def overlap(q, d):                                   # rows of shard q whose sequence position falls into block d
            lo, hi = max(pad + 1 + bounds[q], d * Tr), min(pad + 1 + bounds[q + 1], (d + 1) * Tr)
            return max(hi - lo, 0)

        self.send_counts = [overlap(r, d) for d in range(W)]
        self.recv_counts = [overlap(q, r) for q in range(W)]
        a = max(pad + 1, r * Tr)
        self.run = (a - r * Tr, a - r * Tr + sum(self.recv_counts))      # block rows [start, end) the received rows fill, in order
        self.place = None
        self.Tr, self.comm = Tr, comm


class _AssembleTokens(torch.autograd.Function):
    """This rank's block [Tr, E] of the token sequence: its share of the exchanged rows (``rows_local``: my rows in ascending token order)
    and - ``tail`` [k, E], replicated, sequence rows tail_pos .. tail_pos + k - 1 - the tail rows the block owns.  Everything else is zero
    (front padding, the cls slot).  Backward: the reverse exchange; d tail summed over the owners (every rank needs it: Merge is replicated)."""

    @staticmethod
    def forward(ctx, rows_local, tail, plan: ExchangePlan, tail_pos):
        comm, Tr = plan.comm, plan.Tr
        E = rows_local.shape[1]
        inplace = getattr(plan, "block", None)                # (world 1: the producer wrote its rows into the block already)
        if inplace is not None:
            a, b = plan.run
            assert rows_local.data_ptr() == inplace[a:b].data_ptr() and rows_local.shape[0] == b - a
            block = inplace
            block[:a].zero_()
            block[b:].zero_()
            got = None
        else:
            got = _all_to_all_rows(comm, rows_local.contiguous(), plan.send_counts, plan.recv_counts)
        if got is None:
            pass
        elif plan.place is None:                             # bag order (or world 1): one contiguous run
            a, b = plan.run
            block = torch.empty((Tr, E), device=rows_local.device)
            block[:a].zero_()
            block[b:].zero_()
            block[a:b].copy_(got)
        else:
            block = torch.zeros((Tr, E), device=rows_local.device)
            if got.shape[0]:
                block.index_copy_(0, plan.place, got)
        own = None
        if tail is not None:
            k = tail.shape[0]
            g0 = comm.rank * Tr
            a, b = max(tail_pos, g0), min(tail_pos + k, g0 + Tr)
            if a < b:
                own = (a - tail_pos, b - tail_pos, a - g0)
                block[a - g0:b - g0].copy_(tail[a - tail_pos:b - tail_pos])
        ctx.plan, ctx.own, ctx.n_local = plan, own, rows_local.shape[0]
        ctx.tail_shape = None if tail is None else tuple(tail.shape)
        return block

    @staticmethod
    def backward(ctx, dblock):
        plan, comm = ctx.plan, ctx.plan.comm
        dblock = dblock.contiguous()
        if plan.place is None:
            dgot = dblock[plan.run[0]:plan.run[1]]
        else:
            dgot = dblock.index_select(0, plan.place) if plan.place.numel() else dblock[:0]
        drows = _all_to_all_rows(comm, dgot, plan.recv_counts, plan.send_counts)
        dtail = None
        if ctx.tail_shape is not None:
            dtail = torch.zeros(ctx.tail_shape, device=dblock.device)
            if ctx.own is not None:
                i0, i1, b0 = ctx.own
                dtail[i0:i1].copy_(dblock[b0:b0 + (i1 - i0)])
            comm.all_reduce_sum(dtail)
        return drows, dtail, None, None


class _GatherMergeRows(torch.autograd.Function):
    """The rows to merge as a replicated [R, E] block: every rank's rows at their positions of the merge list, summed (one non-zero
    contributor per row: exact).  Backward: Merge runs replicated, so every rank holds the same dX and reads its own rows off it."""

    @staticmethod
    def forward(ctx, Hm_local, merge_pos, R, comm):
        full = torch.zeros((R, Hm_local.shape[1]), device=Hm_local.device)
        if merge_pos.numel():
            full.index_copy_(0, merge_pos, Hm_local.contiguous())
        ctx.pos = merge_pos
        return comm.all_reduce_sum(full)

    @staticmethod
    def backward(ctx, dfull):
        return dfull.index_select(0, ctx.pos), None, None, None


def seq_layout(n_tok, world):
    """(pad, T, Tr) of a sequence of cls + n_tok tokens over ``world`` ranks (nystrom_attention.py:70-73 pads to 256 at the FRONT)."""
    n = 1 + n_tok
    pad = (NY.LANDMARKS - n % NY.LANDMARKS) % NY.LANDMARKS
    T = pad + n
    if NY.LANDMARKS % world or (T // world) % 64:
        raise L.MhimxError(f"sharded TransMIL: {world} ranks do not divide a sequence of {T} rows into aligned blocks (world | 256, "
                           "T / world a multiple of 64)")
    return pad, T, T // world


def transmil_step(tr, x_local, label, perm=None, ids_shuffle=None, i=None):
    """One MHIM(TransMIL) train step of sharded.ShardedBagTrainer ``tr`` (teacher -> select -> student -> head -> backward -> all-reduce ->
    Adam + EMA).  Returns (logits [C], losses [3]); identical on every rank."""
    s, t, fl, cm = tr.s, tr.t, tr.flat, tr.comm
    gv = fl.grad_views
    x = s._check_x(x_local)
    n, dev, E = x.shape[0], x.device, s.mlp_dim
    W = cm.world
    counts = tr.counts if tr.counts is not None else [n] * W
    N, lo = sum(counts), sum(counts[:cm.rank])
    assert counts[cm.rank] == n, "counts[rank] must equal the local row count"
    bounds = [sum(counts[:q]) for q in range(W + 1)]
    shared_seed, local_seed = tr._seeds()
    mix = lambda a: (local_seed + a * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF

    # ---- teacher: feature rows -> token blocks -> sequence-parallel encoder with the cls attention -> per-token score -> all-gather
    pre = None
    with torch.no_grad():
        p = t.dropout_p if t.training else 0.0
        if s.single_projection_ok(x) and t.single_projection_ok(x) and s.act == t.act:
            # teacher AND student feature rows of all local bag rows in ONE pass over the raw rows (as FusedTrainer: the reference's student
            # projects every row before it masks, mhim.py:335-336); the student's token rows are then a gather, its projection gradient the
            # matrix-core-image pair on (d tokens, d out / d pre in fp16)
            heads = [ops.ProjHead(ops.pair_planes(t.feature[0].weight.data), t.feature[0].bias.data, drop_p=p, drop_seed=mix(1)),
                     ops.ProjHead(ops.pair_planes(s.feature[0].weight.data), s.feature[0].bias.data, drop_p=s.dropout_p, drop_seed=mix(4),
                                  want_dact=True)]
            pad_t, T_t, Tr_t = seq_layout(N, W)
            tblock = None
            if W == 1:                                      # the teacher's feature rows land in its token block (no copy into it)
                tblock = torch.empty((Tr_t, E), device=dev)
                heads[0].out = tblock[pad_t + 1:pad_t + 1 + N]
            ops.bag_project(x, heads, act=L.act_code(s.act, _FEATURE_ACTS))
            Ht, pre = heads[0].out, (heads[1].out, heads[1].dact)
        else:
            tblock = None
            Ht = t._feature(x, None, p, mix(1))
        pad_t, T_t, Tr_t = seq_layout(N, W)
        plan_t = IdentityPlan(N, bounds, pad_t, Tr_t, cm)
        plan_t.block = tblock
        blk = _AssembleTokens.apply(Ht, None, plan_t, 0)
        del Ht
        t_feat, attn, v = sharded_sattention(t.online_encoder, blk, pad_t, 1 + N, cm, return_attn=True, seeds=(mix(2), mix(3)),
                                             training=t.training)
        if t.attn2score:
            sc_loc = t._trans_score(v, attn[0])                                          # [Tr]: garbage on the pad / cls rows, cut below
            score = cm.all_gather(sc_loc.contiguous()).reshape(-1)[pad_t + 1:].view(1, -1)
        else:
            a = attn[t.attn_layer].contiguous()                                         # [8, Tr] -> [1, 8, N] (mhim.py:224-225)
            score = cm.all_gather(a).permute(1, 0, 2).reshape(NY.HEADS, -1)[:, pad_t + 1:].contiguous().unsqueeze(0)
        del blk, attn, v

    # ---- select: replicated
    rows, len_keep, Lk, R = s.student_rows(N, i, score, perm=perm, ids_shuffle=ids_shuffle, generator=None, seed=shared_seed)
    from .sharded import partition_rows
    if W == 1:                                                                           # (no host sync: every row is local)
        rows_local, n_stay, merge_pos = rows, Lk, torch.arange(rows.numel() - Lk, device=dev)
    else:
        rows_local, n_stay, merge_pos = partition_rows(rows, Lk, lo, n)
    n_loc = rows_local.numel()
    # a shard without kept rows is not supported - and EVERY rank must say so: the row list and the bounds are replicated, so every rank
    # counts every shard's kept rows alike and all of them raise together (a rank-local raise left the others hanging in the next collective)
    if W > 1:
        owner_all = torch.bucketize(rows[:len_keep], torch.as_tensor(bounds[1:-1], dtype=torch.int64, device=rows.device), right=True)
        per_shard = torch.bincount(owner_all, minlength=W).tolist()
    else:
        per_shard = [n_loc]
    if min(per_shard) == 0:
        raise L.MhimxError(f"a shard without kept rows is not supported (bag too small for this many ranks): kept rows per shard {per_shard}")
    k = s.merge.k
    pad_s, T_s, Tr_s = seq_layout(Lk + k, W)
    plan_x = ExchangePlan(rows[:Lk], bounds, pad_s, Tr_s, cm)
    assert plan_x.local is None or plan_x.local.numel() == n_stay

    # ---- student forward (autograd over kernel-backed nodes; parameter .grad = views of the flat gradient buffer)
    first = cm.rank == 0
    pd = dict(s.named_parameters())
    par = lambda name: pd[name] if first else pd[name].detach()                          # replicated terms: counted once in the SUM
    plan_f = BagPlan(rows=rows_local, L=n_loc, Lk=n_stay, R=n_loc - n_stay, drop_seed=mix(4), mca_seed=shared_seed, training=True)
    plan_f.pre = pre
    H = _FeatureFn.apply(s, x, plan_f, s.feature[0].weight, s.feature[0].bias)           # [n_loc, E]: stay rows first, then rows to merge
    Hm = _GatherMergeRows.apply(H[n_stay:], merge_pos, R, cm)
    plan_m = BagPlan(rows=None, L=R, Lk=0, R=R, drop_seed=0, mca_seed=shared_seed, training=True)
    z_tok = _MergeFn.apply(s, plan_m, Hm, *[par(nm) for nm in _MergeFn.NAMES])           # replicated (the EMA of the queries too)
    blk = _AssembleTokens.apply(H[:n_stay], z_tok, plan_x, pad_s + 1 + Lk)
    z = sharded_sattention(s.online_encoder, blk, pad_s, 1 + Lk + k, cm, seeds=(mix(5), mix(6)), training=True)

    # ---- head (replicated)
    t_in = t_feat.view(-1) if tr.aux_alpha != 0. else None
    d_wp = gv["predictor.weight"] if first else torch.empty_like(gv["predictor.weight"])
    d_bp = gv["predictor.bias"] if first else torch.empty_like(gv["predictor.bias"])
    logits, losses, g_z, _, _ = ops.head_fwd_bwd(z.detach(), t_in, s.predictor.weight.data, s.predictor.bias.data, label,
                                                 temp_t=float(s.temp_t), main_alpha=tr.main_alpha, aux_alpha=tr.aux_alpha,
                                                 d_wp=d_wp, d_bp=d_bp, accumulate=first)

    # ---- backward into the flat buffer, ONE all-reduce, Adam + EMA
    for name in fl.train_names:
        pd[name].grad = None
    torch.autograd.backward([z], [g_z])
    views, grads = [], []
    for name in fl.train_names:
        p_ = pd[name]
        if p_.grad is not None:
            views.append(gv[name])
            grads.append(p_.grad.reshape(gv[name].shape))
        p_.grad = None
    if grads:
        torch._foreach_add_(views, grads)
    cm.all_reduce_sum(fl.grad[:fl.n_train])
    tr.step_count += 1
    ops.tick(tr.opt_step)
    ops.adam_ema(fl.student, fl.grad, fl.m, fl.v, None if fl.same_teacher else fl.teacher, fl.n_train, tr.step_count, lr=tr.lr, beta1=tr.betas[0], beta2=tr.betas[1],
                 eps=tr.eps, weight_decay=tr.wd, grad_scale=1.0, ema_mm=tr.mm, zero_grad=True, step_dev=tr.opt_step)
    tr.last = {"logits": logits, "losses": losses, "patch_num": N, "keep_num": Lk + k, "rows": rows, "len_keep": len_keep,
               "score": score, "teacher_feat": t_feat}
    return logits, losses
