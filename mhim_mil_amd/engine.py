"""Train-step API for the MHIM path.

``CommonMIL`` mirrors the hook object the reference trainer calls (engines/common_mil.py:1-68): same
method names, keyword arguments and return tuples, so ``BaseTrainer.train/validate`` can drive the HIP
model unchanged.

``FusedTrainer`` is the MI355X-native step the benchmark times: teacher forward -> select -> student
forward -> head (CE + distillation) -> hand-derived backward -> [RCCL all-reduce of ONE flat gradient
buffer] -> fused Adam + EMA-teacher kernel.  It bypasses autograd and the per-parameter Python loops of
base_engine.py:155-167 (one flat fp32 buffer each for student, teacher, gradient, Adam m and v), which is
what makes a tens-of-microseconds bag step reachable at all.
"""
from __future__ import annotations

from typing import Optional

import os

import torch

from . import mhim as mh
from . import ops
from .mhim import MHIM, BagPlan


class CommonMIL:
    """Hook object of the reference trainer (engines/common_mil.py)."""

    def __init__(self, args=None, fused=None, graph_cache=0) -> None:
        self.training = True
        self.fused = fused                 # optional optim.FusedAdamEMA: forward_func may run the native forward + backward (its docstring)
        # graph_cache = K > 0 (with fused=): the native forward + backward of the K most recent bag SHAPES as captured hipGraphs - the second
        # bag of a shape is captured (into buffers of its own: the bag is copied in, one launch), every later one replays; other shapes
        # run eagerly as before.  MHIM(ABMIL), one process, accumulation_steps == 1 (FusedTrainer.shape_cached).
        self.graph_cache = int(graph_cache)

    def _native_step(self, tr, bag, label, n_iter, extra):
        """The native forward + backward of one bag: a replay of the shape's captured graph when there is one (graph_cache), else eager."""
        if self.graph_cache > 0 and not extra:
            out = tr.shape_cached("forward_backward", bag, label, i=n_iter, cache=self.graph_cache)
            if out is not None:
                return out
        logits, losses = tr.forward_backward(bag, label, i=n_iter, **extra)
        return logits, losses, tr.last["patch_num"], tr.last["keep_num"]

    def init_func_train(self, args, **kwargs):
        self.training = True

    def init_func_val(self, args, **kwargs):
        self.training = False

    def after_get_data_func(self, args, **kwargs):
        pass

    def after_backward_func(self, args, **kwargs):
        pass

    def final_train_func(self, args, **kwargs):
        pass

    def forward_func(self, args, model, model_ema, bag, label, criterion, batch_size, i, epoch, n_iter, pos, **kwargs):
        """-> (logits, label, aux_loss, patch_num, keep_num, pad_ratio, kn_std)   (common_mil.py:14-48)"""
        dsmil = getattr(args, "baseline", "attn") == "dsmil"
        fz = self.fused
        if (fz is not None and model is fz.model and model_ema is fz.model_ema and model.training and not dsmil
                and args.model == fz.trainer.model_kind and type(criterion) is torch.nn.CrossEntropyLoss and criterion.weight is None
                and criterion.label_smoothing == 0.0 and criterion.reduction == "mean" and batch_size == 1):
            # the loop's loss is main_alpha * CE + aux_alpha * aux (/ accumulation_steps), base_engine.py:99-102: the native step computes
            # exactly that and its gradient (no autograd graph); the loop gets LEAVES - its criterion / backward touch no parameter
            tr = fz.trainer
            tr.main_alpha, tr.aux_alpha = float(getattr(args, "main_alpha", 1.0)), float(args.aux_alpha)
            extra = {k: kwargs[k] for k in ("perm", "ids_shuffle") if k in kwargs}
            logits, losses, patch_num, keep_num = self._native_step(tr, bag, label.view(-1)[:1], n_iter, extra)
            lg = logits.detach().view(batch_size, -1).clone().requires_grad_(True)     # (clone: a replay rewrites the graph's buffers)
            aux = losses[2].detach().clone().requires_grad_(True)
            return lg, label, aux, patch_num, keep_num, 0., 0.
        if args.model == "mhim":
            teacher_feat, score = (None, None)
            if model_ema is not None:
                teacher_feat, score = model_ema.forward_teacher(bag)
            if args.aux_alpha == 0.:
                teacher_feat = None                                        # common_mil.py:24
            if dsmil and teacher_feat is not None:
                teacher_feat = teacher_feat[0]                              # common_mil.py:27: cls_tea[0] = B [C,E]
            # the reference trainer calls forward_func with loader=, device=, others=, idx=, feat= ... (base_engine.py:77-93) and its
            # CommonMIL hands none of them to the model (common_mil.py:30); only the parity tests' injected draws go through
            extra = {k: kwargs[k] for k in ("perm", "ids_shuffle", "drop_mask") if k in kwargs}
            logits, aux_loss, patch_num, keep_num = model(bag, score, teacher_feat, i=n_iter, **extra)
            if dsmil:                                                       # common_mil.py:28
                logits = 0.5 * logits[0].view(batch_size, -1) + 0.5 * logits[1].view(batch_size, -1)
        elif args.model == "mhim_pure":
            logits, aux_loss, patch_num, keep_num = model.pure(bag)
            if dsmil:                                                       # common_mil.py:34-35
                logits = 0.5 * logits[0].view(batch_size, -1) + 0.5 * logits[1].view(batch_size, -1)
        else:
            raise NotImplementedError(f"model {args.model!r} is outside the MHIM hot path")
        return logits, label, aux_loss, patch_num, keep_num, 0., 0.

    def validate_func(self, args, model, bag, label, criterion, batch_size, i, pos, epoch=None, **kwargs):
        """-> (logits, label)   (common_mil.py:56-68)"""
        if args.model not in ("mhim", "mhim_pure"):
            raise NotImplementedError(f"model {args.model!r} is outside the MHIM hot path")
        logits = model.forward_test(bag)
        if getattr(args, "baseline", "attn") == "dsmil":                    # common_mil.py:59-60,66-67
            logits = logits[0]
            logits = 0.5 * logits[0] + 0.5 * logits[1]
        return logits, label


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """The reference's per-iteration cosine schedule (utils.py:199-210): used for `mm_sche` (EMA momentum mm -> 1,
    modules/__init__.py:177-181) and `mrh_sche` (HAM ratio mask_ratio_h -> 0, modules/__init__.py:72-75)."""
    import numpy as np
    warmup_iters = warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * iters / len(iters)))
    sched = np.concatenate((warm, sched))
    assert len(sched) == epochs * niter_per_ep
    return sched


class FlatState:
    """Flat fp32 buffers behind a student/teacher pair.

    Layout: [trainable parameters in named_parameters() order | non-trainable parameters (merge.global_q_mm)].
    Every nn.Parameter of both models becomes a view into its flat buffer, so state_dict()/load_state_dict()
    keep working and the fused optimiser touches each byte once.
    """

    def __init__(self, student: MHIM, teacher: Optional[MHIM], step_merges=True):
        named = list(student.named_parameters())          # de-duplicated (global_q alias appears once)
        # parameters the forward never touches get no gradient; torch.optim.Adam skips those (no update, no weight decay),
        # so they sit in the non-trainable tail here (DSMIL never uses MHIM.predictor: mhim.py:264-265)
        unused = set(getattr(student, "unused_parameter_names", lambda: ())())
        if not step_merges:                               # 'mhim_pure' never calls Merge: its parameters get no gradient either
            unused |= {n for n, _ in named if n.startswith("merge.")}
        self.train_names = [n for n, p in named if p.requires_grad and n not in unused]
        # the projection's weight and bias lead the buffer: their gradient is the last to become final in a backward, so a data-
        # parallel step all-reduces [the rest] first, overlapped with the projection's weight-gradient GEMM (FusedTrainer._mid_hook)
        head = [n for n in ("feature.0.weight", "feature.0.bias") if n in self.train_names]
        self.train_names = head + [n for n in self.train_names if n not in head]
        self.fixed_names = [n for n, p in named if not p.requires_grad or n in unused]
        self.names = self.train_names + self.fixed_names
        dev = named[0][1].device
        sizes = {n: p.numel() for n, p in named}
        self.offsets, off = {}, 0
        for n in self.names:
            self.offsets[n] = off
            off += (sizes[n] + 3) // 4 * 4                  # keep every tensor 16-byte aligned
            if n == self.train_names[-1]:
                self.n_train = off
        self.n_all = (off + 63) // 64 * 64         # (a multiple of every world size up to 64: the mesh form of the gradient exchange -
        self.student = self._adopt(student, dev)   #  reduce-scatter + all-gather on count / world slices - takes the whole buffer)
        # --tea_type same (modules/__init__.py:211-212: the teacher IS the student; base_engine.py:157-158: no EMA): one buffer, adopted once
        self.same_teacher = teacher is not None and teacher is student
        self.teacher = self.student if self.same_teacher else (self._adopt(teacher, dev) if teacher is not None else None)
        self.grad = torch.zeros(self.n_all, device=dev)
        self.m = torch.zeros(self.n_train, device=dev)
        self.v = torch.zeros(self.n_train, device=dev)
        self.grad_views = {}
        pd = dict(student.named_parameters())
        for n in self.train_names:
            self.grad_views[n] = self.grad[self.offsets[n]:self.offsets[n] + sizes[n]].view_as(pd[n])
        self.step = 0

    def _adopt(self, model, dev):
        flat = torch.zeros(self.n_all, device=dev)
        pd = dict(model.named_parameters())
        for n in self.names:
            p = pd[n]
            view = flat[self.offsets[n]:self.offsets[n] + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        return flat


class QueryChain:
    """How Merge's in-forward EMA of the global queries (merge.py:142-143) composes over the ranks of a data-parallel update.

    A single process with accumulation_steps = W runs the EMA bag after bag: q <- mm q + (1 - mm) z_r, r = 0 .. W-1.  W ranks each see
    the step's first queries q0 (nothing orders them), so the same chain is applied to the tokens z_r their forwards produced:
        q <- mm^W q0 + (1 - mm) sum_r mm^(W-1-r) z_r
    - rank r puts w_r z_r, w_r = (1 - mm) mm^(W-1-r), into its slice of the flat buffer's tail, the ONE SUM all-reduce adds them, and
    every rank finishes with mm^W q0 + the sum.  The same contract as an accumulation window (FusedTrainer.window_step, oracle
    ``train_window(q_ema="window")``): it differs from the bag-after-bag order only through d z_r / d q (second order in 1 - mm), where
    averaging the ranks' one-step results would slow the EMA down W-fold (first order)."""

    def __init__(self, offset, numel, mm, world, rank):
        self.off, self.n = int(offset), int(numel)
        self.w = (1.0 - mm) * mm ** (world - 1 - rank)
        self.decay = mm ** world
        self.tokens = None                       # [k, E] tokens of this rank's forward (set by the step)


def sync_flat_gradient(grad: torch.Tensor, student: torch.Tensor, n_train: int, world: int, group=None, comm=None, chain: QueryChain = None) -> float:
    """The ONE collective of a data-parallel update (RCCL on GPUs; any backend works, gloo in the CPU tests).

    Everything that must agree across ranks rides in the same flat buffer: the gradient (first n_train floats) and,
    in the tail, the non-trainable parameters that each rank mutated in its own forward (merge.global_q_mm's EMA,
    merge.py:142-143).  After the SUM all-reduce the tail is averaged back into ``student``; the
    returned scale (1/world) is applied to the gradient inside the fused Adam kernel.  No data-path collective exists:
    bags are independent units.  ``comm``: a ``comm.NativeComm`` (the C-ABI's own RCCL handle, ``mhimx_comm_allreduce``) instead of
    torch.distributed - the same collective for hosts that do not run torch.distributed.
    """
    if world <= 1:
        return 1.0
    fill_tail(grad, student, n_train, chain)
    if comm is not None:
        comm.allreduce(grad)
    else:
        torch.distributed.all_reduce(grad, group=group)
    scale = 1.0 / world
    finish_tail(grad, student, n_train, scale, chain)
    return scale


def fill_tail(grad, student, n_train, chain=None):
    """Before the SUM: the tail of the flat buffer carries the non-trainable parameters (averaged back afterwards); with a QueryChain
    the global queries' slice carries this rank's weighted tokens instead."""
    grad[n_train:].copy_(student[n_train:])
    if chain is not None and chain.tokens is not None:
        torch.mul(chain.tokens.reshape(-1), chain.w, out=grad[chain.off:chain.off + chain.n])


def finish_tail(grad, student, n_train, scale, chain=None):
    if chain is not None and chain.tokens is not None:
        q0 = student[chain.off:chain.off + chain.n].clone()
        student[n_train:].copy_(grad[n_train:] * scale)
        torch.add(grad[chain.off:chain.off + chain.n], q0, alpha=chain.decay, out=student[chain.off:chain.off + chain.n])
        chain.tokens = None
    else:
        student[n_train:].copy_(grad[n_train:] * scale)
    grad[n_train:].zero_()


# Accumulation windows: both GEMMs of the window's bags as ONE launch each (mhimx_bag_project_multi / mhimx_bag_wgrad_multi).  MEASURED
# (round 3, 8 bags of the c2 shape, same box): the projection 8 x 61.2 -> 470 us, the weight gradient 8 x (44 + 3 of slab sum) -> 219 us -
# (tools/exp_wgrad_multi.py) and the window not at all (1.52 ms on 4 streams, 2.55 ms on one, with or without): the Merge backward's tail that rode in each bag's
# weight-gradient launch becomes a launch of its own (8 x 17.8 us), and the window on 4 streams is bound by how the graph's branches
# overlap (average concurrency 1.7, profiles/r03_window_timeline.md), not by kernel time.  Opt-in until the tail rides elsewhere.
_STEP_IMAGES = os.environ.get("MHIMX_STEP_IMAGES", "1") != "0"       # TransMIL: the step's weight images in the preparation launch
_FOREACH_GRADS = os.environ.get("MHIMX_FOREACH_GRADS", "1") != "0"   # autograd's per-parameter gradient adds as ONE multi-tensor launch
_WINDOW_BATCHED = os.environ.get("MHIMX_WINDOW_BATCHED", "1") != "0"      # (round 6) mhimx_window_run for windows of same-shaped bags
_WINDOW_WGRAD = os.environ.get("MHIMX_WINDOW_WGRAD", "0") != "0"
_WINDOW_PROJECT = os.environ.get("MHIMX_WINDOW_PROJECT", "0") != "0"


class _SplitStep:
    """A captured data-parallel step: graph(forward + backward + tail fill) -> eager all-reduce of the flat buffer -> graph(tail
    average + Adam + EMA)."""

    def __init__(self, trainer, g_fb, g_up):
        self.tr, self.g_fb, self.g_up = trainer, g_fb, g_up

    def replay(self):
        tr = self.tr
        self.g_fb.replay()                                     # ... + the buffer's tail filled
        tr._all_reduce(tr.flat.grad)
        self.g_up.replay()                                     # the tail averaged back + Adam + EMA
        ops.step_images(None)                                  # (nothing on the host installed any; a replay must not leave any either)


class FusedTrainer:
    """One-call MHIM(ABMIL) train step on flat buffers (the benchmarked path)."""

    def __init__(self, student: MHIM, teacher: Optional[MHIM], lr=2e-4, weight_decay=1e-5, betas=(0.9, 0.999), eps=1e-8,
                 mm=0.9997, main_alpha=1.0, aux_alpha=0.5, accumulation_steps=1, process_group=None, model="mhim", mm_sche=None,
                 clip_grad=None, lr_sche=None):
        if model not in ("mhim", "mhim_pure"):
            raise mh.L.MhimxError(f"FusedTrainer: model={model!r} (the path knows 'mhim' and 'mhim_pure')")
        if model == "mhim" and teacher is None:
            raise mh.L.MhimxError("FusedTrainer(model='mhim') needs the EMA teacher (modules/__init__.py:176-214 builds it)")
        if model == "mhim" and not student.merge_enable:
            raise mh.L.MhimxError("FusedTrainer(model='mhim') needs merge_enable=True: the reference's MHIM.forward rejects the Identity "
                                  "merge (mhim.py:82,351)")
        self.s, self.t = student, teacher
        self.flat = FlatState(student, teacher, step_merges=(model == "mhim"))
        self.lr, self.wd, self.betas, self.eps, self.mm = lr, weight_decay, betas, eps, mm
        self.main_alpha, self.aux_alpha = main_alpha, aux_alpha
        self.accum = max(1, int(accumulation_steps))
        self.pg = process_group
        self.comm = None                 # optional comm.NativeComm: the flat-gradient all-reduce through mhimx_comm_allreduce (C-ABI) instead
        self.world = torch.distributed.get_world_size(process_group) if self._dist() else 1
        self.model_kind = model
        # data parallel, one bag per rank per update, single-pass ABMIL step: the queries' EMA is chained over the ranks (QueryChain)
        self._chain, self._chain_tokens, self._q_scratch = None, None, None
        # Participation is a property of the MODEL (ABMIL baseline with Merge), never of a bag: every path an ABMIL step can take - the
        # single-pass step, the generic step of a small / odd-shaped bag, a bag with no rows to merge - leaves this rank's tokens for the
        # chain (forward_backward raises if one did not), so that no two ranks finish an update with different formulas
        if (self.world > 1 and self.accum == 1 and model == "mhim" and student.merge_enable and student.baseline == "attn"
                and "merge.global_q_mm" in self.flat.offsets):
            rank = torch.distributed.get_rank(process_group)
            self._chain = QueryChain(self.flat.offsets["merge.global_q_mm"], student.merge.global_q_mm.numel(), float(student.merge.g_q_mm),
                                     self.world, rank)
        self._micro = 0
        self.last = {}
        dev = self.flat.student.device
        # device-resident counters: hipGraph replays freeze kernel arguments, so the dropout stream position and the
        # Adam step count live in HBM and are advanced by one-thread kernels that are part of the captured step
        self.tick = torch.zeros(1, dtype=torch.int64, device=dev)
        self.opt_step = torch.zeros(1, dtype=torch.int64, device=dev)
        self._defer = ops.ReduceList()                 # final gradient reductions of a step, flushed as one launch
        # data parallel: the flat gradient is all-reduced in two pieces, [projection weight + bias | everything else + tail]; the
        # second piece is final before the backward's longest kernel and its all-reduce overlaps it (eager steps only)
        self.overlap_comm = True
        self._work_a = None
        self._capturing = False
        fo = self.flat.offsets
        self._split = fo["feature.0.bias"] + (student.feature[0].bias.numel() + 3) // 4 * 4 if "feature.0.bias" in fo else 0
        if not (self.flat.names[:2] == ["feature.0.weight", "feature.0.bias"]):
            self._split = 0                            # unexpected parameter order: one all-reduce of the whole buffer
        student._tick = self.tick
        if teacher is not None:
            teacher._tick = self.tick
        # optional EMA-momentum schedule (one value per optimiser step): a device table indexed by the device-resident step
        # counter, so it also advances under hipGraph replay
        self.mm_table = None if mm_sche is None else torch.as_tensor(mm_sche, dtype=torch.float32).to(dev).contiguous()
        # --clip_grad (base_engine.py:115-119: clip_grad_norm_ on the accumulated gradient right before optimizer.step) and a per-update
        # learning-rate schedule (train_utils.py:69-77, base_engine.py:152-153: scheduler.step() once per update): one value per update in a
        # device table read by the optimiser kernel at the device step counter - both keep working under hipGraph replay
        self.clip_grad = None if clip_grad is None else float(clip_grad)
        self.lr_table = None if lr_sche is None else torch.as_tensor(lr_sche, dtype=torch.float32).to(dev).contiguous()
        self._clip_ws = torch.empty(1024, device=dev) if self.clip_grad else None
        self._g_extra = None
        self._fold_now, self._fold_list = False, None      # (train_step: the backward's last reductions inside the update kernel)
        self.fold_reductions = os.environ.get("MHIMX_FOLD_REDUCTIONS", "1") != "0"
        self.ride_prep = os.environ.get("MHIMX_RIDE_PREP", "1") != "0"   # the student-side preparation jobs in the teacher's scorer launch
        self._graph_pool = None
        self._cap_stream = None
        self._side = None                  # the step executor's second stream (mhimx_step_cfg.side_stream: the step as a DAG), made on first use
        # (off by default: measured slower - every fork / join between two queues of a hipGraph costs 5-13 us, profiles/r06_dag.md)
        self.step_dag = os.environ.get("MHIMX_STEP_DAG", "0") != "0"
        # the whole step as ONE C call (mhimx_step_run, csrc/step.hip): the same launches as _forward_backward_nat + _apply, enqueued by the
        # library itself - a bag costs one ctypes call instead of ~1700 interpreter calls (the eager step was host-bound).  MHIMX_STEP_EXEC=0:
        # the Python orchestration (kept: it is the executor's specification and takes every case the executor refuses)
        self.use_executor = os.environ.get("MHIMX_STEP_EXEC", "1") != "0"
        self._exec = None
        self.exec_max_rows = 262144        # include/mhimx.h MHIMX_STEP_MAX_ROWS
        self.single_pass = True            # ABMIL: one projection launch for teacher + student, bag-ordered buffers (when shapes allow)
        self.window_streams = 4            # accumulation windows (window_step): HIP streams the window's bags are issued on
        # (round 6) a window of same-shaped bags as ONE C call with every launch over all its bags (mhimx_window_run, csrc/step.hip);
        # MHIMX_WINDOW_BATCHED=0: the bags on HIP streams, as rounds 3-5 had it
        self.window_batched = _WINDOW_BATCHED
        self._rows_cache = {}

    def _dist(self):
        return torch.distributed.is_available() and torch.distributed.is_initialized()

    # -------------------------------------------------------------------------------------------------
    def forward_backward(self, bag, label, perm=None, ids_shuffle=None, i=None):
        """Teacher fwd + select + student fwd + head + backward into the flat gradient buffer (accumulating)."""
        s, t, fl = self.s, self.t, self.flat
        x = s._check_x(bag)
        # parameter-only preparation of the step: weight transposes, paired-plane weights, query snapshot
        # (kept on the launch stream: a parallel branch in the captured hipGraph costs ~60 us of cross-queue signalling on
        # ROCm 7.2 — measured, profiles/ r01 notes — against ~40 us of kernels it would hide)
        # together with the two device counters (dropout stream position, Adam step) it is ONE launch
        if self._nat_ok(x, i):
            if self._exec_ok(x, i, perm, ids_shuffle):
                return self._exec_step(x, label, i)
            return self._forward_backward_nat(x, label, perm, ids_shuffle, i)
        prep_s = prep_t = None
        jobs = [(ops.PREP_TICK, None, self.tick)]
        if self._micro == 0:
            jobs.append((ops.PREP_TICK, None, self.opt_step))
        if s.baseline == "attn":
            if self.model_kind == "mhim":
                jt, prep_t = t.prep_jobs(backward=False)
                jobs += jt
            js, prep_s = s.prep_jobs(backward=True)
            jobs += js
        if s.baseline == "selfattn" and _STEP_IMAGES and x.shape[0] >= 2048:
            # TransMIL: the 12 weight images of the step (to_qkv / to_out of both layers: teacher and student as stored, the student's
            # also transposed for the data gradients) in this launch instead of 13 pair + 4 transpose launches along the step
            table = {}
            for model, backward in ((t if self.model_kind == "mhim" else None, False), (s, True)):
                if model is None:
                    continue
                for layer in (model.online_encoder.layer1, model.online_encoder.layer2):
                    for w in (layer.attn.to_qkv.weight.data, layer.attn.to_out[0].weight.data):
                        img = torch.empty_like(w)
                        jobs.append((ops.PREP_PAIR, w, img))
                        table[(w.data_ptr(), False)] = img
                        if backward:
                            img_t = torch.empty((w.shape[1], w.shape[0]), device=w.device)
                            jobs.append((ops.PREP_PAIR_T, w, img_t))
                            table[(w.data_ptr(), True)] = img_t
            ops.step_images(table)
        # one bf16 hi/lo image of the bag for both projections: it depends on nothing either, so it rides in the same launch
        xp = None
        if s.baseline == "attn" and s._pairable(x):
            xp = torch.empty_like(x)
            jobs.append((ops.PREP_PAIR, x, xp))
        ops.prep_batch(jobs)
        ps = x.shape[0]
        first = self._micro == 0
        gv = fl.grad_views
        if self.model_kind == "mhim":
            pre, seed_s = None, None
            if (self.single_pass and s.baseline in ("selfattn", "dsmil") and s.single_projection_ok(x) and t.single_projection_ok(x)
                    and s.act == t.act):
                # TransMIL / DSMIL: teacher AND student feature rows of all N bag rows in ONE pass over the raw bag (the reference's
                # student projects every row before it masks, mhim.py:335-336); the student's token rows are a gather, its projection
                # gradient the matrix-core-image pair on (d tokens, d out / d pre in fp16)
                tok_t = None
                if s.baseline == "selfattn":                 # the teacher's rows land under a free first row: [cls ; rows] without a copy
                    tok_t = torch.empty((1 + x.shape[0], t.feature[0].weight.shape[0]), device=x.device)
                heads = [ops.ProjHead(ops.pair_planes(t.feature[0].weight.data), t.feature[0].bias.data,
                                      drop_p=t.dropout_p if t.training else 0.0, drop_seed=t._next_seed(teacher=True),
                                      out=None if tok_t is None else tok_t[1:]),
                         None]
                seed_s = s._next_seed()
                heads[1] = ops.ProjHead(ops.pair_planes(s.feature[0].weight.data), s.feature[0].bias.data, drop_p=s.dropout_p,
                                        drop_seed=seed_s, want_dact=True)
                ops.bag_project(x, heads, act=mh.L.act_code(s.act, mh._FEATURE_ACTS), drop_tick=self.tick)
                pre = (heads[1].out, heads[1].dact)
                teacher_feat, score = t.forward_teacher(x, H=heads[0].out, tok_full=tok_t)
            else:
                teacher_feat, score = t.forward_teacher(x, xp=xp, w1p=None if prep_t is None else prep_t["w1p"],
                                                        wa_frag=None if prep_t is None else prep_t.get("wa_frag"))
            mf = s.baseline == "attn"                       # [merge | stay] rows: the pool reads [stay | merged tokens] contiguously
            rows, len_keep, Lk, R = s.student_rows(ps, i, score, perm=perm, ids_shuffle=ids_shuffle, merge_first=mf)
            plan = BagPlan(rows=rows, L=len_keep, Lk=Lk, R=R, drop_seed=s._next_seed() if seed_s is None else seed_s,
                           mca_seed=s._next_seed(), training=True, merge_first=mf)
            plan.pre = pre
            if self._chain is not None:                     # data parallel: the queries stay q0 until the update (QueryChain)
                if self._q_scratch is None:
                    self._q_scratch = torch.empty((s.merge.k, s.mlp_dim), device=x.device)
                plan.q_out = self._q_scratch
            keep_num = Lk + s.merge.k
        else:
            teacher_feat = None
            plan = s._plan_all_rows(ps)
            plan.training = True
            keep_num = ps
        if s.baseline == "selfattn":
            return self._selfattn_forward_backward(x, label, plan, teacher_feat, keep_num, first)
        if s.baseline == "dsmil":
            return self._dsmil_forward_backward(x, label, plan, teacher_feat, keep_num)
        merge_on = s.merge_enable
        if self.model_kind != "mhim":
            s.merge_enable = False
        try:
            z, saved = s._bag_forward(x, plan, xp=xp, prep=prep_s)
            if self._chain is not None and self.model_kind == "mhim":
                # this rank's term of the chain: the tokens its Merge produced; a bag with no rows to merge leaves the queries alone in the
                # reference (its EMA step is skipped) - handing the chain the current queries does the same up to second order in 1 - mm
                tok = saved.get("z_tok")
                self._chain.tokens = self._chain_tokens = tok if tok is not None else s.merge.global_q_mm.data.view(s.merge.k, -1).clone()
            t_in = teacher_feat.view(-1) if (teacher_feat is not None and self.aux_alpha != 0.) else None
            logits, losses, g_z, _, _ = ops.head_fwd_bwd(
                z, t_in, s.predictor.weight.data, s.predictor.bias.data, label, temp_t=float(s.temp_t),
                main_alpha=self.main_alpha, aux_alpha=self.aux_alpha, inv_accum=1.0 / self.accum,
                d_wp=gv["predictor.weight"], d_bp=gv["predictor.bias"], accumulate=not first)
            if first:
                # the six final gradient reductions of the backward (slab sums, column partials) run as ONE launch
                hook = self._mid_hook if (self.overlap_comm and self.comm is None and self.world > 1 and self.accum == 1 and not self._capturing
                                          and self._split > 0) else None
                s._bag_backward(x, plan, saved, g_z, out=gv, defer=self._defer, mid_hook=hook)
                ops.reduce_flush(self._defer)
            else:                                   # gradient accumulation: fresh buffers, then add (rare path)
                g = s._bag_backward(x, plan, saved, g_z)
                for n, v in g.items():
                    gv[n].add_(v)
        finally:
            s.merge_enable = merge_on
        self._micro += 1
        self.last = {"logits": logits, "losses": losses, "patch_num": ps, "keep_num": keep_num}
        return logits, losses

    def _nat_ok(self, x, i=None):
        """True when the bag takes the single-pass ABMIL step (_forward_backward_nat): no host read-back anywhere, fixed launch shapes."""
        s, t = self.s, self.t
        return bool(self.single_pass and s.baseline == "attn" and s.bag_ordered_ok(x) and (self.model_kind != "mhim" or (
            t is not None and t.bag_ordered_ok(x) and not t.merge_test and s.merge_enable and s.v2_counts(x.shape[0], i) is not None)))

    # ------------------------------------------------------------------------------------------------- the step behind the C-ABI
    def _exec_ok(self, x, i=None, perm=None, ids_shuffle=None, window=False):
        """True when mhimx_step_run takes this bag's step: the single-pass ABMIL step with device-drawn subsets, one process, one bag per
        update, no injected draws (csrc/step.hip: check_cfg).  ``window``: as a bag of mhimx_window_run (several bags per update)."""
        s, t = self.s, self.t
        if ops.KERNEL_EVENT_HOOK is not None:          # (a caller brackets single launches with events: only the Python orchestration can)
            return False
        # (round 6) a data-parallel rank takes it too - forward + backward with update = 0, the EMA of the global queries sent to the
        # QueryChain's scratch (mhimx_step_cfg.q_out), then the all-reduce and mhimx_optim_step as always - unless the eager step overlaps
        # its all-reduce with the backward (the mid-backward hook lives in the Python orchestration)
        hooked = (self.overlap_comm and self.comm is None and self.world > 1 and self.accum == 1 and not self._capturing and self._split > 0)
        if not (self.use_executor and self.model_kind == "mhim" and (self.accum == 1 or window) and not hooked and perm is None
                and ids_shuffle is None and self.ride_prep and s.training and s.n_classes <= 4 and s._op_prec != "f32"
                and s.merge.k * 8 <= 48 and x.shape[1] % 256 == 0 and x.stride(0) % 4 == 0 and x.shape[0] * x.stride(0) * 4 < (1 << 32)):
            return False
        # (csrc/step.hip:check_cfg and mhimx_step_run's own argument checks, mirrored: a bag they would refuse takes the Python path instead
        # of raising - ADVICE r5)
        mg = s.merge
        att = s.online_encoder.attention.attention
        if not (s.mlp_dim == 512 and att[0].weight.shape[0] == 128 and x.data_ptr() % 16 == 0 and x.stride(0) >= x.shape[1] and x.stride(1) == 1
                and mg.attn.to_q.weight.shape[0] == 512 and mg.attn.to_kv.weight.shape[0] == 1024 and 64 <= x.shape[0] <= self.exec_max_rows):
            return False
        c = s.v2_counts(x.shape[0], i)
        if c is None:
            return False
        # up to 16 384 rows: both random subsets drawn inside the one-workgroup select; above (round 6): the multi-workgroup select with the
        # draws as keyed permutations - the launches MHIM.student_rows issues for such bags
        if not (s.device_draw_ok(x.shape[0], i) if x.shape[0] <= 16384 else (s.baseline == "attn" and c[0] <= 16384)):
            return False
        return 1 <= c[0] and c[3] >= 1 and 1 <= c[4] <= 32768

    def _exec_cfg(self):
        """The mhimx_step_cfg of this trainer: parameter / gradient pointers into the flat buffers (stable for the trainer's lifetime), the
        model's hyper-parameters; the optimiser's scalars are refreshed on every call."""
        s, t, fl = self.s, self.t, self.flat
        L = mh.L
        ex = self._exec
        key = (s.feature[0].weight.data_ptr(), t.feature[0].weight.data_ptr(), fl.grad.data_ptr())
        if ex is None or ex["key"] != key:
            P = lambda tns: tns.data_ptr()
            def params(m, with_merge):
                att = m.online_encoder.attention
                pr = L.StepParams(w1=P(m.feature[0].weight), b1=P(m.feature[0].bias), wa=P(att.attention[0].weight), wc=P(att.attention[2].weight),
                                  wp=P(m.predictor.weight), bp=P(m.predictor.bias))
                if with_merge:
                    mg = m.merge
                    pr.q, pr.ln_w, pr.ln_b = P(mg.global_q_mm), P(mg.norm.weight), P(mg.norm.bias)
                    pr.wkv, pr.wq, pr.wo, pr.bo = P(mg.attn.to_kv.weight), P(mg.attn.to_q.weight), P(mg.attn.to_out[0].weight), P(mg.attn.to_out[0].bias)
                return pr
            gv = fl.grad_views
            pre = "online_encoder.attention.attention."
            grads = L.StepGrads(w1=P(gv["feature.0.weight"]), b1=P(gv["feature.0.bias"]), wa=P(gv[pre + "0.weight"]), wc=P(gv[pre + "2.weight"]),
                                wp=P(gv["predictor.weight"]), bp=P(gv["predictor.bias"]), ln_w=P(gv["merge.norm.weight"]), ln_b=P(gv["merge.norm.bias"]),
                                wkv=P(gv["merge.attn.to_kv.weight"]), wq=P(gv["merge.attn.to_q.weight"]), wo=P(gv["merge.attn.to_out.0.weight"]),
                                bo=P(gv["merge.attn.to_out.0.bias"]))
            cfg = L.StepCfg(D=s.input_dim, E=s.mlp_dim, A=s.online_encoder.attention.attention[0].weight.shape[0], C=s.n_classes, k=s.merge.k,
                            act=L.act_code(s.act, mh._FEATURE_ACTS), da_act=L.act_code(s.da_act, mh._SCORER_ACTS),
                            student=params(s, True), teacher=params(t, False), grad=grads,
                            p=P(fl.student), g=P(fl.grad), m=P(fl.m), v=P(fl.v), n_train=fl.n_train, n_all=fl.n_all,
                            tick=P(self.tick), opt_step=P(self.opt_step))
            ex = self._exec = {"key": key, "cfg": cfg, "layouts": {}, "ws": None}
        cfg = ex["cfg"]
        if self.step_dag:
            if self._side is None or self._side.device != fl.student.device:
                self._side = torch.cuda.Stream(device=fl.student.device)
            cfg.side_stream = self._side.cuda_stream
        else:
            cfg.side_stream = None
        if self._chain is not None:                    # data parallel: the queries stay q0 until the update (QueryChain)
            if self._q_scratch is None:
                self._q_scratch = torch.empty((s.merge.k, s.mlp_dim), device=fl.student.device)
            cfg.q_out = self._q_scratch.data_ptr()
        else:
            cfg.q_out = None
        cfg.time_project = int(bool(getattr(self, "time_project", False)))
        cfg.attn2score = int(bool(t.attn2score))
        cfg.drop_p_teacher = float(t.dropout_p if t.training else 0.0)
        cfg.drop_p_student = float(s.dropout_p)
        cfg.merge_drop_p, cfg.merge_mm = float(s.merge.dropout), float(s.merge.g_q_mm)
        cfg.temp_t, cfg.main_alpha, cfg.aux_alpha = float(s.temp_t), float(self.main_alpha), float(self.aux_alpha)
        cfg.p_teacher = None if fl.same_teacher else fl.teacher.data_ptr()
        cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay, cfg.ema_mm = self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.mm
        cfg.mm_table, cfg.mm_len = (None, 0) if self.mm_table is None else (self.mm_table.data_ptr(), self.mm_table.numel())
        cfg.lr_table, cfg.lr_len = (None, 0) if self.lr_table is None else (self.lr_table.data_ptr(), self.lr_table.numel())
        return ex

    def _exec_window_ok(self, xs, labels, i=None):
        """True when mhimx_window_run takes this window: 2..8 bags of ONE shape that each pass _exec_ok, up to 16 384 rows (the one-workgroup
        select), merge_k <= 6, one process, no mid-run ratio schedule (csrc/step.hip: check_window)."""
        from . import _lib as LB
        n = len(xs)
        if not (self.window_batched and 2 <= n <= LB.WINDOW_MAX and self.world == 1 and self._chain is None and not self.step_dag
                and self.flat.n_all % 4 == 0 and self.s.merge.k <= 6 and self.s.mrh_sche is None and self.flat.names[0] == "feature.0.weight"):
            return False
        x0 = xs[0]
        if not (x0.dim() == 2 and x0.shape[0] <= 16384 and all(x.shape == x0.shape and x.stride() == x0.stride() and x.device == x0.device for x in xs)):
            return False
        if not all(torch.is_tensor(l) and l.is_cuda and l.dtype == torch.int64 and l.numel() == 1 and l.device == x0.device for l in labels):
            return False
        return all(self._exec_ok(x, i, window=True) for x in xs)

    def _exec_window(self, xs, labels, i, update):
        """One accumulation window as ONE call of mhimx_window_run: every launch between the projections and the weight gradient covers all
        the bags (include/mhimx.h).  Returns ([logits per bag], [losses per bag]); self.last["bags"] holds each bag's views."""
        import ctypes as C
        L = mh.L
        s, t, fl = self.s, self.t, self.flat
        n = len(xs)
        ex = self._exec_cfg()
        N = xs[0].shape[0]
        k, n_sel, len_keep, Lk, R = s.v2_counts(N, i)
        key = ("window", n, N, k, n_sel, Lk)
        ent = ex["layouts"].get(key)
        if ent is None:
            cnt = L.StepCounts(k_top=k, n_sel=n_sel, len_keep=len_keep, Lk=Lk, R=R)
            lay = L.WindowLayout()
            L.check(L.lib().mhimx_window_layout_of(C.byref(ex["cfg"]), n, N, C.byref(cnt), C.byref(lay)), "mhimx_window_layout_of")
            ent = ex["layouts"][key] = (cnt, lay)
        cnt, lay = ent
        dev = xs[0].device
        if torch.cuda.is_current_stream_capturing():
            ws = torch.empty(lay.total, dtype=torch.uint8, device=dev)
        else:
            ws = ex.get("ws_win")
            if ws is None or ws.numel() < lay.total or ws.device != dev:
                # (poisoned once, when it is made: 0xFF bytes are NaNs - a launch that read workspace memory no launch of the window wrote
                # would show up in every eager test instead of depending on what the allocator's block held before)
                ws = ex["ws_win"] = torch.full((lay.total,), 255, dtype=torch.uint8, device=dev)
        # (the order the stream form with MHIMX_WINDOW_PROJECT=1 draws them in: every bag's dropout streams at the projection, then bag
        # after bag the select's and Merge's - the two forms of a window make the same draws)
        seeds = (L.StepSeeds * n)()
        for j in range(n):
            seeds[j].drop_teacher, seeds[j].drop_student = t._next_seed(teacher=True), s._next_seed()
        for j in range(n):
            seeds[j].select, seeds[j].mca = s._next_seed(), s._next_seed()
        Xp = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        Lp = (C.c_void_p * n)(*[l.data_ptr() for l in labels])
        inside = bool(update and not self.clip_grad)              # (clipping needs the norm of the final gradient: the update stays outside)
        L.check(L.lib().mhimx_window_run(ops._stream(), C.byref(ex["cfg"]), n, Xp, xs[0].stride(0), N, Lp, C.byref(cnt), seeds,
                                         fl.step + int(inside), ws.data_ptr(), ws.numel(), int(inside)), "mhimx_window_run")
        km, E = s.merge.k, s.mlp_dim
        bg = lay.bag

        def view(off, j, cnt_, dtype=torch.float32):
            o = off + j * lay.bag_stride
            return ws[o:o + cnt_ * dtype.itemsize].view(dtype)

        per, logits, losses = [], [], []
        for j in range(n):
            Hs = view(bg.H_student, j, (N + km) * E).view(N + km, E)
            lg, ls = view(bg.logits, j, s.n_classes), view(bg.losses, j, 3)
            rows_all = view(bg.rows_all, j, cnt.len_keep + km, torch.int64)
            per.append({"logits": lg, "losses": ls, "patch_num": N, "keep_num": cnt.Lk + km, "rows": rows_all[:cnt.len_keep],
                        "score": view(bg.score, j, N), "R": cnt.R, "tokens": Hs[N:], "H_student": Hs[:N],
                        "H_teacher": view(bg.H_teacher, j, N * E).view(N, E)})
            logits.append(lg); losses.append(ls)
        self.last = dict(per[-1], logits=logits, losses=losses, bags=per, ws=ws)
        self._micro = n
        if inside:
            fl.step += 1
            ops.step_images(None)
            self._micro = 0
        elif update:
            self.update()
        return logits, losses

    def _exec_plan(self, ex, N, i):
        """(counts, layout) of a bag of N rows at iteration i - cached: they follow from N and the HAM ratio alone."""
        import ctypes as C
        L = mh.L
        k, n_sel, len_keep, Lk, R = self.s.v2_counts(N, i)
        key = (N, k, n_sel, Lk)
        ent = ex["layouts"].get(key)
        if ent is None:
            cnt = L.StepCounts(k_top=k, n_sel=n_sel, len_keep=len_keep, Lk=Lk, R=R)
            lay = L.StepLayout()
            L.check(L.lib().mhimx_step_layout_of(C.byref(ex["cfg"]), N, C.byref(cnt), C.byref(lay)), "mhimx_step_layout_of")
            ent = ex["layouts"][key] = (cnt, lay)
        return ent

    def _exec_step(self, x, label, i):
        """forward_backward (+ the update, inside train_step) of one bag as ONE call of mhimx_step_run."""
        import ctypes as C
        L = mh.L
        s, t, fl = self.s, self.t, self.flat
        if not (torch.is_tensor(label) and label.is_cuda and label.dtype == torch.int64 and label.is_contiguous() and label.device == x.device):
            raise L.MhimxError(f"label: expected a contiguous int64 tensor on the bag's device ({x.device})")
        ex = self._exec_cfg()
        N = x.shape[0]
        cnt, lay = self._exec_plan(ex, N, i)
        # a captured step owns its workspace (the graph's memory pool); eager steps share one that only grows
        if torch.cuda.is_current_stream_capturing():
            ws = torch.empty(lay.total, dtype=torch.uint8, device=x.device)
        else:
            ws = ex["ws"]
            if ws is None or ws.numel() < lay.total or ws.device != x.device:
                ws = ex["ws"] = torch.full((int(lay.total * 1.25),), 255, dtype=torch.uint8, device=x.device)      # (poisoned: see _exec_window)
        seeds = L.StepSeeds(drop_teacher=t._next_seed(teacher=True), drop_student=s._next_seed(), select=s._next_seed(), mca=s._next_seed())
        update = bool(self._fold_now)
        L.check(L.lib().mhimx_step_run(ops._stream(), C.byref(ex["cfg"]), x.data_ptr(), x.stride(0), N, label.data_ptr(), C.byref(cnt), C.byref(seeds),
                                       fl.step + int(update), ws.data_ptr(), ws.numel(), int(update)), "mhimx_step_run")
        if update:                                      # (only a step that was enqueued counts: a refused call leaves host and device counters equal)
            fl.step += 1
        km, E = s.merge.k, s.mlp_dim

        def view(off, n, dtype=torch.float32):
            return ws[off:off + n * dtype.itemsize].view(dtype)

        Hs = view(lay.H_student, (N + km) * E).view(N + km, E)
        logits, losses = view(lay.logits, s.n_classes), view(lay.losses, 3)
        rows_all = view(lay.rows_all, cnt.len_keep + km, torch.int64)
        self.last = {"logits": logits, "losses": losses, "patch_num": N, "keep_num": cnt.Lk + km, "rows": rows_all[:cnt.len_keep],
                     "score": view(lay.score, N), "R": cnt.R, "tokens": Hs[N:], "H_student": Hs[:N],
                     "H_teacher": view(lay.H_teacher, N * E).view(N, E), "ws": ws}
        if self._chain is not None:                    # this rank's term of the chain: the tokens its Merge produced
            self._chain.tokens = self._chain_tokens = Hs[N:]
        if update:
            ops.step_images(None)
            self._micro = 0
        else:
            self._micro += 1
        return logits, losses

    def run_steps(self, bags, labels, i=None):
        """len(bags) consecutive complete train steps (one update each) as ONE call of mhimx_step_run_many (SURVEY 7 H4 "run_steps"): the
        bags of a resident dataset, bag after bag, on one workspace.  ``i``: the iteration of the FIRST bag (bag j runs at i + j, as the
        reference's loop counts them - base_engine.py:78) or a sequence with one entry per bag.  Returns the last bag's (logits, losses).
        What the executor's in-launch update does not do - gradient clipping (base_engine.py:115-119), unfolded reductions, accumulation -
        goes bag by bag through train_step, which does."""
        import ctypes as C
        L = mh.L
        n = len(bags)
        if n == 0 or len(labels) != n:
            raise L.MhimxError(f"run_steps: {n} bags, {len(labels)} labels")
        its = list(i) if isinstance(i, (list, tuple)) else [None if i is None else i + j for j in range(n)]
        if len(its) != n:
            raise L.MhimxError(f"run_steps: {len(its)} iteration indices for {n} bags")
        xs = [self.s._check_x(b) for b in bags]
        if self._micro != 0:
            raise L.MhimxError("run_steps: called inside an accumulation window (a fresh update is required)")
        if self.clip_grad or not self.fold_reductions or self.accum != 1 or self.world != 1 or not all(self._exec_ok(x, it) and self._nat_ok(x, it) for x, it in zip(xs, its)):
            out = None
            for b, l, it in zip(bags, labels, its):
                out = self.train_step(b, l, i=it)
            return out
        for x, l in zip(xs, labels):
            if not (torch.is_tensor(l) and l.is_cuda and l.dtype == torch.int64 and l.is_contiguous() and l.device == x.device and l.numel() >= 1):
                raise L.MhimxError(f"run_steps: every label must be a contiguous int64 tensor on its bag's device ({x.device})")
        ex = self._exec_cfg()
        plans = [self._exec_plan(ex, x.shape[0], it) for x, it in zip(xs, its)]
        total = max(p[1].total for p in plans)
        ws = ex["ws"]
        if ws is None or ws.numel() < total or ws.device != xs[0].device:
            ws = ex["ws"] = torch.empty(int(total * 1.25), dtype=torch.uint8, device=xs[0].device)
        Xp = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        ld = (C.c_int64 * n)(*[x.stride(0) for x in xs])
        Ns = (C.c_int64 * n)(*[x.shape[0] for x in xs])
        lab = (C.c_void_p * n)(*[l.data_ptr() for l in labels])
        cnts = (L.StepCounts * n)(*[p[0] for p in plans])
        seeds = (L.StepSeeds * n)()
        for j in range(n):
            seeds[j] = L.StepSeeds(drop_teacher=self.t._next_seed(teacher=True), drop_student=self.s._next_seed(), select=self.s._next_seed(),
                                   mca=self.s._next_seed())
        L.check(L.lib().mhimx_step_run_many(ops._stream(), C.byref(ex["cfg"]), n, Xp, ld, Ns, lab, cnts, seeds, self.flat.step + 1, ws.data_ptr(),
                                            ws.numel()), "mhimx_step_run_many")
        self.flat.step += n
        ops.step_images(None)
        lay = plans[-1][1]
        return ws[lay.logits:lay.logits + 4 * self.s.n_classes].view(torch.float32), ws[lay.losses:lay.losses + 12].view(torch.float32)

    def _forward_backward_nat(self, x, label, perm, ids_shuffle, i):
        """The single-pass ABMIL step: ONE projection launch computes the teacher's and the student's feature rows from the raw bag
        (the reference's student projects all N rows before it masks, mhim.py:335-336); the rows stay in bag order and the scorer,
        Merge, their backwards and the projection's weight-gradient GEMM gather the rows that take part by index."""
        first = self._micro == 0
        # caller-supplied tensors are validated HERE: inside the pinned step the wrappers skip their argument checks (ops._chk)
        for tns, nm in ((label, "label"), (perm, "perm"), (ids_shuffle, "ids_shuffle")):
            if torch.is_tensor(tns) and not (tns.is_cuda and tns.dtype == torch.int64 and tns.is_contiguous() and tns.device == x.device):
                raise mh.L.MhimxError(f"{nm}: expected a contiguous int64 tensor on the bag's device ({x.device}), got {tns.dtype} on {tns.device}")
        with ops.pinned_stream():                              # (one bag, one stream: the launches' stream is looked up once)
            res = self._nat_prep([x], i, with_opt_tick=first, split=self.ride_prep)
            prep_t, preps = res[0], res[1]
            preps[0]["_ride_jobs"] = res[2] if len(res) > 2 else None
            hook = self._mid_hook if (first and self.overlap_comm and self.comm is None and self.world > 1 and self.accum == 1
                                      and not self._capturing and self._split > 0) else None
            out = self._nat_bag(x, label, prep_t, preps[0], self.flat.grad_views, accumulate=not first, perm=perm, ids_shuffle=ids_shuffle,
                                i=i, mid_hook=hook)
        self._micro += 1
        return out

    def _nat_prep(self, xs, i, with_opt_tick=True, split=False):
        """The parameter-only preparation for the bags ``xs`` that share the current weights (one bag, or an accumulation window) as ONE
        launch: device counters, paired-plane / fragment / transposed weight images of both models, and - per bag, because it lives at the
        head of that bag's Merge workspace - the query side of the projection-free Merge.  Returns (teacher prep, [student prep per bag]).
        ``split`` (one bag, MHIM): only what the projection and the teacher's scorer read - the counters, both projection weight images,
        the teacher's scorer image - is launched here; everything else (the student's scorer images, the backward's transposes, the query
        snapshot, the Merge preparation's chain: ~2/3 of the launch's time) is returned as a third value and rides in the teacher's scorer
        launch (ops.abmil_pool_fwd(ride_jobs=)), off the head of the step's chain."""
        s, t = self.s, self.t
        mhim = self.model_kind == "mhim"
        dev = xs[0].device
        jobs = [(ops.PREP_TICK, None, self.tick)]
        if with_opt_tick:
            jobs.append((ops.PREP_TICK, None, self.opt_step))
        prep_t = None
        if mhim:
            jt, prep_t = t.prep_jobs(backward=False)
            jobs += jt
        merge_on = s.merge_enable
        if not mhim:
            s.merge_enable = False
        try:
            Rs = [s.v2_counts(x.shape[0], i)[4] if mhim else 0 for x in xs]
            lean = mhim and s.merge.k * 8 <= 48 and all(1 <= R <= 32768 for R in Rs) and s._op_prec != "f32"
            js, prep_s = s.prep_jobs(backward=True, lean_merge=lean)
            preps = []
            for R in Rs:
                pb = dict(prep_s)
                if lean:
                    # the parameter-only part of the student's Merge (LayerNorm of the queries, their projection, the score vectors) rides
                    # in the preparation launch: off the teacher -> select -> student chain
                    mw_prep = s._merge_w(None)
                    pb["merge_ws"] = mw_prep.ws_for(R, dev)
                    pb["_merge_prep_w"] = mw_prep                          # (keeps the weight struct alive until the launch is enqueued)
                    js.append((ops.PREP_MERGE, mw_prep, (pb["merge_ws"], R)))
                preps.append(pb)
            late = []
            if split and mhim and len(xs) == 1:
                w1_ptr = s.feature[0].weight.data_ptr()
                early = [j for j in js if j[0] == ops.PREP_PAIR and j[1].data_ptr() == w1_ptr]
                late = [j for j in js if not any(j is e for e in early)]
                js = early
            ops.prep_batch(jobs + js)
        finally:
            s.merge_enable = merge_on
        return (prep_t, preps, late) if split else (prep_t, preps)

    def _nat_heads(self, x, prep_t, prep_s):
        """The student's feature buffer [N + k, E] (the k rows behind the bag: Merge's tokens) and both models' projection heads of a bag."""
        s, t = self.s, self.t
        mhim = self.model_kind == "mhim"
        k = s.merge.k if mhim else 0
        Hbuf = torch.empty((x.shape[0] + k, s.mlp_dim), device=x.device)
        heads = []
        if mhim:
            p_t = t.dropout_p if t.training else 0.0                # the trainer keeps the teacher in train mode
            heads.append(ops.ProjHead(prep_t["w1p"], t.feature[0].bias.data, drop_p=p_t, drop_seed=t._next_seed(teacher=True)))
        heads.append(ops.ProjHead(prep_s["w1p"], s.feature[0].bias.data, drop_p=s.dropout_p, drop_seed=s._next_seed(), out=Hbuf, want_dact=True))
        return Hbuf, heads

    def _nat_bag(self, x, label, prep_t, prep_s, gv, accumulate=False, perm=None, ids_shuffle=None, i=None, mid_hook=None, q_out=None, slot=0,
                 wgrad_park=None, projected=None):
        """One bag of the single-pass step after its preparation: projection (teacher + student), teacher pool, select, Merge, student
        pool, head, backward.  ``gv``: the gradient views to fill (``accumulate``: add to them); ``q_out``: where Merge's EMA-updated
        queries go (default: the parameter itself, merge.py:142-143)."""
        s, t = self.s, self.t
        mhim = self.model_kind == "mhim"
        ps, E = x.shape[0], s.mlp_dim
        dev = x.device
        merge_on = s.merge_enable
        if not mhim:
            s.merge_enable = False
        try:
            k = s.merge.k if mhim else 0
            act = mh.L.act_code(s.act, mh._FEATURE_ACTS)
            if projected is None:
                Hbuf, heads = self._nat_heads(x, prep_t, prep_s)
                ops.bag_project(x, heads, act=act, drop_tick=self.tick)
            else:                                                 # an accumulation window projected all its bags in one launch
                Hbuf, heads = projected
            DACT = heads[-1].dact
            teacher_feat, rows_all, score = None, None, None
            if mhim:
                wp = t.predictor.weight.data if t.attn2score else None
                st_t = ops.abmil_pool_fwd(t._scorer(prep_t.get("wa_frag")), heads[0].out, None, wp=wp,
                                          bp=t.predictor.bias.data if t.attn2score else None, ride_jobs=prep_s.get("_ride_jobs"), no_backward=True)
                score = st_t.pscore if t.attn2score else ops.softmax_from_stats(st_t.s, st_t.stats)
                teacher_feat = st_t.z
                _, _, len_keep, Lk, R = s.v2_counts(ps, i)
                # [rows to merge | rows that stay | the k token rows ps .. ps+k-1]
                key = (ps, len_keep, k, slot)                         # (one list per bag of a window: the bags may run concurrently)
                rows_all = self._rows_cache.get(key)
                if rows_all is None:
                    rows_all = torch.empty(len_keep + k, dtype=torch.int64, device=dev)
                    rows_all[len_keep:] = torch.arange(ps, ps + k, device=dev)
                    self._rows_cache[key] = rows_all
                s.student_rows(ps, i, score.view(1, -1), perm=perm, ids_shuffle=ids_shuffle, merge_first=True, rows_out=rows_all)
                plan = BagPlan(rows=rows_all[:len_keep], L=len_keep, Lk=Lk, R=R, mca_seed=s._next_seed(), training=True, merge_first=True)
                if q_out is None and self._chain is not None:         # data parallel: the queries stay q0 until the update (QueryChain)
                    if self._q_scratch is None:
                        self._q_scratch = torch.empty((k, E), device=dev)
                    q_out = self._q_scratch
                    self._chain.tokens = self._chain_tokens = Hbuf[ps:]
                plan.q_out = q_out
                keep_num = Lk + k
            else:
                plan = BagPlan(rows=None, L=ps, Lk=ps, R=0, training=True)
                keep_num = ps
            z, saved = s._bag_forward_nat(x, plan, Hbuf, DACT, rows_all, prep_s)
            t_in = teacher_feat.view(-1) if (teacher_feat is not None and self.aux_alpha != 0.) else None
            logits, losses, g_z, _, _ = ops.head_fwd_bwd(
                z, t_in, s.predictor.weight.data, s.predictor.bias.data, label, temp_t=float(s.temp_t),
                main_alpha=self.main_alpha, aux_alpha=self.aux_alpha, inv_accum=1.0 / self.accum,
                d_wp=gv["predictor.weight"], d_bp=gv["predictor.bias"], accumulate=accumulate)
            # the six final gradient reductions of the backward (slab sums, column partials) run as ONE launch
            # train_step (one bag per update, one process, no clipping): nothing reads the gradient between this backward and the update,
            # so no reduction launch stands between them - the reductions that are final before the weight-gradient product starts ride in
            # its launch (ride_tail), and the product's own split-K slab sum is folded into the update kernel (_apply: fold)
            fold = self._fold_now and not accumulate and wgrad_park is None and mid_hook is None
            s._bag_backward_nat(x, plan, saved, g_z, gv, defer=self._defer, mid_hook=mid_hook, accumulate=accumulate, wgrad_park=wgrad_park,
                                ride_tail=fold)
            if fold:
                self._fold_list = self._defer
            else:
                ops.reduce_flush(self._defer)
        finally:
            s.merge_enable = merge_on
        # (kept for inspection / parity tests: under graph replay these are the static buffers the replay rewrites)
        self.last = {"logits": logits, "losses": losses, "patch_num": ps, "keep_num": keep_num, "rows": plan.rows,
                     "score": score, "R": plan.R, "tokens": Hbuf[ps:] if mhim else None,
                     "H_student": Hbuf[:ps], "H_teacher": heads[0].out if mhim else None}
        return logits, losses

    # ------------------------------------------------------------------------------------------------- accumulation windows
    def window_ok(self, bags, i=None):
        """The batched window takes the single-pass ABMIL step for every bag (the shapes _forward_backward_nat takes), one process."""
        s, t = self.s, self.t
        if self.model_kind != "mhim" or s.baseline != "attn" or not self.single_pass or s.mrh_sche is not None:
            return False
        for b in bags:
            x = b[0] if b.dim() == 3 else b
            if not (x.is_cuda and s.bag_ordered_ok(x) and t.bag_ordered_ok(x) and not t.merge_test and s.merge_enable
                    and s.v2_counts(x.shape[0], i) is not None and s.device_draw_ok(x.shape[0], i)
                    and ops.bag_wgrad_ok(x, s.mlp_dim, s.v2_counts(x.shape[0], i)[2])):
                return False
        if (_WINDOW_WGRAD or _WINDOW_PROJECT) and len({tuple((b[0] if b.dim() == 3 else b).shape) for b in bags}) > 1:
            return False                                   # the opt-in multi-bag launches size their workspace from bag 0
        return True

    def _window_state(self, n_streams, dev):
        """Side streams and their gradient slabs (stream 0 = the caller's stream accumulates straight into the flat gradient)."""
        st = getattr(self, "_win", None)
        if st is None or len(st["streams"]) < n_streams - 1:
            fl = self.flat
            slabs = torch.zeros((max(n_streams - 1, 1), fl.n_all), device=dev)
            pd = dict(self.s.named_parameters())
            views = [{n: slabs[j, fl.offsets[n]:fl.offsets[n] + pd[n].numel()].view_as(pd[n]) for n in fl.train_names}
                     for j in range(n_streams - 1)]
            st = self._win = {"streams": [torch.cuda.Stream() for _ in range(n_streams - 1)], "slabs": slabs, "views": views,
                              "defers": [ops.ReduceList() for _ in range(n_streams - 1)]}
        return st

    def window_step(self, bags, labels, i=None, n_streams=None, perms=None, shuffles=None, update=True):
        """ONE optimiser update over ``accumulation_steps`` bags that share the current weights (--accumulation_steps,
        base_engine.py:29,47-49,100-119,146-167), as a batch instead of a loop:

        * the parameter-only preparation (weight images, Merge's query side) runs ONCE for the window, Adam + EMA once (as in the loop);
        * the k bags are independent until the gradient sum (teacher and student weights are fixed inside a window), so they are issued
          round-robin on ``n_streams`` HIP streams: bag j's latency-bound launches (select, Merge, scorers, head: one to a few dozen
          workgroups each) run beside bag j'+1's projection and weight-gradient GEMMs instead of on one serial chain.  Stream 0 (the
          caller's) accumulates into the flat gradient, every other stream into its own slab; ONE reduce launch sums the slabs;
        * Merge's in-forward EMA of the global queries (merge.py:142-143): every bag attends with the window's first queries, and the
          window ends with the same chain of EMA steps on the tokens those forwards produced (oracle ``train_window(q_ema="window")``;
          differs from the reference's bag-after-bag order in second order of 1 - merge_mm).
        Each bag's loss is scaled by 1 / accumulation_steps in the head kernel (base_engine.py:102).  Capturable (``capture_window``).
        Returns ([logits per bag], [losses per bag])."""
        k = len(bags)
        assert k == self.accum and len(labels) == k, "window_step takes exactly accumulation_steps bags"
        xs = [self.s._check_x(b) for b in bags]
        if not self.window_ok(xs, i) or perms is not None or self.world > 1:
            outs = [self.train_step(b, l, i=i, **({} if perms is None else {"perm": perms[j], "ids_shuffle": shuffles[j]}))
                    for j, (b, l) in enumerate(zip(bags, labels))]
            return [o[0] for o in outs], [o[1] for o in outs]
        assert self._micro == 0, "window_step starts a fresh accumulation window"
        if self._exec_window_ok(xs, labels, i):
            return self._exec_window(xs, labels, i, update)
        S = max(1, min(int(n_streams or self.window_streams), k))
        dev = xs[0].device
        s, fl = self.s, self.flat
        prep_t, preps = self._nat_prep(xs, i, with_opt_tick=True)
        st = self._window_state(S, dev)
        km, E = s.merge.k, s.mlp_dim
        q_new = torch.empty((k, km, E), device=dev)                   # per-bag EMA outputs (unused: the chain below runs on the tokens)
        main = torch.cuda.current_stream()
        ev0 = torch.cuda.Event()
        ev0.record(main)
        logits, losses, tokens, per_bag = [None] * k, [None] * k, [None] * k, []
        park = [] if _WINDOW_WGRAD else None                          # the bags' dPre images: ONE weight-gradient product for the window
        proj = [None] * k
        if _WINDOW_PROJECT and k <= 8:
            # both models' projections of ALL the window's bags in one launch (mhimx_bag_project_multi): the weights are the window's, a bag's
            # last row tiles drain their 51 MB of outputs under the next bag's loops, and no projection fills every CU later, in the middle
            # of the other streams' small launches
            proj = [self._nat_heads(x, prep_t, pp) for x, pp in zip(xs, preps)]
            ops.bag_project_multi(xs, [h for _, h in proj], act=mh.L.act_code(s.act, mh._FEATURE_ACTS), drop_tick=self.tick)
            ev0.record(main)                                          # (the side streams start behind the projection)
        keep_defer = self._defer
        try:
            for j in range(k):
                lane = j % S
                stream = main if lane == 0 else st["streams"][lane - 1]
                gv = fl.grad_views if lane == 0 else st["views"][lane - 1]
                if lane and j < S:
                    stream.wait_event(ev0)
                self._defer = keep_defer if lane == 0 else st["defers"][lane - 1]
                with torch.cuda.stream(stream):
                    logits[j], losses[j] = self._nat_bag(xs[j], labels[j], prep_t, preps[j], gv, accumulate=(j >= S), i=i,
                                                         q_out=q_new[j], slot=j, wgrad_park=park, projected=proj[j])
                    tokens[j] = self.last["tokens"]
                    per_bag.append(self.last)
        finally:
            self._defer = keep_defer
        for lane in range(1, S):
            e = torch.cuda.Event()
            e.record(st["streams"][lane - 1])
            main.wait_event(e)
        if park:
            # dW1 (+)= sum_bags dPre_b^T X_b as ONE launch (mhimx_bag_wgrad_multi): every bag owns its slabs of k-steps, and what a launch pays
            # once - row tables, first tiles, 34 MB of split-K slabs written and summed - is paid once per window (8 x (36 + 9) -> 236 + 16 us)
            ops.bag_wgrad_multi(park, fl.grad_views["feature.0.weight"], accumulate=False, defer=self._defer)
            ops.reduce_flush(self._defer)
        if S > 1:                                                     # the other streams' slabs: summed into the gradient by the optimiser
            if update and self.world == 1:                            # kernel itself (mhimx_optim_args.g_extra); else by one reduce launch
                self._g_extra = st["slabs"][:S - 1]
            else:
                lst = ops.ReduceList()
                ops.reduce_slabs_job(lst, st["slabs"][:S - 1], fl.grad, accumulate=True)
                ops.reduce_flush(lst)
        # the window's EMA chain on the k token sets: q <- mm^k q + (1 - mm) sum_j mm^(k-1-j) z_j
        mm = float(s.merge.g_q_mm)
        w = self._win_w.get(k) if hasattr(self, "_win_w") else None
        if w is None:                                                 # (device arithmetic only: capture-safe)
            if not hasattr(self, "_win_w"):
                self._win_w = {}
            w = self._win_w[k] = ((1.0 - mm) * mm ** torch.arange(k - 1, -1, -1, device=dev, dtype=torch.float64)).float()
        Z = torch.stack(tokens)                                       # [k, km, E]
        q = s.merge.global_q_mm.data.view(km, E)
        q.mul_(mm ** k).add_((Z * w.view(k, 1, 1)).sum(0))
        self._micro = k
        if update:
            self.update()
        self.last = dict(self.last, logits=logits, losses=losses, bags=per_bag)
        return logits, losses

    def capture_window(self, bags, labels, warmup=1, n_streams=None, **kw):
        """Capture one whole accumulation window (window_step) on static (bags, labels) into ONE hipGraph whose branches are the
        window's streams; ``graph.replay()`` runs prep, the k bags, the slab sum, Adam + EMA."""
        if self.world > 1:
            raise mh.L.MhimxError("capture_window: one process (data-parallel ranks exchange gradients between windows: use capture())")
        if not self.window_ok([self.s._check_x(b) for b in bags], kw.get("i")):
            raise mh.L.MhimxError("capture_window: these bags / this model do not take the single-pass ABMIL step")
        self._capturing = True
        try:
            if self._cap_stream is None:
                self._cap_stream = torch.cuda.Stream()
            cs = self._cap_stream
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                for _ in range(warmup):
                    self.window_step(bags, labels, n_streams=n_streams, **kw)
            torch.cuda.current_stream().wait_stream(cs)
            torch.cuda.synchronize()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._graph_pool, stream=cs):
                self.window_step(bags, labels, n_streams=n_streams, **kw)
            return g
        finally:
            self._capturing = False

    def _bind_grads(self):
        """Every trainable parameter's .grad IS its view of the flat gradient buffer: autograd accumulates straight into the
        buffer the single all-reduce and the fused Adam read."""
        if not getattr(self, "_grads_bound", False):
            pd = dict(self.s.named_parameters())
            for n in self.flat.train_names:
                pd[n].grad = self.flat.grad_views[n]
            self._grads_bound = True

    def _dsmil_forward_backward(self, x, label, plan, teacher_feat, keep_num):
        """DSMIL student (scope row N1): kernel-backed autograd primitives (dsmil.py); loss, CE on the mixed logits and the
        per-class distillation with all three output gradients come from ONE head kernel."""
        from . import dsmil as DS
        from .mhim import _FeatureFn
        s = self.s
        self._bind_grads()
        if self.model_kind == "mhim":
            lb, li, B = s._dsmil_student(x, plan)
        else:
            lb, li, B, _ = s.online_encoder(_FeatureFn.apply(s, x, plan, s.feature[0].weight, s.feature[0].bias))
        Bt = teacher_feat[0].contiguous() if (teacher_feat is not None and self.aux_alpha != 0.) else None
        losses, g_lb, g_li, g_B = DS.dsmil_head(lb.detach(), li.detach(), label, B.detach().contiguous(), Bt, float(s.temp_t),
                                                self.main_alpha, self.aux_alpha, 1.0 / self.accum)
        self._backward_into_flat([lb, li, B], [g_lb, g_li, g_B])
        logits = 0.5 * (lb.detach() + li.detach())
        self._micro += 1
        self.last = {"logits": logits, "losses": losses, "patch_num": x.shape[0], "keep_num": keep_num}
        return logits, losses

    def _selfattn_forward_backward(self, x, label, plan, teacher_feat, keep_num, first):
        """TransMIL student: the encoder is a graph of kernel-backed autograd primitives (nystrom.py); every parameter's
        .grad IS its view of the flat gradient buffer, so autograd accumulates straight into the buffer the single
        all-reduce and the fused Adam read."""
        s, fl = self.s, self.flat
        gv = fl.grad_views
        self._bind_grads()
        if self.model_kind == "mhim":
            z = s._selfattn_student(x, plan)
        else:
            from .mhim import _FeatureFn
            z = s._encode(_FeatureFn.apply(s, x, plan, s.feature[0].weight, s.feature[0].bias))
        t_in = teacher_feat.view(-1) if (teacher_feat is not None and self.aux_alpha != 0.) else None
        # predictor + CE + distillation and their gradients in one kernel; accumulate: the buffer is zero after each update
        logits, losses, g_z, _, _ = ops.head_fwd_bwd(
            z.detach(), t_in, s.predictor.weight.data, s.predictor.bias.data, label, temp_t=float(s.temp_t),
            main_alpha=self.main_alpha, aux_alpha=self.aux_alpha, inv_accum=1.0 / self.accum,
            d_wp=gv["predictor.weight"], d_bp=gv["predictor.bias"], accumulate=True)
        self._backward_into_flat([z], [g_z])
        self._micro += 1
        self.last = {"logits": logits, "losses": losses, "patch_num": x.shape[0], "keep_num": keep_num}
        return logits, losses

    def _backward_into_flat(self, outs, g_outs):
        """torch.autograd.backward(outs, g_outs) with the parameter gradients landing in the flat buffer.  With .grad bound to the buffer's views autograd
        ADDS every parameter's gradient in a launch of its own (~30 launches of ~5 us, most of them for a few hundred floats, strung
        along the backward's chain); here .grad is empty during the backward - autograd just keeps the tensors our backward functions
        return - and ONE multi-tensor add moves them all into the buffer (which is zero after every update)."""
        if not _FOREACH_GRADS:
            torch.autograd.backward(outs, g_outs)
            return
        fl = self.flat
        pd = getattr(self, "_train_params", None)
        if pd is None:
            named = dict(self.s.named_parameters())
            pd = self._train_params = [(named[n], fl.grad_views[n]) for n in fl.train_names]
        for p, _ in pd:
            p.grad = None
        torch.autograd.backward(outs, g_outs)
        views, grads = [], []
        for p, v in pd:
            if p.grad is not None:
                views.append(v)
                grads.append(p.grad.reshape(v.shape))
            p.grad = v
        if grads:
            torch._foreach_add_(views, grads)

    def _mid_hook(self):
        """Data parallel, eager steps: everything but the projection's gradient (the head of the flat buffer) is final - start its
        all-reduce (with the global-query tail) on RCCL's stream while mul_colsum + the dW1 GEMM + its slab reduction still run."""
        fl = self.flat
        fill_tail(fl.grad, fl.student, fl.n_train, self._chain)
        self._work_a = torch.distributed.all_reduce(fl.grad[self._split:], group=self.pg, async_op=True)

    def update(self):
        """All-reduce (data parallel) + fused Adam + EMA teacher.  Call once per ``accumulation_steps`` bags."""
        fl = self.flat
        if self._chain is not None and self._chain.tokens is None:
            raise mh.L.MhimxError("data-parallel update: this rank's step left no Merge tokens for the query chain (QueryChain) - the ranks "
                                  "would finish the update with different formulas for merge.global_q_mm")
        if self._work_a is not None:                       # overlapped form: the rest of the buffer, then wait for both halves
            work_b = torch.distributed.all_reduce(fl.grad[:self._split], group=self.pg, async_op=True)
            self._work_a.wait()
            work_b.wait()
            self._work_a = None
            scale = 1.0 / self.world
            finish_tail(fl.grad, fl.student, fl.n_train, scale, self._chain)
        else:
            scale = sync_flat_gradient(fl.grad, fl.student, fl.n_train, self.world, self.pg, self.comm, self._chain)
        self._apply(scale)

    def _all_reduce(self, buf):
        """The ONE collective of an update: SUM over the ranks (RCCL through torch.distributed, or the C-ABI's own handle)."""
        if self.comm is not None:
            self.comm.allreduce(buf)
        else:
            torch.distributed.all_reduce(buf, group=self.pg)

    def _apply(self, scale):
        fl = self.flat
        # the weights change below: this step's prepared weight images (ops.step_images, keyed by storage address) are stale from here
        # on, whoever runs the update - update(), or the captured data-parallel step (graph | all-reduce | graph), which never passes
        # through update() (an eval forward between replays picked up images of the pre-update weights)
        ops.step_images(None)
        fl.step += 1                                   # (the device-side counter was advanced by the step's prep launch)
        ops.optim_step(fl.student, fl.grad, fl.m, fl.v, fl.teacher if (self.model_kind == "mhim" and not fl.same_teacher) else None, fl.n_train,
                       fl.step, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.wd,
                       grad_scale=scale, ema_mm=self.mm, zero_grad=True, step_dev=self.opt_step, mm_table=self.mm_table,
                       lr_table=self.lr_table, g_extra=self._g_extra, clip_norm=self.clip_grad, ws=self._clip_ws, fold=self._fold_list)
        self._g_extra = None
        self._fold_list = None
        self._micro = 0

    def capture(self, bag, label, warmup=2, **kw):
        """Capture one whole train step on (bag, label) into a hipGraph and return it; ``graph.replay()`` then costs one
        launch instead of ~80.  The step must already have run eagerly (lazy one-time setup such as
        hipFuncSetAttribute cannot happen under capture), hence the warm-up calls."""
        assert self.accum == 1, "graph capture covers a full step (accumulation_steps == 1)"
        if self.s.mrh_sche is not None:
            # a captured graph freezes host-side schedule values: k, n_sel and len_keep are launch arguments and buffer shapes
            raise mh.L.MhimxError("capture(): the HAM-ratio schedule (mrh_sche) changes the number of masked rows per iteration; "
                                  "run such a model with eager train_step calls, or capture one graph per schedule value")
        # (a learning-rate schedule given as lr_sche and the EMA momentum schedule are device tables indexed by the device step counter:
        # they advance under replay; the scalar lr is a by-value kernel argument)
        self._capturing = True                             # (collectives stay outside the graphs: no mid-backward all-reduce)
        try:
            return self._capture(bag, label, warmup, **kw)
        finally:
            self._capturing = False

    def _capture(self, bag, label, warmup, **kw):
        # warm up ON the capture stream: autograd's gradient-accumulation nodes (TransMIL student) remember the stream they
        # were created on, and a node living on another stream would need a cross-stream event inside the capture
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream()
        cs = self._cap_stream
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            for _ in range(warmup):
                self.train_step(bag, label, **kw)
        torch.cuda.current_stream().wait_stream(cs)
        torch.cuda.synchronize()
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()
        if self.world > 1:
            # data parallel: the gradient all-reduce stays OUTSIDE the graphs (compute | RCCL all-reduce | optimiser): a
            # collective inside a captured graph depends on the RCCL build, and a rank that replays while another launches
            # eagerly would deadlock - two graph launches and one eager collective per step cost ~20 us of host time
            # The tail traffic around the collective (the non-trainable parameters / this rank's weighted Merge tokens into the buffer's
            # tail before the SUM, their average back into the student after it) is captured with the neighbouring graph: a replayed step
            # is graph | one collective | graph, nothing else on the host.  capture_error_mode "thread_local": the process group's
            # watchdog thread polls its events while we capture (a "global" capture would be invalidated by it).
            fl = self.flat
            g_fb, g_up = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_fb, pool=self._graph_pool, stream=cs, capture_error_mode="thread_local"):
                self.forward_backward(bag, label, **kw)        # (leaves the QueryChain's tokens: static buffer of this capture)
                fill_tail(fl.grad, fl.student, fl.n_train, self._chain)
            ops.step_images(None)                              # (graph-static images of the capture-time weights: never hand them out)
            self._all_reduce(fl.grad)
            scale = 1.0 / self.world
            with torch.cuda.graph(g_up, pool=self._graph_pool, stream=cs, capture_error_mode="thread_local"):
                finish_tail(fl.grad, fl.student, fl.n_train, scale, self._chain)
                self._apply(scale)
            fl.grad.zero_()                                # (the capture-time all-reduce summed stale values)
            return _SplitStep(self, g_fb, g_up)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._graph_pool, stream=cs):
            self.train_step(bag, label, **kw)
        return g

    def capture_steps(self, bags, labels, warmup=1, **kw):
        """ONE hipGraph of len(bags) consecutive COMPLETE train steps (each with its own update), bag after bag: a trainer that walks a
        resident dataset pays one graph launch per chunk of bags instead of one per bag (~6-8 us of launch latency between two replays of
        18-kernel graphs at c2: 2 % of a step).  One process, accumulation_steps == 1; the same step sequence as calling train_step on
        the bags in order (the dropout / draw streams advance through the device counters: replay-safe)."""
        assert self.accum == 1 and self.world == 1, "capture_steps: one process, one bag per update"
        if self.s.mrh_sche is not None:
            raise mh.L.MhimxError("capture_steps(): the HAM-ratio schedule changes the launch shapes per iteration")
        self._capturing = True
        try:
            if self._cap_stream is None:
                self._cap_stream = torch.cuda.Stream()
            cs = self._cap_stream
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                for _ in range(warmup):
                    for b, l in zip(bags, labels):
                        self.train_step(b, l, **kw)
            torch.cuda.current_stream().wait_stream(cs)
            torch.cuda.synchronize()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._graph_pool, stream=cs):
                for b, l in zip(bags, labels):
                    self.train_step(b, l, **kw)
            return g
        finally:
            self._capturing = False

    def shape_cached(self, what, bag, label, i=None, cache=64):
        """``what`` = 'forward_backward' or 'train_step' on ``bag`` through a cache of captured hipGraphs keyed by the bag's SHAPE: the first
        bag of a shape runs eagerly (it is the kernels' warm-up), the second is captured - into buffers of the graph's own: the bag is copied
        into ONE buffer all the graphs share (they never run concurrently; so do they their memory pool), one launch - and replayed, every later
        one replays.  A dataset of bags of many sizes replays from its second epoch on.  The draw / dropout streams advance through the device
        step counter: replays draw fresh masks.  Only the single-pass ABMIL step is cached (``_nat_ok``: no host read-back, fixed launch
        shapes - a v1 mask ratio or a shape the one-pass kernels do not take runs eagerly), one process, accumulation_steps == 1.  The key
        holds everything a capture bakes in on the host: shape, the HAM-ratio schedule's row counts, the models' train / eval state, the
        Merge switch, the loss weights and - for 'train_step' - the optimiser's scalars.  ``cache``: the K most recently USED shapes stay.
        Returns (logits, losses, patch_num, keep_num) - the graph's static output buffers: read them before the next call - or None when
        the step cannot be cached (run it eagerly)."""
        x = bag[0] if bag.dim() == 3 else bag
        s, t = self.s, self.t
        if not (self.accum == 1 and self.world == 1 and x.is_cuda and x.dim() == 2 and not self._capturing and self._micro == 0):
            return None
        nat = self._nat_ok(x, i)
        # (round 5) the TransMIL / DSMIL students: their step runs autograd over kernel-backed nodes whose gradient-accumulation nodes belong to
        # the stream they were first run on, so EVERY step of such a shape runs on the trainer's capture stream - eager on the first two
        # visits (the second is the capture's warm-up), captured on the third.  v2 recipe only (a v1 ratio reads a count back).
        auto = (not nat) and s.baseline in ("selfattn", "dsmil") and (self.model_kind != "mhim" or s.v2_counts(x.shape[0], i) is not None)
        if not (nat or auto):
            return None
        # (a HAM-ratio schedule, --mrh_sche, changes the row counts - the launch shapes - every few hundred iterations: they are part of the key)
        counts = s.v2_counts(x.shape[0], i) if self.model_kind == "mhim" else None
        st = getattr(self, "_shape_graphs", None)
        if st is None:
            st = self._shape_graphs = {"graphs": {}, "seen": {}, "arena": None, "bad": set()}
        key = (what, tuple(x.shape), x.dtype, x.device.index, counts, s.training, None if t is None else t.training, s.merge_enable,
               float(self.main_alpha), float(self.aux_alpha), tuple(label.shape))
        # the optimiser's by-value scalars are baked into a captured update; scalars that come from a DEVICE table (lr_table / mm_table,
        # indexed by the device step counter) are not.  They are kept beside the graph, not in its key (with a host-side
        # scheduler every step had a new key, nothing was ever replayed and the visit counts grew without bound): a shape whose scalars
        # changed is captured again, and one whose scalars keep changing runs eagerly.
        scal = None
        if what == "train_step":
            scal = (None if self.lr_table is not None else float(self.lr), tuple(self.betas), float(self.eps), float(self.wd),
                    None if self.mm_table is not None else float(self.mm))
        if key in st["bad"]:
            return None
        ent = st["graphs"].get(key)
        if ent is not None and ent[8] != scal:
            st["graphs"].pop(key)
            ent = None
            n_re = st.setdefault("rescaled", {})
            n_re[key] = n_re.get(key, 0) + 1
            if n_re[key] > 2:
                st["bad"].add(key)
                st["seen"].pop(key, None)
                return None
            st["seen"][key] = 2                                  # (the shape is warm: capture on this visit)
        if len(st["seen"]) > 4096:                               # (visit counts of shapes that never came back)
            st["seen"] = {k_: v_ for k_, v_ in st["seen"].items() if k_ in st["graphs"]}
        fn = self.train_step if what == "train_step" else self.forward_backward
        if auto and self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream()
        if ent is None:
            st["seen"][key] = st["seen"].get(key, 0) + 1
            if st["seen"][key] < (3 if auto else 2):
                if auto:
                    cur = torch.cuda.current_stream()
                    self._cap_stream.wait_stream(cur)
                    with torch.cuda.stream(self._cap_stream):
                        logits, losses = fn(bag, label, i=i)
                    cur.wait_stream(self._cap_stream)
                else:
                    logits, losses = fn(bag, label, i=i)
                return logits, losses, self.last["patch_num"], self.last["keep_num"]
            while len(st["graphs"]) >= max(1, int(cache)):        # the least recently used shape makes room
                st["graphs"].pop(next(iter(st["graphs"])))
            need = x.numel()
            ar = st["arena"]
            if ar is None or ar.numel() < need or ar.dtype != x.dtype or ar.device != x.device:
                ar = st["arena"] = torch.empty(need, dtype=x.dtype, device=x.device)
            xs, ls = ar[:need].view(x.shape), torch.empty_like(label)
            if self._cap_stream is None:
                self._cap_stream = torch.cuda.Stream()
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            cs = self._cap_stream
            cs.wait_stream(torch.cuda.current_stream())
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            micro, host_step = self._micro, self.flat.step
            self._capturing = True
            try:
                with torch.cuda.graph(g, pool=self._graph_pool, stream=cs):      # (recorded, not run: the replay below is this bag's step)
                    logits, losses = fn(xs, ls, i=i)
            except Exception as exc:
                # a step that turned out not to be capturable (a host read-back somewhere): this shape runs eagerly from now on.  Argument
                # errors of the library (a label on the wrong device, a refused shape) are the caller's bug, not a capture problem: they
                # propagate; the first capture failure of every other kind is kept for inspection
                if isinstance(exc, mh.L.MhimxError):
                    self._capturing = False
                    self._micro, self.flat.step = micro, host_step
                    self._fold_list = None
                    self._defer = ops.ReduceList()
                    ops.step_images(None)
                    raise
                st["bad"].add(key)
                st.setdefault("errors", []).append(repr(exc))
                self._micro, self.flat.step = micro, host_step
                self._fold_list = None
                self._defer = ops.ReduceList()
                ops.step_images(None)
                torch.cuda.synchronize()
                return None
            finally:
                self._capturing = False
            ops.step_images(None)
            self._micro, self.flat.step = micro, host_step
            ent = st["graphs"][key] = (g, xs, ls, logits, losses, self.last["patch_num"], self.last["keep_num"], dict(self.last), scal)
        else:
            st["graphs"][key] = st["graphs"].pop(key)             # most recently used: to the end of the (insertion-ordered) dict
        g, xs, ls, logits, losses, pn, kn, last, _ = ent
        xs.copy_(x)
        ls.copy_(label)
        g.replay()
        if what == "train_step":
            self.flat.step += 1                                # (the host's count of updates: the device counter advanced in the graph)
            ops.step_images(None)
        else:
            self._micro += 1
        self.last = last
        return logits, losses, pn, kn

    def train_step(self, bag, label, **kw):
        # (a step that is followed by its update right here may leave its last reductions to the update kernel: _nat_bag)
        self._fold_now = self.fold_reductions and self.accum == 1 and self.world == 1 and not self.clip_grad
        try:
            out = self.forward_backward(bag, label, **kw)
        finally:
            self._fold_now = False
        if self._micro >= self.accum:
            self.update()
        return out
