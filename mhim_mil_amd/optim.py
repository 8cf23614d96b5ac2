"""A fused optimiser for the REFERENCE trainer's own loop (engines/base_engine.py:76-167 unchanged).

The two-line swap of INTEGRATION.md section 2 keeps the reference's host loop: ``loss.backward()``, ``optimizer.step()``,
``optimizer.zero_grad()`` and the per-parameter EMA of the teacher (base_engine.py:155-167) - with ``torch.optim.Adam`` and the
Python EMA loop that is ~60 launches per update around ~20 kernels of model.  ``FusedAdamEMA`` is a ``torch.optim.Optimizer`` that
does all of it in ONE launch (``mhimx_optim_step``: Adam with the L2 term in the gradient - train_utils.py:58-65 - and the EMA of
every parameter, ``merge.global_q_mm`` included) over flat buffers:

    optimizer = FusedAdamEMA(model, model_ema, lr=args.lr, weight_decay=args.weight_decay, mm=args.mm,
                             mm_sche=model_others['mm_sche'], main_alpha=args.main_alpha, aux_alpha=args.aux_alpha,
                             accumulation_steps=args.accumulation_steps)      # instead of train_utils.py:58-65's torch.optim.Adam
    engine = CommonMIL(args, fused=optimizer)                                  # optional: the native forward + backward as well

* every parameter of both models becomes a view into one flat fp32 buffer per model (state_dict / load_state_dict keep working),
  every ``.grad`` a view of one flat gradient buffer: autograd accumulates straight into what the update kernel reads;
* ``step()`` is one launch; it also zeroes the gradient, so ``zero_grad()`` has nothing left to do (``.grad`` stays bound);
* the teacher's EMA happens inside ``step()``.  The loop's own EMA (``for param_q, param_k in zip(model.parameters(),
  model_ema.parameters())``, base_engine.py:166-167) must then find nothing to update: the adopted teacher's ``parameters()`` yields
  nothing (``MHIM.parameters`` honours ``_ema_owned``; ``named_parameters`` / ``state_dict`` / ``load_state_dict`` are untouched).
  The momentum of update t is ``mm_sche[t - 1]`` when a schedule is given (modules/__init__.py:177-181; the loop indexes it by
  ``epoch * len(loader) + batch_idx``, which is the same count when accumulation_steps == 1), else ``mm``;
* learning-rate schedulers work as with any optimiser: they write ``param_groups[0]['lr']``, ``step()`` reads it;
* ``--clip_grad`` (dispatch_clip_grad on the parameters' ``.grad``: views of the flat buffer) works unchanged; the GradScaler path
  (``--amp``) is not supported - the path computes in fp32-class arithmetic;
* ``CommonMIL(args, fused=optimizer)``: ``forward_func`` then runs the NATIVE teacher forward + select + student forward + head + backward
  (FusedTrainer.forward_backward: hand-derived backward, one projection launch for both models, no autograd graph) and hands the loop
  leaf tensors - the loop's ``criterion`` / ``loss.backward()`` run on two leaves and touch no parameter.  Taken only when the loop's
  loss IS ``main_alpha * CrossEntropyLoss(logits, label) + aux_alpha * aux_loss`` (base_engine.py:99-102: a plain
  ``nn.CrossEntropyLoss``); anything else falls back to the autograd path, which lands in the same flat gradient.
  ``CommonMIL(args, fused=optimizer, graph_cache=K)`` replays that native step as a captured hipGraph for the K most recent bag shapes
  (c2, same box: 1.75 ms for the plain swap -> 1.29 fused optimiser -> 0.69 native step eager -> 0.37 replayed).
"""
from __future__ import annotations

import torch

from . import ops
from .engine import FusedTrainer


def schedule_at_updates(sche, accumulation_steps=1, len_loader=None):
    """A per-BAG schedule (base_engine.py:161-162) at the bags that trigger an update (base_engine.py:47-49) - the per-UPDATE table the
    optimiser kernel indexes with its device step counter."""
    acc = max(1, int(accumulation_steps))
    if sche is None or (acc == 1 and not len_loader):
        return sche
    sche = list(sche)
    if not len_loader:
        return sche[acc - 1::acc]
    n_l = int(len_loader)
    idx = [e * n_l + b for e in range(-(-len(sche) // n_l)) for b in range(n_l) if (b + 1) % acc == 0 or b == n_l - 1]
    return [sche[j] for j in idx if j < len(sche)]


class FusedAdamEMA(torch.optim.Optimizer):
    def __init__(self, model, model_ema=None, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, mm=0.9997, mm_sche=None,
                 main_alpha=1.0, aux_alpha=0.5, accumulation_steps=1, model_kind=None, len_loader=None):
        if not any(p.is_cuda for p in model.parameters()):
            raise ValueError("FusedAdamEMA: move the model to the GPU first (the flat buffers live where the parameters are)")
        kind = model_kind or ("mhim" if model_ema is not None else "mhim_pure")
        # the loop indexes the momentum schedule by BAG (mm_sche[epoch * len(loader) + batch_idx], base_engine.py:161-162) and reads it at the
        # bag that triggers the update; the update kernel indexes its table by UPDATE.  The bags that trigger one (base_engine.py:47-49:
        # ``need_update = last_batch or (batch_idx + 1) % accumulation_steps == 0``, the count restarting every epoch) are every
        # accumulation_steps-th bag of an epoch AND its last one: with ``len_loader`` given the table holds the schedule at exactly those
        # bags - ceil(len / acc) per epoch (the plain [acc - 1::acc] slice ran ahead of the reference whenever len % acc != 0,
        # and the update kernel then clamped to mm ~ 1 early).  Without ``len_loader`` the slice stays (exact when acc divides len).
        mm_sche = schedule_at_updates(mm_sche, accumulation_steps, len_loader)
        self.trainer = FusedTrainer(model, model_ema, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps, mm=mm, main_alpha=main_alpha,
                                    aux_alpha=aux_alpha, accumulation_steps=accumulation_steps, model=kind, mm_sche=mm_sche)
        self.flat = self.trainer.flat
        named = dict(model.named_parameters())
        params = [named[n] for n in self.flat.train_names]
        for n in self.flat.train_names:
            named[n].grad = self.flat.grad_views[n]            # autograd accumulates into the flat buffer
        self.trainer._grads_bound = True
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model, self.model_ema = model, model_ema
        self._views = [(named[n], self.flat.grad_views[n]) for n in self.flat.train_names]
        # --tea_type same (modules/__init__.py:211-212): model_ema IS model - the reference runs no EMA then (base_engine.py:157-158) and the
        # student must keep its parameters() (the optimiser's, clip_grad's): nothing is adopted twice, nothing is hidden
        if model_ema is not None and model_ema is not model:
            model_ema._ema_owned = True                          # base_engine.py:166-167 finds no parameter left to update

    def close(self):
        """Give the teacher its ``parameters()`` back (the fused EMA ends here): call when the optimiser is dropped."""
        if self.model_ema is not None and getattr(self.model_ema, "_ema_owned", False):
            self.model_ema._ema_owned = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None if closure is None else closure()
        for p, v in self._views:                                # (a loop that set .grad to None or swapped it: fold it back)
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.add_(p.grad.reshape(v.shape))
                p.grad = v
        g = self.param_groups[0]
        tr = self.trainer
        tr.lr, tr.betas, tr.eps, tr.wd = float(g["lr"]), tuple(g["betas"]), g["eps"], g["weight_decay"]
        if tr._micro == 0:                                      # the gradient came through autograd: the native forward (which advances the
            ops.tick(tr.opt_step)                               # device-resident Adam step counter in its preparation launch) did not run
        tr.update()                                             # ONE launch: Adam + EMA, the gradient zeroed (+ the data-parallel sum when ranks > 1)
        return loss

    def zero_grad(self, set_to_none: bool = False):
        """Nothing to launch: ``step()`` zeroed the flat gradient.  Between updates of an accumulation window the gradient must
        survive, and the reference only calls this after an update (base_engine.py:151)."""
        if self.flat.step == 0:
            self.flat.grad.zero_()
