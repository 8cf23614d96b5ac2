"""Bag feeder for the MHIM step.

The reference hands every bag to the model from host memory: ``FeatClsDataset.__getitem__`` loads one ``pt_files/<slide>.pt``
feature matrix per step (datasets/dataset_feat.py:63-111, ``persistence`` keeps them in host RAM) and the DataLoader batch is
copied to the GPU inside the train loop (engines/base_engine.py:52-60).  At 0.68 ms per step a 41 MB bag cannot cross PCIe in
time (0.73 ms from pinned memory), so the MI355X-first feeder has two modes:

* ``resident=True``  — every bag lives in HBM (288 GB holds ~7 000 bags of 10 000 x 1024 fp32): iteration is pointer hand-over,
  and a captured hipGraph per bag (``FusedTrainer.capture``) can be replayed directly.
* ``resident=False`` — bags stay in pinned host memory; the NEXT bag's H2D copy runs on a copy stream into the other half of a
  double buffer while the current step computes (events in both directions: compute waits for its bag, the copy waits until
  the buffer's previous step has finished).

Sources are tensors / arrays already in memory or paths of ``torch.save``d feature matrices ([N, D] float).

``BagLoader`` is the same feeder behind the reference's LOADER seam: it yields the batch dictionaries the reference's train /
validate loops unpack (datasets/dataset_feat.py:93-111 through a batch_size-1 DataLoader and datasets/data_utils.PrefetchLoader,
:484-521): ``{'input': bag [1, N, D] on the device, 'target': label [1] int64 on the device[, 'idx': [name]]}``, with ``len()``.
``BaseTrainer.train`` (engines/base_engine.py:46-60, ``args.prefetch`` set: the batch is already on the device) consumes it as is.
"""
from __future__ import annotations

from typing import Iterable, Sequence

import torch


def _load(src):
    if isinstance(src, (str, bytes)) or hasattr(src, "__fspath__"):
        try:
            t = torch.load(src, weights_only=True)              # dataset_feat.py:86-89
        except Exception:                                          # noqa: BLE001 — the reference retries without weights_only
            t = torch.load(src, weights_only=False)
    else:
        t = src
    t = torch.as_tensor(t)
    if t.dim() == 3 and t.shape[0] == 1:
        t = t[0]
    if t.dim() != 2:
        raise ValueError(f"a bag is a [N, D] feature matrix, got shape {tuple(t.shape)}")
    return t.float().contiguous()


class BagFeeder:
    """Iterates (bag [N,D] on the device, label [1] int64 on the device, index) over ``order`` (default: all bags in turn)."""

    def __init__(self, bags: Sequence, labels: Sequence[int], device="cuda", resident=True, order: Iterable[int] | None = None):
        if len(bags) != len(labels):
            raise ValueError("one label per bag")
        self.device = torch.device(device)
        self.resident = bool(resident)
        self.labels = [torch.tensor([int(l)], device=self.device) for l in labels]
        self.order = list(range(len(bags))) if order is None else list(order)
        host = [_load(b) for b in bags]
        if self.resident:
            self.dev = [h.to(self.device) for h in host]           # one copy, then the bags never move again
            self.host = None
        else:
            self.host = [h.pin_memory() for h in host]
            rows, width = max(h.shape[0] for h in host), host[0].shape[1]
            if any(h.shape[1] != width for h in host):
                raise ValueError("all bags must share the feature width D")
            self._buf = [torch.empty((rows, width), device=self.device) for _ in range(2)]
            self._copy = torch.cuda.Stream(device=self.device)
            self._landed = [torch.cuda.Event() for _ in range(2)]
            self._released = [torch.cuda.Event() for _ in range(2)]

    def __len__(self):
        return len(self.order)

    def bag(self, i):
        """Resident mode: the device tensor of bag i (stable address: safe to capture in a hipGraph)."""
        if not self.resident:
            raise RuntimeError("bag(i) needs resident=True")
        return self.dev[i]

    def _start_copy(self, slot, idx):
        with torch.cuda.stream(self._copy):
            self._copy.wait_event(self._released[slot])            # the step that last used this half has finished
            n = self.host[idx].shape[0]
            self._buf[slot][:n].copy_(self.host[idx], non_blocking=True)
            self._landed[slot].record(self._copy)

    def __iter__(self):
        if self.resident:
            for idx in self.order:
                yield self.dev[idx], self.labels[idx], idx
            return
        main = torch.cuda.current_stream(self.device)
        for slot in range(2):
            self._released[slot].record(main)
        if self.order:
            self._start_copy(0, self.order[0])
        for pos, idx in enumerate(self.order):
            slot = pos % 2
            if pos + 1 < len(self.order):
                self._start_copy(1 - slot, self.order[pos + 1])    # overlaps this step's compute
            main.wait_event(self._landed[slot])
            yield self._buf[slot][:self.host[idx].shape[0]], self.labels[idx], idx
            self._released[slot].record(main)                      # everything the consumer enqueued on `main` is ordered before


class BagLoader:
    """The reference's loader contract over a BagFeeder (PrefetchLoader stand-in: batches arrive on the device)."""

    def __init__(self, bags: Sequence, labels: Sequence[int], device="cuda", resident=True, order: Iterable[int] | None = None,
                 names: Sequence[str] | None = None, return_id=False):
        self.feeder = BagFeeder(bags, labels, device=device, resident=resident, order=order)
        self.names = list(names) if names is not None else [str(i) for i in range(len(bags))]
        self.return_id = bool(return_id)
        self.device = self.feeder.device

    def __len__(self):
        return len(self.feeder)

    def __iter__(self):
        for bag, label, idx in self.feeder:
            batch = {"input": bag.unsqueeze(0), "target": label}            # the default collate of a batch_size-1 DataLoader
            if self.return_id:
                batch["idx"] = [self.names[idx]]                             # dataset_feat.py:108-109
            yield batch
