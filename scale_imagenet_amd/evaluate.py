"""Evaluation loop: the caller of the hot path (counterpart of ``test()``, main.py:242-284).

Reproduces the reference's metric definitions -- cross-entropy loss, top-1 / top-5 running
means weighted by batch size (utils/bar_show.py:110-148), the final ``Acc..`` line
(main.py:284) -- around ``model(inputs)``.  The JPEG pipeline, TensorBoard and the terminal
progress bar of the reference are out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, Tuple

import torch
import torch.nn.functional as F


class RunningMean:
    """Batch-size weighted running mean (``AverageMeter``, utils/bar_show.py:127-148)."""

    def __init__(self, name: str = ""):
        self.name = name
        self.total = 0.0
        self.count = 0
        self.last = 0.0

    def update(self, value: float, n: int = 1):
        self.last = float(value)
        self.total += float(value) * n
        self.count += n

    @property
    def avg(self) -> float:
        return self.total / self.count if self.count else 0.0


def topk_percent(logits: torch.Tensor, targets: torch.Tensor, ks=(1, 5)):
    """Percent of rows whose target is among the k largest logits (utils/bar_show.py:110-124)."""
    order = logits.topk(max(ks), dim=1).indices
    hits = order.eq(targets.reshape(-1, 1))
    return [100.0 * hits[:, :k].any(dim=1).float().mean().item() for k in ks]


@dataclass
class EvalResult:
    loss: float
    top1: float
    top5: float
    images: int


class _Pending:
    """Device-side metrics of one batch, read back only when its lane is needed again."""

    def __init__(self, loss, hits1, hits5, n, event):
        self.loss, self.hits1, self.hits5, self.n, self.event = loss, hits1, hits5, n, event


@torch.no_grad()
def evaluate(model: torch.nn.Module, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]],
             device: torch.device, log_every: int = 0, inflight: int = 1) -> EvalResult:
    """main.py:242-284: ``model.eval()``, no_grad, per batch loss / top-1 / top-5.

    ``inflight`` > 1 keeps that many batches in flight on separate HIP streams and model lanes
    (``model.set_lanes``): the loss / top-k of a batch stay on the device until its lane comes
    round again, instead of the reference's ``.item()`` after every batch, so the ramp and tail
    of one batch's kernels overlap the next batch.  The metrics are the same numbers in the same
    order of accumulation."""
    model.eval()
    loss_m, top1_m, top5_m = RunningMean("Loss"), RunningMean("Acc@1"), RunningMean("Acc@5")
    use_lanes = inflight > 1 and device.type == "cuda" and hasattr(model, "set_lanes")
    if use_lanes:
        model.set_lanes(inflight)
        streams = [torch.cuda.Stream(device) for _ in range(inflight)]
    pending = []

    def retire(p: _Pending, i: int):
        if p.event is not None:
            p.event.synchronize()
        loss_m.update(p.loss.item(), p.n)
        top1_m.update(100.0 * p.hits1.item(), p.n)
        top5_m.update(100.0 * p.hits5.item(), p.n)
        if log_every and i % log_every == 0:
            print("Loss: %.3f | Acc1: %.3f%% Acc5: %.3f%% " % (loss_m.avg, top1_m.avg, top5_m.avg), flush=True)

    def metrics(outputs, targets):
        loss = F.cross_entropy(outputs, targets)
        order = outputs.topk(5, dim=1).indices
        hits = order.eq(targets.reshape(-1, 1))
        return loss, hits[:, :1].any(dim=1).float().mean(), hits[:, :5].any(dim=1).float().mean()

    done = 0
    for i, (inputs, targets) in enumerate(batches):
        if use_lanes:
            lane = i % inflight
            if len(pending) == inflight:              # this lane's previous batch: read its metrics now
                retire(pending.pop(0), done)
                done += 1
            with torch.cuda.stream(streams[lane]):
                inputs = inputs.to(device, non_blocking=True)
                targets = targets.to(device, non_blocking=True)
                outputs = model(inputs, lane=lane)
                loss, h1, h5 = metrics(outputs, targets)
                ev = torch.cuda.Event()
                ev.record(streams[lane])
            pending.append(_Pending(loss, h1, h5, inputs.size(0), ev))
        else:
            inputs = inputs.to(device, non_blocking=True)
            targets = targets.to(device, non_blocking=True)
            outputs = model(inputs)
            loss, h1, h5 = metrics(outputs, targets)
            retire(_Pending(loss, h1, h5, inputs.size(0), None), i)
    for p in pending:
        retire(p, done)
        done += 1
    # the range flag of the split operands is reported on the next call of a plan: without this, an overflow in the
    # last (or only) batch would end in silently invalid metrics
    inner = getattr(model, "module", model)              # (nn.DataParallel wrapper, main.py:192)
    if hasattr(inner, "check_range"):
        inner.check_range()
    print("Acc..", top1_m.avg, top5_m.avg)
    return EvalResult(loss_m.avg, top1_m.avg, top5_m.avg, loss_m.count)
