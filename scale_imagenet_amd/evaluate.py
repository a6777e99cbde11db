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


@torch.no_grad()
def evaluate(model: torch.nn.Module, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]],
             device: torch.device, log_every: int = 0) -> EvalResult:
    """main.py:242-284: ``model.eval()``, no_grad, per batch loss / top-1 / top-5."""
    model.eval()
    loss_m, top1_m, top5_m = RunningMean("Loss"), RunningMean("Acc@1"), RunningMean("Acc@5")
    for i, (inputs, targets) in enumerate(batches):
        inputs = inputs.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        outputs = model(inputs)
        loss = F.cross_entropy(outputs, targets)
        a1, a5 = topk_percent(outputs, targets, (1, 5))
        n = inputs.size(0)
        loss_m.update(loss.item(), n)
        top1_m.update(a1, n)
        top5_m.update(a5, n)
        if log_every and i % log_every == 0:
            print("Loss: %.3f | Acc1: %.3f%% Acc5: %.3f%% " % (loss_m.avg, top1_m.avg, top5_m.avg), flush=True)
    print("Acc..", top1_m.avg, top5_m.avg)
    return EvalResult(loss_m.avg, top1_m.avg, top5_m.avg, loss_m.count)
