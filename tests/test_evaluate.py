"""CPU: the evaluation loop (counterpart of test(), main.py:242-284) reproduces the reference's
metric definitions on the golden batch (logits, loss and top-5 captured from the reference)."""
import numpy as np
import torch

from _util import golden_npz, spec_and_state
from oracle import ttnet_float as OF
from scale_imagenet_amd import synth
from scale_imagenet_amd.evaluate import RunningMean, evaluate, topk_percent


class _OracleModel(torch.nn.Module):
    """Stands in for the device model on CPU (the product has no CPU path)."""

    def __init__(self, variant):
        super().__init__()
        self.spec, st = spec_and_state(variant)
        self.sd = OF.to_torch_state(st)

    def forward(self, x):
        return OF.forward(x, self.sd, self.spec)


def test_evaluate_matches_reference_metrics(capsys):
    g = golden_npz("xsmall")
    n = int(g["n_images"])
    x = torch.from_numpy(synth.synth_images(n))
    t = torch.from_numpy(synth.synth_targets(n))
    batches = [(x[:5], t[:5]), (x[5:], t[5:])]                   # ragged batches: means are size-weighted
    res = evaluate(_OracleModel("xsmall"), batches, torch.device("cpu"))
    assert res.images == n
    assert abs(res.loss - float(g["loss"])) < 1e-5
    hit1 = float((g["argmax"] == synth.synth_targets(n)).mean() * 100)
    hit5 = float((g["top5_idx"] == synth.synth_targets(n)[:, None]).any(1).mean() * 100)
    assert abs(res.top1 - hit1) < 1e-9 and abs(res.top5 - hit5) < 1e-9
    assert "Acc.." in capsys.readouterr().out                    # main.py:284


def test_topk_and_running_mean():
    logits = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.3]])
    assert topk_percent(logits, torch.tensor([1, 2]), (1, 2)) == [50.0, 100.0]
    m = RunningMean()
    m.update(1.0, 1)
    m.update(4.0, 3)
    assert m.avg == (1.0 + 12.0) / 4 and m.count == 4
