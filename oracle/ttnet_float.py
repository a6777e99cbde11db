"""ORACLE (test infrastructure, not product code) -- float-mode CPU restatement.

A PyTorch-CPU restatement of the reference's eval-mode ``forward()`` for the TTNet
ImageNet variants, written from the reference's op sequence, one function per
reference function.  It is the checker for the HIP path and the timed CPU baseline
(``cpu_baseline.kind = "port"``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product package never does.

Pinning: ``oracle/gen_golden.py`` (run once in the build container, where the
reference is importable) checked that this file reproduces the imported reference's
logits and every intermediate stage bit for bit on the synthetic state_dict, and wrote
the reference's own outputs to ``tests/golden/``; ``tests/test_oracle_golden.py``
re-checks this file against those fixtures on every run.

It keeps the reference's incidental work on purpose (the inert ``randint_like`` of the
thresholded act, the two ``clone()`` per Block_TT, the ``contiguous()`` of the
interleave) so that its timing is the reference's op sequence, not a tidied one.

All citations are file:line under the upstream repository root.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from scale_imagenet_amd.spec import BlockTTSpec, MultiHeadSpec, VariantSpec, pad_table

BN_EPS = 1e-5  # nn.BatchNorm2d / BatchNorm1d default


def binarize01_stem(x: torch.Tensor) -> torch.Tensor:
    """models/model_utils/netbin.py:193 -- ``(inp >= 0).to(inp.dtype)``."""
    return (x >= 0).to(x.dtype)


def binarize01_thresholded(x: torch.Tensor, T: float = 0.0) -> torch.Tensor:
    """models/TT_FHE_SMALL.py:186-192.  At T = 0 the ``maks`` term is identically zero
    but ``randint_like`` still runs; kept so the baseline pays for it."""
    all_ones = 1.0 * (x >= T / 2)
    maks = 1.0 * (x < T / 2) - 1.0 * (x < -T / 2)
    random = torch.randint_like(x, 2).to(x.dtype)
    res = all_ones + maks * random
    return res.to(x.dtype)


def _bn(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    return F.batch_norm(x, sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"],
                        sd[f"{prefix}.weight"], sd[f"{prefix}.bias"], False, 0.1, BN_EPS)


def block_tt(x: torch.Tensor, sd: Dict[str, torch.Tensor], b: BlockTTSpec,
             pre_act: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """models/TT_FHE_SMALL.py:307-320 (``Block_TT.forward``)."""
    _input_layer = x.clone()                                                    # :310
    if b.padding != 0:
        x = F.pad(x, (b.padding,) * 4, mode="constant", value=0.0)             # :311-312
    out = F.conv2d(x, sd[f"{b.name}.conv1.weight"], None, b.stride, 0, 1, b.groups)
    out = F.gelu(_bn(out, sd, f"{b.name}.bn1"))                                 # :313
    out = _bn(F.conv2d(out, sd[f"{b.name}.conv2.weight"], None, 1, 0, 1, b.groups), sd, f"{b.name}.bn2")
    if pre_act is not None:
        pre_act.append(out)
    if b.last:
        out = F.relu(out)                                                       # :315
    else:
        out = binarize01_thresholded(out, 0.0)                                  # :318
    _output_layer = out.clone()                                                 # :319
    return out


def multihead_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], blk: MultiHeadSpec, variant: str,
                    taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """models/TT_general_imagenet_v2_small.py:78-148."""
    out3 = block_tt(x, sd, blk.conv3)                                           # :88
    out2 = block_tt(x, sd, blk.conv2)                                           # :89
    out1 = block_tt(x, sd, blk.conv1)                                           # :90
    if blk.stride == 2:
        out4 = binarize01_thresholded(F.avg_pool2d(x, 2) - 0.5)                 # :93
        out3 = binarize01_thresholded(F.avg_pool2d(out3, 2) - 0.5)              # :94
    else:
        out4 = x                                                                # :96
    w = x.shape[-1]
    tbl = pad_table(variant)
    if w not in tbl:
        raise RuntimeError(f"no branch padding rule for width {w}")
    p1, p2, p34 = tbl[w]                                                        # :98-139
    if any(p1):
        out1 = F.pad(out1, p1)
    if any(p2):
        out2 = F.pad(out2, p2)
    out3 = F.pad(out3, p34)
    out4 = F.pad(out4, p34)
    if taps is not None:
        taps[f"{blk.name}.out1"], taps[f"{blk.name}.out2"] = out1, out2
        taps[f"{blk.name}.out3"], taps[f"{blk.name}.out4"] = out3, out4
    outf = torch.cat((out1, out2, out3, out4), dim=1)                           # :142
    n, c, hh, ww = outf.shape
    outf = outf.view(n, 4, c // 4, hh, ww).transpose(1, 2).contiguous().view(n, c, hh, ww)  # :144-147
    return block_tt(outf, sd, blk.convf)                                        # :148


def classifier_scale(x: torch.Tensor, sd: Dict[str, torch.Tensor], head: str) -> torch.Tensor:
    """models/TT_general_imagenet_v2_small.py:229-236 + Polynome_ACT :213-215."""
    x = F.linear(x, sd[f"{head}.lin1.weight"])
    x = F.batch_norm(x, sd[f"{head}.BN2.running_mean"], sd[f"{head}.BN2.running_var"],
                     sd[f"{head}.BN2.weight"], sd[f"{head}.BN2.bias"], False, 0.1, BN_EPS)
    x = 0.47 + 0.50 * x + 0.09 * x ** 2
    return F.linear(x, sd[f"{head}.lin2.weight"], sd[f"{head}.lin2.bias"])


def stem(x: torch.Tensor, sd: Dict[str, torch.Tensor], pre: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """features[0:4]: AvgPool2d(2), Conv2d(3,p,7,2,3), BatchNorm2d, Binarize01Act
    (models/TT_general_imagenet_v2_small.py:168-169, :183-184; netbin.py:193)."""
    x = F.avg_pool2d(x, 2)
    x = F.conv2d(x, sd["features.1.weight"], None, 2, 3)
    x = _bn(x, sd, "features.2")
    if pre is not None:
        pre.append(x)
    return binarize01_stem(x)


@torch.no_grad()
def forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], spec: VariantSpec,
            taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """``SeqBinModelHelper.forward`` (netbin.py:703-708) = ``self.features(x)``."""
    pre: List[torch.Tensor] = []
    x = stem(x, sd, pre)
    if taps is not None:
        taps["stem.pre"] = pre[0]
        taps["features.3"] = x
    for blk in spec.blocks:
        x = multihead_block(x, sd, blk, spec.variant, taps)
        if taps is not None:
            taps[blk.name] = x
    x = F.avg_pool2d(x, 2)                                                      # :197
    x = x.view(x.size(0), -1)                                                   # utils.py:266-267
    if taps is not None:
        taps["flatten"] = x
    head = f"features.{4 + len(spec.blocks) + 2}"
    return classifier_scale(x, sd, head)


@torch.no_grad()
def forward_valexnet(x: torch.Tensor, sd: Dict[str, torch.Tensor], spec,
                     taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """TT_FHE_XSMALL_vAlexnet eval forward (models/TT_FHE_XSMALL_vAlexnet.py): features =
    [VGG conv3x3+bias, ReLU, BatchNorm2d, MaxPool2d(3), thresholded act (:606-614), the stride-1
    block (:490-583), Flatten, Classifier_scale without polynomial (:663-675)]."""
    h = F.relu(F.conv2d(x, sd["features.0.weight"], sd["features.0.bias"], 1, 1))
    h = _bn(h, sd, "features.2")
    h = F.max_pool2d(h, 3)
    if taps is not None:
        taps["stem.pre"] = h
    h = binarize01_thresholded(h, 0.0)
    if taps is not None:
        taps["features.4"] = h
    out3 = block_tt(h, sd, spec.conv3)                                          # :497
    out2 = block_tt(h, sd, spec.conv2)                                          # :498
    out1 = block_tt(h, sd, spec.conv1)                                          # :499
    out4 = h                                                                    # stride 1: :505
    assert h.shape[-1] == 10, "only the 32x32 geometry (W = 10 rule, :544-550) is restated"
    p1, p2, p34 = spec.pads
    out1, out2, out3, out4 = F.pad(out1, p1), F.pad(out2, p2), F.pad(out3, p34), F.pad(out4, p34)
    outf = torch.cat((out1, out2, out3, out4), dim=1)                           # :575 (no interleave, no convf)
    if taps is not None:
        taps["features.5"] = outf
    z = outf.reshape(outf.size(0), -1)
    z = F.linear(z, sd["features.7.lin1.weight"])
    z = F.batch_norm(z, sd["features.7.BN2.running_mean"], sd["features.7.BN2.running_var"],
                     sd["features.7.BN2.weight"], sd["features.7.BN2.bias"], False, 0.1, BN_EPS)
    return F.linear(z, sd["features.7.lin2.weight"], sd["features.7.lin2.bias"])


def to_torch_state(np_state) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(v.copy()) for k, v in np_state.items()}


# ---- caller-side metrics (main.py:242-284, utils/bar_show.py:110-148) -------------------

def topk_accuracy(logits: torch.Tensor, target: torch.Tensor, ks=(1, 5)):
    """Percent of rows whose target is among the k largest logits
    (same definition as utils/bar_show.py:110-124)."""
    top = logits.topk(max(ks), dim=1, largest=True, sorted=True).indices
    hit = top.eq(target.view(-1, 1))
    return [hit[:, :k].any(dim=1).float().sum() * (100.0 / target.numel()) for k in ks]
