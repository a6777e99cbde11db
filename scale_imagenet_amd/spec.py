"""Architecture description of the TTNet ImageNet variants, as plain data.

Nothing here computes: it states, for each variant, the geometry of every
truth-table block (``Block_TT``) so that the nn.Module mirror (``ttnet.py``), the
C-ABI plan (``csrc/``), the synthetic weight generator (``synth.py``) and the
oracle all agree on shapes and state_dict keys.

Reference geometry (file:line under the upstream repository):
  * small   models/TT_general_imagenet_v2_small.py:24-76 (block ctor), :154-203 (net)
  * xsmall  models/TT_general_imagenet_v2_xsmall.py:24-76
  * full    models/TT_general_imagenet_v2.py:24-76 (fan-in 30, kernels (6,5)/(5,6))
  * Block_TT ctor  models/TT_FHE_SMALL.py:281-305
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

T_EXPAND = 8  # Block_TT's ``t``: mid channels = 8 * in_planes (TT_FHE_SMALL.py:281)


@dataclass(frozen=True)
class BlockTTSpec:
    """One ``Block_TT`` (TT_FHE_SMALL.py:278-320)."""
    name: str               # state_dict prefix, e.g. "features.4.Block_conv1"
    in_planes: int
    out_planes: int
    kh: int
    kw: int
    stride: int
    padding: int
    groups: int
    last: bool = False      # ends in ReLU (float) instead of the binary act

    @property
    def cin_g(self) -> int:
        return self.in_planes // self.groups

    @property
    def mid_g(self) -> int:
        return T_EXPAND * self.in_planes // self.groups

    @property
    def cout_g(self) -> int:
        return self.out_planes // self.groups

    @property
    def fan_in_bits(self) -> int:
        """n = number of input bits one output bit depends on."""
        return self.cin_g * self.kh * self.kw

    def out_hw(self, h: int, w: int) -> Tuple[int, int]:
        ho = (h + 2 * self.padding - self.kh) // self.stride + 1
        wo = (w + 2 * self.padding - self.kw) // self.stride + 1
        return ho, wo


@dataclass(frozen=True)
class MultiHeadSpec:
    """One 4-branch block (TT_general_imagenet_v2_small.py:21-148)."""
    name: str               # "features.4"
    in_planes: int
    out_planes: int         # the cfg entry; convf emits 2*out_planes (or 4*in if last)
    stride: int
    last: bool
    conv1: BlockTTSpec
    conv2: BlockTTSpec
    conv3: BlockTTSpec
    convf: BlockTTSpec
    in_hw: Tuple[int, int] = (0, 0)      # block input size
    out_hw: Tuple[int, int] = (0, 0)     # common size of the four branches after their padding


@dataclass(frozen=True)
class VariantSpec:
    variant: str
    p: int                                  # nfilter * tfilter
    image_hw: Tuple[int, int]
    blocks: Tuple[MultiHeadSpec, ...]
    fcsize: int
    inter: int
    n_classes: int
    feat_chw: Tuple[int, int, int]          # shape entering Flatten

    def block_tts(self) -> List[BlockTTSpec]:
        out = []
        for b in self.blocks:
            out += [b.conv1, b.conv2, b.conv3, b.convf]
        return out


# Per-variant constants: (dw kernel of conv1, dw kernel of conv2, dw padding, fan-in of the
# grouped 1x1 blocks).  small :28,:35,:41-42,:63 ; xsmall same lines ; full same lines.
_VARIANTS = {
    "small": dict(k1=(4, 4), k2=(4, 4), pad=2, gsize=16),
    "xsmall": dict(k1=(2, 2), k2=(2, 2), pad=1, gsize=4),
    "full": dict(k1=(6, 5), k2=(5, 6), pad=3, gsize=30),
}

# Shape-keyed zero padding of the four branches before the concat
# (TT_general_imagenet_v2_small.py:98-139).  Keyed by the block INPUT width; value is
# (pad of out1, pad of out2, pad of out3/out4), each (left, right, top, bottom).
# Only the rows reachable at 224x224 / --layers 0..4 are listed; an unknown width is an error
# here (the reference would fall through and fail in torch.cat).  Stride-1 blocks (--layers
# 3/4) use the same width-keyed rows: out3 = conv3(x) and out4 = x keep the input size (:95-96).
_Z = (0, 0, 0, 0)
PAD_TABLE_SMALL: Dict[int, Tuple[Tuple[int, int, int, int], ...]] = {
    56: (_Z, _Z, (1, 0, 1, 0)),     # :107-109  pad0 = ZeroPad2d((1,0,1,0))
    29: (_Z, _Z, (0, 1, 0, 1)),     # :110-114  pad2
    15: (_Z, _Z, (0, 1, 0, 1)),     # :123-125
    8: (_Z, _Z, (0, 1, 0, 1)),      # :126-128
    16: (_Z, _Z, (0, 1, 0, 1)),     # :120-122
    30: (_Z, _Z, (0, 1, 0, 1)),     # :137-139
    57: (_Z, _Z, (0, 1, 0, 1)),     # :115-119
    58: (_Z, _Z, (0, 1, 0, 1)),     # :134-136
}
PAD_TABLE_FULL: Dict[int, Tuple[Tuple[int, int, int, int], ...]] = {
    56: (_Z, _Z, (1, 0, 1, 0)),                              # v2.py:107-109
    29: ((0, 0, 0, 1), (0, 1, 0, 0), (0, 2, 0, 2)),          # v2.py:110-114
    16: (_Z, _Z, (0, 1, 0, 1)),                              # v2.py:115-117
}


def pad_table(variant: str):
    return PAD_TABLE_FULL if variant == "full" else PAD_TABLE_SMALL


def _cfg(p: int, layers: int, variant: str):
    # TT_general_imagenet_v2_small.py:172-181
    if layers == 0:
        return [(p, 2), (2 * p, 2)]
    if layers == 1:
        return [(p, 2), (2 * p, 2), (4 * p, 2)]
    if layers == 2:
        return [(p, 2), (2 * p, 2), (4 * p, 2), (8 * p, 2)]
    if layers == 3:
        return [p, (2 * p, 2), (4 * p, 2), (8 * p, 2)]
    if layers == 4 and variant != "full":
        return [p, (2 * p, 2), 2 * p, (4 * p, 2), (8 * p, 2)]
    raise ValueError(f"unsupported --layers {layers} for variant {variant}")


def make_spec(variant: str = "small", nfilter: int = 8, tfilter: int = 8, layers: int = 1,
              image_hw: Tuple[int, int] = (224, 224)) -> VariantSpec:
    if variant not in _VARIANTS:
        raise ValueError(f"unknown variant {variant!r}")
    v = _VARIANTS[variant]
    p = nfilter * tfilter
    cfg = _cfg(p, layers, variant)
    gs = v["gsize"]
    # stem: AvgPool2d(2) then Conv2d(3,p,7,2,3)  (:168-169)
    h = (image_hw[0] // 2 + 6 - 7) // 2 + 1
    w = (image_hw[1] // 2 + 6 - 7) // 2 + 1
    in_planes = p
    last_out = cfg[-1] if isinstance(cfg[-1], int) else cfg[-1][0]
    last = False
    blocks = []
    for i, x in enumerate(cfg):
        out_planes = x if isinstance(x, int) else x[0]
        stride = 1 if isinstance(x, int) else x[1]
        if out_planes == last_out:
            last = True                      # :192-193 (sticky)
        name = f"features.{4 + i}"
        if in_planes % gs or (4 * in_planes) % gs:
            # int(in/g) must divide in_planes or nn.Conv2d raises (full model at p=64)
            if in_planes % (in_planes // gs) or (4 * in_planes) % ((4 * in_planes) // gs):
                raise ValueError(
                    f"in_channels must be divisible by groups ({variant}, in_planes={in_planes})")
        g3 = in_planes // gs
        gf = 4 * in_planes // gs
        c1 = BlockTTSpec(f"{name}.Block_conv1", in_planes, in_planes, *v["k1"], stride, v["pad"], in_planes)
        c2 = BlockTTSpec(f"{name}.Block_conv2", in_planes, in_planes, *v["k2"], stride, v["pad"], in_planes)
        c3 = BlockTTSpec(f"{name}.Block_conv3", in_planes, in_planes, 1, 1, 1, 0, g3)
        cf_out = 4 * in_planes if last else 2 * out_planes
        cf = BlockTTSpec(f"{name}.Block_convf", 4 * in_planes, cf_out, 1, 1, 1, 0, gf, last=last)
        if stride == 1 and variant != "small":
            raise NotImplementedError("stride-1 multi-head blocks are built for the small variant only")
        tbl = pad_table(variant)
        if w not in tbl or h != w:
            raise ValueError(f"{name}: no branch-padding rule for input {h}x{w} "
                             f"(reference pad table is keyed by width, :98-139)")
        p1, p2, p34 = tbl[w]
        h1, w1 = c1.out_hw(h, w)
        h2, w2 = c2.out_hw(h, w)
        s1 = (h1 + p1[2] + p1[3], w1 + p1[0] + p1[1])
        s2 = (h2 + p2[2] + p2[3], w2 + p2[0] + p2[1])
        s34 = (h // stride + p34[2] + p34[3], w // stride + p34[0] + p34[1])      # stride 1: out3 / out4 keep the input size
        if not (s1 == s2 == s34):
            raise ValueError(f"{name}: branch shapes differ after padding: {s1} {s2} {s34}")
        blocks.append(MultiHeadSpec(name, in_planes, out_planes, stride, last, c1, c2, c3, cf, (h, w), s1))
        h, w = s1
        in_planes = 2 * out_planes
    c_last = blocks[-1].convf.out_planes
    fh, fw = h // 2, w // 2                 # AvgPool2d(2) :197
    fcsize = c_last * fh * fw
    return VariantSpec(variant, p, tuple(image_hw), tuple(blocks), fcsize, 1000, 1000,
                       (c_last, fh, fw))


def state_dict_layout(spec: VariantSpec) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """Keys, shapes and dtypes in the reference's registration order.

    Mirrors what ``state_dict()`` of the reference module returns (the
    committed fixture tests/golden/state_layout_small.json was captured from the
    imported reference and pins this function).
    """
    L: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()

    def bn(prefix: str, c: int):
        L[f"{prefix}.weight"] = ((c,), "float32")
        L[f"{prefix}.bias"] = ((c,), "float32")
        L[f"{prefix}.running_mean"] = ((c,), "float32")
        L[f"{prefix}.running_var"] = ((c,), "float32")
        L[f"{prefix}.num_batches_tracked"] = ((), "int64")

    def block_tt(b: BlockTTSpec):
        mid = T_EXPAND * b.in_planes
        L[f"{b.name}.conv1.weight"] = ((mid, b.cin_g, b.kh, b.kw), "float32")
        bn(f"{b.name}.bn1", mid)
        L[f"{b.name}.conv2.weight"] = ((b.out_planes, mid // b.groups, 1, 1), "float32")
        bn(f"{b.name}.bn2", b.out_planes)
        L[f"{b.name}.act.grad_scale"] = ((), "float32")

    L["features.1.weight"] = ((spec.p, 3, 7, 7), "float32")
    bn("features.2", spec.p)
    L["features.3.grad_scale"] = ((), "float32")
    for blk in spec.blocks:
        block_tt(blk.conv1)
        block_tt(blk.conv2)
        block_tt(blk.conv3)
        L[f"{blk.name}.act.grad_scale"] = ((), "float32")
        block_tt(blk.convf)
    head = f"features.{4 + len(spec.blocks) + 2}"
    L[f"{head}.lin1.weight"] = ((spec.inter, spec.fcsize), "float32")
    bn(f"{head}.BN2", spec.inter)
    L[f"{head}.lin2.weight"] = ((spec.n_classes, spec.inter), "float32")
    L[f"{head}.lin2.bias"] = ((spec.n_classes,), "float32")
    return L


# ---- CIFAR "vAlexnet" variant (models/TT_FHE_XSMALL_vAlexnet.py:434-676) ----------------------
@dataclass(frozen=True)
class VAlexSpec:
    """TT_FHE_XSMALL_vAlexnet: VGG16 features[0:2] stem (Conv2d(3,64,3,pad 1)+bias, ReLU; the
    reference fetches pretrained weights, :594-604), BatchNorm2d, MaxPool2d(3), thresholded act,
    ONE stride-1 4-branch block with (3,2)/(2,3) windows and 8-channel 1x1 groups, plain concat
    (no interleave, no convf, :575-583), Flatten, Classifier_scale(30976, 10, 100) (:663-675)."""
    variant: str = "valexnet"
    image_hw: Tuple[int, int] = (32, 32)
    p: int = 64
    pooled: int = 10                      # 32 // 3
    out_hw: Tuple[int, int] = (11, 11)    # branches after the W = 10 padding rule (:544-550)
    fcsize: int = 256 * 11 * 11
    inter: int = 100
    n_classes: int = 10
    conv1: BlockTTSpec = BlockTTSpec("features.5.Block_conv1", 64, 64, 3, 2, 1, 1, 64)
    conv2: BlockTTSpec = BlockTTSpec("features.5.Block_conv2", 64, 64, 2, 3, 1, 1, 64)
    conv3: BlockTTSpec = BlockTTSpec("features.5.Block_conv3", 64, 64, 1, 1, 1, 0, 8)
    # zero padding of (out1, out2, out3/out4) as (left, right, top, bottom), :544-550
    pads: Tuple[Tuple[int, int, int, int], ...] = ((0, 0, 0, 1), (0, 1, 0, 0), (0, 1, 0, 1))

    def block_tts(self) -> List[BlockTTSpec]:
        return [self.conv1, self.conv2, self.conv3]


def valexnet_layout(spec: "VAlexSpec" = None) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """The 57 state_dict entries in the reference's order; the stem conv is registered twice
    (as ``VGG_Model16_0`` and as ``features.0``), so its two tensors appear under both names."""
    spec = spec or VAlexSpec()
    L: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    for pre in ("VGG_Model16_0", "features.0"):
        L[f"{pre}.weight"] = ((64, 3, 3, 3), "float32")
        L[f"{pre}.bias"] = ((64,), "float32")

    def bn(prefix, c):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            L[f"{prefix}.{leaf}"] = ((c,), "float32")
        L[f"{prefix}.num_batches_tracked"] = ((), "int64")
    bn("features.2", 64)
    L["features.4.grad_scale"] = ((), "float32")
    for b in spec.block_tts():
        mid = T_EXPAND * b.in_planes
        L[f"{b.name}.conv1.weight"] = ((mid, b.cin_g, b.kh, b.kw), "float32")
        bn(f"{b.name}.bn1", mid)
        L[f"{b.name}.conv2.weight"] = ((b.out_planes, mid // b.groups, 1, 1), "float32")
        bn(f"{b.name}.bn2", b.out_planes)
        L[f"{b.name}.act.grad_scale"] = ((), "float32")
    L["features.7.lin1.weight"] = ((spec.inter, spec.fcsize), "float32")
    bn("features.7.BN2", spec.inter)
    L["features.7.lin2.weight"] = ((spec.n_classes, spec.inter), "float32")
    L["features.7.lin2.bias"] = ((spec.n_classes,), "float32")
    return L
