"""ORACLE (test infrastructure, not product code) -- bit-mode CPU restatement.

numpy restatement of the truth-table view of the reference's gate path: every
binarised ``Block_TT`` (models/TT_FHE_SMALL.py:307-320) is a lookup table
``{0,1}^n -> {0,1}^(C_out/G)`` per group, enumerated exactly the way the reference's own
exporter enumerates it (``get_TT_block_all_filter``, models/TT_FHE_SMALL.py:322-342):
pattern ``x`` is the n-bit binary expansion of the table index, MSB first, laid out as
``[c_in_group, kh, kw]``; zero padding is bit 0.

The table entries are evaluated in float64 with an exact erf from the float32
parameters; an entry whose float64 pre-activation is within ``NEAR_TIE`` of zero is a
*near tie*: the reference's own float32 arithmetic (oneDNN conv, oneDNN/ATen gelu) may
round such an entry to either side, so it is excluded from bit-exact claims and
reported.  Everything downstream of the tables is integer work and is
bit exact.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
from scipy.special import erf as _erf

from scale_imagenet_amd.spec import BlockTTSpec, MultiHeadSpec, VariantSpec, pad_table

BN_EPS = 1e-5
NEAR_TIE = 1e-5          # |pre-activation| below this: sign is implementation-defined


def fold_bn(sd: Dict[str, np.ndarray], prefix: str) -> Tuple[np.ndarray, np.ndarray]:
    """Eval-mode BatchNorm as y = x*scale + shift, float64."""
    var = sd[f"{prefix}.running_var"].astype(np.float64)
    scale = sd[f"{prefix}.weight"].astype(np.float64) / np.sqrt(var + BN_EPS)
    shift = sd[f"{prefix}.bias"].astype(np.float64) - sd[f"{prefix}.running_mean"].astype(np.float64) * scale
    return scale, shift


def gelu64(x: np.ndarray) -> np.ndarray:
    return 0.5 * x * (1.0 + _erf(x * 0.7071067811865476))


def enumerate_patterns(n: int) -> np.ndarray:
    """[2^n, n] uint8, row i = binary expansion of i, MSB first (TT_FHE_SMALL.py:330)."""
    idx = np.arange(1 << n, dtype=np.uint32)
    return ((idx[:, None] >> np.arange(n - 1, -1, -1, dtype=np.uint32)[None, :]) & 1).astype(np.uint8)


def block_pre_activation(sd: Dict[str, np.ndarray], b: BlockTTSpec, groups: Optional[List[int]] = None
                         ) -> np.ndarray:
    """float64 pre-activation bn2(conv2(gelu(bn1(conv1(pattern))))) for every pattern.
    Returns [len(groups), 2^n, cout_g]."""
    n = b.fan_in_bits
    if n > 20:
        raise ValueError(f"{b.name}: fan-in {n} is too wide to enumerate")
    pats = enumerate_patterns(n).astype(np.float64)                  # [2^n, n]
    w1 = sd[f"{b.name}.conv1.weight"].astype(np.float64).reshape(b.groups, b.mid_g, n)
    w2 = sd[f"{b.name}.conv2.weight"].astype(np.float64).reshape(b.groups, b.cout_g, b.mid_g)
    s1, t1 = fold_bn(sd, f"{b.name}.bn1")
    s2, t2 = fold_bn(sd, f"{b.name}.bn2")
    s1, t1 = s1.reshape(b.groups, b.mid_g), t1.reshape(b.groups, b.mid_g)
    s2, t2 = s2.reshape(b.groups, b.cout_g), t2.reshape(b.groups, b.cout_g)
    gl = list(range(b.groups)) if groups is None else groups
    out = np.empty((len(gl), 1 << n, b.cout_g), dtype=np.float64)
    for i, g in enumerate(gl):
        mid = pats @ w1[g].T                                          # [2^n, mid_g]
        mid = gelu64(mid * s1[g] + t1[g])
        out[i] = (mid @ w2[g].T) * s2[g] + t2[g]
    return out


def build_lut(sd: Dict[str, np.ndarray], b: BlockTTSpec, groups: Optional[List[int]] = None):
    """(bits uint8 [G, 2^n, cout_g], near_tie bool same shape) in the canonical index order.
    For a ``last`` block returns (relu(pre) float32, near_tie)."""
    pre = block_pre_activation(sd, b, groups)
    near = np.abs(pre) < NEAR_TIE
    if b.last:
        return np.maximum(pre, 0.0).astype(np.float32), near
    return (pre >= 0).astype(np.uint8), near


# ---- evaluation on bit tensors [N,C,H,W] uint8 --------------------------------------------

def window_index(x: np.ndarray, b: BlockTTSpec) -> np.ndarray:
    """Canonical table index for every output position.  x: [N,C,H,W] uint8.
    Returns uint32 [N, G, Ho, Wo]."""
    n_, c, h, w = x.shape
    p = b.padding
    if p:
        x = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))
    ho, wo = b.out_hw(h, w)
    cg = b.cin_g
    idx = np.zeros((n_, b.groups, ho, wo), dtype=np.uint32)
    xg = x.reshape(n_, b.groups, cg, x.shape[2], x.shape[3])
    nbits = b.fan_in_bits
    j = 0
    for ci in range(cg):
        for kh in range(b.kh):
            for kw in range(b.kw):
                sl = xg[:, :, ci, kh:kh + b.stride * (ho - 1) + 1:b.stride, kw:kw + b.stride * (wo - 1) + 1:b.stride]
                idx |= sl.astype(np.uint32) << np.uint32(nbits - 1 - j)
                j += 1
    return idx


def apply_lut(x: np.ndarray, table: np.ndarray, b: BlockTTSpec) -> np.ndarray:
    """x [N,C,H,W] bits -> [N,Cout,Ho,Wo] (uint8 bits, or float32 for a last block)."""
    idx = window_index(x, b)                                         # [N,G,Ho,Wo]
    n_, g, ho, wo = idx.shape
    out = np.empty((n_, g, b.cout_g, ho, wo), dtype=table.dtype)
    for gi in range(g):
        out[:, gi] = np.moveaxis(table[gi][idx[:, gi]], -1, 1)       # [N,Ho,Wo,cout_g] -> [N,cout_g,Ho,Wo]
    return out.reshape(n_, g * b.cout_g, ho, wo)


def apply_direct(x: np.ndarray, sd: Dict[str, np.ndarray], b: BlockTTSpec,
                 near: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
    """Block_TT.forward (models/TT_FHE_SMALL.py:307-320) on bit inputs, evaluated directly in
    float64 with an exact erf -- for blocks too wide to enumerate (fan-in 30 of the full
    model) and as a cross-check of the tables.  Returns bits (uint8) or relu'd float32 (last).
    ``near[b.name]`` receives the near-tie mask |pre| < NEAR_TIE."""
    import torch
    import torch.nn.functional as F
    xt = torch.from_numpy(x.astype(np.float64))
    if b.padding:
        xt = F.pad(xt, (b.padding,) * 4)
    w1 = torch.from_numpy(sd[f"{b.name}.conv1.weight"].astype(np.float64))
    w2 = torch.from_numpy(sd[f"{b.name}.conv2.weight"].astype(np.float64))
    s1, t1 = fold_bn(sd, f"{b.name}.bn1")
    s2, t2 = fold_bn(sd, f"{b.name}.bn2")
    h = F.conv2d(xt, w1, None, b.stride, 0, 1, b.groups).numpy()
    h = gelu64(h * s1[None, :, None, None] + t1[None, :, None, None])
    pre = F.conv2d(torch.from_numpy(h), w2, None, 1, 0, 1, b.groups).numpy()
    pre = pre * s2[None, :, None, None] + t2[None, :, None, None]
    if near is not None:
        near[b.name] = np.abs(pre) < NEAR_TIE
    if b.last:
        return np.maximum(pre, 0.0).astype(np.float32)
    return (pre >= 0).astype(np.uint8)


def majority2x2(x: np.ndarray) -> np.ndarray:
    """act(AvgPool2d(2)(x) - 0.5): floor-cropped 2x2 windows, 1 iff at least 2 of 4 set
    (models/TT_general_imagenet_v2_small.py:93-94)."""
    h2, w2 = x.shape[2] // 2, x.shape[3] // 2
    x = x[:, :, :2 * h2, :2 * w2].astype(np.uint8)
    s = x[:, :, 0::2, 0::2] + x[:, :, 0::2, 1::2] + x[:, :, 1::2, 0::2] + x[:, :, 1::2, 1::2]
    return (s >= 2).astype(np.uint8)


def _zpad(x: np.ndarray, p) -> np.ndarray:
    l, r, t, bm = p
    return np.pad(x, ((0, 0), (0, 0), (t, bm), (l, r)))


def multihead_block_bits(x: np.ndarray, luts: Dict[str, np.ndarray], blk: MultiHeadSpec, variant: str,
                         taps: Optional[Dict[str, np.ndarray]] = None, sd=None, near=None) -> np.ndarray:
    """models/TT_general_imagenet_v2_small.py:78-148 on bits.  A Block_TT without an entry in
    ``luts`` is evaluated directly (float64) from ``sd``."""
    def run(inp, b):
        if luts is not None and b.name in luts:
            return apply_lut(inp, luts[b.name], b)
        return apply_direct(inp, sd, b, near)
    out3 = run(x, blk.conv3)
    out2 = run(x, blk.conv2)
    out1 = run(x, blk.conv1)
    out4 = majority2x2(x) if blk.stride == 2 else x              # :91-96
    out3 = majority2x2(out3) if blk.stride == 2 else out3
    p1, p2, p34 = pad_table(variant)[x.shape[-1]]
    out1, out2, out3, out4 = _zpad(out1, p1), _zpad(out2, p2), _zpad(out3, p34), _zpad(out4, p34)
    if taps is not None:
        for nm, t in (("out1", out1), ("out2", out2), ("out3", out3), ("out4", out4)):
            taps[f"{blk.name}.{nm}"] = t
    n_, c, hh, ww = out1.shape
    outf = np.stack((out1, out2, out3, out4), axis=2).reshape(n_, 4 * c, hh, ww)   # channel 4c+branch
    return run(outf, blk.convf)


def build_all_luts(sd: Dict[str, np.ndarray], spec: VariantSpec):
    luts, near = {}, {}
    for b in spec.block_tts():
        luts[b.name], near[b.name] = build_lut(sd, b)
    return luts, near


def head64(feat: np.ndarray, sd: Dict[str, np.ndarray], head: str) -> np.ndarray:
    """Classifier_scale in float64 (TT_general_imagenet_v2_small.py:229-236)."""
    z = feat.astype(np.float64) @ sd[f"{head}.lin1.weight"].astype(np.float64).T
    s, t = fold_bn(sd, f"{head}.BN2")
    z = z * s + t
    z = 0.47 + 0.50 * z + 0.09 * z * z
    return z @ sd[f"{head}.lin2.weight"].astype(np.float64).T + sd[f"{head}.lin2.bias"].astype(np.float64)


def valexnet_from_stem_bits(bits: np.ndarray, sd: Dict[str, np.ndarray], spec, luts) -> Tuple[np.ndarray, np.ndarray]:
    """vAlexnet (models/TT_FHE_XSMALL_vAlexnet.py:490-583, :663-675) from the binarised stem
    output [N,64,10,10]: the concatenated block output bits [N,256,11,11] and float64 logits."""
    out3 = apply_lut(bits, luts[spec.conv3.name], spec.conv3)
    out2 = apply_lut(bits, luts[spec.conv2.name], spec.conv2)
    out1 = apply_lut(bits, luts[spec.conv1.name], spec.conv1)
    p1, p2, p34 = spec.pads
    y = np.concatenate((_zpad(out1, p1), _zpad(out2, p2), _zpad(out3, p34), _zpad(bits, p34)), axis=1)
    z = y.reshape(y.shape[0], -1).astype(np.float64) @ sd["features.7.lin1.weight"].astype(np.float64).T
    s, t = fold_bn(sd, "features.7.BN2")
    z = z * s + t
    return y, z @ sd["features.7.lin2.weight"].astype(np.float64).T + sd["features.7.lin2.bias"].astype(np.float64)


def forward_from_stem_bits(bits: np.ndarray, sd: Dict[str, np.ndarray], spec: VariantSpec, luts,
                           taps: Optional[Dict[str, np.ndarray]] = None, near=None) -> np.ndarray:
    """Gate path + float tail from the binarised stem output. Returns float64 logits.
    ``luts`` may be None (every block evaluated directly in float64)."""
    x = bits
    for blk in spec.blocks:
        x = multihead_block_bits(x, luts, blk, spec.variant, taps, sd, near)
        if taps is not None:
            taps[blk.name] = x
    n_, c, h, w = x.shape
    x = x.astype(np.float64)[:, :, :2 * (h // 2), :2 * (w // 2)]
    x = 0.25 * (x[:, :, 0::2, 0::2] + x[:, :, 0::2, 1::2] + x[:, :, 1::2, 0::2] + x[:, :, 1::2, 1::2])
    feat = x.reshape(n_, -1)
    if taps is not None:
        taps["flatten"] = feat
    return head64(feat, sd, f"features.{4 + len(spec.blocks) + 2}")


def stem_pre64(x: np.ndarray, sd: Dict[str, np.ndarray]) -> np.ndarray:
    """float64 stem pre-activation [N,p,56,56] (avgpool2 -> 7x7/s2 conv pad 3 -> BN)."""
    x = x.astype(np.float64)
    x = 0.25 * (x[:, :, 0::2, 0::2] + x[:, :, 0::2, 1::2] + x[:, :, 1::2, 0::2] + x[:, :, 1::2, 1::2])
    w = sd["features.1.weight"].astype(np.float64)
    n_, c, h, ww = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (3, 3), (3, 3)))
    ho, wo = (h + 6 - 7) // 2 + 1, (ww + 6 - 7) // 2 + 1
    out = np.zeros((n_, w.shape[0], ho, wo))
    for kh in range(7):
        for kw in range(7):
            sl = xp[:, :, kh:kh + 2 * (ho - 1) + 1:2, kw:kw + 2 * (wo - 1) + 1:2]   # [N,3,ho,wo]
            out += np.einsum("nchw,oc->nohw", sl, w[:, :, kh, kw])
    s, t = fold_bn(sd, "features.2")
    return out * s[None, :, None, None] + t[None, :, None, None]


# ---- packing helpers shared by the tests (layouts of include/ttnet.h) ------------------------

def pack_rows(bits: np.ndarray) -> np.ndarray:
    """[N,C,H,W] bits -> row-packed uint64 [N,C,H]; pixel x is bit x (LSB first)."""
    n_, c, h, w = bits.shape
    assert w <= 64
    sh = np.arange(w, dtype=np.uint64)
    return (bits.astype(np.uint64) << sh).sum(axis=-1, dtype=np.uint64)


def pack_channels(bits: np.ndarray) -> np.ndarray:
    """[N,C,H,W] bits -> group-planar channel words uint16 [N,C/16,H,W]; channel 16q+k is bit k."""
    n_, c, h, w = bits.shape
    assert c % 16 == 0
    b = bits.reshape(n_, c // 16, 16, h, w).astype(np.uint16)
    sh = np.arange(16, dtype=np.uint16).reshape(1, 1, 16, 1, 1)
    return (b << sh).sum(axis=2, dtype=np.uint16)


def unpack_channels(words: np.ndarray, c: int) -> np.ndarray:
    n_, q, h, w = words.shape
    sh = np.arange(16, dtype=np.uint16).reshape(1, 1, 16, 1, 1)
    b = ((words[:, :, None] >> sh) & 1).astype(np.uint8)             # [N,Q,16,H,W]
    return b.reshape(n_, q * 16, h, w)[:, :c].copy()


def unpack_rows(words: np.ndarray, w: int) -> np.ndarray:
    sh = np.arange(w, dtype=np.uint64)
    return ((words[..., None] >> sh) & np.uint64(1)).astype(np.uint8)
