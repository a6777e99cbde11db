"""Deterministic synthetic inputs and weights.

There is no checkpoint and no ImageNet here, so parity and the benchmark both run on
synthetic data.  The generator is counter based (splitmix64) and uses only integer
arithmetic plus IEEE add/mul/div/sqrt, so this container (where the golden vectors are
captured from the imported reference) and the GPU box produce bit-identical tensors.
``torch.manual_seed`` is deliberately not used: the reference constructor consumes the
global stream (dry run on ``torch.rand(1,3,224,224)``, ``randint_like`` in every act).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
IMAGE_SEED = 0xC0FFEE
# utils/preprocess.py:107-108 (Normalize of the eval transform)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def splitmix64(seed: int, idx: np.ndarray) -> np.ndarray:
    """z = mix(seed + (idx+1)*golden) for a uint64 index array."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (idx.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _uniform01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    idx = np.arange(n, dtype=np.uint64) + np.uint64(stream) * np.uint64(1 << 40)
    return (splitmix64(seed, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _normalish(seed: int, n: int) -> np.ndarray:
    """Zero-mean unit-variance bell (sum of four uniforms): adds and one multiply only,
    so it is bit-reproducible on every host (no libm)."""
    s = _uniform01(seed, n, 1) + _uniform01(seed, n, 2) + _uniform01(seed, n, 3) + _uniform01(seed, n, 4)
    return (s - 2.0) * 1.7320508075688772  # var(sum of 4 U(0,1)) = 1/3


def synth_images_u8(n: int, first: int = 0, hw: Tuple[int, int] = (224, 224),
                    seed: int = IMAGE_SEED) -> np.ndarray:
    """uint8 [n,3,H,W]: mean of four block-noise layers (cell sizes 1, 4, 16, 56 px), so the
    images carry low-frequency structure that survives the stem (white noise collapses to
    nearly image-independent activations after two blocks).  A pixel depends only on
    (global image index, c, y, x): any shard of a batch (``first`` = index of its first
    image) sees the same pixels.  Integer arithmetic only."""
    h, w = hw
    acc = np.zeros((n, 3, h, w), dtype=np.uint32)
    ni = (np.arange(n, dtype=np.uint64) + np.uint64(first)).reshape(n, 1, 1, 1)
    c = np.arange(3, dtype=np.uint64).reshape(1, 3, 1, 1)
    y = np.arange(h, dtype=np.uint64).reshape(1, 1, h, 1)
    x = np.arange(w, dtype=np.uint64).reshape(1, 1, 1, w)
    for lvl, cell in enumerate((1, 4, 16, 56)):
        idx = (((ni * np.uint64(3) + c) * np.uint64(4) + np.uint64(lvl)) * np.uint64(1024)
               + y // np.uint64(cell)) * np.uint64(1024) + x // np.uint64(cell)
        acc += (splitmix64(seed, idx) & np.uint64(0xFF)).astype(np.uint32)
    return (acc >> 2).astype(np.uint8)


def normalize_u8(u8: np.ndarray) -> np.ndarray:
    """ToTensor + Normalize of the eval transform (utils/preprocess.py:104-108), fp32."""
    x = u8.astype(np.float32) / np.float32(255.0)
    mean = np.asarray(IMAGENET_MEAN, dtype=np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(IMAGENET_STD, dtype=np.float32).reshape(1, 3, 1, 1)
    return ((x - mean) / std).astype(np.float32)


def synth_images(n: int, first: int = 0, hw: Tuple[int, int] = (224, 224),
                 seed: int = IMAGE_SEED) -> np.ndarray:
    return normalize_u8(synth_images_u8(n, first, hw, seed))


def synth_targets(n: int, first: int = 0, n_classes: int = 1000) -> np.ndarray:
    return ((np.arange(n, dtype=np.int64) + first) % n_classes).astype(np.int64)


def synth_tensor(key: str, shape: Tuple[int, ...], dtype: str, weight_seed: int = 0) -> np.ndarray:
    """One state_dict entry.  Distribution by role (after SURVEY §8d, rescaled so that the
    random network stays input-sensitive through all three blocks -- with PyTorch's default
    1/sqrt(fan_in) bound and unit running_var every truth table is nearly constant):
    conv / linear weight and bias ~ U(-b, b), b = sqrt(3/fan_in) (unit-gain); BN gamma ~
    U(0.5,1.5), beta and running_mean ~ bell(0, 0.1^2), running_var ~ U(0.1,0.3);
    num_batches_tracked = 1; grad_scale = 1.  The classifier's BatchNorm1d statistics are
    the exception: see ``synth_state_dict``."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = fnv1a64(key) ^ (weight_seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.ones(shape, dtype=np.int64)
    if leaf == "grad_scale":
        return np.ones(shape, dtype=np.float32)
    if leaf == "running_var":
        v = 0.1 + 0.2 * _uniform01(seed, n)
    elif leaf == "running_mean":
        v = 0.1 * _normalish(seed, n)
    elif leaf == "weight" and len(shape) == 1:          # BatchNorm gamma
        v = 0.5 + _uniform01(seed, n)
    elif leaf == "bias" and ".lin" not in key:           # BatchNorm beta
        v = 0.1 * _normalish(seed, n)
    elif leaf == "weight":                               # conv / linear
        fan_in = int(np.prod(shape[1:]))
        b = np.sqrt(3.0 / float(fan_in))
        v = (2.0 * _uniform01(seed, n) - 1.0) * b
    elif leaf == "bias":                                 # linear bias: fan_in of its layer
        raise ValueError("linear bias needs fan_in; use synth_state_dict")
    else:
        raise ValueError(f"no synthetic rule for {key}")
    return v.astype(np.float32).reshape(shape)


def synth_state_dict(spec, weight_seed: int = 0, calibrated: bool = True) -> "OrderedDict[str, np.ndarray]":
    """Synthetic state_dict for a ``VariantSpec`` in the reference's key order.

    ``calibrated``: the classifier's BatchNorm1d running statistics are not random; they
    are the statistics of the synthetic lin1 outputs over 64 synthetic images (as training
    would have left them), stored as data in ``data/synth_head_bn_<variant>.npz`` by
    oracle/gen_golden.py.  With random statistics the classifier emits one constant class
    for every image, which would make the top-1 check vacuous."""
    from .spec import state_dict_layout, valexnet_layout
    layout = valexnet_layout(spec) if spec.variant == "valexnet" else state_dict_layout(spec)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, (shape, dtype) in layout.items():
        if key.endswith(".bias") and ".lin" in key:
            wshape = layout[key[:-4] + "weight"][0]
            seed = fnv1a64(key) ^ (weight_seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
            b = np.sqrt(3.0 / float(wshape[1]))
            out[key] = ((2.0 * _uniform01(seed, shape[0]) - 1.0) * b).astype(np.float32)
        else:
            out[key] = synth_tensor(key, shape, dtype, weight_seed)
        assert out[key].dtype == np.dtype(dtype), key
    if spec.variant == "valexnet":          # the stem conv is one module under two names
        out["VGG_Model16_0.weight"] = out["features.0.weight"]
        out["VGG_Model16_0.bias"] = out["features.0.bias"]
    if calibrated:
        if weight_seed != 0:
            raise ValueError("calibrated head statistics exist for weight_seed 0 only")
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data",
                            f"synth_head_bn_{spec.variant}.npz")
        with np.load(path) as z:
            head = "features.7.BN2" if spec.variant == "valexnet" else f"features.{4 + len(spec.blocks) + 2}.BN2"
            for leaf in ("running_mean", "running_var"):
                assert z[leaf].shape == out[f"{head}.{leaf}"].shape
                out[f"{head}.{leaf}"] = z[leaf].astype(np.float32)
    return out
