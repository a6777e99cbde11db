"""ORACLE (test infrastructure, not product code) -- Resize + CenterCrop of the eval transform.

The reference's ``transforms.Resize(256)`` / ``CenterCrop(224)`` (utils/preprocess.py:104-105) are
torchvision calling Pillow.  PINNED TO PILLOW: Pillow 12.x is importable in the build container;
``oracle/gen_golden.py resize`` runs ``Image.resize(BILINEAR)`` on seeded images, asserts that this file
is byte-identical to it on 9 geometries and commits Pillow's outputs as ``tests/golden/ref_resize.npz``;
``tests/test_pil_resize_oracle.py`` re-checks this file against that fixture (and against Pillow directly
where it imports), ``tests/test_gpu_preprocess.py`` checks the HIP kernels (csrc/preproc.hip) against the
fixture.  torchvision is NOT importable here: its two integer rules (output size, crop offsets) are restated
from its published source and are the only unpinned part.  This file restates the algorithms in numpy:

  * Pillow 9/10 ``src/libImaging/Resample.c``: ``precompute_coeffs`` (support = filter support x
    max(scale, 1): antialiasing), ``normalize_coeffs_8bpc`` (22-bit fixed point, round half away),
    ``ImagingResampleHorizontal_8bpc`` then ``ImagingResampleVertical_8bpc`` with a uint8
    intermediate, ``clip8``; bilinear filter ``1 - |x|`` on [-1, 1];
  * torchvision ``transforms.functional``: ``_compute_resized_output_size`` (shorter side -> size,
    longer side ``int(size * long / short)``, unchanged if the shorter side already matches) and
    ``center_crop`` (offsets ``int(round((size - crop) / 2.0))``, Python's round-half-even).

Only ``tests/`` may import this module.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _coeffs(in_size: int, out_size: int):
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)], dtype=np.float64)
        ww = 0.0
        for v in w:                      # the C loop sums left to right
            ww += v
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One 8-bit pass along ``axis`` of an [H,W,3] uint8 image."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    bounds, kk = _coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(cnt):
            acc += img[xmin + x] * kk[xx, x]
        out[xx] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def resized_size(h: int, w: int, size: int):
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w <= h:
        return int(size * h / w), size
    return size, int(size * w / h)


def resize_center_crop(img: np.ndarray, size: int = 256, crop: int = 224) -> np.ndarray:
    """[H,W,3] uint8 -> [crop,crop,3] uint8, as Resize(size) -> CenterCrop(crop) on a PIL image."""
    h, w, _ = img.shape
    nh, nw = resized_size(h, w, size)
    out = img
    if nw != w:
        out = resample_axis(out, nw, 1)          # horizontal pass first
    if nh != h:
        out = resample_axis(out, nh, 0)
    top, left = int(round((nh - crop) / 2.0)), int(round((nw - crop) / 2.0))
    return out[top:top + crop, left:left + crop].copy()
