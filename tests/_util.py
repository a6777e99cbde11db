"""Shared helpers for the tests (test infrastructure; may import oracle/)."""
import hashlib
import json
import os
from argparse import Namespace
from functools import lru_cache

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

VARIANT_ARGS = {
    "small": dict(nfilter=8, tfilter=8, layers=1),
    "xsmall": dict(nfilter=8, tfilter=8, layers=1),
    "full": dict(nfilter=6, tfilter=10, layers=1),
}


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def args_for(variant: str) -> Namespace:
    return Namespace(groups=[1, None, 4, None], **VARIANT_ARGS.get(variant, VARIANT_ARGS["small"]))


@lru_cache(maxsize=None)
def golden_npz(variant: str):
    with np.load(os.path.join(GOLD, f"ref_{variant}.npz")) as z:
        return {k: z[k] for k in z.files}


@lru_cache(maxsize=None)
def golden_json(variant: str):
    with open(os.path.join(GOLD, f"ref_luts_{variant}.json")) as f:
        return json.load(f)


@lru_cache(maxsize=None)
def golden_layout(variant: str):
    with open(os.path.join(GOLD, f"state_layout_{variant}.json")) as f:
        return json.load(f)


@lru_cache(maxsize=None)
def spec_and_state(variant: str):
    from scale_imagenet_amd.spec import VAlexSpec, make_spec
    from scale_imagenet_amd.synth import synth_state_dict
    spec = VAlexSpec() if variant == "valexnet" else make_spec(variant, **VARIANT_ARGS[variant])
    return spec, synth_state_dict(spec)


RESIZE_GEOMETRIES = [(375, 500), (500, 333), (256, 256), (300, 256), (256, 341), (224, 224), (1200, 900), (333, 500),
                     (480, 640)]


def resize_test_images(n: int, h: int, w: int, seed: int) -> np.ndarray:
    """uint8 [n,h,w,3] test images for the Resize / CenterCrop checks: 8-pixel blocks plus noise (edges and
    flats).  numpy's PCG64 stream is stable across versions, so oracle/gen_golden.py (which feeds these to
    Pillow) and the tests regenerate identical inputs."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(n, h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    img = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)[:, :h, :w].astype(np.int16)
    img += rng.integers(-20, 21, size=img.shape, dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


@lru_cache(maxsize=None)
def golden_resize():
    with np.load(os.path.join(GOLD, "ref_resize.npz")) as z:
        return {k: z[k] for k in z.files}
