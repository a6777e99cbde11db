"""CPU: properties of the Resize / CenterCrop restatement (oracle/pil_resize.py).  Pillow and torchvision
are not importable here, so these are the algorithm's own invariants, not a comparison with Pillow."""
import numpy as np

from oracle import pil_resize as PR


def test_output_size_and_crop_rules():
    assert PR.resized_size(375, 500, 256) == (256, 341)          # int(256 * 500 / 375)
    assert PR.resized_size(500, 333, 256) == (384, 256)          # int(256 * 500 / 333)
    assert PR.resized_size(256, 300, 256) == (256, 300)          # shorter side already matches: unchanged
    assert PR.resized_size(300, 256, 256) == (300, 256)
    img = np.arange(375 * 500 * 3, dtype=np.uint64).reshape(375, 500, 3).astype(np.uint8)
    out = PR.resize_center_crop(img)
    assert out.shape == (224, 224, 3) and out.dtype == np.uint8


def test_constant_and_identity():
    img = np.full((300, 400, 3), 137, dtype=np.uint8)
    assert (PR.resize_center_crop(img) == 137).all()             # normalised coefficients sum to one (within rounding)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(256, 256, 3), dtype=np.uint8)
    assert np.array_equal(PR.resize_center_crop(img), img[16:240, 16:240])      # no resampling, offsets 16


def test_coefficients():
    bounds, kk = PR._coeffs(512, 256)                             # exact 2x down-scaling: support 2, taps (1,3,3,1)/8
    assert (bounds[5] == (9, 4)).all()
    assert kk[5, :4].tolist() == [1 << 19, 3 << 19, 3 << 19, 1 << 19]
    bounds, kk = PR._coeffs(100, 200)                             # up-scaling: plain bilinear, support 1
    assert bounds[:, 1].max() <= 2
    assert (kk.sum(axis=1) >= (1 << 22) - 2).all() and (kk.sum(axis=1) <= (1 << 22) + 2).all()
    img = np.zeros((4, 1024, 3), dtype=np.uint8)
    img[:, ::2] = 255                                             # alternating columns, halved: 127.5 -> 128 (round half up of the fixed-point sum)
    out = PR.resample_axis(img, 512, 1)
    assert set(np.unique(out[:, 4:-4]).tolist()) <= {127, 128}
