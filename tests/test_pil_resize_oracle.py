"""CPU: the Resize / CenterCrop restatement (oracle/pil_resize.py) against Pillow's own output -- the committed
fixture tests/golden/ref_resize.npz (oracle/gen_golden.py resize: Pillow 12.x ``Image.resize(BILINEAR)`` +
torchvision's crop offsets on seeded images) and, where Pillow is importable, Pillow itself -- plus the
algorithm's own invariants.  torchvision is not importable in the build container: its size and crop rules
(two integer formulas) are restated."""
import numpy as np
import pytest

from _util import RESIZE_GEOMETRIES, golden_resize, resize_test_images, sha
from oracle import pil_resize as PR


@pytest.mark.parametrize("h,w", RESIZE_GEOMETRIES)
def test_restatement_matches_the_pillow_fixture(h, w):
    g = golden_resize()
    x = resize_test_images(2, h, w, seed=h * 1000 + w)
    crops = [PR.resize_center_crop(x[i]) for i in range(2)]
    assert [sha(c) for c in crops] == g[f"sha_{h}x{w}"].tolist()
    if f"crop_{h}x{w}" in g:
        assert np.array_equal(crops[0], g[f"crop_{h}x{w}"])


@pytest.mark.parametrize("h,w", [(375, 500), (500, 333), (257, 256), (1200, 900), (231, 640), (256, 256)])
def test_restatement_matches_pillow_directly(h, w):
    Image = pytest.importorskip("PIL.Image")
    x = resize_test_images(1, h, w, seed=17 * h + w)[0]
    nh, nw = PR.resized_size(h, w, 256)
    im = Image.fromarray(x)
    if (nh, nw) != (h, w):
        im = im.resize((nw, nh), Image.BILINEAR)
    a = np.asarray(im)
    top, left = int(round((nh - 224) / 2.0)), int(round((nw - 224) / 2.0))
    assert np.array_equal(PR.resize_center_crop(x), a[top:top + 224, left:left + 224])


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
