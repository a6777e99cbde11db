"""GPU: Resize(256) + CenterCrop(224) on the device against Pillow's own output (the committed
fixture tests/golden/ref_resize.npz, written by oracle/gen_golden.py resize from Pillow 12.x) and against the
numpy restatement of Pillow's 8-bit bilinear resampling (oracle/pil_resize.py, itself pinned to that fixture by
tests/test_pil_resize_oracle.py); then the whole eval transform + forward against the float32 path."""
import numpy as np
import pytest
import torch

from _util import RESIZE_GEOMETRIES, args_for, golden_resize, resize_test_images, sha, spec_and_state
from oracle import pil_resize as PR
from scale_imagenet_amd import _lib, preprocess, synth, ttnet

pytestmark = pytest.mark.gpu


_images = resize_test_images


@pytest.mark.parametrize("h,w", RESIZE_GEOMETRIES)
def test_resize_center_crop_matches_pillow(h, w):
    """preproc.hip byte for byte against Pillow's output on the same seeded images (fixture) and against
    oracle/pil_resize.py."""
    dev = torch.device("cuda", 0)
    x = _images(2, h, w, seed=h * 1000 + w)
    g = golden_resize()
    if min(PR.resized_size(h, w, 256)) < 224:
        with pytest.raises(_lib.TTNetError):
            preprocess.resize_center_crop_u8(torch.from_numpy(x).to(dev))
        return
    got = preprocess.resize_center_crop_u8(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert got.shape == (2, 224, 224, 3)
    assert [sha(got[i]) for i in range(2)] == g[f"sha_{h}x{w}"].tolist(), (h, w, "differs from Pillow's output")
    if f"crop_{h}x{w}" in g:
        assert np.array_equal(got[0], g[f"crop_{h}x{w}"])
    for i in range(2):
        want = PR.resize_center_crop(x[i])
        assert np.array_equal(got[i], want), (h, w, int(np.abs(got[i].astype(int) - want.astype(int)).max()))


def test_eval_transform_then_forward():
    """Resize + crop on the GPU, ToTensor + Normalize in the stem: equal to the float32 path fed with the
    oracle's crop (up to stem near ties, as test_uint8_input_fused_normalise)."""
    dev = torch.device("cuda", 0)
    spec, st = spec_and_state("small")
    m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    m = m.to(dev).eval().reserve(4)
    x = _images(3, 375, 500, seed=7)
    with torch.no_grad():
        y = preprocess.imgnet_eval_forward(m, torch.from_numpy(x).to(dev)).cpu().numpy()
        crop = np.stack([PR.resize_center_crop(x[i]) for i in range(3)])
        xf = synth.normalize_u8(np.ascontiguousarray(crop.transpose(0, 3, 1, 2)))
        want = m(torch.from_numpy(xf).to(dev)).cpu().numpy()
    assert y.shape == (3, 1000)
    assert (np.abs(y - want).max(axis=1) <= 1e-5).sum() >= 2          # (an image may cross a stem near tie)
    with pytest.raises(RuntimeError):
        preprocess.resize_center_crop_u8(torch.zeros((1, 3, 300, 300), dtype=torch.uint8, device=dev))
