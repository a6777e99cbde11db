"""CPU: state_dict layout, geometry and the synthetic generator against committed fixtures."""
import json
import os

import numpy as np
import pytest
import torch

from _util import GOLD, VARIANT_ARGS, args_for, golden_layout, sha, spec_and_state
from scale_imagenet_amd import synth
from scale_imagenet_amd.spec import make_spec, state_dict_layout


@pytest.mark.parametrize("variant", ["small", "xsmall", "full"])
def test_layout_matches_reference_capture(variant):
    g = golden_layout(variant)
    spec = make_spec(variant, **VARIANT_ARGS[variant])
    layout = state_dict_layout(spec)
    assert [k for k, _, _ in g["keys"]] == list(layout.keys())
    for k, shape, dtype in g["keys"]:
        assert tuple(shape) == layout[k][0], k
        assert dtype == layout[k][1], k
    assert len(layout) == 174


def test_small_geometry_matches_survey():
    spec = make_spec("small")
    assert spec.fcsize == 16384 and spec.feat_chw == (1024, 4, 4)
    lookups = 0
    h = 56
    for b in spec.blocks:
        ho = b.conv1.out_hw(h, h)[0]
        lookups += 2 * b.in_planes * ho * ho                      # conv1, conv2: one lookup per output bit
        lookups += (b.in_planes // 16) * h * h                    # conv3: one per pixel and group
        if not b.last:
            lookups += (4 * b.in_planes // 16) * ho * ho          # convf
        h = ho
    assert lookups == 241544                                       # SURVEY §8a A4 total


def test_full_does_not_construct_at_p64():
    with pytest.raises(ValueError):
        make_spec("full", nfilter=8, tfilter=8)


@pytest.mark.parametrize("variant", ["small", "xsmall"])
def test_module_state_dict_is_drop_in(variant):
    from scale_imagenet_amd import ttnet
    cls = {"small": ttnet.TT_vf_19lv3_imgnet_small, "xsmall": ttnet.TT_vf_19lv3_imgnet_xsmall}[variant]
    m = cls(args_for(variant))
    g = golden_layout(variant)
    sd = m.state_dict()
    assert [k for k, _, _ in g["keys"]] == list(sd.keys())
    for k, shape, dtype in g["keys"]:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == "torch." + dtype
    assert sum(p.numel() for p in m.parameters()) == g["n_params"]
    # strict load of a DataParallel-style checkpoint (main.py:181-192, :222)
    spec, st = spec_and_state(variant)
    m.load_state_dict({"module." + k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    assert torch.equal(m.state_dict()["features.1.weight"], torch.from_numpy(st["features.1.weight"]))
    with pytest.raises(RuntimeError):
        bad = dict(st)
        bad.pop("features.1.weight")
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in bad.items()}, strict=True)


def test_module_refuses_cpu_and_training():
    from scale_imagenet_amd import ttnet
    m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
    with pytest.raises(RuntimeError, match="eval"):
        m(torch.zeros(1, 3, 224, 224))
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 3, 224, 224))
    with pytest.raises(NotImplementedError):
        m.features[4](torch.zeros(1, 64, 56, 56))


def test_synth_is_pinned():
    """The generator must produce the same bytes here and on the GPU box: pinned hashes."""
    with open(os.path.join(GOLD, "synth_pins.json")) as f:
        pins = json.load(f)
    assert sha(synth.synth_images_u8(2)) == pins["images_u8_2"]
    assert sha(synth.synth_images(2)) == pins["images_f32_2"]
    assert sha(synth.synth_images(1, first=1)) == sha(synth.synth_images(2)[1:2])
    spec, st = spec_and_state("small")
    for k in pins["small"]:
        assert sha(st[k]) == pins["small"][k], k


def test_synth_shards_are_consistent():
    a = synth.synth_images_u8(6)
    b = np.concatenate([synth.synth_images_u8(3, first=0), synth.synth_images_u8(3, first=3)])
    assert np.array_equal(a, b)
    assert synth.synth_targets(4, first=998).tolist() == [998, 999, 0, 1]


def test_valexnet_layout_and_module():
    from scale_imagenet_amd import ttnet
    from scale_imagenet_amd.spec import valexnet_layout
    g = golden_layout("valexnet")
    layout = valexnet_layout()
    assert [k for k, _, _ in g["keys"]] == list(layout.keys()) and len(layout) == 57
    m = ttnet.TT_FHE_XSMALL_vAlexnet(args_for("valexnet"))
    sd = m.state_dict()
    assert [k for k, _, _ in g["keys"]] == list(sd.keys())
    assert sum(p.numel() for p in m.parameters()) == g["n_params"]
    spec, st = spec_and_state("valexnet")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    assert m.VGG_Model16_0.weight is m.features[0].weight
