"""CPU: the oracle against the golden vectors captured from the imported reference
(oracle/gen_golden.py).  This is what pins oracle/ on every run."""
import numpy as np
import pytest
import torch

from _util import golden_json, golden_npz, sha, spec_and_state
from oracle import ttnet_bits as OB
from oracle import ttnet_float as OF
from scale_imagenet_amd import synth


def _float_forward(variant, n):
    spec, st = spec_and_state(variant)
    sd = OF.to_torch_state(st)
    x = torch.from_numpy(synth.synth_images(n, hw=spec.image_hw))
    taps = {}
    y = OF.forward(x, sd, spec, taps)
    return spec, st, y, taps


@pytest.mark.parametrize("variant", ["small", "xsmall", "full"])
def test_float_oracle_reproduces_reference(variant):
    g, j = golden_npz(variant), golden_json(variant)
    n = int(g["n_images"])
    spec, st, y, taps = _float_forward(variant, n)
    # same container, same torch build: bit for bit.  (On another CPU the float32 noise of
    # oneDNN may differ; the bound below is the north-star tolerance.)
    assert np.abs(y.numpy() - g["logits"]).max() <= 1e-5
    assert np.array_equal(y.argmax(1).numpy(), g["argmax"])
    mism = 0
    for k, info in j["stages"].items():
        if k == "flatten":
            continue
        rows = OB.pack_rows(taps[k].numpy().astype(np.uint8))
        if sha(rows) != info["rows_sha256"]:
            mism += 1
    assert mism == 0, f"{mism} stages differ from the reference capture"
    loss = torch.nn.functional.cross_entropy(y, torch.from_numpy(synth.synth_targets(n))).item()
    assert abs(loss - float(g["loss"])) < 1e-5


def test_bit_oracle_xsmall_end_to_end():
    """Truth tables (float64) + integer evaluation reproduce every reference stage."""
    variant = "xsmall"
    g, j = golden_npz(variant), golden_json(variant)
    n = int(g["n_images"])
    spec, st, y, taps = _float_forward(variant, n)
    luts, near = OB.build_all_luts(st, spec)
    for name, info in j["luts"].items():
        if info["kind"] == "bits":
            assert sha(np.packbits(luts[name], axis=1, bitorder="little")) == info["f64_sha256"], name
            assert int(near[name].sum()) == info["near_ties"]
    bits = taps["features.3"].numpy().astype(np.uint8)
    bt = {}
    logits = OB.forward_from_stem_bits(bits, st, spec, luts, bt)
    for k, info in j["stages"].items():
        if k in ("flatten", "features.3"):
            continue
        assert sha(OB.pack_rows(bt[k])) == info["rows_sha256"], k
    assert np.abs(logits - g["logits"]).max() < 2e-5
    assert np.array_equal(logits.argmax(1), g["argmax"])


def test_bit_oracle_small_first_block():
    """n = 16 tables of the small model: block features.4 on the golden stem bits."""
    variant = "small"
    g, j = golden_npz(variant), golden_json(variant)
    spec, st = spec_and_state(variant)
    blk = spec.blocks[0]
    luts = {}
    for b in (blk.conv1, blk.conv2, blk.conv3, blk.convf):
        luts[b.name], near = OB.build_lut(st, b)
        info = j["luts"][b.name]
        flips = np.array(info["ref_differs_from_f64_at"], dtype=np.int64).reshape(-1, 3)
        assert sha(np.packbits(luts[b.name], axis=1, bitorder="little")) == info["f64_sha256"], b.name
        assert near[tuple(flips.T)].all() if len(flips) else True
        # apply the reference's own near-tie decisions -> the reference's exact table
        for gi, idx, o in flips:
            luts[b.name][gi, idx, o] ^= 1
        assert sha(np.packbits(luts[b.name], axis=1, bitorder="little")) == info["ref_sha256"], b.name
    rows = g["rows:features.3"]
    bits = OB.unpack_rows(rows, 56)
    out = OB.multihead_block_bits(bits, luts, blk, variant)
    assert np.array_equal(OB.pack_rows(out), g["rows:features.4"])


def test_stem_float64_agrees_with_reference_bits():
    g = golden_npz("small")
    spec, st = spec_and_state("small")
    x = synth.synth_images(2)
    pre = OB.stem_pre64(x, st)
    bits = (pre >= 0).astype(np.uint8)
    ref = OB.unpack_rows(g["rows:features.3"], 56)
    diff = np.argwhere(bits != ref)
    # any disagreement must sit on a near tie
    assert all(abs(pre[tuple(d)]) < OB.NEAR_TIE for d in diff)


def test_pack_roundtrip():
    rng = np.random.default_rng(0)
    b = rng.integers(0, 2, size=(2, 32, 5, 29), dtype=np.uint8)
    assert np.array_equal(OB.unpack_rows(OB.pack_rows(b), 29), b)
    assert np.array_equal(OB.unpack_channels(OB.pack_channels(b), 32), b)


def test_valexnet_oracles_reproduce_reference():
    """Config 5 (CIFAR vAlexnet): float oracle == reference capture; bit oracle (float64 tables)
    reproduces every reference stage from the reference's stem bits."""
    g, j = golden_npz("valexnet"), golden_json("valexnet")
    spec, st = spec_and_state("valexnet")
    n = int(g["n_images"])
    taps = {}
    y = OF.forward_valexnet(torch.from_numpy(synth.synth_images(n, hw=(32, 32))), OF.to_torch_state(st), spec, taps)
    assert np.abs(y.numpy() - g["logits"]).max() <= 1e-5 and np.array_equal(y.argmax(1).numpy(), g["argmax"])
    for k, info in j["stages"].items():
        assert sha(OB.pack_rows(taps[k].numpy().astype(np.uint8))) == info["rows_sha256"], k
    luts = {}
    for b in spec.block_tts():
        luts[b.name], near = OB.build_lut(st, b)
        assert sha(np.packbits(luts[b.name], axis=1, bitorder="little")) == j["luts"][b.name]["f64_sha256"]
        assert j["luts"][b.name]["ref_differs_from_f64_at"] == []
    yb, logits = OB.valexnet_from_stem_bits(taps["features.4"].numpy().astype(np.uint8), st, spec, luts)
    assert sha(OB.pack_rows(yb)) == j["stages"]["features.5"]["rows_sha256"]
    assert np.abs(logits - g["logits"]).max() < 2e-5


@pytest.mark.parametrize("layers", [3, 4])
def test_float_oracle_stride1_depths_match_reference(layers):
    """--layers 3 / 4 (stride-1 blocks, TT_general_imagenet_v2_small.py:95-96, :178-181): the oracle
    reproduces the logits and stage bits the imported reference produced (oracle/gen_golden.py depth)."""
    import os
    from _util import GOLD
    from scale_imagenet_amd.spec import make_spec, state_dict_layout
    with np.load(os.path.join(GOLD, f"ref_small_l{layers}.npz")) as z:
        g = {k: z[k] for k in z.files}
    spec = make_spec("small", 8, 8, layers)
    assert len(state_dict_layout(spec)) == int(g["n_keys"])
    st = synth.synth_state_dict(spec, calibrated=False)
    n = int(g["n_images"])
    taps = {}
    y = OF.forward(torch.from_numpy(synth.synth_images(n)), OF.to_torch_state(st), spec, taps)
    assert np.array_equal(y.numpy(), g["logits"])
    for name, want in zip(g["stage_names"], g["stage_sha"]):
        assert sha(OB.pack_rows(taps[str(name)].numpy().astype(np.uint8))) == str(want), name
