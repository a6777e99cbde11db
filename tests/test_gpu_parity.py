"""GPU parity tests: the HIP path, called through the C ABI (libttnet.so via the nn.Module
mirror), against the golden vectors captured from the imported reference and against the
oracle on the same seeded inputs.

Bars (north star): integer gate path bit exact; logits within 1e-5; top-1 exact match.
Float -> bit boundaries (truth-table entries, stem bits) are exact except at near ties
(|pre-activation| < 1e-5), which are listed in the fixtures and reported, never hidden.
"""
import os

import numpy as np
import pytest
import torch

import json

from _util import GOLD, args_for, golden_json, golden_npz, sha, spec_and_state
from oracle import ttnet_bits as OB
from oracle import ttnet_float as OF
from scale_imagenet_amd import _lib, synth, ttnet

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-5          # the north star's bound; the only absolute logit tolerance in this file
# The reference's own float32 head is not exact: tests/golden/ref_spread.json (written by
# oracle/gen_golden.py from the imported reference) records, per variant, how far its committed
# logits sit from the float64 evaluation of the same head on its own features (ref_vs_exact:
# 5.8e-6 / 1.0e-5 / 6.8e-6) and how far they move with its thread count alone (4.4e-6 / 6.3e-6 /
# 1.9e-6) at logits of magnitude ~3.  So: |hip - exact| <= 1e-5, and against the reference capture
# the allowance is 1e-5 + that committed distance.
with open(os.path.join(GOLD, "ref_spread.json")) as _f:
    _SPREAD = json.load(_f)
REF_SPREAD = _SPREAD["variants"]
# the same for the depth / width fixtures (uncalibrated heads, |logit| ~16): keyed by fixture file
GEOMETRY_SPREAD = {v["fixture"]: v for v in _SPREAD.get("geometries", {}).values()}


def ref_allowance(variant):
    return LOGIT_TOL + REF_SPREAD[variant]["ref_vs_exact"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda", 0)


CLASSES = {"small": ttnet.TT_vf_19lv3_imgnet_small, "xsmall": ttnet.TT_vf_19lv3_imgnet_xsmall,
           "full": ttnet.TT_vf_19lv3_imgnet}
_MODELS, _TAPS = {}, {}


@pytest.fixture(scope="module", params=["small", "xsmall", "full"])
def variant(request):
    return request.param


def _model(variant, dev):
    if variant not in _MODELS:
        spec, st = spec_and_state(variant)
        m = CLASSES[variant](args_for(variant))
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
        m = m.to(dev).eval()
        with torch.no_grad():
            m(torch.from_numpy(synth.synth_images(1)).to(dev))       # builds the plan and the tables
        torch.cuda.synchronize()
        _MODELS[variant] = m
    return _MODELS[variant]


@pytest.fixture(scope="module")
def model(variant, dev):
    return _model(variant, dev)


@pytest.fixture(scope="module")
def small_model(dev):
    return _model("small", dev)


def scaled_tol(ref):
    """1e-5 is stated for logits of the reference's scale (|logit| <= ~3 on the golden images).  Inputs
    far from the calibration set (random stem bits, uncalibrated heads) give logits k times larger,
    where float32 itself no longer resolves 1e-5 (an ulp at 64 is 7.6e-6): the bound scales with k."""
    return LOGIT_TOL * max(1.0, float(np.abs(ref).max()) / 4.0)


@pytest.fixture(scope="module")
def oracle_small_rows():
    """features.5 of the 8 golden images as the reference computes it (row-packed)."""
    return OB.pack_rows(_oracle_taps("small")[1]["features.5"].astype(np.uint8))


@pytest.fixture(scope="module")
def oracle_taps(variant):
    return _oracle_taps(variant)


def _oracle_taps(variant):
    """Reference-identical stages for the 8 golden images (the float oracle is pinned to the
    reference bit for bit by tests/test_oracle_golden.py)."""
    if variant not in _TAPS:
        spec, st = spec_and_state(variant)
        sd = OF.to_torch_state(st)
        x = torch.from_numpy(synth.synth_images(int(golden_npz(variant)["n_images"])))
        taps = {}
        y = OF.forward(x, sd, spec, taps)
        _TAPS[variant] = (y.numpy(), {k: v.numpy() for k, v in taps.items()})
    return _TAPS[variant]


def _stage_width(spec, stage):
    if stage == "features.3":
        return 56
    for b in spec.blocks:
        if stage.startswith(b.name):
            return b.out_hw[1]
    raise KeyError(stage)


def test_library_is_the_native_one(small_model):
    assert _lib.load().ttnet_version().startswith(b"ttnet-mi355x")
    assert small_model._any_plan().query("fcsize") == 16384
    assert small_model._any_plan().query("n_state_tensors") == 174


def test_truth_tables_match_float64_oracle(model, variant):
    """GPU-built tables == float64 numpy tables (hash), near-tie counts equal; and they
    differ from the reference's own float32 tables only at the listed near ties."""
    if variant == "full":           # fan-in 30: no tables exist (2^30 entries per output bit)
        with pytest.raises(_lib.TTNetError, match="no truth tables"):
            model.get_table("features.4.Block_conv3")
        return
    j = golden_json(variant)
    spec, st = spec_and_state(variant)
    ties = model.near_ties()
    for b in spec.block_tts():
        info = j["luts"][b.name]
        tab = model.get_table(b.name)
        if b.last:           # the float table of the last block: every group, in chunks (float64 -> float32 once)
            worst = 0.0
            for g0 in range(0, b.groups, 8):
                gl = list(range(g0, min(b.groups, g0 + 8)))
                ref, _ = OB.build_lut(st, b, groups=gl)
                worst = max(worst, float(np.abs(tab[gl] - ref).max()))
            assert worst <= 1e-6, (b.name, worst)
            continue
        got = sha(np.packbits(tab, axis=1, bitorder="little"))
        if got != info["f64_sha256"]:                  # an erf ulp could move an exact tie: then
            ref, near = OB.build_lut(st, b)            # every difference must be a near tie
            d = np.argwhere(ref != tab)
            assert len(d) > 0 and near[tuple(d.T)].all(), f"{b.name}: table differs outside the near-tie set"
        assert ties[b.name] == info["near_ties"], b.name
        flips = np.array(info["ref_differs_from_f64_at"], dtype=np.int64).reshape(-1, 3)
        patched = tab.copy()
        for gi, idx, o in flips:
            patched[gi, idx, o] ^= 1
        assert sha(np.packbits(patched, axis=1, bitorder="little")) == info["ref_sha256"], b.name


def test_stem_bits(model, variant, dev, oracle_taps):
    """Float stem: bits equal the reference except (possibly) at near ties."""
    g = golden_npz(variant)
    n = int(g["n_images"])
    x = torch.from_numpy(synth.synth_images(n)).to(dev)
    with torch.no_grad():
        model(x)
    rows = model.read_stage("features.3", n)
    ref_bits = oracle_taps[1]["features.3"].astype(np.uint8)
    bits = OB.unpack_rows(rows, 56)
    diff = np.argwhere(bits != ref_bits)
    pre = oracle_taps[1]["stem.pre"]
    for d in diff:
        assert abs(pre[tuple(d)]) < OB.NEAR_TIE, f"stem bit {tuple(d)} differs away from a tie: pre={pre[tuple(d)]}"
    print(f"stem: {len(diff)} of {bits.size} bits differ from the reference (all near ties)")
    if len(diff) == 0:       # then the packed words are the committed ones, bit for bit
        assert np.array_equal(rows[:2], g["rows:features.3"])
    else:                    # only images that own one of the listed differences may differ
        dirty = set(int(d[0]) for d in diff)
        for i in (0, 1):
            assert i in dirty or np.array_equal(rows[i], g["rows:features.3"][i])


def test_stem_bits_batch64_vs_float64(small_model, dev):
    """The fp16 x 2 split stem on 64 images (12.8 M outputs) against the float64 oracle: every
    bit outside the near-tie band |pre| < 1e-5 equals (pre >= 0)."""
    n = 64
    spec, st = spec_and_state("small")
    xh = synth.synth_images(n)
    pre = OB.stem_pre64(xh, st)
    with torch.no_grad():
        small_model(torch.from_numpy(xh).to(dev))
    bits = OB.unpack_rows(small_model.read_stage("features.3", n), 56)
    want = (pre >= 0).astype(np.uint8)
    bad = np.argwhere(bits != want)
    worst = max((abs(pre[tuple(d)]) for d in bad), default=0.0)
    print(f"stem (64 images): {len(bad)} of {bits.size} bits differ from float64, largest |pre| there {worst:.2e}")
    assert worst < OB.NEAR_TIE


def _with_reference_tables(model, variant):
    """Patch the GPU tables with the reference's own near-tie decisions (fixture)."""
    j = golden_json(variant)
    spec, _ = spec_and_state(variant)
    saved = {}
    if variant == "full":
        return saved
    for b in spec.block_tts():
        if b.last:
            continue
        flips = np.array(j["luts"][b.name]["ref_differs_from_f64_at"], dtype=np.int64).reshape(-1, 3)
        if len(flips):
            tab = model.get_table(b.name)
            saved[b.name] = tab.copy()
            for gi, idx, o in flips:
                tab[gi, idx, o] ^= 1
            model.set_table(b.name, tab)
    return saved


def test_gate_path_bit_exact_and_logits(model, variant, dev, oracle_taps):
    """Integer gate path from the reference's stem bits: every stage of every image is bit
    identical to the reference capture; logits within 1e-5; top-1 equal."""
    g, j = golden_npz(variant), golden_json(variant)
    n = int(g["n_images"])
    saved = _with_reference_tables(model, variant)
    try:
        stem_rows = OB.pack_rows(oracle_taps[1]["features.3"].astype(np.uint8))
        assert sha(stem_rows) == j["stages"]["features.3"]["rows_sha256"]
        rows_t = torch.from_numpy(stem_rows.view(np.int64)).to(dev)
        with torch.no_grad():
            logits = model.forward_from_stem_bits(rows_t).cpu().numpy()
        for stage, info in j["stages"].items():
            if stage in ("flatten", "features.3"):
                continue
            got = model.read_stage(stage, n)
            assert [sha(got[i]) for i in range(n)] == info["per_image_sha256"], stage
        feat = model.read_stage("flatten", n)
        assert np.abs(feat[:2] - g["features_flat"]).max() <= 1e-5
        # Float head.  The reference's own float32 head is not exact: on these features it sits
        # up to ~6e-6 (small) / ~1.1e-5 (xsmall) from the float64 evaluation of the same head,
        # and moves by ~5e-6 with its thread count.  So: within 1e-5 of the exact head, and
        # within 1e-5 + (the reference's own deviation from exact) of the reference capture.
        spec, st = spec_and_state(variant)
        exact = OB.head64(oracle_taps[1]["flatten"], st, f"features.{4 + len(spec.blocks) + 2}")
        ref_dev = float(np.abs(g["logits"] - exact).max())
        print(f"{variant}: |gpu-exact| {np.abs(logits - exact).max():.2e}  |ref-exact| {ref_dev:.2e} "
              f"(fixture {REF_SPREAD[variant]['ref_vs_exact']:.2e})  |gpu-ref| {np.abs(logits - g['logits']).max():.2e}")
        # (ref_dev is recomputed from THIS box's oracle features, whose float32 convolutions differ in the
        # last bits from the build container's: 6.0e-6 vs 5.8e-6 for TT-small; the allowance is the fixture's)
        assert np.abs(logits - exact).max() <= LOGIT_TOL
        assert np.abs(logits - g["logits"]).max() <= ref_allowance(variant)
        assert np.array_equal(logits.argmax(1), g["argmax"])
    finally:
        for name, tab in saved.items():
            model.set_table(name, tab)


def test_end_to_end_forward(model, variant, dev, oracle_taps):
    """model(inputs) exactly as main.py:261 calls it, GPU-built float64 tables."""
    g = golden_npz(variant)
    n = int(g["n_images"])
    spec, _ = spec_and_state(variant)
    x = torch.from_numpy(synth.synth_images(n)).to(dev)
    with torch.no_grad():
        y = model(x).cpu().numpy()
    # which images touch a table entry where float64 and the reference's float32 disagree?
    clean = np.ones(n, dtype=bool)
    for stage in ("features.4", "features.5"):
        got = OB.unpack_rows(model.read_stage(stage, n), _stage_width(spec, stage))
        ref = oracle_taps[1][stage].astype(np.uint8)
        clean &= (got == ref).reshape(n, -1).all(axis=1)
    print(f"end to end: {int(clean.sum())}/{n} images bit identical to the reference through the gate path")
    assert clean.sum() >= n - 2
    exact = OB.head64(oracle_taps[1]["flatten"], spec_and_state(variant)[1], f"features.{4 + len(spec.blocks) + 2}")
    assert np.abs(y[clean] - exact[clean]).max() <= LOGIT_TOL
    assert np.abs(y[clean] - g["logits"][clean]).max() <= ref_allowance(variant)
    assert np.array_equal(y[clean].argmax(1), g["argmax"][clean])
    # an image that crossed a listed near tie (at most 2 of them, asserted above) carries a few flipped
    # bits through the blocks: its logits are NOT covered by the 1e-5 claim; they stay close (sanity)
    if (~clean).any():
        assert np.abs(y[~clean] - g["logits"][~clean]).max() < 1e-2


def test_full_batch_properties(small_model, dev, oracle_small_rows):
    """BASELINE size (batch 256): determinism, batch-composition invariance of the integer
    path, and agreement of a 256-image forward with 8-image forwards."""
    model = small_model
    n = 256
    x = torch.from_numpy(synth.synth_images(n)).to(dev)
    with torch.no_grad():
        y1 = model(x)
        s4 = model.read_stage("features.4", n).copy()
        s5 = model.read_stage("features.5", n).copy()
        f1 = model.read_stage("flatten", n).copy()
        y2 = model(x)
    assert torch.equal(y1, y2), "forward is not deterministic"
    assert np.array_equal(s4, model.read_stage("features.4", n))
    with torch.no_grad():
        ya = model(x[8:16])
        s4a = model.read_stage("features.4", 8)
        fa = model.read_stage("flatten", 8)
    assert np.array_equal(s4a, s4[8:16]), "gate bits depend on batch composition"
    assert np.array_equal(fa, f1[8:16])
    assert (ya - y1[8:16]).abs().max().item() <= LOGIT_TOL
    g = golden_npz("small")
    # the 8 golden images lead the batch: those whose gate bits equal the reference's are top-1 equal
    same = (s5[:8] == oracle_small_rows).reshape(8, -1).all(axis=1)
    assert same.sum() >= 6
    assert np.array_equal(y1[:8].argmax(1).cpu().numpy()[same], g["argmax"][same])
    assert s5.shape == (n, 256, 15)
    assert len(set(y1.argmax(1).tolist())) > 20          # the synthetic classifier is not degenerate


def test_large_batch_rounds_equal_small_batches(dev):
    """Batches large enough that a workgroup of the block-fused gate kernel walks several rounds (more images
    per workgroup than its LDS scratch holds: 600 images = 19 per workgroup of the first block, rounds of 8,
    the last one partial) must give exactly the bits and logits of the same images run 200 at a time."""
    spec, st = spec_and_state("small")
    m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    m = m.to(dev).eval().reserve(600)
    base = torch.from_numpy(synth.synth_images(200)).to(dev)
    x = torch.cat([base, base.flip(0), base.roll(7, 0)])          # 600 images, every one a known image
    with torch.no_grad():
        y = m(x).clone()
        s4 = m.read_stage("features.4", 600).copy()
        s5 = m.read_stage("features.5", 600).copy()
        o3 = m.read_stage("features.6.out3", 600).copy()
        ya = m(base).clone()
        a4 = m.read_stage("features.4", 200).copy()
        a5 = m.read_stage("features.5", 200).copy()
        b3 = m.read_stage("features.6.out3", 200).copy()
    assert np.array_equal(s4[:200], a4) and np.array_equal(s4[200:400], a4[::-1]) and np.array_equal(s4[400:], np.roll(a4, 7, 0))
    assert np.array_equal(s5[:200], a5) and np.array_equal(s5[400:], np.roll(a5, 7, 0))
    assert np.array_equal(o3[:200], b3) and np.array_equal(o3[200:400], b3[::-1])
    # logits: lin1 splits K differently at M = 600 than at M = 200 (another float32 summation order)
    want = torch.cat([ya, ya.flip(0), ya.roll(7, 0)])
    assert (y - want).abs().max().item() <= LOGIT_TOL
    assert torch.equal(y.argmax(1), want.argmax(1))


def test_graph_replay_matches_plain_launches(small_model, dev):
    """From the third forward with one batch size the C ABI replays a captured hipGraph; the
    replay must equal the plain launches bit for bit, with fresh input / output buffers."""
    model = small_model
    n = 24
    xs = [torch.from_numpy(synth.synth_images(n, first=100 * i)).to(dev) for i in range(3)]
    model.set_profiling(True)                       # profiling mode = plain launches
    with torch.no_grad():
        want = [model(x).clone() for x in xs]
    model.set_profiling(False)
    plan = model._any_plan()
    before = plan.query("graph_replays")
    with torch.no_grad():
        for _ in range(3):
            model(xs[0])                            # warm-up calls and the capture
        got = [model(x).clone() for x in xs]        # replays with patched pointers
        rows = model.read_stage("features.3", n).copy()
    torch.cuda.synchronize()
    assert plan.query("graphs_enabled") == 1, _lib.load().ttnet_last_error()     # gfx950: capture must work
    assert plan.query("graph_replays") > before
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    assert rows.shape[0] == n


def test_lanes_in_flight_match_lane0(dev):
    """Three batches in flight on three lanes / streams (shared tables, separate activation
    workspaces) give the logits of the same batches run one at a time; evaluate() pipelined
    equals evaluate() serial."""
    from scale_imagenet_amd.evaluate import evaluate
    spec, st = spec_and_state("small")
    m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    m = m.to(dev).eval().reserve(16)
    xs = [torch.from_numpy(synth.synth_images(16, first=16 * i)).to(dev) for i in range(6)]
    with torch.no_grad():
        want = [m(x).clone() for x in xs]
    m.set_lanes(3)
    assert m._any_plan().query("lanes") == 3
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    got = [None] * len(xs)
    with torch.no_grad():
        for rep in range(4):                        # plain launches first, then captured graphs per lane
            for i, x in enumerate(xs):
                with torch.cuda.stream(streams[i % 3]):
                    got[i] = m(x, lane=i % 3)
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    with pytest.raises(Exception):
        m(xs[0], lane=5)
    batches = [(x.cpu(), torch.from_numpy(synth.synth_targets(16, first=16 * i))) for i, x in enumerate(xs)]
    a = evaluate(m, batches, dev, inflight=1)
    b = evaluate(m, batches, dev, inflight=3)
    assert (a.loss, a.top1, a.top5, a.images) == (b.loss, b.top1, b.top5, b.images)


def test_uint8_input_fused_normalise(model, variant, dev):
    """uint8 HWC input with ToTensor + Normalize fused into the stem.  Bits equal
    (pre >= 0) of the float64 oracle on the normalised float32 tensor outside the near-tie band;
    where the stem bits agree with the float32-input path, the logits are identical."""
    spec, st = spec_and_state(variant)
    n = 16
    u8 = synth.synth_images_u8(n)
    xf = synth.normalize_u8(u8)
    x_u8 = torch.from_numpy(np.ascontiguousarray(u8.transpose(0, 2, 3, 1))).to(dev)
    with torch.no_grad():
        y_f = model(torch.from_numpy(xf).to(dev)).clone()
        rows_f = model.read_stage("features.3", n).copy()
        for _ in range(4):                         # plain launches, then the captured graph
            y_u = model.forward_u8(x_u8).clone()
        rows_u = model.read_stage("features.3", n).copy()
    pre = OB.stem_pre64(xf, st)
    bits = OB.unpack_rows(rows_u, 56)
    bad = np.argwhere(bits != (pre >= 0).astype(np.uint8))
    worst = max((abs(pre[tuple(d)]) for d in bad), default=0.0)
    print(f"{variant} uint8 stem: {len(bad)} of {bits.size} bits differ from float64, largest |pre| there {worst:.2e}")
    assert worst < OB.NEAR_TIE
    same = (rows_u == rows_f).reshape(n, -1).all(axis=1)
    assert same.sum() >= n - 2
    assert torch.equal(y_u[torch.from_numpy(same).to(dev)], y_f[torch.from_numpy(same).to(dev)])
    # other constants: mean 0 / std 1 is plain ToTensor
    model.set_input_norm((0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    try:
        x01 = (u8[:2].astype(np.float32) / np.float32(255.0)).astype(np.float32)
        with torch.no_grad():
            model.forward_u8(x_u8[:2])
            r_u = model.read_stage("features.3", 2).copy()
        p01 = OB.stem_pre64(x01, st)
        b01 = OB.unpack_rows(r_u, 56)
        bad = np.argwhere(b01 != (p01 >= 0).astype(np.uint8))
        assert max((abs(p01[tuple(d)]) for d in bad), default=0.0) < OB.NEAR_TIE
    finally:
        model.set_input_norm(synth.IMAGENET_MEAN, synth.IMAGENET_STD)
    with pytest.raises(RuntimeError):
        model.forward_u8(x_u8.permute(0, 3, 1, 2).contiguous())          # CHW is not the contract


@pytest.mark.parametrize("nfilter,tfilter", [(4, 8), (16, 8)])
def test_uint8_input_other_widths(dev, nfilter, tfilter):
    """The uint8 stem (two products, normalisation folded into the weights, border-class corrections) with one and with four
    32-channel M-tiles (p = 32, 128): bits of the float64 oracle on the normalised tensor outside the near-tie band."""
    from argparse import Namespace
    from scale_imagenet_amd.spec import make_spec
    spec = make_spec("small", nfilter, tfilter, 1)
    st = synth.synth_state_dict(spec, calibrated=False)
    m = ttnet.TT_vf_19lv3_imgnet_small(Namespace(nfilter=nfilter, tfilter=tfilter, layers=1, groups=[1, None, 4, None]))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to(dev).eval().reserve(4)
    u8 = synth.synth_images_u8(4)
    # the borders carry the corrections: make them matter (bright first / last rows and columns)
    u8[:, :, :4, :] = 255
    u8[:, :, -4:, :] = 3
    u8[:, :, :, :4] = 200
    u8[:, :, :, -4:] = 17
    xf = synth.normalize_u8(u8)
    with torch.no_grad():
        m.forward_u8(torch.from_numpy(np.ascontiguousarray(u8.transpose(0, 2, 3, 1))).to(dev))
    bits = OB.unpack_rows(m.read_stage("features.3", 4), 56)
    pre = OB.stem_pre64(xf, st)
    bad = np.argwhere(bits != (pre >= 0).astype(np.uint8))
    worst = max((abs(pre[tuple(d)]) for d in bad), default=0.0)
    print(f"p = {nfilter * tfilter} uint8 stem: {len(bad)} of {bits.size} bits differ from float64, largest |pre| there {worst:.2e}")
    assert worst < OB.NEAR_TIE


def test_truth_table_export_from_gpu_tables(dev, tmp_path):
    """the files exported from the GPU-built table of an x-small block equal the
    files the reference's own exporter wrote (fixture from oracle/gen_golden.py)."""
    import json
    from _util import GOLD
    m = _model("xsmall", dev)
    with open(os.path.join(GOLD, "ref_export_xsmall.json")) as f:
        g = json.load(f)
    table = m.get_table(g["block"])
    compared = 0
    for f_str, want in g["filters"].items():
        fi = int(f_str)
        if table[fi, :, 0].astype(int).tolist() != want["column"]:
            continue
        got = m.export_truth_tables(g["block"], str(tmp_path / f_str), g["blockici"], g["sousblockici"], filters=[fi])[fi]
        assert (got["dnf"], got["cnf"]) == (want["dnf"], want["cnf"])
        files = {n: open(tmp_path / f_str / n).read() for n in sorted(os.listdir(tmp_path / f_str))}
        assert files == want["files"]
        compared += 1
    assert compared >= 10


def test_ragged_batches_plan_regrowth_and_lanes(dev):
    """Batch sizes around the 32-image tile edges, a forward larger than the reserved workspace
    (the plan is rebuilt with the lanes it had), and both input kinds on both lanes: every result
    equals the corresponding rows of one large forward."""
    spec, st = spec_and_state("small")
    m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    m = m.to(dev).eval().reserve(40).set_lanes(2)
    u8 = synth.synth_images_u8(70)
    x = torch.from_numpy(synth.normalize_u8(u8)).to(dev)
    with torch.no_grad():
        small = {n: m(x[:n]).clone() for n in (1, 2, 31, 33, 40)}
        big = m(x).clone()                           # 70 > 40: the plan grows
        assert m._any_plan().query("lanes") == 2 and m._any_plan().query("max_batch") >= 70
        for n, y in small.items():
            assert torch.equal(y, big[:n]), n
        for rep in range(4):
            a, b = m(x, lane=0), m(x, lane=1)
        xu = torch.from_numpy(np.ascontiguousarray(u8[:33].transpose(0, 2, 3, 1))).to(dev)
        for rep in range(4):
            yu = m.forward_u8(xu, lane=rep % 2)
    torch.cuda.synchronize()
    assert torch.equal(a, big) and torch.equal(b, big)
    assert (yu - big[:33]).abs().max().item() <= 1e-3      # equal unless a stem near tie flips


def test_majority_and_padding_edges(model, variant, dev):
    """Edge inputs of the integer path: all-zero and all-one stem bits, checked against the
    bit oracle with the GPU's own tables."""
    spec, st = spec_and_state(variant)
    luts = None if variant == "full" else {b.name: model.get_table(b.name) for b in spec.block_tts()}
    for fill in (0, 1):
        bits = np.full((2, spec.p, 56, 56), fill, dtype=np.uint8)
        bits[1, ::3, ::5, ::7] ^= 1
        rows_t = torch.from_numpy(OB.pack_rows(bits).view(np.int64)).to(dev)
        with torch.no_grad():
            y = model.forward_from_stem_bits(rows_t).cpu().numpy()
        bt = {}
        ref = OB.forward_from_stem_bits(bits, st, spec, luts, bt)
        for stage in ("features.4.out1", "features.4.out3", "features.4.out4", "features.4", "features.5",
                      "features.6.out2", "features.6.out3"):
            assert np.array_equal(model.read_stage(stage, 2), OB.pack_rows(bt[stage])), (fill, stage)
        print(f"{variant} fill={fill}: |logit| max {np.abs(ref).max():.2f}, |gpu - exact| {np.abs(y - ref).max():.2e}")
        assert np.abs(y - ref).max() <= scaled_tol(ref)  # ref: the float64 head on the bit oracle's features


def test_random_bits_against_bit_oracle(model, variant, dev):
    spec, st = spec_and_state(variant)
    luts = None if variant == "full" else {b.name: model.get_table(b.name) for b in spec.block_tts()}
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2, size=(3, spec.p, 56, 56), dtype=np.uint8)
    rows_t = torch.from_numpy(OB.pack_rows(bits).view(np.int64)).to(dev)
    with torch.no_grad():
        y = model.forward_from_stem_bits(rows_t).cpu().numpy()
    bt = {}
    ref = OB.forward_from_stem_bits(bits, st, spec, luts, bt)
    for stage in bt:
        if stage == spec.blocks[-1].name:                 # float output of the last block: see "flatten"
            continue
        if stage == "flatten":
            # float32 table entries (float64 values rounded once) averaged in float32, read back
            # from lin1's fp16 x 2 operand format (22 significant bits)
            assert np.abs(model.read_stage("flatten", 3) - bt[stage]).max() <= 5e-7 * max(1.0, np.abs(bt[stage]).max()) + 1e-6
        else:
            assert np.array_equal(model.read_stage(stage, 3), OB.pack_rows(bt[stage])), stage
    print(f"{variant} random bits: |logit| max {np.abs(ref).max():.2f}, |gpu - exact| {np.abs(y - ref).max():.2e}")
    assert np.abs(y - ref).max() <= scaled_tol(ref)


def test_errors_are_loud(small_model, dev):
    import ctypes as C
    model = small_model
    lib = _lib.load()
    plan = model._any_plan()
    x = torch.zeros((1, 3, 224, 224), device=dev)
    out = torch.empty((1, 1000), device=dev)
    st = lib.ttnet_forward(plan.handle, C.c_void_p(x.data_ptr()), plan.max_batch + 1, C.c_void_p(out.data_ptr()), None)
    assert st == -1 and b"max_batch" in lib.ttnet_last_error()
    shape = (C.c_int64 * 1)(7)
    buf = np.zeros(7, dtype=np.float32)
    assert lib.ttnet_plan_set_tensor(plan.handle, b"features.99.weight", buf.ctypes.data_as(C.c_void_p), shape, 1, 0, 0) == -1
    assert b"unexpected key" in lib.ttnet_last_error()
    assert lib.ttnet_plan_set_tensor(plan.handle, b"features.2.weight", buf.ctypes.data_as(C.c_void_p), shape, 1, 0, 0) == -1
    assert b"size mismatch" in lib.ttnet_last_error()
    # a fresh plan refuses to run before finalize, and finalize names the missing key
    desc = _lib.NetDesc(0, 8, 8, 1, 224, 224, 4, 0)
    h = C.c_void_p()
    assert lib.ttnet_plan_create(C.byref(desc), 0, C.byref(h)) == 0
    assert lib.ttnet_forward(h, C.c_void_p(x.data_ptr()), 1, C.c_void_p(out.data_ptr()), None) == -2
    assert lib.ttnet_plan_finalize(h, None) == -2 and b"missing key" in lib.ttnet_last_error()
    lib.ttnet_plan_destroy(h)
    desc = _lib.NetDesc(2, 8, 8, 1, 224, 224, 4, 0)      # full at p = 64: in_channels not divisible by groups
    assert lib.ttnet_plan_create(C.byref(desc), 0, C.byref(h)) == -1
    desc = _lib.NetDesc(0, 8, 5, 1, 224, 224, 4, 0)      # small at p = 40: a fan-in of 20, no truth-table kernel (p = 16..64 in steps of 16 are built)
    assert lib.ttnet_plan_create(C.byref(desc), 0, C.byref(h)) == -4 and b"% 16" in lib.ttnet_last_error()
    desc = _lib.NetDesc(0, 20, 8, 1, 224, 224, 4, 0)     # small at p = 160: beyond the stem kernel's four M-tiles
    assert lib.ttnet_plan_create(C.byref(desc), 0, C.byref(h)) == -4
    desc = _lib.NetDesc(0, 4, 8, 3, 224, 224, 4, 0)      # --layers 3 at p = 32: the stride-1 blocks are built for p = 64
    assert lib.ttnet_plan_create(C.byref(desc), 0, C.byref(h)) == -4
    with pytest.raises(RuntimeError):
        model(torch.zeros((1, 3, 32, 32), device=dev))


def test_state_change_rebuilds_tables(dev):
    spec, st = spec_and_state("small")
    m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    m = m.to(dev).eval().reserve(4)
    x = torch.from_numpy(synth.synth_images(2)).to(dev)
    with torch.no_grad():
        y0 = m(x).clone()
        m.features[9].lin2.bias.add_(1.0)
        y1 = m(x)
    assert (y1 - y0 - 1.0).abs().max().item() < 1e-5


def test_cabi_comm_single_rank(dev):
    """ttnet_comm_* / ttnet_allgather_logits over RCCL with a world of one (the N>1 wiring is
    covered by the gloo tests on CPU; an 8-GPU node is the driver's to launch)."""
    import ctypes as C
    lib = _lib.load()
    uid = (C.c_char * 128)()
    _lib.check(lib.ttnet_comm_unique_id(uid))
    comm = C.c_void_p()
    _lib.check(lib.ttnet_comm_create(uid, 0, 1, 0, C.byref(comm)))
    local = torch.arange(4 * 1000, device=dev, dtype=torch.float32).reshape(4, 1000)
    out = torch.zeros_like(local)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.ttnet_allgather_logits(comm, C.c_void_p(local.data_ptr()), 4, 1000, C.c_void_p(out.data_ptr()),
                                          C.c_void_p(stream)))
    torch.cuda.synchronize()
    assert torch.equal(out, local)
    lib.ttnet_comm_destroy(comm)


def _check_geometry_against_the_oracle(dev, nfilter, tfilter, layers, golden=None, variant="small"):
    """Three independent checks of a (p, --layers) geometry:
      1. every GPU-built truth table equals the float64 oracle table (OB.build_lut), the float table of the last
         block included, ALL of its groups (in chunks of 8, built on a thread pool);
      2. the gate path on the GPU's stem bits is bit-identical to the bit oracle RUN ON THE ORACLE'S
         OWN TABLES at every stage, and the logits are within the tolerance of the exact head;
      3. golden = a fixture of the imported reference's own per-stage hashes and logits (oracle/gen_golden.py):
         every block output must be hash-identical to the reference capture, top-1 equal, logits within tolerance."""
    from argparse import Namespace
    from concurrent.futures import ThreadPoolExecutor      # (numpy / scipy release the GIL: ~100 s of table building otherwise)
    from scale_imagenet_amd.spec import make_spec
    tag = f"{variant} p={nfilter * tfilter} --layers {layers}"
    spec = make_spec(variant, nfilter, tfilter, layers)
    st = synth.synth_state_dict(spec, calibrated=False)
    m = CLASSES[variant](Namespace(nfilter=nfilter, tfilter=tfilter, layers=layers, groups=[1, None, 4, None]))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to(dev).eval().reserve(4)
    x = synth.synth_images(3)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    stem_rows = m.read_stage("features.3", 3)
    bits = OB.unpack_rows(stem_rows, 56)
    # 0. the stem (one, two or four M-tiles by p): bits of the float oracle's stem except at near ties
    taps = {}
    OF.forward(torch.from_numpy(x[:2]), OF.to_torch_state(st), spec, taps)
    ref_bits, pre = taps["features.3"].numpy().astype(np.uint8), taps["stem.pre"].numpy()
    d = np.argwhere(bits[:2] != ref_bits)
    assert all(abs(pre[tuple(i)]) < OB.NEAR_TIE for i in d), f"{tag}: stem bits differ away from a near tie"
    # 1. tables: GPU float64 == numpy float64 (differences only where the oracle itself sees a near tie)
    luts = {}
    with ThreadPoolExecutor(max_workers=8) as ex:
        built = dict(zip([b.name for b in spec.block_tts() if not b.last],
                         ex.map(lambda b: OB.build_lut(st, b), [b for b in spec.block_tts() if not b.last])))
        for b in spec.block_tts():
            tab = m.get_table(b.name)
            if b.last:
                chunks = [list(range(g0, min(b.groups, g0 + 8))) for g0 in range(0, b.groups, 8)]
                for gl, (ref, _) in zip(chunks, ex.map(lambda gl: OB.build_lut(st, b, groups=gl), chunks)):
                    assert np.abs(tab[gl] - ref).max() <= 1e-6, (b.name, gl)
                luts[b.name] = tab
                continue
            ref, near = built[b.name]
            d = np.argwhere(ref != tab)
            assert near[tuple(d.T)].all(), f"{tag} {b.name}: table differs outside the near-tie set"
            luts[b.name] = ref
            if len(d):                                  # run the bit oracle on exactly what the GPU looks up
                luts[b.name] = tab
                print(f"{tag} {b.name}: {len(d)} near-tie entries differ from numpy float64")
    # 2. integer path vs the bit oracle on the oracle's tables
    bt = {}
    ref = OB.forward_from_stem_bits(bits, st, spec, luts, bt)
    for stage, want in bt.items():
        if stage == "flatten" or stage == spec.blocks[-1].name:
            continue
        assert np.array_equal(m.read_stage(stage, 3), OB.pack_rows(want)), (tag, stage)
    assert y.shape == (3, 1000)
    print(f"{tag}: |logit| max {np.abs(ref).max():.1f} (uncalibrated head), |gpu - exact| {np.abs(y - ref).max():.2e}")
    assert np.abs(y - ref).max() <= scaled_tol(ref), (tag, np.abs(y - ref).max())
    # 3. the reference's own capture (oracle/gen_golden.py depth / width)
    if golden is not None:
        with np.load(os.path.join(GOLD, golden)) as z:
            want, names, shas = z["logits"], [str(v) for v in z["stage_names"]], [str(v) for v in z["stage_sha"]]
        k = want.shape[0]
        assert sha(stem_rows[:k]) == shas[names.index("features.3")], f"{tag}: stem bits differ from the reference capture"
        differ = [nm for nm, hs in zip(names, shas) if nm != "features.3" and sha(m.read_stage(nm, 3)[:k]) != hs]
        # (round 2 only printed this: every block output of the fixture's images is hash-identical to the reference, and a
        # float32-vs-float64 near tie of a table entry these images reach would have to be listed here to be excused)
        assert not differ, f"{tag}: stages {differ} differ from the reference capture"
        assert np.array_equal(y[:k].argmax(1), want.argmax(1))
        # |hip - exact| <= 1e-5 scaled to these logits' magnitude, plus the committed distance of the reference's own float32
        # head from the exact one on this very fixture (tests/golden/ref_spread.json "geometries": 0.8e-5 .. 1.5e-5)
        assert np.abs(y[:k] - want).max() <= scaled_tol(want) + GEOMETRY_SPREAD[golden]["ref_vs_exact"]
        print(f"{tag}: {len(names) - 1} block outputs hash-identical to the reference capture, |gpu - reference| "
              f"{np.abs(y[:k] - want).max():.2e} on {k} images")


@pytest.mark.parametrize("layers", [0, 2, 3, 4])
def test_other_depths_against_the_oracle(dev, layers):
    """--layers 0 / 2 (two and four stride-2 blocks) and 3 / 4 (a stride-1 first block, and a second
    one at 29x29; TT_general_imagenet_v2_small.py:172-181); for --layers 3 / 4 also against
    tests/golden/ref_small_l<k>.npz (the imported reference's own per-stage hashes and logits)."""
    _check_geometry_against_the_oracle(dev, 8, 8, layers, f"ref_small_l{layers}.npz" if layers >= 3 else None)


@pytest.mark.parametrize("nfilter,tfilter", [(4, 8), (6, 8), (2, 8), (12, 8), (16, 8)])
def test_other_widths_against_the_oracle(dev, nfilter, tfilter):
    """p = nfilter * tfilter other than main.py's default 64 (TT_general_imagenet_v2_small.py:165-167): 32, 48, 16, 96, 128.
    Every p with p % 16 == 0 and p <= 128 keeps the fan-in of 16 and is built (one, two or four M-tiles in the stem
    kernel); p = 32 is also pinned to the imported reference (tests/golden/ref_small_p32.npz)."""
    p = nfilter * tfilter
    _check_geometry_against_the_oracle(dev, nfilter, tfilter, 1, "ref_small_p32.npz" if p == 32 else None)


@pytest.mark.parametrize("nfilter,tfilter", [(4, 8), (6, 8)])
def test_xsmall_other_widths_against_the_oracle(dev, nfilter, tfilter):
    """The x-small variant (fan-in 4) at p = 32 and p = 48 (TT_general_imagenet_v2_xsmall.py: groups = C / 4)."""
    _check_geometry_against_the_oracle(dev, nfilter, tfilter, 1, None, "xsmall")


@pytest.mark.parametrize("layers", [0, 2])
def test_xsmall_other_depths_against_the_oracle(dev, layers):
    """The x-small variant at --layers 0 / 2 (TT_general_imagenet_v2_xsmall.py:172-177: two and four stride-2 blocks)."""
    _check_geometry_against_the_oracle(dev, 8, 8, layers, None, "xsmall")


def test_full_depth_0_against_the_float_oracle(dev):
    """The full (fan-in 30) variant at --layers 0 (TT_general_imagenet_v2.py:161-162: two blocks): no tables to compare, so
    every stage against the float oracle's taps (bit-identical away from near ties of the float32 oracle) and the logits
    against the float64 head on the GPU's own features."""
    from argparse import Namespace
    from scale_imagenet_amd.spec import make_spec
    spec = make_spec("full", 6, 10, 0)
    st = synth.synth_state_dict(spec, calibrated=False)
    m = ttnet.TT_vf_19lv3_imgnet(Namespace(nfilter=6, tfilter=10, layers=0, groups=[1, None, 4, None]))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to(dev).eval().reserve(4)
    x = synth.synth_images(3)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    taps = {}
    OF.forward(torch.from_numpy(x), OF.to_torch_state(st), spec, taps)
    for stage in ["features.3"] + [b.name for b in spec.blocks[:-1]]:
        got = OB.unpack_rows(m.read_stage(stage, 3), _stage_width(spec, stage))
        want = taps[stage].numpy().astype(np.uint8)
        bad = int((got != want).sum())
        print(f"full --layers 0 {stage}: {bad} of {want.size} bits differ from the float32 oracle")
        assert bad <= 2, (stage, bad)                # (a float32 near tie of the oracle's own; none seen)
    feat = m.read_stage("flatten", 3)
    exact = OB.head64(feat, st, f"features.{4 + len(spec.blocks) + 2}")
    assert np.abs(feat - taps["flatten"].numpy()).max() <= 2e-5 * max(1.0, float(np.abs(feat).max()))
    assert np.abs(y - exact).max() <= scaled_tol(exact), np.abs(y - exact).max()


def test_unbuilt_widths_are_refused_loudly(dev):
    """What the reference constructs and this build does not (ttnet.h): p > 64, and p with another fan-in than 16."""
    from argparse import Namespace
    for nf, tf in ((20, 8), (5, 8)):                 # p = 160 (> 128), p = 40 (fan-in 20)
        try:
            m = ttnet.TT_vf_19lv3_imgnet_small(Namespace(nfilter=nf, tfilter=tf, layers=1, groups=[1, None, 4, None]))
        except (ValueError, RuntimeError, AssertionError):
            continue                                 # (the spec itself may refuse the geometry)
        m = m.to(dev).eval()
        with pytest.raises(_lib.TTNetError) as ei, torch.no_grad():
            m(torch.zeros((1, 3, 224, 224), device=dev))
        assert ei.value.status == -4                 # TTNET_E_UNSUPPORTED


def test_reload_after_capture_uses_the_new_weights(dev):
    """A captured hipGraph bakes in by-value arguments derived from the weights (lin2's 1/prescale):
    three forwards (-> capture), then load_state_dict with lin2.weight x 4 (its prescale crosses two
    powers of two) and a changed BatchNorm: the next forward must equal a fresh model's, and the plan
    must have dropped its graphs and go on to capture and replay new ones."""
    spec, st = spec_and_state("small")
    m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    m = m.to(dev).eval().reserve(8)
    x = torch.from_numpy(synth.synth_images(8)).to(dev)
    with torch.no_grad():
        for _ in range(4):
            y_old = m(x).clone()
    plan = m._any_plan()
    assert plan.query("graphs_enabled") == 1 and plan.query("graph_replays") >= 1 and plan.query("graphs_cached") == 1
    st2 = {k: v.copy() for k, v in st.items()}
    st2["features.9.lin2.weight"] *= 4.0
    st2["features.9.BN2.weight"] = (st2["features.9.BN2.weight"] * 0.5).astype(np.float32)
    m.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in st2.items()})     # a DataParallel checkpoint
    with torch.no_grad():
        y_new = m(x).clone()
    assert plan.query("graphs_cached") == 0 and plan.query("graph_drops") >= 1
    fresh = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
    fresh.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st2.items()})
    fresh = fresh.to(dev).eval().reserve(8)
    with torch.no_grad():
        want = fresh(x).clone()
        replays = plan.query("graph_replays")
        for _ in range(4):
            y_again = m(x).clone()
    assert torch.equal(y_new, want) and torch.equal(y_again, want)
    assert not torch.equal(y_new, y_old)
    assert plan.query("graph_replays") > replays and plan.query("graphs_cached") == 1


def test_graph_cache_eviction_and_pointer_patching(dev):
    """Round 1 saw one abort in this area ('The graph-replay abort'): the kernel-parameter
    arrays handed to hipGraphExecKernelNodeSetParams must be storage the plan owns, also after the
    GraphEntry has been moved into the cache map.  Ten batch sizes overflow the 8-entry cache (entries
    are evicted only after a device synchronisation), every replay gets fresh input / output buffers,
    and every result must equal the plain launches."""
    spec, st = spec_and_state("small")
    m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    m = m.to(dev).eval().reserve(16)
    xs = torch.from_numpy(synth.synth_images(16)).to(dev)
    m.set_profiling(True)                           # plain launches
    with torch.no_grad():
        want = m(xs).clone()
    m.set_profiling(False)
    plan = m._any_plan()
    sizes = [3, 4, 5, 6, 7, 9, 10, 11, 12, 13]
    with torch.no_grad():
        for n in sizes:
            for rep in range(4):                    # two plain calls, the capture, one replay with new buffers
                y = m(xs[:n].clone())
            assert torch.equal(y, want[:n]), n
        for n in sizes:                             # evicted sizes are captured again, cached ones replayed
            for rep in range(3):
                y = m(xs[:n].clone())
            assert torch.equal(y, want[:n]), n
    torch.cuda.synchronize()
    assert plan.query("graphs_enabled") == 1, _lib.load().ttnet_last_error()
    assert plan.query("graph_drops") >= 2 and plan.query("graphs_cached") <= 8
    assert plan.query("graph_captures") >= len(sizes)


def test_dataparallel_wrapper_as_main_py_builds_it(dev):
    """main.py:192 wraps the model in nn.DataParallel and :222 loads a `module.`-prefixed checkpoint
    INTO THE WRAPPER; :251 / :261 then call wrapper.eval() and wrapper(inputs).  With one visible
    device DataParallel calls the module directly (no replicate)."""
    spec, st = spec_and_state("small")
    g = golden_npz("small")
    net = torch.nn.DataParallel(ttnet.TT_vf_19lv3_imgnet_small(args_for("small")), device_ids=[0]).cuda()
    ckpt = {"model_state_dict": {"module." + k: torch.from_numpy(v.copy()) for k, v in st.items()}}
    assert list(net.state_dict().keys()) == list(ckpt["model_state_dict"].keys())
    net.load_state_dict(ckpt["model_state_dict"])           # strict, as main.py:222
    net.eval()
    n = int(g["n_images"])
    x = torch.from_numpy(synth.synth_images(n))
    with torch.no_grad():
        y = net(x.cuda(non_blocking=True))
        plain = _model("small", dev)(x.to(dev))
    assert y.device.type == "cuda" and y.shape == (n, 1000)
    assert torch.equal(y, plain)
    assert (y.argmax(1).cpu().numpy() == g["argmax"]).sum() >= n - 2     # (near-tie images: test_end_to_end_forward)


def test_values_outside_the_split_range_are_loud(dev):
    """The float stages carry f32 operands as two fp16 terms after a x16 prescale: |v| < 4094
    (ttnet.h).  The reference has no such limit, so an input beyond it must not pass silently: the
    kernel raises the plan's sticky flag and the next call fails with TTNET_E_RANGE until the flag
    has been read."""
    spec, st = spec_and_state("small")
    m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    m = m.to(dev).eval().reserve(4)
    x = torch.from_numpy(synth.synth_images(2)).to(dev)
    with torch.no_grad():
        y0 = m(x).clone()
        plan = m._any_plan()
        assert plan.query("range_overflow") == 0
        bad = x.clone()
        bad[1, 2, 100:104, 50:54] = 30000.0          # pooled value 30000 -> 480000 after the prescale
        m(bad)                                       # asynchronous: returns before the kernel has run
        torch.cuda.synchronize()
        with pytest.raises(_lib.TTNetError) as ei:
            m(x)
        assert ei.value.status == -6 and b"range" in _lib.load().ttnet_last_error()
        assert plan.query("range_overflow") == 1     # read and clear
        assert plan.query("range_overflow") == 0
        assert torch.equal(m(x), y0)
        nan = x.clone()
        nan[0, 0, 0, 0] = float("nan")
        m(nan)
        torch.cuda.synchronize()
        with pytest.raises(_lib.TTNetError):
            m.read_stage("features.3", 2)
        assert plan.query("range_overflow") == 1
    # a classifier whose BatchNorm1d blows the polynomial's input up leaves the range inside the head
    st2 = {k: v.copy() for k, v in st.items()}
    st2["features.9.BN2.weight"] = (st2["features.9.BN2.weight"] * 1e4).astype(np.float32)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st2.items()})
    with torch.no_grad():
        m(x)
        torch.cuda.synchronize()
        assert plan.query("range_overflow") == 1
    # an overflow in the LAST batch of an evaluation: no further forward would report it, evaluate() must
    from scale_imagenet_amd import evaluate as E
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    tg = torch.from_numpy(synth.synth_targets(2))
    E.evaluate(m, [(x.cpu(), tg), (x.cpu(), tg)], dev)                      # clean run: no error
    with pytest.raises(RuntimeError, match="range of the split operands"):
        E.evaluate(m, [(x.cpu(), tg), (bad.cpu(), tg)], dev)
    with torch.no_grad():
        assert torch.equal(m(x), y0)                                         # the flag was read and cleared


def test_valexnet_config5(dev):
    """BASELINE config 5: TT_FHE_XSMALL_vAlexnet, CIFAR 32x32 (non-square 3x2 / 2x3 windows at
    stride 1, 8-channel 1x1 groups, no convf).  The stem weights are synthetic like the rest: the
    reference's pretrained-VGG fetch is unavailable offline (gen_golden.py uses a local stand-in)."""
    g, j = golden_npz("valexnet"), golden_json("valexnet")
    spec, st = spec_and_state("valexnet")
    n = int(g["n_images"])
    m = ttnet.TT_FHE_XSMALL_vAlexnet(args_for("valexnet"))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m = m.to(dev).eval().reserve(256)
    x = synth.synth_images(n, hw=(32, 32))
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert y.shape == (n, 10)
    # tables: GPU float64 == numpy float64 == the reference's own float32 module (no near-tie flips)
    for b in spec.block_tts():
        tab = m.get_table(b.name)
        assert sha(np.packbits(tab, axis=1, bitorder="little")) == j["luts"][b.name]["f64_sha256"], b.name
        assert j["luts"][b.name]["f64_sha256"] == j["luts"][b.name]["ref_sha256"]
    # stem bits: equal to the reference except at near ties of the float stem
    taps = {}
    y_ref = OF.forward_valexnet(torch.from_numpy(x), OF.to_torch_state(st), spec, taps).numpy()
    stem = OB.unpack_rows(m.read_stage("features.4", n), 10)
    d = np.argwhere(stem != taps["features.4"].numpy().astype(np.uint8))
    pre = taps["stem.pre"].numpy()
    assert all(abs(pre[tuple(i)]) < OB.NEAR_TIE for i in d)
    # block + head from the reference's stem bits: every stage hash, logits vs the exact head
    rows = OB.pack_rows(taps["features.4"].numpy().astype(np.uint8))
    with torch.no_grad():
        y2 = m.forward_from_stem_bits(torch.from_numpy(rows.view(np.int64)).to(dev)).cpu().numpy()
    got = m.read_stage("features.5", n)
    assert [sha(got[i]) for i in range(n)] == j["stages"]["features.5"]["per_image_sha256"]
    luts = {b.name: m.get_table(b.name) for b in spec.block_tts()}
    _, exact = OB.valexnet_from_stem_bits(taps["features.4"].numpy().astype(np.uint8), st, spec, luts)
    ref_dev = float(np.abs(g["logits"] - exact).max())
    assert np.abs(y2 - exact).max() <= LOGIT_TOL
    assert np.abs(y2 - g["logits"]).max() <= LOGIT_TOL + ref_dev
    assert np.array_equal(y2.argmax(1), g["argmax"])
    if len(d) == 0:
        assert np.abs(y - y2).max() == 0.0
    assert np.array_equal(m.read_stage("flatten", 2)[0], taps["features.5"].numpy().reshape(n, -1)[0])


def test_full_fast_path_is_the_float64_path(dev, monkeypatch):
    """Full variant (fan-in 30): the grouped 1x1 and depthwise blocks are evaluated in split-fp16 / float32 and
    only the outputs inside the evaluation's error bound are redone in float64 (gate_full.hip).  The emitted
    bits must be exactly those of the all-float64 path (TTNET_FULL_EXACT=1) -- on every stage of a batch large
    enough that thousands of outputs go through the float64 pass -- and a share of the outputs, but not most
    of them, must have been listed."""
    spec, st = spec_and_state("full")
    m = ttnet.TT_vf_19lv3_imgnet(args_for("full"))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    m = m.to(dev).eval().reserve(24)
    x = torch.from_numpy(synth.synth_images(24)).to(dev)
    stages = [b.name for b in spec.blocks[:-1]]

    def run():
        with torch.no_grad():
            y = m(x).cpu().numpy()
        return y, {s: m.read_stage(s, 24).copy() for s in stages}

    monkeypatch.delenv("TTNET_FULL_EXACT", raising=False)
    plan_before = None
    y_fast, st_fast = run()
    plan_before = m._any_plan()
    listed_pw, listed_dw = plan_before.query("full_listed_pw"), plan_before.query("full_listed_dw")
    monkeypatch.setenv("TTNET_FULL_EXACT", "1")
    y_exact, st_exact = run()
    assert plan_before.query("full_listed_pw") == listed_pw          # (the exact path lists nothing)
    for s in stages:
        assert np.array_equal(st_fast[s], st_exact[s]), s
    # the last block's features are float32 in the fast path, float64 rounded once in the exact one
    assert np.abs(y_fast - y_exact).max() <= LOGIT_TOL
    # the bound's margin: scaled down a hundredfold it still lists every output whose sign float32 gets wrong (none
    # differs; tools/full_tau_margin.py: the same down to 1/300 on 64 images) -- the shipped bound is the worst-case one
    monkeypatch.delenv("TTNET_FULL_EXACT", raising=False)
    monkeypatch.setenv("TTNET_FULL_TAU_SCALE", "0.01")
    _, st_tight = run()
    monkeypatch.delenv("TTNET_FULL_TAU_SCALE", raising=False)
    for s in stages:
        assert np.array_equal(st_tight[s], st_exact[s]), s
    pixel_groups = 24 * sum(b.conv3.groups * b.in_hw[0] * b.in_hw[1] + (0 if b.last else b.convf.groups * b.out_hw[0] * b.out_hw[1])
                            for b in spec.blocks)
    print(f"full: {listed_pw} of {pixel_groups} (pixel, group) pairs and {listed_dw} depthwise outputs redone in float64")
    assert 0 < listed_pw < pixel_groups // 8 and 0 < listed_dw


def test_unaligned_input_view(small_model, dev):
    """The stem reads the input with 16-byte loads (ttnet.h); a view that starts 4 bytes into a buffer still works through
    the module (it is copied), and the raw C ABI refuses it loudly."""
    import ctypes as C
    x = torch.from_numpy(synth.synth_images(2)).to(dev)
    flat = torch.empty(x.numel() + 1, device=dev)
    flat[1:] = x.reshape(-1)
    view = flat[1:].view_as(x)
    assert view.data_ptr() % 16 == 4
    with torch.no_grad():
        assert torch.equal(small_model(view), small_model(x))
    plan = small_model._any_plan()
    out = torch.empty((2, 1000), device=dev)
    lib = _lib.load()
    st = lib.ttnet_forward(plan.handle, C.c_void_p(view.data_ptr()), 2, C.c_void_p(out.data_ptr()), None)
    assert st == -1 and b"aligned" in lib.ttnet_last_error()
    # ... also once a graph for this batch size is cached (round 2's check sat in the plain path only: a replay took
    # any pointer), and a refused call must not turn graph replay off for the plan
    for _ in range(5):
        assert lib.ttnet_forward(plan.handle, C.c_void_p(x.data_ptr()), 2, C.c_void_p(out.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert plan.query("graphs_cached") >= 1 and plan.query("graphs_enabled") == 1
    replays = plan.query("graph_replays")
    st = lib.ttnet_forward(plan.handle, C.c_void_p(view.data_ptr()), 2, C.c_void_p(out.data_ptr()), None)
    assert st == -1 and b"aligned" in lib.ttnet_last_error()
    assert plan.query("graph_replays") == replays and plan.query("graphs_enabled") == 1
    assert lib.ttnet_forward(plan.handle, C.c_void_p(x.data_ptr()), 2, C.c_void_p(out.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert plan.query("graph_replays") == replays + 1
    with torch.no_grad():
        assert torch.equal(out, small_model(x))
