"""ORACLE tooling -- golden-vector generator.  Runs ONLY in the build container.

Imports the upstream reference from /root/reference (read-only; Python, importable with
empty ``torchvision`` stubs), loads the build's synthetic state_dict
into the reference's own model classes and records what the REFERENCE computes:

  tests/golden/state_layout_<variant>.json   state_dict keys / shapes / dtypes, in order
  tests/golden/ref_<variant>.npz             logits, CE loss, top-1/top-5, per-stage
                                             SHA-256 of the packed activation bits, and
                                             (first images only) the packed bits themselves
  tests/golden/ref_luts_<variant>.json       SHA-256 of every truth table as exported by
                                             the reference's own enumerator
                                             (Block_TT.get_TT_block_all_filter,
                                             models/TT_FHE_SMALL.py:322-342) and the list
                                             of entries where it differs from the float64
                                             oracle table (must all be near ties)
  tests/golden/ref_spread.json               the reference's own logit spread (threads, batch split)
                                             and its distance from the float64 head (gen_spread)
  scale_imagenet_amd/data/synth_head_bn_<variant>.npz
                                             calibrated BatchNorm1d statistics of the
                                             synthetic classifier (data, see synth.py)

It also asserts that oracle/ttnet_float.py reproduces the reference bit for bit, which is
what pins the oracle.  Nothing of the reference travels: only arrays and hashes are
written.  Usage:  python oracle/gen_golden.py [small xsmall full]
"""
from __future__ import annotations

import contextlib
import hashlib
import io
import json
import os
import sys
import types

sys.dont_write_bytecode = True          # /root/reference is writable by root: leave no __pycache__
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from scale_imagenet_amd.spec import make_spec, state_dict_layout
from scale_imagenet_amd import synth
from oracle import ttnet_float as OF
from oracle import ttnet_bits as OB

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "scale_imagenet_amd", "data")
REF = "/root/reference"

VARIANT_ARGS = {
    "small": dict(nfilter=8, tfilter=8, layers=1),
    "xsmall": dict(nfilter=8, tfilter=8, layers=1),
    "full": dict(nfilter=6, tfilter=10, layers=1),     # p=64 does not construct
}
BATCH = {"small": 8, "xsmall": 8, "full": 4}
DUMP_IMAGES = 2           # images whose packed stage bits are stored in full
CALIB_IMAGES = 64
CALIB_FIRST = 100000      # calibration images are disjoint from every test batch


def import_reference(variant: str, layers: int = None, nfilter: int = None, tfilter: int = None):
    for name in ("torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.datasets"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision.transforms"].Normalize = object
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from argparse import Namespace
    with contextlib.redirect_stdout(io.StringIO()):
        if variant == "small":
            from models.TT_general_imagenet_v2_small import TT_vf_19lv3_imgnet_small as M
        elif variant == "xsmall":
            from models.TT_general_imagenet_v2_xsmall import TT_vf_19lv3_imgnet_xsmall as M
        else:
            from models.TT_general_imagenet_v2 import TT_vf_19lv3_imgnet as M
        a = dict(VARIANT_ARGS[variant])
        if layers is not None:
            a["layers"] = layers
        if nfilter is not None:
            a["nfilter"], a["tfilter"] = nfilter, tfilter
        m = M(Namespace(groups=[1, None, 4, None], **a)).eval()
    return m


def gen_valexnet():
    """Config 5: TT_FHE_XSMALL_vAlexnet (CIFAR 32x32).  The reference constructor fetches
    pretrained VGG16 / DenseNet weights (models/TT_FHE_XSMALL_vAlexnet.py:594-597); there is no
    network here, so ``torchvision.models`` is a LOCAL un-pretrained stand-in that only
    reproduces the ``features[0:2]`` topology (Conv2d(3,64,3,padding=1) with bias, ReLU) --
    the stem weights then come from the synthetic state_dict like every other tensor."""
    import torch.nn as nn
    from scale_imagenet_amd.spec import VAlexSpec, valexnet_layout
    for name in ("torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.datasets",
                 "torchvision.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tv = sys.modules["torchvision"]
    tv.transforms, tv.utils = sys.modules["torchvision.transforms"], sys.modules["torchvision.utils"]
    tv.datasets, tv.models = sys.modules["torchvision.datasets"], sys.modules["torchvision.models"]
    sys.modules["torchvision.transforms"].Normalize = object

    class _VGGStandIn(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(nn.Conv2d(3, 64, 3, padding=1), nn.ReLU(inplace=True))
    tv.models.vgg16 = lambda pretrained=False: _VGGStandIn()
    tv.models.densenet121 = lambda pretrained=False: None
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from argparse import Namespace
    with contextlib.redirect_stdout(io.StringIO()):
        from models.TT_FHE_XSMALL_vAlexnet import TT_FHE_XSMALL_vAlexnet
        m = TT_FHE_XSMALL_vAlexnet(Namespace(nfilter=8, tfilter=8, layers=1, groups=[1, None, 4, None])).eval()
    spec = VAlexSpec()
    layout = valexnet_layout(spec)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(layout.keys())
    for k, v in ref_sd.items():
        assert tuple(v.shape) == layout[k][0] and str(v.dtype) == "torch." + layout[k][1], k
    with open(os.path.join(GOLD, "state_layout_valexnet.json"), "w") as f:
        json.dump({"variant": "valexnet", "n_params": int(sum(p.numel() for p in m.parameters())),
                   "keys": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in ref_sd.items()]},
                  f, indent=0)
    # calibrate the head BatchNorm1d on 64 synthetic images
    st0 = synth.synth_state_dict(spec, calibrated=False)
    sd0 = OF.to_torch_state(st0)
    xc = torch.from_numpy(synth.synth_images(CALIB_IMAGES, first=CALIB_FIRST, hw=(32, 32)))
    tp = {}
    OF.forward_valexnet(xc, sd0, spec, tp)
    z = tp["features.5"].reshape(CALIB_IMAGES, -1).double() @ sd0["features.7.lin1.weight"].double().t()
    os.makedirs(DATA, exist_ok=True)
    np.savez(os.path.join(DATA, "synth_head_bn_valexnet.npz"), running_mean=z.mean(0).float().numpy(),
             running_var=z.var(0, unbiased=False).float().numpy())
    st = synth.synth_state_dict(spec)
    sd = OF.to_torch_state(st)
    m.load_state_dict(sd, strict=True)
    n = 16
    x = torch.from_numpy(synth.synth_images(n, hw=(32, 32)))
    ref_taps = {}
    hooks = [m.features[3].register_forward_hook(lambda mod, i, o: ref_taps.__setitem__("stem.pre", o.detach().clone())),
             m.features[4].register_forward_hook(lambda mod, i, o: ref_taps.__setitem__("features.4", o.detach().clone())),
             m.features[5].register_forward_hook(lambda mod, i, o: ref_taps.__setitem__("features.5", o.detach().clone()))]
    with torch.no_grad():
        y_ref = m(x)
    for h in hooks:
        h.remove()
    taps = {}
    y = OF.forward_valexnet(x, sd, spec, taps)
    assert torch.equal(y, y_ref), "valexnet: oracle logits differ from the reference"
    for k in ("stem.pre", "features.4", "features.5"):
        assert torch.equal(taps[k], ref_taps[k]), k
    print(f"[valexnet] oracle/ttnet_float.py == reference on {n} images (all stages, bit for bit)")
    tgt = torch.from_numpy(synth.synth_targets(n, n_classes=10))
    out = {"logits": y_ref.numpy(), "argmax": y_ref.argmax(1).numpy(), "n_images": np.int64(n),
           "loss": np.float64(torch.nn.functional.cross_entropy(y_ref, tgt).item())}
    stages = {}
    for k in ("features.4", "features.5"):
        rows = OB.pack_rows(taps[k].numpy().astype(np.uint8))
        stages[k] = {"shape": list(taps[k].shape), "rows_sha256": sha(rows),
                     "per_image_sha256": [sha(rows[i]) for i in range(n)]}
        out["rows:" + k] = rows[:DUMP_IMAGES]
    pre = ref_taps["stem.pre"].numpy()
    near = np.argwhere(np.abs(pre) < OB.NEAR_TIE)
    out["stem_near_tie_idx"] = near.astype(np.int32)
    np.savez_compressed(os.path.join(GOLD, "ref_valexnet.npz"), **out)
    luts = {}
    for b, modname in ((spec.conv1, "Block_conv1"), (spec.conv2, "Block_conv2"), (spec.conv3, "Block_conv3")):
        mod = getattr(m.features[5], modname)
        with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
            pats = torch.from_numpy(OB.enumerate_patterns(b.fan_in_bits).astype(np.float32))
            xin = pats.reshape(-1, b.cin_g, b.kh, b.kw).repeat(1, b.groups, 1, 1)
            if b.padding:      # feed the pattern as the padded window: crop the centre of the padded output
                res = mod.forward(xin, compute_final_mask_noise=False).numpy()
                res = res[:, :, b.padding * 1, b.padding * 1] if res.shape[-1] > 1 else res
            else:
                res = mod.forward(xin, compute_final_mask_noise=False).numpy()
        res = res.reshape(res.shape[0], b.groups, b.cout_g).transpose(1, 0, 2).astype(np.uint8)
        mine, near_tab = OB.build_lut(st, b)
        diff = np.argwhere(res != mine)
        assert near_tab[tuple(diff.T)].all(), f"{b.name}: table mismatch outside the near-tie set"
        luts[b.name] = {"kind": "bits", "shape": list(res.shape),
                        "ref_sha256": sha(np.packbits(res, axis=1, bitorder="little")),
                        "f64_sha256": sha(np.packbits(mine, axis=1, bitorder="little")),
                        "near_ties": int(near_tab.sum()), "ref_differs_from_f64_at": diff.tolist()}
        print(f"[valexnet] {b.name}: flips vs f64: {len(diff)}")
    with open(os.path.join(GOLD, "ref_luts_valexnet.json"), "w") as f:
        json.dump({"variant": "valexnet", "stages": stages, "luts": luts, "near_tie_threshold": OB.NEAR_TIE,
                   "torch": torch.__version__, "batch": n}, f, indent=0)
    print(f"[valexnet] logits range [{y_ref.min().item():.3f}, {y_ref.max().item():.3f}] argmax {y_ref.argmax(1).tolist()}")


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def calibrate_head(variant: str, spec):
    """BatchNorm1d running stats = statistics of lin1 outputs over CALIB_IMAGES synthetic
    images, as training would have left them; without this the random classifier's output is
    one constant class for every image."""
    st = synth.synth_state_dict(spec, calibrated=False)
    sd = OF.to_torch_state(st)
    head = f"features.{4 + len(spec.blocks) + 2}"
    zs = []
    for i in range(0, CALIB_IMAGES, 16):
        x = torch.from_numpy(synth.synth_images(16, first=CALIB_FIRST + i, hw=spec.image_hw))
        taps = {}
        OF.forward(x, sd, spec, taps)
        zs.append(taps["flatten"].double() @ sd[f"{head}.lin1.weight"].double().t())
    z = torch.cat(zs)
    mean = z.mean(0).float().numpy()
    var = z.var(0, unbiased=False).float().numpy()
    os.makedirs(DATA, exist_ok=True)
    np.savez(os.path.join(DATA, f"synth_head_bn_{variant}.npz"), running_mean=mean, running_var=var)
    print(f"[{variant}] calibrated head BN: mean|mean| {np.abs(mean).mean():.4f}  mean var {var.mean():.5f}")


def reference_truth_table(block_tt_module, b) -> np.ndarray:
    """The reference's own arithmetic on every input pattern; returns [G, 2^n, cout_g]
    (uint8 bits, or float32 for a ``last`` block).

    Depthwise blocks go through the reference's exporter ``get_TT_block_all_filter``
    (models/TT_FHE_SMALL.py:322-342).  That exporter only works for one input channel per
    group (it appends ``in_planes - cin_g`` copies of the whole cin_g-channel pattern, so a
    grouped 1x1 block gets 244 instead of 64 channels and conv1 raises); for those blocks
    the same enumeration (MSB first over (c, kh, kw), :330-334) is fed to the reference
    module's own ``forward``."""
    with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
        if b.cin_g == 1:
            res = np.asarray(block_tt_module.get_TT_block_all_filter("cpu", 0, 0))
        else:
            pats = torch.from_numpy(OB.enumerate_patterns(b.fan_in_bits).astype(np.float32))
            x = pats.reshape(-1, b.cin_g, b.kh, b.kw).repeat(1, b.groups, 1, 1)
            res = block_tt_module.forward(x, compute_final_mask_noise=False).numpy()
    if res.ndim == 4 and res.shape[-1] > 1:
        # For a padded block the exporter pads the pattern and forward() pads it again
        # (:340-342 then :311-312), so the result is a small map whose entry at offset
        # 2*pad/stride is the window that covers exactly the enumerated pattern.
        assert (2 * b.padding) % b.stride == 0
        off = 2 * b.padding // b.stride
        res = res[:, :, off, off]
    res = res.reshape(res.shape[0], b.groups, b.cout_g).transpose(1, 0, 2)
    return res.astype(np.float32) if b.last else res.astype(np.uint8)


def gen_export_golden(n_filters: int = 12):
    """tests/golden/ref_export_xsmall.json: what the reference's own exporter
    (Block_TT.get_TT_block_1filter -> for_1_filter / save_cnf_dnf, models/TT_FHE_SMALL.py:344-431)
    writes for the first filters of features.4.Block_conv1 of the x-small model (n = 4 inputs,
    one of the sizes its get_expresion_methode1 handles).  The exporter's first step
    (get_TT_block_all_filter) returns a small map for a padded block; its centre entry is the
    window that covers the enumerated pattern (see reference_truth_table), and that [2^n, C]
    table is handed to the reference's own per-filter code unchanged."""
    import tempfile
    variant = "xsmall"
    spec = make_spec(variant, **{k: v for k, v in VARIANT_ARGS[variant].items() if k != "layers"})
    st = synth.synth_state_dict(spec)
    m = import_reference(variant)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
    m.eval()
    blk = spec.blocks[0].conv1
    mod = m.features[4].Block_conv1
    tab = reference_truth_table(mod, blk)                 # [G, 2^n, 1] bits; also sets mod.df / mod.n
    mod.res_numpy = tab[:, :, 0].T.astype(np.float32)     # [2^n, C] as the exporter keeps it
    out = {"block": blk.name, "n": int(blk.fan_in_bits), "blockici": 0, "sousblockici": 1, "filters": {}}
    with tempfile.TemporaryDirectory() as d:
        for f in range(n_filters):
            before = set(os.listdir(d))
            with contextlib.redirect_stdout(io.StringIO()):
                mod.blockici, mod.sousblockici = 0, 1
                cnf, dnf, cnf3 = mod.get_TT_block_1filter(f, d + os.sep)
            files = {}
            for name in sorted(set(os.listdir(d)) - before):
                if name.endswith(".npy"):
                    continue                              # 2^n x 2^n chararray of a constant filter: not reproduced
                with open(os.path.join(d, name)) as fh:
                    files[name] = fh.read()
            out["filters"][str(f)] = {"column": tab[f, :, 0].astype(int).tolist(),
                                      "dnf": None if dnf is None else str(dnf), "cnf": None if cnf is None else str(cnf),
                                      "cnf_with_y": None if cnf3 is None else str(cnf3), "files": files}
    import sympy, pandas
    out["sympy"], out["pandas"] = sympy.__version__, pandas.__version__
    with open(os.path.join(GOLD, "ref_export_xsmall.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print(f"[export] {len(out['filters'])} filters of {blk.name}: "
          f"{sum(1 for v in out['filters'].values() if v['dnf'])} with expressions")


def gen_depth_golden(layers: int, n: int = 2, nfilter: int = 8, tfilter: int = 8, out_name: str = None):
    """tests/golden/ref_small_l<layers>.npz: --layers 3 / 4 of TT-small (a stride-1 first block,
    TT_general_imagenet_v2_small.py:178-181, :95-96) on n synthetic images -- the reference's logits
    and per-stage hashes; also asserts oracle/ttnet_float.py == reference bit for bit.
    With nfilter x tfilter != 64 (``width``): tests/golden/ref_small_p<p>.npz, another width of the same net
    (:165-167)."""
    spec = make_spec("small", nfilter, tfilter, layers)
    m = import_reference("small", layers, nfilter, tfilter)
    layout = state_dict_layout(spec)
    assert list(m.state_dict().keys()) == list(layout.keys()), "state_dict key order differs"
    st = synth.synth_state_dict(spec, calibrated=False)
    sd = OF.to_torch_state(st)
    m.load_state_dict(sd, strict=True)
    x = torch.from_numpy(synth.synth_images(n))
    ref_taps = {}
    feats = m.features
    hooks = [feats[3].register_forward_hook(lambda mod, inp, out: ref_taps.__setitem__("features.3", out.detach().clone()))]
    for i, blk in enumerate(spec.blocks):
        hooks.append(feats[4 + i].register_forward_hook(
            lambda mod, inp, out, name=blk.name: ref_taps.__setitem__(name, out.detach().clone())))
    with torch.no_grad():
        y_ref = m(x)
    for h in hooks:
        h.remove()
    taps = {}
    y = OF.forward(x, sd, spec, taps)
    assert torch.equal(y, y_ref), f"layers={layers}: oracle logits differ from the reference"
    stages = {}
    for k in ["features.3"] + [b.name for b in spec.blocks[:-1]]:
        assert torch.equal(taps[k], ref_taps[k]), (layers, k)
        stages[k] = sha(OB.pack_rows(ref_taps[k].numpy().astype(np.uint8)))
    np.savez_compressed(os.path.join(GOLD, out_name or f"ref_small_l{layers}.npz"), logits=y_ref.numpy(), n_images=n,
                        stage_names=np.array(list(stages.keys())), stage_sha=np.array(list(stages.values())),
                        n_keys=len(layout))
    print(f"[small p={nfilter * tfilter} --layers {layers}] oracle == reference on {n} images; {len(layout)} state keys; "
          f"shapes {[tuple(ref_taps[b.name].shape[1:]) for b in spec.blocks]}")


def gen_spread(variants=("small", "xsmall", "full")):
    """tests/golden/ref_spread.json: how far the REFERENCE's own float32 logits move with things that
    are not part of its definition -- its intra-op thread count and the batch split -- and how far
    they sit from the float64 evaluation of the same head on the reference's own features.  This is
    the committed justification of the parity tests' allowance next to the north star's 1e-5:
    |hip - reference| <= 1e-5 + ref_vs_exact[variant]."""
    out = {"torch": torch.__version__, "variants": {}}
    for variant in variants:
        spec = make_spec(variant, **VARIANT_ARGS[variant])
        m = import_reference(variant)
        st = synth.synth_state_dict(spec)
        m.load_state_dict(OF.to_torch_state(st), strict=True)
        n = BATCH[variant]
        x = torch.from_numpy(synth.synth_images(n, hw=spec.image_hw))
        flat = {}
        h = m.features[4 + len(spec.blocks) + 1].register_forward_hook(lambda mod, i, o: flat.__setitem__("f", o.detach().clone()))
        with torch.no_grad():
            torch.set_num_threads(8)
            y8 = m(x).clone()
            ysplit = torch.cat([m(x[: n // 2]), m(x[n // 2:])])
            m(x)
            torch.set_num_threads(1)
            y1 = m(x).clone()
            torch.set_num_threads(8)
        h.remove()
        exact = OB.head64(flat["f"].numpy(), st, f"features.{4 + len(spec.blocks) + 2}")
        with np.load(os.path.join(GOLD, f"ref_{variant}.npz")) as z:
            assert np.array_equal(z["logits"], y8.numpy()), "the committed logits are the 8-thread full-batch run"
        out["variants"][variant] = {
            "images": n,
            "logit_abs_max": float(y8.abs().max()),
            "ref_vs_exact": float(np.abs(y8.numpy() - exact).max()),
            "threads_1_vs_8": float((y1 - y8).abs().max()),
            "split_vs_full": float((ysplit - y8).abs().max()),
            "threads_1_vs_exact": float(np.abs(y1.numpy() - exact).max()),
        }
        print(f"[spread {variant}]", out["variants"][variant])
    # the other geometries with a reference capture (ref_small_l3 / _l4 / _p32.npz: uncalibrated heads, |logit| up to ~16):
    # how far the reference's own float32 logits sit from the float64 head on its own features -- what backs the scaled
    # allowance of tests/test_gpu_parity.py::_check_geometry_against_the_oracle for these fixtures
    out["geometries"] = {}
    for name, (nf, tf, layers) in {"small_l3": (8, 8, 3), "small_l4": (8, 8, 4), "small_p32": (4, 8, 1)}.items():
        spec = make_spec("small", nf, tf, layers)
        m = import_reference("small", layers, nf, tf)
        st = synth.synth_state_dict(spec, calibrated=False)
        m.load_state_dict(OF.to_torch_state(st), strict=True)
        fixture = f"ref_small_l{layers}.npz" if nf * tf == 64 else f"ref_small_p{nf * tf}.npz"
        with np.load(os.path.join(GOLD, fixture)) as z:
            want, n = z["logits"], int(z["n_images"])
        x = torch.from_numpy(synth.synth_images(n))
        flat = {}
        h = m.features[4 + len(spec.blocks) + 1].register_forward_hook(lambda mod, i, o: flat.__setitem__("f", o.detach().clone()))
        with torch.no_grad():
            y = m(x).clone()
        h.remove()
        assert np.array_equal(want, y.numpy()), f"{fixture}: the committed logits are this run"
        exact = OB.head64(flat["f"].numpy(), st, f"features.{4 + len(spec.blocks) + 2}")
        out["geometries"][name] = {"images": n, "fixture": fixture, "logit_abs_max": float(y.abs().max()),
                                   "ref_vs_exact": float(np.abs(y.numpy() - exact).max())}
        print(f"[spread {name}]", out["geometries"][name])
    with open(os.path.join(GOLD, "ref_spread.json"), "w") as f:
        json.dump(out, f, indent=1)


def gen_resize():
    """tests/golden/ref_resize.npz: what PILLOW ITSELF computes for Resize(256) + CenterCrop(224)
    (utils/preprocess.py:104-105 = torchvision calling ``Image.resize(size[::-1], BILINEAR)`` and cropping).
    Pillow is importable in the build container (12.x); torchvision is not, so its two integer rules
    (``_compute_resized_output_size`` and ``center_crop``'s offsets) are restated here and in
    oracle/pil_resize.py.  Per geometry: SHA-256 of Pillow's crop of two seeded test images
    (tests/_util.py:resize_test_images) and, for two geometries, the full crop of the first image."""
    from PIL import Image
    import PIL
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _util import RESIZE_GEOMETRIES, resize_test_images
    from oracle import pil_resize as PR
    out = {"pillow_version": np.array(PIL.__version__)}
    for (h, w) in RESIZE_GEOMETRIES:
        x = resize_test_images(2, h, w, seed=h * 1000 + w)
        nh, nw = PR.resized_size(h, w, 256)          # torchvision's size rule (restated)
        crops = []
        for i in range(2):
            im = Image.fromarray(x[i])
            if (nh, nw) != (h, w):
                im = im.resize((nw, nh), Image.BILINEAR)
            a = np.asarray(im)
            top, left = int(round((nh - 224) / 2.0)), int(round((nw - 224) / 2.0))      # torchvision center_crop
            crops.append(a[top:top + 224, left:left + 224].copy())
            assert np.array_equal(crops[-1], PR.resize_center_crop(x[i])), (h, w, "oracle/pil_resize.py != Pillow")
        out[f"sha_{h}x{w}"] = np.array([sha(c) for c in crops])
        if (h, w) in ((375, 500), (1200, 900)):
            out[f"crop_{h}x{w}"] = crops[0]
    np.savez_compressed(os.path.join(GOLD, "ref_resize.npz"), **out)
    print(f"[resize] Pillow {PIL.__version__}: {len(RESIZE_GEOMETRIES)} geometries x 2 images, "
          f"oracle/pil_resize.py byte-identical on all")


def main(variants):
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    for variant in variants:
        if variant == "valexnet":
            gen_valexnet()
            continue
        spec = make_spec(variant, **VARIANT_ARGS[variant])
        layout = state_dict_layout(spec)
        m = import_reference(variant)
        ref_sd = m.state_dict()
        assert list(ref_sd.keys()) == list(layout.keys()), "state_dict key order differs"
        for k, v in ref_sd.items():
            assert tuple(v.shape) == layout[k][0] and str(v.dtype) == "torch." + layout[k][1], k
        with open(os.path.join(GOLD, f"state_layout_{variant}.json"), "w") as f:
            json.dump({"variant": variant, "args": VARIANT_ARGS[variant],
                       "n_params": int(sum(p.numel() for p in m.parameters())),
                       "keys": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in ref_sd.items()]},
                      f, indent=0)

        calibrate_head(variant, spec)
        st = synth.synth_state_dict(spec)
        sd = OF.to_torch_state(st)
        m.load_state_dict(sd, strict=True)

        n = BATCH[variant]
        x_np = synth.synth_images(n, hw=spec.image_hw)
        x = torch.from_numpy(x_np)
        tgt = torch.from_numpy(synth.synth_targets(n))

        # --- the reference itself, with forward hooks on its own modules for the stage taps
        ref_taps = {}
        feats = m.features
        hooks = []
        def tap(name):
            return lambda mod, inp, out: ref_taps.__setitem__(name, out.detach().clone())
        hooks.append(feats[2].register_forward_hook(tap("stem.pre")))
        hooks.append(feats[3].register_forward_hook(tap("features.3")))
        for i, blk in enumerate(spec.blocks):
            hooks.append(feats[4 + i].register_forward_hook(tap(blk.name)))
            hooks.append(feats[4 + i].Block_convf.register_forward_pre_hook(
                lambda mod, inp, name=blk.name: ref_taps.__setitem__(name + ".outf", inp[0].detach().clone())))
        hooks.append(feats[4 + len(spec.blocks) + 1].register_forward_hook(tap("flatten")))
        with torch.no_grad():
            y_ref = m(x)
        for h in hooks:
            h.remove()

        # --- pin the float oracle: identical to the reference, every stage
        taps = {}
        y = OF.forward(x, sd, spec, taps)
        assert torch.equal(y, y_ref), f"{variant}: oracle logits differ from the reference"
        for k in ("stem.pre", "features.3", "flatten") + tuple(b.name for b in spec.blocks):
            assert torch.equal(taps[k], ref_taps[k]), f"{variant}: oracle stage {k} differs"
        for blk in spec.blocks:            # de-interleave the reference's convf input: channel 4c+branch
            outf = ref_taps[blk.name + ".outf"]
            for br in range(4):
                assert torch.equal(outf[:, br::4], taps[f"{blk.name}.out{br + 1}"]), (blk.name, br)
        print(f"[{variant}] oracle/ttnet_float.py == reference on {n} images (all stages, bit for bit)")

        loss = torch.nn.functional.cross_entropy(y_ref, tgt).item()
        top = y_ref.topk(5, dim=1).indices
        out = {
            "logits": y_ref.numpy(), "argmax": y_ref.argmax(1).numpy(),
            "loss": np.float64(loss), "top5_idx": top.numpy(),
            "n_images": np.int64(n),
        }
        stage_sha = {}
        bit_stages = ["features.3"] + [b.name for b in spec.blocks if not b.last]
        for blk in spec.blocks:
            bit_stages += [f"{blk.name}.out{i}" for i in (1, 2, 3, 4)]
        for k in bit_stages:
            bits = taps[k].numpy().astype(np.uint8)
            assert set(np.unique(bits).tolist()) <= {0, 1}
            rows = OB.pack_rows(bits)
            stage_sha[k] = {"shape": list(bits.shape), "rows_sha256": sha(rows),
                            "per_image_sha256": [sha(rows[i]) for i in range(n)],
                            "ones": int(bits.sum())}
            out["rows:" + k] = rows[:DUMP_IMAGES]
        out["features_flat"] = ref_taps["flatten"].numpy()[:DUMP_IMAGES]
        stage_sha["flatten"] = {"shape": list(ref_taps["flatten"].shape), "sha256": sha(ref_taps["flatten"].numpy())}
        # stem near ties (|pre| tiny): those bits are implementation-defined
        pre = ref_taps["stem.pre"].numpy()
        near = np.argwhere(np.abs(pre) < OB.NEAR_TIE)
        out["stem_near_tie_idx"] = near.astype(np.int32)
        out["stem_near_tie_pre"] = pre[tuple(near.T)].astype(np.float32)
        np.savez_compressed(os.path.join(GOLD, f"ref_{variant}.npz"), **out)

        # --- truth tables as the reference's own exporter gives them
        lut_info = {}
        if variant != "full":             # the exporter computes self.k ** 2: no tuple kernels, no n=30
            for i, blk in enumerate(spec.blocks):
                for b, modname in ((blk.conv1, "Block_conv1"), (blk.conv2, "Block_conv2"),
                                   (blk.conv3, "Block_conv3"), (blk.convf, "Block_convf")):
                    mod = getattr(feats[4 + i], modname)
                    ref_tab = reference_truth_table(mod, b)
                    mine, near_tab = OB.build_lut(st, b)
                    if b.last:
                        d = np.abs(ref_tab - mine)
                        lut_info[b.name] = {"kind": "float", "max_abs_diff_vs_f64": float(d.max()),
                                            "shape": list(ref_tab.shape)}
                    else:
                        diff = np.argwhere(ref_tab != mine)
                        assert near_tab[tuple(diff.T)].all(), f"{b.name}: table mismatch outside the near-tie set"
                        lut_info[b.name] = {"kind": "bits", "shape": list(ref_tab.shape),
                                            "ref_sha256": sha(np.packbits(ref_tab, axis=1, bitorder="little")),
                                            "f64_sha256": sha(np.packbits(mine, axis=1, bitorder="little")),
                                            "near_ties": int(near_tab.sum()),
                                            "ref_differs_from_f64_at": diff.tolist()}
                    print(f"[{variant}] {b.name}: {lut_info[b.name]['kind']} table "
                          f"{'flips vs f64: %d' % len(lut_info[b.name].get('ref_differs_from_f64_at', [])) if not b.last else 'max diff %.2e' % lut_info[b.name]['max_abs_diff_vs_f64']}")
        with open(os.path.join(GOLD, f"ref_luts_{variant}.json"), "w") as f:
            json.dump({"variant": variant, "stages": stage_sha, "luts": lut_info,
                       "loss": loss, "near_tie_threshold": OB.NEAR_TIE,
                       "torch": torch.__version__, "batch": n}, f, indent=0)
        print(f"[{variant}] logits range [{y_ref.min().item():.3f}, {y_ref.max().item():.3f}] "
              f"argmax {y_ref.argmax(1).tolist()} loss {loss:.4f}")


if __name__ == "__main__":
    if sys.argv[1:] == ["export"]:
        gen_export_golden()
    elif sys.argv[1:] == ["spread"]:
        gen_spread()
    elif sys.argv[1:] == ["resize"]:
        gen_resize()
    elif sys.argv[1:2] == ["width"]:                  # e.g. `width 4 8`: p = 32
        nf, tf = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) >= 4 else (4, 8)
        gen_depth_golden(1, nfilter=nf, tfilter=tf, out_name=f"ref_small_p{nf * tf}.npz")
    elif sys.argv[1:2] == ["depth"]:
        for L in [int(a) for a in sys.argv[2:]] or [3, 4]:
            gen_depth_golden(L)
    else:
        main(sys.argv[1:] or ["small", "xsmall", "full", "valexnet"])
        gen_export_golden()
        for L in (3, 4):
            gen_depth_golden(L)
        gen_spread()
        gen_resize()
        gen_depth_golden(1, nfilter=4, tfilter=8, out_name="ref_small_p32.npz")
